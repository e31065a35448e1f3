# A/B: 4 x 2 waves of 32 x 64 per workgroup (4 waves per SIMD, 128-row tile) against the 16-row-operand kernel at the 128-row tile and the default 120-row tile
cd $GRAFT_REPO_ROOT
O=gpurun_out/kseg
mkdir -p $O
run() { name=$1; shift; envs=""; while [ "$1" != "--" ]; do export "$1"; envs="$envs ${1%%=*}"; shift; done; shift
  python bench.py --no-cpu-baseline --no-c2 --no-iterative "$@" > $O/w8_$name.json 2> $O/w8_$name.err
  for e in $envs; do unset $e; done
  python3 -c "
import json,sys
d=json.loads(open('$O/w8_$name.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$name'.ljust(24), round(d['value'],1), 'it/s dense', round(r['avg_launch_ms'],4), 'frac', round(r['frac'],3), d['config']['checksum'], d['config']['steps_by_type']['hessian_mults'])"
}
run base -- 
run tm128 PMH_FXO_TM=128 --
run tm128_w8 PMH_FXO_TM=128 PMH_FXO_WAVES8=1 --
run tm128_w8_s1024 PMH_FXO_TM=128 PMH_FXO_WAVES8=1 PMH_FXO_SLOTS=1024 --
PMH_FXO_TM=128 PMH_FXO_WAVES8=1 timeout -k 10 300 python -m pytest tests/test_gpu_explicit.py -x -q -m gpu -k "orbit" 2>&1 | tail -n 3
