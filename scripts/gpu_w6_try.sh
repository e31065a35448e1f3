# A/B: 3 x 2 waves of 40 x 64 per workgroup (3 waves per SIMD) against 2 x 2 of 60 x 64; the short-piece rule at nel 63, 1/8 share
cd $GRAFT_REPO_ROOT
O=gpurun_out/kseg
mkdir -p $O
run() { name=$1; shift; args=""; while [ "$1" != "--" ]; do export "$1"; envs="$envs ${1%%=*}"; shift; done; shift
  python bench.py --no-cpu-baseline --no-c2 --no-iterative "$@" > $O/w6_$name.json 2> $O/w6_$name.err
  for e in $envs; do unset $e; done; envs=""
  python3 -c "
import json,sys
d=json.loads(open('$O/w6_$name.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$name'.ljust(24), round(d['value'],1), 'it/s dense', round(r['avg_launch_ms'],4), 'frac', round(r['frac'],3), d['config']['checksum'], d['config']['steps_by_type']['hessian_mults'])"
}
run base -- 
run waves6 PMH_FXO_WAVES6=1 --
run nel63s8_rule -- --nel 63 --steps 60 --warmup 4 --sim-world 8
run nel63s8_512 PMH_FXO_SLOTS=512 -- --nel 63 --steps 60 --warmup 4 --sim-world 8
PMH_FXO_WAVES6=1 timeout -k 10 300 python -m pytest tests/test_gpu_explicit.py -x -q -m gpu -k "orbit" 2>&1 | tail -n 3
