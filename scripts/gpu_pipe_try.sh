# A/B: software-pipelined operand reads in the orbit GEMM's inner block (PMH_FXO_PIPE=1)
cd $GRAFT_REPO_ROOT
O=gpurun_out/kseg
mkdir -p $O
run() { name=$1; shift; envs=""; while [ "$1" != "--" ]; do export "$1"; envs="$envs ${1%%=*}"; shift; done; shift
  python bench.py --no-cpu-baseline --no-c2 --no-iterative "$@" > $O/pipe_$name.json 2> $O/pipe_$name.err
  for e in $envs; do unset $e; done
  python3 -c "
import json,sys
d=json.loads(open('$O/pipe_$name.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$name'.ljust(24), round(d['value'],1), 'it/s dense', round(r['avg_launch_ms'],4), 'frac', round(r['frac'],3), d['config']['checksum'], d['config']['steps_by_type']['hessian_mults'])"
}
run base -- 
run pipe PMH_FXO_PIPE=1 --
run base2 -- 
run pipe2 PMH_FXO_PIPE=1 --
