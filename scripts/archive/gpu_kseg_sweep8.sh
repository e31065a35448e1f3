# orbit GEMM plan knobs at the 1/8 share (--sim-world 8): dense (GEMM + fin) time per launch from the bench's own events
cd $GRAFT_REPO_ROOT
O=gpurun_out/kseg
mkdir -p $O
W=${W:-8}
run() { name=$1; shift; for kv in "$@"; do export "$kv"; done
  python bench.py --no-cpu-baseline --no-c2 --no-iterative --sim-world $W > $O/sw8_$name.json 2> $O/sw8_$name.err
  for kv in "$@"; do unset "${kv%%=*}"; done
  python3 -c "
import json,sys
d=json.loads(open('$O/sw8_$name.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$name'.ljust(24), round(d['value'],1), 'it/s dense', round(r['avg_launch_ms'],4), 'frac', round(r['frac'],3))"
}
run default
run slots384 PMH_FXO_SLOTS=384
run slots256 PMH_FXO_SLOTS=256
run slots192 PMH_FXO_SLOTS=192
run slots128 PMH_FXO_SLOTS=128
run aligned PMH_FXO_NO_STREAMK=1
run nokseg PMH_FXO_NO_KSEG=1
