# orbit GEMM plan knobs: dense (GEMM + fin) time per launch from the bench's own events
cd $GRAFT_REPO_ROOT
O=gpurun_out/kseg
mkdir -p $O
run() { name=$1; shift; for kv in "$@"; do export "$kv"; done
  python bench.py --no-cpu-baseline --no-c2 --no-iterative --steps 100 > $O/sw_$name.json 2> $O/sw_$name.err
  for kv in "$@"; do unset "${kv%%=*}"; done
  python3 -c "
import json,sys
d=json.loads(open('$O/sw_$name.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$name'.ljust(24), round(d['value'],1), 'it/s dense', round(r['avg_launch_ms'],4), 'frac', round(r['frac'],3))"
}
run default
run nokseg PMH_FXO_NO_KSEG=1
run nokseg_aligned PMH_FXO_NO_KSEG=1 PMH_FXO_NO_STREAMK=1
run aligned PMH_FXO_NO_STREAMK=1
run slots480 PMH_FXO_SLOTS=480
run slots448 PMH_FXO_SLOTS=448
run slots384 PMH_FXO_SLOTS=384
run slots256 PMH_FXO_SLOTS=256
run segmin1764 PMH_FXO_SEGMIN=1700
run segmin100 PMH_FXO_SEGMIN=100
