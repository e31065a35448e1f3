cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-c2 "$@" > gpurun_out/sw.json 2>gpurun_out/sw.err || tail -n 3 gpurun_out/sw.err; python3 -c "
import json
d=json.load(open('gpurun_out/sw.json')); c=d['config']['steps_by_type']; print('$PMH_SYNC_SPIN $*', round(d['value'],1), round(d['ms_per_step'],4), round(c['ms_per_operator_apply'],4), c['outer'], c['cg'], c['expansion'], c['hessian_mults'])"; }
export PMH_SYNC_SPIN=0
run --sim-world 8
export PMH_SYNC_SPIN=1
run --sim-world 8
export PMH_SYNC_SPIN=0
run --sim-world 8
export PMH_SYNC_SPIN=1
run --sim-world 8
run --no-iterative
