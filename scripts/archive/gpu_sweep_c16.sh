cd $GRAFT_REPO_ROOT
run() { python bench.py --workload c2 --no-cpu-baseline --steps 200 --warmup 20 > gpurun_out/sw.json 2>gpurun_out/sw.err || tail -n 3 gpurun_out/sw.err; python3 -c "
import json
d=json.load(open('gpurun_out/sw.json')); r=d['roofline']; print('NO_COL16=$PMH_SPMV_NO_COL16', round(d['value'],1), round(d['ms_per_step'],4), round(r['frac'],4), r.get('avg_launch_ms'), d.get('whole_iteration_GBs'))"; }
export PMH_SPMV_NO_COL16=1
run
unset PMH_SPMV_NO_COL16
run
export PMH_SPMV_NO_COL16=1
run
unset PMH_SPMV_NO_COL16
run
