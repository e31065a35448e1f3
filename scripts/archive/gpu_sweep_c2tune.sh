cd $GRAFT_REPO_ROOT
run() { python bench.py --workload c2 --no-cpu-baseline --steps 200 --warmup 20 > gpurun_out/sw.json 2>gpurun_out/sw.err || tail -n 3 gpurun_out/sw.err; python3 -c "
import json
d=json.load(open('gpurun_out/sw.json')); r=d['roofline']; print('TUNE=$PMH_SPMV_TUNE', round(d['value'],1), round(d['ms_per_step'],4), round(r['frac'],4), r.get('avg_launch_ms'))"; }
run
for t in 1024,2,3 2048,2,1 2048,2,3 512,2,1 1024,1,1; do export PMH_SPMV_TUNE=$t; run; done
