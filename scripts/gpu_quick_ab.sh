# quick check after a change of the small dual-space kernels: tests of the paths they sit on, the default bench line, the 1/8 share, per-kernel averages of the 1/8 share
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_feti.py tests/test_gpu_explicit.py tests/test_gpu_examples.py tests/test_gpu_configs2_full.py -x -q -m gpu > gpurun_out/quick_tests.log 2>&1; tail -n 3 gpurun_out/quick_tests.log
for w in 1 8; do
  a=""; [ $w = 8 ] && a="--sim-world 8"
  python bench.py --no-cpu-baseline --no-iterative --no-c2 $a 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('N=$w', round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms/step; per application', round(d['config']['steps_by_type']['ms_per_operator_apply'],4), 'dense', round(r['avg_launch_ms'],4), d['config']['steps_by_type'])"
done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_quick8 -- python3 $R/bench.py --no-cpu-baseline --no-iterative --no-c2 --sim-world 8 > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/prof_quick8/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:16]:
    print("%-60s calls %6s avg_us %8.2f total_ms %8.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
