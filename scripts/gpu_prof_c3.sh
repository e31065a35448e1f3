# kernel statistics of the configs[3] shape (64 subdomains of 21^3 elements, dense 384 x 384 coarse problem) on one GPU
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3 -- python3 $R/bench.py --no-cpu-baseline --no-c2 --no-iterative --sub 4,4,4 --nel 21 --dense-coarse --steps 648 > $R/gpurun_out/prof_c3.json 2>/dev/null
python3 - <<PY
import csv,glob,json
f=glob.glob("$R/gpurun_out/prof_c3/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
d=json.load(open("$R/gpurun_out/prof_c3.json")); print(d["value"], d["ms_per_step"], d["config"]["steps_by_type"])
for r in rows[:40]:
    if int(r["Calls"]) >= 300: print("%-72s calls %6s avg_us %8.2f total_ms %8.2f" % (r["Name"][:72], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
