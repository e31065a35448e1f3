# final round-2 measurements: default bench at N = 1, rehearsals, the configs[3] shape, kernel profiles
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_r02_default.json 2> gpurun_out/bench_r02_default.err; tail -n 1 gpurun_out/bench_r02_default.err
for n in 2 4 8; do python bench.py --no-cpu-baseline --no-c2 --sim-world $n > gpurun_out/bench_r02_sim$n.json 2>/dev/null; echo sim $n done; done
python bench.py --no-cpu-baseline --no-c2 --no-iterative --sub 4,4,4 --nel 21 --dense-coarse > gpurun_out/bench_r02_c3.json 2>/dev/null; echo c3 done
python bench.py --no-cpu-baseline --no-c2 --sub 4,4,4 --nel 21 --dense-coarse --sim-world 8 > gpurun_out/bench_r02_c3_sim8.json 2>/dev/null; echo c3 sim8 done
python3 - <<'PY'
import json
for f in ["default", "sim2", "sim4", "sim8", "c3", "c3_sim8"]:
    d = json.load(open("gpurun_out/bench_r02_%s.json" % f)); r = d["roofline"]
    print(f, round(d["value"], 1), round(d["ms_per_step"], 3), round(r["frac"], 3), round(r["avg_launch_ms"], 4), d["config"]["kplus"]["storage"], d["config"]["kplus"]["assemble_seconds"], d["config"]["steps_by_type"])
d = json.load(open("gpurun_out/bench_r02_default.json"))
print(d["iterative"]["value"], d["strict_fp64"]["value"], d["cpu_baseline"]["value"], d["configs1"]["value"], d["configs1"]["roofline"]["frac"], d["roofline"]["traffic"])
PY
bash scripts/gpu_prof_r02.sh
