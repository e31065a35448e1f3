# final round-3 measurements, part A: the default bench line (every block), the driver's window, the strong-scaling rehearsals
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -n 1 $O/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_window.json 2> $O/bench_driver_window.err; tail -n 1 $O/bench_driver_window.err
for n in 2 4 8; do python bench.py --no-cpu-baseline --no-c2 --no-iterative --sim-world $n > $O/bench_sim$n.json 2>/dev/null; echo sim $n done; done
for nel in 63 79; do
  python bench.py --no-cpu-baseline --no-c2 --no-iterative --nel $nel --steps 60 --warmup 4 > $O/bench_nel${nel}_n1.json 2> $O/bench_nel${nel}_n1.err; echo nel $nel N=1 done
  python bench.py --no-cpu-baseline --no-c2 --no-iterative --nel $nel --steps 60 --warmup 4 --sim-world 8 > $O/bench_nel${nel}_sim8.json 2> $O/bench_nel${nel}_sim8.err; echo nel $nel sim8 done
done
python3 - <<'PY'
import json
O="gpurun_out/r03/"
for f in ["bench_default","bench_driver_window","bench_sim2","bench_sim4","bench_sim8","bench_nel63_n1","bench_nel63_sim8","bench_nel79_n1","bench_nel79_sim8"]:
    try:
        d=json.loads(open(O+f+".json").read().strip().splitlines()[-1]); r=d["roofline"]; c=d["config"]["steps_by_type"]
        print(f, round(d["value"],1), "it/s", round(d["ms_per_step"],3), "ms/step; ms/apply", round(c["ms_per_operator_apply"],4), "dense", round(r["avg_launch_ms"],4), "frac", round(r["frac"],3), d["config"]["kplus"]["storage"], "setup", d["config"]["setup_seconds"])
    except Exception as e: print(f, "FAILED", e)
PY
