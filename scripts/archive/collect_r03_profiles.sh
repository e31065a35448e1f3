# gpurun_out/r03/* (scripts/gpu_final_r03_{a,b,c}.sh) -> profiles/r03_*: bench lines, per-step kernel tables, kernel stats, PMC summaries, the rehearsal table
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/r03
P=profiles
cp $O/bench_default.json $P/r03_bench_default_1gpu.json
cp $O/bench_driver_window.json $P/r03_bench_driver_window_1gpu.json
for n in 2 4 8; do cp $O/bench_sim$n.json $P/r03_bench_rehearsal$n.json; done
for nel in 63 79; do cp $O/bench_nel${nel}_n1.json $P/r03_bench_nel${nel}_1gpu.json; cp $O/bench_nel${nel}_sim8.json $P/r03_bench_nel${nel}_rehearsal8.json; done
ks() { find $O/prof_$1 -name "*kernel_stats.csv" | tail -n 1; }
cp $(ks n1) $P/r03_kernel_stats_n1.csv
cp $(ks sim8) $P/r03_kernel_stats_rehearsal8.csv
cp $(ks c3) $P/r03_kernel_stats_configs3.csv
cp $(ks svm) $P/r03_kernel_stats_configs4.csv
cp $(ks c2) $P/r03_kernel_stats_configs1.csv
python3 scripts/per_step.py $P/r03_kernel_stats_n1.csv $O/prof_n1.json > $P/r03_per_step_kernels_n1.txt
python3 scripts/per_step.py $P/r03_kernel_stats_rehearsal8.csv $O/prof_sim8.json > $P/r03_per_step_kernels_rehearsal8.txt
python3 scripts/per_step.py $P/r03_kernel_stats_configs3.csv $O/prof_c3.json > $P/r03_per_step_kernels_configs3.txt
cp gpurun_out/r03_pmc_gemm_sq.txt $P/r03_pmc_gemm_sq.txt
cp gpurun_out/r03_pmc_traffic_feti_explicit.json $P/r03_pmc_traffic_feti_explicit.json
for n in configs3 configs4 c2 feti_iterative general; do [ -f $O/pmc_traffic_$n.json ] && cp $O/pmc_traffic_$n.json $P/r03_pmc_traffic_$n.json; done
[ -s $O/contact_solve_configs2.jsonl ] && cp $O/contact_solve_configs2.jsonl $P/r03_contact_solve_configs2.jsonl
python3 - <<'PY' > profiles/r03_strong_scaling_rehearsal.txt
import json
O="gpurun_out/r03/"
print("# Strong-scaling REHEARSALS on ONE MI355X (bench.py --sim-world N: one rank's 1/N share of the k range of the orbit GEMM, everything replicated run in full,")
print("# no collective: the all-reduce that ends B Y and the coarse-problem reductions are NOT in these numbers).  No N > 1 hardware run exists for this round.")
print("# columns: case, it/s, ms/step, ms per operator application, dense apply (GEMM + fin) ms, fraction of the fp64 matrix peak on the rank's share")
for f,lab in [("bench_default","nel 43  N=1"),("bench_sim2","nel 43  1/2"),("bench_sim4","nel 43  1/4"),("bench_sim8","nel 43  1/8"),("bench_nel63_n1","nel 63  N=1"),("bench_nel63_sim8","nel 63  1/8"),("bench_nel79_n1","nel 79  N=1"),("bench_nel79_sim8","nel 79  1/8")]:
    d=json.loads(open(O+f+".json").read().strip().splitlines()[-1]); r=d["roofline"]; c=d["config"]["steps_by_type"]
    print("%-12s %8.1f it/s  %7.3f ms/step  %7.4f ms/apply  dense %7.4f ms  frac %.3f  steps %d (%s)" % (lab, d["value"], d["ms_per_step"], c["ms_per_operator_apply"], r["avg_launch_ms"], r["frac"], d["steps"], ", ".join("%s %s" % (k, c[k]) for k in ("outer","cg","expansion","hessian_mults","operator_applies"))))
PY
cat profiles/r03_strong_scaling_rehearsal.txt
