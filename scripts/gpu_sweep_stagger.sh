# orbit GEMM: start offset between the two workgroups of a CU (PMH_FXO_STAGGER x ~1000 cycles); avg_launch_ms of the dense apply from bench.py's event pairs
R=$GRAFT_REPO_ROOT
for st in ${STAGGERS:-0 1 2 3 4 6 0 2}; do
  PMH_FXO_STAGGER=$st python3 $R/bench.py --no-cpu-baseline --no-c2 --no-iterative --steps 108 --warmup 5 > $R/gpurun_out/stagger_$st.json 2> $R/gpurun_out/stagger_$st.err || { tail -3 $R/gpurun_out/stagger_$st.err; continue; }
  python3 - $R/gpurun_out/stagger_$st.json $st <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("stagger %s: dense apply %.4f ms  frac %.3f  %.1f it/s  ms/apply %.4f  checksum %s" % (sys.argv[2], r["avg_launch_ms"], r["frac"], d["value"], d["config"]["steps_by_type"]["ms_per_operator_apply"], d["config"]["checksum"]["norm_lambda_child_after_last_step"]))
PY
done
