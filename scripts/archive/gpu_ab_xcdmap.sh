# A/B of the orbit GEMM's work-item order (PMH_FXO_NO_XCDMAP) with default-policy loads of A; checksum must not move
R=$GRAFT_REPO_ROOT
for v in "" "PMH_FXO_NO_XCDMAP=1" "" "PMH_FXO_NO_XCDMAP=1"; do
  env $v python3 $R/bench.py --no-cpu-baseline --no-c2 --no-iterative --steps 108 --warmup 5 > $R/gpurun_out/ab_xcd.json 2> $R/gpurun_out/ab_xcd.err || { tail -3 $R/gpurun_out/ab_xcd.err; continue; }
  python3 - $R/gpurun_out/ab_xcd.json "$v" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("[%s] dense apply %.4f ms  frac %.3f  %.1f it/s  ms/apply %.4f  checksum %s" % (sys.argv[2], r["avg_launch_ms"], r["frac"], d["value"], d["config"]["steps_by_type"]["ms_per_operator_apply"], d["config"]["checksum"]["norm_lambda_child_after_last_step"]))
PY
done
