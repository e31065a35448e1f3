# A/B of the folded vector phases (PMH_NO_VEC_EPI) and of ||B u|| riding on the projector's kernels (PMH_NO_AUX_NORMG): checksums and counts must not move
R=$GRAFT_REPO_ROOT
for v in "" "PMH_NO_VEC_EPI=1 PMH_NO_AUX_NORMG=1" "PMH_NO_VEC_EPI=1" "PMH_NO_AUX_NORMG=1" "" "PMH_NO_VEC_EPI=1 PMH_NO_AUX_NORMG=1"; do
  for st in "--steps 216 --warmup 8" "--steps 20 --warmup 5"; do
  env $v python3 $R/bench.py --no-cpu-baseline --no-c2 --no-iterative $st > $R/gpurun_out/ab_fuse.json 2> $R/gpurun_out/ab_fuse.err || { tail -3 $R/gpurun_out/ab_fuse.err; continue; }
  python3 - $R/gpurun_out/ab_fuse.json "$v $st" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]; c=d["config"]["steps_by_type"]
print("[%s] %.1f it/s  %.4f ms/step  ms/apply %.4f  dense %.4f  cg %d exp %d mults %d outer %d  checksum %s  full solve %s" % (sys.argv[2], d["value"], d["ms_per_step"], c["ms_per_operator_apply"], r["avg_launch_ms"], c["cg"], c["expansion"], c["hessian_mults"], c["outer"], d["config"]["checksum"]["norm_lambda_child_after_last_step"], d["full_solve"]["solve_seconds"]))
PY
  done
done
