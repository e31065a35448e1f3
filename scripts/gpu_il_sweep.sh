#!/bin/bash
# Sweep of the load / product interleaving pattern of k_fxo_gemm16 (compile-time: rebuilt on the GPU box per variant).  VARIANTS="noil 2_4 3_4" bash scripts/gpu_il_sweep.sh
set -o pipefail
B="python bench.py --steps 216 --warmup 8 --no-c2 --no-iterative --no-cpu-baseline --no-dual-spmv"
for v in ${VARIANTS:-noil 2_4 3_4 2_6 1_3}; do
  case $v in
    noil) extra="-DFXO_NO_INTERLEAVE" ;;
    *) extra="-DFXO_IL_MFMA=${v%_*} -DFXO_IL_VALU=${v#*_}" ;;
  esac
  rm -f permon_amd/csrc/fshared.o
  make -C permon_amd/csrc -j8 -s all EXTRA="$extra ${EXTRA_ALL}" > gpurun_out/il_build.log 2>&1 || { tail -5 gpurun_out/il_build.log; exit 1; }
  for slots in 512 256; do
    env PMH_FXO_SLOTS=$slots $B --details gpurun_out/il_$v.json > gpurun_out/il_$v.line 2> gpurun_out/il_$v.err || { tail -5 gpurun_out/il_$v.err; exit 1; }
    python - <<P
import json
d = json.load(open("gpurun_out/il_$v.json"))
r = d["roofline"]
print("$v slots $slots: %.1f it/s, dense apply %.4f ms, %.1f TFLOP/s = %.3f of peak, checksum %s" % (d["value"], r["avg_launch_ms"], r["achieved"], r["frac"], d["config"]["checksum"]["norm_lambda_child_after_last_step"]))
P
  done
done
