#!/bin/bash
# A/B of the orbit GEMM on the two fp64 shapes of the matrix pipe (k_fxo_gemm4<15>: 4x4x4_4b, k_fxo_gemm16<NI>: 16x16x4): tests of the orbit storage with either kernel, then the
# headline window of the bench with each.   gpurun -- bash scripts/gpu_ab_mfma16.sh
set -o pipefail
mkdir -p gpurun_out
if [ -z "$NOTESTS" ]; then
  python -m pytest tests/test_gpu_explicit.py -x -q > gpurun_out/ab16_tests.log 2>&1 || { tail -30 gpurun_out/ab16_tests.log; exit 1; }
  tail -2 gpurun_out/ab16_tests.log
fi
B="python bench.py --steps 216 --warmup 8 --no-c2 --no-iterative --no-cpu-baseline --no-dual-spmv"
for v in ${VARIANTS:-base m16 m16_128}; do
  case $v in
    base) env="PMH_FXO_MFMA4=1" ;;
    m16) env="" ;;
    s256) env="PMH_FXO_MFMA4=1 PMH_FXO_SLOTS=256" ;;
    m16s256) env="PMH_FXO_SLOTS=256" ;;
    m16_*) env="PMH_FXO_TM=${v#m16_}" ;;
  esac
  env $env $B --details gpurun_out/ab16_$v.json > gpurun_out/ab16_$v.line 2> gpurun_out/ab16_$v.err || { tail -5 gpurun_out/ab16_$v.err; exit 1; }
  python - <<P
import json
d = json.load(open("gpurun_out/ab16_$v.json"))
r = d["roofline"]
print("$v: %.1f it/s, dense apply %.4f ms, issued %.3f GFLOP -> %.1f TFLOP/s = %.3f of peak, checksum %s, kernel %s" % (d["value"], r["avg_launch_ms"], r["flops_per_launch"] / 1e9, r["achieved"], r["frac"], d["config"]["checksum"], d["config"]["kplus"].get("orbit_gemm")))
P
done
