# PMC passes on bench.py ITSELF (nothing between `--` and the program): the orbit GEMM of the headline path (k_fxo_gemm* / k_fxo_fin).
# Separate passes per counter group (SQ: 8 slots; FETCH_SIZE / WRITE_SIZE cannot share a pass), --kernel-trace / --stats only in their own run.
# usage (on the GPU box): PMH_GIT=<commit> TAG=r03 bash scripts/gpu_pmc_bench.sh     -> gpurun_out/${TAG}_pmc_gemm_sq.txt, gpurun_out/${TAG}_pmc_traffic_feti_explicit.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r03}
export PMH_BENCH_NO_TIMING=1
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-c2 --no-iterative"
OUT=$R/gpurun_out/${TAG}_pmc_gemm_sq.txt
echo "# rocprofv3 --pmc <set> --kernel-include-regex k_fxo_ -- python3 bench.py $ARGS   (commit ${PMH_GIT:-unknown}); averages per launch, summed over all SEs/XCDs as rocprofv3 reports them" > $OUT
i=0
for set in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc_b$i
  rocprofv3 --pmc $set --kernel-include-regex "k_fxo_" --output-format csv -d $R/gpurun_out/pmc_b$i -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_b$i.log 2>&1
  tail -n 1 $R/gpurun_out/pmc_b$i.log | cut -c1-120
  f=$(find $R/gpurun_out/pmc_b$i -name "*counter_collection.csv" | head -n 1)
  [ -n "$f" ] && python3 - "$f" >> $OUT <<'PY'
import csv,sys,collections
v=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])): v[(r["Kernel_Name"].split("(")[0][:24], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k,c),x in sorted(v.items()): print("%-26s %-28s launches %4d  avg per launch %.6g" % (k, c, len(x), sum(x)/len(x)))
PY
  rm -rf $R/gpurun_out/pmc_b$i
done
cat $OUT
# HBM traffic of the same launches
mkdir -p $R/gpurun_out/pmc_bt
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-include-regex "k_fxo_" --output-format csv -d $R/gpurun_out/pmc_bt/pmc_$C -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_bt_$C.log 2>&1
done
python3 $R/scripts/pmc_parse.py $R/gpurun_out/pmc_bt "${PMH_GIT:-unknown}" "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, FETCH_SIZE x2) --kernel-include-regex k_fxo_ -- python3 bench.py $ARGS" \
  && cp $R/gpurun_out/pmc_bt/pmc_traffic.json $R/gpurun_out/${TAG}_pmc_traffic_feti_explicit.json
rm -rf $R/gpurun_out/pmc_bt/pmc_FETCH_SIZE $R/gpurun_out/pmc_bt/pmc_WRITE_SIZE
