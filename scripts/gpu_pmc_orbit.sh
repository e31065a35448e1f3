cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VALU"; do
  rm -rf $R/gpurun_out/pmc_og
  rocprofv3 --pmc $set --kernel-include-regex "k_gemm" --output-format csv -d $R/gpurun_out/pmc_og -- $R/scripts/micro/bin/og_2_2_16 715 48 33288 28 > /dev/null 2>&1
  f=$(find $R/gpurun_out/pmc_og -name "*counter_collection.csv" | head -n 1)
  python3 - "$f" <<'PY'
import csv,sys,collections
v=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])): v[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,x in v.items(): print("%-28s avg per launch %.4g" % (k, sum(x)/len(x)))
PY
done
rm -rf $R/gpurun_out/pmc_og
