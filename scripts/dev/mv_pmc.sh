# Round 6: what the multi-right-hand-side product waits for -- SQ / TA / TCP / TCC counters of k_mv_spmv on ONE 43^3 block (scripts/micro/mv_spmv_time.py), one counter group per pass
# (no trace options with --pmc).   gpurun -- bash scripts/dev/mv_pmc.sh   ->  gpurun_out/dev/mv_pmc.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dev/mv_pmc; mkdir -p $O
pass() { # name, counters...
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-include-regex "k_mv_spmv" --output-format csv -d $O/$name -- python3 $R/scripts/micro/mv_spmv_time.py 42 > $O/$name.log 2>&1
  echo "pass $name done"
}
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
# (crashes rocprofv3 on this image: signal 6, the run hangs) pass ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
# pass tcp1 TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
# pass tcp2 TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum
# pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python3 - <<'PY' > $R/gpurun_out/dev/mv_pmc.txt
import csv, glob, os, collections
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/dev/mv_pmc"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        key = "fp64" if "<double, double" in k or "Idd" in k else ("fp32" if "<float, float" in k or "Iff" in k else "fp16")
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key in ("fp64", "fp32", "fp16"):
    print("== k_mv_spmv, %s entries: average per launch" % key)
    for c, v in sorted(acc[key].items()):
        print("   %-40s %16.0f   (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
cat $R/gpurun_out/dev/mv_pmc.txt
