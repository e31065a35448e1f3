# HBM traffic per launch of the roofline kernels from PMC counters (separate --pmc passes, FETCH_SIZE doubled: MI355X_MICROARCH.md).
# Counter collection is restricted to the kernel of interest with --kernel-include-regex (the set-up solves of the explicit operators
# alone are ~11 M launches of OTHER kernels; they run uncounted).  PMH_GIT = the commit being measured (the box has no .git); writes
# profiles-ready JSON with a _meta record into gpurun_out/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PMH_BENCH_NO_TIMING=1
run() { # name, regex, bench args...
  name=$1; rx=$2; shift; shift
  mkdir -p $R/gpurun_out/pmc_$name
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-include-regex "$rx" --output-format csv -d $R/gpurun_out/pmc_$name/pmc_$C -- python3 $R/bench.py "$@" > $R/gpurun_out/pmc_${name}_$C.log 2>&1
    tail -n 1 $R/gpurun_out/pmc_${name}_$C.log | cut -c1-160
  done
  python3 $R/scripts/pmc_parse.py $R/gpurun_out/pmc_$name "$PMH_GIT" "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, FETCH_SIZE x2) --kernel-include-regex '$rx' -- python3 bench.py $*" \
    && cp $R/gpurun_out/pmc_$name/pmc_traffic.json $R/gpurun_out/r02_pmc_traffic_$name.json
  rm -rf $R/gpurun_out/pmc_$name/pmc_FETCH_SIZE $R/gpurun_out/pmc_$name/pmc_WRITE_SIZE
}
# the dense apply of the explicit operators: same kernels, same storage and sizes as configs[2], filled with a byte pattern instead of
# 51 s of set-up solves (scripts/symv_tune.py) -- counters on every launch of a 30-apply loop
runtune() {
  name=feti_explicit
  mkdir -p $R/gpurun_out/pmc_$name
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-include-regex "k_fx_symv" --output-format csv -d $R/gpurun_out/pmc_$name/pmc_$C/sym -- python3 $R/scripts/symv_tune.py sym 43 8 > $R/gpurun_out/pmc_${name}_$C.log 2>&1
    rocprofv3 --pmc $C --kernel-include-regex "k_fxs_" --output-format csv -d $R/gpurun_out/pmc_$name/pmc_$C/class -- python3 $R/scripts/symv_tune.py class 43 8 >> $R/gpurun_out/pmc_${name}_$C.log 2>&1
    rocprofv3 --pmc $C --kernel-include-regex "k_fxs_sym" --output-format csv -d $R/gpurun_out/pmc_$name/pmc_$C/class_sym -- python3 $R/scripts/symv_tune.py class_sym 43 8 >> $R/gpurun_out/pmc_${name}_$C.log 2>&1
    rocprofv3 --pmc $C --kernel-include-regex "k_fxo_" --output-format csv -d $R/gpurun_out/pmc_$name/pmc_$C/class_orbit -- python3 $R/scripts/symv_tune.py class_orbit 43 8 >> $R/gpurun_out/pmc_${name}_$C.log 2>&1
    tail -n 1 $R/gpurun_out/pmc_${name}_$C.log | cut -c1-160
  done
  python3 $R/scripts/pmc_parse.py $R/gpurun_out/pmc_$name "$PMH_GIT" "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, FETCH_SIZE x2) --kernel-include-regex k_fx_symv|k_fxs_|k_fxo_ -- python3 scripts/symv_tune.py sym|class|class_sym|class_orbit 43 8 (the configs[2] operators, byte-pattern fill)" \
    && cp $R/gpurun_out/pmc_$name/pmc_traffic.json $R/gpurun_out/r02_pmc_traffic_$name.json
  rm -rf $R/gpurun_out/pmc_$name/pmc_FETCH_SIZE $R/gpurun_out/pmc_$name/pmc_WRITE_SIZE
}
runtune
echo "explicit done"
[ -n "$PMH_PMC_ONLY_EXPLICIT" ] && exit 0
run feti_iterative "bsr3" --no-cpu-baseline --no-c2 --no-iterative --kplus iterative --steps 20 --warmup 2
echo "iterative done"
run c2 "k_spmv_stream|k_spmv_ell|k_step_update|k_dir_update" --workload c2 --no-cpu-baseline --steps 50 --warmup 5
