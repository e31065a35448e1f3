# round-6 measurements for profiles/: rocprofv3 kernel stats of the TIMED regions (default bench at N = 1, the driver's window, the 1/8 share, configs[3], the inner-Krylov path, the
# irregular 'staircase' partition at 21^3- and 43^3-scale), of configs[4] / configs[1] / the FETI dual SpMV, and PMC passes in runs of their own (SQ counters + HBM traffic of the orbit
# GEMM from bench.py itself; HBM traffic of the other blocks).   PMH_GIT=<commit> gpurun -- bash scripts/gpu_final_r06.sh [a|n|b|c]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
PART=${1:-anbc}
export PMH_GIT
prof() { # name, args... : kernel stats of the roctx-selected timed region of bench.py
  name=$1; shift
  rocprofv3 --kernel-trace --marker-trace --stats --selected-regions --output-format csv -d $O/prof_$name -- python3 $R/bench.py "$@" --details $O/prof_${name}_details.json > $O/prof_$name.json 2> $O/prof_$name.err
  find $O/prof_$name -name "*kernel_trace.csv" -delete; find $O/prof_$name -name "*marker_api_trace.csv" -delete
  python3 $R/scripts/per_step.py $(find $O/prof_$name -name "*kernel_stats.csv" | tail -n 1) $O/prof_${name}_details.json > $O/per_step_$name.txt 2>> $O/prof_$name.err
  head -n 3 $O/per_step_$name.txt
}
if [[ $PART == *a* ]]; then
  export PMH_BENCH_ROCTX=1 PMH_BENCH_NO_TIMING=1
  prof n1 --no-cpu-baseline --no-c2 --no-iterative
  prof drv --steps 20 --warmup 5 --no-cpu-baseline --no-c2 --no-iterative
  prof sim8 --no-cpu-baseline --no-c2 --no-iterative --sim-world 8
  prof c3 --no-cpu-baseline --no-c2 --no-iterative --sub 4,4,4 --nel 21 --dense-coarse --steps 108
  prof iter --no-cpu-baseline --no-c2 --no-iterative --kplus iterative --steps 20 --warmup 5
  prof nosym21 --no-cpu-baseline --no-c2 --no-iterative --partition staircase --nel 21 --steps 108 --warmup 8
  unset PMH_BENCH_ROCTX PMH_BENCH_NO_TIMING
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_svm -- python3 $R/bench.py --workload svm --steps 60 --warmup 6 > $O/prof_svm.json 2> $O/prof_svm.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -- python3 $R/bench.py --workload c2 --steps 300 --warmup 30 --no-cpu-baseline > $O/prof_c2.json 2> $O/prof_c2.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dual -- python3 $R/scripts/dual_spmv_only.py > $O/prof_dual.json 2> $O/prof_dual.err
  find $O -name "*kernel_trace.csv" -delete
  echo "kernel stats done"
fi
if [[ $PART == *n* ]]; then # the 43^3-scale irregular partition: 184 k set-up solves (~ 4.5 min) before the traced region
  export PMH_BENCH_ROCTX=1 PMH_BENCH_NO_TIMING=1 PMH_PROGRESS=1
  prof nosym43 --no-cpu-baseline --no-c2 --no-iterative --partition staircase --nel 43 --steps 108 --warmup 8
  unset PMH_BENCH_ROCTX PMH_BENCH_NO_TIMING PMH_PROGRESS
fi
if [[ $PART == *b* ]]; then
  TAG=r06 bash $R/scripts/gpu_pmc_bench.sh > $O/pmc_bench.log 2>&1; tail -n 8 $O/pmc_bench.log
  mv $R/gpurun_out/r06_pmc_gemm_sq.txt $R/gpurun_out/r06_pmc_traffic_feti_explicit.json $O/ 2>/dev/null
fi
if [[ $PART == *c* ]]; then
  export PMH_BENCH_NO_TIMING=1
  run() { # name, program, regex, args...
    name=$1; prog=$2; rx=$3; shift; shift; shift
    if [[ -n "$PMC_ONLY" && "$PMC_ONLY" != *$name* ]]; then return; fi # PMC_ONLY="feti_iterative nosym21": a subset
    mkdir -p $O/pmc_$name
    for C in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $C --kernel-include-regex "$rx" --output-format csv -d $O/pmc_$name/pmc_$C -- python3 $R/$prog "$@" > $O/pmc_${name}_$C.log 2>&1
    done
    python3 $R/scripts/pmc_parse.py $O/pmc_$name "$PMH_GIT" "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, FETCH_SIZE x2) --kernel-include-regex '$rx' -- python3 $prog $*" && cp $O/pmc_$name/pmc_traffic.json $O/pmc_traffic_$name.json
    rm -rf $O/pmc_$name
  }
  run dual_spmv scripts/dual_spmv_only.py "k_spmv_stream|k_bsr3"
  run configs3 bench.py "k_fxo_" --no-cpu-baseline --no-c2 --no-iterative --sub 4,4,4 --nel 21 --dense-coarse --steps 40 --warmup 4
  run configs4 bench.py "k_svm" --workload svm --steps 20 --warmup 2
  run c2 bench.py "k_spmv_stream|k_spmv_ell|k_step_update|k_dir_update" --workload c2 --no-cpu-baseline --steps 50 --warmup 5
  run feti_iterative bench.py "bsr3|k_mv_spmv" --no-cpu-baseline --no-c2 --no-iterative --kplus iterative --steps 20 --warmup 2
  run general bench.py "k_fxo_" --no-cpu-baseline --no-c2 --no-iterative --young distinct --nel 43 --steps 40 --warmup 4
  run nosym21 bench.py "k_fx_symv" --no-cpu-baseline --no-c2 --no-iterative --partition staircase --nel 21 --steps 40 --warmup 4
  echo "pmc done"
fi
