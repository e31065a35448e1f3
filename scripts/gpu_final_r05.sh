# round-5 measurements for profiles/: rocprofv3 kernel stats (timed region of the default bench at N = 1 and for the 1/8 share; configs[3]; configs[4]; configs[1]), PMC passes in runs of
# their own (SQ counters and HBM traffic of the orbit GEMM from bench.py itself; HBM traffic of the FETI dual SpMV block, of configs[3], configs[4], configs[1], the inner-Krylov path, the
# general decomposition).   PMH_GIT=<commit> gpurun -- bash scripts/gpu_final_r05.sh [a|b|c]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
PART=${1:-abc}
export PMH_GIT
if [[ $PART == *a* ]]; then
  export PMH_BENCH_ROCTX=1 PMH_BENCH_NO_TIMING=1
  rocprofv3 --kernel-trace --marker-trace --stats --selected-regions --output-format csv -d $O/prof_n1 -- python3 $R/bench.py --no-cpu-baseline --no-c2 --no-iterative > $O/prof_n1.json 2> $O/prof_n1.err
  rocprofv3 --kernel-trace --marker-trace --stats --selected-regions --output-format csv -d $O/prof_drv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-c2 --no-iterative > $O/prof_drv.json 2> $O/prof_drv.err
  rocprofv3 --kernel-trace --marker-trace --stats --selected-regions --output-format csv -d $O/prof_sim8 -- python3 $R/bench.py --no-cpu-baseline --no-c2 --no-iterative --sim-world 8 > $O/prof_sim8.json 2> $O/prof_sim8.err
  rocprofv3 --kernel-trace --marker-trace --stats --selected-regions --output-format csv -d $O/prof_c3 -- python3 $R/bench.py --no-cpu-baseline --no-c2 --no-iterative --sub 4,4,4 --nel 21 --dense-coarse --steps 108 > $O/prof_c3.json 2> $O/prof_c3.err
  unset PMH_BENCH_ROCTX PMH_BENCH_NO_TIMING
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_svm -- python3 $R/bench.py --workload svm --steps 60 --warmup 6 > $O/prof_svm.json 2> $O/prof_svm.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -- python3 $R/bench.py --workload c2 --steps 300 --warmup 30 --no-cpu-baseline > $O/prof_c2.json 2> $O/prof_c2.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dual -- python3 $R/scripts/dual_spmv_only.py > $O/prof_dual.json 2> $O/prof_dual.err
  find $O -name "*kernel_trace.csv" -delete; find $O -name "*marker_api_trace.csv" -delete
  for n in n1 drv sim8 c3; do python3 $R/scripts/per_step.py $(find $O/prof_$n -name "*kernel_stats.csv" | tail -n 1) $O/prof_$n.json > $O/per_step_$n.txt; done
  head -n 4 $O/per_step_n1.txt; head -n 3 $O/per_step_sim8.txt
  echo "kernel stats done"
fi
if [[ $PART == *b* ]]; then
  TAG=r05 bash $R/scripts/gpu_pmc_bench.sh > $O/pmc_bench.log 2>&1; tail -n 8 $O/pmc_bench.log
  mv $R/gpurun_out/r05_pmc_gemm_sq.txt $R/gpurun_out/r05_pmc_traffic_feti_explicit.json $O/ 2>/dev/null
fi
if [[ $PART == *c* ]]; then
  export PMH_BENCH_NO_TIMING=1
  run() { # name, program, regex, args...
    name=$1; prog=$2; rx=$3; shift; shift; shift
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
  run feti_iterative bench.py "bsr3" --no-cpu-baseline --no-c2 --no-iterative --kplus iterative --steps 20 --warmup 2
  run general bench.py "k_fxo_" --no-cpu-baseline --no-c2 --no-iterative --young distinct --nel 43 --steps 40 --warmup 4
  echo "pmc done"
fi
