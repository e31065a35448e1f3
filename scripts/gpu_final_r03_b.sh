# final round-3 measurements, part B: rocprofv3 kernel stats (timed region of the default bench at N = 1 and for the 1/8 share; configs[3]; configs[4]; configs[1]),
# PMC passes (separate runs: SQ counters and HBM traffic of the orbit GEMM from bench.py itself; traffic of configs[3], configs[4], configs[1], the inner-Krylov path), the one-call contact solve
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
export PMH_BENCH_ROCTX=1 PMH_BENCH_NO_TIMING=1
rocprofv3 --kernel-trace --marker-trace --stats --selected-regions --output-format csv -d $O/prof_n1 -- python3 $R/bench.py --no-cpu-baseline --no-c2 --no-iterative > $O/prof_n1.json 2> $O/prof_n1.err
rocprofv3 --kernel-trace --marker-trace --stats --selected-regions --output-format csv -d $O/prof_sim8 -- python3 $R/bench.py --no-cpu-baseline --no-c2 --no-iterative --sim-world 8 > $O/prof_sim8.json 2> $O/prof_sim8.err
rocprofv3 --kernel-trace --marker-trace --stats --selected-regions --output-format csv -d $O/prof_c3 -- python3 $R/bench.py --no-cpu-baseline --no-c2 --no-iterative --sub 4,4,4 --nel 21 --dense-coarse --steps 108 > $O/prof_c3.json 2> $O/prof_c3.err
unset PMH_BENCH_ROCTX
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_svm -- python3 $R/bench.py --workload svm --steps 60 --warmup 6 > $O/prof_svm.json 2> $O/prof_svm.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -- python3 $R/bench.py --workload c2 --steps 300 --warmup 30 --no-cpu-baseline > $O/prof_c2.json 2> $O/prof_c2.err
find $O -name "*kernel_trace.csv" -delete; find $O -name "*marker_api_trace.csv" -delete
find $O -name "*kernel_stats.csv"
echo "kernel stats done"
export PMH_GIT TAG=r03
bash $R/scripts/gpu_pmc_bench.sh > $O/pmc_bench.log 2>&1; tail -n 6 $O/pmc_bench.log
cd /tmp
run() { # name, regex, bench args...
  name=$1; rx=$2; shift; shift
  mkdir -p $O/pmc_$name
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-include-regex "$rx" --output-format csv -d $O/pmc_$name/pmc_$C -- python3 $R/bench.py "$@" > $O/pmc_${name}_$C.log 2>&1
  done
  python3 $R/scripts/pmc_parse.py $O/pmc_$name "$PMH_GIT" "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, FETCH_SIZE x2) --kernel-include-regex '$rx' -- python3 bench.py $*" && cp $O/pmc_$name/pmc_traffic.json $O/pmc_traffic_$name.json
  rm -rf $O/pmc_$name
}
run configs3 "k_fxo_" --no-cpu-baseline --no-c2 --no-iterative --sub 4,4,4 --nel 21 --dense-coarse --steps 40 --warmup 4
run configs4 "k_svm" --workload svm --steps 20 --warmup 2
run c2 "k_spmv_stream|k_spmv_ell|k_step_update|k_dir_update" --workload c2 --no-cpu-baseline --steps 50 --warmup 5
run feti_iterative "bsr3" --no-cpu-baseline --no-c2 --no-iterative --kplus iterative --steps 20 --warmup 2
run general "k_fx_symv" --no-cpu-baseline --no-c2 --no-iterative --young distinct --nel 21 --steps 40 --warmup 4
echo "pmc done"
cd $R && PMH_CONTACT_TIMING=1 python scripts/contact_solve_c2.py 43 all > $O/contact_solve_configs2.jsonl 2> $O/contact_solve_configs2.err; cat $O/contact_solve_configs2.jsonl | cut -c1-300
