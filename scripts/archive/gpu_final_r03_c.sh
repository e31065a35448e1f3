# final round-3 measurements, part B2: HBM traffic (PMC, separate FETCH_SIZE / WRITE_SIZE passes) of configs[3], configs[4], configs[1], the inner-Krylov path, the general decomposition; the one-call contact solve
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
export PMH_GIT PMH_BENCH_NO_TIMING=1
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
