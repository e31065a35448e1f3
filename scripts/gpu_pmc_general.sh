cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O/pmc_general
export PMH_BENCH_NO_TIMING=1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-include-regex "k_fxo_" --output-format csv -d $O/pmc_general/pmc_$C -- python3 $R/bench.py --no-cpu-baseline --no-c2 --no-iterative --young distinct --nel 43 --steps 40 --warmup 4 > $O/pmc_general_$C.log 2>&1
done
python3 $R/scripts/pmc_parse.py $O/pmc_general "$PMH_GIT" "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, FETCH_SIZE x2) --kernel-include-regex 'k_fxo_' -- python3 bench.py --no-cpu-baseline --no-c2 --no-iterative --young distinct --nel 43 --steps 40 --warmup 4" && cp $O/pmc_general/pmc_traffic.json $O/pmc_traffic_general.json
rm -rf $O/pmc_general
