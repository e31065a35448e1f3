cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PMH_BENCH_NO_TIMING=1
name=c2; rx="k_spmv_stream|k_spmv_ell|k_step_update|k_dir_update"
mkdir -p $R/gpurun_out/pmc_$name
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-include-regex "$rx" --output-format csv -d $R/gpurun_out/pmc_$name/pmc_$C -- python3 $R/bench.py --workload c2 --no-cpu-baseline --steps 50 --warmup 5 > $R/gpurun_out/pmc_${name}_$C.log 2>&1
  tail -n 1 $R/gpurun_out/pmc_${name}_$C.log | cut -c1-160
done
python3 $R/scripts/pmc_parse.py $R/gpurun_out/pmc_$name "$PMH_GIT" "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, FETCH_SIZE x2) --kernel-include-regex '$rx' -- python3 bench.py --workload c2 --no-cpu-baseline --steps 50 --warmup 5" && cp $R/gpurun_out/pmc_$name/pmc_traffic.json $R/gpurun_out/r02_pmc_traffic_$name.json
rm -rf $R/gpurun_out/pmc_$name/pmc_FETCH_SIZE $R/gpurun_out/pmc_$name/pmc_WRITE_SIZE
