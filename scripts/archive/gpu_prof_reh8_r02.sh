cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_examples.py tests/test_gpu_explicit.py tests/test_gpu_configs2_full.py -x -q 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PMH_BENCH_ROCTX=1 PMH_BENCH_NO_TIMING=1
rocprofv3 --kernel-trace --marker-trace --stats --selected-regions --output-format csv -d $R/gpurun_out/prof_ex8 -- python3 $R/bench.py --no-cpu-baseline --no-c2 --sim-world 8 > $R/gpurun_out/prof_ex8.json 2> $R/gpurun_out/prof_ex8.err
find $R/gpurun_out/prof_ex8 \( -name "*kernel_trace.csv" -o -name "*marker_api_trace.csv" \) -delete
f=$(ls -t $R/gpurun_out/prof_ex8/*/*kernel_stats.csv | head -n 1)
python3 $R/scripts/per_step.py $f $R/gpurun_out/prof_ex8.json > $R/gpurun_out/per_step_reh8.txt; cut -c1-60,100-170 $R/gpurun_out/per_step_reh8.txt | head -24
