# per-kernel time of the TIMED REGION of the default bench (explicit local dual operators) at N = 1 and for the 1/8 share of an
# 8-GPU run, rocprofv3 --kernel-trace --stats restricted to the region between roctxProfilerResume / Pause (bench.py, PMH_BENCH_ROCTX)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PMH_BENCH_ROCTX=1 PMH_BENCH_NO_TIMING=1
rocprofv3 --kernel-trace --marker-trace --stats --selected-regions --output-format csv -d $R/gpurun_out/prof_ex1 -- python3 $R/bench.py --no-cpu-baseline --no-c2 --no-iterative > $R/gpurun_out/prof_ex1.json 2> $R/gpurun_out/prof_ex1.err
rocprofv3 --kernel-trace --marker-trace --stats --selected-regions --output-format csv -d $R/gpurun_out/prof_ex8 -- python3 $R/bench.py --no-cpu-baseline --no-c2 --sim-world 8 > $R/gpurun_out/prof_ex8.json 2> $R/gpurun_out/prof_ex8.err
find $R/gpurun_out/prof_ex1 $R/gpurun_out/prof_ex8 \( -name "*kernel_trace.csv" -o -name "*marker_api_trace.csv" \) -delete
find $R/gpurun_out/prof_ex1 $R/gpurun_out/prof_ex8 -name "*kernel_stats.csv"
tail -n 2 $R/gpurun_out/prof_ex1.err; tail -n 2 $R/gpurun_out/prof_ex8.err
