# one GPU round: parity tests, smoke, the default bench line, rocprof kernel stats of both workloads
set -x
python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke | tail -2
mkdir -p gpurun_out
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -2 gpurun_out/bench_default.err; cut -c1-600 gpurun_out/bench_default.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_c2 -- python3 $GRAFT_REPO_ROOT/bench.py --workload c2 --steps 100 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_c2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_feti -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-c2 > $GRAFT_REPO_ROOT/gpurun_out/prof_feti.log 2>&1
# keep only the stats summaries (kernel traces are large)
find $GRAFT_REPO_ROOT/gpurun_out/prof_c2 $GRAFT_REPO_ROOT/gpurun_out/prof_feti -name "*kernel_trace.csv" -delete
find $GRAFT_REPO_ROOT/gpurun_out -name "*kernel_stats.csv" | head
