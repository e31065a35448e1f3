set -x
python -m pytest tests -x -q -m gpu 2>&1 | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
mkdir -p gpurun_out
python bench.py --steps 300 --warmup 30 > gpurun_out/bench_r01_a.json 2> gpurun_out/bench_r01_a.err; cat gpurun_out/bench_r01_a.json; tail -3 gpurun_out/bench_r01_a.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_a -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_a.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/prof_a.log
find $GRAFT_REPO_ROOT/gpurun_out/prof_a -name "*stats*" | head
