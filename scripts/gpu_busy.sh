# GPU-busy time per bench step: kernel-time difference of a 22-step and a 12-step run (same set-up)
cd /tmp && export TMPDIR=/tmp
for n in 12 22; do
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/busy_$n -- python3 $GRAFT_REPO_ROOT/bench.py --steps $n --warmup 2 --no-cpu-baseline --no-c2 > $GRAFT_REPO_ROOT/gpurun_out/busy_$n.json 2> $GRAFT_REPO_ROOT/gpurun_out/busy_$n.err
find $GRAFT_REPO_ROOT/gpurun_out/busy_$n -name "*kernel_trace.csv" -delete
done
