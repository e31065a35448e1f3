# rocprofv3 kernel stats of the configs[2] bench step (MG-preconditioned K^+); summary only
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_feti_mg -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-c2 > $GRAFT_REPO_ROOT/gpurun_out/prof_feti_mg.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/prof_feti_mg -name "*kernel_trace.csv" -delete
find $GRAFT_REPO_ROOT/gpurun_out/prof_feti_mg -name "*kernel_stats.csv" | head
