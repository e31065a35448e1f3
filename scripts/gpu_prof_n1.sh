# kernel stats per step of the default N=1 workload: difference of a 22- and a 12-step run = 10 steps
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in 12 22; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/n1_$n -- python3 $R/bench.py --steps $n --warmup 2 --no-cpu-baseline --no-c2 > $R/gpurun_out/n1_$n.json 2> $R/gpurun_out/n1_$n.err
find $R/gpurun_out/n1_$n -name "*kernel_trace.csv" -delete
done
