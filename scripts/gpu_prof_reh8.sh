# kernel stats of the 1-block-per-GPU rehearsal (rank 0's share of an 8-GPU run): difference of a 22- and a 12-step run = 10 steps
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in 12 22; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/reh8_$n -- python3 $R/bench.py --sim-world 8 --steps $n --warmup 2 --no-cpu-baseline --no-c2 > $R/gpurun_out/reh8_$n.json 2> $R/gpurun_out/reh8_$n.err
find $R/gpurun_out/reh8_$n -name "*kernel_trace.csv" -delete
done
