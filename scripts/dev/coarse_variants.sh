cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dev; mkdir -p $O
for v in 0 1 2 3 4 5; do
export PMH_DEV_COARSE=$v PMH_BENCH_NO_TIMING=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cv_$v -- python3 $R/bench.py --no-cpu-baseline --no-c2 --no-iterative --kplus iterative --steps 10 --warmup 2 > $O/cv_$v.json 2> $O/cv_$v.err
find $O/cv_$v -name "*kernel_trace.csv" -delete
echo "variant $v: $(grep -E "coarse_mfma|rt_dot|k_mvc_project" $(find $O/cv_$v -name "*kernel_stats.csv" | tail -n 1) | cut -d, -f1-4 | cut -c1-40,60- | tr "\n" " ")"
done
