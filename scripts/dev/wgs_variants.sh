cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dev; mkdir -p $O
for v in 256 512 1024 2048; do
export PMH_DEV_WGS=$v PMH_BENCH_NO_TIMING=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/wv_$v -- python3 $R/bench.py --no-cpu-baseline --no-c2 --no-iterative --kplus iterative --steps 10 --warmup 2 > $O/wv_$v.json 2> $O/wv_$v.err
find $O/wv_$v -name "*kernel_trace.csv" -delete
echo "wgs $v:"; grep -E "k_mvc_" $(find $O/wv_$v -name "*kernel_stats.csv" | tail -n 1) | awk -F, '{print "   ", substr($1,1,22), $(NF-6), $(NF-4)}'
done
