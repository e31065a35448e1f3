# kernel durations of the orbit GEMM and its fin kernel with and without k segments
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/kseg
mkdir -p $O
export PMH_BENCH_NO_TIMING=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kseg -- python3 $R/bench.py --no-cpu-baseline --no-c2 --no-iterative --steps 100 > $O/prof_kseg.json 2> $O/prof_kseg.err
export PMH_FXO_NO_KSEG=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_nokseg -- python3 $R/bench.py --no-cpu-baseline --no-c2 --no-iterative --steps 100 > $O/prof_nokseg.json 2> $O/prof_nokseg.err
find $O -name "*kernel_trace.csv" -delete
for d in prof_kseg prof_nokseg; do echo == $d; f=$(find $O/$d -name "*kernel_stats.csv"); grep -E "k_fxo_gemm|k_fxo_fin|k_rows_then|k_gt_fused" $f | cut -d, -f1-5 | cut -c1-200; done
