# final round-3 measurements, part B1: rocprofv3 kernel stats (timed region of the default bench at N = 1 and for the 1/8 share; configs[3]; configs[4]; configs[1]),
# PMC passes (separate runs: SQ counters and HBM traffic of the orbit GEMM from bench.py itself; traffic of configs[3], configs[4], configs[1], the inner-Krylov path), the one-call contact solve
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
export PMH_BENCH_ROCTX=1 PMH_BENCH_NO_TIMING=1
rocprofv3 --kernel-trace --marker-trace --stats --selected-regions --output-format csv -d $O/prof_n1 -- python3 $R/bench.py --no-cpu-baseline --no-c2 --no-iterative > $O/prof_n1.json 2> $O/prof_n1.err
rocprofv3 --kernel-trace --marker-trace --stats --selected-regions --output-format csv -d $O/prof_sim8 -- python3 $R/bench.py --no-cpu-baseline --no-c2 --no-iterative --sim-world 8 > $O/prof_sim8.json 2> $O/prof_sim8.err
rocprofv3 --kernel-trace --marker-trace --stats --selected-regions --output-format csv -d $O/prof_c3 -- python3 $R/bench.py --no-cpu-baseline --no-c2 --no-iterative --sub 4,4,4 --nel 21 --dense-coarse --steps 108 > $O/prof_c3.json 2> $O/prof_c3.err
unset PMH_BENCH_ROCTX
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_svm -- python3 $R/bench.py --workload svm --steps 60 --warmup 6 > $O/prof_svm.json 2> $O/prof_svm.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -- python3 $R/bench.py --workload c2 --steps 300 --warmup 30 --no-cpu-baseline > $O/prof_c2.json 2> $O/prof_c2.err
find $O -name "*kernel_trace.csv" -delete; find $O -name "*marker_api_trace.csv" -delete
find $O -name "*kernel_stats.csv"
echo "kernel stats done"
export PMH_GIT TAG=r03
bash $R/scripts/gpu_pmc_bench.sh > $O/pmc_bench.log 2>&1; tail -n 6 $O/pmc_bench.log
