# per-step kernel tables of the timed region (roctx-selected) of the default bench at N = 1 (and, with SIM8=1, for the 1/8 share)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
export PMH_BENCH_ROCTX=1 PMH_BENCH_NO_TIMING=1
FLAGS="--no-cpu-baseline --no-c2 --no-iterative --no-configs3 --no-svm --no-contact-solve --no-dual-spmv"
rocprofv3 --kernel-trace --marker-trace --stats --selected-regions --output-format csv -d $O/prof_n1 -- python3 $R/bench.py $FLAGS > $O/prof_n1.json 2> $O/prof_n1.err
if [ -n "$SIM8" ]; then rocprofv3 --kernel-trace --marker-trace --stats --selected-regions --output-format csv -d $O/prof_sim8 -- python3 $R/bench.py $FLAGS --sim-world 8 > $O/prof_sim8.json 2> $O/prof_sim8.err; fi
find $O -name "*kernel_trace.csv" -delete; find $O -name "*marker_api_trace.csv" -delete
for n in n1 sim8; do [ -d $O/prof_$n ] && python3 $R/scripts/per_step.py $(find $O/prof_$n -name "*kernel_stats.csv" | tail -n 1) $O/prof_$n.json > $O/per_step_$n.txt; done
head -n 30 $O/per_step_n1.txt; [ -f $O/per_step_sim8.txt ] && head -n 30 $O/per_step_sim8.txt
