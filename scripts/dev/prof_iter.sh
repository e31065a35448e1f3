# kernel stats of the timed region of the inner-Krylov bench (dev helper): gpurun -- bash scripts/dev/prof_iter.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dev; mkdir -p $O; name=iter_${1:-x}
export PMH_BENCH_ROCTX=1 PMH_BENCH_NO_TIMING=1
rocprofv3 --kernel-trace --marker-trace --stats --selected-regions --output-format csv -d $O/prof_$name -- python3 $R/bench.py --no-cpu-baseline --no-c2 --no-iterative --kplus iterative --steps 20 --warmup 5 --details $O/prof_${name}_details.json > $O/prof_$name.json 2> $O/prof_$name.err
find $O/prof_$name -name "*kernel_trace.csv" -delete; find $O/prof_$name -name "*marker_api_trace.csv" -delete
python3 $R/scripts/per_step.py $(find $O/prof_$name -name "*kernel_stats.csv" | tail -n 1) $O/prof_${name}_details.json > $O/per_step_$name.txt 2>> $O/prof_$name.err
head -n 30 $O/per_step_$name.txt | cut -c1-60,100-200
