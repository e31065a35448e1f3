# per-kernel table of the inner-Krylov bench at another block size: gpurun -- bash scripts/dev/prof_iter_nel.sh <nel>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dev; mkdir -p $O; name=iter_nel$1
export PMH_BENCH_ROCTX=1 PMH_BENCH_NO_TIMING=1
rocprofv3 --kernel-trace --marker-trace --stats --selected-regions --output-format csv -d $O/prof_$name -- python3 $R/bench.py --no-cpu-baseline --no-c2 --no-iterative --kplus iterative --nel $1 --steps 10 --warmup 2 --details $O/prof_${name}_details.json > $O/prof_$name.json 2> $O/prof_$name.err
find $O/prof_$name -name "*kernel_trace.csv" -delete; find $O/prof_$name -name "*marker_api_trace.csv" -delete
python3 $R/scripts/per_step.py $(ls -t $(find $O/prof_$name -name "*kernel_stats.csv") | head -n 1) $O/prof_${name}_details.json > $O/per_step_$name.txt 2>> $O/prof_$name.err
head -n 24 $O/per_step_$name.txt | cut -c1-60,100-200
