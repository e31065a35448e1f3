# strong-scaling rehearsal on ONE GPU: rank 0's share of an N-GPU run (8/N blocks, no collective), plain timing + kernel stats at N=8
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for n in 1 2 4 8; do
  python3 $R/bench.py --sim-world $n --steps 20 --warmup 2 --no-cpu-baseline --no-c2 > $R/gpurun_out/reh_$n.json 2> $R/gpurun_out/reh_$n.err
  python3 -c "import json;d=json.load(open('$R/gpurun_out/reh_$n.json'));print('sim-world',$n,'ms/step',d['ms_per_step'])"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_reh8 -- python3 $R/bench.py --sim-world 8 --steps 20 --warmup 2 --no-cpu-baseline --no-c2 > $R/gpurun_out/prof_reh8.log 2>&1
find $R/gpurun_out/prof_reh8 -name "*kernel_trace.csv" -delete
find $R/gpurun_out/prof_reh8 -name "*kernel_stats.csv"
