# split-K of the orbit GEMM at a 1/8 and a 1/4 share of the k range (bench.py --sim-world N): time per operator application
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for N in 8 4; do
  for S in 28 20 14 9; do
    PMH_FXO_SPLIT=$S PMH_FXO_MINCH=4 timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-c2 --no-iterative --sim-world $N > $R/gpurun_out/split_${N}_$S.json 2>/dev/null
    python3 -c "
import json,sys
d=json.load(open('$R/gpurun_out/split_${N}_$S.json')); r=d['roofline']; print('N=$N S=$S', round(d['ms_per_step'],4), 'ms/apply', round(d['config']['steps_by_type']['ms_per_operator_apply'],4), 'dense ms', round(r['avg_launch_ms'],4))"
  done
done
