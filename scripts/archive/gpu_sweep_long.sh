cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-c2 --sim-world 8 > gpurun_out/sw.json 2>/dev/null; python3 -c "
import json
d=json.load(open('gpurun_out/sw.json')); print('$1', round(d['ms_per_step'],4), round(d['config']['steps_by_type']['ms_per_operator_apply'],4), d['config']['steps_by_type']['cg'], d['config']['steps_by_type']['expansion'])"; }
run base
PMH_LONG_NT=0 run nt0
PMH_LONG_NT=0 PMH_LONG_CHUNK=2048 run nt0_2048
PMH_LONG_NT=0 PMH_LONG_CHUNK=1024 run nt0_1024
PMH_LONG_NT=1 PMH_LONG_CHUNK=1024 run nt1_1024
