# SVM paired passes: tests, then configs[4] with and without the pairing
cd $GRAFT_REPO_ROOT
O=gpurun_out/svm
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_svm.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -n 40 $O/tests.log; exit 1; }
tail -n 2 $O/tests.log
python bench.py --workload svm --steps 60 --warmup 6 > $O/svm_paired.json 2> $O/svm_paired.err
PMH_SVM_NO_PAIRING=1 python bench.py --workload svm --steps 60 --warmup 6 > $O/svm_separate.json 2> $O/svm_separate.err
python3 - <<'PY'
import json
for f in ["svm_paired","svm_separate"]:
    try:
        d=json.loads(open("gpurun_out/svm/"+f+".json").read().strip().splitlines()[-1]); r=d["roofline"]
        print(f, round(d["value"],1), "it/s", round(d["ms_per_step"],3), "ms/step frac", round(r["frac"],3), d["config"]["steps_by_type"], d["config"].get("checksum"))
    except Exception as e: print(f, "FAILED", e, open("gpurun_out/svm/"+f+".err").read()[-600:])
PY
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/svm/prof -- python3 $R/bench.py --workload svm --steps 30 --warmup 4 > $R/gpurun_out/svm/prof.json 2> $R/gpurun_out/svm/prof.err
find $R/gpurun_out/svm/prof -name "*kernel_trace.csv" -delete
python3 - <<'PY'
import csv,glob,os
f=sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/svm/prof/**/*kernel_stats.csv",recursive=True))[-1]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"])>1.0: print(r["Name"][:60].ljust(62), r["Calls"], round(float(r["AverageNs"])/1e3,1), "us", r["Percentage"])
PY
