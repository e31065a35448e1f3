cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for a in "43 8" "43 8 8"; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_orbit -- python3 $R/scripts/symv_tune.py class_orbit $a > $R/gpurun_out/prof_orbit.txt 2>&1
f=$(ls -t $R/gpurun_out/prof_orbit/*/*kernel_stats.csv | head -n 1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith("k_fxo") : print(r["Name"][:20], r["Calls"], r["AverageNs"])
PY
done
find $R/gpurun_out/prof_orbit -name "*kernel_trace.csv" -delete
