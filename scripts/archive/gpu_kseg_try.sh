# k-segment pruning + piece partition of the orbit GEMM: parity tests of the orbit storage, the headline bench with the plan printed, kernel durations (with A/B knobs)
cd $GRAFT_REPO_ROOT
O=gpurun_out/kseg
mkdir -p $O
echo skip tests

PMH_FXO_VERBOSE=1 python bench.py --no-cpu-baseline --no-c2 --no-iterative > $O/bench_kseg.json 2> $O/bench_kseg.err; grep "PMH_FX_CLASS_ORBIT" $O/bench_kseg.err | cut -c1-400
python3 - <<'PY'
import json
for f in ["bench_kseg"]:
    d=json.loads(open("gpurun_out/kseg/"+f+".json").read().strip().splitlines()[-1]); r=d["roofline"]; c=d["config"]["steps_by_type"]
    print(f, round(d["value"],1), "it/s", round(d["ms_per_step"],4), "dense", round(r["avg_launch_ms"],4), "frac", round(r["frac"],3), "checksum", d["config"].get("checksum"), c)
PY
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/kseg
export PMH_BENCH_NO_TIMING=1
prof() { # name, env assignments...
  name=$1; shift
  for kv in "$@"; do export "$kv"; done
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- python3 $R/bench.py --no-cpu-baseline --no-c2 --no-iterative --steps 100 > $O/prof_$name.json 2> $O/prof_$name.err
  for kv in "$@"; do unset "${kv%%=*}"; done
  find $O/prof_$name -name "*kernel_trace.csv" -delete
  f=$(find $O/prof_$name -name "*kernel_stats.csv")
  echo "== $name: $(grep -E 'k_fxo_gemm|k_fxo_fin' $f | cut -d, -f1,2,4 | sed 's/(int[^"]*"/"/' | tr '\n' ' ')"
}
prof streamk
prof aligned PMH_FXO_NO_STREAMK=1

prof nokseg_aligned PMH_FXO_NO_KSEG=1 PMH_FXO_NO_STREAMK=1
