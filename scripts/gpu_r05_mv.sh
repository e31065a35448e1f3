# round 5: the multi-right-hand-side K^+ in the set-up of the explicit operators -- A/B against one column per block (PMH_NO_MULTI_RHS=1):
#   the one-call contact solve (orbit storage, 723 solves), the non-congruent problem without symmetry at 21^3 (and at 43^3 with FULL=1)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
for mode in mv one; do
  if [ $mode = one ]; then export PMH_NO_MULTI_RHS=1; else unset PMH_NO_MULTI_RHS; fi
  PMH_CONTACT_TIMING=1 python scripts/contact_timing.py > $O/contact_timing_$mode.txt 2>&1
  grep "^call" $O/contact_timing_$mode.txt
  python bench.py --young distinct --no-explicit-symmetry --nel 20 --no-c2 --no-cpu-baseline --steps 40 --warmup 4 --details $O/general_nosym21_${mode}_details.json > $O/general_nosym21_$mode.json 2> $O/general_nosym21_$mode.err
  python - $O/general_nosym21_${mode}_details.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); k=d["config"]["kplus"]
print("no symmetry, 21^3:", {x: k.get(x) for x in ("assemble_seconds","assemble_solves","storage","assemble_multi_rhs")}, "value", d["value"], d["config"].get("checksum"))
PY
  if [ -n "$FULL" ]; then
    python bench.py --young distinct --no-explicit-symmetry --nel 42 --no-c2 --no-cpu-baseline --steps 40 --warmup 4 --details $O/general_nosym43_${mode}_details.json > $O/general_nosym43_$mode.json 2> $O/general_nosym43_$mode.err
    python - $O/general_nosym43_${mode}_details.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); k=d["config"]["kplus"]
print("no symmetry, 43^3:", {x: k.get(x) for x in ("assemble_seconds","assemble_solves","storage","assemble_multi_rhs")}, "value", d["value"], d["config"].get("checksum"))
PY
  fi
done
