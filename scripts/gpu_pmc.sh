# HBM traffic of the dominant kernel from PMC counters (separate passes, as MI355X_MICROARCH.md prescribes)
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$C -- python3 $R/bench.py --steps 48 --warmup 4 --no-cpu-baseline > $R/gpurun_out/pmc_$C.log 2>&1
  tail -1 $R/gpurun_out/pmc_$C.log | cut -c1-200
done
python3 $R/scripts/pmc_parse.py $R/gpurun_out
