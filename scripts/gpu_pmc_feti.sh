# HBM traffic of the FETI dual SpMV (K_i blocks) from PMC counters, separate passes
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmcf_$C -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-c2 > $R/gpurun_out/pmcf_$C.log 2>&1
  tail -1 $R/gpurun_out/pmcf_$C.log | cut -c1-200
done
mkdir -p $R/gpurun_out/pmcf && rm -rf $R/gpurun_out/pmcf/pmc_FETCH_SIZE $R/gpurun_out/pmcf/pmc_WRITE_SIZE
mv $R/gpurun_out/pmcf_FETCH_SIZE $R/gpurun_out/pmcf/pmc_FETCH_SIZE; mv $R/gpurun_out/pmcf_WRITE_SIZE $R/gpurun_out/pmcf/pmc_WRITE_SIZE
python3 $R/scripts/pmc_parse.py $R/gpurun_out/pmcf && cp $R/gpurun_out/pmcf/pmc_traffic.json $R/gpurun_out/pmc_traffic_feti.json
# keep only the summary (the per-dispatch CSVs are tens of MB)
rm -rf $R/gpurun_out/pmcf/pmc_FETCH_SIZE $R/gpurun_out/pmcf/pmc_WRITE_SIZE
# torch-first import check: libpermonhip must work on torch's bundled HIP runtime + RCCL (the N>1 launch path)
cd $R
PMH_COMM_FORCE=1 python3 - <<'PY'
import torch
torch.cuda.init(); print("torch", torch.__version__, torch.cuda.get_device_name(0))
import numpy as np, permon_amd as pa
ctx = pa.Context(0)
ctx.comm_init(0, 1, ctx.comm_unique_id())   # real RCCL communicator of size 1 (PMH_COMM_FORCE keeps the collectives on)
v = ctx.vec_from(np.arange(8.0))
pa._lib.check(ctx.L.pmh_comm_allreduce_sum(ctx.h, v.p, 8)); ctx.barrier()
assert np.array_equal(v.to_numpy(), np.arange(8.0))
import __graft_entry__ as g
ctx.close(); g.smoke(); print("torch-first + RCCL OK")
PY
PMH_COMM_FORCE=1 python3 - <<'PY'
import numpy as np, permon_amd as pa
ctx = pa.Context(0)
ctx.comm_init(0, 1, ctx.comm_unique_id())
v = ctx.vec_from(np.arange(8.0))
pa._lib.check(ctx.L.pmh_comm_allreduce_sum(ctx.h, v.p, 8)); ctx.barrier()
assert np.array_equal(v.to_numpy(), np.arange(8.0)); print("rocm RCCL OK")
PY
