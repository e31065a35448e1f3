#!/usr/bin/env python3
"""bench.py -- headline benchmark of the PERMON QPS hot path on MI355X.

Metric (BASELINE.json): QPS iterations/sec + CSR SpMV GB/s (% of HBM roofline).
Workload at N=1 (BASELINE.json configs[1]): synthetic SPD CSR, 5-point Laplacian on a 3162 x 3162 grid
(n = 9 998 244 rows, nnz = 49 978 572), box-constrained, MPGP, fp64, one MI355X.
A "step" is one MPGP iteration (one pass of the hot loop, src/qps/impls/mpgp/mpgp.c:511-641).

  python bench.py --gpus N --steps K --warmup W          (N=1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N>1)

Prints ONE JSON line on rank 0.  Inputs are resident in HBM before the timed region starts.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured copy ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--grid", type=int, default=3162, help="nx = ny of the 5-pt Laplacian (3162 -> configs[1])")
    ap.add_argument("--variant", default="obstacle", choices=["obstacle", "twosided"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-its", type=int, default=24, help="MPGP iterations of the bounded CPU-baseline sample")
    return ap.parse_args()


def cpu_baseline(p, its):
    """The oracle (a port of the reference's unfused op sequence, OpenMP over rows) timed on the host cores
    on a bounded sample: `its` MPGP iterations of the same workload.  Reported baseline, not the target."""
    from oracle import oracle as O

    # threads = the cores this process may actually run on (a cgroup/affinity mask can be far below cpu_count)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # cgroup v2 CPU quota (the GPU box: cpu.max = 1600000 100000 -> 16 CPUs of a 2 x 64-core EPYC 9575F)
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    os.environ["OMP_NUM_THREADS"] = str(cores)
    os.environ.setdefault("OMP_PROC_BIND", "close")
    A = O.Csr(p["n"], p["n"], p["rowptr"], p["col"], p["val"])
    op = O.Op(p["n"], csr=A, omp=True)
    box = O.Box(p["n"], lb=p["lb"], ub=p["ub"])
    # maxeig supplied => no power method inside the timed call; max_it bounds the sample
    lam, _ = O.max_eigenvalue(op, omp=True)
    t, done = O.time_mpgp(op, p["b"], p["x0"], box, reps=1, omp=True, maxeig=lam, max_it=its - 1)
    t_spmv = O.time_spmv(A, p["b"], reps=3, omp=True)
    return {
        "value": done / t,
        "unit": "QPS iterations/s",
        "cores": cores,
        "kind": "port",
        "sample": "%d MPGP iterations of the same %d-row workload, oracle/permon_oracle.c with OpenMP on %d threads "
                  "(os.cpu_count()=%d; reference op order, one pass per PETSc call); SpMV alone %.1f GB/s" % (done, p["n"], cores, os.cpu_count() or 0, (12.0 * A.val.size + 20.0 * p["n"]) / t_spmv / 1e9),
    }


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus %d needs a torch.distributed.run launch with %d ranks" % (a.gpus, a.gpus))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist_

        dist = dist_
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import permon_amd as pa
    from permon_amd import problems as P

    ctx = pa.Context(local_rank)
    t0 = time.time()
    p = P.laplace2d_box(a.grid, a.grid, variant=a.variant)
    n, nnz = p["n"], int(p["val"].size)
    A = pa.CsrMat(ctx, n, n, p["rowptr"], p["col"], p["val"])
    op = pa.Op.from_csr(A)
    qp = pa.QP(ctx)
    qp.SetOperator(op)
    qp.SetRhs(ctx.vec_from(p["b"]))
    x = ctx.vec_from(p["x0"])
    qp.SetInitialVector(x)
    qp.SetBox(None, ctx.vec_from(p["lb"]), ctx.vec_from(p["ub"]) if p["ub"] is not None else None)
    qps = pa.QPS(ctx)
    qps.SetQP(qp)
    qps.SetType("mpgp")
    qps.SetUp()  # power method (<= 50 SpMV), alpha = 2/lambda_max: set-up, outside the timed region
    t_setup = time.time() - t0

    def barrier():
        ctx.sync()
        if dist is not None:
            import torch

            dist.barrier()
            torch.cuda.synchronize()

    # warm-up: W untimed steps
    qps.RunFixed(a.warmup)
    x.set_numpy(p["x0"])
    pa._lib.check(ctx.L.pmh_mpgp_reset_statistics(qps.h))
    A.timing_enable(2 * a.steps + 8)
    barrier()
    t1 = time.perf_counter()
    st = qps.RunFixed(a.steps)
    barrier()
    dt = time.perf_counter() - t1
    if dist is not None:
        import torch

        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    assert st.iteration == a.steps, (st.iteration, a.steps)
    n_p1, ms_p1 = A.timing_get(3)  # fused MPGP phase-P1 SpMV launches (the dominant kernel)
    n_sub, ms_sub = A.timing_get(2)
    has_ub = p["ub"] is not None
    b_spmv = 12.0 * nnz + 20.0 * n
    b_p1 = b_spmv + (32.0 if has_ub else 24.0) * n  # + reads of g, x, lb (ub) in the fused epilogue
    vec_extra = 16.0 * n if has_ub else 0.0
    b_cg, b_prop, b_exp = b_spmv + 112.0 * n + vec_extra, b_spmv + 104.0 * n + vec_extra, 2 * b_spmv + 112.0 * n + vec_extra
    alg_bytes = st.ncg * b_cg + st.nprop * b_prop + st.nexp * b_exp
    achieved = b_p1 / (ms_p1 / n_p1 * 1e-3) / 1e9 if n_p1 else 0.0

    # HBM bytes per launch of the same kernel from rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE),
    # collected by scripts/gpu_pmc.sh with this very command and committed under profiles/
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
        for k, v in pmc.items():
            if k.startswith("void k_spmv_stream<3,") and a.grid == 3162 and a.variant == "obstacle":
                traffic = v["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass

    out = {
        "metric": "QPS iterations/sec + CSR SpMV GB/s (% HBM roofline)",
        "value": world * a.steps / dt,
        "unit": "QPS iterations/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "configs[1]: 5-pt Laplacian %dx%d (n=%d, nnz=%d) MPGP box QP (%s), fp64, CSR int32" % (a.grid, a.grid, n, nnz, a.variant),
            "parallelism": "1 GPU" if world == 1 else "%d independent replicas (configs[1] is a single-GPU config; replicas only)" % world,
            "steps_by_type": {"cg": st.ncg, "expansion": st.nexp, "proportioning": st.nprop, "hessian_mults": st.nmv},
            "setup_seconds": round(t_setup, 2),
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "k_spmv_stream<MPGP epilogue> (Ap = A p fused with p'Ap, g'p, QPCFeas)",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "algorithmic_bytes_per_launch": b_p1,
            "launches_timed": n_p1,
            "avg_launch_ms": ms_p1 / n_p1 if n_p1 else None,
            "spmv_only_GBs": b_spmv / (ms_p1 / n_p1 * 1e-3) / 1e9 if n_p1 else None,
            "whole_iteration_GBs": alg_bytes / dt / 1e9,
            "whole_iteration_frac": alg_bytes / dt / 1e9 / HBM_PEAK_GBS,
        },
    }
    if n_sub:
        out["roofline"]["gradient_spmv_avg_ms"] = ms_sub / n_sub
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(p, a.cpu_its)
        except Exception as e:  # noqa: BLE001 - the baseline leg must not kill the GPU number
            out["cpu_baseline"] = {"value": None, "unit": "QPS iterations/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}
    if rank == 0:
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
