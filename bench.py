#!/usr/bin/env python3
"""bench.py -- headline benchmark of the PERMON QPS hot path on MI355X.

Metric (BASELINE.json): QPS iterations/sec + CSR SpMV GB/s (% of HBM roofline), 1/2/4/8 MI355X.

Workloads
  feti (default, every N): BASELINE.json configs[2] -- 3-D elasticity TFETI, 2x2x2 cubic subdomains of 43^3 Q1
      elements (N = 2 044 416 primal dof, K_i: 255 552 rows / 19.77 M nnz each, n_lambda = 102 268 incl. 7 744
      contact rows), rigid obstacle, SMALXE + MPGP on the dual QP with F = B K^+ B'.  The 8 subdomain blocks
      are sharded over the N GPUs (8/N per GPU, STRONG scaling: the north-star's own scaling target); dual
      vectors are replicated, B u is summed with one RCCL all-reduce per F apply.  This is the configuration the
      north-star quotes both of its targets on (>= 60 % HBM roofline on the FETI dual SpMV at 1 GPU, >= 6x
      iterations/s at 8 GPUs); it fits one GPU (2.2 GB).
      A "step" = one inner MPGP iteration of SMALXE (mpgp.c:511-641) on A_rho = P F P + rho Q, i.e. one F apply
      (block-wise CG K^+ on every subdomain), two projector applies, the ||G u|| of the injected convergence test
      and the fused vector phases.
  c2: BASELINE.json configs[1] -- 5-pt Laplacian 3162^2 (n = 9 998 244, nnz = 49 978 572) box QP, MPGP, one GPU.
      At N = 1 it is ALSO run and reported in the same JSON line under "configs1" (with its own roofline object).

  python bench.py --gpus N --steps K --warmup W                    (any N: with N > 1 and no WORLD_SIZE in the environment bench.py starts its N ranks itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Rank 0 prints ONE compact JSON line (< 4 KB: the contract's keys, `roofline`, `cpu_baseline`, one-number summaries of the secondary blocks) as the LAST line of
stdout and writes the full object (every block with its notes, ~30 KB) to bench_details.json (--details PATH).  Inputs are resident in HBM before the timed region starts.
"""
import argparse
import json
import math
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
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="feti", choices=["feti", "c2", "svm"])
    ap.add_argument("--svm-n", type=int, default=5000000, help="svm: total number of samples (configs[4]: 5 M x 64)")
    ap.add_argument("--grid", type=int, default=3162, help="c2: nx = ny of the 5-pt Laplacian")
    ap.add_argument("--variant", default="obstacle", choices=["obstacle", "twosided"])
    ap.add_argument("--nel", type=int, default=43, help="feti: Q1 elements per subdomain edge (43 -> configs[2])")
    ap.add_argument("--sub", default="2,2,2", help="feti: subdomain grid sx,sy,sz (2,2,2 -> configs[2]; 4,4,4 with --nel 21 --dense-coarse -> the shape of configs[3]: 64 subdomains, 8 per GPU at N = 8, 384 x 384 coarse problem)")
    ap.add_argument("--orth-form", choices=["implicit", "explicit"], default="implicit",
                    help="feti: QPTOrthonormalizeEq's form (-qp_E_orth_form; the reference's default is implicit, qptransform.c:647): implicit keeps G = R'B' as sparse as it is and applies "
                         "the small dense T = chol(GG')^{-1} inside the finishing launch of G v; explicit hands the filled T G to the library (3.4 x the non-zeros for configs[2]). Same operator either way")
    ap.add_argument("--dense-coarse", action="store_true", help="feti: keep G as it comes (no QPTOrthonormalizeEq): the projector applies the dense (GG')^{-1} (GG' assembled by the fp64-MFMA kernel)")
    ap.add_argument("--kplus-rtol", type=float, default=1e-9, help="feti: relative tolerance of the block-wise CG K^+")
    ap.add_argument("--kplus", choices=["explicit", "iterative"], default="explicit", help="feti: how F = B K^+ B' applies K^+: explicit = the dense local dual operators W_b = (K_b^+)[Gamma_b, Gamma_b] "
                    "(assembled once by K^+ solves, then ONE fp64 GEMV per apply; the exact path and the faster one at every N), iterative = an inner block-wise Krylov solve per apply")
    ap.add_argument("--explicit-storage", choices=["auto", "class_orbit", "class_sym", "class", "sym", "full"], default="auto",
                    help="feti: the dense local dual operators per block as their lower block-triangle (sym: SYMV, 4 n^2 bytes per apply) or in full (full: GEMV, 8 n^2), or ONE full matrix per class "
                         "of congruent blocks applied to 8 blocks' vectors per pass (class: 8 n_c^2 for the whole class; class_sym: its lower block-triangle in 16x16 tiles, 4 n_c^2, both products of a tile on the "
                         "fp64 matrix instruction; class_orbit: only the rows of the orbit representatives under the cube's symmetries, applied as a GEMM on the fp64 matrix instruction); "
                         "auto = class_orbit when the blocks are congruent cubes with >= 16 symmetries, class_sym when congruent, else sym")
    ap.add_argument("--no-explicit-symmetry", action="store_true", help="feti: assemble every row of the class-shared explicit operator by its own K^+ solve instead of one solve per orbit of rows under "
                    "the cube's 48 signed coordinate permutations (checked against K; pmh_fexplicit_set_class_symmetry)")
    ap.add_argument("--no-stripe", action="store_true", help="feti at N > 1: every rank keeps the explicit operators of its OWN blocks instead of an even share of 128-row stripes of all blocks")
    ap.add_argument("--explicit-rtol", type=float, default=1e-12, help="feti: tolerance of the set-up solves of the explicit operators")
    ap.add_argument("--explicit-slots", type=int, default=8, help="feti: a rank with fewer (congruent) blocks than this assembles with a replica solver of this many slots")
    ap.add_argument("--no-iterative", action="store_true", help="feti at N=1: skip the secondary passes on the inner-Krylov K^+ (fp16-PC and strict fp64)")
    ap.add_argument("--kplus-pc", choices=["mg", "jacobi"], default="mg", help="feti: PC of the inner CG of K^+ (-mat_inv_pc_type): multigrid V-cycle or Jacobi")
    ap.add_argument("--mg-precision", choices=["fp16", "fp32", "fp64"], default="fp16", help="feti: precision of the V-cycle (it only preconditions the fp64 CG)")
    ap.add_argument("--mg-min-nodes", type=int, default=0, help="feti: the hierarchy stops coarsening at <= this many nodes per block (dense block pseudo-inverse there); 0 = by blocks per GPU")
    ap.add_argument("--mg-builder", choices=["c", "python"], default="c", help="feti: who builds the V-cycle hierarchy of the GPU solver: pmh_mg_create_box inside the library, or permon_amd.feti.box_mg_hierarchy (scipy)")
    ap.add_argument("--mg-degree", type=int, default=2, help="feti: Chebyshev degree of the V-cycle smoother")
    ap.add_argument("--regularize", action="store_true", help="feti: K^+ = K_reg^{-1} with K_reg = MatRegularize(K, R) (the reference's default, -regularize 1) instead of the Moore-Penrose wrapping "
                    "P_R K^- P_R (-regularize 0 -qpt_dualize_Kplus_mp); identical on the projected dual problem, the V-cycle hierarchy is then built per block on K_reg")
    ap.add_argument("--no-bsr3", action="store_true", help="feti: keep K x of the inner CG on the CSR kernel instead of the 3x3-block kernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c2", action="store_true", help="feti at N=1: the headline alone -- skip every secondary block (configs[1], configs[3], configs[4], general, contact_solve)")
    ap.add_argument("--sim-world", type=int, default=0, help="feti, testing: this single process takes the share rank 0 would have in a run on SIM_WORLD GPUs (8/SIM_WORLD blocks, no collective): per-GPU launch-latency rehearsal of the strong-scaling run")
    ap.add_argument("--cpu-its", type=int, default=24, help="c2: MPGP iterations of the bounded CPU-baseline sample")
    ap.add_argument("--cpu-its-feti", type=int, default=5, help="feti: MPGP iterations of the bounded CPU-baseline sample (median of their times; cut at ~45 s)")
    ap.add_argument("--young", default="", help="feti: Young's moduli of the subdomains, comma separated, or 'distinct' (1, 1.25, 1.5 ...): a heterogeneous body whose blocks K_s = E_s K_1 all differ -- "
                    "the non-congruent case: no class sharing, no symmetry set-up, per-block symmetric storage (k_fx_symv, HBM-bound)")
    ap.add_argument("--general-nel", type=int, default=43, help="feti at N=1: elements per edge of the secondary 'general' block (8 subdomains of 8 different materials: no two blocks congruent; one class per block, "
                    "each on the closure of its touched set under the cube's 48 symmetries -> one K^+ solve per orbit: 43 -> ~15 s of set-up; with --no-explicit-symmetry every column of every W_b by its own "
                    "K^+ solve: 21 -> ~20 s, 43 -> minutes); 0 = skip")
    ap.add_argument("--partition", choices=["", "staircase", "lshape"], default="", help="feti: cut the (2 nel)^3-element cube into 8 subdomains that are NOT boxes (permon_amd.feti.irregular_partition: staircase "
                    "interfaces / L-shaped bodies) instead of the 2 x 2 x 2 cubes: no congruence, no symmetry, no box hierarchy -- K^+ on the algebraic hierarchy (pmh_mg_create_sa), per-block explicit operators (k_fx_symv)")
    ap.add_argument("--nosym-nel", type=int, default=21, help="feti at N=1: half the elements per edge of the secondary 'general_nosym' block (the irregular 'staircase' partition of a (2 nel)^3-element cube; "
                    "every column of every W_b by its own K^+ solve, 8 per block at a time: 21 -> seconds, 43 -> ~2 min of set-up); 0 = skip")
    ap.add_argument("--c2-steps", type=int, default=2500, help="feti at N=1: MPGP iterations of the secondary configs[1] block (from x0 = 0 the first ~hundreds of iterations are pure CG; the expansion steps start once the iterate reaches the obstacle)")
    ap.add_argument("--no-configs3", action="store_true", help="feti at N=1: skip the secondary configs[3] block (4x4x4 subdomains of 21^3 elements, dense 384 x 384 coarse problem)")
    ap.add_argument("--no-svm", action="store_true", help="feti at N=1: skip the secondary configs[4] block (5 M x 64 SVM dual)")
    ap.add_argument("--no-contact-solve", action="store_true", help="feti at N=1: skip the one-call contact solve (pmh_feti_contact_solve: set-up + solve = time to solution)")
    ap.add_argument("--cpu-direct-nel", type=int, default=21, help="feti: subdomain size at which the CPU baseline factors K_reg with scipy's SuperLU (the reference's direct K^+); the 43^3 block itself would take ~20 min and ~10 GB")
    ap.add_argument("--details", default=os.path.join(ROOT, "bench_details.json"), help="where rank 0 writes the full result object (the stdout line is the compact summary)")
    ap.add_argument("--dry-launch", action="store_true", help="testing: with --gpus N > 1 start the N ranks as usual, but every rank only reports its rendezvous environment and exits (no GPU, no torch)")
    ap.add_argument("--no-cpu-pool", action="store_true", help="feti at N=1: do not start the worker processes of the configs[3] CPU leg (the reference's op sequence with a direct K^+, timed live)")
    ap.add_argument("--no-dual-spmv", action="store_true", help="feti at N=1: skip the HBM-streaming measurement of MatMult_BlockDiag (8 DISTINCT K_i, one device copy each) at the headline size")
    return ap.parse_args()


_HOST_THREADS = None
CPU_POOL = None  # oracle.direct_pool.DirectPool of the configs[3] CPU leg (started in main() before the GPU is initialised)
_DIRECT_ROWS = []  # cpu_baseline_direct's measured (nel, n, factor_s, solve_s) rows: the configs[3] block re-uses the measured 21^3 solve


def host_threads():
    """Threads the CPU baseline may really use: affinity mask capped by the cgroup CPU quota.  Evaluated once, before
    any OpenMP runtime is loaded (libgomp reads OMP_NUM_THREADS at load time and may re-bind the main thread)."""
    global _HOST_THREADS
    if _HOST_THREADS is not None:
        return _HOST_THREADS
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # cgroup v2 quota (the GPU box: cpu.max = 1600000 100000 -> 16 CPUs of a 2 x 64-core EPYC 9575F)
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    # several ranks on one node share its cores: every rank takes cores // (ranks on the node) for its numpy / OpenMP work AND for the library's host-side builders
    # (pmh_set_knob "host_threads" reads PMH_HOST_THREADS once): 8 ranks x 16 threads on a 16-CPU cgroup was the first-run risk of the N = 8 bench
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")) or 1))
    cores = max(1, cores // local_world)
    _HOST_THREADS = cores
    os.environ["OMP_NUM_THREADS"] = str(cores)
    os.environ["PMH_HOST_THREADS"] = str(cores)
    for k in ("OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[k] = str(cores)
    return cores


def cpu_model():
    """The host CPU's model name (core count AND model stated next to every CPU baseline)."""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform

    return platform.processor() or platform.machine()


HOST_TRANSPORT = os.environ.get("PMH_BENCH_TRANSPORT") == "host"  # test mode: the ranks share ONE GPU, the collectives ride on the library's host transport over gloo


def dist_max(dist, value):
    """MAX of a host scalar over the ranks (the contract's max-over-ranks time)."""
    import torch

    tt = torch.tensor([value], dtype=torch.float64, device="cpu" if HOST_TRANSPORT else "cuda")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt.item())


def global_norm(dist, local_norm):
    """2-norm of a row-distributed vector from the ranks' local norms (the SVM workload shards the samples)."""
    if dist is None:
        return local_norm
    import torch

    tt = torch.tensor([local_norm * local_norm], dtype=torch.float64, device="cpu" if HOST_TRANSPORT else "cuda")
    dist.all_reduce(tt, op=dist.ReduceOp.SUM)
    return float(tt.item()) ** 0.5


def measured_ceiling(ctx, n=1 << 26, reps=10):
    """On-box HBM ceiling (SURVEY 8d asks for it next to the 8 TB/s spec): device copy (read n, write n) and the
    VecWAXPY triad (read 2n, write n) on 512 MiB vectors, HIP-event timed on the launch stream."""
    x, y, w = ctx.vec(n), ctx.vec(n), ctx.vec(n)
    x.set(1.0)
    y.set(2.0)
    res = {}
    for name, fn, nbytes in (("copy_GBs", lambda: ctx.L.pmh_vec_copy(ctx.h, n, x.p, w.p), 16.0 * n), ("triad_GBs", lambda: w.waxpy(0.5, x, y), 24.0 * n)):
        for _ in range(2):
            fn()
        ctx.sync()
        ctx.timer_start()
        for _ in range(reps):
            fn()
        ms = ctx.timer_stop() / reps
        res[name] = nbytes / ms / 1e6
    for v in (x, y, w):
        v.free()
    return res


# ------------------------------------------------------------------------------------------------------------------
# configs[1]: single CSR, MPGP
# ------------------------------------------------------------------------------------------------------------------
def cpu_baseline_c2(p, its):
    """The oracle (a port of the reference's unfused op sequence, OpenMP over rows) timed on the host cores on a
    bounded sample: `its` MPGP iterations of the same workload.  Reported baseline, not the target."""
    from oracle import oracle as O

    cores = host_threads()
    A = O.Csr(p["n"], p["n"], p["rowptr"], p["col"], p["val"])
    op = O.Op(p["n"], csr=A, omp=True)
    box = O.Box(p["n"], lb=p["lb"], ub=p["ub"])
    lam, _ = O.max_eigenvalue(op, omp=True)  # maxeig supplied => no power method inside the timed call
    t, done = O.time_mpgp(op, p["b"], p["x0"], box, reps=1, omp=True, maxeig=lam, max_it=its - 1)
    t_spmv = O.time_spmv(A, p["b"], reps=3, omp=True)
    op1 = O.Op(p["n"], csr=A, omp=False)  # "all host cores and also 1 core" -- the plain (non-OpenMP) build of the oracle on a shorter sample
    t1, done1 = O.time_mpgp(op1, p["b"], p["x0"], box, reps=1, omp=False, maxeig=lam, max_it=max(2, its // 6) - 1)
    return {
        "value": done / t, "unit": "QPS iterations/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
        "one_core": {"value": done1 / t1, "unit": "QPS iterations/s", "cores": 1, "sample": "%d iterations of the same workload on one thread" % done1},
        "sample": "%d MPGP iterations of the same %d-row workload, oracle/permon_oracle.c with OpenMP on %d threads "
                  "(os.cpu_count()=%d, cgroup quota honoured; reference op order, one pass per PETSc call); SpMV alone %.1f GB/s"
                  % (done, p["n"], cores, os.cpu_count() or 0, (12.0 * A.val.size + 20.0 * p["n"]) / t_spmv / 1e9),
    }


def run_c2(ctx, a, steps, warmup, cpu=True, whole_solves=False):
    """whole_solves: time WHOLE MPGP solves from x0 (expansion and proportioning steps included, the real stopping rule; restarted until at least `steps`
    iterations are done) instead of exactly `steps` iterations from x0 with the verdict of the test ignored (the first few hundred are pure CG steps)."""
    import permon_amd as pa
    from permon_amd import problems as P

    t0 = time.time()
    p = P.laplace2d_box(a.grid, a.grid, variant=a.variant)
    n, nnz = p["n"], int(p["val"].size)
    A = pa.CsrMat(ctx, n, n, p["rowptr"], p["col"], p["val"])
    qp = pa.QP(ctx)
    qp.SetOperator(pa.Op.from_csr(A))
    qp.SetRhs(ctx.vec_from(p["b"]))
    x = ctx.vec_from(p["x0"])
    qp.SetInitialVector(x)
    qp.SetBox(None, ctx.vec_from(p["lb"]), ctx.vec_from(p["ub"]) if p["ub"] is not None else None)
    qps = pa.QPS(ctx)
    qps.SetQP(qp)
    qps.SetType("mpgp")
    qps.SetUp()  # power method (<= 50 SpMV), alpha = 2/lambda_max: set-up, outside the timed region
    t_setup = time.time() - t0
    qps.RunFixed(warmup)
    x.set_numpy(p["x0"])
    pa._lib.check(ctx.L.pmh_mpgp_reset_statistics(qps.h))
    nsolves, reasons = 0, []
    if whole_solves:
        # QPS defaults (qps.c:73-76): rtol 1e-5, max_it 10 000 = SURVEY 8d's configs[1] set-up
        A.timing_enable(2 * 10001 * max(1, -(-steps // 10001)) + 64)
        ctx.sync()
        t1 = time.perf_counter()
        done = 0
        while done < steps:
            x.set_numpy(p["x0"])
            st = qps.Solve()
            done += int(st.iteration)
            nsolves += 1
            reasons.append(int(st.reason))
            if st.iteration == 0:
                break
        ctx.sync()
        dt = time.perf_counter() - t1
        steps = done
    else:
        A.timing_enable(2 * steps + 64)
        ctx.sync()
        t1 = time.perf_counter()
        st = qps.RunFixed(steps)
        ctx.sync()
        dt = time.perf_counter() - t1
        assert st.iteration == steps, (st.iteration, steps)
    n_p1, ms_p1 = A.timing_get(3)  # fused MPGP phase-P1 SpMV launches (the dominant kernel)
    has_ub = p["ub"] is not None
    b_spmv = 12.0 * nnz + 20.0 * n
    b_p1 = b_spmv + (32.0 if has_ub else 24.0) * n  # + reads of g, x, lb (ub) in the fused epilogue
    ve = 16.0 * n if has_ub else 0.0
    alg = st.ncg * (b_spmv + 112.0 * n + ve) + st.nprop * (b_spmv + 104.0 * n + ve) + st.nexp * (2 * b_spmv + 112.0 * n + ve)
    achieved = b_p1 / (ms_p1 / n_p1 * 1e-3) / 1e9 if n_p1 else 0.0
    res = {
        "value": steps / dt, "unit": "QPS iterations/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup,
        "workload": "configs[1]: 5-pt Laplacian %dx%d (n=%d, nnz=%d) MPGP box QP (%s), fp64, CSR int32" % (a.grid, a.grid, n, nnz, a.variant),
        "steps_by_type": {"cg": st.ncg, "expansion": st.nexp, "proportioning": st.nprop, "hessian_mults": st.nmv},
        "timed": ("%d whole solve(s) from x0 (rtol 1e-5, max_it 10 000; reasons %s): the real step mix" % (nsolves, reasons)) if whole_solves else "exactly %d iterations from x0, verdict of the convergence test ignored" % steps,
        "algorithmic_bytes_per_step_type": {"cg": b_spmv + 112.0 * n + ve, "expansion": 2 * b_spmv + 112.0 * n + ve, "proportioning": b_spmv + 104.0 * n + ve,
                                            "note": "SURVEY 8d: B_cg = B_spmv + 112 n, B_exp = 2 B_spmv + 112 n, B_prop = B_spmv + 104 n (+16 n with an upper bound); whole_iteration_GBs = sum over the timed steps / wall time"},
        "setup_seconds": round(t_setup, 2),
        "roofline": {
            "bound": "hbm", "kernel": "k_spmv_ell<MPGP epilogue> (uniformly short rows: slot-major device copy, one thread per row, no LDS staging; k_spmv_stream otherwise): Ap = A p fused with p'Ap, g'p, QPCFeas",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            **dict(zip(("traffic", "traffic_source"), (lambda t: t if t[0] is not None else pmc_lookup("void k_spmv_stream<3,", "r04_pmc_traffic_c2.json"))(pmc_lookup("void k_spmv_ell<3,", "r04_pmc_traffic_c2.json")) if (a.grid == 3162 and a.variant == "obstacle") else (None, "not the configuration of the committed PMC pass"))),
            "algorithmic_bytes_per_launch": b_p1, "launches_timed": n_p1, "avg_launch_ms": ms_p1 / n_p1 if n_p1 else None,
            "whole_iteration_GBs": alg / dt / 1e9, "whole_iteration_frac": alg / dt / 1e9 / HBM_PEAK_GBS,
            "note": "algorithmic bytes are SURVEY 8d's CSR figure (12 B per non-zero: fp64 value + int32 column); the kernel streams a device-private copy of the columns as 16-bit offsets "
                    "per row block (banded matrix), so its HBM traffic (PMC) is BELOW the algorithmic bytes: 2 B per non-zero less (rows padded to the longest row: +2 % for the 5-point Laplacian)",
        },
    }
    A.timing_enable(0)
    if cpu:
        try:
            res["cpu_baseline"] = cpu_baseline_c2(p, a.cpu_its)
        except Exception as e:  # noqa: BLE001 - the baseline leg must not kill the GPU number
            res["cpu_baseline"] = {"value": None, "unit": "QPS iterations/s", "cores": host_threads(), "kind": "port", "sample": "failed: %r" % (e,)}
    return res


# ------------------------------------------------------------------------------------------------------------------
# configs[4]: PermonSVM-style hinge-loss dual, dense-row Hessian, samples sharded by rows over the GPUs
# ------------------------------------------------------------------------------------------------------------------
def svm_roofline(N, n, d, world, st, dt, passes):
    """configs[4]: the SURVEY 8d figure counts X twice per Hessian application (2*8*N*d + 40*N).  Inside MPGP the library pairs the second pass of one application
    with the first pass of the next wherever the step allows it (svm.hip, "paired passes"), so `achieved` on that figure can exceed the HBM peak: `streamed_GBs` is what
    was actually moved (the operator counts its passes over X: pmh_op_svm_dual_passes; + 17 n-vectors read or written per expansion step by the two fused passes), `frac_streamed`
    its fraction of the peak -- the number to judge the kernels by."""
    b_H = 2.0 * 8 * n * d + 40.0 * n
    achieved = st.nmv * b_H / dt / 1e9
    steps = max(1, st.ncg + st.nexp + st.nprop)
    streamed = (passes * 8.0 * n * d + steps * 17.0 * 8.0 * n) / dt / 1e9  # per expansion step the two fused passes read y, p, g, x, lb, ub / y, x+, b, lb, ub and write Ap, x+ / g, gf, p, x: 17 vectors
    traffic, tsrc = (None, "not the configuration of the committed PMC pass")
    if N == 5000000 and world == 1:
        # HBM bytes per Hessian application from the committed PMC pass: all k_svm* launches, divided by the applications (= the launches of the second-pass kernels)
        try:
            c4file = next(f_ for f_ in ("r05_pmc_traffic_configs4.json", "r04_pmc_traffic_configs4.json") if os.path.exists(os.path.join(ROOT, "profiles", f_)))
            pmc = json.load(open(os.path.join(ROOT, "profiles", c4file)))
            # a Hessian application of the steady state (a run of expansion steps) is ONE launch of either paired kernel + the 64-column sum before it
            pk = [v["hbm_bytes_per_launch"] for k, v in pmc.items() if k != "_meta" and ("k_svm_x64_grad" in k or "k_svm_x64_p1<1>" in k)]
            cs = [v["hbm_bytes_per_launch"] for k, v in pmc.items() if k != "_meta" and "k_svm_colsum_feas" in k]
            meta = pmc.get("_meta", {})
            if len(pk) == 2:
                traffic, tsrc = sum(pk) / 2 + (cs[0] if cs else 0.0), "profiles/" + c4file + " @ %s (%s): mean of k_svm_x64_grad and k_svm_x64_p1<1> (one launch = one application in a run of expansion steps) + k_svm_colsum_feas" % (meta.get("git", "?"), meta.get("command", "?"))
        except (OSError, ValueError) as ex:
            tsrc = "no PMC pass: %r" % (ex,)
    b_pair = 8.0 * n * d + 68.0 * n  # one pass over X + half of the 17 vectors the two fused passes of an expansion step read or write
    ach2 = st.nmv * b_pair / dt / 1e9
    return {"bound": "hbm", "kernel": "k_svm_x64_p1 + k_svm_x64_grad (paired passes over X inside MPGP; k_svm_xt64 + k_svm_x64 for a lone application)", "achieved": ach2, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach2 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": b_pair, "traffic": traffic, "traffic_source": tsrc,
            "passes_over_X": passes, "passes_per_application": passes / max(1, st.nmv), "streamed_GBs": streamed, "frac_streamed": streamed / HBM_PEAK_GBS,
            "survey_figure": {"bytes_per_application": b_H, "achieved": achieved, "frac": achieved / HBM_PEAK_GBS,
                              "note": "SURVEY 8d counts X twice per Hessian application (2*8*N*d + 40*N): the separate passes move that (PMH_SVM_NO_PAIRING=1: 0.61-0.67 of the peak on it); the paired passes move about half, "
                                      "so on this figure they 'exceed' the peak"},
            "note": "achieved = bytes of a PAIRED Hessian application (8*N*d + 68*N: X once, half of the 17 vectors of an expansion step's two fused passes) x applies / WHOLE step time (vector phases, the two 64-column "
                    "sums and the finalising launches included); streamed_GBs = what the operator's own pass counter says was moved (lone applications of the set-up's power method stream X twice)"}


def run_svm(ctx, a, steps, warmup, rank, world, dist):
    import permon_amd as pa
    from permon_amd import problems as P

    t0 = time.time()
    N, d = a.svm_n, 64
    lo, hi = rank * N // world, (rank + 1) * N // world
    # every rank draws the same stream and keeps its row slice (chunked to bound host memory)
    rng = np.random.default_rng(7)
    w_true = np.random.default_rng(8).standard_normal(d)
    X = np.empty((hi - lo, d))
    y = np.empty(hi - lo)
    chunk = 500000
    for s in range(0, N, chunk):
        e = min(N, s + chunk)
        Xc = rng.standard_normal((e - s, d))
        yc = np.sign(Xc @ w_true + 0.1 * rng.standard_normal(e - s))
        a0, a1 = max(s, lo), min(e, hi)
        if a1 > a0:
            X[a0 - lo:a1 - lo] = Xc[a0 - s:a1 - s]
            y[a0 - lo:a1 - lo] = yc[a0 - s:a1 - s]
    y[y == 0] = 1.0
    n = hi - lo
    H = pa.MatCreateSVMDual(ctx, X, y)
    del X
    qp = pa.QP(ctx)
    qp.SetOperator(H)
    qp.SetRhs(ctx.vec_from(np.ones(n)))
    x = ctx.vec(n)
    qp.SetInitialVector(x)
    qp.SetBox(None, ctx.vec(n), ctx.vec_from(np.ones(n)))
    qps = pa.QPS(ctx)
    qps.SetQP(qp)
    qps.SetType("mpgp")
    qps.MPGPSetDistributed(world > 1)
    qps.SetUp()
    t_setup = time.time() - t0

    def barrier():
        ctx.sync()
        if dist is not None:
            import torch

            dist.barrier()
            torch.cuda.synchronize()

    qps.RunFixed(warmup)
    x.set(0.0)
    pa._lib.check(ctx.L.pmh_mpgp_reset_statistics(qps.h))
    barrier()
    passes0 = H.passes()
    t1 = time.perf_counter()
    st = qps.RunFixed(steps)
    barrier()
    dt = time.perf_counter() - t1
    passes = H.passes() - passes0  # how many times X was streamed in the timed region (pmh_op_svm_dual_passes)
    if dist is not None:
        import torch

        dt = dist_max(dist, dt)
    comm_rank, comm_size = ctx.comm_rank()
    return {
        "rccl_ranks": comm_size if (world > 1 or os.environ.get("PMH_BENCH_FORCE_DIST")) else None, "checksum": {"norm_x_after_last_step": repr(global_norm(dist if world > 1 else None, float(x.norm())))},
        "value": steps / dt, "ms_per_step": dt / steps * 1e3,
        "workload": "configs[4]: PermonSVM-style hinge-loss dual, N=%d samples x d=%d dense fp64 (%.2f GB), H = diag(y) X X' diag(y) matrix-free, MPGP box 0<=a<=1" % (N, d, N * d * 8 / 1e9),
        "parallelism": "samples sharded by rows over %d GPU(s); w all-reduce (d doubles) per Hessian apply; scalar all-reduces for the MPGP reductions" % world,
        "steps_by_type": {"cg": st.ncg, "expansion": st.nexp, "proportioning": st.nprop, "hessian_mults": st.nmv},
        "setup_seconds": round(t_setup, 1),
        "roofline": svm_roofline(N, n, d, world, st, dt, passes),
    }


# ------------------------------------------------------------------------------------------------------------------
# configs[2]: TFETI contact problem, SMALXE + MPGP on the dual QP, subdomain blocks sharded over the GPUs
# ------------------------------------------------------------------------------------------------------------------
def cpu_baseline_feti(f, G, hier, b_dual, lb_dual, its, rtol, orth=True, budget_s=60.0):
    """The same algorithm as the GPU's ITERATIVE K^+ path on the host cores: the oracle's MPGP (C, reference op order) on
    A_rho = P F P + rho Q, F = B K^+ B', K^+ = block-wise V-cycle-preconditioned CG (oracle/mg_host.py: same hierarchy, Chebyshev(2)/
    Jacobi smoothing, dense coarse pseudo-inverses, fp64 throughout, Moore-Penrose wrapped), sparse products by the OpenMP CSR
    kernel of oracle/permon_oracle.c, vectors in numpy.  (The reference's own K^+ is a sparse Cholesky forward / backward solve per block: that is
    cpu_baseline_direct -- scipy's SuperLU standing in for PETSc Cholesky / MUMPS -- and the line's `cpu_baseline`; this leg is reported next to it as
    `cpu_baseline_iterative`: the host port of the GPU's inner-Krylov path.)
    Bounded sample: `its` MPGP iterations (fewer if one application is too slow for the time budget); every Hessian application
    is timed, the median x applications per iteration is reported."""
    from oracle import oracle as O
    from oracle.mg_host import KplusMG

    cores = host_threads()
    L = O.lib(True)
    import ctypes as C

    A = [O.Csr.from_scipy(a) for a in hier["A"]]
    P = [O.Csr.from_scipy(p) for p in hier["P"]]
    Pt = [O.Csr.from_scipy(p.T.tocsr()) for p in hier["P"]]

    def spmv(tag, l, x):
        M = A[l] if tag == "A" else P[l] if tag == "P" else Pt[l]
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty(M.nrows)
        L.orc_csr_mult(C.byref(M.c), x.ctypes.data_as(C.POINTER(C.c_double)), y.ctypes.data_as(C.POINTER(C.c_double)))
        return y

    Kp = KplusMG(hier["A"][0], f.block_rowstart, hier, R=f.R, rtol=rtol, spmv=spmv)
    B, Gs = f.B.tocsr(), G.tocsr()
    Bt, Gt = B.T.tocsr(), Gs.T.tocsr()
    GGt_inv = None if orth else np.linalg.inv((Gs @ Gt).toarray())
    rho = 1.0  # any positive penalty: the cost per iteration is the same
    stamps = []

    def Q(v):
        w = Gs @ v
        return Gt @ (w if orth else GGt_inv @ w)

    def A_rho(v):
        pv = v - Q(v)
        y = B @ Kp(Bt @ pv)
        y = y - Q(y)
        y = y + rho * (Gt @ (Gs @ v))
        stamps.append(time.perf_counter())
        return y

    n = f.n_lambda
    op = O.Op(n, fn=A_rho)
    # size the sample: one probe application of A_rho, then as many MPGP iterations (<= its, >= 2) as fit the time budget
    A_rho(b_dual)  # warm-up (page faults, thread pool)
    t0 = time.perf_counter()
    A_rho(b_dual)
    t_probe = time.perf_counter() - t0
    its = int(max(3, min(its, budget_s / (2.1 * t_probe))))
    del stamps[:]
    ref = O.mpgp(op, b_dual, np.zeros(n), O.Box(n, lb=lb_dual), maxeig=1.0 + rho, max_it=its)  # stops once iteration > max_it - 1 ... `its` iterations
    # the first application is the initial gradient (set-up of the solve); the iterations own the rest
    per_apply = np.diff(np.asarray(stamps))
    applies_per_it = len(per_apply) / max(1, ref["iteration"])
    med = float(np.median(per_apply)) * applies_per_it
    return {
        "value": 1.0 / med, "unit": "QPS iterations/s", "cores": cores, "kind": "port", "extrapolated": False, "cpu_model": cpu_model(),
        "sample_short": "%d MPGP iterations of the same dual QP, MEASURED on the host: oracle MPGP (reference op order) with the iterative K^+ (block CG + V-cycle) restated on the CPU, OpenMP CSR products on %d threads" % (ref["iteration"], cores),
        "sample": "%d MPGP iterations (oracle/permon_oracle.c, the reference's op order; %.2f Hessian applications each, median application %.2f s => %.1f s per iteration) of the same TFETI dual QP on the host: "
                  "F = B K^+ B' with the GPU's own ITERATIVE K^+ restated on the CPU (oracle/mg_host.py: block-wise CG preconditioned by the same %d-level V-cycle, "
                  "fp64, rtol %.0e, %d CG iterations per application), sparse products by the OpenMP CSR kernel on %d threads (cgroup quota of the box), vectors in numpy; "
                  "%d K products in the sample.  The reference's own K^+ (sparse Cholesky) cannot be built here (no sparse direct solver on the image)"
                  % (ref["iteration"], applies_per_it, float(np.median(per_apply)), med, len(hier["A"]), rtol, Kp.last_its, cores, Kp.n_spmv),
    }


def cpu_baseline_whole_iteration(pool, ctx, f, G0, b_dual, lb_dual, budget_s=12.0, its_cap=40):
    """The reference's op sequence for a TFETI contact QP timed LIVE on the host cores, whole iteration: the oracle's SMALXE + MPGP in the
    reference's unfused operation order (oracle/permon_oracle.c: mpgp.c:511-641, smalxe.c:893-997) on A = P F P (qptransform.c:273-284), F = B K^+ B'
    (qptransform.c:1103-1128) with B / B' as CSR products (gluing.c:47-159), the projector with the dense (G G')^{-1} (qppf.c:454-605) and the reference's DEFAULT K^+:
    a sparse direct factorisation of K_reg = MatRegularize(K, R) per block and one forward / backward substitution per block and application (matinv.c:481-580,
    :734-743) -- SuperLU standing in for PETSc Cholesky / MUMPS, the blocks dealt to `pool.nw` worker processes (the ranks of the reference's run), each holding its own
    factors.  The blocks are congruent: every worker factors the ONE block matrix.  Sample: ONE outer SMALXE iteration with the inner solve capped so that it fits the budget;
    value = inner iterations / wall time of QPSSolve (solve phase only -- the factorisation and QPSSetUp with its power method are reported separately)."""
    import scipy.sparse as sp

    import permon_amd as pa
    from oracle import oracle as O

    if not f.congruent:
        raise ValueError("the live CPU leg factors ONE block matrix: congruent blocks only")
    nb, n_i, n = f.nsub, f.n_i, f.n_lambda
    Kreg, _piv, _rho = pa.MatRegularize(ctx, f.Ki, f.R[:, :n_i])
    t0 = time.perf_counter()
    fnnz = pool.factor(Kreg)
    t_fac = time.perf_counter() - t0
    pool.attach(n_i, nb)
    B = f.B.tocsr()
    Bt = B.T.tocsr()
    G = G0.tocsr()
    pf = O.Qppf(O.Csr.from_scipy(G), orthonormal=False)
    stamps = {"kplus": 0.0, "applies": 0}

    def F(v):
        pool.X[:, :] = (Bt @ v).reshape(nb, n_i)
        t1 = time.perf_counter()
        pool.solve()
        stamps["kplus"] += time.perf_counter() - t1
        stamps["applies"] += 1
        return B @ pool.Y.reshape(-1)

    def A(v):
        return pf.P(F(pf.P(v)))

    op = O.Op(n, fn=A)
    A(b_dual)  # warm-up (page faults, worker caches)
    t0 = time.perf_counter()
    A(b_dual)
    t_probe = time.perf_counter() - t0
    its = int(max(3, min(its_cap, budget_s / (2.2 * t_probe))))
    tm = {}
    stamps["kplus"], stamps["applies"] = 0.0, 0
    ref = O.smalxe(op, b_dual, np.zeros(n), O.Box(n, lb=lb_dual), pf, max_it=1, inner_opts=dict(max_it=its), maxeig_iter=10, timing=tm)  # (10 power iterations: the set-up is not what is measured)
    wall = tm["solve_seconds"]
    inner = max(1, int(ref["inner_iter_accu"]))
    return {
        "value": inner / wall, "unit": "QPS iterations/s", "cores": pool.nw, "kind": "port", "extrapolated": False, "cpu_model": cpu_model(), "measured": "live, this run",
        "kplus": "sparse direct (the reference's algorithm): SuperLU of K_reg per worker process", "inner_iterations": inner, "operator_applies_incl_setup": stamps["applies"], "solve_seconds": wall, "setup_seconds": tm["setup_seconds"],
        "kplus_seconds_incl_setup": stamps["kplus"], "factor_seconds_per_worker": t_fac, "factor_entries": int(fnnz),
        "sample": "WHOLE iteration, measured live (solve phase only): the oracle's SMALXE + MPGP (oracle/permon_oracle.c, the reference's unfused op order) for %d inner iterations (one outer update; %d applications of A = P F P incl. the set-up's power method) "
                  "of this block's dual QP on the host: F = B K^+ B' with CSR B / B' (scipy), the projector with the dense (G G')^{-1} (%d x %d), K^+ = the reference's direct solve -- SuperLU (scipy splu, stand-in "
                  "for PCCHOLESKY / MUMPS, matinv.c:481-580, :734-743) of K_reg, %d blocks dealt to %d worker processes (one factorisation each, %.1f s, not in the figure), vectors through shared memory; "
                  "QPSSolve %.2f s (QPSSetUp %.2f s apart), %.0f %% of both inside the K^+ solves; %s" % (inner, stamps["applies"], G.shape[0], G.shape[0], nb, pool.nw, t_fac, wall, tm["setup_seconds"],
                                                                                                   100.0 * stamps["kplus"] / (wall + tm["setup_seconds"]), cpu_model()),
        "sample_short": "live: oracle SMALXE+MPGP, %d inner its, direct K^+ (SuperLU) on %d worker processes, B/B'/projector included" % (inner, pool.nw),
    }


def cpu_baseline_direct(ctx, nel_full, nel_factor, applies_per_step, nblocks=8):
    """The reference's own K^+ on the host: MATINV factors K_reg = MatRegularize(K, R) once per block (PCCHOLESKY / MUMPS, src/mat/impls/inv/matinv.c:481-580)
    and every F = B K^+ B' application is one forward / backward substitution per block (MatMult_Inv -> KSPSolve with KSPPREONLY, matinv.c:734-743), one block
    per MPI rank.  PETSc and MUMPS are absent, so the sparse direct solver of the image stands in: scipy.sparse.linalg.splu (SuperLU, minimum-degree ordering on
    A' + A, symmetric mode, no pivoting) on K_reg from the library's own MatRegularize.  The 43^3 block (255 552 dof, ~1.3e9 factor entries, ~10 GB) would take
    ~20 min to factor, so two smaller congruent cubes are factored, the solve time per block is measured and extrapolated to the full block with the measured
    exponent of the growth in n (said in `sample`).  The 8 blocks are independent: 8 ranks on 8 cores are assumed to solve them in parallel without
    interference; B / B' and the dual-space vector work are not counted."""
    import scipy.sparse.linalg as spla

    import permon_amd as pa

    sizes = sorted({max(5, (nel_factor * 11) // 21), max(7, (nel_factor * 5) // 7), nel_factor})  # 21 -> 11, 15, 21
    rows = []
    for nel in sizes:
        g = pa.CubeFeti((1, 1, 1), nel, contact=False)
        Kreg, piv, rho = pa.MatRegularize(ctx, g.Ki, g.R)
        t0 = time.perf_counter()
        lu = spla.splu(Kreg.tocsc(), permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
        t_fac = time.perf_counter() - t0
        rhs = np.random.default_rng(3).standard_normal(Kreg.shape[0])
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            x = lu.solve(rhs)
            ts.append(time.perf_counter() - t0)
        res = float(np.linalg.norm(Kreg @ x - rhs) / np.linalg.norm(rhs))
        rows.append(dict(nel=nel, n=int(Kreg.shape[0]), factor_s=t_fac, solve_s=float(np.median(ts)), factor_nnz=int(lu.L.nnz + lu.U.nnz), residual=res))
        del lu
    n_full = 3 * (nel_full + 1) ** 3
    b = rows[-1]
    _DIRECT_ROWS[:] = rows
    # The full block size was factored ONCE on this pool's host (round 5, scripts/cpu_splu_full.py: 484 s and 24 GB for the 43^3 block -- not something a bench run can repeat)
    # together with the largest of the sizes this run factors: the committed full-size solve time is carried to THIS host by the ratio of the two measurements of that size.
    once = None
    try:
        full_m = json.load(open(os.path.join(ROOT, "profiles", "r05_splu_%d.json" % nel_full)))
        cal_m = json.load(open(os.path.join(ROOT, "profiles", "r05_splu_%d.json" % b["nel"])))
        if full_m["n"] == n_full and cal_m["n"] == b["n"] and nel_full != b["nel"]:
            ratio = b["solve_s"] / cal_m["solve_seconds_median"]
            once = dict(t_solve_full=full_m["solve_seconds_median"] * ratio, ratio=ratio, full=full_m, cal=cal_m)
    except (OSError, ValueError, KeyError):
        once = None

    def fit(key, default):  # least-squares slope of log t over log n through the measured sizes
        pts = [(math.log(r["n"]), math.log(r[key])) for r in rows if r[key] > 0]
        if len(pts) < 2:
            return default
        mx, my = sum(p_[0] for p_ in pts) / len(pts), sum(p_[1] for p_ in pts) / len(pts)
        return sum((p_[0] - mx) * (p_[1] - my) for p_ in pts) / sum((p_[0] - mx) ** 2 for p_ in pts)
    expo, expo_f = fit("solve_s", 4.0 / 3.0), fit("factor_s", 2.0)
    t_solve_full = b["solve_s"] * (n_full / b["n"]) ** expo if nel_full != b["nel"] else b["solve_s"]
    t_fac_full = b["factor_s"] * (n_full / b["n"]) ** expo_f if nel_full != b["nel"] else b["factor_s"]
    if once is not None:
        fm, cm = once["full"], once["cal"]
        return {
            "value": 1.0 / (applies_per_step * once["t_solve_full"] * math.ceil(nblocks / max(1, min(nblocks, host_threads())))), "unit": "QPS iterations/s", "cores": min(nblocks, host_threads()), "kind": "port", "cpu_model": cpu_model(),
            "kplus": "sparse direct (the reference's algorithm)",
            # NOT a same-run measurement: the full-size solve time was measured once on another box of this pool and is carried here by a calibration ratio; only the K^+ solves are counted
            "extrapolated": True, "calibrated_from": "profiles/r05_splu_%d.json x (this run's %d^3 solve / the committed %d^3 solve)" % (nel_full, b["nel"], b["nel"]), "counts": "K+ solves only (B / B', projector and vector work not counted; perfect concurrency of the blocks assumed)",
            "measured_once": "profiles/r05_splu_%d.json" % nel_full, "sizes_measured": [r["nel"] for r in rows] + [nel_full],
            "solve_seconds_per_block_full_size": once["t_solve_full"], "solve_seconds_per_block_full_size_as_measured": fm["solve_seconds_median"], "factor_seconds_full_size_measured": fm["factor_seconds"],
            "factor_entries_full_size": fm["factor_nnz"], "factor_max_rss_GB": fm["max_rss_GB"], "calibration_ratio_this_host": once["ratio"],
            "solve_seconds_per_block_measured": {("%d^3" % r["nel"]): round(r["solve_s"], 4) for r in rows}, "factor_seconds_measured": {("%d^3" % r["nel"]): round(r["factor_s"], 2) for r in rows},
            "model_from_small_sizes": {"solve_seconds_per_block": t_solve_full, "factor_seconds": t_fac_full, "growth_exponent_solve": expo, "growth_exponent_factor": expo_f},
            "sample_short": "reference's direct K^+: splu (SuperLU) of K_reg per block, 1 fwd/bwd solve per block and F apply; %d^3 block MEASURED once (factor %.0f s, solve %.3f s, %s), x %.2f from this run's %d^3 solve; %.2f applies/it, %d blocks on %d cores"
                            % (nel_full, fm["factor_seconds"], fm["solve_seconds_median"], "profiles/r05_splu_%d.json" % nel_full, once["ratio"], b["nel"], applies_per_step, nblocks, nblocks),
            "sample": "the reference's direct K^+ (matinv.c:481-580, :734-743): sparse factorisation of K_reg = MatRegularize(K, R) per subdomain, one forward/backward substitution per block and F application; "
                      "scipy.sparse.linalg.splu (SuperLU, MMD on A'+A, symmetric mode) stands in for PETSc Cholesky / MUMPS.  The %d^3 block (n = %d) was factored ONCE on this pool's host (%s: factor %.0f s, %.3g factor entries, "
                      "%.0f GB, solve %.4f s, residual %.1e; scripts/cpu_splu_full.py) -- not repeatable inside a bench run -- together with the %d^3 block (solve %.4f s), which THIS run factors again (solve %.4f s): the full-size "
                      "solve time is carried over by that ratio (%.2f): %.4f s per block and application.  Factored in this run: %s.  x %.2f F applications per QPS iteration (the GPU run's own mix), %d subdomain blocks on %d cores "
                      "in parallel (one block per rank, as the reference runs), B / B' and dual-space vector work not counted"
                      % (nel_full, n_full, fm.get("cpu_model"), fm["factor_seconds"], fm["factor_nnz"], fm["max_rss_GB"], fm["solve_seconds_median"], fm["residual"], b["nel"], cm["solve_seconds_median"], b["solve_s"],
                         once["ratio"], once["t_solve_full"], "; ".join("%d^3 (n = %d): factor %.1f s, solve %.4f s" % (r["nel"], r["n"], r["factor_s"], r["solve_s"]) for r in rows), applies_per_step, nblocks, nblocks),
        }
    return {
        "value": 1.0 / (applies_per_step * t_solve_full * math.ceil(nblocks / max(1, min(nblocks, host_threads())))), "unit": "QPS iterations/s", "cores": min(nblocks, host_threads()), "kind": "port", "cpu_model": cpu_model(),
        "counts": "K+ solves only (B / B', projector and vector work not counted; perfect concurrency of the blocks assumed)",
        "solve_seconds_per_block_measured": {("%d^3" % r["nel"]): round(r["solve_s"], 4) for r in rows}, "factor_seconds_measured": {("%d^3" % r["nel"]): round(r["factor_s"], 2) for r in rows},
        "solve_seconds_per_block_full_size": t_solve_full, "factor_seconds_full_size_extrapolated": t_fac_full, "growth_exponent_solve": expo, "growth_exponent_factor": expo_f,
        "extrapolated": nel_full != b["nel"], "sizes_measured": [r["nel"] for r in rows],
        "sample": "the reference's direct K^+ (matinv.c:481-580, :734-743): sparse factorisation of K_reg = MatRegularize(K, R) per subdomain, one forward/backward substitution per block and F application; "
                  "scipy.sparse.linalg.splu (SuperLU, MMD on A'+A, symmetric mode) stands in for PETSc Cholesky / MUMPS.  Factored on this host: %s; residuals %s.  "
                  "%s  x %.2f F applications per QPS iteration (the GPU run's own mix), %d subdomain blocks on %d cores in parallel (one block per rank, as the reference runs), B / B' and dual-space vector work not counted"
                  % ("; ".join("%d^3 elements (n = %d): factor %.1f s, %.3g factor entries, solve %.4f s" % (r["nel"], r["n"], r["factor_s"], r["factor_nnz"], r["solve_s"]) for r in rows),
                     ", ".join("%.1e" % r["residual"] for r in rows),
                     ("The %d^3 block (n = %d) is EXTRAPOLATED from these with the measured growth of the solve time, n^%.2f: %.2f s per block and application (factorisation n^%.2f: ~%.0f s, not run)."
                      % (nel_full, n_full, expo, t_solve_full, expo_f, t_fac_full)) if nel_full != b["nel"] else "Measured at the full block size.",
                     applies_per_step, nblocks, nblocks),
    }


def pmc_lookup(prefix, fname, combine="mean", contains=None):
    """(HBM bytes per launch, provenance) of a kernel from a committed rocprofv3 PMC pass (profiles/<fname>, written by
    scripts/gpu_pmc*.sh with the git state it measured).  (None, reason) when the file or the kernel is missing: the line then
    carries no traffic figure rather than a stale one."""
    # the newest committed pass of that name wins (profiles/r05_* over r04_*: the callers name the round-4 file the figure first came from)
    for newer in ("r05_" + fname[4:], "r06_" + fname[4:]) if fname.startswith("r04_") else ():
        if os.path.exists(os.path.join(ROOT, "profiles", newer)):
            fname = newer
    path = os.path.join(ROOT, "profiles", fname)
    try:
        pmc = json.load(open(path))
    except (OSError, ValueError) as ex:
        return None, "no PMC pass: %r" % (ex,)
    prefixes = (prefix,) if isinstance(prefix, str) else tuple(prefix)
    keep = lambda k: k != "_meta" and (contains is None or any(c in k for c in contains))
    meta = pmc.get("_meta", {})
    if combine == "sum":
        # the kernels of ONE operation, launched once each: every element of `prefix` (a name prefix, or a tuple of alternative prefixes for the
        # instantiations of one kernel) must be in the file -- a partial sum would be a wrong figure, not a missing one
        total = 0.0
        for e in prefixes:
            alts = (e,) if isinstance(e, str) else tuple(e)
            hits = [v for k, v in pmc.items() if keep(k) and k.startswith(alts)]
            if not hits:
                return None, "kernel %r not in profiles/%s" % (alts, fname)
            total += sum(v["hbm_bytes_per_launch"] * v["launches"] for v in hits) / sum(v["launches"] for v in hits)
    else:
        flat = tuple(a for e in prefixes for a in ((e,) if isinstance(e, str) else tuple(e)))
        hits = [v for k, v in pmc.items() if keep(k) and k.startswith(flat)]
        if not hits:
            return None, "kernel %r not in profiles/%s" % (flat, fname)
        total = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in hits) / sum(v["launches"] for v in hits)  # launch-weighted mean over the instantiations of one kernel
    return (total,
            "profiles/%s @ %s (%s)" % (fname, meta.get("git", "git state not recorded"), meta.get("command", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH_SIZE doubled")))


def dual_spmv_hbm(ctx, f, reps=20):
    """The north star's "FETI dual SpMV" on HBM at the size of the run: MatMult_BlockDiag (matblockdiag.c:190-201) y_i = K_i x_i over the subdomain blocks with DISTINCT
    K_i (K_s = E_s K_1, E_s = 1, 1.25, ...: no two equal, so nothing can be shared or served from L2) -- (a) the CSR kernel on SURVEY 8d's bytes 12 nnz + 20 n, (b) the
    3x3-block kernel with one device copy per block on the bytes it stores, both also on the CSR figure .  HIP-event pairs
    around every launch on the launch stream (pmh_blockdiag_timing_*)."""
    import permon_amd as pa

    nsub, n_i, Ki = f.nsub, f.n_i, f.Ki.tocsr()
    Ki.sort_indices()
    nnz_i = Ki.nnz
    ip = np.empty(nsub * n_i + 1, dtype=np.int32)
    ci = np.empty(nsub * nnz_i, dtype=np.int32)
    va = np.empty(nsub * nnz_i, dtype=np.float64)
    ip[0] = 0
    for s_ in range(nsub):
        ip[s_ * n_i + 1:(s_ + 1) * n_i + 1] = Ki.indptr[1:] + s_ * nnz_i
        ci[s_ * nnz_i:(s_ + 1) * nnz_i] = Ki.indices + s_ * n_i
        np.multiply(Ki.data, 1.0 + 0.25 * s_, out=va[s_ * nnz_i:(s_ + 1) * nnz_i])
    n = nsub * n_i
    A = pa.CsrMat(ctx, n, n, ip, ci, va)
    del ip, ci, va
    K = pa.MatBlockDiag(ctx, np.arange(nsub + 1, dtype=np.int64) * n_i, A)
    x, y = ctx.vec_from(np.random.default_rng(11).standard_normal(n)), ctx.vec(n)

    def timed():
        for _ in range(3):
            K.mult(x, y)
        K.timing_enable(reps + 8)
        for _ in range(reps):
            K.mult(x, y)
        ctx.sync()
        r = K.timing_get()
        K.timing_enable(0)
        return r

    out = {"what": "MatMult_BlockDiag y_i = K_i x_i, %d subdomain blocks with DISTINCT matrices (K_s = E_s K_1; one device copy each, every byte from HBM), n = %d, nnz = %d" % (nsub, n, nsub * nnz_i),
           "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "launches_timed": reps}
    nl, ms, csr_b, hbm_b, cp = timed()
    y_csr = y.to_numpy().copy()
    out["csr"] = {"kernel": "k_spmv_stream (row-blocked CSR: coalesced tile loads, LDS partial sums, wavefront shuffle reduction)", "avg_launch_ms": ms / nl, "algorithmic_bytes_per_launch": csr_b,
                  "bytes_note": "SURVEY 8d: 12 nnz + 20 n", "achieved": csr_b / (ms / nl) / 1e6, "frac": csr_b / (ms / nl) / 1e6 / HBM_PEAK_GBS, "device_copies": cp}
    try:
        K.enable_bsr3(share=False)
        nl, ms, csr_b, hbm_b, cp = timed()
        dev = float(np.max(np.abs(y.to_numpy() - y_csr)) / max(np.max(np.abs(y_csr)), 1e-300))
        out["bsr3"] = {"kernel": "k_bsr3<double> (3x3 blocks, 8.44 B per non-zero stored)", "avg_launch_ms": ms / nl, "algorithmic_bytes_per_launch": hbm_b, "bytes_note": "the bytes the 3x3-block copy stores (76 B per block + pointers) + x + y",
                       "achieved": hbm_b / (ms / nl) / 1e6, "frac": hbm_b / (ms / nl) / 1e6 / HBM_PEAK_GBS, "csr_equivalent_GBs": csr_b / (ms / nl) / 1e6, "csr_equivalent_frac": csr_b / (ms / nl) / 1e6 / HBM_PEAK_GBS,
                       "device_copies": cp, "max_rel_diff_vs_csr_kernel": dev}
    except RuntimeError as ex:
        out["bsr3"] = {"failed": repr(ex)}
    x.free(), y.free()
    K.destroy()
    A.destroy()
    # SURVEY 8d names B and B' beside K_i ("each measured with its own B_spmv"): MatMult_Gluing / MatMultTranspose_Gluing (gluing.c:47-159) of the run's own gluing -- 1-2 entries per
    # dual row, a few MB in all: LATENCY-bound launches, reported as what they are (the K_i product above carries the roofline target)
    try:
        B = pa.MatGluing(ctx, f.N, f.n_lambda, f.leaves_row, f.leaves_root, f.leaves_sign)
        lam, xx = ctx.vec_from(np.random.default_rng(12).standard_normal(f.n_lambda)), ctx.vec(f.N)
        nleaf = int(np.size(f.leaves_row))
        res = {}
        for name, fn, rows in (("Bt_lambda (MatMult_Gluing)", lambda: B.mult(lam, xx), f.N), ("B_u (MatMultTranspose_Gluing)", lambda: B.mult_transpose(xx, lam), f.n_lambda)):
            for _ in range(5):
                fn()
            ctx.sync()
            ctx.timer_start()
            for _ in range(50):
                fn()
            ms_b = ctx.timer_stop() / 50
            b_spmv = 12.0 * nleaf + 20.0 * rows
            res[name] = {"avg_ms": ms_b, "algorithmic_bytes_per_launch": b_spmv, "achieved_GBs": b_spmv / ms_b / 1e6, "frac": b_spmv / ms_b / 1e6 / HBM_PEAK_GBS}
        out["gluing"] = dict(res, note="n_lambda = %d, %d leaves.  B' lambda writes the whole primal vector (%.1f MB by SURVEY's 12 nnz + 20 rows: a bandwidth-bound launch); B u gathers 1-2 entries per dual row "
                                       "(%.1f MB: launch latency, not bandwidth).  The K_i product moves 45 x / 470 x these bytes and carries the roofline target" % (f.n_lambda, nleaf, (12.0 * nleaf + 20.0 * f.N) / 1e6, (12.0 * nleaf + 20.0 * f.n_lambda) / 1e6))
        lam.free(), xx.free()
        B.destroy()
    except Exception as ex:  # noqa: BLE001
        out["gluing"] = {"failed": repr(ex)}
    # headline of the block: the CSR kernel on the CSR bytes (what the metric names)
    full = nsub == 8 and f.nel == 43
    t_csr, src_csr = pmc_lookup("void k_spmv_stream<0, 2048", "r04_pmc_traffic_dual_spmv.json") if full else (None, "not the configuration of the committed PMC pass")
    t_bsr, src_bsr = pmc_lookup("void k_bsr3<double", "r04_pmc_traffic_dual_spmv.json") if full else (None, "not the configuration of the committed PMC pass")
    out["csr"]["traffic"], out["csr"]["traffic_source"] = t_csr, src_csr
    if "failed" not in out["bsr3"]:
        out["bsr3"]["traffic"], out["bsr3"]["traffic_source"] = t_bsr, src_bsr
    out.update({"kernel": "k_spmv_stream", "achieved": out["csr"]["achieved"], "frac": out["csr"]["frac"], "algorithmic_bytes_per_launch": out["csr"]["algorithmic_bytes_per_launch"],
                "avg_launch_ms": out["csr"]["avg_launch_ms"], "traffic": t_csr, "traffic_source": src_csr})
    return out


def run_feti(ctx, a, steps, warmup, rank, world, dist):
    import scipy.sparse as sp

    import permon_amd as pa
    from permon_amd.chain import FetiDualQP

    t0 = time.time()
    sub = tuple(int(v) for v in a.sub.split(","))
    nsub = sub[0] * sub[1] * sub[2]
    orth = not a.dense_coarse
    young = None
    if getattr(a, "young", ""):
        young = [1.0 + 0.25 * i for i in range(nsub)] if a.young == "distinct" else [float(v) for v in a.young.split(",")]
    partition = getattr(a, "partition", "")
    if partition:  # the (2 nel)^3-element cube re-cut into 8 subdomains that are NOT boxes (staircase interfaces / L-shapes): no congruence, no symmetry, no box hierarchy
        from permon_amd import feti as _feti

        if sub != (2, 2, 2):
            raise SystemExit("--partition cuts the 2,2,2 cube")
        f = _feti.MeshFeti(_feti.irregular_partition(a.nel, partition), contact=True, young=young)
        congruent = False
    else:
        f = pa.CubeFeti(sub, a.nel, contact=True, young=young)
        congruent = f.congruent
    implicit = orth and a.orth_form == "implicit"
    G, e = f.coarse(orthonormalize=orth and not implicit)  # implicit: G0 = R'B', e0; the library orthonormalises (pmh_qppf_create orthonormal = 2)
    if nsub % world:
        raise SystemExit("feti workload: the %d subdomains must divide over the ranks" % nsub)
    per = nsub // world
    if a.sim_world and world == 1:
        if nsub % a.sim_world:
            raise SystemExit("--sim-world must divide the number of subdomains")
        per = nsub // a.sim_world
    local = f.subset(range(rank * per, (rank + 1) * per))
    t_gen = time.time() - t0
    t0 = time.time()
    nn = a.nel + 1
    blocks = [f.block_K(rank * per + i) for i in range(per)]
    if a.regularize:  # MatRegularize per block (the kernel bases, hence the fixing DOFs, differ from block to block)
        from permon_amd.chain import regularize_blocks

        local["Kreg"] = regularize_blocks(ctx, local)[0]
        blocks = [local["Kreg"][i * f.n_i:(i + 1) * f.n_i, i * f.n_i:(i + 1) * f.n_i].tocsr() for i in range(per)]

    def make_hier(blks, nper):  # Galerkin hierarchy of the congruent cubes (host set-up, seconds)
        # depth by blocks per GPU: with 1-4 blocks the cycle is launch-latency bound, so it stops one level earlier (dense block
        # pseudo-inverse at ~5000 dof, one HBM-streaming launch instead of a smoothed level's seven) -- measured, profiles/
        auto_nodes = 2000 if nper <= (4 if a.mg_precision == "fp16" else 1) else 400
        auto_nodes = min(auto_nodes, nn ** 3 // 8)  # tiny test problems: at least one smoothed level
        return pa.box_mg_hierarchy(blks, [(nn, nn, nn)] * len(blks), 3, min_nodes=a.mg_min_nodes or auto_nodes)

    # the V-cycle hierarchy of the GPU solver is built inside the library (pmh_mg_create_box, host C++); the scipy builder is only
    # run where its output is needed on the host: the CPU baseline leg (rank 0 at N = 1) and --mg-builder python
    need_py_hier = a.kplus_pc == "mg" and not partition and (a.mg_builder == "python" or a.regularize or (world == 1 and not a.sim_world and not a.no_cpu_baseline))
    hier = make_hier(blocks, per) if need_py_hier else None
    use_c_builder = a.kplus_pc == "mg" and (a.mg_builder == "c" or bool(partition)) and not a.regularize
    # blocks that are not boxes: the algebraic hierarchy (pmh_mg_create_sa); tiny test problems: at least one smoothed level
    mg_sa = dict(ndof=3, max_coarse=min(3 * (a.mg_min_nodes or 500), max(6, int(np.diff(f.block_rowstart).min()) // 4)), theta=0.08) if (partition and use_c_builder) else None

    def mg_box(nblk, for_iterative=None, precision=None):
        # (explicit K^+: the rank's own inner-Krylov solver only serves a handful of set-up products -- d = B K^+ f, the replica's columns come from the replica solver --
        # so it does not pay for the 5000-dof dense coarse pseudo-inverse that makes the cycle of a 1-4 block rank faster: 6 s of host set-up at the 1/8 share)
        # Round 6: 8 CONGRUENT blocks run as the 8 columns of ONE block (matinv_mv.hip): the cycle then walks one block's levels, the 1-4 block regime -- the hierarchy stops a level
        # earlier there too (measured on the driver's window: 71.1 -> 76.0 it/s, one CG iteration less per application; +8 s of host set-up for the one dense pseudo-inverse)
        it = (a.kplus != "explicit") if for_iterative is None else for_iterative
        prec = precision or a.mg_precision
        one_block_regime = nblk <= (4 if prec == "fp16" else 1) or (nblk == 8 and congruent and prec != "fp64" and not a.no_bsr3)  # (the strict fp64 cycle runs on the one-column kernels: 8 blocks)
        auto_nodes = 2000 if (one_block_regime and it) else 400
        return dict(dims=[(nn, nn, nn)] * nblk, ndof=3, min_nodes=a.mg_min_nodes or min(auto_nodes, nn ** 3 // 8))
    explicit = None
    replica = {}
    if a.kplus == "explicit":
        def solver_factory(nslots):
            """A K^+ over `nslots` replicas of this rank's (congruent) block: the set-up solves of the explicit operators fill the GPU
            although the rank owns fewer blocks (1 at N = 8)."""
            from permon_amd.feti import csr_block_diag

            Krep = csr_block_diag([f.Ki] * nslots)
            Kb = pa.MatBlockDiag.from_scipy(ctx, np.arange(nslots + 1, dtype=np.int32) * f.n_i, Krep)
            Rb = np.tile(local["R"][:, :f.n_i], (1, nslots))
            M = pa.MatInv(Kb, rtol=a.explicit_rtol, max_it=20000, jacobi=True, nullspace=Rb)
            if not a.no_bsr3:
                M.enable_bsr3()
            if use_c_builder:
                mb = mg_box(nslots)
                M.set_pc_mg_box(Krep, mb["dims"], 3, R=Rb, min_nodes=mb["min_nodes"], degree=a.mg_degree, precision=a.mg_precision)
            elif a.kplus_pc == "mg":
                M.set_pc_mg(make_hier([f.Ki] * nslots, nslots), degree=a.mg_degree, precision=a.mg_precision)
            replica["M"], replica["K"] = M, Kb
            return M

        explicit = dict(rtol=a.explicit_rtol, storage=a.explicit_storage, min_slots=0 if (a.regularize or not congruent) else a.explicit_slots, solver_factory=None if (a.regularize or not congruent) else solver_factory)
        if not a.no_explicit_symmetry and not a.regularize and congruent:  # used by the class-shared storages only
            explicit["symmetry"] = dict(dims=(a.nel + 1,) * 3, ndof=3, orbit=a.explicit_storage in ("auto", "class_orbit"))
        elif partition:
            explicit["storage"] = "sym"  # nothing to share, nothing to find by symmetry: per-block symmetric tiles (k_fx_symv), every column by its own K^+ solve (8 per block at a time)
        elif not a.no_explicit_symmetry and not a.regularize and not congruent and a.explicit_storage in ("auto", "class_orbit"):  # (several GPUs: every rank its own blocks' classes -- owner computes)
            # cubes of different materials: one class per block, each on the CLOSURE of its touched set under the cube's group (the whole boundary): all 48 operations survive, one K^+
            # solve per orbit instead of one per touched dof, and the apply is the orbit GEMM (one per class) instead of the HBM-bound stream over every full W_b
            explicit["storage"] = "class_orbit"
            explicit["symmetry"] = dict(dims=(a.nel + 1,) * 3, ndof=3, orbit=True, close=True)
        nshare = world if world > 1 else a.sim_world
        if nshare > 1 and a.explicit_storage != "full" and not a.regularize and not a.no_stripe and congruent:
            # every cube is congruent: each rank takes an even share of 128-row stripes of ALL W_b (the blocks' n_Gamma differ by 1.43 x)
            explicit["stripe"] = (rank, nshare, dict(n_x=f.N, block_rowstart=f.block_rowstart, leaves_row=f.leaves_row, leaves_root=f.leaves_root, leaves_sign=f.leaves_sign))
    q = FetiDualQP(ctx, local, G, e, f.c, f.lb, orthonormal="implicit" if implicit else orth, kplus_rtol=a.kplus_rtol, mg_hierarchy=None if use_c_builder else hier, mg_box=mg_box(per) if (use_c_builder and not mg_sa) else None, mg_sa=mg_sa,
                   mg_degree=a.mg_degree, mg_precision=a.mg_precision, bsr3=not a.no_bsr3, regularize=a.regularize, explicit=explicit)
    has_mg = a.kplus_pc == "mg"

    def switch_mg(precision, for_iterative=None):
        """Replace the V-cycle of the inner KSP by one in another precision (same builder; for_iterative: the depth the inner-Krylov path takes, see mg_box)."""
        old = q.Kplus.mg
        if mg_sa:
            q.Kplus.set_pc_mg_sa(local["K"], 3, R=local["R"], max_coarse=mg_sa["max_coarse"], theta=mg_sa["theta"], degree=a.mg_degree, precision=precision)
        elif use_c_builder:
            mb = mg_box(per, for_iterative, precision)
            q.Kplus.set_pc_mg_box(local["K"], mb["dims"], 3, R=local["R"], min_nodes=mb["min_nodes"], degree=a.mg_degree, precision=precision)
        else:
            q.Kplus.set_pc_mg(hier, degree=a.mg_degree, precision=precision)
        old.destroy()
    if replica:  # the replica solver is set-up scaffolding: release it
        if getattr(replica["M"], "mg", None) is not None:
            replica["M"].mg.destroy()
        replica["M"].destroy()
        replica["K"].destroy()
        replica["K"].K.destroy()
    qps = q.make_smalxe()  # QPSSetUp_SMALXE: lambda_max(PFP) by the power method, rho, M1, inner MPGP
    t_setup = time.time() - t0

    def barrier():
        ctx.sync()
        if dist is not None:
            import torch

            dist.barrier()
            torch.cuda.synchronize()

    def roctx(resume):
        """PMH_BENCH_ROCTX=1 under `rocprofv3 --selected-regions`: only the timed region is traced (the set-up solves of the explicit
        operators alone are ~11 M launches)."""
        if not os.environ.get("PMH_BENCH_ROCTX"):
            return
        import ctypes

        lib = ctypes.CDLL("/opt/rocm/lib/librocprofiler-sdk-roctx.so")
        (lib.roctxProfilerResume if resume else lib.roctxProfilerPause)(ctypes.c_uint64(0))

    collective = {}

    def timed_pass(nsteps, nwarm):
        """W untimed + exactly K timed inner MPGP iterations of the real SMALXE solver loop (restarting from lambda = 0 whenever the
        solve converges: configs[2] takes 108 iterations), bracketed by barriers; max over ranks."""
        if nwarm:
            qps.RunFixedSolve(nwarm)
        if dist is not None and want_timing:
            import ctypes as _C

            pa._lib.check(ctx.L.pmh_comm_timing_enable(ctx.h, 4 * nsteps + 64))  # event pairs around the all-reduce that ends every B u
        barrier()
        roctx(True)
        t1 = time.perf_counter()
        cnt = qps.RunFixedSolve(nsteps)
        barrier()
        dt = time.perf_counter() - t1
        roctx(False)
        if dist is not None:
            import torch

            dt = dist_max(dist, dt)
            if want_timing:
                import ctypes as _C

                nc, msc, byc = _C.c_int(), _C.c_double(), _C.c_double()
                pa._lib.check(ctx.L.pmh_comm_timing_get(ctx.h, _C.byref(nc), _C.byref(msc), _C.byref(byc)))
                pa._lib.check(ctx.L.pmh_comm_timing_enable(ctx.h, 0))
                if nc.value:
                    collective.update({"allreduces_timed": nc.value, "ms_per_allreduce": msc.value / nc.value, "bytes_per_allreduce": byc.value / nc.value,
                                       "ms_per_allreduce_max_over_ranks": dist_max(dist, msc.value / nc.value), "share_of_step_time": msc.value * 1e-3 / dt,
                                       "transport": "host (gloo, TEST MODE)" if HOST_TRANSPORT else "RCCL"})
        return dt, cnt

    want_timing = not os.environ.get("PMH_BENCH_NO_TIMING")
    full_size = (a.nel == 43 and a.sub == "2,2,2" and world == 1 and not a.sim_world and congruent and not partition)
    kreg_text = " on K_reg = MatRegularize(K, R)" if a.regularize else ""
    pc_text = ("multigrid-preconditioned CG (Galerkin V-cycle in %s built by %s, dense coarse solve at <= %d nodes per block, Chebyshev(%d)/Jacobi smoothing)"
               % (a.mg_precision, ("pmh_mg_create_sa (smoothed aggregation)" if mg_sa else "pmh_mg_create_box") if use_c_builder else "permon_amd.feti.box_mg_hierarchy", (mg_sa["max_coarse"] // 3) if mg_sa else mg_box(per)["min_nodes"], a.mg_degree)) if has_mg else "Jacobi-CG"

    def iterative_pass(nsteps, nwarm, precision):
        """The inner-Krylov K^+ (the reference's iterative MATINV path): block-wise CG with the V-cycle PC in `precision`."""
        os.environ.setdefault("PMH_TIMING_STRIDE", "5")  # event pairs around every 5th K x launch (coprime to the 4 fine-level launches of a cycle)
        if want_timing:
            q.Kplus.timing_enable(8000)
            if has_mg:
                q.Kplus.mg.timing_enable(8000)
        _, spmv1 = q.Kplus.last_iterations()
        mgs1 = q.Kplus.mg.fine_spmv() if has_mg else 0
        dt, cnt = timed_pass(nsteps, nwarm)
        kits, spmv2 = q.Kplus.last_iterations()
        mgs2 = q.Kplus.mg.fine_spmv() if has_mg else 0
        stride = int(os.environ.get("PMH_TIMING_STRIDE", "1"))
        n_cg, ms_cg, b_cg = q.Kplus.timing_get() if want_timing else (0, 0.0, 0.0)
        cg_GBs = b_cg / (ms_cg / n_cg * 1e-3) / 1e9 if n_cg else 0.0
        kname = ("k_bsr3<double>: the fp64 K x of the block CG inside K^+ = the FETI dual SpMV (3x3 blocks, 8.44 B per non-zero)" if not a.no_bsr3
                 else "k_spmv_stream<plain, 2048-nnz tile, 8 lanes/row> on blockdiag(K_i): the FETI dual SpMV inside K^+")
        kpat = "void k_bsr3<double" if not a.no_bsr3 else "void k_spmv_stream<0, 2048,"
        mv_active = q.Kplus.multi_rhs_active()
        if mv_active:  # 8 congruent blocks = the 8 columns of one block: the product streams ONE ELL copy of K_1 from HBM and serves 8 vectors
            kname = "k_mv_spmv<double, 8 columns>: the fp64 K x of the block CG inside K^+ on interleaved multivectors (ELL 3x3 blocks, 8 congruent blocks as the 8 columns of one)"
            kpat = "void k_mv_spmv<double"
        traffic, tsrc = pmc_lookup(kpat, "r04_pmc_traffic_feti_iterative.json") if full_size else (None, "not the configuration of the committed PMC pass")
        nrep = q.Kplus.bsr3_replicas() if not a.no_bsr3 else 1
        roof = {"bound": "hbm", "kernel": kname, "achieved": cg_GBs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": cg_GBs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc,
                "algorithmic_bytes_per_launch": b_cg, "launches_timed": n_cg, "avg_launch_ms": ms_cg / n_cg if n_cg else None, "timing_stride": stride,
                "share_of_step_time": (ms_cg * 1e-3) * stride / dt if n_cg else None, "blocks_per_device_copy": nrep}
        if mv_active:
            roof["blocks_per_device_copy"] = 8
            roof["note"] = "8 congruent blocks run as the 8 columns of ONE block (matinv_mv.hip): bytes = the ELL copy of K_1 once + the 8 x and 8 y (HBM)"
        elif nrep > 1:  # congruent blocks share ONE device copy: achieved / frac are on what HBM delivers; the block-diagonal figure (every K_i counted) is an L2-served rate
            roof["bound"] = "l2"  # (the kernel is NOT at `frac` of HBM: most of what it reads comes from the XCDs' L2; the HBM-streaming form is the feti_dual_spmv block / PMH_BSR_NO_SHARE=1)
            b_bd = nrep * (b_cg - 16.0 * local["n_x"]) + 16.0 * local["n_x"]
            roof["blockdiag_figure_bytes"] = b_bd
            roof["blockdiag_figure_GBs"] = b_bd / (ms_cg / n_cg * 1e-3) / 1e9 if n_cg else None
            roof["note"] = "%d congruent blocks share one device copy of K_i: bytes = that copy once + x + y (HBM); blockdiag_figure_* counts every K_i (SURVEY 8d) and is served by the XCDs' L2, not an HBM rate" % nrep
        if has_mg and want_timing:
            n_k, ms_k, b_k = q.Kplus.mg.timing_get()
            pk = b_k / (ms_k / n_k * 1e-3) / 1e9 if n_k else 0.0
            ppat = {"fp16": ("void k_bsr3<_Float16", "_Z6k_bsr3IDF16_"), "fp32": "void k_bsr3<float", "fp64": "void k_bsr3<double"}[precision]
            # the fine-level instantiations only (tile size 1024; the second level runs the 512 ones)
            ptraffic, ptsrc = pmc_lookup(ppat, "r04_pmc_traffic_feti_iterative.json", contains=("Li1024E", ", 1024>")) if (full_size and precision != "fp64") else (None, "fp64 cycle: same kernel as the CG product" if precision == "fp64" else "not the configuration of the committed PMC pass")
            roof["preconditioner"] = {"kernel": "k_bsr3<%s>: fine-level K x of the V-cycle (%s B per non-zero)" % {"fp16": ("_Float16 entries, float vectors", "2.44"), "fp32": ("float", "4.44"), "fp64": ("double", "8.44")}[precision],
                                      "achieved": pk, "frac": pk / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": b_k, "launches_timed": n_k, "avg_launch_ms": ms_k / n_k if n_k else None,
                                      "share_of_step_time": (ms_k * 1e-3) * stride / dt if n_k else None, "traffic": ptraffic, "traffic_source": ptsrc}
        q.Kplus.timing_enable(0)
        if has_mg:
            q.Kplus.mg.timing_enable(0)
        return {"value": nsteps / dt, "ms_per_step": dt / nsteps * 1e3, "steps": nsteps, "warmup": nwarm, "steps_by_type": cnt,
                "kplus": "block-wise %s%s (rtol %.0e)" % ((pc_text.replace("<= %d nodes" % mg_box(per)["min_nodes"], "<= %d nodes" % mg_box(per, True, precision)["min_nodes"]) if (has_mg and use_c_builder and not mg_sa) else pc_text).replace(a.mg_precision, precision), kreg_text, a.kplus_rtol),
                "cg_spmv_per_step": (spmv2 - spmv1) / max(nsteps + nwarm, 1), "vcycle_fine_spmv_per_step": (mgs2 - mgs1) / max(nsteps + nwarm, 1), "last_block_cg_iterations": kits,
                "roofline": roof}

    extra = {}
    if a.kplus == "explicit":
        E = q.E
        if want_timing:
            E.timing_enable(4 * (steps + warmup) + 256, 1)
        if warmup:
            qps.RunFixedSolve(warmup)
        n_w, ms_w, _ = E.timing_get() if want_timing else (0, 0.0, 0.0)
        fk_w = getattr(E, "first_kernel_ms", 0.0) if want_timing else 0.0
        dt, cnt = timed_pass(steps, 0)
        n_k, ms_k, b_k = E.timing_get() if want_timing else (0, 0.0, E.gemv_bytes)
        ms_first = (getattr(E, "first_kernel_ms", 0.0) - fk_w) if want_timing else 0.0  # the first kernel of the dense apply alone (orbit storage: the GEMM; sym: k_fx_symv)
        n_k, ms_k = n_k - n_w, ms_k - ms_w  # the timed region only
        E.timing_enable(0)
        cnt["operator_applies"] = n_k  # F applies of the timed region: the inner Hessian multiplications + SMALXE's objective evaluation per outer iteration
        cnt["ms_per_operator_apply"] = dt * 1e3 / n_k if n_k else None
        achieved = b_k / (ms_k / n_k * 1e-3) / 1e9 if n_k else 0.0
        n_solves, asm_s = E.assemble_stats()
        storage_used = q.explicit_storage
        flops_k = E.apply_flops()
        ppref = {"class_orbit": (("void k_fxo_gemm16<",), "k_fxo_fin"), "class_sym": ("k_fxs_symm8", "k_fxs_symfin"), "class": ("k_fxs_gemm8", "k_fxs_fin"), "sym": ("void k_fx_symv<", "k_fx_symv_fin"), "full": ("void k_fx_gemv<",)}[storage_used]
        # the committed PMC passes (scripts/gpu_final_r04.sh): the headline, the configs[3] block, the general (non-congruent) block
        pmc_file = None
        if world == 1 and not a.sim_world:
            pmc_file = ("r04_pmc_traffic_feti_explicit.json" if (full_size and congruent) else
                        "r04_pmc_traffic_configs3.json" if (a.nel == 21 and a.sub == "4,4,4" and congruent) else
                        "r04_pmc_traffic_general.json" if (a.nel == 43 and a.sub == "2,2,2" and not congruent and not partition) else
                        "r06_pmc_traffic_nosym21.json" if (partition == "staircase" and a.nel == 21) else None)
        traffic, tsrc = pmc_lookup(ppref, pmc_file, combine="sum") if pmc_file else (None, "not the configuration of a committed PMC pass")
        roofline = {
            "bound": "hbm", "kernel": ("k_fxs_symm8 (+ k_fxs_symfin): Y = W_c X, ONE symmetric dense fp64 matrix W_c = (K^+)[U_c, U_c] per class of congruent blocks, kept as its lower block-triangle in 16x16 tiles "
                                       "(every stored byte read once) and applied to the blocks' vectors together: 8 right-hand sides per pass, both products of a tile (W_IJ X_J and W_IJ' X_I) on the fp64 matrix "
                                       "instruction v_mfma_f64_4x4x4_4b (4 flop per byte: 1/3 of the fp64 MFMA peak at the HBM rate, so still HBM-bound; the FETI dual operator apply, SURVEY 8d dense path)" if storage_used == "class_sym" else
                                       "k_fxs_gemm8 (+ k_fxs_fin): Y = W_c X, ONE full dense fp64 matrix W_c = (K^+)[U_c, U_c] per class of congruent blocks applied to the blocks' vectors together "
                                       "(8 right-hand sides per pass, the lane owns its output columns: no reduction across lanes; the FETI dual operator apply, SURVEY 8d dense path)" if storage_used == "class" else
                                       "k_fx_symv (+ k_fx_symv_fin): y_b = W_b x_b on the lower block-triangle of the symmetric dense fp64 local dual operators W_b = (K_b^+)[Gamma_b, Gamma_b], every stored byte read once, "
                                       "all blocks of the rank in one launch (the FETI dual operator apply, SURVEY 8d dense path)" if storage_used == "sym" else
                                       "k_fx_gemv: y_b = W_b x_b, the dense fp64 local dual operators W_b = (K_b^+)[Gamma_b, Gamma_b] of all blocks of the rank in one launch (the FETI dual operator apply, SURVEY 8d dense path)"),
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc,
            "algorithmic_bytes_per_launch": b_k, "launches_timed": n_k, "avg_launch_ms": ms_k / n_k if n_k else None, "timing_stride": 1,
            "timed_over": "the timed region" if n_k else "not timed", "share_of_step_time": (ms_k * 1e-3) / dt if n_k else None,
        }
        if storage_used == "class_orbit":  # compute-bound: the fp64 matrix peak is the roofline (78.6 TFLOP/s dense, AMD's MI355X figure; scripts/micro/mfma_f64.hip measures 72 for this instruction)
            fl_issued, fl_dense = E.apply_flops_detail()
            ms_gemm = ms_first if ms_first > 0 else ms_k  # the GEMM kernel's own launches (event after the kernel, before k_fxo_fin): the dominant kernel the roofline is about
            tf = fl_issued / (ms_gemm / n_k * 1e-3) / 1e12 if n_k else 0.0  # what the matrix cores execute (padded tiles, skipped k segments not counted) / the kernel's launch time
            tf_legacy = flops_k / (ms_k / n_k * 1e-3) / 1e12 if n_k else 0.0
            roofline.update({
                "bound": "mfma", "achieved": tf, "peak": 78.6, "unit": "TFLOP/s", "frac": tf / 78.6, "flops_per_launch": fl_issued,
                "avg_launch_ms": ms_gemm / n_k if n_k else None, "dense_apply_ms": ms_k / n_k if n_k else None, "finishing_kernel_ms": (ms_k - ms_gemm) / n_k if n_k else None,
                "share_of_step_time": (ms_gemm * 1e-3) / dt if n_k else None, "dense_apply_share_of_step_time": (ms_k * 1e-3) / dt if n_k else None,
                "frac_dense_apply": (fl_issued / (ms_k / n_k * 1e-3) / 1e12 / 78.6) if n_k else None,
                # USEFUL flops: what the dense symmetric product y_b = W_b x_b of every block would take, 2 sum_b n_Gamma_b^2 -- the GEMM executes more (padded row tiles, the columns of
                # operations a row tile lists for only some of its rows) to move 48 x fewer bytes; frac_useful prices the launch on that basis
                "flops_useful": 2.0 * float(np.sum(np.asarray(E.n_gamma, dtype=np.float64) ** 2)),
                "frac_useful": (2.0 * float(np.sum(np.asarray(E.n_gamma, dtype=np.float64) ** 2)) / (ms_gemm / n_k * 1e-3) / 1e12 / 78.6) if n_k else None,
                "flops_listed_legacy": flops_k, "frac_legacy_r02": tf_legacy / 78.6, "flops_unpruned_product": fl_dense,
                "flops_note": "flops_per_launch = what the matrix cores execute: every chunk of the padded 120 x 128 tiles the workgroups multiply, the k segments a unit skips (structurally zero B) "
                              "not counted.  flops_listed_legacy = rounds 2-3's count (rows x listed columns x ALL of n_c: the skipped segments still in it, the padding not): frac_legacy_r02 is on that. "
                              "flops_unpruned_product = every (representative, operation, block) (%.2f of it is listed)" % (flops_k / max(fl_dense, 1.0)),
                "kernel": "k_fxo_gemm16<NI, NWM> (+ k_fxo_fin) (v_mfma_f64_16x16x4_f64; workgroup tile 144 / 112 / 80 rows with 1 x 4 waves or 128 / 96 with 2 x 2, whichever pads the representatives' rows least: 715 -> 5 x 144): W_c is invariant under the %d signed coordinate permutations of the cube, so only the rows of the %d orbit representatives are stored (%.2f GB instead of %.2f GB of symmetric tiles) and "
                          "Y = W_c X becomes the GEMM (representatives) x (operations x 8 right-hand sides) over n_c on the fp64 matrix instruction: %.0f flop per stored byte, compute-bound; B is gathered from the L2-resident SIGNED multivector "
                          "(+x and -x per entry: one index per (operation, dof) addresses the signed value, nothing touches a loaded value before the products), the chunk loop is one basic block with the next chunk's loads placed between the products, split-K partial tiles summed in a fixed order; the representatives are ordered by which (operation, block) columns their rows are needed for and "
                          "every row tile multiplies its own column list only (the FETI dual operator apply, SURVEY 8d dense path)"
                          % (q.explicit_symmetries, n_solves - len(E.n_gamma) if n_solves > len(E.n_gamma) else n_solves, E.dense_bytes / 1e9, 4.0 * float(E.class_union(0).size) ** 2 / 1e9, flops_k / max(E.dense_bytes, 1)),
                "hbm_bytes_algorithmic": b_k, "hbm_GBs": achieved,
                "note": "the HBM-bound form of the same apply (--explicit-storage class_sym: k_fxs_symm8, 4.65 GB per apply at 0.77-0.80 of the 8 TB/s peak) takes 0.73-0.75 ms; this form moves 48 x fewer bytes"})
        elif b_k > 1.5 * E.dense_bytes:  # more than 8 blocks per class: one pass over W_c per group of 8
            roofline["note"] = ("W_c (%.2f GB stored on this rank) is streamed once per group of 8 blocks, %.1f passes per apply: the re-reads are served by the 256 MB Infinity Cache / L2, "
                                "so `achieved` is an on-chip rate here, not an HBM rate (the HBM bound applies to the 8-blocks-per-GPU case of configs[2])" % (E.dense_bytes / 1e9, b_k / E.dense_bytes))
        kplus_cfg = {"path": "explicit", "storage": storage_used, "setup_symmetries": getattr(q, "explicit_symmetries", 1), "n_gamma": [int(v) for v in E.n_gamma], "dense_GB": round(E.dense_bytes / 1e9, 2), "assemble_seconds": round(asm_s, 1), "assemble_solves": int(n_solves), "assemble_multi_rhs": bool(getattr(q, "explicit_multi_rhs", False)),
                     "assemble_rtol": a.explicit_rtol, "assemble_solver": "this rank's K^+ (%s), one unit right-hand side per block and application, congruent blocks share their columns%s" % (pc_text, ", one solve per orbit of rows under the %d symmetries of the cube (checked against K, a batch of rows re-solved directly)" % q.explicit_symmetries if getattr(q, "explicit_symmetries", 1) > 1 else "")
                     if not replica else "a %d-slot replica K^+ of the rank's congruent block(s) (%s)" % (a.explicit_slots, pc_text)}
        if storage_used == "class_orbit" and world == 1 and not a.sim_world:  # the GEMM's shape: representatives (rows), their padding to the row tile, operations x 8 right-hand sides (columns), n_c (k)
            import ctypes as _C

            n_rep = int(n_solves - len(E.n_gamma)) if n_solves > len(E.n_gamma) else int(n_solves)
            tm_, mp_ = _C.c_int(), _C.c_int()
            pa._lib.check(ctx.L.pmh_fexplicit_orbit_row_tile(max(1, n_rep), _C.byref(tm_), _C.byref(mp_)))
            kplus_cfg["orbit_gemm"] = {"representatives": n_rep, "row_tile": tm_.value, "padded_rows": mp_.value, "columns": 8 * int(q.explicit_symmetries), "k": int(E.class_union(0).size)}
        kplus_text = "the explicit local dual operators W_b = (K_b^+)[Gamma_b, Gamma_b] (dense fp64, n_Gamma %d-%d, %s%.1f GB on this rank; assembled once by %d K^+ solves at rtol %.0e in %.0f s)" % (
            int(E.n_gamma.min()), int(E.n_gamma.max()), "the congruent blocks share ONE matrix on the union of their Gamma, " if storage_used in ("class", "class_sym", "class_orbit") else "", E.dense_bytes / 1e9, n_solves, a.explicit_rtol, asm_s)
        precision_note = "fp64 throughout: the dense blocks, the GEMV and everything in the dual space are fp64; reduced precision exists only inside the V-cycle that preconditions the SET-UP solves (their CG, residual test at rtol %.0e and solutions are fp64)" % a.explicit_rtol
        if not a.sim_world and not a.no_iterative:  # the inner-Krylov path next to it: fp16-PC default and (one GPU) strict fp64
            # N > 1 (round 6): every rank runs this pass too -- the strong-scaling figure of the path that shards one subdomain per GPU and streams every K_i from HBM at
            # every N stands next to the explicit one in the same line (at N = 1 its like-for-like partner is `iterative_distinct_blocks`: 8 different K_i, nothing shared)
            q.Kplus.attach_explicit(None)
            if has_mg and use_c_builder and not mg_sa and not getattr(a, "_secondary", False) and mg_box(per, True)["min_nodes"] != mg_box(per)["min_nodes"]:
                switch_mg(a.mg_precision, for_iterative=True)  # the set-up solves of the explicit operators ran on the shallow-coarse hierarchy; the inner-Krylov pass takes its own depth
            extra["iterative"] = iterative_pass(min(steps, 108), 4, a.mg_precision)
            if world == 1 and has_mg and a.mg_precision != "fp64" and not getattr(a, "_no_strict", False):
                switch_mg("fp64")
                extra["strict_fp64"] = iterative_pass(min(steps, 108), 2, "fp64")
                extra["strict_fp64"]["note"] = "every operator, vector and the V-cycle in fp64 (the reference's arithmetic throughout)"
            q.Kplus.attach_explicit(E)
    else:
        r = iterative_pass(steps, warmup, a.mg_precision)
        dt, cnt = r["ms_per_step"] * steps / 1e3, r["steps_by_type"]
        roofline = r["roofline"]
        roofline["timed_over"] = "the timed region"
        kplus_cfg = {"path": "iterative", "pc": a.kplus_pc, "cg_spmv_per_step": r["cg_spmv_per_step"], "vcycle_fine_spmv_per_step": r["vcycle_fine_spmv_per_step"], "last_block_cg_iterations": r["last_block_cg_iterations"]}
        kplus_text = "block-wise %s K^+%s (rtol %.0e)" % (pc_text, kreg_text, a.kplus_rtol)
        precision_note = ("the reduced precision (%s) lives ONLY in the V-cycle that preconditions the block CG of K^+: that CG's operator, residual, stopping test (rtol %.0e) and solution "
                          "are fp64, as is everything in the dual space" % (a.mg_precision, a.kplus_rtol)) if has_mg else "fp64 throughout"
        if world == 1 and not a.sim_world and has_mg and a.mg_precision != "fp64" and not a.no_iterative:
            switch_mg("fp64")
            extra["strict_fp64"] = iterative_pass(min(steps, 108), 2, "fp64")
    # the K x of the CG inside K^+ (k_bsr3<double>) as the solver runs it, timed on three K^+ applications with the bench's own event pairs.  With congruent blocks ONE device
    # copy serves all of them: `achieved` / `frac` are on the bytes that copy + the vectors take from HBM (<= the peak); the figure of the block-diagonal product (every K_i
    # counted) is `blockdiag_figure_GBs`, an L2-served speed-up, not an HBM rate
    kx = None
    if world == 1 and not a.sim_world and not a.no_bsr3 and want_timing:
        os.environ["PMH_TIMING_STRIDE"] = "1"
        rhsv, uv = ctx.vec_from(np.random.default_rng(5).standard_normal(local["n_x"])), ctx.vec(local["n_x"])
        q.Kplus.timing_enable(512)
        for _ in range(3):
            q.Kplus.mult(rhsv, uv)
        n_x, ms_x, b_x = q.Kplus.timing_get()
        q.Kplus.timing_enable(0)
        kplus_its = q.Kplus.last_iterations()[0]
        rhsv.free(), uv.free()
        if n_x and q.Kplus.multi_rhs_active():
            gbs = b_x / (ms_x / n_x * 1e-3) / 1e9
            kx = {"bound": "hbm", "kernel": "k_mv_spmv<double, 8 columns>", "what": "Y = K_1 X of the block-wise CG inside K^+ as pmh_matinv_mult runs it on 8 congruent blocks: the 8 blocks' vectors are the 8 columns of "
                                                                                    "ONE block (matinv_mv.hip), the ELL copy of K_1 (3x3 blocks) streamed once per launch",
                  "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": b_x, "launches_timed": n_x, "avg_launch_ms": ms_x / n_x,
                  "matrix_copies_on_device": 1, "blocks_per_copy": 8,
                  "note": "the one-column kernel k_bsr3<double> (8 replicas of one device copy served from L2; the set-up solves of non-congruent blocks and every N > 1 run use it) is timed with PMH_NO_KPLUS_MV=1"}
        elif n_x:
            nrep = q.Kplus.bsr3_replicas()
            gbs = b_x / (ms_x / n_x * 1e-3) / 1e9
            b_bd = nrep * (b_x - 16.0 * local["n_x"]) + 16.0 * local["n_x"]
            kx = {"bound": "hbm", "kernel": "k_bsr3<double>", "what": "y = K x of the block-wise CG inside K^+ as the solver runs it (3x3 blocks, 8.44 B per non-zero; the set-up solves of the explicit operators and the inner-Krylov path run on it)",
                  "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": b_x, "launches_timed": n_x, "avg_launch_ms": ms_x / n_x,
                  "matrix_copies_on_device": 1 if nrep > 1 else per, "blocks_per_copy": nrep,
                  "blockdiag_figure_bytes": b_bd, "blockdiag_figure_GBs": b_bd / (ms_x / n_x * 1e-3) / 1e9,
                  "note": ("the %d congruent blocks share ONE device copy of K_i (pmh_bsr3_from_csr compares them entry by entry): the replicas of a tile run back to back on one XCD and read it from that L2. "
                           "achieved/frac count what HBM delivers (the copy once + x + y); blockdiag_figure_GBs counts every K_i as SURVEY 8d does and is NOT an HBM rate. The HBM-streaming "
                           "product (distinct K_i, a copy each) is the feti_dual_spmv block" % nrep) if nrep > 1 else
                          "one device copy per block: every byte of the block-diagonal product is streamed from HBM once"}
    if kx is not None:
        kx["kplus_cg_iterations_random_rhs"] = int(kplus_its)
        kx["kplus_rtol"] = a.kplus_rtol
    # the north star's "FETI dual SpMV" on HBM at this size: 8 DISTINCT K_i, one device copy each (dual_spmv_hbm)
    dual = None
    if world == 1 and not a.sim_world and want_timing and not getattr(a, "no_dual_spmv", False) and not getattr(a, "_secondary", False) and not partition:
        try:
            dual = dual_spmv_hbm(ctx, f)
        except Exception as ex:  # noqa: BLE001 - never at the cost of the headline
            dual = {"failed": repr(ex)}
    # one whole solve from lambda = 0 to the outer tolerance (the REAL stopping rule): what a user waits for after the set-up
    q.lam.set(0.0)
    qps_full = q.make_smalxe()  # a fresh solver object (QPS_SMALXE.state = 1 as after QPSCreate; the throughput passes above leave it at 3)
    ctx.sync()
    t1 = time.perf_counter()
    st_full = qps_full.Solve()
    ctx.sync()
    t_solve = time.perf_counter() - t1
    # the same solve and the same throughput window with the EXTENSION pmh_smalxe_set_reuse_products (off in everything above: the headline keeps the reference's
    # operation sequence): A_rho u carried from the inner solve's last gradient, two operator applications less per outer iteration
    reuse = None
    if a.kplus == "explicit" and world == 1 and not a.sim_world:
        lam_ref = q.lam.to_numpy().copy()
        q.lam.set(0.0)
        qps_r = q.make_smalxe()
        qps_r.SMALXESetReuseProducts(True)
        ctx.sync()
        t1 = time.perf_counter()
        st_r = qps_r.Solve()
        ctx.sync()
        t_r = time.perf_counter() - t1
        lam_r = q.lam.to_numpy().copy()
        if warmup:
            qps_r.RunFixedSolve(warmup)
        ctx.sync()
        t1 = time.perf_counter()
        cnt_r = qps_r.RunFixedSolve(steps)
        ctx.sync()
        dt_r = time.perf_counter() - t1
        reuse = {"what": "EXTENSION, not the reference's operation sequence (off in the headline): pmh_smalxe_set_reuse_products carries A_rho u from the last gradient of an inner solve into the Lagrangian "
                         "(QPComputeObjective's own product, smalxe.c:982) and into the next inner solve's first gradient (mpgp.c:500): g' = g + rho B'B u",
                 "value": steps / dt_r, "unit": "QPS iterations/s", "ms_per_step": dt_r / steps * 1e3, "steps": steps, "warmup": warmup, "steps_by_type": cnt_r,
                 "full_solve": {"solve_seconds": t_r, "outer_iterations": int(st_r.iteration), "inner_iterations": int(st_r.inner_iter_accu), "hessian_mults": int(st_r.inner.nmv), "reason": int(st_r.reason)},
                 "reference_sequence": {"solve_seconds": t_solve, "outer_iterations": int(st_full.iteration), "inner_iterations": int(st_full.inner_iter_accu), "hessian_mults": int(st_full.inner.nmv)},
                 "rel_diff_lambda": float(np.linalg.norm(lam_r - lam_ref) / max(np.linalg.norm(lam_ref), 1e-300))}
        q.lam.set_numpy(lam_ref)  # (the checksum below is of the reference-sequence solve)
    full_solve = {"solve_seconds": t_solve, "outer_iterations": int(st_full.iteration), "inner_iterations": int(st_full.inner_iter_accu), "reason": int(st_full.reason),
                  "setup_seconds": round(t_setup, 2), "time_to_solution_seconds": round(t_setup + t_solve, 3),
                  "note": "set-up = everything between the generated problem and the first solver iteration as bench.py orchestrates it (uploads, 3x3-block copies, multigrid hierarchy, explicit operators by K^+ solves, "
                          "coarse problem, dual chain, SMALXE set-up with its power method); solve = pmh_smalxe_solve to rtol 1e-5"}
    comm_rank, comm_size = ctx.comm_rank()
    res = {
        "value": steps / dt, "ms_per_step": dt / steps * 1e3, "full_solve": full_solve,
        "workload": "%s: 3-D elasticity TFETI, %dx%dx%d cubic subdomains of %d^3 Q1 elements (N=%d dof, K_i %d rows / %d nnz, n_lambda=%d "
                    "incl. %d contact rows), rigid obstacle, SMALXE+MPGP on the dual QP (the real solver loop, restarted when it converges), F = B K^+ B' through %s%s"
                    % (("configs[2]-like, IRREGULAR PARTITION ('%s': the same body cut into 8 subdomains that are not boxes -- no congruence, no symmetry, no box hierarchy; rows per block %d-%d)" % (partition, int(np.diff(f.block_rowstart).min()), int(np.diff(f.block_rowstart).max()))) if partition else
                       ("configs[2]-like, HETEROGENEOUS (Young's moduli %s: no two subdomain matrices are equal -- the general, non-congruent case)" % ", ".join("%g" % v for v in f.young)) if not congruent else
                       "configs[2]" if (sub == (2, 2, 2) and orth and a.nel == 43) else "configs[3]" if (nsub == 64 and a.nel == 21 and not orth) else "configs[3]-shaped" if nsub == 64 else "configs[2]-like",
                       sub[0], sub[1], sub[2], a.nel, f.N, int(np.diff(f.block_rowstart).max()), int(max(f.block_K(s_).nnz for s_ in range(f.nsub))), f.n_lambda, f.n_ineq,
                       kplus_text, "" if orth else ", coarse problem: dense %d x %d (GG')^{-1}" % (G.shape[0], G.shape[0])),
        "parallelism": ("%d subdomain block(s) per GPU on %d GPU(s) (K_i, K^+, the set-up solves); dual vectors replicated; one RCCL all-reduce (n_lambda doubles) per F apply" % (per, world))
                       + ("; the dense local dual operators are applied in 128-row stripes dealt evenly over the GPUs (the cubes are congruent: every rank assembles its stripes of every W_b with its own K^+)" if (explicit and "stripe" in explicit) else "")
                       + (" [REHEARSAL --sim-world %d: rank 0's share only, no collective; not a result]" % a.sim_world if (a.sim_world and world == 1) else ""),
        "workload_short": "%s: 3-D elasticity TFETI contact, %dx%dx%d subdomains of %d^3 Q1 elements (N=%d dof, n_lambda=%d), SMALXE+MPGP on the dual QP, F = B K^+ B' with K^+ %s" % (
            ("configs[2]-like, irregular partition '%s'" % partition) if partition else "configs[2]-like heterogeneous" if not congruent else "configs[2]" if (sub == (2, 2, 2) and orth and a.nel == 43) else "configs[3]" if (nsub == 64 and a.nel == 21 and not orth) else "configs[2]-like",
            sub[0], sub[1], sub[2], a.nel, f.N, f.n_lambda, ("explicit (dense local dual operators, storage %s)" % q.explicit_storage) if a.kplus == "explicit" else "iterative (block CG, %s PC)" % a.kplus_pc),
        "parallelism_short": "%d subdomain block(s)/GPU on %d GPU(s), strong scaling; dual vectors replicated, one RCCL all-reduce (n_lambda doubles) per F apply%s" % (
            per, world, " [REHEARSAL --sim-world %d]" % a.sim_world if (a.sim_world and world == 1) else ""),
        "rccl_ranks": comm_size if (world > 1 or os.environ.get("PMH_BENCH_FORCE_DIST")) else None,
        "steps_by_type": cnt, "precision_note": precision_note, "kplus": kplus_cfg,
        "checksum": {"norm_lambda_child_after_last_step": repr(float(q.lam.norm()))},  # bitwise comparable between runs (deterministic reductions)
        "generate_seconds": round(t_gen, 1), "setup_seconds": round(t_setup, 1),
        "setup_seconds_max_over_ranks": round(dist_max(dist, t_setup), 1) if dist is not None else None, "generate_seconds_max_over_ranks": round(dist_max(dist, t_gen), 1) if dist is not None else None,
        "host_threads_per_rank": host_threads(), "collective": collective or None,
        "coarse_problem": (lambda s: {"m": q.pf.m, "GGt_mfma_ms": round(s[0], 3), "GGt_TFLOPs": round(s[1] / (s[0] * 1e-3) / 1e12, 2) if s[0] > 0 else None,
                                      "host_cholesky_inverse_ms": round(s[2], 2)})(q.pf.setup_stats()) if (q.pf is not None and not orth) else {"m": q.pf.m if q.pf is not None else 0, "orthonormal_G": "implicit: T G0 with T = chol(G0 G0')^{-1} applied in the finishing launch of G0 v, G0 kept sparse (%d non-zeros)" % G.nnz if implicit else True},
        "roofline": roofline, "feti_dual_spmv": dual, "kplus_cg_product": kx, "reuse_products": reuse,
    }
    res.update(extra)
    return res, f, G, hier, q.b.to_numpy(), q.lb_new.to_numpy()  # (the CPU baseline leg re-uses the generated problem)


def kernel_name_only(k):
    """'k_fxo_gemm / k_fxo_gemm4<NA> (row tile ...): ...' -> 'k_fxo_gemm / k_fxo_gemm4<NA>'"""
    if not isinstance(k, str):
        return k
    cut = len(k)
    for tok in (" (", ": "):
        i = k.find(tok)
        if i > 0:
            cut = min(cut, i)
    return k[:cut][:64]


def _num(v, digits=5):
    if isinstance(v, float):
        return float("%.*g" % (digits, v)) if math.isfinite(v) else None
    return v


def compact_line(out, details_path):
    """The driver's line: the contract's keys, `roofline` and `cpu_baseline` trimmed to numbers and names, one-number summaries of the secondary blocks.  Everything else
    (notes, provenance, per-block detail) is in the details file."""
    def roof(r):
        if not isinstance(r, dict):
            return None
        keep = {"bound": r.get("bound"), "kernel": kernel_name_only(r.get("kernel")), "achieved": _num(r.get("achieved")), "peak": r.get("peak"), "unit": r.get("unit"), "frac": _num(r.get("frac"), 4),
                "traffic": _num(r.get("traffic"), 6), "avg_launch_ms": _num(r.get("avg_launch_ms")), "launches_timed": r.get("launches_timed")}
        if r.get("traffic") is not None:  # where the HBM bytes come from: a committed rocprofv3 --pmc pass of this kernel (not measured in this run), with the state it measured
            keep["traffic_source"] = str(r.get("traffic_source") or "")[:44]
        if r.get("bound") == "mfma":
            keep["flops_per_launch"] = _num(r.get("flops_per_launch"), 6)
            keep["hbm_bytes_algorithmic"] = _num(r.get("hbm_bytes_algorithmic"), 6)
        else:
            keep["algorithmic_bytes_per_launch"] = _num(r.get("algorithmic_bytes_per_launch"), 6)
        for k in ("share_of_step_time", "whole_iteration_frac", "frac_streamed", "frac_legacy_r02", "frac_useful"):
            if r.get(k) is not None:
                keep[k] = _num(r[k], 4)
        return keep

    def block(b):
        if not isinstance(b, dict):
            return None
        if b.get("failed"):
            return {"value": None, "failed": str(b["failed"])[:120]}
        o = {"value": _num(b.get("value")), "ms_per_step": _num(b.get("ms_per_step"))}
        if isinstance(b.get("roofline"), dict):
            o["roofline_bound"], o["roofline_frac"], o["kernel"] = b["roofline"].get("bound"), _num(b["roofline"].get("frac"), 4), kernel_name_only(b["roofline"].get("kernel"))
            if b["roofline"].get("frac_streamed") is not None:
                o["frac_streamed"] = _num(b["roofline"]["frac_streamed"], 4)
            if b["roofline"].get("whole_iteration_frac") is not None:
                o["whole_iteration_frac"] = _num(b["roofline"]["whole_iteration_frac"], 4)
        if isinstance(b.get("cpu_baseline"), dict) and b["cpu_baseline"].get("value") is not None:  # a block with its own CPU leg (configs[3]: the reference's op sequence, live)
            cbb = b["cpu_baseline"]
            o["cpu_baseline"] = {"value": _num(cbb.get("value")), "cores": cbb.get("cores"), "kind": cbb.get("kind"), "extrapolated": bool(cbb.get("extrapolated", False)), "measured": cbb.get("measured")}
        return o

    cfg = out.get("config", {})
    c = {k: out.get(k) for k in ("metric", "value", "unit")}
    # what makes windows of different length comparable sits right behind `value`: the driver's 20-step window is the START of a solve (more F applications per step)
    if out.get("ms_per_operator_apply") is not None:
        c["ms_per_operator_apply"], c["applies_per_step"] = _num(out.get("ms_per_operator_apply")), _num(out.get("applies_per_step"))
    c.update({k: out.get(k) for k in ("n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")})
    c["value"], c["ms_per_step"] = _num(c["value"], 7), _num(c["ms_per_step"], 7)
    c["config"] = {"workload": cfg.get("workload_short") or str(cfg.get("workload", ""))[:300], "parallelism": str(cfg.get("parallelism_short") or cfg.get("parallelism", ""))[:160], "rccl_ranks": cfg.get("rccl_ranks")}
    if cfg.get("transport"):
        c["config"]["transport"] = cfg["transport"]
    if cfg.get("collective"):  # N > 1: the all-reduce that ends every B u, HIP-event timed on the launch stream (pmh_comm_timing_*)
        c["config"]["collective"] = {k: _num(v) if isinstance(v, float) else v for k, v in cfg["collective"].items()}
    for k in ("host_threads_per_rank", "setup_seconds_max_over_ranks", "generate_seconds_max_over_ranks"):
        if cfg.get(k) is not None:
            c["config"][k] = cfg[k]
    if cfg.get("checksum"):
        c["config"]["checksum"] = cfg["checksum"]
    if cfg.get("steps_by_type"):
        c["config"]["steps_by_type"] = {k: v for k, v in cfg["steps_by_type"].items() if k in ("cg", "expansion", "proportioning", "hessian_mults", "operator_applies", "solves", "outer")}
    c["roofline"] = roof(out.get("roofline"))
    mc = out.get("roofline", {}).get("measured_ceiling")
    if isinstance(mc, dict):
        c["roofline"]["measured_ceiling"] = {k: _num(v, 4) for k, v in mc.items()}
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        c["cpu_baseline"] = {"value": _num(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"), "extrapolated": bool(cb.get("extrapolated", False)), "counts": (str(cb.get("counts")).split(" (")[0] if cb.get("counts") else None), "calibrated_from": (str(cb.get("calibrated_from")).split(" x ")[0] if cb.get("calibrated_from") else None),
                             "cpu_model": str(cb.get("cpu_model") or cpu_model())[:40], "sample": str(cb.get("sample_short") or cb.get("sample", ""))[:90]}
    if isinstance(cb, dict) and cb.get("kplus"):
        c["cpu_baseline"]["kplus"] = cb["kplus"]
        if cb.get("measured_once"):
            c["cpu_baseline"]["measured_once"] = cb["measured_once"]
    ci = out.get("cpu_baseline_iterative")
    if isinstance(ci, dict):
        c["cpu_baseline_iterative"] = {"value": _num(ci.get("value")), "cores": ci.get("cores"), "kind": ci.get("kind"), "kplus": ci.get("kplus")}
    fd = out.get("feti_dual_spmv")
    if isinstance(fd, dict):
        if fd.get("failed"):
            c["feti_dual_spmv"] = {"failed": str(fd["failed"])[:120]}
        else:
            c["feti_dual_spmv"] = {"bound": "hbm", "kernel": kernel_name_only(fd.get("kernel")), "achieved": _num(fd.get("achieved")), "peak": fd.get("peak"), "unit": "GB/s", "frac": _num(fd.get("frac"), 4),
                                   "algorithmic_bytes_per_launch": _num(fd.get("algorithmic_bytes_per_launch"), 6), "avg_launch_ms": _num(fd.get("avg_launch_ms")), "traffic": _num(fd.get("traffic"), 6),
                                   "device_copies": (fd.get("csr") or {}).get("device_copies"),
                                   "bsr3": {k: _num((fd.get("bsr3") or {}).get(k), 4) for k in ("achieved", "frac", "csr_equivalent_frac", "avg_launch_ms") if (fd.get("bsr3") or {}).get(k) is not None}}
    for k in ("applies_per_step", "ms_per_operator_apply", "time_to_solution_s"):
        if out.get(k) is not None:
            c[k] = _num(out[k])
    if isinstance(out.get("full_solve"), dict):
        c["full_solve"] = {k: _num(out["full_solve"].get(k)) for k in ("solve_seconds", "outer_iterations", "inner_iterations", "reason", "setup_seconds")}
    for k in ("iterative", "iterative_distinct_blocks", "strict_fp64", "general", "general_nosym", "configs1", "configs3", "configs4", "reuse_products"):
        if k in out:
            c[k] = block(out[k])
    if isinstance(c.get("general"), dict) and isinstance(out["general"], dict) and isinstance(out["general"].get("kplus"), dict):  # the non-congruent block's set-up: K^+ solves and their time
        kp = out["general"]["kplus"]
        c["general"].update({k: kp.get(k) for k in ("storage", "assemble_seconds", "assemble_solves", "assemble_multi_rhs") if kp.get(k) is not None})
    if isinstance(c.get("general_nosym"), dict) and isinstance(out["general_nosym"], dict) and isinstance(out["general_nosym"].get("kplus"), dict):  # blocks that are not boxes: nothing shared, nothing by symmetry
        kp = out["general_nosym"]["kplus"]
        c["general_nosym"].update({k: kp.get(k) for k in ("storage", "assemble_seconds", "assemble_solves", "assemble_multi_rhs") if kp.get(k) is not None})
        kxn = out["general_nosym"].get("kplus_cg_product") or {}
        if kxn.get("kplus_cg_iterations_random_rhs") is not None:
            c["general_nosym"]["kplus_cg_iterations"] = kxn["kplus_cg_iterations_random_rhs"]
    if isinstance(out.get("contact_solve"), dict):
        c["contact_solve"] = {k: out["contact_solve"].get(k) for k in ("setup_seconds", "solve_seconds", "time_to_solution_seconds", "failed") if out["contact_solve"].get(k) is not None}
    c["details"] = os.path.relpath(details_path, ROOT) if details_path.startswith(ROOT) else details_path
    line = json.dumps(c, separators=(",", ":"))
    if len(line) >= 4000:  # never hand the driver a line it cannot take: drop the summaries of the secondary blocks first
        for k in ("reuse_products", "contact_solve", "full_solve", "cpu_baseline_iterative", "strict_fp64", "iterative_distinct_blocks", "general", "iterative", "general_nosym", "configs4", "configs3", "configs1", "full_solve", "cpu_baseline_iterative"):
            c.pop(k, None)
            line = json.dumps(c, separators=(",", ":"))
            if len(line) < 4000:
                break
    return line


def launch_ranks(a):
    """`python bench.py --gpus N` with N > 1 and no rendezvous in the environment: THIS process -- which has not touched the GPU and never does -- starts the N ranks as
    child processes (one per GPU: RANK = LOCAL_RANK = 0..N-1, WORLD_SIZE = N, MASTER_ADDR 127.0.0.1, a free MASTER_PORT: what torch.distributed.run would export), waits
    for them, forwards rank 0's line as its own last stdout line and exits non-zero if any rank failed (the other ranks are then ended by PID).  No exec of a process that
    initialised the GPU, no retry."""
    import socket
    import subprocess
    import tempfile

    n = a.gpus
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    tmp = tempfile.mkdtemp(prefix="pmh_bench_")
    procs, outs = [], []
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    share = max(1, cores // n)  # the node's CPUs dealt over the ranks: numpy / OpenMP threads and the library's host-side builders (host_threads() in the child arrives at the same number)
    deadline = time.time() + float(os.environ.get("PMH_BENCH_DEADLINE_S", "1500"))  # watchdog: a hung rank must not hang the launcher (children are ended by PID, exit code 124)
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), GROUP_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS=str(share), PMH_HOST_THREADS=str(share), OPENBLAS_NUM_THREADS=str(share), MKL_NUM_THREADS=str(share))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
        fo = open(os.path.join(tmp, "rank%d.out" % r), "w+")
        outs.append(fo)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=fo, stderr=None))
    rc, alive = 0, set(range(n))
    while alive:
        if time.time() > deadline and rc == 0:
            rc = 124
            sys.stderr.write("bench.py: the ranks did not finish within PMH_BENCH_DEADLINE_S; ending them\n")
            for o in alive:
                procs[o].terminate()
            t_kill = time.time() + 10
            while any(procs[o].poll() is None for o in alive) and time.time() < t_kill:
                time.sleep(0.1)
            for o in alive:
                if procs[o].poll() is None:
                    procs[o].kill()
        for r in sorted(alive):
            c = procs[r].poll()
            if c is None:
                continue
            alive.discard(r)
            if c != 0 and rc == 0:
                rc = c
                sys.stderr.write("bench.py: rank %d exited with code %d; ending the other ranks\n" % (r, c))
                for o in alive:
                    procs[o].terminate()
        time.sleep(0.05)
    texts = []
    for fo in outs:
        fo.seek(0)
        texts.append(fo.read())
        fo.close()
    for r in range(1, n):  # the other ranks print nothing on stdout in a real run; whatever they did print goes to stderr
        if texts[r].strip() and not a.dry_launch:
            sys.stderr.write("[rank %d stdout] %s\n" % (r, texts[r].strip()[-2000:]))
    if rc != 0:
        sys.stderr.write(texts[0][-2000:])
        raise SystemExit(rc if rc > 0 else 1)
    if a.dry_launch:
        kids = [json.loads([ln for ln in t.splitlines() if ln.strip()][-1]) for t in texts]
        print(json.dumps({"dry_launch": True, "launcher_pid": os.getpid(), "n_children": n, "children": kids}))
        return
    lines = [ln for ln in texts[0].splitlines() if ln.strip()]
    for ln in lines[:-1]:
        sys.stderr.write(ln + "\n")
    print(lines[-1])


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:  # before anything initialises the GPU (or imports torch)
        return launch_ranks(a)
    if a.dry_launch:
        if os.environ.get("PMH_BENCH_TEST_HANG"):  # (tests/test_bench_host.py: a rank that never finishes, for the launcher's watchdog)
            time.sleep(3600)
        print(json.dumps({"dry_launch": True, "pid": os.getpid(), "ppid": os.getppid(), **{k.lower(): os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}, "argv": sys.argv[1:],
                          "host_threads": host_threads(), "omp_num_threads": os.environ.get("OMP_NUM_THREADS"), "pmh_host_threads": os.environ.get("PMH_HOST_THREADS")}))
        return
    # stdout carries ONE line, the last thing this process prints: whatever a library writes to file descriptor 1 meanwhile (RCCL prints a version banner when a
    # communicator is created) goes to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    host_threads()
    if os.environ.get("PMH_BENCH_RANK_DEADLINE_S"):  # per-rank watchdog (a launcher that does not watch, e.g. torch.distributed.run): leave with a message instead of hanging in a collective
        import threading

        def _late():
            sys.stderr.write("bench.py: rank %s gave up after PMH_BENCH_RANK_DEADLINE_S\n" % os.environ.get("RANK", "0"))
            sys.stderr.flush()
            os._exit(124)

        _wd = threading.Timer(float(os.environ["PMH_BENCH_RANK_DEADLINE_S"]), _late)
        _wd.daemon = True
        _wd.start()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and not os.environ.get("PMH_BENCH_FORCE_DIST"):
        raise SystemExit("--gpus %d but the rendezvous environment says WORLD_SIZE = %d" % (a.gpus, world))
    dist = None
    force_dist = bool(os.environ.get("PMH_BENCH_FORCE_DIST"))  # exercise the N>1 code path on a single rank (testing)
    if world > 1 or force_dist:
        import torch
        import torch.distributed as dist_

        dist = dist_
        if HOST_TRANSPORT:
            # PMH_BENCH_TRANSPORT=host (tests): every rank on device 0, gloo between the processes, the library's collectives through pmh_comm_set_host_transport -- what runs is
            # bench.py's own N > 1 orchestration (shares of the operator, barriers, max over ranks) and the library's distributed arithmetic; RCCL cannot put two ranks on one device
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    global CPU_POOL
    if (rank == 0 and world == 1 and not force_dist and a.workload == "feti" and not a.sim_world and not a.no_cpu_baseline and not a.no_c2 and not a.no_configs3 and not a.no_cpu_pool
            and a.kplus == "explicit" and not a.young and not a.partition):
        # the worker processes of the configs[3] CPU leg (the reference's direct K^+, one factorisation per worker: oracle/direct_pool.py) are started HERE, before this
        # process initialises the GPU -- they import numpy / scipy only and idle until that leg runs
        try:
            from oracle.direct_pool import DirectPool

            CPU_POOL = DirectPool(host_threads())
        except Exception as ex:  # noqa: BLE001 - the CPU leg then says why it is missing
            sys.stderr.write("bench.py: no CPU worker pool: %r\n" % (ex,))
            CPU_POOL = None

    import permon_amd as pa

    ctx = pa.Context(local_rank)
    if (world > 1 or force_dist) and HOST_TRANSPORT:
        import torch

        def host_transport(op, arr):
            if op == 2 or arr.size == 0:
                dist.barrier()
                return
            dist.all_reduce(torch.from_numpy(arr), op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MIN)  # a view of the library's pinned staging buffer: reduced in place

        ctx.comm_set_host_transport(rank, world, host_transport)
    elif world > 1 or force_dist:
        import torch

        idt = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            idt.copy_(torch.tensor(list(ctx.comm_unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, 0)
        ctx.comm_init(rank, world, bytes(idt.cpu().tolist()))

    if a.workload == "c2":
        steps, warmup = a.steps or 300, a.warmup if a.warmup is not None else 30
        r = run_c2(ctx, a, steps, warmup, cpu=(rank == 0 and world == 1 and not a.no_cpu_baseline))
        if dist is not None:
            import torch

            r["ms_per_step"] = dist_max(dist, r["ms_per_step"])
            r["value"] = 1e3 / r["ms_per_step"]
        out = {
            "metric": "QPS iterations/sec + CSR SpMV GB/s (% HBM roofline)", "value": world * r["value"], "unit": "QPS iterations/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": r["ms_per_step"], "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": r["workload"], "workload_short": r["workload"], "parallelism": "1 GPU" if world == 1 else "%d independent replicas (configs[1] is a single-CSR, single-GPU config: replicas only)" % world,
                       "steps_by_type": r["steps_by_type"], "setup_seconds": r["setup_seconds"]},
            "roofline": r["roofline"],
        }
        if "cpu_baseline" in r:
            out["cpu_baseline"] = r["cpu_baseline"]
    elif a.workload == "svm":
        steps, warmup = a.steps or 100, a.warmup if a.warmup is not None else 10
        r = run_svm(ctx, a, steps, warmup, rank, world, dist)
        out = {
            "metric": "QPS iterations/sec + CSR SpMV GB/s (% HBM roofline)", "value": r["value"], "unit": "QPS iterations/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": r["ms_per_step"], "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": r["workload"], "workload_short": r["workload"], "parallelism": r["parallelism"], "rccl_ranks": r["rccl_ranks"], "steps_by_type": r["steps_by_type"], "setup_seconds": r["setup_seconds"], "checksum": r["checksum"]},
            "roofline": r["roofline"],
        }
    else:
        steps, warmup = a.steps or 216, a.warmup if a.warmup is not None else 8  # 216 = two full configs[2] solves (108 inner iterations each)
        r, f, G, hier, b_dual, lb_dual = run_feti(ctx, a, steps, warmup, rank, world, dist)
        out = {
            "metric": "QPS iterations/sec + CSR SpMV GB/s (% HBM roofline)", "value": r["value"], "unit": "QPS iterations/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": r["ms_per_step"], "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": r["workload"], "workload_short": r["workload_short"], "parallelism": r["parallelism"], "parallelism_short": r["parallelism_short"], "rccl_ranks": r["rccl_ranks"], "steps_by_type": r["steps_by_type"], "kplus": r["kplus"], "coarse_problem": r["coarse_problem"], "checksum": r["checksum"],
                       "precision_note": r["precision_note"], "generate_seconds": r["generate_seconds"], "setup_seconds": r["setup_seconds"],
                       "setup_seconds_max_over_ranks": r["setup_seconds_max_over_ranks"], "generate_seconds_max_over_ranks": r["generate_seconds_max_over_ranks"], "host_threads_per_rank": r["host_threads_per_rank"],
                       "collective": r["collective"]},
            "roofline": r["roofline"],
        }
        # what lets two windows of the same solver be compared (the driver's 20 + 5 steps are the START of a solve: many short outer iterations, ~3.1 F
        # applications per step; the default 216 steps are two whole solves, 1.9 per step): F applications per step and the time of one of them, all included
        sbt = r["steps_by_type"]
        out["applies_per_step"] = (sbt["operator_applies"] / steps) if sbt.get("operator_applies") else None
        out["ms_per_operator_apply"] = sbt.get("ms_per_operator_apply")
        out["time_to_solution_s"] = r["full_solve"]["time_to_solution_seconds"]
        out["full_solve"] = r["full_solve"]
        out["feti_dual_spmv"] = r["feti_dual_spmv"]
        out["kplus_cg_product"] = r["kplus_cg_product"]
        if r.get("reuse_products"):
            out["reuse_products"] = r["reuse_products"]
        for k in ("iterative", "strict_fp64"):
            if k in r:
                out[k] = r[k]
        if rank == 0 and world == 1 and not a.sim_world:
            applies_per_step = out["applies_per_step"] or 1.9
            if not a.no_cpu_baseline and not a.regularize:
                # `cpu_baseline` = the REFERENCE's algorithm on the host: the sparse direct K^+ (factor once, one forward / backward solve per block and application), its solve
                # phase measured at the full block size (once, committed: profiles/r05_splu_43.json) and carried to this host by a size this run factors itself -- a like-for-like
                # partner of the GPU headline's exact (explicit) K^+.  Where no full-size measurement is committed for the block size it is the three-size extrapolation
                # (`extrapolated: true`).  `cpu_baseline_iterative` = the host port of the GPU's inner-Krylov K^+ (oracle/mg_host.py), measured in this run.
                if hier is not None:
                    try:
                        out["cpu_baseline_iterative"] = cpu_baseline_feti(f, G, hier, b_dual, lb_dual, max(4, a.cpu_its_feti), a.kplus_rtol, orth=(not a.dense_coarse) and a.orth_form == "explicit")  # implicit form: G0 with the dense (G0 G0')^{-1} = the same projector
                        out["cpu_baseline_iterative"]["kplus"] = "block CG + V-cycle (the GPU's iterative path restated)"
                    except Exception as ex:  # noqa: BLE001
                        out["cpu_baseline_iterative"] = {"value": None, "unit": "QPS iterations/s", "cores": 1, "kind": "port", "sample": "failed: %r" % (ex,)}
                try:
                    out["cpu_baseline"] = cpu_baseline_direct(ctx, a.nel, min(a.cpu_direct_nel, a.nel), applies_per_step, nblocks=len(f.block_rowstart) - 1)
                except Exception as ex:  # noqa: BLE001
                    out["cpu_baseline"] = out.get("cpu_baseline_iterative") or {"value": None, "unit": "QPS iterations/s", "cores": 1, "kind": "port", "sample": "failed: %r" % (ex,)}
            del f, G, hier
            import copy

            if a.no_c2:  # --no-c2 = the headline alone: no secondary block at all
                a.general_nel, a.nosym_nel, a.no_configs3, a.no_svm, a.no_contact_solve = 0, 0, True, True, True

            def secondary(name, fn):
                """A secondary block must never cost the headline: failures are recorded, not raised."""
                t0 = time.time()
                try:
                    blk = fn()
                except Exception as ex:  # noqa: BLE001
                    blk = {"value": None, "failed": repr(ex)}
                blk["block_seconds"] = round(time.time() - t0, 1)
                out[name] = blk

            def feti_block(**over):
                a2 = copy.copy(a)
                a2._secondary = True
                for k_, v_ in over.items():
                    setattr(a2, k_, v_)
                full = run_feti(ctx, a2, over.get("_steps", 108), 8, 0, 1, None)
                r2 = full[0]
                if over.get("_keep") is not None:  # (the block's generated problem and dual QP, for its CPU leg)
                    over["_keep"].update(f=full[1], G=full[2], b_dual=full[4], lb_dual=full[5])
                sb = r2["steps_by_type"]
                return {"value": r2["value"], "unit": "QPS iterations/s", "ms_per_step": r2["ms_per_step"], "steps": over.get("_steps", 108), "warmup": 8, "workload": r2["workload"],
                        "applies_per_step": sb["operator_applies"] / over.get("_steps", 108) if sb.get("operator_applies") else None, "ms_per_operator_apply": sb.get("ms_per_operator_apply"),
                        "steps_by_type": sb, "kplus": r2["kplus"], "coarse_problem": r2["coarse_problem"], "full_solve": r2["full_solve"], "setup_seconds": r2["setup_seconds"], "roofline": r2["roofline"], "kplus_cg_product": r2["kplus_cg_product"],
                        **({"iterative": r2["iterative"]} if "iterative" in r2 else {})}

            if a.general_nel and not a.young:
                # the general (non-congruent) path of the explicit operators: 8 subdomains of 8 different materials -> no class sharing, no set-up by symmetry,
                # per-block symmetric storage applied by k_fx_symv (HBM-bound); every column of every W_b by its own K^+ solve
                # its inner-Krylov pass = MatMult_Inv on 8 DISTINCT K_i: one device copy per block, every byte from HBM (no congruence to lean on) -- the 1-GPU partner of the
                # N > 1 runs' `iterative` figure (one block per GPU at N = 8)
                secondary("general", lambda: feti_block(young="distinct", nel=a.general_nel, sub="2,2,2", dense_coarse=False, no_iterative=False, _no_strict=True, explicit_storage="auto"))
                if isinstance(out.get("general"), dict) and isinstance(out["general"].get("iterative"), dict):
                    out["iterative_distinct_blocks"] = dict(out["general"].pop("iterative"), workload="the `general` block's problem: 8 subdomains of 8 different materials, K^+ by the inner Krylov solver on 8 distinct K_i (every byte from HBM)")
            if a.nosym_nel and not a.young and not a.partition:
                # blocks that are NOT boxes (round 6): the same body cut along staircases -- the set-up leans on nothing (algebraic hierarchy, one K^+ column per touched dof, 8 at a time),
                # the apply is the HBM-bound k_fx_symv over every block's own W_b
                secondary("general_nosym", lambda: feti_block(partition="staircase", nel=a.nosym_nel, sub="2,2,2", dense_coarse=False, no_iterative=True, explicit_storage="sym", young=""))
            if not a.no_configs3:
                # BASELINE configs[3]: 4 x 4 x 4 subdomains (64, 8 per GPU at N = 8) of 21^3 elements, G NOT orthonormalised: the projector applies the dense 384 x 384 (G G')^{-1},
                # G G' assembled on the fp64 matrix cores (coarse_problem.GGt_TFLOPs)
                keep3 = {}
                secondary("configs3", lambda: feti_block(sub="4,4,4", nel=21, dense_coarse=True, no_iterative=True, young="", explicit_storage="auto", _keep=keep3))
                row = next((r_ for r_ in _DIRECT_ROWS if r_["nel"] == 21), None)
                if CPU_POOL is not None and keep3.get("f") is not None and out["configs3"].get("value"):
                    # the reference's op sequence timed LIVE on this host, whole iteration: oracle SMALXE + MPGP, direct K^+ on worker processes, B / B' / projector included
                    try:
                        out["configs3"]["cpu_baseline"] = cpu_baseline_whole_iteration(CPU_POOL, ctx, keep3["f"], keep3["G"], keep3["b_dual"], keep3["lb_dual"])
                    except Exception as ex:  # noqa: BLE001
                        out["configs3"]["cpu_baseline"] = {"value": None, "unit": "QPS iterations/s", "kind": "port", "sample": "failed: %r" % (ex,)}
                    finally:
                        CPU_POOL.close()
                    keep3.clear()
                elif row and out["configs3"].get("applies_per_step"):
                    # the reference's direct K^+ for THIS block size was measured by cpu_baseline_direct (SuperLU forward/backward solve of one 21^3 block): no extrapolation here
                    cores = host_threads()
                    t_apply = math.ceil(64 / cores) * row["solve_s"]
                    out["configs3"]["cpu_baseline"] = {"value": 1.0 / (out["configs3"]["applies_per_step"] * t_apply), "unit": "QPS iterations/s", "cores": cores, "kind": "port", "extrapolated": True, "cpu_model": cpu_model(),
                                                       "counts": "K+ solves only (a model: measured solve time x rounds; the live whole-iteration leg needs the worker pool, --no-cpu-pool was given or it failed to start)",
                                                       "sample": "measured: SuperLU (scipy splu, stand-in for the reference's PCCHOLESKY / MUMPS K^+, matinv.c:734-743) forward/backward solve of one 21^3 block = %.4f s; 64 blocks over %d cores "
                                                                 "(%d rounds per F application, perfect parallelism assumed) x %.2f F applications per iteration; B / B' and dual-space work not counted" % (row["solve_s"], cores, math.ceil(64 / cores), out["configs3"]["applies_per_step"])}
            if not a.no_svm:
                def svm_block():
                    rs = run_svm(ctx, a, 60, 6, 0, 1, None)
                    return {"value": rs["value"], "unit": "QPS iterations/s", "ms_per_step": rs["ms_per_step"], "steps": 60, "warmup": 6, "workload": rs["workload"], "steps_by_type": rs["steps_by_type"],
                            "setup_seconds": rs["setup_seconds"], "roofline": rs["roofline"]}
                secondary("configs4", svm_block)
            if not a.no_contact_solve and a.kplus == "explicit" and not a.young and not a.regularize and not a.dense_coarse:
                def contact_block():
                    fc = pa.CubeFeti(tuple(int(v) for v in a.sub.split(",")), a.nel, contact=True)
                    t0 = time.perf_counter()
                    _, _, stc = pa.FETIContactSolve(ctx, fc, explicit=True, explicit_storage="class_orbit", explicit_symmetry=True)
                    wall = time.perf_counter() - t0
                    sx = stc.smalxe
                    return {"what": "pmh_feti_contact_solve: the whole contact solve in ONE library call (multigrid hierarchy, explicit operators, SMALXE + MPGP, rigid-body recovery), host CSR in, u out",
                            "wall_seconds_incl_upload": round(wall, 3), "setup_seconds": round(stc.setup_seconds, 3), "explicit_assembly_seconds": round(stc.explicit_seconds, 3), "explicit_solves": int(stc.explicit_solves),
                            "solve_seconds": round(stc.solve_seconds, 4), "time_to_solution_seconds": round(stc.setup_seconds + stc.solve_seconds, 3), "outer": int(sx.iteration), "inner": int(sx.inner_iter_accu),
                            "hessian_mults": int(sx.inner.nmv), "active_contact_rows": int(stc.n_active)}
                secondary("contact_solve", contact_block)
                if isinstance(out.get("contact_solve"), dict) and out["contact_solve"].get("time_to_solution_seconds"):
                    # the product path for a whole solve is the ONE library call; full_solve keeps the same solve as bench.py's Python orchestration sets it up (the step-rate harness)
                    out["time_to_solution_s"] = out["contact_solve"]["time_to_solution_seconds"]
                    out["time_to_solution_source"] = "contact_solve (pmh_feti_contact_solve: set-up %.2f s + solve %.3f s); the same solve set up by bench.py's Python orchestration: full_solve (%.2f s)" % (
                        out["contact_solve"]["setup_seconds"], out["contact_solve"]["solve_seconds"], out["full_solve"]["time_to_solution_seconds"])
            if not a.no_c2:
                secondary("configs1", lambda: run_c2(ctx, a, a.c2_steps, 30, cpu=not a.no_cpu_baseline, whole_solves=True))
    if rank == 0:
        try:
            out["roofline"]["measured_ceiling"] = measured_ceiling(ctx)
            if out["roofline"].get("bound") == "mfma":  # compute-bound headline kernel: the comparable measured ceiling is the instruction's issue rate, not the copy rate
                out["roofline"]["measured_ceiling"]["mfma_f64_4x4x4_TFLOPs"] = 72.0  # scripts/micro/mfma_f64.hip, 2 waves per SIMD (1.9 GHz held under the GEMM's load: 62)
                out["roofline"]["frac_of_measured_instruction_rate"] = out["roofline"]["achieved"] / 72.0
            else:
                out["roofline"]["frac_of_measured_copy"] = out["roofline"]["achieved"] / out["roofline"]["measured_ceiling"]["copy_GBs"]
        except Exception as ex:  # noqa: BLE001
            out["roofline"]["measured_ceiling"] = "failed: %r" % (ex,)
        if HOST_TRANSPORT and dist is not None:
            out["config"]["transport"] = "host (gloo) -- TEST MODE: the %d ranks share one GPU (PMH_BENCH_TRANSPORT=host); not a multi-GPU measurement" % world
        try:
            with open(a.details, "w") as fh:
                json.dump(out, fh, indent=1)
                fh.write("\n")
        except OSError as ex:
            sys.stderr.write("bench.py: could not write %s: %r\n" % (a.details, ex))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    os.close(real_stdout)
    if rank == 0:
        print(compact_line(out, a.details), flush=True)


if __name__ == "__main__":
    main()
