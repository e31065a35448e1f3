"""World-size-2 worker for tests/test_gpu_dist_transport.py: the LIBRARY's distributed arithmetic (pmh_mpgp `distributed`, pmh_gluing_mult_transpose, the SVM w exchange,
the grouped MPGP scalars) on two processes that share the box's one GPU.  RCCL cannot put two ranks on one device, so the collectives ride on the library's host transport
(pmh_comm_set_host_transport) carried by gloo -- the arithmetic on either side of the exchange is the code the RCCL build runs.

usage: dist_transport_worker.py {svm|feti_iterative|feti_explicit}   (RANK / WORLD_SIZE / MASTER_PORT in the environment)"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import permon_amd as pa  # noqa: E402

CALLS = {0: 0, 1: 0, 2: 0}


def transport(op, arr):
    CALLS[op] += 1
    if op == 2 or arr.size == 0:
        dist.barrier()
        return
    t = torch.from_numpy(arr)  # a view of the library's pinned staging buffer: reduced in place
    dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MIN)


def bcast_ok(flag):
    t = torch.tensor([1.0 if flag else 0.0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item() == 1.0)


def case_svm(ctx, rank, world):
    N, d = 6000, 64
    rng = np.random.default_rng(7)
    X = rng.standard_normal((N, d))
    y = np.sign(X @ np.random.default_rng(8).standard_normal(d) + 0.1 * rng.standard_normal(N))
    y[y == 0] = 1.0

    def solve(Xl, yl, distributed, max_it):
        n = Xl.shape[0]
        H = pa.MatCreateSVMDual(ctx, np.ascontiguousarray(Xl), np.ascontiguousarray(yl))
        qp = pa.QP(ctx)
        qp.SetOperator(H)
        qp.SetRhs(ctx.vec_from(np.ones(n)))
        x = ctx.vec(n)
        qp.SetInitialVector(x)
        qp.SetBox(None, ctx.vec(n), ctx.vec_from(np.ones(n)))
        qps = pa.QPS(ctx)
        qps.SetQP(qp)
        qps.SetType("mpgp")
        qps.SetTolerances(rtol=1e-6, max_it=max_it)
        qps.MPGPSetDistributed(distributed)
        st = qps.Solve()
        return st, x.to_numpy().copy()

    def objective(a):  # 1/2 a'Ha - 1'a with H = diag(y) X X' diag(y)
        w = X.T @ (y * a)
        return 0.5 * float(w @ w) - float(a.sum())

    # the single-rank references: the whole sample set on this process, no transport.  PAIRING (argv[2]): the paired passes over X inside MPGP (the default; on two ranks w and
    # the feasible step length are completed across the ranks between the passes) or the separate passes (PMH_SVM_NO_PAIRING=1)
    from permon_amd._lib import check as _check

    _check(ctx.L.pmh_set_knob(b"svm_pairing", 0 if (len(sys.argv) > 2 and sys.argv[2] == "separate") else 1))
    p_before = 0
    ref60, x60 = solve(X, y, False, 60)
    ref, x_ref = solve(X, y, False, 10000)
    assert ref.reason > 0, ref.reason
    ctx.comm_set_host_transport(rank, world, transport)
    lo, hi = rank * N // world, (rank + 1) * N // world

    def same_on_all_ranks(vals, what):
        t = torch.tensor(vals, dtype=torch.float64)
        t2 = t.clone()
        dist.all_reduce(t2, op=dist.ReduceOp.MAX)
        assert torch.equal(t, t2), "ranks disagree on " + what

    # (1) 60 iterations: the same decisions step by step (the Hessian has rank 64 on 6000 unknowns -- over thousands of iterations a rounding-level difference in the order of
    # the sums moves the iteration count by a percent, so the exact comparison is made on a trajectory short enough to stay together) and x to rounding
    st, x_loc = solve(X[lo:hi], y[lo:hi], True, 60)
    assert CALLS[0] > st.nmv and CALLS[1] >= 1, CALLS  # w per Hessian application + the grouped scalars; QPCFeas's MIN went out too
    got = (st.iteration, st.nmv, st.ncg, st.nexp, st.nprop, st.reason)
    exp = (ref60.iteration, ref60.nmv, ref60.ncg, ref60.nexp, ref60.nprop, ref60.reason)
    same_on_all_ranks(got + (st.rnorm,), "the counters / the replicated residual norm")
    assert got == exp, (got, exp)
    err = np.linalg.norm(x_loc - x60[lo:hi]) / max(np.linalg.norm(x60), 1e-300)
    assert err <= 1e-10, err
    # active sets: identical up to components that sit within rounding of a bound (the sums behind the step are taken in a different order on two ranks)
    tol = 10 * np.finfo(float).eps
    flips = int(np.count_nonzero((np.abs(x_loc) <= tol) != (np.abs(x60[lo:hi]) <= tol)) + np.count_nonzero((np.abs(x_loc - 1.0) <= tol) != (np.abs(x60[lo:hi] - 1.0) <= tol)))
    assert flips <= 3, "active sets differ in %d components" % flips
    assert abs(st.rnorm - ref60.rnorm) <= 1e-9 * ref60.rnorm
    # (2) the whole solve: converged on both, the same minimum, iteration counts within a few percent
    st2, x2 = solve(X[lo:hi], y[lo:hi], True, 10000)
    same_on_all_ranks((st2.iteration, st2.nmv, st2.reason, st2.rnorm), "the full solve")
    assert st2.reason == ref.reason and abs(st2.iteration - ref.iteration) <= 0.1 * ref.iteration, (st2.iteration, ref.iteration)
    xs = [torch.zeros(N // world, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(xs, torch.from_numpy(x2))
    xa = torch.cat(xs).numpy()
    f2, f1 = objective(xa), objective(x_ref)
    assert abs(f2 - f1) <= 1e-8 * abs(f1), (f2, f1)
    return "svm: 60 its on 2 ranks %s == 1 rank, |x - x_ref| = %.1e; full solve %d vs %d its, objective %.10e vs %.10e; %d sum / %d min exchanges" % (got, err, st2.iteration, ref.iteration, f2, f1, CALLS[0], CALLS[1])


def case_feti(ctx, rank, world, explicit):
    from permon_amd.chain import FetiDualQP

    f = pa.CubeFeti((2, 2, 2), 4, contact=True)
    G, e = f.coarse()
    kw = dict(kplus_rtol=1e-12, explicit=dict(rtol=1e-12, storage="sym") if explicit else None)
    # single-rank reference on this process (all 8 blocks, no transport)
    q1 = FetiDualQP(ctx, f.subset(range(f.nsub)), G, e, f.c, f.lb, **kw)
    lam = np.random.default_rng(3).standard_normal(f.n_lambda)
    y1 = ctx.vec(f.n_lambda)
    q1.F.mult(ctx.vec_from(lam), y1)
    F1 = y1.to_numpy().copy()
    st1 = q1.solve_smalxe(rtol=1e-6)
    lam1 = q1.lam.to_numpy().copy()
    assert st1.reason > 0
    # two ranks: 4 blocks each, lambda replicated, B u summed through the transport
    ctx.comm_set_host_transport(rank, world, transport)
    per = f.nsub // world
    loc = f.subset(range(rank * per, (rank + 1) * per))
    q = FetiDualQP(ctx, loc, G, e, f.c, f.lb, **kw)
    c0 = CALLS[0]
    # (a) pmh_gluing_mult_transpose alone: the partial B u of the ranks sum to the whole
    u = np.random.default_rng(5).standard_normal(f.N)
    lo, hi = f.block_rowstart[rank * per], f.block_rowstart[(rank + 1) * per]
    t = ctx.vec(f.n_lambda)
    q.B.mult_transpose(ctx.vec_from(u[lo:hi]), t)
    assert CALLS[0] == c0 + 1
    Bu = f.B @ u
    assert np.linalg.norm(t.to_numpy() - Bu) <= 1e-13 * np.linalg.norm(Bu)
    # (b) F lambda
    y = ctx.vec(f.n_lambda)
    q.F.mult(ctx.vec_from(lam), y)
    errF = np.linalg.norm(y.to_numpy() - F1) / np.linalg.norm(F1)
    assert errF <= 1e-10, errF
    # (c) the whole SMALXE + MPGP solve: same counts, same multipliers, bit-identical replicated state on both ranks
    st = q.solve_smalxe(rtol=1e-6)
    got = (st.iteration, st.inner_iter_accu, st.inner.nmv, st.inner.ncg, st.inner.nexp, st.inner.nprop, st.reason)
    exp = (st1.iteration, st1.inner_iter_accu, st1.inner.nmv, st1.inner.ncg, st1.inner.nexp, st1.inner.nprop, st1.reason)
    assert got == exp, (got, exp)
    lam2 = q.lam.to_numpy().copy()
    errl = np.linalg.norm(lam2 - lam1) / np.linalg.norm(lam1)
    assert errl <= 1e-8, errl
    tl = torch.from_numpy(lam2.copy())
    tm = tl.clone()
    dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    assert torch.equal(tl, tm), "replicated lambda differs between the ranks"
    return "feti (%s K^+): 2 ranks %s == 1 rank, |F - F_1| = %.1e, |lambda - lambda_1| = %.1e, %d exchanges" % ("explicit" if explicit else "iterative", got, errF, errl, CALLS[0])


def main():
    case = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["MASTER_PORT"], rank=rank, world_size=world)
    ctx = pa.Context(0)  # both ranks on the box's one GPU
    try:
        msg = case_svm(ctx, rank, world) if case == "svm" else case_feti(ctx, rank, world, case == "feti_explicit")
        ok = True
    except Exception:  # noqa: BLE001
        import traceback

        traceback.print_exc()
        msg, ok = "failed", False
    ok_all = bcast_ok(ok)
    ctx.comm_set_host_transport(0, 1, None)
    ctx.close()
    dist.destroy_process_group()
    if not (ok and ok_all):
        sys.exit(1)
    print("rank %d ok: %s" % (rank, msg))


if __name__ == "__main__":
    main()
