"""GPU tests of BASELINE configs[4]: dense-row SVM dual Hessian + MPGP, and the row-distributed scalar mode."""
import os

import numpy as np
import pytest

import permon_amd as pa
from permon_amd import problems as P
from permon_amd._lib import check

pytestmark = pytest.mark.gpu


def _solve(ctx, p, distributed=False, rtol=1e-6):
    H = pa.MatCreateSVMDual(ctx, p["X"], p["y"])
    qp = pa.QP(ctx)
    qp.SetOperator(H)
    qp.SetRhs(ctx.vec_from(p["b"]))
    x = ctx.vec_from(p["x0"])
    qp.SetInitialVector(x)
    qp.SetBox(None, ctx.vec_from(p["lb"]), ctx.vec_from(p["ub"]))
    qps = pa.QPS(ctx)
    qps.SetQP(qp)
    qps.SetType("mpgp")
    qps.SetTolerances(rtol=rtol)
    qps.MPGPSetDistributed(distributed)
    st = qps.Solve()
    return H, st, x.to_numpy()


@pytest.mark.parametrize("N,d", [(3000, 64), (1111, 37), (500, 130)])
def test_svm_hessian_apply(N, d):
    ctx = pa.Context(0)
    p = P.svm_dual(N, d)
    H = pa.MatCreateSVMDual(ctx, p["X"], p["y"])
    a = np.random.default_rng(1).uniform(0, 1, N)
    out = ctx.vec(N)
    H.mult(ctx.vec_from(a), out)
    ref = p["y"] * (p["X"] @ (p["X"].T @ (p["y"] * a)))
    assert np.linalg.norm(out.to_numpy() - ref) <= 1e-12 * np.linalg.norm(ref)
    ctx.close()


def test_svm_mpgp_vs_oracle(oracle):
    ctx = pa.Context(0)
    p = P.svm_dual(4000, 64)
    H, st, x = _solve(ctx, p)
    X, y = p["X"], p["y"]
    op = oracle.Op(p["n"], fn=lambda a: y * (X @ (X.T @ (y * a))))
    ref = oracle.mpgp(op, p["b"], p["x0"], oracle.Box(p["n"], lb=p["lb"], ub=p["ub"]), rtol=1e-6)
    assert st.reason == ref["reason"] == 2
    # ~1700 iterations on a rank-deficient Hessian: summation-order rounding moves the count by a few per cent
    assert abs(st.iteration - ref["iteration"]) <= max(3, ref["iteration"] // 6)
    # the dual solution of an SVM is not unique in a (H is rank d): compare what is -- w = X'(y o a) and the objective
    w, w_ref = X.T @ (y * x), X.T @ (y * ref["x"])
    assert np.linalg.norm(w - w_ref) <= 1e-3 * np.linalg.norm(w_ref)
    f = lambda a: 0.5 * np.dot(X.T @ (y * a), X.T @ (y * a)) - a.sum()
    assert abs(f(x) - f(ref["x"])) <= 1e-6 * abs(f(ref["x"]))
    astol = 10 * np.finfo(float).eps  # the reference counts |x - bound| <= astol as on the bound (qpc.c:28)
    assert x.min() >= -astol and x.max() <= 1.0 + astol
    ctx.close()


def test_svm_mpgp_first_iterations_equal_the_oracle(oracle):
    """The whole solve above can only be compared by what it converges to (1 700 iterations on a rank-deficient Hessian amplify the rounding of the dense sums).  The TRAJECTORY is
    compared here: the first 60 iterations take the same steps (CG / expansion / proportioning, Hessian multiplications) and reach the same iterate and the same three gradient norms
    as the oracle's MPGP on a numpy Hessian, on both pass forms."""
    ctx = pa.Context(0)
    p = P.svm_dual(4000, 64)
    X, y = p["X"], p["y"]
    op = oracle.Op(p["n"], fn=lambda a: y * (X @ (X.T @ (y * a))))
    ref = oracle.mpgp(op, p["b"], p["x0"], oracle.Box(p["n"], lb=p["lb"], ub=p["ub"]), rtol=1e-30, max_it=60)
    for pairing in (True, False):
        check(ctx.L.pmh_set_knob(b"svm_pairing", 1 if pairing else 0))
        try:
            H = pa.MatCreateSVMDual(ctx, X, y)
            qp = pa.QP(ctx)
            qp.SetOperator(H)
            qp.SetRhs(ctx.vec_from(p["b"]))
            x = ctx.vec_from(p["x0"])
            qp.SetInitialVector(x)
            qp.SetBox(None, ctx.vec_from(p["lb"]), ctx.vec_from(p["ub"]))
            qps = pa.QPS(ctx)
            qps.SetQP(qp)
            qps.SetType("mpgp")
            qps.SetTolerances(rtol=1e-30, max_it=60)
            st = qps.Solve()
        finally:
            check(ctx.L.pmh_set_knob(b"svm_pairing", 1))
        assert (st.iteration, st.reason) == (ref["iteration"], ref["reason"]) and st.reason == -3  # DIVERGED_ITS: it > max_it (qps.c:694)
        assert (st.ncg, st.nexp, st.nprop, st.nmv) == (ref["ncg"], ref["nexp"], ref["nprop"], ref["nmv"])
        assert np.linalg.norm(x.to_numpy() - ref["x"]) <= 1e-10 * np.linalg.norm(ref["x"])
        for k in ("rnorm", "gfnorm", "gcnorm"):
            assert abs(getattr(st, k) - ref[k]) <= 1e-9 * max(ref["rnorm"], 1e-300), (k, getattr(st, k), ref[k])
    ctx.close()


@pytest.mark.parametrize("N", [6000, 4003, 777])
def test_svm_paired_passes_equal_separate_passes(monkeypatch, N):
    """svm.hip "paired passes": inside MPGP the second pass over X of one Hessian application also does the first pass of the next one (X'(y o p) while the
    gradient is formed, X'(y o x+) for the prepared expansion step while Ap is formed).  Against the separate passes (PMH_SVM_NO_PAIRING=1): the same solve --
    reason, step types within the rounding of the partial sums' order, the same w and objective -- from fewer passes over X."""
    ctx = pa.Context(0)
    p = P.svm_dual(N, 64)  # (4003, 777: rows past the last full group of 8 a wave has in flight)
    X, y = p["X"], p["y"]
    Hp, st_p, x_p = _solve(ctx, p)
    check(ctx.L.pmh_set_knob(b"svm_pairing", 0))
    try:
        Hs, st_s, x_s = _solve(ctx, p)
    finally:
        check(ctx.L.pmh_set_knob(b"svm_pairing", 1))
    assert st_p.reason == st_s.reason == 2
    assert abs(st_p.iteration - st_s.iteration) <= max(3, st_s.iteration // 6) and st_p.nexp > 0
    w_p, w_s = X.T @ (y * x_p), X.T @ (y * x_s)
    assert np.linalg.norm(w_p - w_s) <= 1e-3 * np.linalg.norm(w_s)
    f = lambda a: 0.5 * np.dot(X.T @ (y * a), X.T @ (y * a)) - a.sum()
    assert abs(f(x_p) - f(x_s)) <= 1e-6 * abs(f(x_s))
    astol = 10 * np.finfo(float).eps
    assert x_p.min() >= -astol and x_p.max() <= 1.0 + astol
    # separate passes: two per Hessian application; paired: the P1 after every gradient split saves one, an expansion step that follows such a P1 another one
    # (per counted application; the count also holds the power iterations of the set-up and the P1 passes a proportioning step discards)
    per_s, per_p = Hs.passes() / st_s.nmv, Hp.passes() / st_p.nmv
    assert 2.0 <= per_s <= 2.2
    assert per_p <= per_s - 0.9 * (st_p.nexp - 1) / st_p.nmv, (per_p, per_s, st_p.nexp, st_p.nmv)
    # a fixed run of steps is reproducible bit for bit (no atomics, fixed orders)
    _, st_q, x_q = _solve(ctx, p)
    assert st_q.iteration == st_p.iteration and np.array_equal(x_q, x_p)
    ctx.close()


def test_distributed_scalar_mode_single_rank_communicator(monkeypatch):
    """Row-distributed vectors: every reduction goes through the grouped RCCL all-reduce.  On a 1-rank communicator
    (PMH_COMM_FORCE keeps the collectives on) the result must equal the local mode bit for bit (the local mode on the separate passes the
    distributed mode takes: the paired passes of svm.hip sum other elements per workgroup)."""
    os.environ["PMH_COMM_FORCE"] = "1"
    try:
        ctx = pa.Context(0)
        check(ctx.L.pmh_set_knob(b"svm_pairing", 0))
        ctx.comm_init(0, 1, ctx.comm_unique_id())
        p = P.svm_dual(2000, 64)
        _, st_d, x_d = _solve(ctx, p, distributed=True)
        _, st_l, x_l = _solve(ctx, p, distributed=False)
        assert (st_d.iteration, st_d.nmv, st_d.ncg, st_d.nexp, st_d.nprop, st_d.reason) == (st_l.iteration, st_l.nmv, st_l.ncg, st_l.nexp, st_l.nprop, st_l.reason)
        assert np.array_equal(x_d, x_l)
        # and on a CSR operator (fused epilogue path, speculation off in distributed mode)
        q = P.ex1(100)
        A = pa.CsrMat(ctx, q["n"], q["n"], q["rowptr"], q["col"], q["val"])
        for dist in (True, False):
            qp = pa.QP(ctx)
            qp.SetOperator(pa.Op.from_csr(A))
            qp.SetRhs(ctx.vec_from(q["b"]))
            qp.SetInitialVector(ctx.vec_from(q["x0"]))
            qp.SetBox(None, ctx.vec_from(q["lb"]), None)
            qps = pa.QPS(ctx)
            qps.SetQP(qp)
            qps.SetType("mpgp")
            qps.MPGPSetDistributed(dist)
            st = qps.Solve()
            assert (st.iteration, st.nmv, st.ncg, st.nexp, st.nprop) == (181, 200, 156, 18, 7)
        check(ctx.L.pmh_set_knob(b"svm_pairing", 1))
        ctx.close()
    finally:
        del os.environ["PMH_COMM_FORCE"]
        from permon_amd import _lib

        _lib.load().pmh_set_knob(b"svm_pairing", 1)
