"""GPU tests of BASELINE configs[4]: dense-row SVM dual Hessian + MPGP, and the row-distributed scalar mode."""
import os

import numpy as np
import pytest

import permon_amd as pa
from permon_amd import problems as P

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


def test_distributed_scalar_mode_single_rank_communicator():
    """Row-distributed vectors: every reduction goes through the grouped RCCL all-reduce.  On a 1-rank communicator
    (PMH_COMM_FORCE keeps the collectives on) the result must equal the local mode bit for bit."""
    os.environ["PMH_COMM_FORCE"] = "1"
    try:
        ctx = pa.Context(0)
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
        ctx.close()
    finally:
        del os.environ["PMH_COMM_FORCE"]
