"""A seeded sweep of random box QPs through both MPGP drivers of the library against the oracle: sizes around the kernels' block boundaries (1 ... 4 097), SPD matrices with ragged
rows (band + random symmetric fill, some rows with the diagonal only), bounds that mix finite values with -inf / +inf on either side, feasible and infeasible initial vectors,
expansion variants drawn at random.  Asserted per problem: the same iteration counters (total, Hessian multiplications, CG / expansion / proportioning steps), the same reason, the
iterate to 1e-10 of its norm, the same active set (|x - bound| <= astol), and the three gradient norms to 1e-9.  Index bookkeeping must agree EXACTLY (SURVEY 8d's parity flags)."""
import numpy as np
import pytest
import scipy.sparse as sp

import permon_amd as pa

pytestmark = pytest.mark.gpu

SIZES = (1, 2, 3, 17, 64, 255, 256, 257, 1000, 1024, 2049, 4097)
EXP = (("std", "fixed"), ("std", "opt"), ("std", "optapprox"), ("projcg", "fixed"), ("gfgr", "bb"), ("gf", "opt"), ("g", "optapprox"))


def _problem(seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(SIZES[seed % len(SIZES)])
    band = sp.diags([-1.0, -1.0], [-1, 1], shape=(n, n)) if n > 1 else sp.csr_matrix((1, 1))
    k = int(rng.integers(0, 3 * n + 1))
    i, j = rng.integers(0, n, size=k), rng.integers(0, n, size=k)
    keep = i != j
    fill = sp.csr_matrix((rng.uniform(-0.5, 0.5, size=int(keep.sum())), (i[keep], j[keep])), shape=(n, n))
    off = (band + fill + fill.T).tolil()
    lone = rng.random(n) < 0.1  # rows (and columns) with the diagonal only
    for r in np.nonzero(lone)[0]:
        off[r, :] = 0.0
        off[:, r] = 0.0
    off = off.tocsr()
    off.eliminate_zeros()
    diag = np.asarray(abs(off).sum(axis=1)).ravel() + rng.uniform(0.5, 2.0, size=n)  # strictly diagonally dominant with a margin: SPD, condition number O(10)
    M = (off + sp.diags(diag)).tocsr()
    M.sort_indices()
    b = rng.standard_normal(n)
    lb, ub = rng.uniform(-1.0, 0.0, size=n), rng.uniform(0.0, 1.0, size=n)
    kind = rng.integers(0, 4, size=n)
    lb[kind == 1] = -np.inf
    ub[kind == 2] = np.inf
    lb[kind == 3], ub[kind == 3] = -np.inf, np.inf
    x0 = np.zeros(n) if seed % 3 == 0 else rng.uniform(-2.0, 2.0, size=n)  # infeasible starts are projected by the solver (mpgp.c:497)
    return M, b, x0, lb, ub, EXP[seed % len(EXP)]


@pytest.fixture(scope="module")
def ctx():
    c = pa.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("seed", range(28))
def test_random_box_qp_parity(ctx, oracle, seed):
    M, b, x0, lb, ub, (exptype, explen) = _problem(seed)
    n = M.shape[0]
    ref = oracle.mpgp(oracle.Op(n, csr=oracle.Csr.from_scipy(M)), b, x0, oracle.Box(n, lb=lb, ub=ub), rtol=1e-8, max_it=5000, exptype=exptype, explengthtype=explen)
    for unfused in (False, True):
        A = pa.CsrMat(ctx, n, n, M.indptr, M.indices, M.data)
        qp = pa.QP(ctx)
        qp.SetOperator(pa.Op.from_csr(A))
        qp.SetRhs(ctx.vec_from(b))
        x = ctx.vec_from(x0)
        qp.SetInitialVector(x)
        qp.SetBox(None, ctx.vec_from(lb), ctx.vec_from(ub))
        qps = pa.QPS(ctx)
        qps.SetQP(qp)
        qps.SetType("mpgp")
        qps.SetTolerances(rtol=1e-8, max_it=5000)
        qps.MPGPSetExpansionType(exptype, explen)
        qps.MPGPSetUnfused(unfused)
        st = qps.Solve()
        xs = x.to_numpy()
        tag = (seed, n, exptype, explen, unfused)
        assert (st.iteration, st.reason, st.nmv, st.ncg, st.nexp, st.nprop) == (ref["iteration"], ref["reason"], ref["nmv"], ref["ncg"], ref["nexp"], ref["nprop"]), tag
        assert np.linalg.norm(xs - ref["x"]) <= 1e-10 * max(np.linalg.norm(ref["x"]), 1e-300), tag
        astol = 10 * np.finfo(float).eps  # qpc.c:28
        for bound in (lb, ub):
            fin = np.isfinite(bound)
            assert np.array_equal(np.abs(xs[fin] - bound[fin]) <= astol, np.abs(ref["x"][fin] - bound[fin]) <= astol), tag
        assert np.all(xs >= lb - astol) and np.all(xs <= ub + astol), tag
        for k in ("rnorm", "gfnorm", "gcnorm"):
            assert abs(getattr(st, k) - ref[k]) <= 1e-9 * max(ref["rnorm"], ref["norm_rhs"] * 1e-8) + 4e-16 * max(ref["norm_rhs"], 1.0), (tag, k)  # (+ the rounding level of a residual that is exactly zero)


def _eq_problem(seed):
    rng = np.random.default_rng(5000 + seed)
    n = int((20, 64, 257, 1000)[seed % 4])
    m = int((1, 3, 8)[seed % 3])
    band = sp.diags([-1.0, 2.0 + rng.uniform(0.2, 1.0), -1.0], [-1, 0, 1], shape=(n, n)).tocsr()  # SPD, condition number O(10)
    G0 = sp.random(m, n, density=min(1.0, 12.0 / n), random_state=int(seed), data_rvs=rng.standard_normal).toarray() + 0.0
    G0[np.arange(m), rng.choice(n, size=m, replace=False)] += 1.0  # full row rank
    Q, _ = np.linalg.qr(G0.T)  # orthonormal rows
    orth = seed % 2 == 0
    G = sp.csr_matrix(Q.T if orth else G0)
    G.sort_indices()
    b = rng.standard_normal(n)
    lb = rng.uniform(-0.5, 0.0, size=n)
    lb[rng.random(n) < 0.3] = -np.inf
    return band, G, orth, b, lb


@pytest.mark.parametrize("seed", range(12))
def test_random_equality_constrained_qp_smalxe_parity(ctx, oracle, seed):
    """min 1/2 x'Ax - b'x  s.t.  G x = 0, x >= lb on random data through SMALXE (inner MPGP): outer iterations, M1 / rho updates, the state machine AND the inner iteration total
    exactly as the oracle's, on the default path (fused chain / fused projector epilogues) and on the separate launches (knobs chain = 0, gt_fusion = 0, MPGP unfused): measured
    with tests/tools/inner_totals.py, all 12 seeds x 4 paths agree to the iteration (round 4 allowed 2 %).  The solution to 1e-6 -- orthonormal and general G."""
    A, G, orth, b, lb = _eq_problem(seed)
    n = A.shape[0]
    pfo = oracle.Qppf(oracle.Csr.from_scipy(G), orthonormal=orth)
    ref = oracle.smalxe(oracle.Op(n, csr=oracle.Csr.from_scipy(A)), b, np.zeros(n), oracle.Box(n, lb=lb), pfo, rtol=1e-7)
    try:
        for chain, gtf, unfused in ((1, 1, False), (0, 1, False), (0, 0, True)):
            pa._lib.check(ctx.L.pmh_set_knob(b"chain", chain))
            pa._lib.check(ctx.L.pmh_set_knob(b"gt_fusion", gtf))
            Ad = pa.CsrMat(ctx, n, n, A.indptr, A.indices, A.data)
            qp = pa.QP(ctx)
            qp.SetOperator(pa.Op.from_csr(Ad))
            qp.SetRhs(ctx.vec_from(b))
            x = ctx.vec(n)
            qp.SetInitialVector(x)
            qp.SetBox(None, ctx.vec_from(lb), None)
            qp.SetEq(pa.QPPF.from_scipy(ctx, G, orthonormal=orth))
            qps = pa.QPS(ctx)
            qps.SetQP(qp)
            qps.SetType("smalxe")
            qps.SetTolerances(rtol=1e-7)
            if unfused:
                qps.MPGPSetUnfused(True)
            st = qps.Solve()
            tag = (seed, n, G.shape[0], orth, chain, gtf, unfused)
            assert (st.reason, st.iteration, st.M1_updates, st.rho_updates, st.state) == (ref["reason"], ref["iteration"], ref["M1_updates"], ref["rho_updates"], ref["state"]), tag
            assert st.inner_iter_accu == ref["inner_iter_accu"], tag
            xs = x.to_numpy()
            assert np.linalg.norm(xs - ref["u"]) <= 1e-6 * max(np.linalg.norm(ref["u"]), 1e-300), tag
            assert np.linalg.norm(G @ xs) <= 1e-6 * max(np.linalg.norm(b), 1.0) and np.all(xs >= lb - 1e-14), tag
    finally:
        ctx.L.pmh_set_knob(b"chain", 1)
        ctx.L.pmh_set_knob(b"gt_fusion", 1)
