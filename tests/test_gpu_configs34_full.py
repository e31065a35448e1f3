"""BASELINE configs[3] and configs[4] at the sizes bench.py TIMES them (both were only tested in miniature).
configs[3]: 4x4x4 subdomains of 21^3 Q1 elements (64 blocks, 8 per GPU at N = 8), G not orthonormalised: dense 384 x 384 coarse inverse, GG' on the fp64 matrix
cores; the orbit-storage explicit operators (what the `configs3` block of the bench line runs) against the inner-Krylov K^+ on the same dual QP.
configs[4]: the 5 M x 64 SVM dual: size-independent properties of the Hessian application (symmetry, linearity, the known first column block) and the paired passes of
MPGP against the separate ones."""
import os

import numpy as np
import pytest

import permon_amd as pa
from permon_amd._lib import check
from permon_amd.chain import FetiDualQP

pytestmark = pytest.mark.gpu


def _counts(st):
    return dict(outer=st.iteration, inner=st.inner_iter_accu, nmv=st.inner.nmv, ncg=st.inner.ncg, nexp=st.inner.nexp, nprop=st.inner.nprop)


def test_configs3_full_size_orbit_operators_against_inner_krylov():
    ctx = pa.Context(0)
    f = pa.CubeFeti((4, 4, 4), 21, contact=True)
    assert f.nsub == 64 and f.congruent
    G0, e0 = f.coarse(orthonormalize=False)
    assert G0.shape[0] == 384
    q = FetiDualQP(ctx, f.subset(range(64)), G0, e0, f.c, f.lb, orthonormal=False, kplus_rtol=1e-9, mg_box=dict(dims=[(22, 22, 22)] * 64, ndof=3, min_nodes=400), mg_degree=2, mg_precision="fp16",
                   bsr3=True, explicit=dict(rtol=1e-12, storage="class_orbit", symmetry=dict(dims=(22, 22, 22), ndof=3)))
    assert q.explicit_storage == "class_orbit" and q.explicit_symmetries == 48 and q.pf.m == 384
    s = q.pf.setup_stats()
    assert s[0] > 0 and s[1] > 0  # GG' went through the fp64-MFMA kernel
    q.lam.set(0.0)
    st_ex = q.solve_smalxe(rtol=1e-5)
    lam_ex = q.lam.to_numpy().copy()
    q.qps.Destroy()
    assert st_ex.reason == 2
    # F lambda: orbit GEMM against the block-wise CG (tight tolerance), at the solution and on a random vector
    n = f.n_lambda
    xr = ctx.vec_from(np.random.default_rng(21).standard_normal(n))
    y1, y2 = ctx.vec(n), ctx.vec(n)
    q.F.mult(xr, y1)
    q.Kplus.attach_explicit(None)
    q.Kplus.set_tolerances(1e-12, max_it=400)
    q.F.mult(xr, y2)
    a, b = y1.to_numpy(), y2.to_numpy()
    assert np.linalg.norm(a - b) <= 1e-9 * np.linalg.norm(b), np.linalg.norm(a - b) / np.linalg.norm(b)
    q.Kplus.set_tolerances(1e-9, max_it=20000)
    # the same solve through the inner-Krylov K^+: same counts, same multipliers
    q.lam.set(0.0)
    st_it = q.solve_smalxe(rtol=1e-5)
    lam_it = q.lam.to_numpy().copy()
    q.qps.Destroy()
    q.Kplus.attach_explicit(q.E)
    assert st_it.reason == 2 and _counts(st_it) == _counts(st_ex), (_counts(st_it), _counts(st_ex))
    assert np.linalg.norm(lam_it - lam_ex) <= 1e-6 * np.linalg.norm(lam_ex)
    # solution properties: G lambda = e through the dense coarse inverse, dual feasibility
    lam = q.dual_solution()
    assert lam[f.n_eq:].min() >= -1e-12
    assert np.linalg.norm(G0 @ lam - e0) <= 1e-5 * max(1.0, np.linalg.norm(e0))
    ctx.close()


def test_configs4_full_size_hessian_properties_and_pairing():
    ctx = pa.Context(0)
    N, d = 5000000, 64
    rng = np.random.default_rng(7)
    w_true = np.random.default_rng(8).standard_normal(d)
    X = np.empty((N, d))
    y = np.empty(N)
    for s in range(0, N, 500000):
        X[s:s + 500000] = rng.standard_normal((500000, d))
        y[s:s + 500000] = np.sign(X[s:s + 500000] @ w_true + 0.1 * rng.standard_normal(500000))
    y[y == 0] = 1.0
    H = pa.MatCreateSVMDual(ctx, X, y)
    r2 = np.random.default_rng(5)
    u, v = r2.standard_normal(N), r2.standard_normal(N)
    du, dv, hu, hv, hw = ctx.vec_from(u), ctx.vec_from(v), ctx.vec(N), ctx.vec(N), ctx.vec(N)
    H.mult(du, hu)
    H.mult(dv, hv)
    Hu, Hv = hu.to_numpy(), hv.to_numpy()
    # symmetry v'Hu = u'Hv and positive semi-definiteness u'Hu = ||X'(y o u)||^2
    assert abs(v @ Hu - u @ Hv) <= 1e-11 * (np.linalg.norm(v) * np.linalg.norm(Hu))
    wu = X.T @ (y * u)
    assert abs(u @ Hu - wu @ wu) <= 1e-11 * (wu @ wu)
    # the rows: (H u)_i = y_i x_i . w on a sample of rows
    idx = r2.integers(0, N, 4096)
    ref = y[idx] * (X[idx] @ wu)
    assert np.max(np.abs(Hu[idx] - ref)) <= 1e-10 * np.max(np.abs(ref))
    # linearity
    dw = ctx.vec_from(2.0 * u - 0.5 * v)
    H.mult(dw, hw)
    assert np.linalg.norm(hw.to_numpy() - (2.0 * Hu - 0.5 * Hv)) <= 1e-11 * np.linalg.norm(Hu)
    del X
    # MPGP on the box 0 <= a <= 1: the paired passes over X (one pass per application in a run of expansion steps) against the separate passes: same steps, same iterate
    def run(no_pairing):
        check(ctx.L.pmh_set_knob(b"svm_pairing", 0 if no_pairing else 1))  # (the same operator H serves both runs: the switch is read per product)
        qp = pa.QP(ctx)
        qp.SetOperator(H)
        qp.SetRhs(ctx.vec_from(np.ones(N)))
        x = ctx.vec(N)
        qp.SetInitialVector(x)
        qp.SetBox(None, ctx.vec(N), ctx.vec_from(np.ones(N)))
        qps = pa.QPS(ctx)
        qps.SetQP(qp)
        qps.SetType("mpgp")
        qps.SetUp()
        p0 = H.passes()
        st = qps.RunFixed(30)
        return (st.ncg, st.nexp, st.nprop, st.nmv), x.to_numpy().copy(), H.passes() - p0
    c_sep, x_sep, p_sep = run(True)
    c_pair, x_pair, p_pair = run(False)
    check(ctx.L.pmh_set_knob(b"svm_pairing", 1))
    assert c_sep == c_pair, (c_sep, c_pair)
    assert np.linalg.norm(x_pair - x_sep) <= 1e-10 * np.linalg.norm(x_sep)
    assert 2 * c_sep[3] <= p_sep <= 2 * (c_sep[3] + 2) and p_pair < 0.75 * p_sep  # two passes over X per application when separate (the speculated product after the last step included); the pairing removes a good part of them
    ctx.close()
