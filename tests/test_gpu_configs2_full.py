"""BASELINE configs[2] at FULL size (2x2x2 cubes of 43^3 Q1 elements, 2 044 416 dof, n_lambda = 102 268) with the bench defaults:
the contact problem is SOLVED through every K^+ the bench line reports -- the explicit local dual operators (headline), the
inner-Krylov K^+ with the fp16 V-cycle PC and with the strict fp64 cycle -- and must give the same SMALXE / MPGP counts
(10 outer / 108 inner / 185 Hessian multiplications / 41 CG + 67 expansion) and a feasible, complementary solution.
One module-scoped problem (generation + hierarchy + explicit assembly ~ 1 min on an MI355X)."""
import numpy as np
import pytest

import permon_amd as pa
from permon_amd.chain import FetiDualQP

pytestmark = pytest.mark.gpu

COUNTS = dict(outer=10, inner=108, nmv=185, ncg=41, nexp=67, nprop=0)


@pytest.fixture(scope="module")
def c2():
    ctx = pa.Context(0)
    f = pa.CubeFeti((2, 2, 2), 43, contact=True)
    assert f.N == 2044416 and f.n_lambda == 102268 and f.n_ineq == 7744
    G, e = f.coarse(orthonormalize=True)
    hier = pa.box_mg_hierarchy([f.Ki] * 8, [(44, 44, 44)] * 8, 3, min_nodes=400)  # bench default at 8 blocks per GPU
    q = FetiDualQP(ctx, f.subset(range(8)), G, e, f.c, f.lb, orthonormal=True, kplus_rtol=1e-9, mg_hierarchy=hier, mg_degree=2, mg_precision="fp16", bsr3=True,
                   explicit=dict(rtol=1e-12))
    yield ctx, f, G, e, hier, q
    ctx.close()


def _counts(st):
    return dict(outer=st.iteration, inner=st.inner_iter_accu, nmv=st.inner.nmv, ncg=st.inner.ncg, nexp=st.inner.nexp, nprop=st.inner.nprop)


def _solve(q):
    q.lam.set(0.0)
    st = q.solve_smalxe(rtol=1e-5)
    q.qps.Destroy()
    return st, q.lam.to_numpy()


def test_full_size_solve_same_counts_on_every_kplus(c2):
    ctx, f, G, e, hier, q = c2
    E = q.E
    assert E.n_gamma.tolist() == [24384, 18880, 24384, 18880, 22578, 17031, 22578, 17031] and E.assemble_stats()[0] == 33288  # congruent cubes share the 33 288 boundary columns
    st_ex, lam_ex = _solve(q)  # explicit local dual operators
    assert st_ex.reason == 2 and _counts(st_ex) == COUNTS
    q.Kplus.attach_explicit(None)
    st_16, lam_16 = _solve(q)  # inner-Krylov K^+, fp16 V-cycle PC (rtol 1e-9)
    assert st_16.reason == 2 and _counts(st_16) == COUNTS
    # K^+ f with the fp16 PC vs the fp64 PC at 2.04 M dof
    rhs = ctx.vec_from(np.random.default_rng(3).standard_normal(f.N))
    u16, u64 = ctx.vec(f.N), ctx.vec(f.N)
    q.Kplus.set_tolerances(1e-11, max_it=200)
    q.Kplus.mult(rhs, u16)
    its16 = q.Kplus.last_iterations()[0]
    old = q.Kplus.mg
    q.Kplus.set_pc_mg(hier, degree=2, precision="fp64")
    old.destroy()
    q.Kplus.mult(rhs, u64)
    its64 = q.Kplus.last_iterations()[0]
    a, b = u16.to_numpy(), u64.to_numpy()
    assert np.linalg.norm(a - b) <= 1e-8 * np.linalg.norm(b) and abs(its16 - its64) <= 3
    q.Kplus.set_tolerances(1e-9, max_it=20000)
    st_64, lam_64 = _solve(q)  # strict fp64
    assert st_64.reason == 2 and _counts(st_64) == COUNTS
    q.Kplus.attach_explicit(E)
    for lam in (lam_16, lam_64):
        assert np.linalg.norm(lam - lam_ex) <= 1e-6 * np.linalg.norm(lam_ex)
    assert abs(st_16.rnorm - st_ex.rnorm) <= 1e-5 * st_ex.rnorm and abs(st_64.rnorm - st_ex.rnorm) <= 1e-5 * st_ex.rnorm


def test_full_size_solution_properties(c2):
    """Size-independent properties of the solution: G lambda = e, dual feasibility, no penetration, glued interfaces, complementarity."""
    ctx, f, G, e, hier, q = c2
    st, _ = _solve(q)
    assert st.reason == 2
    lam = q.dual_solution()
    n = f.n_lambda
    assert lam[f.n_eq:].min() >= -1e-12
    assert np.linalg.norm(G @ lam - e) <= 1e-5 * max(1.0, np.linalg.norm(e))
    u, Fl_minus_d = q.primal_solution(G)
    Ru = f.kernel_matrix()
    tight = (np.arange(n) < f.n_eq) | (lam > 1e-8 * np.abs(lam).max())
    BR = (f.B @ Ru).toarray()
    alpha = np.linalg.lstsq(BR[tight], Fl_minus_d[tight], rcond=None)[0]
    uu = u + Ru @ alpha
    Bu, scale = f.B @ uu, np.abs(uu).max()
    assert np.abs(Bu[:f.n_eq] - f.c[:f.n_eq]).max() <= 1e-3 * scale
    assert (Bu[f.n_eq:] - f.c[f.n_eq:]).max() <= 1e-3 * scale
    gap = f.c[f.n_eq:] - Bu[f.n_eq:]
    assert np.abs(lam[f.n_eq:] * gap).max() <= 1e-3 * scale * np.abs(lam).max()
    assert 7000 < (lam[f.n_eq:] > 0).sum() <= 7744  # the contact zone (7 483 active rows in profiles/r01_solve_configs2_smalxe.jsonl)
    # the explicit F and the inner-Krylov F agree on the solution
    y1, y2 = ctx.vec(n), ctx.vec(n)
    lv = ctx.vec_from(lam)
    q.F.mult(lv, y1)
    q.Kplus.attach_explicit(None)
    q.Kplus.set_tolerances(1e-12, max_it=200)
    q.F.mult(lv, y2)
    q.Kplus.set_tolerances(1e-9, max_it=20000)
    q.Kplus.attach_explicit(q.E)
    assert np.linalg.norm(y1.to_numpy() - y2.to_numpy()) <= 1e-9 * np.linalg.norm(y2.to_numpy())


@pytest.mark.parametrize("storage", ["class_orbit", "class_sym"])
def test_full_size_headline_path(c2, storage):
    """The bench's headline configurations at full size: ONE symmetric class matrix for the 8 congruent cubes -- "class_orbit": only the rows of the 715 orbit
    representatives under the cube's 48 symmetries, applied as a GEMM on the fp64 matrix instruction (k_fxo_gemm, the default); "class_sym": the lower
    block-triangle in tiles (k_fxs_symm8), assembled by symmetry -- G orthonormalised implicitly: the same solver counts as every other K^+, the same dual
    solution, and F lambda equal to the per-block operators' (33 288 direct solves) to the set-up tolerance."""
    ctx, f, G, e, hier, q = c2
    st_ref, lam_ref = _solve(q)
    G0, e0 = f.coarse(orthonormalize=False)
    qh = FetiDualQP(ctx, f.subset(range(8)), G0, e0, f.c, f.lb, orthonormal="implicit", kplus_rtol=1e-9, mg_hierarchy=hier, mg_degree=2, mg_precision="fp16", bsr3=True,
                    explicit=dict(rtol=1e-12, storage=storage, symmetry=dict(dims=(44, 44, 44), ndof=3, orbit=storage == "class_orbit")))
    assert qh.explicit_symmetries == 48 and qh.explicit_storage == storage
    n_solves, secs = qh.E.assemble_stats()
    assert 700 <= n_solves <= 715 + 64 and qh.E.dense_bytes < (0.2e9 if storage == "class_orbit" else 4.7e9)
    st, lam = _solve(qh)
    assert st.reason == 2 and _counts(st) == COUNTS
    assert np.linalg.norm(lam - lam_ref) <= 1e-6 * np.linalg.norm(lam_ref)
    x = ctx.vec_from(np.random.default_rng(12).standard_normal(f.n_lambda))
    y1, y2 = ctx.vec(f.n_lambda), ctx.vec(f.n_lambda)
    q.F.mult(x, y1)
    qh.F.mult(x, y2)
    assert np.linalg.norm(y1.to_numpy() - y2.to_numpy()) <= 1e-9 * np.linalg.norm(y1.to_numpy())
