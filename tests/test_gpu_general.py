"""The general (non-congruent) path of the explicit local dual operators at a size where it is not a toy: 2 x 2 x 2 subdomains of 27^3 elements (526 848 dof) made of
8 DIFFERENT materials (CubeFeti(young=...): K_s = E_s K_1, no two blocks bit-identical), so that nothing the congruent-cube headline relies on applies --
pmh_csr_block_classes finds 8 classes, `auto` falls to per-block symmetric storage (PMH_FX_SYM, k_fx_symv: the HBM-bound kernel), every column of every W_b
comes from its own K^+ solve (no class sharing, no set-up by symmetry).  Checked against the inner-Krylov K^+ the reference would run (MATINV's KSP): F to 1e-9,
sampled columns of the W_b against direct solves, identical SMALXE / MPGP counts and the same dual solution for the contact problem.
(On this per-block path the full 43^3 size needs 24 384 K^+ applications for the set-up, ~5 min.  Since round 4 symmetric boxes of different materials take the orbit storage with
every class on the CLOSURE of its touched set under the cube's group -- test_general_closed_orbit_path below; the driver-run bench line carries the 43^3 case on it as `general`.)"""
import numpy as np
import pytest

import permon_amd as pa
from permon_amd.chain import FetiDualQP

pytestmark = pytest.mark.gpu
NEL = 27


@pytest.fixture(scope="module")
def general():
    ctx = pa.Context(0)
    young = [1.0 + 0.25 * i for i in range(8)]
    f = pa.CubeFeti((2, 2, 2), NEL, contact=True, young=young)
    assert not f.congruent and f.N == 8 * 3 * (NEL + 1) ** 3
    G, e = f.coarse(orthonormalize=True)
    loc = f.subset(range(8))
    nn = NEL + 1
    box = dict(dims=[(nn, nn, nn)] * 8, ndof=3, min_nodes=400)
    qi = FetiDualQP(ctx, loc, G, e, f.c, f.lb, kplus_rtol=1e-12, mg_box=box, mg_precision="fp16", bsr3=True)
    qe = FetiDualQP(ctx, loc, G, e, f.c, f.lb, kplus_rtol=1e-12, mg_box=box, mg_precision="fp16", bsr3=True,
                    explicit=dict(rtol=1e-12, storage="auto", symmetry=dict(dims=(nn, nn, nn), ndof=3)))  # the symmetry hint must be ignored: 8 classes
    qc = FetiDualQP(ctx, loc, G, e, f.c, f.lb, kplus_rtol=1e-12, mg_box=box, mg_precision="fp16", bsr3=True,
                            explicit=dict(rtol=1e-12, storage="class_orbit", symmetry=dict(dims=(nn, nn, nn), ndof=3, close=True)))  # every class on the closure of its touched set
    yield ctx, f, loc, qi, qe, qc
    ctx.close()


def test_general_decomposition_takes_the_per_block_path(general):
    ctx, f, loc, qi, qe, _qc = general
    cls = pa.csr_block_classes(loc["block_rowstart"], loc["K"])
    assert sorted(cls.tolist()) == list(range(8))  # no two blocks are equal
    assert qe.explicit_storage == "sym" and qe.explicit_symmetries == 1
    n_solves, seconds = qe.E.assemble_stats()
    assert n_solves == int(qe.E.n_gamma.sum())  # one K^+ solve per touched dof of every block: nothing shared
    # the blocks touch different numbers of dofs (Dirichlet / contact faces): 2 x 2 distinct sizes
    assert len(set(qe.E.n_gamma.tolist())) == 4


def test_general_closed_orbit_path(general):
    """The same 8-material decomposition through the orbit storage with the class sets closed under the cube's group (round 4): 48 operations per class, 24 x fewer set-up solves
    than one per touched dof, F equal to the inner-Krylov F and to the per-block operators, the same SMALXE counts."""
    ctx, f, loc, qi, qe, qc = general
    assert qc.explicit_storage == "class_orbit" and qc.explicit_symmetries == 48
    n_closed, _ = qc.E.assemble_stats()
    n_open, _ = qe.E.assemble_stats()
    assert n_closed * 20 < n_open
    lam = np.random.default_rng(12).standard_normal(f.n_lambda)
    lv, y0, y1, y2 = ctx.vec_from(lam), ctx.vec(f.n_lambda), ctx.vec(f.n_lambda), ctx.vec(f.n_lambda)
    qi.F.mult(lv, y0)
    qe.F.mult(lv, y1)
    qc.F.mult(lv, y2)
    n0 = np.linalg.norm(y0.to_numpy())
    assert np.linalg.norm(y2.to_numpy() - y0.to_numpy()) <= 1e-9 * n0 and np.linalg.norm(y2.to_numpy() - y1.to_numpy()) <= 1e-9 * n0
    qi.lam.set(0.0)
    qc.lam.set(0.0)
    si, sc = qi.solve_smalxe(rtol=1e-5), qc.solve_smalxe(rtol=1e-5)
    assert (si.reason, si.iteration, si.inner_iter_accu, si.inner.ncg, si.inner.nexp) == (sc.reason, sc.iteration, sc.inner_iter_accu, sc.inner.ncg, sc.inner.nexp)


def test_general_F_equals_inner_krylov_F(general):
    ctx, f, loc, qi, qe, _qc = general
    rng = np.random.default_rng(11)
    for _ in range(2):
        lam = rng.standard_normal(f.n_lambda)
        lv, y0, y1 = ctx.vec_from(lam), ctx.vec(f.n_lambda), ctx.vec(f.n_lambda)
        qi.F.mult(lv, y0)
        qe.F.mult(lv, y1)
        a, b = y0.to_numpy(), y1.to_numpy()
        assert np.linalg.norm(a - b) <= 1e-9 * np.linalg.norm(a)


def test_general_blocks_equal_direct_solves_and_scale_with_the_material(general):
    """Columns of W_b = (K_b^+)[Gamma_b, Gamma_b] against K^+ e_j by the iterative solver; and W_b = W_1-like / E_b: blocks 0 and 2 touch the same dofs
    (same position in the decomposition up to the y mirror), so E_0 W_0 and E_2 W_2 agree on their common shape only through the solver -- what IS exact is
    the scaling K_s = E_s K_1 => W_s(E) = W_s(1) / E, checked on the diagonal entries of two blocks with equal n_Gamma."""
    ctx, f, loc, qi, qe, _qc = general
    E = qe.E
    rs = np.asarray(loc["block_rowstart"])
    rng = np.random.default_rng(5)
    for b in (0, 5):
        n_g = int(E.n_gamma[b])
        W, gam = E.block(b)  # Gamma_b as rank-local primal indices
        gam = gam - rs[b]
        assert np.allclose(W, W.T, rtol=0, atol=1e-10 * np.abs(W).max())
        for j in rng.choice(n_g, size=3, replace=False):
            rhs = np.zeros(f.N)
            rhs[rs[b] + gam[j]] = 1.0
            u = ctx.vec(f.N)
            qi.Kplus.mult(ctx.vec_from(rhs), u)
            col = u.to_numpy()[rs[b] + gam]
            assert np.linalg.norm(W[:, j] - col) <= 1e-8 * np.linalg.norm(col)


def test_general_contact_solve_same_counts_as_inner_krylov(general):
    ctx, f, loc, qi, qe, _qc = general
    res = []
    for q in (qi, qe):
        q.lam.set(0.0)
        st = q.solve_smalxe(rtol=1e-5)
        res.append((st, q.dual_solution()))
    (si, li), (se, le) = res
    assert si.reason == 2 and se.reason == 2
    assert (si.iteration, si.inner_iter_accu, si.inner.nmv, si.inner.ncg, si.inner.nexp, si.inner.nprop) == (se.iteration, se.inner_iter_accu, se.inner.nmv, se.inner.ncg, se.inner.nexp, se.inner.nprop)
    assert np.linalg.norm(li - le) <= 1e-6 * np.linalg.norm(li)
    # feasibility and complementarity of the contact multipliers (dual box: lambda_I >= 0)
    lamI = le[f.n_eq:]
    assert lamI.min() >= -1e-12 and (lamI > 0).sum() > 0


def test_block_without_any_symmetry_against_pinv():
    """A decomposition NOTHING leans on: two subdomains, the second one of a GRADED material (the modulus varies from element to element: E = 2 (1 + 3 x + y^2 + 0.5 x z)),
    so that no two blocks are congruent and no coordinate permutation of the cube leaves K_2 invariant.  The explicit operators are assembled column by column through the
    multi-right-hand-side K^+ (8 columns per block and application, matinv_mv.hip; the reference's column-blocked MatInvExplicitly_Inv, matinv.c:640-730) and every W_b is
    compared with numpy.linalg.pinv(K_b) on Gamma_b; the same assembly one column at a time gives the same operators."""
    import scipy.sparse as sp

    ctx = pa.Context(0)
    nel = 6
    f = pa.CubeFeti((2, 1, 1), nel, contact=True, young=[1.0, 2.0], graded={1: lambda x, y, z: 1.0 + 3.0 * x + y * y + 0.5 * x * z})
    assert not f.congruent
    nn = nel + 1
    K1 = f.block_K(1)
    from permon_amd.feti import box_symmetries

    ops = box_symmetries((nn, nn, nn), 3, K1)
    assert len(ops[0]) == 1  # only the identity survives the check against the graded matrix
    G, e = f.coarse(orthonormalize=True)
    loc = f.subset(range(2))
    box = dict(dims=[(nn, nn, nn)] * 2, ndof=3, min_nodes=27)
    W = {}
    for mv in (True, False):
        q = FetiDualQP(ctx, loc, G, e, f.c, f.lb, kplus_rtol=1e-13, mg_box=box, mg_precision="fp32", bsr3=True, explicit=dict(rtol=1e-13, storage="sym", multi_rhs=mv))
        assert q.explicit_storage == "sym" and q.explicit_multi_rhs == mv
        assert q.E.assemble_stats()[0] == int(q.E.n_gamma.sum())
        W[mv] = [q.E.block(b) for b in range(2)]
    for b in range(2):
        Kb = f.block_K(b).toarray()
        Kp = np.linalg.pinv(Kb, rcond=1e-10, hermitian=True)
        Wb, g = W[True][b]
        g = g - int(f.block_rowstart[b])  # Gamma_b is given in the rank's primal numbering
        ref = Kp[np.ix_(g, g)]
        assert np.max(np.abs(Wb - ref)) <= 1e-9 * np.max(np.abs(ref)), b
        assert np.max(np.abs(Wb - W[False][b][0])) <= 1e-10 * np.max(np.abs(ref)), b
    ctx.close()
