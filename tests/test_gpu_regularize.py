"""GPU tests of the reference's default K^+ path: MatRegularize (permonmatregularize.c) feeding MATINV (matinv.c:449-459),
against the CPU oracle and against the Moore-Penrose path the other FETI tests use."""
import numpy as np
import pytest
import scipy.sparse as sp

import permon_amd as pa
from permon_amd.chain import FetiDualQP, regularize_blocks
from permon_amd.feti import CubeFeti, box_mg_hierarchy

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = pa.Context(0)
    yield c
    c.close()


def _block(f):
    K = f.Ki.tocsr()
    K.sort_indices()
    return K, np.ascontiguousarray(f.R[:, :f.n_i])


@pytest.mark.parametrize("physics", ["poisson", "elasticity"])
def test_power_method_restart_and_regularize_vs_oracle(ctx, oracle, physics):
    f = CubeFeti((2, 1, 1), 3, physics, contact=False)
    K, R = _block(f)
    p = K.shape[0]
    Kd = pa.CsrMat(ctx, p, p, K.indptr, K.indices, K.data)
    # MatGetMaxEigenvalue(K_loc, NULL, &rho, 1, 20): v = 1 is a kernel vector of the floating block -> RAND48 restart branch
    lam, its = pa.Op.from_csr(Kd).max_eigenvalue(tol=1.0, maxits=20)
    lam_o, its_o = oracle.max_eigenvalue(oracle.Op(p, csr=oracle.Csr.from_scipy(K)), tol=1.0, maxits=20)
    # the first Rayleigh quotient is rounding noise around 0 (+-1e-17): its sign decides whether the loop stops after 2 or 3
    # iterations, on the CPU as on the GPU; with the same count the values agree to the reduction order of the dots
    assert its in (2, 3) and its_o in (2, 3) and lam > 0
    if its == its_o:
        assert abs(lam - lam_o) <= 1e-10 * abs(lam_o)
    lmax = np.linalg.eigvalsh(K.toarray()).max()
    assert 0.05 * lmax < lam <= lmax * (1 + 1e-12)
    Kreg, piv, rho = pa.MatRegularize(ctx, K, R)
    rp, ci, va, piv_o = oracle.regularize_csr(oracle.Csr.from_scipy(K), R, rho)
    assert piv.tolist() == piv_o.tolist()  # index bookkeeping: exact
    assert np.array_equal(Kreg.indptr, rp) and np.array_equal(Kreg.indices, ci)
    assert np.abs(Kreg.data - va).max() <= 1e-14 * np.abs(va).max()
    # with the oracle's rho handed over the values agree to rounding
    Kreg2, _, _ = pa.MatRegularize(ctx, K, R, rho=lam_o)
    assert np.abs(Kreg2.data - va).max() <= 1e-14 * np.abs(va).max()


def test_kplus_on_regularized_blocks(ctx, oracle):
    """K^+ = K_reg^{-1} block-wise (no null-space projection): against the dense inverse, against the oracle's block CG on the
    same K_reg, and as a generalised inverse of K (K K^+ f = f for f in range(K))."""
    f = CubeFeti((2, 2, 1), 2, contact=False)
    loc = f.subset(range(f.nsub))
    Kreg, pivots, rhos = regularize_blocks(ctx, loc)
    assert len(pivots) == f.nsub and all(len(pv) == 6 for pv in pivots)
    Kb = pa.MatBlockDiag.from_scipy(ctx, loc["block_rowstart"], Kreg)
    Kplus = pa.MatInv(Kb, rtol=1e-13, nullspace=None)
    rng = np.random.default_rng(12)
    rhs = rng.standard_normal(f.N)
    u = ctx.vec(f.N)
    Kplus.mult(ctx.vec_from(rhs), u)
    ref = np.linalg.solve(Kreg.toarray(), rhs)
    assert np.linalg.norm(u.to_numpy() - ref) <= 1e-9 * np.linalg.norm(ref)
    Mo = oracle.MatInv(oracle.Csr.from_scipy(Kreg), loc["block_rowstart"], None, rtol=1e-13)
    assert np.linalg.norm(u.to_numpy() - Mo.mult(rhs)) <= 1e-9 * np.linalg.norm(ref)
    g = f.K @ rng.standard_normal(f.N)  # in range(K)
    Kplus.mult(ctx.vec_from(g), u)
    assert np.linalg.norm(f.K @ u.to_numpy() - g) <= 1e-9 * np.linalg.norm(g)


def test_contact_tfeti_regularized_equals_moore_penrose_path(ctx, oracle):
    """On the projected dual problem the choice of the generalised inverse is invisible (P kills the B R alpha components):
    SMALXE+MPGP on F_reg = B K_reg^{-1} B' (the reference's default, -regularize 1) and on F_mp = B K^+_mp B' take the same
    iterations and reach the same lambda; the oracle runs the same chain with a dense K_reg^{-1}."""
    f = CubeFeti((2, 2, 2), 2, contact=True)
    G, e = f.coarse(orthonormalize=True)
    loc = f.subset(range(f.nsub))
    Kb, _ = _block(f)
    # the reference's rho (power method, tol 1, <= 20 its) stops after 2 or 3 iterations depending on the sign of a rounding-noise
    # Rayleigh quotient: both sides get the oracle's value so that they regularise with the same matrix
    rho, _ = oracle.max_eigenvalue(oracle.Op(Kb.shape[0], csr=oracle.Csr.from_scipy(Kb)), tol=1.0, maxits=20)
    loc_reg = dict(loc)
    loc_reg["Kreg"] = regularize_blocks(ctx, loc, rho=rho)[0]
    q_reg = FetiDualQP(ctx, loc_reg, G, e, f.c, f.lb, orthonormal=True, kplus_rtol=1e-13, regularize=True)
    q_mp = FetiDualQP(ctx, dict(loc), G, e, f.c, f.lb, orthonormal=True, kplus_rtol=1e-13, regularize=False)
    s_reg, s_mp = q_reg.solve_smalxe(), q_mp.solve_smalxe()
    assert s_reg.reason == s_mp.reason == 2
    assert s_reg.iteration == s_mp.iteration
    assert abs(s_reg.inner_iter_accu - s_mp.inner_iter_accu) <= max(2, s_mp.inner_iter_accu // 50)
    l_reg, l_mp = q_reg.dual_solution(), q_mp.dual_solution()
    assert np.linalg.norm(l_reg - l_mp) <= 1e-4 * np.linalg.norm(l_mp)
    # oracle: dense chain on K_reg built by the oracle's own MatRegularize (rho from its own power method)
    inv_blocks = []
    for s_ in range(f.nsub):  # the kernel bases differ from block to block (rotations about the global origin): so may the pivots
        Rb = np.ascontiguousarray(f.R[:, s_ * f.n_i:(s_ + 1) * f.n_i])
        rp, ci, va, _ = oracle.regularize_csr(oracle.Csr.from_scipy(Kb), Rb, rho)
        inv_blocks.append(np.linalg.inv(sp.csr_matrix((va, ci, rp), shape=Kb.shape).toarray()))
    Kri = sp.block_diag(inv_blocks).toarray()
    Bd = f.B.toarray()
    Fd = Bd @ Kri @ Bd.T
    pfo = oracle.Qppf(oracle.Csr.from_scipy(G), orthonormal=True)
    d = Bd @ (Kri @ f.f) - f.c
    lam_t = pfo.half_Q_transpose(e)
    b_bar = d - Fd @ lam_t
    n = f.n_lambda
    ref = oracle.smalxe(oracle.Op(n, fn=lambda x: pfo.P(Fd @ pfo.P(x))), pfo.P(b_bar), np.zeros(n), oracle.Box(n, lb=f.lb - lam_t), pfo)
    assert np.linalg.norm(q_reg.d.to_numpy() - d) <= 1e-9 * np.linalg.norm(d)
    assert (s_reg.reason, s_reg.iteration) == (ref["reason"], ref["iteration"])
    assert abs(s_reg.inner_iter_accu - ref["inner_iter_accu"]) <= max(2, ref["inner_iter_accu"] // 50)
    assert np.linalg.norm(q_reg.lam.to_numpy() - ref["u"]) <= 1e-4 * np.linalg.norm(ref["u"])


@pytest.mark.parametrize("precision", ["fp64", "fp16"])
def test_multigrid_pc_on_regularized_blocks(ctx, precision):
    """The V-cycle hierarchy built on K_reg (Galerkin, coarse level now regular): few CG iterations, same K^+ f as Jacobi-CG."""
    f = CubeFeti((2, 1, 1), 8, contact=False)
    loc = f.subset(range(f.nsub))
    Kreg, pivots, _ = regularize_blocks(ctx, loc)
    nn, n_i = f.nel + 1, f.n_i
    Kr_i = Kreg[:n_i, :n_i].tocsr()
    H = box_mg_hierarchy([Kr_i] * f.nsub, [(nn, nn, nn)] * f.nsub, 3, min_nodes=27)
    rhs = np.random.default_rng(3).standard_normal(f.N)
    out = []
    for mg in (False, True):
        Kb = pa.MatBlockDiag.from_scipy(ctx, loc["block_rowstart"], Kreg)
        Kplus = pa.MatInv(Kb, rtol=1e-11, nullspace=None)
        Kplus.enable_bsr3()
        if mg:
            Kplus.set_pc_mg(H, degree=2, precision=precision)
        u = ctx.vec(f.N)
        Kplus.mult(ctx.vec_from(rhs), u)
        out.append((u.to_numpy(), Kplus.last_iterations()[0]))
    (u_j, it_j), (u_m, it_m) = out
    assert it_m <= 30 and it_m < it_j / 4
    assert np.linalg.norm(u_m - u_j) <= 1e-8 * np.linalg.norm(u_j)
