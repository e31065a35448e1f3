"""GPU parity tests of the FETI side of the path (MATGLUING, QPPF, MATINV, F, SMALXE, PCPG) against the CPU oracle."""
import numpy as np
import pytest
import scipy.sparse as sp

import permon_amd as pa
from permon_amd import problems as P
from permon_amd.chain import FetiDualQP

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = pa.Context(0)
    yield c
    c.close()


def test_gluing_bit_exact(ctx, oracle):
    rng = np.random.default_rng(11)
    n_x, n_l, nleaf = 5000, 1300, 4000
    rows = rng.integers(0, n_x, nleaf).astype(np.int32)
    roots = rng.integers(0, n_l, nleaf).astype(np.int32)
    signs = rng.choice([-1.0, 1.0, 1 / np.sqrt(2), -1 / np.sqrt(3)], nleaf)
    Bd = pa.MatGluing(ctx, n_x, n_l, rows, roots, signs)
    Bo = oracle.Gluing(n_x, n_l, rows, roots, signs)
    lam, x = rng.standard_normal(n_l), rng.standard_normal(n_x)
    xd, ld = ctx.vec(n_x), ctx.vec(n_l)
    Bd.mult(ctx.vec_from(lam), xd)
    assert np.array_equal(xd.to_numpy(), Bo.mult(lam))
    Bd.mult_transpose(ctx.vec_from(x), ld)
    assert np.array_equal(ld.to_numpy(), Bo.mult_transpose(x))


@pytest.mark.parametrize("orth", [False, True])
def test_qppf_vs_oracle(ctx, oracle, orth):
    f = pa.CubeFeti((2, 2, 1), 2)
    G, e = f.coarse(orthonormalize=orth)
    pf = pa.QPPF.from_scipy(ctx, G, orthonormal=orth)
    pfo = oracle.Qppf(oracle.Csr.from_scipy(G), orthonormal=orth)
    rng = np.random.default_rng(3)
    v = rng.standard_normal(f.n_lambda)
    vd, yd = ctx.vec_from(v), ctx.vec(f.n_lambda)
    pf.ApplyQ(vd, yd)
    q_ref = pfo.Q(v)
    assert np.max(np.abs(yd.to_numpy() - q_ref)) <= 1e-12 * np.max(np.abs(q_ref))
    pf.ApplyP(vd, yd)
    assert np.max(np.abs(yd.to_numpy() - pfo.P(v))) <= 1e-12 * np.max(np.abs(v))
    # P is a projector onto null(G)
    gd = ctx.vec(G.shape[0])
    pf.ApplyG(yd, gd)
    assert np.max(np.abs(gd.to_numpy())) <= 1e-12
    if not orth:
        # coarse solve: GG' assembled by the fp64-MFMA kernel (k_ggt_mfma), inverted on the host, applied as a dense GEMV
        xm = rng.standard_normal(G.shape[0])
        ym = ctx.vec(G.shape[0])
        pf.ApplyCP(ctx.vec_from(xm), ym)
        ref_cp = np.linalg.solve((G @ G.T).toarray(), xm)
        assert np.linalg.norm(ym.to_numpy() - ref_cp) <= 1e-10 * np.linalg.norm(ref_cp)
    ed = ctx.vec_from(e)
    pf.ApplyHalfQTranspose(ed, yd)
    assert np.max(np.abs(yd.to_numpy() - pfo.half_Q_transpose(e))) <= 1e-12 * max(1.0, np.max(np.abs(e)))


@pytest.mark.parametrize("size", [(2, 2, 1, 2), (2, 2, 2, 16)])
def test_qppf_implicit_orthonormalisation(ctx, size, monkeypatch):
    """pmh_qppf_create(orthonormal = 2): G0 = R'B' kept sparse, T = chol(G0 G0')^{-1} applied in the finishing launch of G0 v -- every slot acts as the
    explicitly orthonormalised T G0 does (QPTOrthonormalizeEq, -qp_E_orth_form implicit vs explicit); the second size has long rows (the chunked
    G v kernels) and short G0' rows (k_gt_fused1 of the penalised operator)."""
    f = pa.CubeFeti(size[:3], size[3], contact=True)
    G0, e0 = f.coarse(orthonormalize=False)
    Ge, ee = f.coarse(orthonormalize=True)
    assert Ge.nnz > 1.5 * G0.nnz  # what the implicit form saves
    pi, pe = pa.QPPF.from_scipy(ctx, G0, orthonormal="implicit"), pa.QPPF.from_scipy(ctx, Ge, orthonormal=True)
    assert np.linalg.norm(pi.orth_rhs(e0) - ee) <= 1e-11 * np.linalg.norm(ee)
    rng = np.random.default_rng(8)
    v, w = rng.standard_normal(f.n_lambda), rng.standard_normal(G0.shape[0])
    vd, wd = ctx.vec_from(v), ctx.vec_from(w)
    for name, arg, n_out in (("ApplyQ", vd, f.n_lambda), ("ApplyP", vd, f.n_lambda), ("ApplyGtG", vd, f.n_lambda), ("ApplyG", vd, G0.shape[0]), ("ApplyHalfQ", vd, G0.shape[0]),
                             ("ApplyHalfQTranspose", wd, f.n_lambda), ("ApplyCP", wd, G0.shape[0])):
        ya, yb = ctx.vec(n_out), ctx.vec(n_out)
        getattr(pi, name)(arg, ya)
        getattr(pe, name)(arg, yb)
        ref = yb.to_numpy()
        assert np.linalg.norm(ya.to_numpy() - ref) <= 1e-11 * max(np.linalg.norm(ref), 1e-3 * np.linalg.norm(v)), name
    # the penalised projected operator rho Q x + P A P x (its fused G' epilogues against the unfused sequence: same bits; against the explicit form)
    n = f.n_lambda
    Dm = sp.diags(1.0 + rng.random(n)).tocsr()
    D = pa.Op.from_csr(pa.CsrMat(ctx, n, n, Dm.indptr, Dm.indices, Dm.data))
    out = []
    pd = pa.QPPF.from_scipy(ctx, G0)  # G as it is, dense (G G')^{-1}: the fused form is k_gt_dual1 + k_gt_fused1 (its penalty term is rho G'G, not rho Q)
    for pf in (pi, pe, pd):
        Ap = pa.MatCreatePenalized(pa.MatCreateProjected(D, pf, symmetric=True), pf, 2.5)
        y1, y2 = ctx.vec(n), ctx.vec(n)
        pa._lib.check(ctx.L.pmh_op_mult(Ap.h, vd.p, y1.p))
        ctx.L.pmh_set_knob(b"gt_fusion", 0)
        pa._lib.check(ctx.L.pmh_op_mult(Ap.h, vd.p, y2.p))
        ctx.L.pmh_set_knob(b"gt_fusion", 1)
        assert np.array_equal(y1.to_numpy(), y2.to_numpy())
        out.append(y1.to_numpy())
    assert np.linalg.norm(out[0] - out[1]) <= 1e-11 * np.linalg.norm(out[1])
    G0d = G0.toarray()
    proj = lambda z: z - G0d.T @ np.linalg.solve(G0d @ G0d.T, G0d @ z)
    ref = 2.5 * (G0d.T @ (G0d @ v)) + proj(Dm @ proj(v))
    assert np.linalg.norm(out[2] - ref) <= 1e-10 * np.linalg.norm(ref)
    if size[3] > 4:
        return
    # the whole dual chain and the contact solve: same counts, same solution
    loc, res = f.subset(range(f.nsub)), []
    for G, e, orth in ((G0, e0, "implicit"), (Ge, ee, True)):
        q = FetiDualQP(ctx, loc, G, e, f.c, f.lb, orthonormal=orth, kplus_rtol=1e-12)
        st = q.solve_smalxe(rtol=1e-6)
        res.append((st, q.dual_solution()))
    (sa, la), (sb, lb_) = res
    assert (sa.iteration, sa.inner_iter_accu, sa.reason, sa.inner.ncg, sa.inner.nexp, sa.inner.nprop) == (sb.iteration, sb.inner_iter_accu, sb.reason, sb.inner.ncg, sb.inner.nexp, sb.inner.nprop)
    assert np.linalg.norm(la - lb_) <= 1e-8 * np.linalg.norm(la)


def _dense_ops(f):
    Kd = f.K.toarray()
    Kp = np.linalg.pinv(Kd, rcond=1e-12, hermitian=True)
    Bd = f.B.toarray()
    return Kp, Bd, Bd @ Kp @ Bd.T


@pytest.mark.parametrize("physics", ["poisson", "elasticity"])
def test_matinv_block_cg_is_pseudoinverse(ctx, physics):
    f = pa.CubeFeti((2, 1, 1), 2, physics=physics)
    Kp, _, _ = _dense_ops(f)
    K = pa.MatBlockDiag.from_scipy(ctx, f.block_rowstart, f.K)
    Kplus = pa.MatInv(K, rtol=1e-13, nullspace=f.R)
    rng = np.random.default_rng(5)
    rhs = rng.standard_normal(f.N)
    u = ctx.vec(f.N)
    Kplus.mult(ctx.vec_from(rhs), u)
    ref = Kp @ rhs
    assert np.linalg.norm(u.to_numpy() - ref) <= 1e-9 * np.linalg.norm(ref)
    its, total = Kplus.last_iterations()
    assert 0 < its < 500 and total >= its
    y = ctx.vec(f.N)
    K.mult(u, y)  # MatMult_BlockDiag
    assert np.linalg.norm(y.to_numpy() - f.K @ u.to_numpy()) <= 1e-12 * np.linalg.norm(rhs)


def test_matinv_load_in_the_kernel(ctx):
    """A block whose whole load lies in the kernel of K (a uniform body force on an interior floating subdomain: feti/ex71.c's slabs): P_R f is rounding residue, which is NOT
    in the range of the singular K -- the block CG must not iterate on it (it did until round 4: 1e-4 absolute error in K^+ f, 29 instead of 27 dual iterations on ex71 TEST 2
    with the Moore-Penrose K^+).  K^+ f of such a block is exactly 0; the other block is untouched by the floor."""
    f = pa.CubeFeti((2, 1, 1), 3, physics="elasticity")
    Kp, _, _ = _dense_ops(f)
    K = pa.MatBlockDiag.from_scipy(ctx, f.block_rowstart, f.K)
    R = np.asarray(f.R)
    rs = np.asarray(f.block_rowstart)
    rng = np.random.default_rng(2)
    g = rng.standard_normal(f.N)
    g[rs[1]:rs[2]] = (R.T @ np.array([3.0, -2.0, 5.0, 0.7, -1.1, 0.3]))[rs[1]:rs[2]]  # block 1: a rigid-body load (a row of R holds one kernel vector of every block)
    for rtol in (1e-10, 1e-14):
        Kplus = pa.MatInv(K, rtol=rtol, max_it=20000, nullspace=f.R)
        u = ctx.vec(f.N)
        Kplus.mult(ctx.vec_from(g), u)
        u = u.to_numpy()
        ref = Kp @ g
        assert np.all(u[rs[1]:rs[2]] == 0.0) and np.linalg.norm(ref[rs[1]:rs[2]]) <= 1e-12 * np.linalg.norm(g)
        assert np.linalg.norm(u - ref) <= max(1e2 * rtol, 1e-12) * np.linalg.norm(ref)
        its, _ = Kplus.last_iterations()
        assert 0 < its < 2000


def test_matinv_left_generalised_inverse(ctx):
    """-qpt_dualize_Kplus_left (qptransform.c:997-1062): K^+ = K^- P_R with K^- the solve that leaves the null-pivot dofs at zero.  Against dense numpy: K^- = the inverse of
    K without the fixing dofs, zero-padded; K K^+ g = P_R g away from the fixing dofs; the result is NOT orthogonal to the kernel (the Moore-Penrose form's is)."""
    import scipy.sparse as sp

    f = pa.CubeFeti((2, 1, 1), 2, physics="poisson")  # two floating blocks, kernel = constants
    Kd, N, rs = f.K.toarray(), f.N, np.asarray(f.block_rowstart)
    fix = [int(rs[0]), int(rs[1])]  # the first dof of every block
    Kfix = Kd.copy()
    Kfix[fix, :] = 0.0
    Kfix[:, fix] = 0.0
    Kfix[fix, fix] = 1.0
    K = pa.MatBlockDiag.from_scipy(ctx, f.block_rowstart, sp.csr_matrix(Kfix))
    Kplus = pa.MatInv(K, rtol=1e-13, nullspace=f.R)
    Kplus.set_left_inverse(fix)
    g = np.random.default_rng(11).standard_normal(N)
    u = ctx.vec(N)
    Kplus.mult(ctx.vec_from(g), u)
    u = u.to_numpy()
    R = np.asarray(f.R)
    Pg = g.copy()
    for b in range(len(rs) - 1):  # block-wise: a row of R holds one kernel vector of EVERY block
        Rb = R[:, rs[b]:rs[b + 1]]
        Pg[rs[b]:rs[b + 1]] -= Rb.T @ (Rb @ g[rs[b]:rs[b + 1]])
    keep = np.setdiff1d(np.arange(N), fix)
    ref = np.zeros(N)
    ref[keep] = np.linalg.solve(Kd[np.ix_(keep, keep)], Pg[keep])
    assert np.linalg.norm(u - ref) <= 1e-9 * np.linalg.norm(ref) and np.all(u[fix] == 0.0)
    assert np.linalg.norm((Kd @ u - Pg)[keep]) <= 1e-9 * np.linalg.norm(Pg)  # the kept equations hold; the dropped ones follow from R'P_R g = 0
    assert np.linalg.norm(Kd @ u - Pg) <= 1e-8 * np.linalg.norm(Pg)
    kern = lambda w: np.sqrt(sum(np.linalg.norm(R[:, rs[b]:rs[b + 1]] @ w[rs[b]:rs[b + 1]]) ** 2 for b in range(len(rs) - 1)))  # noqa: E731
    assert kern(u) > 1e-3 * np.linalg.norm(u)  # a kernel component stays: this is K^- P_R, not P_R K^- P_R
    Kplus.set_left_inverse([])
    v = ctx.vec(N)
    Kplus.mult(ctx.vec_from(g), v)
    assert kern(v.to_numpy()) <= 1e-10 * np.linalg.norm(v.to_numpy())


def test_feti_dual_operator_and_lumped_pc(ctx):
    f = pa.CubeFeti((2, 2, 1), 2)
    Kp, Bd, Fd = _dense_ops(f)
    loc = f.subset(range(f.nsub))
    K = pa.MatBlockDiag.from_scipy(ctx, loc["block_rowstart"], loc["K"])
    Kplus = pa.MatInv(K, rtol=1e-13, nullspace=loc["R"])
    B = pa.MatGluing(ctx, loc["n_x"], loc["n_lambda"], loc["leaves_row"], loc["leaves_root"], loc["leaves_sign"])
    F = pa.MatCreateFetiDual(B, Kplus)
    rng = np.random.default_rng(9)
    lam = rng.standard_normal(f.n_lambda)
    y = ctx.vec(f.n_lambda)
    F.mult(ctx.vec_from(lam), y)
    ref = Fd @ lam
    assert np.linalg.norm(y.to_numpy() - ref) <= 1e-9 * np.linalg.norm(ref)
    pc = pa.PCDualLumpedOp(B, K)
    pc.mult(ctx.vec_from(lam), y)
    ref = Bd @ (f.K @ (Bd.T @ lam))
    assert np.linalg.norm(y.to_numpy() - ref) <= 1e-12 * np.linalg.norm(ref)


def _oracle_dual(oracle, f, G, e, orth):
    """The same chain on the CPU with a dense Moore-Penrose K^+ (oracle side of the parity test)."""
    Kp, Bd, Fd = _dense_ops(f)
    pfo = oracle.Qppf(oracle.Csr.from_scipy(G), orthonormal=orth)
    d = Bd @ (Kp @ f.f) - f.c
    lam_t = pfo.half_Q_transpose(e)
    b_bar = d - Fd @ lam_t
    lb_new = f.lb - lam_t
    return Fd, pfo, d, lam_t, b_bar, lb_new


def test_smalxe_contact_tfeti_vs_oracle(ctx, oracle):
    f = pa.CubeFeti((2, 2, 2), 2, contact=True)
    G, e = f.coarse(orthonormalize=True)
    Fd, pfo, d, lam_t, b_bar, lb_new = _oracle_dual(oracle, f, G, e, True)
    n = f.n_lambda
    A_or = oracle.Op(n, fn=lambda x: pfo.P(Fd @ pfo.P(x)))
    ref = oracle.smalxe(A_or, pfo.P(b_bar), np.zeros(n), oracle.Box(n, lb=lb_new), pfo)
    assert ref["reason"] > 0

    q = FetiDualQP(ctx, f.subset(range(f.nsub)), G, e, f.c, f.lb, orthonormal=True, kplus_rtol=1e-13)
    assert np.linalg.norm(q.d.to_numpy() - d) <= 1e-9 * np.linalg.norm(d)
    assert np.linalg.norm(q.b.to_numpy() - pfo.P(b_bar)) <= 1e-9 * np.linalg.norm(b_bar)
    st = q.solve_smalxe()
    assert st.reason == ref["reason"]
    assert st.iteration == ref["iteration"]
    assert abs(st.inner_iter_accu - ref["inner_iter_accu"]) <= max(2, ref["inner_iter_accu"] // 50)
    lam_child = q.lam.to_numpy()
    assert np.linalg.norm(lam_child - ref["u"]) <= 1e-4 * np.linalg.norm(ref["u"])
    # KKT of the dual problem: feasibility of the multipliers and of the primal solution
    lam = q.dual_solution()
    assert np.all(lam[f.n_eq:] >= -1e-10)
    # primal recovery (QPTDualizePostSolve): u = K^+(f - B'lambda) + R alpha with alpha fitted on the rows that
    # must be tight (equalities and active contact rows): B u - c = (d - F lambda) + (B R) alpha
    u, Fl_minus_d = q.primal_solution(G)
    Ru = f.kernel_matrix()
    tight = (np.arange(n) < f.n_eq) | (lam > 1e-8)
    BR = (f.B @ Ru).toarray()
    alpha = np.linalg.lstsq(BR[tight], Fl_minus_d[tight], rcond=None)[0]
    uu = u + Ru @ alpha
    scale = np.max(np.abs(uu))
    assert np.max(np.abs((f.B @ uu)[:f.n_eq] - f.c[:f.n_eq])) <= 1e-3 * scale
    assert np.max((f.B @ uu)[f.n_eq:] - f.c[f.n_eq:]) <= 1e-3 * scale
    # equilibrium: K u = f - B' lambda
    # (f - B'lambda has a kernel component of the size of the solver tolerance: G lambda = e holds to rtol)
    assert np.linalg.norm(f.K @ uu - (f.f - f.B.T @ lam)) <= 1e-4 * np.linalg.norm(f.f)


def test_pcpg_linear_tfeti_vs_oracle(ctx, oracle):
    f = pa.CubeFeti((2, 2, 1), 2, contact=False)
    G, e = f.coarse(orthonormalize=False)
    Fd, pfo, d, lam_t, b_bar, lb_new = _oracle_dual(oracle, f, G, e, False)
    n = f.n_lambda
    ref = oracle.pcpg(oracle.Op(n, fn=lambda x: Fd @ x), b_bar, np.zeros(n), pfo, rtol=1e-8)
    q = FetiDualQP(ctx, f.subset(range(f.nsub)), G, e, f.c, f.lb, orthonormal=False, kplus_rtol=1e-13)
    st = q.solve_pcpg(rtol=1e-8)
    assert (st.reason, st.iteration) == (ref["reason"], ref["iteration"])
    assert np.linalg.norm(q.lam.to_numpy() - ref["x"]) <= 1e-6 * np.linalg.norm(ref["x"])
    # lumped preconditioner: fewer iterations, same solution (feti/output/ex71_2_*: 66 -> 26 its)
    Bd = f.B.toarray()
    refl = oracle.pcpg(oracle.Op(n, fn=lambda x: Fd @ x), b_bar, np.zeros(n), pfo, rtol=1e-8, pc=lambda w: Bd @ (f.K @ (Bd.T @ w)))
    q2 = FetiDualQP(ctx, f.subset(range(f.nsub)), G, e, f.c, f.lb, orthonormal=False, kplus_rtol=1e-13)
    st2 = q2.solve_pcpg(rtol=1e-8, lumped=True)
    assert (st2.reason, st2.iteration) == (refl["reason"], refl["iteration"])
    assert st2.iteration < st.iteration
    assert np.linalg.norm(q2.lam.to_numpy() - refl["x"]) <= 1e-6 * np.linalg.norm(refl["x"])


def _ex3_dual_dense(n):
    p = P.ex3_primal(n)
    K = sp.csr_matrix((p["val"], p["col"], p["rowptr"]), shape=(n, n)).toarray()
    Lc = np.linalg.cholesky(K)
    Kinv = np.linalg.solve(Lc.T, np.linalg.solve(Lc, np.eye(n)))
    B = np.diag(p["BI_diag"])
    return B @ Kinv @ B.T, B @ (Kinv @ p["b"]) - p["cI"]


def _dense_csr(ctx, F):
    Fs = sp.csr_matrix(F)
    Fs.sort_indices()
    return pa.CsrMat(ctx, F.shape[0], F.shape[1], Fs.indptr, Fs.indices, Fs.data)


def test_ex3_goldens_on_gpu(ctx, goldens):
    """Reference goldens ex3_1.out (dualised MPGP) and ex3_nullspace.out (SMALXE with a 0-row BE) through the C ABI."""
    n = 100
    F, d = _ex3_dual_dense(n)
    op = pa.Op.from_csr(_dense_csr(ctx, F))
    g = goldens["ex3_1"]["solves"][0]
    qp = pa.QP(ctx)
    qp.SetOperator(op)
    qp.SetRhs(ctx.vec_from(d))
    qp.SetInitialVector(ctx.vec(n))
    qp.SetBox(None, ctx.vec(n), None)
    qps = pa.QPS(ctx)
    qps.SetQP(qp)
    qps.SetType("mpgp")
    st = qps.Solve()
    assert (st.iteration, st.nmv, st.ncg, st.nexp, st.nprop, st.reason) == (g["iterations"], g["nmv"], g["ncg"], g["nexp"], g["nprop"], g["reason"])
    # -qp_chain_view_kkt of ex3_1.out: four lines of the dual (box) QP, then four of the primal QP after QPTDualizePostSolve
    kk = goldens["ex3_1"]["kkt"]
    assert len(kk) == 8

    def same(val, printed):
        return ("%.2e" % val) == printed or (float(printed) < 1e-12 and abs(val) < 1e-12)

    import re

    for line, ref in zip(qps.ViewKKT(), kk[:4]):
        m = re.match(r"r = (.*?)\s*= (\S+)\s+rO?/\|\|b\|\| = (\S+)", line)
        assert line.startswith("r = " + ref["name"]) and same(float(m.group(2)), ref["r"]) and same(float(m.group(3)), ref["r_rel"]), (line, ref)
    p3 = P.ex3_primal(n)
    Kd = sp.csr_matrix((p3["val"], p3["col"], p3["rowptr"]), shape=(n, n)).toarray()
    BI, lam = np.diag(p3["BI_diag"]), qp.x.to_numpy()
    xp = np.linalg.solve(Kd, p3["b"] - BI.T @ lam)  # u = K^+(f - B' lambda), qptransform.c:812-815 (K regular here)
    nb = np.linalg.norm(p3["b"])
    gap = BI @ xp - p3["cI"]
    vals = [np.linalg.norm(Kd @ xp - p3["b"] + BI.T @ lam), np.linalg.norm(np.maximum(gap, 0.0)), np.linalg.norm(np.minimum(lam, 0.0)), abs(lam @ gap)]
    assert vals[0] <= 1e-13 and float(kk[4]["r"]) <= 1e-13  # rounding level on both sides
    for v, ref in zip(vals[1:], kk[5:]):
        assert same(v, ref["r"]) and same(v / nb, ref["r_rel"]), (v, v / nb, ref)

    outer, inner = goldens["ex3_nullspace"]["solves"]
    G0 = pa.CsrMat(ctx, 0, n, np.zeros(1, dtype=np.int32), np.zeros(0, dtype=np.int32), np.zeros(0))
    qp2 = pa.QP(ctx)
    qp2.SetOperator(op)
    qp2.SetRhs(ctx.vec_from(d))
    qp2.SetInitialVector(ctx.vec(n))
    qp2.SetBox(None, ctx.vec(n), None)
    qp2.SetEq(pa.QPPF(ctx, G0, orthonormal=True))
    qps2 = pa.QPS(ctx)
    qps2.SetQP(qp2)
    qps2.SetDefaultType()  # BE present -> SMALXE (qps.c:443-444)
    assert qps2.type == "smalxe"
    s2 = qps2.Solve()
    assert (s2.iteration, s2.reason, s2.inner_iter_accu) == (outer["iterations"], outer["reason"], outer["inner_iterations"])
    i2 = s2.inner
    assert (i2.reason, i2.nmv, i2.ncg, i2.nexp, i2.nprop) == (inner["reason"], inner["nmv"], inner["ncg"], inner["nexp"], inner["nprop"])
    # ex3_nullspace.out prints 13 KKT lines: the penalised QP, the dual QP with its (empty) equality constraint, the primal QP.
    # Entries of rounding size (1.08e-19 in the golden: a multiplier that is -1e-19 instead of 0) are compared by magnitude.
    kn = goldens["ex3_nullspace"]["kkt"]
    lam2 = qp2.x.to_numpy()
    box = qps2.ViewKKT()  # the dual QP's box lines (same numbers on lines 0-3 and 6-8 of the golden)
    for line, ref in zip(box, kn[:4]):
        m = re.match(r"r = (.*?)\s*= (\S+)\s+rO?/\|\|b\|\| = (\S+)", line)
        assert line.startswith("r = " + ref["name"])
        if float(ref["r"]) < 1e-15:
            assert float(m.group(2)) < 1e-12
        else:
            assert (m.group(2), m.group(3)) == (ref["r"], ref["r_rel"]), (line, ref)
    assert [k["r"] for k in kn[6:9]] == [k["r"] for k in kn[1:4]]
    xp2 = np.linalg.solve(Kd, p3["b"] - BI.T @ lam2)
    gap2 = BI @ xp2 - p3["cI"]
    v2 = [np.linalg.norm(np.maximum(gap2, 0.0)), abs(lam2 @ gap2)]
    for v, ref in zip(v2, (kn[10], kn[12])):
        assert same(v, ref["r"]) and same(v / nb, ref["r_rel"]), (v, v / nb, ref)
    assert np.linalg.norm(Kd @ xp2 - p3["b"] + BI.T @ lam2) <= 1e-13 and np.linalg.norm(np.minimum(lam2, 0.0)) <= 1e-15


def test_config3_shape_64_subdomains_dense_coarse_solve(ctx, oracle):
    """BASELINE configs[3] in miniature: 4x4x4 = 64 subdomains packed on one GPU (one concatenated block-diagonal CSR,
    64 independent block-CG solves per K^+ apply), G NOT orthonormalised => the coarse problem is the dense 384 x 384
    (GG')^{-1} applied on the device, SMALXE penalty term rho*G'G, inner MPGP with its own power method."""
    f = pa.CubeFeti((4, 4, 4), 1, contact=True)
    assert f.nsub == 64
    G, e = f.coarse(orthonormalize=False)
    assert G.shape[0] == 384
    Fd, pfo, d, lam_t, b_bar, lb_new = _oracle_dual(oracle, f, G, e, False)
    n = f.n_lambda
    A_or = oracle.Op(n, fn=lambda x: pfo.P(Fd @ pfo.P(x)))
    ref = oracle.smalxe(A_or, pfo.P(b_bar), np.zeros(n), oracle.Box(n, lb=lb_new), pfo)
    q = FetiDualQP(ctx, f.subset(range(f.nsub)), G, e, f.c, f.lb, orthonormal=False, kplus_rtol=1e-13)
    st = q.solve_smalxe()
    assert st.reason == ref["reason"] and st.reason > 0
    assert st.iteration == ref["iteration"]
    assert abs(st.inner_iter_accu - ref["inner_iter_accu"]) <= max(3, ref["inner_iter_accu"] // 25)
    # two solutions of an ill-conditioned dual QP, each converged to rtol 1e-5: they agree to a few 1e-3
    assert np.linalg.norm(q.lam.to_numpy() - ref["u"]) <= 5e-3 * np.linalg.norm(ref["u"])
    # G lambda = e to the solver tolerance (equality constraint of the dual QP)
    lam = q.dual_solution()
    assert np.linalg.norm(G @ lam - e) <= 1e-4 * max(1.0, np.linalg.norm(e))


def test_extension_matches_gluing_and_dense(ctx):
    """MATEXTENSION (the reference's default B type): B' lambda and B u through the condensed CSR + index sets must equal
    MATGLUING's result on the same constraints (both compute the same B' lambda / B u)."""
    f = pa.CubeFeti((2, 2, 1), 2, contact=True)
    Bt = f.B.T.tocsr()  # N x n_lambda: rows = primal dofs; condensed to the interface rows
    ris = np.flatnonzero(np.diff(Bt.indptr) > 0).astype(np.int32)
    cis = np.arange(f.n_lambda, dtype=np.int32)
    Acond = Bt[ris].tocsr()
    Acond.sort_indices()
    A = pa.CsrMat(ctx, Acond.shape[0], Acond.shape[1], Acond.indptr, Acond.indices, Acond.data)
    TA = pa.MatExtension(ctx, f.N, f.n_lambda, A, ris, cis)
    Bg = pa.MatGluing(ctx, f.N, f.n_lambda, f.leaves_row, f.leaves_root, f.leaves_sign)
    rng = np.random.default_rng(2)
    lam, u = rng.standard_normal(f.n_lambda), rng.standard_normal(f.N)
    x1, x2 = ctx.vec(f.N), ctx.vec(f.N)
    TA.mult(ctx.vec_from(lam), x1)
    Bg.mult(ctx.vec_from(lam), x2)
    assert np.array_equal(x1.to_numpy(), x2.to_numpy())
    assert np.allclose(x1.to_numpy(), f.B.T @ lam, rtol=1e-14, atol=1e-14)
    l1, l2 = ctx.vec(f.n_lambda), ctx.vec(f.n_lambda)
    TA.mult_transpose(ctx.vec_from(u), l1)
    Bg.mult_transpose(ctx.vec_from(u), l2)
    assert np.allclose(l1.to_numpy(), l2.to_numpy(), rtol=1e-14, atol=1e-14)
    assert np.allclose(l1.to_numpy(), f.B @ u, rtol=1e-13, atol=1e-14)
    with pytest.raises(pa.PermonHipError):
        pa.MatExtension(ctx, f.N, f.n_lambda, A, np.zeros_like(ris), cis)  # repeated row index


def test_contact_tfeti_end_to_end_properties(ctx):
    """Whole chain at a size the oracle cannot follow (222 k dof, n_lambda = 23 700; the 2.04 M-dof configs[2] run of the
    same script is recorded in profiles/r01_solve_configs2_smalxe.jsonl): dualise -> homogenise -> project -> SMALXE+MPGP,
    then size-independent properties of the solution -- dual feasibility, G lambda = e, and primal feasibility / contact
    complementarity after the rigid-body recovery."""
    f = pa.CubeFeti((2, 2, 2), 20, contact=True)
    G, e = f.coarse(orthonormalize=True)
    q = FetiDualQP(ctx, f.subset(range(8)), G, e, f.c, f.lb, orthonormal=True, kplus_rtol=1e-9)
    st = q.solve_smalxe(rtol=1e-5)
    assert st.reason == 2 and st.iteration < 100 and st.inner.nmv == st.inner.ncg + 2 * st.inner.nexp + st.inner.nprop + st.iteration
    lam = q.dual_solution()
    n = f.n_lambda
    assert lam[f.n_eq:].min() >= -1e-12  # lambda_I >= 0
    assert np.linalg.norm(G @ lam - e) <= 1e-5 * max(1.0, np.linalg.norm(e))
    u, Fl_minus_d = q.primal_solution(G)
    Ru = f.kernel_matrix()
    tight = (np.arange(n) < f.n_eq) | (lam > 1e-8 * np.abs(lam).max())
    alpha = np.linalg.lstsq((f.B @ Ru).toarray()[tight], Fl_minus_d[tight], rcond=None)[0]
    uu = u + Ru @ alpha
    Bu, scale = f.B @ uu, np.abs(uu).max()
    assert np.abs(Bu[:f.n_eq] - f.c[:f.n_eq]).max() <= 1e-3 * scale  # glued + Dirichlet
    assert (Bu[f.n_eq:] - f.c[f.n_eq:]).max() <= 1e-3 * scale  # no penetration
    gap = f.c[f.n_eq:] - Bu[f.n_eq:]
    assert np.abs(lam[f.n_eq:] * gap).max() <= 1e-3 * scale * np.abs(lam).max()  # complementarity
    assert (lam[f.n_eq:] > 0).sum() > 100  # a genuine contact zone


def test_mat_add_and_transpose_slots(ctx):
    """The remaining Mat op slots the reference fills (multadd / multtranspose / multtransposeadd of MATBLOCKDIAG
    matblockdiag.c:743-746, MATGLUING gluing.c:281-284, MATEXTENSION extension.c:1115-1118), incl. the in-place form."""
    rng = np.random.default_rng(21)
    f = pa.CubeFeti((2, 2, 1), 2, contact=True)
    N, nl = f.N, f.n_lambda
    # a non-symmetric block-diagonal matrix so that the transpose slots are really exercised
    blocks = [sp.random(f.n_i, f.n_i, density=0.05, random_state=5 + s, format="csr") + sp.identity(f.n_i) for s in range(f.nsub)]
    Kd = sp.block_diag(blocks, format="csr")
    K = pa.MatBlockDiag.from_scipy(ctx, f.block_rowstart, Kd)
    x, y1 = rng.standard_normal(N), rng.standard_normal(N)
    xd, y1d, yd = ctx.vec_from(x), ctx.vec_from(y1), ctx.vec(N)
    K.mult_transpose(xd, yd)
    assert np.allclose(yd.to_numpy(), Kd.T @ x, rtol=1e-13, atol=1e-13)
    K.mult_add(xd, y1d, yd)
    assert np.allclose(yd.to_numpy(), y1 + Kd @ x, rtol=1e-13, atol=1e-13)
    K.mult_transpose_add(xd, y1d, yd)
    assert np.allclose(yd.to_numpy(), y1 + Kd.T @ x, rtol=1e-13, atol=1e-13)
    yin = ctx.vec_from(y1)
    K.mult_add(xd, yin, yin)  # v2 can be the same as v3 (matblockdiag.c:227)
    assert np.allclose(yin.to_numpy(), y1 + Kd @ x, rtol=1e-13, atol=1e-13)
    # gluing
    Bg = pa.MatGluing(ctx, N, nl, f.leaves_row, f.leaves_root, f.leaves_sign)
    lam, l1 = rng.standard_normal(nl), rng.standard_normal(nl)
    out_x, out_l = ctx.vec(N), ctx.vec(nl)
    Bg.mult_add(ctx.vec_from(lam), y1d, out_x)
    assert np.allclose(out_x.to_numpy(), y1 + f.B.T @ lam, rtol=1e-13, atol=1e-13)
    Bg.mult_transpose_add(xd, ctx.vec_from(l1), out_l)
    assert np.allclose(out_l.to_numpy(), l1 + f.B @ x, rtol=1e-13, atol=1e-13)
    # extension: same operator through the condensed CSR + index sets
    Bt = f.B.T.tocsr()
    ris = np.flatnonzero(np.diff(Bt.indptr) > 0).astype(np.int32)
    Acond = Bt[ris].tocsr()
    Acond.sort_indices()
    A = pa.CsrMat(ctx, Acond.shape[0], Acond.shape[1], Acond.indptr, Acond.indices, Acond.data)
    TA = pa.MatExtension(ctx, N, nl, A, ris, np.arange(nl, dtype=np.int32))
    TA.mult_add(ctx.vec_from(lam), y1d, out_x)
    assert np.allclose(out_x.to_numpy(), y1 + f.B.T @ lam, rtol=1e-13, atol=1e-13)
    TA.mult_transpose_add(xd, ctx.vec_from(l1), out_l)
    assert np.allclose(out_l.to_numpy(), l1 + f.B @ x, rtol=1e-13, atol=1e-13)
    xin = ctx.vec_from(y1)
    TA.mult_add(ctx.vec_from(lam), xin, xin)
    assert np.allclose(xin.to_numpy(), y1 + f.B.T @ lam, rtol=1e-13, atol=1e-13)


def test_replicated_dual_arithmetic_is_bitwise_reproducible(ctx):
    """The multi-GPU design replicates the dual-space MPGP / SMALXE arithmetic on every rank and relies on it being
    deterministic (fixed reduction trees, no atomics on the data path): two independent set-ups + solves of the same contact
    problem on one GPU must agree bit for bit -- multipliers, counters and the final residual -- although the number of
    enqueued no-op launches differs between them (the K^+ driver adapts its look-ahead to the previous iteration counts)."""
    f = pa.CubeFeti((2, 2, 2), 3, contact=True)
    G, e = f.coarse(orthonormalize=True)
    runs = []
    for rep in range(2):
        q = FetiDualQP(ctx, f.subset(range(f.nsub)), G, e, f.c, f.lb, orthonormal=True, kplus_rtol=1e-10)
        if rep:  # perturb the driver's look-ahead state: a different number of idle launches, same arithmetic
            w = ctx.vec(f.N)
            q.Kplus.mult(ctx.vec_from(np.ones(f.N)), w)
        st = q.solve_smalxe()
        runs.append((q.lam.to_numpy(), st.iteration, st.inner_iter_accu, st.inner.nmv, st.rnorm))
    a, b = runs
    assert a[1:4] == b[1:4] and a[4] == b[4]
    assert np.array_equal(a[0], b[0])
