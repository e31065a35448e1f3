"""GPU parity tests of the explicit local dual operators (pmh_fexplicit): the exact-K^+ path of F = B K^+ B'
(MatInvExplicitly_Inv, src/mat/impls/inv/matinv.c:670-730, restricted to the dofs B touches)."""
import numpy as np
import pytest
import scipy.sparse as sp

import permon_amd as pa
from permon_amd.chain import FetiDualQP

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = pa.Context(0)
    yield c
    c.close()


def _dense_F(f):
    """Oracle side: F = B K^+ B' with the Moore-Penrose inverse of every block (numpy pinv)."""
    Kp = np.linalg.pinv(f.Ki.toarray(), rcond=1e-10, hermitian=True)
    B = f.B.toarray()
    F = np.zeros((f.n_lambda, f.n_lambda))
    for s in range(f.nsub):
        Bs = B[:, s * f.n_i:(s + 1) * f.n_i]
        F += Bs @ Kp @ Bs.T
    return F, Kp


def test_block_classes_host():
    f = pa.CubeFeti((2, 1, 1), 2)
    K = sp.block_diag([f.Ki, f.Ki, 2.0 * f.Ki, f.Ki], format="csr")
    cls = pa.csr_block_classes(np.arange(5) * f.n_i, K)
    assert cls.tolist() == [0, 0, 1, 0]


@pytest.mark.parametrize("share,storage", [(True, "sym"), (False, "sym"), (True, "full"), (True, "class"), (True, "class_sym"), (False, "class_sym")])
def test_explicit_blocks_vs_pinv(ctx, share, storage):
    """nel = 2: every W_b equals the dense pseudo-inverse of K_b on Gamma_b; F through the explicit path equals B pinv(K) B'."""
    f = pa.CubeFeti((2, 2, 1), 2, contact=True)
    G, e = f.coarse()
    q = FetiDualQP(ctx, f.subset(range(f.nsub)), G, e, f.c, f.lb, kplus_rtol=1e-13, explicit=dict(rtol=1e-13, share_congruent=share, storage=storage))
    Fref, Kp = _dense_F(f)
    n_solves, secs = q.E.assemble_stats()
    if share:  # the four congruent cubes share the union of their Gamma sets
        assert n_solves < int(q.E.n_gamma.sum())
    else:
        assert n_solves == int(q.E.n_gamma.sum())
    for b in range(f.nsub):
        W, g = q.E.block(b)
        gl = g - b * f.n_i
        assert np.all(np.diff(g) > 0) and gl.min() >= 0 and gl.max() < f.n_i
        ref = Kp[np.ix_(gl, gl)]
        assert np.max(np.abs(W - ref)) <= 1e-10 * np.max(np.abs(ref))
        assert np.max(np.abs(W - W.T)) <= 1e-11 * np.max(np.abs(ref))
    rng = np.random.default_rng(5)
    lam = rng.standard_normal(f.n_lambda)
    y = ctx.vec(f.n_lambda)
    q.F.mult(ctx.vec_from(lam), y)  # the chain's F applies through E now
    assert np.linalg.norm(y.to_numpy() - Fref @ lam) <= 1e-10 * np.linalg.norm(Fref @ lam)
    y2 = ctx.vec(f.n_lambda)
    q.E.mult(ctx.vec_from(lam), y2)
    assert np.array_equal(y.to_numpy(), y2.to_numpy())


@pytest.mark.parametrize("storage", ["sym", "full", "class", "class_sym", "class_sym/3"])
def test_explicit_vs_iterative_F(ctx, storage, monkeypatch):
    """2x2x2 cubes, nel = 6: F_dense lambda vs the iterative B K^+ B' lambda (rtol 1e-13) <= 1e-10; the dense kernel (SYMV on the
    lower block-triangle / GEMV on the full matrix) vs numpy, and bitwise reproducible."""
    if "/" in storage:  # the persistent grid of k_fxs_symm8 cut down to 3 workgroups: several items (mega band, column range) per workgroup
        storage, nwg = storage.split("/")
        monkeypatch.setenv("PMH_FXM_NWG", nwg)
    f = pa.CubeFeti((2, 2, 2), 6, contact=True)
    G, e = f.coarse()
    loc = f.subset(range(f.nsub))
    q_it = FetiDualQP(ctx, loc, G, e, f.c, f.lb, kplus_rtol=1e-13)
    q_ex = FetiDualQP(ctx, loc, G, e, f.c, f.lb, kplus_rtol=1e-13, explicit=dict(rtol=1e-13, storage=storage))
    rng = np.random.default_rng(9)
    for _ in range(3):
        lam = rng.standard_normal(f.n_lambda)
        ya, yb = ctx.vec(f.n_lambda), ctx.vec(f.n_lambda)
        q_it.F.mult(ctx.vec_from(lam), ya)
        q_ex.F.mult(ctx.vec_from(lam), yb)
        ref = ya.to_numpy()
        assert np.linalg.norm(yb.to_numpy() - ref) <= 1e-10 * np.linalg.norm(ref)
    # chain vectors agree (d, b: computed with the respective F)
    assert np.linalg.norm(q_ex.b.to_numpy() - q_it.b.to_numpy()) <= 1e-9 * np.linalg.norm(q_it.b.to_numpy())
    if storage in ("class", "class_sym"):  # multivector numbering: the dense kernel is covered through F above; reproducibility here
        ntot, _ = q_ex.E.compressed_size()
        xm = ctx.vec_from(rng.standard_normal(ntot))
        ya, yb = ctx.vec(ntot), ctx.vec(ntot)
        q_ex.E.dense_mult(xm, ya)
        q_ex.E.dense_mult(xm, yb)
        assert np.array_equal(ya.to_numpy(), yb.to_numpy())
        for b in range(f.nsub):  # every W_b is a principal sub-matrix of the shared W_c
            W, g = q_ex.E.block(b)
            assert np.max(np.abs(W - W.T)) <= 1e-10 * np.max(np.abs(W))
        return
    # the dense kernel alone against numpy, block by block
    ntot, gs = q_ex.E.compressed_size()
    xh = rng.standard_normal(ntot)
    for b in range(f.nsub):  # pad entries (odd n_Gamma) must be zero in x
        xh[gs[b] + q_ex.E.n_gamma[b]:gs[b + 1]] = 0.0
    yh, yh2 = ctx.vec(ntot), ctx.vec(ntot)
    q_ex.E.dense_mult(ctx.vec_from(xh), yh)
    q_ex.E.dense_mult(ctx.vec_from(xh), yh2)
    assert np.array_equal(yh.to_numpy(), yh2.to_numpy())  # fixed summation order
    yh = yh.to_numpy()
    for b in range(f.nsub):
        W, _ = q_ex.E.block(b)
        n = q_ex.E.n_gamma[b]
        ref = W @ xh[gs[b]:gs[b] + n]
        assert np.max(np.abs(yh[gs[b]:gs[b] + n] - ref)) <= 1e-13 * np.max(np.abs(W)) * np.linalg.norm(xh) * np.sqrt(n)


@pytest.mark.parametrize("sub,nel,nsym", [((2, 2, 2), 2, 48), ((2, 2, 2), 4, 48), ((2, 2, 1), 5, 8)])
def test_explicit_setup_by_symmetry(ctx, sub, nel, nsym):
    """PMH_FX_CLASS_SYM assembled with the cube's symmetries (feti.box_symmetries, checked against K): one K^+ solve per orbit of rows of W_c
    (up to 48 x fewer) + one batch of direct solves as the library's self-check; every W_b still equals pinv(K_b) on Gamma_b, F = B pinv(K) B'.
    A permutation that is NOT a symmetry makes the assembly fail loudly."""
    f = pa.CubeFeti(sub, nel, contact=True)
    G, e = f.coarse()
    nn = nel + 1
    loc = f.subset(range(f.nsub))
    q0 = FetiDualQP(ctx, loc, G, e, f.c, f.lb, kplus_rtol=1e-13, explicit=dict(rtol=1e-13, storage="class_sym"))
    q = FetiDualQP(ctx, loc, G, e, f.c, f.lb, kplus_rtol=1e-13, explicit=dict(rtol=1e-13, storage="class_sym", symmetry=dict(dims=(nn, nn, nn), ndof=3, orbit=False)))
    # 2 x 2 x 2 cubes: the union of the touched faces is the whole boundary, closed under all 48 operations; 2 x 2 x 1: the operations that keep
    # the touched set (no top face) are kept, the others dropped
    assert q.explicit_symmetries == nsym
    n0, n1 = q0.E.assemble_stats()[0], q.E.assemble_stats()[0]
    assert n1 < n0 / (nsym / 8.0) + 8 * f.nsub  # orbits of <= nsym rows (rows on symmetry planes have shorter ones) + the self-check batch (one row per slot: 8 columns per block on the multi-right-hand-side K^+)
    Fref, Kp = _dense_F(f)
    for b in range(f.nsub):
        W, g = q.E.block(b)
        gl = g - b * f.n_i
        ref = Kp[np.ix_(gl, gl)]
        assert np.max(np.abs(W - ref)) <= 1e-10 * np.max(np.abs(ref))
    lam = np.random.default_rng(6).standard_normal(f.n_lambda)
    y = ctx.vec(f.n_lambda)
    q.F.mult(ctx.vec_from(lam), y)
    assert np.linalg.norm(y.to_numpy() - Fref @ lam) <= 1e-10 * np.linalg.norm(Fref @ lam)
    # not a symmetry: the identity + a swap of two touched dofs
    q2 = FetiDualQP(ctx, loc, G, e, f.c, f.lb, kplus_rtol=1e-13)
    cls = np.zeros(f.nsub, dtype=np.int32)
    E = pa.MatExplicitDual(q2.B, q2.K, storage="class_sym", block_class=cls)
    u = E.class_union(0)
    perm = np.tile(np.arange(f.n_i, dtype=np.int32), (2, 1))
    perm[1, u[0]], perm[1, u[-1]] = u[-1], u[0]
    E.set_class_symmetry(0, perm, np.ones((2, f.n_i), dtype=np.int8))
    with pytest.raises(Exception, match="symmetr"):
        E.assemble(q2.Kplus, slot_class=cls, block_class=cls, rtol=1e-13)
    E.destroy()


@pytest.mark.parametrize("sub,nel,nsym", [((2, 2, 2), 2, 48), ((2, 2, 2), 5, 48), ((2, 2, 1), 5, 8)])
def test_explicit_orbit_storage(ctx, sub, nel, nsym):
    """PMH_FX_CLASS_ORBIT: only the rows of W_c of the orbit representatives under the cube's symmetries are stored; the dense apply is the GEMM
    (representatives) x (operations x 8 right-hand sides) on the fp64 matrix instruction.  Every W_b rebuilt from it equals pinv(K_b) on Gamma_b,
    F = B pinv(K) B', bitwise reproducible; three ranks' shares of the representatives sum to F."""
    f = pa.CubeFeti(sub, nel, contact=True)
    G, e = f.coarse()
    nn = nel + 1
    loc = f.subset(range(f.nsub))
    q = FetiDualQP(ctx, loc, G, e, f.c, f.lb, kplus_rtol=1e-13, explicit=dict(rtol=1e-13, storage="class_orbit", symmetry=dict(dims=(nn, nn, nn), ndof=3)))
    assert q.explicit_storage == "class_orbit" and q.explicit_symmetries == nsym
    n_c = q.E.class_union(0).size
    assert q.E.assemble_stats()[0] < n_c / (nsym / 8.0) + 8 * f.nsub and q.E.dense_bytes < 8.0 * n_c * n_c / (nsym / 8.0) and q.E.apply_flops() > 0
    Fref, Kp = _dense_F(f)
    for b in range(f.nsub):
        W, g = q.E.block(b)
        gl = g - b * f.n_i
        ref = Kp[np.ix_(gl, gl)]
        assert np.max(np.abs(W - ref)) <= 1e-10 * np.max(np.abs(ref))
    rng = np.random.default_rng(6)
    lam = rng.standard_normal(f.n_lambda)
    lv, y, y2 = ctx.vec_from(lam), ctx.vec(f.n_lambda), ctx.vec(f.n_lambda)
    q.F.mult(lv, y)
    q.F.mult(lv, y2)
    assert np.array_equal(y.to_numpy(), y2.to_numpy())
    assert np.linalg.norm(y.to_numpy() - Fref @ lam) <= 1e-10 * np.linalg.norm(Fref @ lam)
    # the auto rule: with >= 16 operations "class_sym" + symmetry turns into the orbit storage, with fewer it stays on the symmetric tiles
    q3 = FetiDualQP(ctx, loc, G, e, f.c, f.lb, kplus_rtol=1e-13, explicit=dict(rtol=1e-13, storage="class_sym", symmetry=dict(dims=(nn, nn, nn), ndof=3)))
    assert q3.explicit_storage == ("class_orbit" if nsym >= 16 else "class_sym")
    # several GPUs rehearsed: a contiguous range of the representatives per rank
    glob = dict(n_x=f.N, block_rowstart=f.block_rowstart, leaves_row=f.leaves_row, leaves_root=f.leaves_root, leaves_sign=f.leaves_sign)
    tot = np.zeros(f.n_lambda)
    for r in range(3):
        lr = f.subset([r])
        qr = FetiDualQP(ctx, lr, G, e, f.c, f.lb, kplus_rtol=1e-13)
        qr.assemble_explicit(lr, rtol=1e-13, stripe=(r, 3, glob), storage="class_orbit", symmetry=dict(dims=(nn, nn, nn), ndof=3))
        yr = ctx.vec(f.n_lambda)
        qr.F.mult(lv, yr)
        tot += yr.to_numpy()
    assert np.linalg.norm(tot - Fref @ lam) <= 1e-10 * np.linalg.norm(Fref @ lam)


@pytest.mark.parametrize("tm", [144, 128, 112, 96, 80])
def test_explicit_orbit_row_tiles(ctx, tm, monkeypatch):
    """Every row tile of the orbit GEMM k_fxo_gemm16<NI, NWM> (v_mfma_f64_16x16x4: tiles 144 / 112 / 80 with 1 x 4 waves, 128 / 96 with 2 x 2), forced through PMH_FXO_TM,
    two or more row tiles per class (nel = 20: n_c = 7206, ~165 representatives), F = B pinv(K) B' through each and the blocks rebuilt from the pre-tiled rows.  (The
    4x4x4_4b kernels of rounds 2-3 went at the end of round 6.)"""
    monkeypatch.setenv("PMH_FXO_TM", str(tm))
    nel = 20
    f = pa.CubeFeti((2, 2, 2), nel, contact=True)
    G, e = f.coarse()
    nn = nel + 1
    loc = f.subset(range(f.nsub))
    q = FetiDualQP(ctx, loc, G, e, f.c, f.lb, kplus_rtol=1e-13, explicit=dict(rtol=1e-13, storage="class_orbit", symmetry=dict(dims=(nn, nn, nn), ndof=3)))
    assert q.explicit_storage == "class_orbit" and q.explicit_symmetries == 48
    rng = np.random.default_rng(16)
    lam = rng.standard_normal(f.n_lambda)
    lv, y = ctx.vec_from(lam), ctx.vec(f.n_lambda)
    q.F.mult(lv, y)
    # reference: the iterative K^+ at rtol 1e-13 (a dense pseudo-inverse of 27 783-dof blocks is out of reach)
    q0 = FetiDualQP(ctx, loc, G, e, f.c, f.lb, kplus_rtol=1e-13)
    y0 = ctx.vec(f.n_lambda)
    q0.F.mult(lv, y0)
    assert np.linalg.norm(y.to_numpy() - y0.to_numpy()) <= 1e-9 * np.linalg.norm(y0.to_numpy())
    W, g = q.E.block(3)
    assert np.max(np.abs(W - W.T)) <= 1e-9 * np.max(np.abs(W))
    if tm != 128:  # the row tile changes the tiles' column lists, hence the k segments and the cuts of the k sums (fxo_prepare): the 128-row kernel agrees to rounding
        monkeypatch.setenv("PMH_FXO_TM", "128")
        q1 = FetiDualQP(ctx, loc, G, e, f.c, f.lb, kplus_rtol=1e-13, explicit=dict(rtol=1e-13, storage="class_orbit", symmetry=dict(dims=(nn, nn, nn), ndof=3)))
        y1 = ctx.vec(f.n_lambda)
        q1.F.mult(lv, y1)
        assert np.linalg.norm(y.to_numpy() - y1.to_numpy()) <= 1e-13 * np.linalg.norm(y1.to_numpy())
        y2 = ctx.vec(f.n_lambda)  # and the product is deterministic: the same bits from a second apply
        q1.F.mult(lv, y2)
        assert np.array_equal(y1.to_numpy(), y2.to_numpy())


def test_explicit_orbit_plan_variants(ctx, monkeypatch):
    """The plan of the orbit GEMM (fxo_prepare): k segments of equal non-zero columns of the gathered operand, units = (row tile, segment) with their own column lists,
    equal pieces that may end one unit and begin the next (workgroups with several items).  Every variant of the plan -- one segment, unit-aligned splits, every
    signature its own segment, few long pieces, many short ones -- is the same product: F lambda agrees to rounding with the default plan and with the inner-Krylov K^+."""
    nel = 12
    f = pa.CubeFeti((2, 2, 2), nel, contact=True)
    G, e = f.coarse()
    nn = nel + 1
    loc = f.subset(range(f.nsub))
    lam = np.random.default_rng(21).standard_normal(f.n_lambda)
    lv = ctx.vec_from(lam)

    def F_lambda(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        q = FetiDualQP(ctx, loc, G, e, f.c, f.lb, kplus_rtol=1e-13, explicit=dict(rtol=1e-13, storage="class_orbit", symmetry=dict(dims=(nn, nn, nn), ndof=3)))
        for k in env:
            monkeypatch.delenv(k)
        assert q.explicit_storage == "class_orbit"
        y, y2 = ctx.vec(f.n_lambda), ctx.vec(f.n_lambda)
        q.F.mult(lv, y)
        q.F.mult(lv, y2)
        assert np.array_equal(y.to_numpy(), y2.to_numpy())  # fixed orders: the same bits from a second apply
        return y.to_numpy()

    y0 = F_lambda({})
    q0 = FetiDualQP(ctx, loc, G, e, f.c, f.lb, kplus_rtol=1e-13)
    yi = ctx.vec(f.n_lambda)
    q0.F.mult(lv, yi)
    assert np.linalg.norm(y0 - yi.to_numpy()) <= 1e-9 * np.linalg.norm(y0)
    for env in ({"PMH_FXO_NO_KSEG": "1"}, {"PMH_FXO_NO_STREAMK": "1"}, {"PMH_FXO_NO_KSEG": "1", "PMH_FXO_NO_STREAMK": "1"}, {"PMH_FXO_SEGMIN": "1"}, {"PMH_FXO_SEGMIN": "1", "PMH_FXO_SLOTS": "4096"},
                {"PMH_FXO_SLOTS": "48"}, {"PMH_FXO_SLOTS": "4096", "PMH_FXO_MINCH": "1"}, {"PMH_FXO_SPLIT": "3"}, {"PMH_FXO_NO_PRUNE": "1"}):
        y = F_lambda(env)
        assert np.linalg.norm(y - y0) <= 1e-13 * np.linalg.norm(y0), env


@pytest.mark.parametrize("nel", [3, 7, 11, 13])
def test_explicit_orbit_sizes_against_iterative_kplus(ctx, nel):
    """Odd sizes (row / k / column remainders of the orbit GEMM's tiles; scripts/orbit_size_sweep.py runs more of them): F through the orbit storage equals F
    through the inner-Krylov K^+ to 1e-9, the contact solve takes the same SMALXE / MPGP counts and ends at the same lambda."""
    f = pa.CubeFeti((2, 2, 2), nel, contact=True)
    G, e = f.coarse()
    loc = f.subset(range(f.nsub))
    nn = nel + 1
    qi = FetiDualQP(ctx, loc, G, e, f.c, f.lb, kplus_rtol=1e-12)
    qo = FetiDualQP(ctx, loc, G, e, f.c, f.lb, kplus_rtol=1e-12, explicit=dict(rtol=1e-13, storage="class_orbit", symmetry=dict(dims=(nn, nn, nn), ndof=3)))
    assert qo.explicit_storage == "class_orbit" and qo.explicit_symmetries == 48
    lam = np.random.default_rng(nel).standard_normal(f.n_lambda)
    lv, y0, y1 = ctx.vec_from(lam), ctx.vec(f.n_lambda), ctx.vec(f.n_lambda)
    qi.F.mult(lv, y0)
    qo.F.mult(lv, y1)
    assert np.linalg.norm(y1.to_numpy() - y0.to_numpy()) <= 1e-9 * np.linalg.norm(y0.to_numpy())
    si, so = qi.solve_smalxe(rtol=1e-6), qo.solve_smalxe(rtol=1e-6)
    assert (si.iteration, si.inner_iter_accu, si.inner.ncg, si.inner.nexp, si.reason) == (so.iteration, so.inner_iter_accu, so.inner.ncg, so.inner.nexp, so.reason)
    li, lo = qi.dual_solution(), qo.dual_solution()
    assert np.linalg.norm(li - lo) <= 1e-7 * np.linalg.norm(li)


def test_explicit_contact_solve_same_counts(ctx):
    """Contact TFETI (SMALXE + MPGP) through the explicit F: same outer / inner counts and solution as the iterative K^+."""
    f = pa.CubeFeti((2, 2, 2), 5, contact=True)
    G, e = f.coarse()
    loc = f.subset(range(f.nsub))
    res = []
    for explicit in (None, dict(rtol=1e-13)):
        q = FetiDualQP(ctx, loc, G, e, f.c, f.lb, kplus_rtol=1e-12, explicit=explicit)
        st = q.solve_smalxe(rtol=1e-6)
        res.append((st, q.dual_solution()))
    (sa, la), (sb, lb_) = res
    assert (sa.iteration, sa.inner_iter_accu, sa.reason) == (sb.iteration, sb.inner_iter_accu, sb.reason)
    assert (sa.inner.ncg, sa.inner.nexp, sa.inner.nprop) == (sb.inner.ncg, sb.inner.nexp, sb.inner.nprop)
    assert np.linalg.norm(la - lb_) <= 1e-7 * np.linalg.norm(la)


def test_explicit_replica_solver(ctx):
    """One block on the rank (the 8-GPU share): the columns come from a solver with several replica slots of the same matrix."""
    f = pa.CubeFeti((2, 1, 1), 4, contact=True)
    G, e = f.coarse()
    loc = f.subset([0])
    made = {}

    def factory(nslots):
        Kb = pa.MatBlockDiag.from_scipy(ctx, np.arange(nslots + 1, dtype=np.int32) * f.n_i, sp.block_diag([f.Ki] * nslots, format="csr"))
        made["K"] = Kb
        return pa.MatInv(Kb, rtol=1e-13, max_it=20000, nullspace=np.tile(loc["R"], (1, nslots)))

    q = FetiDualQP(ctx, loc, G, e, f.c, f.lb, kplus_rtol=1e-13, explicit=dict(rtol=1e-13, min_slots=4, solver_factory=factory, multi_rhs=False))  # (with multi_rhs the block's own solver has 8 slots)
    assert made["K"].nblocks == 4
    Kp = np.linalg.pinv(f.Ki.toarray(), rcond=1e-10, hermitian=True)
    W, g = q.E.block(0)
    ref = Kp[np.ix_(g, g)]
    assert np.max(np.abs(W - ref)) <= 1e-10 * np.max(np.abs(ref))
    n_solves, _ = q.E.assemble_stats()
    assert n_solves == q.E.n_gamma[0]


@pytest.mark.parametrize("storage", ["sym", "class", "class_sym"])
def test_striped_shares_sum_to_F(ctx, storage):
    """Several GPUs rehearsed on one: the operator spans all blocks, rank r of 3 keeps the 128-row stripes idx = r (mod 3); the sum of
    the three ranks' applies (the all-reduce) equals F lambda of the unstriped operator, each share is bitwise reproducible, and the
    shares split the set-up solves."""
    # n_Gamma > 128: several stripes per block; "class_sym" deals whole mega bands of 1024 rows of W_c: n_c = 3 534 gives 4 of them
    f = pa.CubeFeti((2, 2, 1), 14 if storage == "class_sym" else 8, contact=True)
    G, e = f.coarse()
    loc = f.subset(range(f.nsub))
    q0 = FetiDualQP(ctx, loc, G, e, f.c, f.lb, kplus_rtol=1e-13, explicit=dict(rtol=1e-13))
    assert q0.E.n_gamma.min() > 256
    lam = np.random.default_rng(2).standard_normal(f.n_lambda)
    lv, y0 = ctx.vec_from(lam), ctx.vec(f.n_lambda)
    q0.F.mult(lv, y0)
    glob = dict(n_x=f.N, block_rowstart=f.block_rowstart, leaves_row=f.leaves_row, leaves_root=f.leaves_root, leaves_sign=f.leaves_sign)
    tot, solves = np.zeros(f.n_lambda), []
    for r in range(3):
        # rank r owns blocks [r] only (a 1-block K^+), but applies its stripes of ALL four W_b
        lr = f.subset([r])
        q = FetiDualQP(ctx, lr, G, e, f.c, f.lb, kplus_rtol=1e-13)
        sym = dict(dims=(f.nel + 1,) * 3, ndof=3, orbit=False) if storage == "class_sym" else None  # the 4-mega-band case also takes its rows from orbit representatives
        E = q.assemble_explicit(lr, rtol=1e-13, stripe=(r, 3, glob), storage=storage, symmetry=sym)
        y, y2 = ctx.vec(f.n_lambda), ctx.vec(f.n_lambda)
        q.F.mult(lv, y)
        q.F.mult(lv, y2)
        assert np.array_equal(y.to_numpy(), y2.to_numpy())
        tot += y.to_numpy()
        solves.append(E.assemble_stats()[0])
    ref = y0.to_numpy()
    assert np.linalg.norm(tot - ref) <= 1e-10 * np.linalg.norm(ref)
    assert max(solves) < q0.E.assemble_stats()[0] and (sum(solves) >= q0.E.assemble_stats()[0] or storage == "class_sym")  # by symmetry: every rank solves (nearly) all orbit representatives, far fewer than rows


def test_contact_solve_one_call(ctx):
    """pmh_feti_contact_solve through its Python binding (the C example runs the same entry from plain C): counts of the Python-orchestrated
    chain, a feasible primal solution with glued interfaces, and the non-explicit path of the same call."""
    f = pa.CubeFeti((2, 2, 1), 6, contact=True)
    G, e = f.coarse(orthonormalize=False)  # orthonormalised implicitly, the default of pmh_feti_contact_solve
    q = FetiDualQP(ctx, f.subset(range(f.nsub)), G, e, f.c, f.lb, orthonormal="implicit", kplus_rtol=1e-9, mg_box=dict(dims=[(7, 7, 7)] * f.nsub, ndof=3, min_nodes=400), mg_precision="fp16", bsr3=True,
                   explicit=dict(rtol=1e-12))
    ref = q.solve_smalxe(rtol=1e-5)
    for explicit in (True, False):
        u, lam, st = pa.FETIContactSolve(ctx, f, explicit=explicit)
        s = st.smalxe
        assert (s.reason, s.iteration, s.inner_iter_accu, s.inner.nmv, s.inner.ncg, s.inner.nexp) == (ref.reason, ref.iteration, ref.inner_iter_accu, ref.inner.nmv, ref.inner.ncg, ref.inner.nexp)
        assert st.coarse_dim == 6 * f.nsub and st.norm_Glambda_minus_e <= 1e-5 and (st.explicit_solves > 0) == explicit
        Bu, scale = f.B @ u, np.abs(u).max()
        assert np.abs(Bu[:f.n_eq] - f.c[:f.n_eq]).max() <= 1e-3 * scale and (Bu[f.n_eq:] - f.c[f.n_eq:]).max() <= 1e-3 * scale
        assert lam[f.n_eq:].min() >= -1e-12 and np.linalg.norm(lam - q.dual_solution()) <= 1e-6 * np.linalg.norm(lam)
        assert np.linalg.norm(f.K @ u - (f.f - f.B.T @ lam)) <= 1e-4 * np.linalg.norm(f.f)  # equilibrium


def test_contact_solve_one_call_non_congruent_cubes(ctx):
    """pmh_feti_contact_solve on cubes of DIFFERENT materials (one class per block) with the orbit storage asked for: the driver closes every class's touched set under the cube's
    group (pmh_box_symmetry_closure + pmh_fexplicit_create_shared_orbit_union), all 48 operations survive, the set-up needs one K^+ solve per orbit of boundary dofs -- and the solve
    is the one the inner-Krylov K^+ gives."""
    f = pa.CubeFeti((2, 2, 1), 6, contact=True, young=[1.0, 1.5, 2.0, 3.0])
    assert not f.congruent
    u0, lam0, st0 = pa.FETIContactSolve(ctx, f, explicit=False)
    u1, lam1, st1 = pa.FETIContactSolve(ctx, f, explicit=True, explicit_storage="class_orbit", explicit_symmetry=True)
    assert st1.explicit_symmetries == 48
    nn = 7
    assert 0 < st1.explicit_solves <= f.nsub * (3 * (nn ** 3 - (nn - 2) ** 3) // 48 + 3 * nn) < f.nsub * 3 * nn * nn  # orbits of the boundary, not the touched dofs one by one
    s0, s1 = st0.smalxe, st1.smalxe
    assert (s0.reason, s0.iteration, s0.inner_iter_accu, s0.inner.ncg, s0.inner.nexp) == (s1.reason, s1.iteration, s1.inner_iter_accu, s1.inner.ncg, s1.inner.nexp)
    assert np.linalg.norm(lam1 - lam0) <= 1e-6 * np.linalg.norm(lam0) and np.linalg.norm(u1 - u0) <= 1e-5 * np.linalg.norm(u0)


def test_explicit_edge_cases(ctx):
    """A block that B does not touch at all (n_Gamma = 0), a block with a single touched dof, non-congruent blocks, n_Gamma around the
    32 / 128 padding boundaries -- F through the explicit operators against the iterative F; striping refuses the full storage."""
    rng = np.random.default_rng(21)
    for ntouch in ((0, 1, 33), (127, 128, 129), (5, 0, 260)):
        f = pa.CubeFeti((3, 1, 1), 4, contact=False)  # three 5^3-node cubes: 375 dofs each
        K = sp.block_diag([f.Ki, 1.5 * f.Ki, f.Ki], format="csr")  # block 1 is not congruent with the others
        rows, roots, vals = [], [], []
        nl = 0
        for b, nt in enumerate(ntouch):
            dofs = np.sort(rng.choice(f.n_i, size=nt, replace=False)) + b * f.n_i
            for d in dofs:  # one or two dual rows per touched dof
                for _ in range(1 + int(rng.integers(0, 2))):
                    rows.append(d), roots.append(int(rng.integers(0, 40))), vals.append(float(rng.choice([-1.0, 1.0, 0.5])))
        nl = 40
        Kd = pa.MatBlockDiag.from_scipy(ctx, f.block_rowstart, K)
        Kp = pa.MatInv(Kd, rtol=1e-13, max_it=5000, nullspace=f.R)
        B = pa.MatGluing(ctx, f.N, nl, np.array(rows, dtype=np.int32), np.array(roots, dtype=np.int32), np.array(vals))
        F_it = pa.MatCreateFetiDual(B, Kp)
        lam = rng.standard_normal(nl)
        y_it, y_ex = ctx.vec(nl), ctx.vec(nl)
        F_it.mult(ctx.vec_from(lam), y_it)
        for storage in ("sym", "full"):
            E = pa.MatExplicitDual(B, Kd, storage=storage)
            assert E.n_gamma.tolist() == list(ntouch)
            cls = pa.csr_block_classes(f.block_rowstart, K)
            assert cls.tolist() == [0, 1, 0]
            E.assemble(Kp, slot_class=cls, block_class=cls, rtol=1e-13)
            Kp.attach_explicit(E)
            F_it.mult(ctx.vec_from(lam), y_ex)  # the same operator object now applies through E
            Kp.attach_explicit(None)
            ref = y_it.to_numpy()
            assert np.linalg.norm(y_ex.to_numpy() - ref) <= 1e-9 * max(np.linalg.norm(ref), 1e-300)
            if storage == "full":
                with pytest.raises(pa.PermonHipError):
                    E.set_stripe(0, 2)
            E.destroy()


def test_explicit_orbit_several_classes(ctx):
    """PMH_FX_CLASS_ORBIT with SEVERAL block classes (one material per subdomain: 8 classes of one block each): every class keeps the operations under which ITS touched
    set is closed (a single block's three interface faces + a Dirichlet or contact face: 2 ... 8 of the cube's 48), has its own representatives, row tile, k segments and
    launch (fxo_prepare / fxo_gemm per class, the workgroup tables sliced per class).  F against the inner-Krylov K^+, the same SMALXE counts."""
    nel = 9
    f = pa.CubeFeti((2, 2, 2), nel, contact=True, young=[1.0 + 0.25 * i for i in range(8)])
    G, e = f.coarse(orthonormalize=True)
    loc = f.subset(range(8))
    nn = nel + 1
    qi = FetiDualQP(ctx, loc, G, e, f.c, f.lb, kplus_rtol=1e-13)
    qo = FetiDualQP(ctx, loc, G, e, f.c, f.lb, kplus_rtol=1e-13, explicit=dict(rtol=1e-13, storage="class_orbit", symmetry=dict(dims=(nn, nn, nn), ndof=3)))
    assert qo.explicit_storage == "class_orbit" and 2 <= qo.explicit_symmetries <= 48
    n_solves, _ = qo.E.assemble_stats()
    assert n_solves < int(qo.E.n_gamma.sum())  # fewer set-up solves than touched dofs: the orbits
    lam = np.random.default_rng(3).standard_normal(f.n_lambda)
    lv, y0, y1 = ctx.vec_from(lam), ctx.vec(f.n_lambda), ctx.vec(f.n_lambda)
    qi.F.mult(lv, y0)
    qo.F.mult(lv, y1)
    assert np.linalg.norm(y1.to_numpy() - y0.to_numpy()) <= 1e-10 * np.linalg.norm(y0.to_numpy())
    si, so = qi.solve_smalxe(rtol=1e-6), qo.solve_smalxe(rtol=1e-6)
    assert si.reason == so.reason == 2
    assert (si.iteration, si.inner_iter_accu, si.inner.ncg, si.inner.nexp) == (so.iteration, so.inner_iter_accu, so.inner.ncg, so.inner.nexp)


def test_explicit_orbit_closed_class_sets(ctx):
    """The same decomposition with the class sets CLOSED under the cube's group (symmetry["close"]: pmh_box_symmetry_closure + pmh_fexplicit_create_shared_orbit_union): every class of one
    block works on the whole boundary of its cube, keeps all 48 operations and needs one K^+ solve per orbit of boundary dofs -- far fewer set-up solves than with the class's own
    faces --, F is the same operator (against the inner-Krylov K^+ and against the un-closed orbit storage), the SMALXE solve takes the same steps."""
    nel = 9
    f = pa.CubeFeti((2, 2, 2), nel, contact=True, young=[1.0 + 0.25 * i for i in range(8)])
    G, e = f.coarse(orthonormalize=True)
    loc = f.subset(range(8))
    nn = nel + 1
    qi = FetiDualQP(ctx, loc, G, e, f.c, f.lb, kplus_rtol=1e-13)
    qo = FetiDualQP(ctx, loc, G, e, f.c, f.lb, kplus_rtol=1e-13, explicit=dict(rtol=1e-13, storage="class_orbit", symmetry=dict(dims=(nn, nn, nn), ndof=3)))
    qc = FetiDualQP(ctx, loc, G, e, f.c, f.lb, kplus_rtol=1e-13, explicit=dict(rtol=1e-13, storage="class_orbit", symmetry=dict(dims=(nn, nn, nn), ndof=3, close=True)))
    assert qc.explicit_storage == "class_orbit" and qc.explicit_symmetries == 48
    n_open, _ = qo.E.assemble_stats()
    n_closed, _ = qc.E.assemble_stats()
    nb_dofs = 3 * (nn ** 3 - (nn - 2) ** 3)  # boundary dofs of a cube: 48 | orbits of them
    assert n_closed < n_open / 3 and n_closed <= 8 * (nb_dofs // 48 + 3 * nn)  # (orbits of points on symmetry planes are shorter than 48)
    lam = np.random.default_rng(3).standard_normal(f.n_lambda)
    lv, y0, y1, y2 = ctx.vec_from(lam), ctx.vec(f.n_lambda), ctx.vec(f.n_lambda), ctx.vec(f.n_lambda)
    qi.F.mult(lv, y0)
    qo.F.mult(lv, y1)
    qc.F.mult(lv, y2)
    n0 = np.linalg.norm(y0.to_numpy())
    assert np.linalg.norm(y2.to_numpy() - y0.to_numpy()) <= 1e-10 * n0 and np.linalg.norm(y2.to_numpy() - y1.to_numpy()) <= 1e-10 * n0
    si, sc = qi.solve_smalxe(rtol=1e-6), qc.solve_smalxe(rtol=1e-6)
    assert si.reason == sc.reason == 2
    assert (si.iteration, si.inner_iter_accu, si.inner.ncg, si.inner.nexp) == (sc.iteration, sc.inner_iter_accu, sc.inner.ncg, sc.inner.nexp)


@pytest.mark.parametrize("block", [0, 5])
def test_explicit_orbit_one_block_per_rank(ctx, block):
    """The reference's own layout -- ONE block per rank (matblockdiag.c:787-788; what the PETSc glue's MatInvAttachExplicitHIP sees): one class of one block.  With the class set closed
    under the cube's group the rank keeps all 48 operations, its multivector records have one slot (8 bytes) and the table-driven GEMM runs on a 64-wide column tile; the rank's
    share of F (its block's B_b W_b B_b') equals the one the inner-Krylov K^+ gives."""
    nel = 8
    f = pa.CubeFeti((2, 2, 2), nel, contact=True)
    G, e = f.coarse(orthonormalize=True)
    loc = f.subset([block])
    nn = nel + 1
    qi = FetiDualQP(ctx, loc, G, e, f.c, f.lb, kplus_rtol=1e-13)
    qc = FetiDualQP(ctx, loc, G, e, f.c, f.lb, kplus_rtol=1e-13, explicit=dict(rtol=1e-13, storage="class_orbit", symmetry=dict(dims=(nn, nn, nn), ndof=3, close=True)))
    assert qc.explicit_storage == "class_orbit" and qc.explicit_symmetries == 48
    n_solves, _ = qc.E.assemble_stats()
    assert n_solves < int(qc.E.n_gamma.sum()) / 10
    rng = np.random.default_rng(8)
    for _ in range(2):
        lam = rng.standard_normal(f.n_lambda)
        lv, y0, y1 = ctx.vec_from(lam), ctx.vec(f.n_lambda), ctx.vec(f.n_lambda)
        qi.F.mult(lv, y0)
        qc.F.mult(lv, y1)
        assert np.linalg.norm(y1.to_numpy() - y0.to_numpy()) <= 1e-10 * np.linalg.norm(y0.to_numpy())


def test_explicit_orbit_mixed_class_sizes(ctx):
    """Classes of 3, 2, 2 and 1 congruent blocks in one decomposition (materials 1, 1, 1, 2, 2, 3, 3, 4): multivector records of 4, 2, 2 and 1 slots, column tiles of 128 and 64,
    hence table-driven GEMM launches per class (no single merged launch) -- F against the inner-Krylov K^+ and the same SMALXE steps."""
    nel = 8
    f = pa.CubeFeti((2, 2, 2), nel, contact=True, young=[1.0, 1.0, 1.0, 2.0, 2.0, 3.0, 3.0, 4.0])
    G, e = f.coarse(orthonormalize=True)
    loc = f.subset(range(8))
    cls = pa.csr_block_classes(loc["block_rowstart"], loc["K"])
    assert sorted(np.bincount(cls).tolist()) == [1, 2, 2, 3]
    nn = nel + 1
    qi = FetiDualQP(ctx, loc, G, e, f.c, f.lb, kplus_rtol=1e-13)
    qc = FetiDualQP(ctx, loc, G, e, f.c, f.lb, kplus_rtol=1e-13, explicit=dict(rtol=1e-13, storage="class_orbit", symmetry=dict(dims=(nn, nn, nn), ndof=3, close=True)))
    assert qc.explicit_storage == "class_orbit" and qc.explicit_symmetries == 48
    rng = np.random.default_rng(4)
    for _ in range(2):
        lam = rng.standard_normal(f.n_lambda)
        lv, y0, y1 = ctx.vec_from(lam), ctx.vec(f.n_lambda), ctx.vec(f.n_lambda)
        qi.F.mult(lv, y0)
        qc.F.mult(lv, y1)
        assert np.linalg.norm(y1.to_numpy() - y0.to_numpy()) <= 1e-10 * np.linalg.norm(y0.to_numpy())
    si, sc = qi.solve_smalxe(rtol=1e-6), qc.solve_smalxe(rtol=1e-6)
    assert (si.reason, si.iteration, si.inner_iter_accu, si.inner.ncg, si.inner.nexp) == (sc.reason, sc.iteration, sc.inner_iter_accu, sc.inner.ncg, sc.inner.nexp)

