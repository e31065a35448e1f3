"""Multi-right-hand-side K^+ (csrc/mv.hip ...): the operator product on interleaved multivectors against scipy, column by column."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

import permon_amd as pa
from permon_amd._lib import check

pytestmark = pytest.mark.gpu
R = 8


@pytest.fixture(scope="module")
def ctx():
    c = pa.Context(0)
    yield c
    c.close()


def _block_matrix(rng, nn, extra):
    """a matrix of full 3 x 3 blocks on nn nodes: chain neighbours + `extra` random block pairs, ragged rows (1 ... many blocks)"""
    pairs = {(i, i) for i in range(nn)} | {(i, i + 1) for i in range(nn - 1)} | {(i + 1, i) for i in range(nn - 1)}
    for _ in range(extra):
        a, b = int(rng.integers(0, nn)), int(rng.integers(0, nn))
        pairs.add((a, b))
    rows, cols, vals = [], [], []
    for (a, b) in pairs:
        blk = rng.standard_normal((3, 3))
        keep = rng.random((3, 3)) < 0.8  # incomplete blocks: an entry a row does not store is a zero of the block
        keep[rng.integers(0, 3), rng.integers(0, 3)] = True
        for q in range(3):
            for c in range(3):
                if keep[q, c]:
                    rows.append(3 * a + q), cols.append(3 * b + c), vals.append(blk[q, c])
    A = sp.csr_matrix((vals, (rows, cols)), shape=(3 * nn, 3 * nn))
    A.sort_indices()
    return A


@pytest.mark.parametrize("nn,extra", [(1, 0), (7, 5), (300, 900), (5000, 20000)])
@pytest.mark.parametrize("storage", [0, 1, 2])
def test_mv_product_against_scipy(ctx, nn, extra, storage):
    rng = np.random.default_rng(nn + storage)
    A = _block_matrix(rng, nn, extra)
    n = A.shape[0]
    X = rng.standard_normal((n, R))
    Ad = pa.CsrMat(ctx, n, n, A.indptr, A.indices, A.data)
    xd, yd = ctx.vec_from(X.reshape(-1)), ctx.vec(n * R)
    ms = C.c_float(0)
    check(ctx.L.pmh_mv_test_spmv(Ad.h, storage, xd.p, yd.p, 1, C.byref(ms)))
    Y = yd.to_numpy().reshape(n, R)
    ref = A @ X
    tol = {0: 1e-14, 1: 2e-6, 2: 3e-3}[storage]
    scale = np.abs(A).dot(np.abs(X)).max()
    assert np.abs(Y - ref).max() <= tol * scale, (nn, storage, np.abs(Y - ref).max() / scale)


def test_mv_wide_rows_are_refused(ctx):
    n = 3 * 2050
    A = sp.csr_matrix(np.ones((n, n)))  # 2050 blocks in a block row: more than the 2048 slots (32 until round 6: an aggregation hierarchy's coarse rows hold 50 ... 600)
    Ad = pa.CsrMat(ctx, n, n, A.indptr, A.indices, A.data)
    xd, yd = ctx.vec(n * R), ctx.vec(n * R)
    assert ctx.L.pmh_mv_test_spmv(Ad.h, 0, xd.p, yd.p, 1, None) != 0


@pytest.mark.parametrize("storage", [0, 1, 2])
def test_mv_long_rows_take_16_lanes_per_block_row(ctx, storage):
    """Block rows of more than 48 blocks (the coarse operators of an aggregation hierarchy) run with 16 lanes per block row (k_mv_spmv<..., 16>): same product."""
    rng = np.random.default_rng(40 + storage)
    nb = 90
    M = rng.standard_normal((3 * nb, 3 * nb)) * (rng.random((3 * nb, 3 * nb)) < 0.8)
    keep = np.kron((rng.random((nb, nb)) < 0.75).astype(float), np.ones((3, 3)))
    A = sp.csr_matrix(M * keep + np.eye(3 * nb))
    A.sort_indices()
    n = A.shape[0]
    X = rng.standard_normal((n, R))
    Ad = pa.CsrMat(ctx, n, n, A.indptr, A.indices, A.data)
    xd, yd = ctx.vec_from(X.reshape(-1)), ctx.vec(n * R)
    check(ctx.L.pmh_mv_test_spmv(Ad.h, storage, xd.p, yd.p, 1, None))
    ref = A @ X
    tol = {0: 1e-13, 1: 2e-6, 2: 3e-3}[storage]
    scale = np.abs(A).dot(np.abs(X)).max()
    assert np.abs(yd.to_numpy().reshape(n, R) - ref).max() <= tol * scale


@pytest.mark.parametrize("case", ["floating", "regularized", "odd_box", "jacobi", "mfma_coarse"])
def test_multi_rhs_kplus_against_the_one_column_solver(ctx, case):
    """U = K^+ F for 8 columns per block (matinv_mv.hip: interleaved multivectors, the V-cycle of mg_mv.hip) against pmh_matinv_mult column by column and against the dense
    pseudo-inverse: floating blocks (K^+ = P_R K^- P_R), regularised blocks (no kernel), a box with a short last coarse interval, and Jacobi-CG without a hierarchy; two
    DIFFERENT blocks (the second one twice as stiff) so that nothing leans on congruence.  Columns of very different size, one of them zero, one in the kernel.
    "mfma_coarse": 15^3-node blocks coarsened once, TWO dense fp16 coarse blocks of 1 536 rows each -- k_mvg_coarse_mfma with more than one block (a workgroup finds its block by its
    first row); against the one-column solver only (the dense pseudo-inverse of 10 125 dofs is not formed)."""
    from permon_amd.feti import CubeFeti
    import scipy.sparse as sp

    nel = {"odd_box": 9, "mfma_coarse": 14}.get(case, 6)
    f = CubeFeti((2, 1, 1), nel, "elasticity", contact=False)
    nn, n_i, N = f.nel + 1, f.n_i, f.N
    Ksp = sp.block_diag([f.K[:n_i, :n_i], 2.0 * f.K[n_i:, n_i:]]).tocsr()
    R = f.R
    if case == "regularized":
        Ksp = (Ksp + sp.diags(np.full(N, 1e-2 * abs(Ksp.diagonal()).max()))).tocsr()
        R = None
    Ksp.sort_indices()
    K = pa.MatBlockDiag.from_scipy(ctx, f.block_rowstart, Ksp)
    M = pa.MatInv(K, rtol=1e-11, nullspace=R)
    if case != "jacobi":
        M.set_pc_mg_box(Ksp, [(nn, nn, nn)] * f.nsub, 3, R=R, min_nodes=512 if case == "mfma_coarse" else 27, degree=2, precision="fp16" if case == "mfma_coarse" else "fp32")
    rng = np.random.default_rng(5)
    F = rng.standard_normal((N, 8)) * (10.0 ** rng.integers(-3, 4, size=8))
    F[:, 3] = 0.0
    if R is not None:
        F[:n_i, 5] = R[0, :n_i]  # block 0, column 5: a load in the kernel -> u = 0 there
    Fd, Ud = ctx.vec_from(F.reshape(-1)), ctx.vec(N * 8)
    its = M.mult_multi(Fd, Ud)
    U = Ud.to_numpy().reshape(N, 8)
    u1 = ctx.vec(N)
    worst = 0
    for r in range(8):
        M.mult(ctx.vec_from(F[:, r].copy()), u1)
        ref = u1.to_numpy()
        worst = max(worst, M.last_iterations()[0])
        assert np.linalg.norm(U[:, r] - ref) <= 1e-8 * max(np.linalg.norm(ref), 1e-300) + (0 if np.linalg.norm(ref) else 1e-300), (case, r)
    assert abs(its - worst) <= 2, (its, worst)
    assert np.all(U[:, 3] == 0.0)
    if R is not None:
        assert np.linalg.norm(U[:n_i, 5]) <= 1e-10
        for b in range(2 if case != "mfma_coarse" else 0):
            Kd = Ksp[b * n_i:(b + 1) * n_i, b * n_i:(b + 1) * n_i].toarray()
            ref = np.linalg.pinv(Kd, rcond=1e-10, hermitian=True) @ F[b * n_i:(b + 1) * n_i]
            for r in (0, 1, 2, 4, 6, 7):
                assert np.linalg.norm(U[b * n_i:(b + 1) * n_i, r] - ref[:, r]) <= 1e-8 * np.linalg.norm(ref[:, r]), (case, b, r)


@pytest.mark.parametrize("prec", ["fp32", "fp16"])
def test_eight_congruent_blocks_run_as_eight_columns(ctx, prec):
    """pmh_matinv_mult on 8 CONGRUENT floating blocks (the cubes of a 2 x 2 x 2 decomposition) with the box V-cycle: the 8 blocks are the 8 columns of ONE block on the
    multi-right-hand-side kernels (knob kplus_mv, the default) -- against the one-column block CG on 8 replicas (knob 0): the same K^+ to 1e-8 per block, iteration counts
    within 2; and against the dense pseudo-inverse.  The first call after a change of the solver (here: tolerances stay, the knob flips) re-decides the path."""
    from permon_amd.feti import CubeFeti

    f = CubeFeti((2, 2, 2), 6, "elasticity", contact=False)
    nn, n_i, N = f.nel + 1, f.n_i, f.N
    Ksp = f.K
    rhs = np.random.default_rng(3).standard_normal(N) * np.repeat(10.0 ** np.arange(-3, 5), n_i)
    out = []
    try:
        for knob in (1, 0):
            check(ctx.L.pmh_set_knob(b"kplus_mv", knob))
            K = pa.MatBlockDiag.from_scipy(ctx, f.block_rowstart, Ksp)
            M = pa.MatInv(K, rtol=1e-11, nullspace=f.R)
            M.enable_bsr3()
            M.set_pc_mg_box(Ksp, [(nn, nn, nn)] * f.nsub, 3, R=f.R, min_nodes=27, degree=2, precision=prec)
            u = ctx.vec(N)
            M.mult(ctx.vec_from(rhs), u)
            M.mult(ctx.vec_from(rhs), u)  # twice: the second application starts from the first one's iteration estimate
            out.append((u.to_numpy(), M.last_iterations()[0]))
    finally:
        ctx.L.pmh_set_knob(b"kplus_mv", 1)
    (ua, ia), (ub, ib) = out
    for b in range(8):
        sl = slice(b * n_i, (b + 1) * n_i)
        assert np.linalg.norm(ua[sl] - ub[sl]) <= 1e-8 * np.linalg.norm(ub[sl]), b
    assert abs(ia - ib) <= 2 and ia < 30
    Kp = np.linalg.pinv(Ksp[:n_i, :n_i].toarray(), rcond=1e-10, hermitian=True)
    for b in (0, 7):
        sl = slice(b * n_i, (b + 1) * n_i)
        ref = Kp @ rhs[sl]
        assert np.linalg.norm(ua[sl] - ref) <= 1e-8 * np.linalg.norm(ref), b


@pytest.mark.parametrize("nel", [14, 22])
def test_large_dense_coarse_block_is_solved_on_the_matrix_cores(ctx, nel):
    """The one-block regime of bench.py's inner-Krylov pass in small: 15^3- (23^3-) node cubes coarsened ONCE, dense coarse blocks of 8^3 (12^3) nodes = 1 536 (5 184, bench.py's)
    rows in fp16 -- multiples of 16 and >= 512, so the 8-column cycle takes k_mvg_coarse_mfma (v_mfma_f32_16x16x4_f32, 16 rows per workgroup, the k range split over its 8
    wavefronts: 48 steps of 32 k's = 6 per wavefront, whole pipeline stages; 162 steps = 21 per wavefront, 15 for the last one: the guarded tails) -- against the one-column
    cycle's k_mg_coarse on 8 replicas: the same K^+ to 1e-8, iteration counts within 2."""
    from permon_amd.feti import CubeFeti

    f = CubeFeti((2, 2, 2), nel, "elasticity", contact=False)
    nn, n_i, N = f.nel + 1, f.n_i, f.N
    Ksp = f.K
    rhs = np.random.default_rng(5).standard_normal(N) * np.repeat(10.0 ** np.arange(-3, 5), n_i)
    out = []
    try:
        for knob in (1, 0):
            check(ctx.L.pmh_set_knob(b"kplus_mv", knob))
            K = pa.MatBlockDiag.from_scipy(ctx, f.block_rowstart, Ksp)
            M = pa.MatInv(K, rtol=1e-11, nullspace=f.R)
            M.enable_bsr3()
            M.set_pc_mg_box(Ksp, [(nn, nn, nn)] * f.nsub, 3, R=f.R, min_nodes=512, degree=2, precision="fp16")
            u = ctx.vec(N)
            M.mult(ctx.vec_from(rhs), u)
            out.append((u.to_numpy(), M.last_iterations()[0]))
    finally:
        ctx.L.pmh_set_knob(b"kplus_mv", 1)
    (ua, ia), (ub, ib) = out
    for b in range(8):
        sl = slice(b * n_i, (b + 1) * n_i)
        assert np.linalg.norm(ua[sl] - ub[sl]) <= 1e-8 * np.linalg.norm(ub[sl]), b
    assert abs(ia - ib) <= 2 and ia < 30, (ia, ib)
    # K K^+ f = P_R f: the solution is one, whatever the path
    r = Ksp @ ua - rhs
    Rb = f.R[:, :n_i]
    G = np.linalg.inv(Rb @ Rb.T)
    for b in range(8):
        sl = slice(b * n_i, (b + 1) * n_i)
        rb = r[sl] + Rb.T @ (G @ (Rb @ rhs[sl]))
        assert np.linalg.norm(rb) <= 1e-9 * np.linalg.norm(rhs[sl]), b


def test_multi_rhs_kplus_on_blocks_of_different_sizes(ctx):
    """ex71's elasticity slabs (DmdaFeti): 7 blocks of two different sizes, one of them non-singular (Dirichlet in the matrix), thin in x, a hierarchy built by the Python
    builder and handed in (pmh_mg_create with node-wise P) -- the 8-column solver against the one-column solver on every column, and the explicit operators assembled
    8 columns at a time against the same assembly one column at a time."""
    from permon_amd.feti import DmdaFeti, box_mg_hierarchy

    prob = DmdaFeti((8, 6, 4), 7, "elasticity")
    dims = [(2, 7, 5), (3, 7, 5)] + [(2, 7, 5)] * 5
    H = box_mg_hierarchy(prob.blocks, dims, 3, min_nodes=27)
    K = pa.MatBlockDiag.from_scipy(ctx, prob.block_rowstart, prob.K)
    M = pa.MatInv(K, rtol=1e-12, nullspace=prob.R)
    M.enable_bsr3()
    M.set_pc_mg(H, degree=2, precision="fp32")
    N = prob.N
    F = np.random.default_rng(21).standard_normal((N, 8))
    Ud = ctx.vec(N * 8)
    its = M.mult_multi(ctx.vec_from(F.reshape(-1)), Ud)
    U = Ud.to_numpy().reshape(N, 8)
    u1 = ctx.vec(N)
    for r in range(8):
        M.mult(ctx.vec_from(F[:, r].copy()), u1)
        ref = u1.to_numpy()
        assert np.linalg.norm(U[:, r] - ref) <= 1e-8 * np.linalg.norm(ref), r
    assert its < 40


def test_multi_rhs_refusals_are_loud(ctx):
    """Where the 8-column solver does not apply the library says so (PMH_ERR_SUP with the reason) instead of computing something else: the left generalised inverse, a
    fp64 V-cycle; and a slot count that is neither one nor eight per block is an argument error of the assembly."""
    from permon_amd.feti import CubeFeti
    from permon_amd._lib import PermonHipError

    f = CubeFeti((2, 1, 1), 4, "elasticity", contact=True)
    nn, N = f.nel + 1, f.N
    K = pa.MatBlockDiag.from_scipy(ctx, f.block_rowstart, f.K)
    Fd, Ud = ctx.vec(N * 8), ctx.vec(N * 8)
    M = pa.MatInv(K, rtol=1e-10, nullspace=f.R)
    M.set_pc_mg_box(f.K, [(nn, nn, nn)] * f.nsub, 3, R=f.R, min_nodes=27, degree=2, precision="fp64")
    with pytest.raises(PermonHipError) as ei:
        M.mult_multi(Fd, Ud)
    assert ei.value.code == 4 and "fp64" in str(ei.value)
    M2 = pa.MatInv(K, rtol=1e-10, nullspace=f.R)
    check(ctx.L.pmh_matinv_set_left_inverse(M2.h, 1, np.zeros(1, dtype=np.int32).ctypes.data_as(C.c_void_p)))
    with pytest.raises(PermonHipError) as ei:
        M2.mult_multi(Fd, Ud)
    assert ei.value.code == 4 and "left" in str(ei.value)
    # the assembly: 3 slots for a solver of 2 blocks
    G, e = f.coarse()
    from permon_amd.chain import FetiDualQP

    q = FetiDualQP(ctx, f.subset(range(2)), G, e, f.c, f.lb, kplus_rtol=1e-10)
    E = q._create_explicit(f.subset(range(2)), "sym", None, np.arange(2, dtype=np.int32), 2)
    sc = np.zeros(3, dtype=np.int32)
    rc = ctx.L.pmh_fexplicit_assemble(E.h, q.Kplus.h, 3, sc.ctypes.data_as(C.c_void_p), np.arange(2, dtype=np.int32).ctypes.data_as(C.c_void_p), 1e-10, 0)
    assert rc == 2  # PMH_ERR_ARG
    E.destroy()
