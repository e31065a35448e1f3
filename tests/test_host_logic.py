"""CPU tests of the host-side producers (problem generators, FETI set-up data) and of the oracle's FETI pieces."""
import numpy as np
import pytest
import scipy.sparse as sp

from permon_amd import problems as P
from permon_amd.feti import CubeFeti, q1_elasticity_element, q1_poisson_element


def test_ex1_matrix_shape_matches_tutorial():
    p = P.ex1(1000)  # BASELINE.json configs[0]
    assert p["col"].size == 2994  # nnz quoted in SURVEY section 8 (C1)
    A = sp.csr_matrix((p["val"], p["col"], p["rowptr"]), shape=(1000, 1000))
    assert abs(A - A.T).max() == 0
    assert A[0, 0] == 1 and A[0, 1] == 0 and A[1, 0] == 0 and A[1, 1] == 2
    assert p["lb"][0] == 0 and p["lb"][-1] == 0 and p["b"][0] == 0


def test_laplace2d_config1_counts():
    rp, ci, va = P.laplace2d_csr(50, 40)
    n = 2000
    assert rp[-1] == 5 * n - 2 * 50 - 2 * 40
    A = sp.csr_matrix((va, ci, rp), shape=(n, n))
    assert abs(A - A.T).max() == 0 and np.all(np.diff(ci[rp[7]:rp[8]]) > 0)
    # configs[1]: 3162^2 -> n = 9 998 244, nnz = 49 978 572 (formula, without building it)
    g = 3162
    assert g * g == 9998244 and 5 * g * g - 4 * g == 49978572


def test_q1_elements():
    Ke = q1_elasticity_element(0.5)
    assert np.allclose(Ke, Ke.T) and Ke.shape == (24, 24)
    ev = np.linalg.eigvalsh(Ke)
    assert np.sum(np.abs(ev) < 1e-12) == 6 and ev.min() > -1e-12  # six rigid-body modes
    Kp = q1_poisson_element(0.25)
    assert np.allclose(Kp.sum(axis=1), 0) and np.linalg.eigvalsh(Kp).min() > -1e-14


@pytest.mark.parametrize("physics,kdim", [("elasticity", 6), ("poisson", 1)])
def test_cube_feti_invariants(physics, kdim):
    f = CubeFeti((2, 2, 1), 2, physics=physics, gluing="full")
    assert f.R.shape == (kdim, f.N)
    assert abs(f.K @ f.R.T).max() < 1e-13  # R spans ker K block-wise
    for s in range(f.nsub):
        Rs = f.R[:, s * f.n_i:(s + 1) * f.n_i]
        assert np.allclose(Rs @ Rs.T, np.eye(kdim))
    # gluing rows: two entries +-1/sqrt(multiplicity); Dirichlet / contact rows: one entry
    B = f.B.tocsr()
    nnz_row = np.diff(B.indptr)
    assert np.all(nnz_row[:f.n_dirichlet] == 1) and np.all(nnz_row[f.n_dirichlet:f.n_eq] == 2) and np.all(nnz_row[f.n_eq:] == 1)
    assert np.allclose(B[f.n_dirichlet:f.n_eq].sum(axis=1), 0)
    G, e = f.coarse(orthonormalize=True)
    assert np.allclose((G @ G.T).toarray(), np.eye(G.shape[0]), atol=1e-12)
    # subsets partition the leaves
    a, b = f.subset([0, 1]), f.subset([2, 3])
    assert len(a["leaves_row"]) + len(b["leaves_row"]) == len(f.leaves_row)
    assert f.lb[:f.n_eq].max() == -np.inf and np.all(f.lb[f.n_eq:] == 0)


def test_config2_sizes_formula():
    """BASELINE.json configs[2] at full size without assembling: 2x2x2 cubes of 44^3 nodes x 3 dof."""
    nn1 = 44
    assert nn1 ** 3 * 3 == 255552 and 8 * 255552 == 2044416
    f = CubeFeti((2, 2, 2), 3)
    # structure check of the row count formulas on a small instance: Dirichlet = dofs on x=0 of the 4 left cubes
    assert f.n_dirichlet == 4 * (3 + 1) ** 2 * 3


def test_oracle_matinv_and_feti_operator(oracle):
    f = CubeFeti((2, 1, 1), 2)
    K = oracle.Csr.from_scipy(f.K)
    M = oracle.MatInv(K, f.block_rowstart, f.R, rtol=1e-13)
    rhs = np.random.default_rng(0).standard_normal(f.N)
    Kp = np.linalg.pinv(f.K.toarray(), rcond=1e-12, hermitian=True)
    assert np.linalg.norm(M.mult(rhs) - Kp @ rhs) <= 1e-10 * np.linalg.norm(Kp @ rhs)
    G, e = f.coarse()
    pf = oracle.Qppf(oracle.Csr.from_scipy(G), orthonormal=True)
    B = oracle.Gluing(f.N, f.n_lambda, f.leaves_row, f.leaves_root, f.leaves_sign)
    Fo = oracle.FetiOp(B, M, pf, rho=0.7, which=1)
    x = np.random.default_rng(1).standard_normal(f.n_lambda)
    Bd = f.B.toarray()
    Fd = Bd @ Kp @ Bd.T
    ref = pf.P(Fd @ pf.P(x)) + 0.7 * pf.Q(x)
    assert np.linalg.norm(Fo.op(x) - ref) <= 1e-10 * np.linalg.norm(ref)


def test_gluing_from_l2g_matches_the_link_rules():
    """pmh_feti_gluing_from_l2g (C++, QPFetiGetBgtSF) against an independent statement of the same rules (gluing_links applied node
    by node in ascending global dof): random partitions with dofs shared by up to 6 subdomains, all three gluing types, with and
    without -SCALE_ON and with excluded (Dirichlet) dofs.  Index bookkeeping: exact."""
    import numpy as np

    from permon_amd.feti import gluing_from_l2g, gluing_links

    rng = np.random.default_rng(5)
    for trial in range(6):
        nsub, nglob = int(rng.integers(2, 7)), int(rng.integers(20, 60))
        l2g = []
        for s in range(nsub):
            g = np.flatnonzero(rng.random(nglob) < 0.55)
            rng.shuffle(g)  # local numbering is not monotone in the global one (as for a DMDA's ghosted numbering)
            l2g.append(g.astype(np.int32))
        start = np.concatenate([[0], np.cumsum([len(g) for g in l2g])])
        exclude = rng.choice(nglob, 4, replace=False) if trial % 2 else None
        for gtype in ("nonred", "full", "orth"):
            for scale in (True, False):
                rows, roots, vals, nl = gluing_from_l2g(l2g, gtype, scale, exclude)
                exp_rows, exp_roots, exp_vals, link = [], [], [], 0
                for gg in range(nglob):
                    if exclude is not None and gg in exclude:
                        continue
                    cp = [(s, int(np.flatnonzero(l2g[s] == gg)[0])) for s in range(nsub) if gg in l2g[s]]
                    for lk in gluing_links(len(cp), gtype, scale):
                        for t, v in lk:
                            exp_rows.append(start[cp[t][0]] + cp[t][1]), exp_roots.append(link), exp_vals.append(v)
                        link += 1
                assert nl == link and rows.tolist() == exp_rows and roots.tolist() == exp_roots
                assert np.array_equal(vals, np.asarray(exp_vals))
    # a dof listed twice in one subdomain is an input error, reported
    import pytest

    from permon_amd import PermonHipError

    with pytest.raises(PermonHipError):
        gluing_from_l2g([np.array([0, 1, 1], dtype=np.int32), np.array([1, 2], dtype=np.int32)], "full")
    with pytest.raises(ValueError):
        gluing_from_l2g([np.array([0, 1], dtype=np.int32)], "bogus")


def test_matis_rhs_split_and_solution_assembly():
    """QPTMatISToBlockDiag's vector part (qptransform.c:2095-2113, post-solve :1945-1949) as host routines of the C ABI."""
    import ctypes as C

    import numpy as np

    from permon_amd import _lib

    L = _lib.load()
    l2g = np.array([0, 1, 2, 2, 3, 1, 4, 2], dtype=np.int32)  # dof 1 twice, dof 2 three times
    b = np.array([1.0, 4.0, 9.0, 5.0, 7.0])
    f = np.zeros(l2g.size)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    _lib.check(L.pmh_qpt_matis_split_rhs(l2g.size, p(l2g), b.size, p(b), p(f)))
    assert f.tolist() == [1.0, 2.0, 3.0, 3.0, 5.0, 2.0, 7.0, 3.0]
    assert np.allclose(np.bincount(l2g, weights=f, minlength=5), b)  # the copies sum up to the assembled load
    u = np.arange(10.0, 18.0)
    x = np.zeros(5)
    _lib.check(L.pmh_qpt_matis_assemble_solution(l2g.size, p(l2g), p(u), x.size, p(x)))
    assert x.tolist() == [10.0, 15.0, 17.0, 14.0, 16.0]  # INSERT_VALUES: the last copy wins, nothing is averaged
    assert L.pmh_qpt_matis_split_rhs(l2g.size, p(l2g), 3, p(b), p(f)) != 0  # l2g out of range: reported


def test_host_kplus_mg_restatement_vs_pinv():
    """oracle/mg_host.py (the CPU baseline's K^+: block-wise V-cycle-preconditioned CG, Moore-Penrose wrapped) against numpy pinv."""
    import permon_amd as pa
    from oracle.mg_host import KplusMG

    f = pa.CubeFeti((2, 1, 1), 4, contact=False)
    nn = f.nel + 1
    H = pa.box_mg_hierarchy([f.Ki] * 2, [(nn, nn, nn)] * 2, 3, min_nodes=27)
    Kp = KplusMG(f.K, f.block_rowstart, H, R=f.R, rtol=1e-12)
    rhs = np.random.default_rng(2).standard_normal(f.N)
    u = Kp(rhs)
    ref1 = np.linalg.pinv(f.Ki.toarray(), rcond=1e-10, hermitian=True)
    ref = np.concatenate([ref1 @ rhs[:f.n_i], ref1 @ rhs[f.n_i:]])
    assert np.linalg.norm(u - ref) <= 1e-9 * np.linalg.norm(ref)
    assert 0 < Kp.last_its < 40


def test_matis_to_blockdiag_matrix_side():
    """QPTMatISToBlockDiag, matrix side (qptransform.c:2007-2150): local matrices + l2g -> block-diagonal CSR, matis->counter,
    interface flags and the sorted i2g, against scipy / numpy on the ex71 decomposition (host routine: no GPU needed)."""
    import ctypes as C

    import scipy.sparse as sp

    import permon_amd as pa

    L = pa.load()
    f = pa.DmdaFeti(cells=(5, 4, 3), size=4, physics="elasticity")
    blocks = [b.tocsr() for b in f.blocks]
    for b in blocks:
        b.sort_indices()
    nd = f.ndof
    l2g = [(np.asarray(g)[:, None] * nd + np.arange(nd)[None, :]).ravel() for g in f.gids]
    start = np.concatenate([[0], np.cumsum([len(g) for g in l2g])]).astype(np.int32)
    cat = np.concatenate(l2g).astype(np.int32)
    ng = int(cat.max()) + 1
    lrp = np.concatenate([b.indptr for b in blocks]).astype(np.int32)
    lci = np.concatenate([b.indices for b in blocks]).astype(np.int32)
    lva = np.concatenate([b.data for b in blocks])
    N, nnz = int(start[-1]), lva.size
    brs, rp, ci, va = np.zeros(len(blocks) + 1, np.int32), np.zeros(N + 1, np.int32), np.zeros(nnz, np.int32), np.zeros(nnz)
    cnt, isb, i2g, n_i2g = np.zeros(N, np.int32), np.zeros(N, np.int32), np.zeros(ng, np.int32), C.c_int()
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    pa._lib.check(L.pmh_qpt_matis_to_blockdiag(len(blocks), p(start), p(cat), ng, p(lrp), p(lci), p(lva), p(brs), p(rp), p(ci), p(va), p(cnt), p(isb), C.byref(n_i2g), p(i2g)))
    K = sp.csr_matrix((va, ci, rp), shape=(N, N))
    ref = sp.block_diag(blocks, format="csr")
    assert (K != ref).nnz == 0 and np.array_equal(brs, start)
    mult = np.bincount(cat, minlength=ng)
    assert np.array_equal(cnt, mult[cat]) and np.array_equal(isb, (mult[cat] > 1).astype(np.int32))
    assert np.array_equal(i2g[:n_i2g.value], np.nonzero(mult > 1)[0])
    # the vector part uses the same counter: b_local = b_global / counter
    bg = np.random.default_rng(0).standard_normal(ng)
    fl = np.zeros(N)
    pa._lib.check(L.pmh_qpt_matis_split_rhs(N, p(cat), ng, p(bg), p(fl)))
    assert np.allclose(fl, bg[cat] / cnt)


def test_stripe_plan_balances_the_dense_bytes():
    """pmh_fexplicit_set_stripe's dealing rule on the n_Gamma of configs[2] (host helper, no GPU): one block per GPU would leave a 1.36 x
    imbalance of the dense bytes (the slowest rank sets the step time); dealt in 128-row stripes every rank gets the mean within 1 %."""
    import ctypes as C

    import permon_amd as pa

    L = pa.load()
    ng = np.array([24384, 18880, 24384, 18880, 22578, 17031, 22578, 17031], dtype=np.int32)
    one_per_gpu = (ng.astype(float) ** 2)
    assert one_per_gpu.max() / one_per_gpu.mean() > 1.3
    for size in (1, 2, 4, 8):
        b = np.zeros(size)
        pa._lib.check(L.pmh_fexplicit_stripe_bytes(ng.size, ng.ctypes.data_as(C.c_void_p), size, b.ctypes.data_as(pa._lib.c_double_p)))
        assert b.min() > 0 and b.max() / b.mean() < 1.01
        tot = b.sum()
    assert abs(tot - 4.0 * (np.ceil(ng / 128) * 128).astype(float).dot((np.ceil(ng / 128) * 128)) ) / tot < 0.02  # ~ 4 n^2 bytes per block


def test_class_sym_plan_balances_the_tile_bytes():
    """PMH_FX_CLASS_SYM at several GPUs (host helper, no GPU): whole mega bands of 1024 rows of the lower block-triangle, dealt from the longest
    down in snake order -- every mega band has one owner and the ranks' tile bytes stay within 3 % of the mean for configs[2] (n_c = 33 288)."""
    import ctypes as C

    import permon_amd as pa

    L = pa.load()
    n_c = 33288
    nmb = -(-(-(-n_c // 256)) // 4)
    for size in (1, 2, 4, 8):
        own, b = np.full(nmb, -1, dtype=np.int32), np.zeros(size)
        pa._lib.check(L.pmh_fexplicit_class_sym_plan(n_c, size, own.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)))
        assert own.min() >= 0 and own.max() == size - 1 and set(own.tolist()) == set(range(size))
        assert b.max() / b.mean() < 1.03
        nsb = -(-n_c // 256)
        assert abs(b.sum() - 8.0 * 256 * 256 * nsb * (nsb + 1) / 2) < 1.0  # the lower block-triangle in 256-row super bands


def test_orbit_row_tile_rule(monkeypatch):
    """PMH_FX_CLASS_ORBIT (host helper, no GPU): the row tile of the orbit GEMM.  Default kernel (k_fxo_gemm16, v_mfma_f64_16x16x4: 16 rows per instruction tile): the tile among
    144, 128, 112, 96, 80 that pads the representatives' rows least (ties: the larger): configs[2]'s 715 -> 5 x 144 = 720."""
    import ctypes as C

    import permon_amd as pa

    L = pa.load()
    monkeypatch.delenv("PMH_FXO_TM", raising=False)

    def rule(M):
        tm, Mp = C.c_int(), C.c_int()
        pa._lib.check(L.pmh_fexplicit_orbit_row_tile(M, C.byref(tm), C.byref(Mp)))
        return tm.value, Mp.value

    assert rule(715) == (144, 720) and rule(144) == (144, 144) and rule(128) == (128, 128) and rule(1024) == (128, 1024)
    assert rule(176) == (96, 192) or rule(176) == (80, 240) or rule(176)[1] == 192  # the configs[3] shape: 176 -> 192 (2 x 96)
    assert rule(176) == (96, 192)
    for M in range(1, 2000, 7):
        tm, Mp = rule(M)
        assert tm in (144, 128, 112, 96, 80) and Mp % tm == 0 and M <= Mp < M + tm
        assert Mp <= min(-(-M // t) * t for t in (144, 128, 112, 96, 80))
    monkeypatch.setenv("PMH_FXO_TM", "112")
    assert rule(715) == (112, 784)
    monkeypatch.delenv("PMH_FXO_TM")
    assert rule(715) == (144, 720)


def test_box_symmetries_c_vs_numpy():
    """pmh_box_symmetries (host C++): the group of signed dof permutations of a box block that leave K invariant -- the same operations as the numpy
    restatement feti.box_symmetries (48 for a cube of Q1 elasticity elements, 8 / 16 for boxes with unequal sides), identity first, and K^+ is
    invariant under every one of them (what the set-up by symmetry relies on)."""
    import ctypes as C

    import permon_amd as pa
    from permon_amd.feti import CubeFeti, box_symmetries

    L = pa.load()
    f = CubeFeti((2, 1, 1), 3, contact=True)
    K = f.Ki.tocsr()
    K.sort_indices()
    n = K.shape[0]

    def c_group(dims, ndof, K=None, n=None):
        d = np.array(dims, dtype=np.int32)
        perm, sign, ns = np.zeros(48 * n, dtype=np.int32), np.zeros(48 * n, dtype=np.int8), C.c_int()
        args = [None, None, None] if K is None else [np.ascontiguousarray(K.indptr, dtype=np.int32), np.ascontiguousarray(K.indices, dtype=np.int32), np.ascontiguousarray(K.data)]
        pa._lib.check(L.pmh_box_symmetries(d.ctypes.data_as(C.c_void_p), ndof, *[a.ctypes.data_as(C.c_void_p) if a is not None else None for a in args], 4000, C.byref(ns),
                                           perm.ctypes.data_as(C.c_void_p), sign.ctypes.data_as(C.c_void_p)))
        return perm.reshape(48, n)[:ns.value], sign.reshape(48, n)[:ns.value]

    pc, sc = c_group((4, 4, 4), 3, K, n)
    pn, sn = box_symmetries((4, 4, 4), 3, K=K)
    assert pc.shape[0] == 48 and np.array_equal(pc[0], np.arange(n)) and sc[0].min() == 1
    assert {(pc[g].tobytes(), sc[g].tobytes()) for g in range(48)} == {(pn[g].astype(np.int32).tobytes(), sn[g].tobytes()) for g in range(48)}
    Kp = np.linalg.pinv(K.toarray(), rcond=1e-10, hermitian=True)
    for g in range(48):
        W = np.empty_like(Kp)
        s = sc[g].astype(float)
        W[np.ix_(pc[g], pc[g])] = Kp * s[:, None] * s[None, :]
        assert np.abs(W - Kp).max() <= 1e-12 * np.abs(Kp).max()
    assert c_group((3, 4, 5), 3, n=180)[0].shape[0] == 8 and c_group((4, 4, 5), 1, n=80)[0].shape[0] == 16
    # a matrix that is NOT invariant (one stiffened entry): the generators that move it are dropped
    K2 = K.copy().tolil()
    K2[0, 0] *= 2.0
    assert c_group((4, 4, 4), 3, K2.tocsr(), n)[0].shape[0] < 48


def test_every_environment_knob_is_documented():
    """The verdict of round 2: ~20 PMH_* environment knobs steer kernels at run time, "a knob table exists, no test pins" it.  Every getenv("PMH_...") of the
    library and every PMH_* the Python side reads must appear in DESIGN.md's knob appendix (with its default there), so that a knob cannot be added silently."""
    import glob
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    design = open(os.path.join(root, "DESIGN.md")).read()
    appendix = design[design.index("### Appendix: environment knobs"):]
    knobs = set()
    for f in glob.glob(os.path.join(root, "permon_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "permon_amd", "csrc", "*.h")):
        knobs |= set(re.findall(r'getenv\("(PMH_[A-Z0-9_]+)"\)', open(f).read()))
    for f in glob.glob(os.path.join(root, "permon_amd", "*.py")) + [os.path.join(root, "bench.py")]:
        knobs |= set(re.findall(r'environ[^\n]*?"(PMH_[A-Z0-9_]+)"', open(f).read()))
    assert len(knobs) > 30
    missing = sorted(k for k in knobs if "`%s`" % k not in appendix and ("`%s=" % k) not in appendix)
    assert not missing, "environment knobs missing from DESIGN.md's appendix: %s" % missing


def test_box_symmetry_closure_host():
    """pmh_box_symmetry_closure: a face of a cube closes to the whole boundary under the 48 operations; under a matrix that breaks the symmetry only the identity is left."""
    import permon_amd as pa
    from permon_amd.mat import box_symmetry_closure

    f = pa.CubeFeti((1, 1, 1), 4, contact=False)
    nn = 5
    K = f.K.tocsr()
    nodes = np.arange(nn ** 3)
    ijk = np.stack([nodes % nn, (nodes // nn) % nn, nodes // (nn * nn)], axis=1)
    face = np.nonzero(ijk[:, 0] == 0)[0]
    rel = (face[:, None] * 3 + np.arange(3)[None, :]).ravel()
    closure, nsym = box_symmetry_closure((nn, nn, nn), 3, K, rel)
    boundary = np.nonzero(((ijk == 0) | (ijk == nn - 1)).any(axis=1))[0]
    assert nsym == 48 and np.array_equal(closure, np.sort((boundary[:, None] * 3 + np.arange(3)[None, :]).ravel()))
    K2 = K.tolil()
    K2[7, 7] += 1.0  # no operation but the identity leaves this matrix invariant ... unless dof 7 is a fixed point of some of them
    closure2, nsym2 = box_symmetry_closure((nn, nn, nn), 3, K2.tocsr(), rel)
    assert nsym2 < 48 and set(rel) <= set(closure2) and closure2.size < closure.size
