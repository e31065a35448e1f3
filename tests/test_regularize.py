"""MatRegularize (src/mat/interface/permonmatregularize.c): the oracle restatement against a hand-followed case and the
algebraic properties the reference relies on, and the product's host routines (C ABI, no device call) against the oracle."""
import ctypes as C

import numpy as np
import scipy.sparse as sp

import permon_amd as pa
from permon_amd import _lib


def _drand48_stream(seed, n):
    """glibc srand48/drand48: X0 = (seed << 16) | 0x330E, X <- (0x5DEECE66D X + 0xB) mod 2^48, value X / 2^48."""
    x = ((seed & 0xFFFFFFFF) << 16) | 0x330E
    out = np.empty(n)
    for i in range(n):
        x = (0x5DEECE66D * x + 0xB) & ((1 << 48) - 1)
        out[i] = x / float(1 << 48)
    return out


def test_pivots_hand_followed_case(oracle):
    # p = 4, d = 2 (columns stored contiguously = column-major p x d).  Following permonmatregularize.c:36-100 by hand:
    # J=1, II=3: largest |entry| over both columns is -9 at (i=1, j=0) -> row 1 <-> row 3, column 0 <-> column 1,
    #            perm = [0,3,2,1]; column 0 is combined so that its last row vanishes.
    # J=0, II=2: column 0 restricted to rows 0..2; its largest entry decides the second pivot.
    R = np.array([[1.0, -9.0, 2.0, 0.5],   # column 0
                  [4.0, 3.0, -1.0, 2.0]])  # column 1
    # after step 1: W(:,1) = [1, .5, 2, -9] (old col 0, rows 1<->3), W(:,0) = old col 1 rows swapped = [4, 2, -1, 3];
    # alpha = -(-9)/3 = 3: W(:,0) = 3*[4,2,-1,3] + [1,.5,2,-9] = [13, 6.5, -1, 0] -> max over rows 0..2 is row 0 (13),
    # swapped with row II = 2: perm = [2,3,0,1] -> pivots = sorted(perm[2:]) = [0, 1]
    assert oracle.regularize_pivots(R).tolist() == [0, 1]
    # a case where the second pivot is not row 0
    R2 = np.array([[0.1, -9.0, 2.0, 0.5], [0.2, 3.0, -1.0, 2.0]])
    # step 1 as above (pivot -9 at row 1): W(:,0) = 3*[.2, 2, -1, 3] + [.1, .5, 2, -9] = [.7, 6.5, -1, 0]: max is row 1 of the
    # swapped numbering, whose perm entry is 3 -> pivots = sorted([3, 1]) = [1, 3]
    assert oracle.regularize_pivots(R2).tolist() == [1, 3]


def _floating_block(nel=2):
    f = pa.CubeFeti((2, 1, 1), nel, contact=False)  # TFETI: Dirichlet rows live in B, every K_i is floating
    K = f.Ki.tocsr()
    K.sort_indices()
    R = np.ascontiguousarray(f.R[:, :f.n_i])
    return K, R


def test_oracle_regularization_properties(oracle):
    K, R = _floating_block(2)
    p, d = K.shape[0], R.shape[0]
    assert d == 6 and np.abs(K @ R.T).max() < 1e-12 * abs(K).max()
    rho = 3.7
    rp, ci, va, piv = oracle.regularize_csr(oracle.Csr.from_scipy(K), R, rho)
    assert piv.tolist() == sorted(set(piv.tolist())) and len(piv) == d
    # the fixing DOFs make R(pivots,:) non-singular (that is what the pivot search is for)
    assert np.linalg.cond(R[:, piv]) < 1e3
    Kreg = sp.csr_matrix((va, ci, rp), shape=(p, p))
    D = (Kreg - K).toarray()
    # K_reg - K lives on the pivots x pivots block and equals rho^2 Q with Q = RI (RI'RI)^{-1} RI' = I (RI square, regular)
    mask = np.zeros((p, p), dtype=bool)
    mask[np.ix_(piv, piv)] = True
    assert np.abs(D[~mask]).max() == 0.0
    assert np.allclose(D[np.ix_(piv, piv)], rho * rho * np.eye(d), rtol=0, atol=1e-12 * rho * rho)
    # SPD, and its inverse is a generalised inverse of K: K K_reg^{-1} K = K
    Kd, Krd = K.toarray(), Kreg.toarray()
    assert np.linalg.eigvalsh(Krd).min() > 1e-8 * np.linalg.eigvalsh(Krd).max()
    assert np.abs(Kd @ np.linalg.solve(Krd, Kd) - Kd).max() <= 1e-9 * np.abs(Kd).max()
    # sorted, duplicate-free rows (PETSc AIJ invariant)
    for i in range(p):
        assert np.all(np.diff(ci[rp[i]:rp[i + 1]]) > 0)
    # d = 0 (no kernel): K_reg = K
    rp0, ci0, va0, piv0 = oracle.regularize_csr(oracle.Csr.from_scipy(K), np.zeros((0, p)), rho)
    assert np.array_equal(rp0, K.indptr) and np.array_equal(ci0, K.indices) and np.array_equal(va0, K.data) and len(piv0) == 0


def test_power_method_null_space_restart(oracle):
    """permonmatutils.c:491-499: v = 1 lies in the kernel of a floating block, A v is replaced by the RAND48 stream
    (seed 0x12345678) and lambda keeps its value in that iteration; MatRegularize calls it with tol = 1, maxits = 20."""
    K, R = _floating_block(2)
    op = oracle.Op(K.shape[0], csr=oracle.Csr.from_scipy(K))
    lam, its = oracle.max_eigenvalue(op, tol=1.0, maxits=20)
    # restate the loop in numpy with the published drand48 recurrence
    n = K.shape[0]
    v = np.ones(n)
    lam_ref, l0, it_ref, rnd = 0.0, 0.0, 0, _drand48_stream(0x12345678, n)
    for i in range(1, 21):
        it_ref = i
        l0 = lam_ref
        Av = K @ v
        vv = v @ v
        lam_ref = (v @ Av) / vv
        if lam_ref < np.finfo(float).eps:
            Av = rnd.copy()
        with np.errstate(divide="ignore", invalid="ignore"):
            relerr = abs(lam_ref - l0) / abs(lam_ref)
        if relerr < 1.0:
            break
        v = Av / np.sqrt(vv)
    assert its == it_ref and its >= 2
    assert abs(lam - lam_ref) <= 1e-9 * abs(lam_ref)
    assert 0.0 < lam <= np.linalg.eigvalsh(K.toarray()).max() * (1 + 1e-12)


def test_product_host_routines_match_oracle(oracle):
    L = _lib.load()
    K, R = _floating_block(3)
    p, d = K.shape[0], R.shape[0]
    # pivots: index bookkeeping, exact
    piv = np.zeros(d, dtype=np.int32)
    _lib.check(L.pmh_mat_regularize_pivots(p, d, R.ctypes.data_as(C.c_void_p), piv.ctypes.data_as(C.c_void_p)))
    assert piv.tolist() == oracle.regularize_pivots(R).tolist()
    rng = np.random.default_rng(4)
    for trial in range(5):  # random full-rank bases, exact agreement of the selected rows
        pp, dd = int(rng.integers(8, 60)), int(rng.integers(1, 7))
        Rr = np.ascontiguousarray(rng.standard_normal((dd, pp)))
        pv = np.zeros(dd, dtype=np.int32)
        _lib.check(L.pmh_mat_regularize_pivots(pp, dd, Rr.ctypes.data_as(C.c_void_p), pv.ctypes.data_as(C.c_void_p)))
        assert pv.tolist() == oracle.regularize_pivots(Rr).tolist()
    # K_reg: same pattern, values to rounding
    rho = 2.5
    rp_o, ci_o, va_o, piv_o = oracle.regularize_csr(oracle.Csr.from_scipy(K), R, rho)
    rp = np.zeros(p + 1, dtype=np.int32)
    ci = np.zeros(K.nnz + d * d, dtype=np.int32)
    va = np.zeros(K.nnz + d * d)
    nnz = C.c_longlong()
    ip, cp, vp = K.indptr.astype(np.int32), K.indices.astype(np.int32), K.data.astype(np.float64)
    _lib.check(L.pmh_mat_regularize_csr(p, ip.ctypes.data_as(C.c_void_p), cp.ctypes.data_as(C.c_void_p), vp.ctypes.data_as(C.c_void_p), d, R.ctypes.data_as(C.c_void_p), rho,
                                        piv.ctypes.data_as(C.c_void_p), rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p), va.ctypes.data_as(C.c_void_p), C.byref(nnz)))
    assert nnz.value == len(ci_o) and np.array_equal(rp, rp_o) and np.array_equal(ci[:nnz.value], ci_o) and piv.tolist() == piv_o.tolist()
    assert np.abs(va[:nnz.value] - va_o).max() <= 1e-14 * np.abs(va_o).max()
    # rank-deficient basis: reported, not silently regularised
    Rbad = np.vstack([R[0], R[0]])
    rc = L.pmh_mat_regularize_csr(p, ip.ctypes.data_as(C.c_void_p), cp.ctypes.data_as(C.c_void_p), vp.ctypes.data_as(C.c_void_p), 2, np.ascontiguousarray(Rbad).ctypes.data_as(C.c_void_p), rho,
                                  piv.ctypes.data_as(C.c_void_p), rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p), va.ctypes.data_as(C.c_void_p), C.byref(nnz))
    assert rc != 0 and b"rank deficient" in L.pmh_last_error()
