"""Mat implementations of the FETI path over the C ABI: MATGLUING, MATBLOCKDIAG, MATINV, QPPF and the shell
operators of the QP transform chain (names follow include/permonmat.h / permonqppf.h)."""
import ctypes as C

import numpy as np

from ._lib import check
from .core import CsrMat, Op, Vec


class MatGluing:
    """MatCreateGluing (src/mat/impls/gluing/gluing.c:216-258): leaves = (local primal dof, lambda index, sign)."""

    def __init__(self, ctx, n_x, n_lambda, leaves_row, leaves_root, leaves_sign):
        lr = np.ascontiguousarray(leaves_row, dtype=np.int32)
        lo = np.ascontiguousarray(leaves_root, dtype=np.int32)
        ls = np.ascontiguousarray(leaves_sign, dtype=np.float64)
        assert lr.size == lo.size == ls.size
        self.ctx, self.n_x, self.n_lambda = ctx, int(n_x), int(n_lambda)
        h = C.c_void_p()
        check(ctx.L.pmh_gluing_create(ctx.h, self.n_x, self.n_lambda, lr.size, lr.ctypes.data_as(C.c_void_p), lo.ctypes.data_as(C.c_void_p),
                                      ls.ctypes.data_as(C.c_void_p), C.byref(h)))
        self.h = h

    def mult(self, lam, x):  # MatMult_Gluing: x = B' lambda
        check(self.ctx.L.pmh_gluing_mult(self.h, lam.p, x.p))

    def mult_transpose(self, x, lam):  # MatMultTranspose_Gluing: lambda = B x (+ all-reduce across GPUs)
        check(self.ctx.L.pmh_gluing_mult_transpose(self.h, x.p, lam.p))

    def mult_add(self, lam, x1, x):  # MatMultAdd_Gluing: x = x1 + B' lambda
        check(self.ctx.L.pmh_gluing_mult_add(self.h, lam.p, x1.p, x.p))

    def mult_transpose_add(self, x, lam1, lam):  # MatMultTransposeAdd_Gluing: lambda = lambda1 + B x
        check(self.ctx.L.pmh_gluing_mult_transpose_add(self.h, x.p, lam1.p, lam.p))

    def destroy(self):
        if self.h:
            self.ctx.L.pmh_gluing_destroy(self.h)
            self.h = None


class MatExtension:
    """MatCreateExtension (src/mat/impls/extension/extension.c): TA = scatter(ris) * A * gather(cis), A condensed CSR."""

    def __init__(self, ctx, n_r, n_c, A, ris, cis):
        ris = np.ascontiguousarray(ris, dtype=np.int32)
        cis = np.ascontiguousarray(cis, dtype=np.int32)
        assert ris.size == A.nrows and cis.size == A.ncols
        self.ctx, self.A, self.n_r, self.n_c = ctx, A, int(n_r), int(n_c)
        h = C.c_void_p()
        check(ctx.L.pmh_extension_create(ctx.h, self.n_r, self.n_c, A.h, ris.ctypes.data_as(C.c_void_p), cis.ctypes.data_as(C.c_void_p), C.byref(h)))
        self.h = h

    def mult(self, c, r):  # MatMult_Extension
        check(self.ctx.L.pmh_extension_mult(self.h, c.p, r.p))

    def mult_transpose(self, r, c):  # MatMultTranspose_Extension
        check(self.ctx.L.pmh_extension_mult_transpose(self.h, r.p, c.p))

    def mult_add(self, c, r1, r):  # MatMultAdd_Extension: r = r1 + TA c
        check(self.ctx.L.pmh_extension_mult_add(self.h, c.p, r1.p, r.p))

    def mult_transpose_add(self, r, c1, c):  # MatMultTransposeAdd_Extension: c = c1 + TA' r
        check(self.ctx.L.pmh_extension_mult_transpose_add(self.h, r.p, c1.p, c.p))

    def destroy(self):
        if self.h:
            self.ctx.L.pmh_extension_destroy(self.h)
            self.h = None


class MatBlockDiag:
    """MatCreateBlockDiag (src/mat/impls/blockdiag/matblockdiag.c:777-854): this rank's subdomain blocks as
    one concatenated CSR + block row offsets."""

    def __init__(self, ctx, block_rowstart, Kcat):
        rs = np.ascontiguousarray(block_rowstart, dtype=np.int32)
        self.ctx, self.K, self.nblocks, self.n = ctx, Kcat, rs.size - 1, Kcat.nrows
        self.block_rowstart = rs
        h = C.c_void_p()
        check(ctx.L.pmh_blockdiag_create(ctx.h, self.nblocks, rs.ctypes.data_as(C.c_void_p), Kcat.h, C.byref(h)))
        self.h = h

    @classmethod
    def from_scipy(cls, ctx, block_rowstart, K):
        K = K.tocsr()
        K.sort_indices()
        return cls(ctx, block_rowstart, CsrMat(ctx, K.shape[0], K.shape[1], K.indptr, K.indices, K.data))

    def mult(self, x, y):  # MatMult_BlockDiag
        check(self.ctx.L.pmh_blockdiag_mult(self.h, x.p, y.p))

    def mult_transpose(self, x, y):  # MatMultTranspose_BlockDiag
        check(self.ctx.L.pmh_blockdiag_mult_transpose(self.h, x.p, y.p))

    def mult_add(self, x, y1, y):  # MatMultAdd_BlockDiag (y1 may be y)
        check(self.ctx.L.pmh_blockdiag_mult_add(self.h, x.p, y1.p, y.p))

    def mult_transpose_add(self, x, y1, y):  # MatMultTransposeAdd_BlockDiag
        check(self.ctx.L.pmh_blockdiag_mult_transpose_add(self.h, x.p, y1.p, y.p))

    def enable_bsr3(self, share=True):
        """MatMult_BlockDiag on the 3x3-block kernel; share: congruent blocks share one device copy (else one copy per block, all of it streamed from HBM)."""
        check(self.ctx.L.pmh_blockdiag_enable_bsr3(self.h, 1 if share else 0))

    def timing_enable(self, max_launches):
        check(self.ctx.L.pmh_blockdiag_timing_enable(self.h, int(max_launches)))

    def timing_get(self):
        """(launches, total ms, CSR bytes 12 nnz + 20 n, HBM bytes of the kernel in use, device copies of the matrix)"""
        n, ms, cb, hb, cp = C.c_int(), C.c_double(), C.c_double(), C.c_double(), C.c_int()
        check(self.ctx.L.pmh_blockdiag_timing_get(self.h, C.byref(n), C.byref(ms), C.byref(cb), C.byref(hb), C.byref(cp)))
        return n.value, ms.value, cb.value, hb.value, cp.value

    def destroy(self):
        if self.h:
            self.ctx.L.pmh_blockdiag_destroy(self.h)
            self.h = None


def MatRegularize(ctx, K, R, rho=None):
    """MatRegularize(K, R, MAT_REG_EXPLICIT) for ONE sequential block (src/mat/interface/permonmatregularize.c:198-287):
    K scipy CSR (p x p), R (d, p) array whose rows span the kernel of K.  Returns (K_reg as scipy CSR, pivots, rho):
    K_reg = K + rho^2 Q with Q the filtered RI (RI'RI)^{-1} RI' on the d fixing DOFs picked by the reference's pivot
    search; rho = MatGetMaxEigenvalue(K, NULL, &rho, 1, 20) on the device unless given."""
    import scipy.sparse as sp

    K = K.tocsr()
    K.sort_indices()
    p = K.shape[0]
    R = np.ascontiguousarray(R, dtype=np.float64).reshape(-1, p) if np.size(R) else np.zeros((0, p))
    d = R.shape[0]
    if rho is None:
        Kd = CsrMat(ctx, p, p, K.indptr, K.indices, K.data)
        op = Op.from_csr(Kd)
        rho, _ = op.max_eigenvalue(tol=1.0, maxits=20)
        op.destroy()
        Kd.destroy()
    rp = np.zeros(p + 1, dtype=np.int32)
    ci = np.zeros(K.nnz + d * d, dtype=np.int32)
    va = np.zeros(K.nnz + d * d)
    piv = np.zeros(max(d, 1), dtype=np.int32)
    nnz = C.c_longlong()
    ip, cp, vp_ = (np.ascontiguousarray(K.indptr, dtype=np.int32), np.ascontiguousarray(K.indices, dtype=np.int32), np.ascontiguousarray(K.data, dtype=np.float64))
    check(ctx.L.pmh_mat_regularize_csr(p, ip.ctypes.data_as(C.c_void_p), cp.ctypes.data_as(C.c_void_p), vp_.ctypes.data_as(C.c_void_p), d,
                                       R.ctypes.data_as(C.c_void_p) if d else None, float(rho), piv.ctypes.data_as(C.c_void_p), rp.ctypes.data_as(C.c_void_p),
                                       ci.ctypes.data_as(C.c_void_p), va.ctypes.data_as(C.c_void_p), C.byref(nnz)))
    Kreg = sp.csr_matrix((va[:nnz.value], ci[:nnz.value], rp), shape=(p, p))
    return Kreg, piv[:d].copy(), float(rho)


class MatInv:
    """MATINV apply (src/mat/impls/inv/matinv.c:734-743) on the iterative per-block KSPCG path."""

    def __init__(self, K, rtol=1e-10, atol=1e-50, max_it=10000, jacobi=True, nullspace=None):
        self.ctx, self.K = K.ctx, K
        h = C.c_void_p()
        check(self.ctx.L.pmh_matinv_create(K.h, float(rtol), float(atol), int(max_it), int(bool(jacobi)), C.byref(h)))
        self.h = h
        if nullspace is not None:
            self.set_nullspace(nullspace)

    def set_nullspace(self, R):
        """R: array (kdim, n) whose rows restricted to a block are that block's orthonormal kernel basis
        (MatInvSetNullSpace + the Moore-Penrose wrapping of QPTDualize)."""
        R = np.ascontiguousarray(R, dtype=np.float64)
        assert R.ndim == 2 and R.shape[1] == self.K.n
        check(self.ctx.L.pmh_matinv_set_nullspace(self.h, R.shape[0], R.ctypes.data_as(C.c_void_p)))

    def set_left_inverse(self, fix_dofs):
        """-qpt_dualize_Kplus_left (QPTDualize qptransform.c:997-1062): K^+ := K^- P_R; fix_dofs = the null-pivot dofs (identity rows / columns in K), [] = off."""
        fx = np.ascontiguousarray(fix_dofs, dtype=np.int32)
        check(self.ctx.L.pmh_matinv_set_left_inverse(self.h, int(fx.size), fx.ctypes.data_as(C.c_void_p) if fx.size else None))

    def set_pc_mg(self, hier, degree=2, lo=0.1, hi=1.1, precision="fp64"):
        """PCMG-like V-cycle as the PC of the inner CG (-mat_inv_pc_type mg): hier is box_mg_hierarchy()'s dict whose
        level 0 is this matrix (the resident CSR of the MATBLOCKDIAG is reused, not uploaded twice).
        precision "fp32": single-precision cycle (3x3-block operators only)."""
        self.mg = MG(self.ctx, hier, degree=degree, lo=lo, hi=hi, fine=self.K.K, precision=precision)
        check(self.ctx.L.pmh_matinv_set_pc_mg(self.h, self.mg.h))
        return self.mg

    def set_pc_mg_box(self, K, dims, ndof, R=None, min_nodes=400, degree=2, precision="fp64"):
        """The same V-cycle PC with the hierarchy built inside libpermonhip (pmh_mg_create_box, host C++): K = the scipy CSR this
        MATINV works on (the host copy of the resident matrix), dims = [(nx, ny, nz)] node boxes of the blocks, R = (kdim, n) kernel
        vectors (zero over non-singular blocks) or None."""
        K = K.tocsr()
        K.sort_indices()
        ip, ci, va = np.ascontiguousarray(K.indptr, dtype=np.int32), np.ascontiguousarray(K.indices, dtype=np.int32), np.ascontiguousarray(K.data, dtype=np.float64)
        rs = np.ascontiguousarray(self._rowstart(), dtype=np.int32)
        dm = np.ascontiguousarray(dims, dtype=np.int32).reshape(-1, 3)
        Rm = np.ascontiguousarray(R, dtype=np.float64) if R is not None and np.size(R) else None
        h = C.c_void_p()
        check(self.ctx.L.pmh_mg_create_box(self.ctx.h, self.K.K.h, rs.size - 1, rs.ctypes.data_as(C.c_void_p), dm.ctypes.data_as(C.c_void_p), int(ndof), ip.ctypes.data_as(C.c_void_p),
                                           ci.ctypes.data_as(C.c_void_p), va.ctypes.data_as(C.c_void_p), Rm.shape[0] if Rm is not None else 0, Rm.ctypes.data_as(C.c_void_p) if Rm is not None else None,
                                           int(min_nodes), int(degree), {"fp64": 0, "fp32": 1, "fp16": 2}[precision], C.byref(h)))
        self.mg = MG.__new__(MG)
        self.mg.ctx, self.mg.h, self.mg.precision = self.ctx, h, precision
        check(self.ctx.L.pmh_matinv_set_pc_mg(self.h, h))
        return self.mg

    def set_pc_mg_sa(self, K, ndof, R=None, nns=None, max_coarse=1500, theta=0.08, degree=2, precision="fp64"):
        """The V-cycle PC with an ALGEBRAIC hierarchy built inside libpermonhip (pmh_mg_create_sa, host C++: smoothed aggregation) -- blocks of any shape, no boxes,
        no caller-supplied P.  K = the scipy CSR this MATINV works on; R = (kdim, n) kernel vectors (zero over non-singular blocks) or None; nns = (m, n)
        near-kernel vectors for the non-singular blocks or None (the ndof translations)."""
        K = K.tocsr()
        K.sort_indices()
        ip, ci, va = np.ascontiguousarray(K.indptr, dtype=np.int32), np.ascontiguousarray(K.indices, dtype=np.int32), np.ascontiguousarray(K.data, dtype=np.float64)
        rs = np.ascontiguousarray(self._rowstart(), dtype=np.int32)
        Rm = np.ascontiguousarray(R, dtype=np.float64) if R is not None and np.size(R) else None
        Nm = np.ascontiguousarray(nns, dtype=np.float64) if nns is not None and np.size(nns) else None
        h = C.c_void_p()
        check(self.ctx.L.pmh_mg_create_sa(self.ctx.h, self.K.K.h, rs.size - 1, rs.ctypes.data_as(C.c_void_p), int(ndof), ip.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p),
                                          va.ctypes.data_as(C.c_void_p), Rm.shape[0] if Rm is not None else 0, Rm.ctypes.data_as(C.c_void_p) if Rm is not None else None,
                                          Nm.shape[0] if Nm is not None else 0, Nm.ctypes.data_as(C.c_void_p) if Nm is not None else None, int(max_coarse), float(theta), int(degree),
                                          {"fp64": 0, "fp32": 1, "fp16": 2}[precision], C.byref(h)))
        self.mg = MG.__new__(MG)
        self.mg.ctx, self.mg.h, self.mg.precision = self.ctx, h, precision
        check(self.ctx.L.pmh_matinv_set_pc_mg(self.h, h))
        return self.mg

    def _rowstart(self):
        return self.K.block_rowstart

    def enable_bsr3(self):
        """K x of the CG on the 3x3-block kernel (MATSEQBAIJ bs=3 role)."""
        check(self.ctx.L.pmh_matinv_enable_bsr3(self.h))

    def timing_enable(self, max_launches):
        check(self.ctx.L.pmh_matinv_timing_enable(self.h, int(max_launches)))

    def bsr3_replicas(self):
        """Congruent blocks served by the ONE device copy of the 3x3-block operator (1: a copy per block, or no 3x3-block copy)."""
        n = C.c_int()
        check(self.ctx.L.pmh_matinv_bsr3_replicas(self.h, C.byref(n)))
        return n.value

    def timing_get(self):
        """(launches, total ms, HBM bytes per launch: the stored matrix once -- shared copies are streamed once -- + x + y) of the CG's K x products since timing_enable."""
        n, ms, b = C.c_int(), C.c_double(), C.c_double()
        check(self.ctx.L.pmh_matinv_timing_get(self.h, C.byref(n), C.byref(ms), C.byref(b)))
        return n.value, ms.value, b.value

    def mult(self, f, u):  # MatMult_Inv
        check(self.ctx.L.pmh_matinv_mult(self.h, f.p, u.p))

    def multi_rhs_active(self):
        """True when mult() runs the solver's 8 congruent blocks as the 8 columns of one block (pmh_matinv_multi_rhs_active)."""
        v = C.c_int()
        check(self.ctx.L.pmh_matinv_multi_rhs_active(self.h, C.byref(v)))
        return bool(v.value)

    def mult_multi(self, F, U):
        """U = K^+ F for 8 columns per block at once (pmh_matinv_mult_multi): F, U vectors of 8 n entries, entry (dof i, column r) at 8 i + r.  Returns the largest iteration count."""
        its = C.c_int()
        check(self.ctx.L.pmh_matinv_mult_multi(self.h, F.p, U.p, C.byref(its)))
        return its.value

    def set_tolerances(self, rtol, atol=1e-50, max_it=10000):  # KSPSetTolerances of the inner KSP
        check(self.ctx.L.pmh_matinv_set_tolerances(self.h, float(rtol), float(atol), int(max_it)))

    def attach_explicit(self, E):
        """F = B K^+ B' built on this MATINV applies through the explicit local dual operators E (None detaches)."""
        check(self.ctx.L.pmh_matinv_attach_explicit(self.h, E.h if E is not None else None))
        self.explicit = E

    def last_iterations(self):
        its, tot = C.c_int(), C.c_longlong()
        check(self.ctx.L.pmh_matinv_last_iterations(self.h, C.byref(its), C.byref(tot)))
        return its.value, tot.value

    def destroy(self):
        if self.h:
            self.ctx.L.pmh_matinv_destroy(self.h)
            self.h = None


def csr_block_classes(block_rowstart, K):
    """Classes of bit-identical diagonal blocks of a block-diagonal scipy CSR (congruent subdomains): array of class ids."""
    from . import _lib

    K = K.tocsr()
    K.sort_indices()
    rs = np.ascontiguousarray(block_rowstart, dtype=np.int32)
    ip, ci, va = np.ascontiguousarray(K.indptr, dtype=np.int32), np.ascontiguousarray(K.indices, dtype=np.int32), np.ascontiguousarray(K.data, dtype=np.float64)
    cls = np.zeros(rs.size - 1, dtype=np.int32)
    ncls = C.c_int()
    check(_lib.load().pmh_csr_block_classes(rs.size - 1, rs.ctypes.data_as(C.c_void_p), ip.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p), va.ctypes.data_as(C.c_void_p),
                                            cls.ctypes.data_as(C.c_void_p), C.byref(ncls)))
    return cls


def box_symmetry_closure(dims, ndof, K, rel):
    """pmh_box_symmetry_closure: the sorted union of the images of the block-relative dofs `rel` under the symmetries of a box of dims nodes x ndof that leave the block's matrix K
    (scipy CSR) invariant; returns (closure, number of operations)."""
    from . import _lib

    K = K.tocsr()
    K.sort_indices()
    ip, ci, va = (np.ascontiguousarray(K.indptr, dtype=np.int32), np.ascontiguousarray(K.indices, dtype=np.int32), np.ascontiguousarray(K.data, dtype=np.float64))
    dm = np.ascontiguousarray(dims, dtype=np.int32)
    rel = np.ascontiguousarray(rel, dtype=np.int32)
    out = np.zeros(int(np.prod(dm)) * int(ndof), dtype=np.int32)
    n_out, nsym = C.c_int(), C.c_int()
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    check(_lib.load().pmh_box_symmetry_closure(p(dm), int(ndof), p(ip), p(ci), p(va), int(rel.size), p(rel) if rel.size else None, C.byref(n_out), p(out), C.byref(nsym)))
    return out[:n_out.value].copy(), nsym.value


class MatExplicitDual:
    """Explicit local dual operators (pmh_fexplicit): W_b = (K_b^+)[Gamma_b, Gamma_b] dense per block, F = Bhat W Bhat'.
    The exact-K^+ path (MatInvExplicitly_Inv, src/mat/impls/inv/matinv.c:670-730, restricted to the dofs B touches)."""

    def __init__(self, B, K, storage="sym", block_class=None, class_extra=None):
        """storage "sym": lower block-triangle + SYMV (half the bytes per apply); "full": row-major + GEMV; "class": congruent blocks
        (block_class from csr_block_classes) share ONE full matrix per class, applied to their vectors together (8 per pass);
        "class_sym": that matrix kept as its lower block-triangle in 16 x 16 tiles (half the bytes, fp64 MFMA kernel);
        "class_orbit": only the rows of the orbit representatives under the class's symmetries (set_box_symmetry / set_class_symmetry before the
        assembly): the dense apply is a GEMM on the fp64 matrix instruction.  class_extra ("class_orbit"): one array of block-relative dofs per class, added to the class's
        touched set (pmh_fexplicit_create_shared_orbit_union) -- e.g. box_symmetry_closure(), so that a class of one box keeps all the box's operations."""
        self.ctx, self.B, self.K, self.storage = B.ctx, B, K, storage
        h = C.c_void_p()
        if storage == "class_orbit" and class_extra is not None:
            bc = np.ascontiguousarray(block_class, dtype=np.int32)
            assert bc.size == K.nblocks and len(class_extra) == int(bc.max()) + 1
            ptr = np.zeros(len(class_extra) + 1, dtype=np.int32)
            ptr[1:] = np.cumsum([len(e) for e in class_extra])
            rel = np.ascontiguousarray(np.concatenate([np.asarray(e, dtype=np.int32) for e in class_extra]) if ptr[-1] else np.zeros(1), dtype=np.int32)
            check(self.ctx.L.pmh_fexplicit_create_shared_orbit_union(B.h, K.h, bc.ctypes.data_as(C.c_void_p), ptr.ctypes.data_as(C.c_void_p), rel.ctypes.data_as(C.c_void_p), C.byref(h)))
        elif storage in ("class", "class_sym", "class_orbit"):
            bc = np.ascontiguousarray(block_class, dtype=np.int32)
            assert bc.size == K.nblocks
            create = {"class": self.ctx.L.pmh_fexplicit_create_shared, "class_sym": self.ctx.L.pmh_fexplicit_create_shared_sym, "class_orbit": self.ctx.L.pmh_fexplicit_create_shared_orbit}[storage]
            check(create(B.h, K.h, bc.ctypes.data_as(C.c_void_p), C.byref(h)))
        else:
            check(self.ctx.L.pmh_fexplicit_create(B.h, K.h, {"full": 0, "sym": 1}[storage], C.byref(h)))
        self.h = h
        nb = C.c_int()
        check(self.ctx.L.pmh_fexplicit_sizes(h, C.byref(nb), None, None, None))
        self.nblocks = nb.value
        ng = np.zeros(self.nblocks, dtype=np.int32)
        db, gb = C.c_longlong(), C.c_double()
        check(self.ctx.L.pmh_fexplicit_sizes(h, None, ng.ctypes.data_as(C.c_void_p), C.byref(db), C.byref(gb)))
        self.n_gamma, self.dense_bytes, self.gemv_bytes = ng, db.value, gb.value

    def set_stripe(self, rank, size):
        """Several GPUs, congruent blocks: this object spans all blocks; the rank keeps the 128-row stripes idx = rank (mod size)."""
        check(self.ctx.L.pmh_fexplicit_set_stripe(self.h, int(rank), int(size)))
        gb = C.c_double()
        check(self.ctx.L.pmh_fexplicit_sizes(self.h, None, None, None, C.byref(gb)))
        self.gemv_bytes = gb.value  # this rank's share

    def class_union(self, cls):
        """The touched dofs of class `cls` relative to the block start, ascending (the row numbering of W_c); class-shared storages."""
        n = C.c_int()
        check(self.ctx.L.pmh_fexplicit_class_union(self.h, int(cls), C.byref(n), None))
        u = np.zeros(max(n.value, 1), dtype=np.int32)
        check(self.ctx.L.pmh_fexplicit_class_union(self.h, int(cls), C.byref(n), u.ctypes.data_as(C.c_void_p)))
        return u[:n.value]

    def set_class_symmetry(self, cls, perm, sign):
        """Set-up by symmetry ("class_sym"): perm[g, i], sign[g, i] = signed permutations of the block's dofs under which K (hence K^+) is
        invariant, operation 0 the identity (feti.box_symmetries); operations that do not map the touched dofs onto themselves are dropped.
        Returns the number of operations used: one K^+ solve per orbit of rows."""
        u = self.class_union(cls)
        nloc = perm.shape[1]
        pos = -np.ones(nloc, dtype=np.int64)
        pos[u] = np.arange(u.size)
        pm = pos[perm[:, u]]  # [nsym, n_c]
        keep = np.all(pm >= 0, axis=1)
        keep[0] = True
        pm = np.ascontiguousarray(pm[keep], dtype=np.int32)
        sg = np.ascontiguousarray(sign[keep][:, u], dtype=np.int8)
        check(self.ctx.L.pmh_fexplicit_set_class_symmetry(self.h, int(cls), int(pm.shape[0]), pm.ctypes.data_as(C.c_void_p), sg.ctypes.data_as(C.c_void_p)))
        return int(pm.shape[0])

    def set_box_symmetry(self, cls, dims, ndof, Kblock):
        """Box-shaped blocks ("class_sym"): the symmetries of the box that leave Kblock (one block of the class, scipy CSR) invariant and map the
        touched dofs onto themselves serve the set-up (pmh_fexplicit_set_box_symmetry).  Returns the number of operations used."""
        Kb = Kblock.tocsr()
        Kb.sort_indices()
        d = np.ascontiguousarray(dims, dtype=np.int32)
        ip, ci, va = (np.ascontiguousarray(Kb.indptr, dtype=np.int32), np.ascontiguousarray(Kb.indices, dtype=np.int32), np.ascontiguousarray(Kb.data, dtype=np.float64))
        n = C.c_int()
        check(self.ctx.L.pmh_fexplicit_set_box_symmetry(self.h, int(cls), d.ctypes.data_as(C.c_void_p), int(ndof), ip.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p),
                                                         va.ctypes.data_as(C.c_void_p), C.byref(n)))
        return n.value

    def refresh_sizes(self):
        """dense_bytes / gemv_bytes after the plan changed (stripe, symmetries, assembly)."""
        db, gb = C.c_longlong(), C.c_double()
        check(self.ctx.L.pmh_fexplicit_sizes(self.h, None, None, C.byref(db), C.byref(gb)))
        self.dense_bytes, self.gemv_bytes = db.value, gb.value

    def apply_flops(self):
        """"class_orbit": useful flops of one dense apply (its roofline is the fp64 MFMA peak); 0 for the streaming storages."""
        f = C.c_double()
        check(self.ctx.L.pmh_fexplicit_apply_flops(self.h, C.byref(f)))
        return f.value

    def apply_flops_detail(self):
        """"class_orbit": (flops the matrix cores execute: padded tiles, flops of the unpruned product over every (representative, operation, block))."""
        a, b = C.c_double(), C.c_double()
        check(self.ctx.L.pmh_fexplicit_apply_flops_detail(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def assemble(self, solver, slot_class=None, block_class=None, rtol=1e-12, max_it=0, multi_rhs=False):
        """One K^+ application of `solver` (a MatInv with solver.K.nblocks slots) per batch of unit right-hand sides.  multi_rhs: 8 columns per block and application on the
        multi-right-hand-side solver (matinv_mv.hip: slot s = column s % 8 of block s // 8; slot_class still names the class of every BLOCK of the solver)."""
        nslots = solver.K.nblocks * (8 if multi_rhs else 1)
        if multi_rhs and slot_class is not None:
            slot_class = np.repeat(np.asarray(slot_class, dtype=np.int32), 8)
        sc = np.ascontiguousarray(slot_class, dtype=np.int32) if slot_class is not None else None
        bc = np.ascontiguousarray(block_class, dtype=np.int32) if block_class is not None else None
        check(self.ctx.L.pmh_fexplicit_assemble(self.h, solver.h, nslots, sc.ctypes.data_as(C.c_void_p) if sc is not None else None,
                                                bc.ctypes.data_as(C.c_void_p) if bc is not None else None, float(rtol), int(max_it)))
        self.refresh_sizes()

    def assemble_stats(self):
        n, t = C.c_longlong(), C.c_double()
        check(self.ctx.L.pmh_fexplicit_assemble_stats(self.h, C.byref(n), C.byref(t)))
        return n.value, t.value

    def block(self, b):
        """(W_b as a numpy array, Gamma_b as rank-local primal indices)."""
        n = int(self.n_gamma[b])
        W, g = np.zeros((n, n)), np.zeros(n, dtype=np.int32)
        check(self.ctx.L.pmh_fexplicit_get_block(self.h, int(b), W.ctypes.data_as(C.c_void_p), g.ctypes.data_as(C.c_void_p)))
        return W, g

    def mult(self, lam, y):  # y = F lambda
        check(self.ctx.L.pmh_fexplicit_mult(self.h, lam.p, y.p))

    def compressed_size(self):
        n = C.c_int()
        gs = np.zeros(self.nblocks + 1, dtype=np.int32)
        check(self.ctx.L.pmh_fexplicit_compressed_size(self.h, C.byref(n), gs.ctypes.data_as(C.c_void_p)))
        return n.value, gs

    def dense_mult(self, xh, yh):
        check(self.ctx.L.pmh_fexplicit_dense_mult(self.h, xh.p, yh.p))

    def timing_enable(self, max_launches, stride=1):
        check(self.ctx.L.pmh_fexplicit_timing_enable(self.h, int(max_launches), int(stride)))

    def timing_get(self):
        n, ms, ms1 = C.c_int(), C.c_double(), C.c_double()
        check(self.ctx.L.pmh_fexplicit_timing_get(self.h, C.byref(n), C.byref(ms), C.byref(ms1)))
        self.first_kernel_ms = ms1.value  # "sym": k_fx_symv alone, the rest of ms is k_fx_symv_fin
        return n.value, ms.value, self.gemv_bytes

    def destroy(self):
        if self.h:
            self.ctx.L.pmh_fexplicit_destroy(self.h)
            self.h = None


class MG:
    """pmh_mg: Galerkin multigrid V-cycle with Chebyshev/Jacobi smoothing and dense coarse pseudo-inverses (PCMG role)."""

    def __init__(self, ctx, hier, degree=2, lo=0.1, hi=1.1, fine=None, precision="fp64"):
        self.ctx = ctx
        self.precision = precision
        A, P = hier["A"], hier["P"]
        self.nlevels = len(A)
        import os

        if os.environ.get("PMH_MG_WINDOW"):  # tuning knob: "lo,hi" fractions of lambda_max(D^-1 A) of the Chebyshev window
            lo, hi = (float(v) for v in os.environ["PMH_MG_WINDOW"].split(","))

        def up(M):
            M = M.tocsr()
            M.sort_indices()
            return CsrMat(ctx, M.shape[0], M.shape[1], M.indptr, M.indices, M.data)

        self.A = [fine if fine is not None else up(A[0])] + [up(a) for a in A[1:]]
        self.P = [up(p) for p in P]
        if self.A[0].nrows != A[0].shape[0]:
            raise ValueError("fine operator size does not match the hierarchy")
        Ah = (C.c_void_p * self.nlevels)(*[a.h for a in self.A])
        Ph = (C.c_void_p * max(1, len(self.P)))(*[p.h for p in self.P])
        lam = np.ascontiguousarray(hier["lambda_max"] if len(self.P) else [1.0], dtype=np.float64)
        crs = np.ascontiguousarray(hier["coarse_rowstart"], dtype=np.int32)
        cpinv = np.ascontiguousarray(hier["coarse_pinv"], dtype=np.float64)
        h = C.c_void_p()
        check(ctx.L.pmh_mg_create(ctx.h, self.nlevels, Ah, Ph, int(degree), lam.ctypes.data_as(C.c_void_p), float(lo), float(hi), crs.size - 1,
                                  crs.ctypes.data_as(C.c_void_p), cpinv.ctypes.data_as(C.c_void_p), {"fp64": 0, "fp32": 1, "fp16": 2}[precision], C.byref(h)))
        self.h = h

    def apply(self, b, x):  # PCApply
        check(self.ctx.L.pmh_mg_apply(self.h, b.p, x.p))

    def timing_enable(self, max_launches):
        check(self.ctx.L.pmh_mg_timing_enable(self.h, int(max_launches)))

    def timing_get(self):
        """(launches, total ms, algorithmic bytes per launch) of the fine-level operator launches of the cycle."""
        n, ms, b = C.c_int(), C.c_double(), C.c_double()
        check(self.ctx.L.pmh_mg_timing_get(self.h, C.byref(n), C.byref(ms), C.byref(b)))
        return n.value, ms.value, b.value

    def fine_spmv(self):
        n = C.c_longlong()
        check(self.ctx.L.pmh_mg_stats(self.h, C.byref(n)))
        return n.value

    def destroy(self):
        if self.h:
            self.ctx.L.pmh_mg_destroy(self.h)
            self.h = None


class QPPF:
    """Projector factory on G (src/qppf/interface/qppf.c): Q = G'(GG')^{-1}G, P = I - Q."""

    def __init__(self, ctx, G, orthonormal=False):
        """orthonormal: False (G as it is, dense (GG')^{-1} in between), True (G has orthonormal rows), "implicit" (G is orthonormalised
        implicitly: the object acts as T G with GG' = LL', T = L^{-1}, but G keeps its sparsity; the reference's -qp_E_orth_form implicit)."""
        self.ctx, self.G, self.orthonormal = ctx, G, bool(orthonormal)
        self.implicit = orthonormal == "implicit"
        self.m, self.n = G.nrows, G.ncols
        h = C.c_void_p()
        check(ctx.L.pmh_qppf_create(ctx.h, G.h, 2 if self.implicit else int(self.orthonormal), C.byref(h)))
        self.h = h

    def orth_rhs(self, e0):
        """e = T e0: the right-hand side of the implicitly orthonormalised constraint."""
        e0 = np.ascontiguousarray(e0, dtype=np.float64)
        e = np.zeros_like(e0)
        check(self.ctx.L.pmh_qppf_orth_rhs(self.h, e0.ctypes.data_as(C.c_void_p), e.ctypes.data_as(C.c_void_p)))
        return e

    @classmethod
    def from_scipy(cls, ctx, G, orthonormal=False):
        G = G.tocsr()
        G.sort_indices()
        return cls(ctx, CsrMat(ctx, G.shape[0], G.shape[1], G.indptr, G.indices, G.data), orthonormal)

    def setup_stats(self):
        """(GG' assembly ms on the matrix cores, its flops, host Cholesky + inverse ms)."""
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        check(self.ctx.L.pmh_qppf_setup_stats(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def ApplyQ(self, v, Qv):
        check(self.ctx.L.pmh_qppf_apply_Q(self.h, v.p, Qv.p))

    def ApplyP(self, v, Pv):
        check(self.ctx.L.pmh_qppf_apply_P(self.h, v.p, Pv.p))

    def ApplyGtG(self, v, y):
        check(self.ctx.L.pmh_qppf_apply_GtG(self.h, v.p, y.p))

    def ApplyCP(self, x, y):
        check(self.ctx.L.pmh_qppf_apply_CP(self.h, x.p, y.p))

    def ApplyG(self, v, Gv):
        check(self.ctx.L.pmh_qppf_apply_G(self.h, v.p, Gv.p))

    def ApplyHalfQ(self, x, y):
        check(self.ctx.L.pmh_qppf_apply_halfQ(self.h, x.p, y.p))

    def ApplyHalfQTranspose(self, x, y):
        check(self.ctx.L.pmh_qppf_apply_halfQ_transpose(self.h, x.p, y.p))

    def destroy(self):
        if self.h:
            self.ctx.L.pmh_qppf_destroy(self.h)
            self.h = None


def MatCreatePenalized(A, pf, rho):  # src/qp/utils/matpenalized.c:212-243
    h = C.c_void_p()
    check(A.ctx.L.pmh_op_create_penalized(A.h, pf.h, float(rho), C.byref(h)))
    return Op(A.ctx, h, A.n, keep=[A, pf])


def MatCreateProjected(A, pf, symmetric=True):  # P*A*P / P*A, qptransform.c:273-284
    h = C.c_void_p()
    check(A.ctx.L.pmh_op_create_projected(A.h, pf.h, int(bool(symmetric)), C.byref(h)))
    return Op(A.ctx, h, A.n, keep=[A, pf])


def MatCreateFetiDual(B, Kplus):  # F = B K^+ B', qptransform.c:1103-1128
    h = C.c_void_p()
    check(B.ctx.L.pmh_op_create_feti_dual(B.h, Kplus.h, C.byref(h)))
    return Op(B.ctx, h, B.n_lambda, keep=[B, Kplus])


def PCDualLumpedOp(B, K):
    """PCDUAL lumped (src/pc/impls/dual/pcdual.c:63-78) as an operator y = B K B' x."""
    ctx = B.ctx

    def fn(xp, yp):
        check(ctx.L.pmh_pc_dual_lumped_apply(B.h, K.h, xp, yp))

    return Op.shell(ctx, B.n_lambda, fn)


def MatCreateSVMDual(ctx, X, y):
    """Matrix-free H = diag(y) X X' diag(y) of the hinge-loss SVM dual (BASELINE configs[4]); X: (n_local, d) row-major."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    n, d = X.shape
    Xd = Vec.from_numpy(ctx, X.ravel())
    yd = Vec.from_numpy(ctx, y)
    h = C.c_void_p()
    check(ctx.L.pmh_op_create_svm_dual(ctx.h, n, d, Xd.p, yd.p, C.byref(h)))
    op = Op(ctx, h, n, keep=[Xd, yd])

    def passes():
        """How many times the operator has streamed X so far (pmh_op_svm_dual_passes)."""
        k = C.c_longlong(0)
        check(ctx.L.pmh_op_svm_dual_passes(op.h, C.byref(k)))
        return int(k.value)

    op.passes = passes
    return op
