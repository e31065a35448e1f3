"""ctypes front end of the CPU oracle (oracle/permon_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (permon_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
EPS = float(np.finfo(np.float64).eps)

MULT_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double))

EXP_TYPES = {"std": 0, "projcg": 1, "gf": 2, "g": 3, "gfgr": 4, "ggr": 5}
EXPLEN_TYPES = {"fixed": 0, "opt": 1, "optapprox": 2, "bb": 3}


class OrcOp(C.Structure):
    _fields_ = [("mult", MULT_FN), ("ctx", C.c_void_p), ("n", C.c_int)]


class OrcCsr(C.Structure):
    _fields_ = [("nrows", C.c_int), ("ncols", C.c_int), ("rowptr", C.c_void_p), ("col", C.c_void_p), ("val", C.c_void_p)]


class OrcBox(C.Structure):
    _fields_ = [("n", C.c_int), ("nis", C.c_int), ("is_", C.c_void_p), ("lb", C.c_void_p), ("ub", C.c_void_p), ("astol", C.c_double)]


class OrcQppf(C.Structure):
    _fields_ = [("m", C.c_int), ("n", C.c_int), ("G", C.c_void_p), ("GGt_chol", C.c_void_p), ("G_left", C.c_void_p), ("Gt_right", C.c_void_p)]


class OrcGluing(C.Structure):
    _fields_ = [("n_x", C.c_int), ("n_lambda", C.c_int), ("n_leaves", C.c_int), ("leaves_row", C.c_void_p), ("leaves_root", C.c_void_p), ("leaves_sign", C.c_void_p)]


def build(force=False):
    """Compile oracle/liborc.so and liborc_omp.so (gcc); no-op when up to date."""
    so = os.path.join(_HERE, "liborc.so")
    src = [os.path.join(_HERE, f) for f in ("permon_oracle.c", "orc_api.c", "permon_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src if os.path.exists(s)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return so


_libs = {}


def lib(omp=False):
    key = "omp" if omp else "serial"
    if key not in _libs:
        build()
        L = C.CDLL(os.path.join(_HERE, "liborc_omp.so" if omp else "liborc.so"))
        L.orc_qps_new.restype = C.c_void_p
        L.orc_smalxe_new.restype = C.c_void_p
        L.orc_smalxe_inner.restype = C.c_void_p
        L.orc_pcpg_new.restype = C.c_void_p
        L.orc_qps_get.restype = C.c_double
        L.orc_smalxe_get.restype = C.c_double
        L.orc_pcpg_get.restype = C.c_double
        L.orc_qps_trace_steps.restype = C.c_char_p
        L.orc_qps_trace_array.restype = C.POINTER(C.c_double)
        L.orc_qps_work.restype = C.POINTER(C.c_double)
        L.orc_max_eigenvalue.restype = C.c_double
        L.orc_box_feas.restype = C.c_double
        L.orc_time_mpgp.restype = C.c_double
        L.orc_time_spmv.restype = C.c_double
        L.orc_objective.restype = C.c_double
        _libs[key] = L
    return _libs[key]


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Csr:
    """Holds a CSR matrix (int32 / fp64) alive together with its C descriptor."""

    def __init__(self, nrows, ncols, rowptr, col, val):
        self.rowptr, self.col, self.val = _i32(rowptr), _i32(col), _f64(val)
        self.nrows, self.ncols = int(nrows), int(ncols)
        self.c = OrcCsr(self.nrows, self.ncols, _p(self.rowptr), _p(self.col), _p(self.val))

    @classmethod
    def from_scipy(cls, A):
        A = A.tocsr()
        A.sort_indices()
        return cls(A.shape[0], A.shape[1], A.indptr, A.indices, A.data)

    def to_scipy(self):
        import scipy.sparse as sp

        return sp.csr_matrix((self.val, self.col, self.rowptr), shape=(self.nrows, self.ncols))


class Op:
    """orc_op: either a CSR operator (native MatMult_SeqAIJ restatement) or a Python callback."""

    def __init__(self, n, csr=None, fn=None, native=None, omp=False):
        self.n = int(n)
        self._keep = []
        L = lib(omp)
        if csr is not None:
            self.csr = csr
            mult = C.cast(L.orc_csr_mult, MULT_FN)
            self.c = OrcOp(mult, C.cast(C.pointer(csr.c), C.c_void_p), self.n)
        elif native is not None:
            fnptr, ctx, keep = native
            self._keep.append(keep)
            self.c = OrcOp(C.cast(fnptr, MULT_FN), ctx, self.n)
        else:
            n_ = self.n

            def _cb(ctx, xp, yp):
                x = np.ctypeslib.as_array(xp, shape=(n_,))
                y = np.ctypeslib.as_array(yp, shape=(n_,))
                y[:] = fn(x.copy())

            self._cb = MULT_FN(_cb)
            self.c = OrcOp(self._cb, None, self.n)

    def __call__(self, x):
        x = _f64(x)
        y = np.empty(self.n)
        self.c.mult(self.c.ctx, _dp(x), _dp(y))
        return y


class Box:
    def __init__(self, n, lb=None, ub=None, is_=None, astol=10 * EPS):
        self.n = int(n)
        self.lb = _f64(lb) if lb is not None else None
        self.ub = _f64(ub) if ub is not None else None
        self.is_ = _i32(is_) if is_ is not None else None
        nis = len(self.is_) if self.is_ is not None else self.n
        self.c = OrcBox(self.n, nis, _p(self.is_), _p(self.lb), _p(self.ub), astol)

    # --- the four QPC kernels -----------------------------------------------------------------
    def project(self, x):
        x = _f64(x)
        Px = np.empty_like(x)
        lib().orc_box_project(C.byref(self.c), _dp(x), _dp(Px))
        return Px

    def feas(self, x, d):
        return lib().orc_box_feas(C.byref(self.c), _dp(_f64(x)), _dp(_f64(d)))

    def grads(self, x, g):
        x, g = _f64(x), _f64(g)
        gf, gc = np.empty_like(x), np.empty_like(x)
        lib().orc_box_grads(C.byref(self.c), _dp(x), _dp(g), _dp(gf), _dp(gc))
        return gf, gc

    def gradreduced(self, x, gf, alpha):
        x, gf = _f64(x), _f64(gf)
        gr = np.empty_like(x)
        lib().orc_box_gradreduced(C.byref(self.c), _dp(x), _dp(gf), C.c_double(alpha), _dp(gr))
        return gr


def spmv(csr, x):
    y = np.empty(csr.nrows)
    lib().orc_csr_mult(C.byref(csr.c), _dp(_f64(x)), _dp(y))
    return y


def spmv_transpose(csr, x):
    y = np.empty(csr.ncols)
    lib().orc_csr_mult_transpose(C.byref(csr.c), _dp(_f64(x)), _dp(y))
    return y


def max_eigenvalue(op, tol=-1.0, maxits=-1, omp=False):
    its = C.c_int(0)
    lam = lib(omp).orc_max_eigenvalue(C.byref(op.c), C.c_double(tol), C.c_int(maxits), C.byref(its))
    return lam, its.value


def regularize_pivots(R):
    """MatRegularize_GetPivots_Private; R: (d, p) array = the p x d kernel basis stored column-major."""
    R = _f64(R)
    d, p = R.shape
    piv = np.zeros(d, dtype=np.int32)
    lib().orc_regularize_pivots(C.c_int(p), C.c_int(d), _dp(R), _p(piv))
    return piv


def regularize_csr(csr, R, rho):
    """MatRegularize (MAT_REG_EXPLICIT) of one block: returns (rowptr, col, val, pivots) of K + rho^2 Q."""
    R = _f64(R)
    d = R.shape[0] if R.size else 0
    nnz = int(csr.rowptr[-1])
    rp = np.zeros(csr.nrows + 1, dtype=np.int32)
    ci = np.zeros(nnz + d * d, dtype=np.int32)
    va = np.zeros(nnz + d * d)
    piv = np.zeros(max(d, 1), dtype=np.int32)
    L = lib()
    L.orc_regularize_csr.restype = C.c_int
    n = L.orc_regularize_csr(C.byref(csr.c), C.c_int(d), _dp(R) if d else None, C.c_double(rho), _p(piv), _p(rp), _p(ci), _dp(va))
    if n < 0:
        raise ValueError("R(pivots,:) is rank deficient")
    return rp, ci[:n].copy(), va[:n].copy(), piv[:d]


def _qps_results(L, q, x, trace):
    keys = ["iteration", "reason", "rnorm", "gfnorm", "gcnorm", "nmv", "ncg", "nexp", "nprop", "nfinc", "nfall", "alpha", "maxeig", "norm_rhs", "ttol"]
    res = {k: L.orc_qps_get(C.c_void_p(q), k.encode()) for k in keys}
    for k in ("iteration", "reason", "nmv", "ncg", "nexp", "nprop", "nfinc", "nfall"):
        res[k] = int(res[k])
    res["x"] = x
    if trace:
        tl = int(L.orc_qps_get(C.c_void_p(q), b"trace_len"))
        res["steps"] = L.orc_qps_trace_steps(C.c_void_p(q))[:tl].decode()
        for i, name in enumerate(("trace_rnorm", "trace_gfnorm", "trace_gcnorm", "trace_alpha")):
            arr = L.orc_qps_trace_array(C.c_void_p(q), i)
            res[name] = np.ctypeslib.as_array(arr, shape=(tl,)).copy()
    return res


def _apply_opts(L, setter, handle, opts):
    for k, v in opts.items():
        if k == "exptype" and isinstance(v, str):
            v = EXP_TYPES[v]
        if k == "explengthtype" and isinstance(v, str):
            v = EXPLEN_TYPES[v]
        rc = setter(C.c_void_p(handle), k.encode(), C.c_double(float(v)))
        if rc != 0:
            raise KeyError("unknown oracle option %r" % k)


def mpgp(op, b, x0, box, trace_cap=0, omp=False, **opts):
    """QPSSolve_MPGP restatement.  Returns dict with x, counters, reason and (optionally) the monitor trace."""
    L = lib(omp)
    b = _f64(b)
    x = _f64(x0).copy()
    q = L.orc_qps_new()
    try:
        L.orc_qps_set_problem(C.c_void_p(q), C.byref(op.c), _dp(b), _dp(x), C.byref(box.c))
        _apply_opts(L, L.orc_qps_set, q, opts)
        if trace_cap:
            L.orc_qps_enable_trace(C.c_void_p(q), C.c_int(trace_cap))
        L.orc_mpgp_solve(C.c_void_p(q))
        res = _qps_results(L, q, x, trace_cap)
        n = op.n
        for i, name in ((0, "gP"), (1, "gf"), (2, "gc"), (3, "g")):
            res[name] = np.ctypeslib.as_array(L.orc_qps_work(C.c_void_p(q), i), shape=(n,)).copy()
    finally:
        L.orc_qps_delete(C.c_void_p(q))
    return res


def time_mpgp(op, b, x0, box, reps=1, omp=False, **opts):
    """Seconds of the solve phase (power method excluded) and the iterations executed."""
    L = lib(omp)
    b, x0 = _f64(b), _f64(x0)
    x = x0.copy()
    q = L.orc_qps_new()
    try:
        L.orc_qps_set_problem(C.c_void_p(q), C.byref(op.c), _dp(b), _dp(x), C.byref(box.c))
        _apply_opts(L, L.orc_qps_set, q, opts)
        its = C.c_int(0)
        t = L.orc_time_mpgp(C.c_void_p(q), _dp(x0), C.c_int(reps), C.byref(its))
    finally:
        L.orc_qps_delete(C.c_void_p(q))
    return t, its.value


def time_spmv(csr, x, reps=5, omp=False):
    y = np.empty(csr.nrows)
    return lib(omp).orc_time_spmv(C.byref(csr.c), _dp(_f64(x)), _dp(y), C.c_int(reps)) / reps


class Qppf:
    """Projector factory on an explicit G (m x n CSR): Q = G'(GG')^{-1}G, P = I - Q."""

    def __init__(self, G, orthonormal=False):
        self.G = G
        self.m, self.n = G.nrows, G.ncols
        self.orthonormal = bool(orthonormal)
        self.G_left = np.zeros(max(self.m, 1))
        self.Gt_right = np.zeros(max(self.m, 1))
        self.chol = None
        if not orthonormal and self.m > 0:
            Gs = G.to_scipy()
            GGt = np.ascontiguousarray((Gs @ Gs.T).toarray(), dtype=np.float64)
            rc = lib().orc_dense_cholesky(C.c_int(self.m), _dp(GGt))
            if rc != 0:
                raise ValueError("GG' is not SPD")
            self.chol = GGt
        self.c = OrcQppf(self.m, self.n, C.cast(C.pointer(G.c), C.c_void_p), _p(self.chol), _p(self.G_left), _p(self.Gt_right))

    def Q(self, v):
        y = np.empty(self.n)
        lib().orc_qppf_apply_Q(C.byref(self.c), _dp(_f64(v)), _dp(y))
        return y

    def P(self, v):
        y = np.empty(self.n)
        lib().orc_qppf_apply_P(C.byref(self.c), _dp(_f64(v)), _dp(y))
        return y

    def half_Q_transpose(self, x):
        y = np.empty(self.n)
        lib().orc_qppf_apply_halfQ_transpose(C.byref(self.c), _dp(_f64(x)), _dp(y))
        return y


def smalxe(op, b, u0, box, pf, omp=False, inner_opts=None, trace_cap=0, timing=None, **opts):
    """QPSSolve_SMALXE restatement (outer AL loop + inner MPGP with the injected convergence test).
    timing: a dict that receives the seconds of QPSSetUp (power method for the largest eigenvalue, rho, M1) and of QPSSolve separately."""
    L = lib(omp)
    b = _f64(b)
    u = _f64(u0).copy()
    s = L.orc_smalxe_new()
    try:
        L.orc_smalxe_set_problem(C.c_void_p(s), C.byref(op.c), _dp(b), _dp(u), C.byref(box.c), C.byref(pf.c), C.c_int(1 if pf.orthonormal else 0))
        _apply_opts(L, L.orc_smalxe_set, s, opts)
        inner = L.orc_smalxe_inner(C.c_void_p(s))
        if inner_opts:
            _apply_opts(L, L.orc_qps_set, inner, inner_opts)
        if trace_cap:
            L.orc_qps_enable_trace(C.c_void_p(inner), C.c_int(trace_cap))
        import time as _time

        t0 = _time.perf_counter()
        L.orc_smalxe_setup(C.c_void_p(s))
        t1 = _time.perf_counter()
        L.orc_smalxe_solve(C.c_void_p(s))
        if timing is not None:
            timing["setup_seconds"], timing["solve_seconds"] = t1 - t0, _time.perf_counter() - t1
        keys = ["M1", "M1_initial", "eta", "rho", "rho_current", "M1_updates", "M1_hits", "eta_hits", "rho_updates", "state", "inner_iter_accu", "normBu", "enorm", "rnorm", "iteration", "reason", "maxeig", "gtol", "lag_neval", "lag_niter"]
        res = {k: L.orc_smalxe_get(C.c_void_p(s), k.encode()) for k in keys}
        for k in ("M1_updates", "M1_hits", "eta_hits", "rho_updates", "state", "inner_iter_accu", "iteration", "reason", "lag_neval", "lag_niter"):
            res[k] = int(res[k])
        res["inner"] = _qps_results(L, inner, u, trace_cap)
        res["u"] = u
    finally:
        L.orc_smalxe_delete(C.c_void_p(s))
    return res


def pcpg(op, b, x0, pf, rtol=1e-5, atol=1e-50, divtol=1e4, max_it=10000, pc=None):
    """QPSSolve_PCPG restatement; pc is an optional python callable y = M^{-1} x.
    pf=None: no equality constraint, i.e. the CG that QPSKSP runs (qpsksp.c:244-250)."""
    L = lib()
    b = _f64(b)
    x = _f64(x0).copy()
    s = L.orc_pcpg_new(C.byref(op.c), _dp(b), _dp(x), C.byref(pf.c) if pf is not None else None, C.c_double(rtol), C.c_double(atol), C.c_double(divtol), C.c_int(max_it))
    keep = None
    try:
        if pc is not None:
            n = op.n

            def _cb(ctx, xp, yp):
                xx = np.ctypeslib.as_array(xp, shape=(n,))
                yy = np.ctypeslib.as_array(yp, shape=(n,))
                yy[:] = pc(xx.copy())

            keep = MULT_FN(_cb)
            L.orc_pcpg_set_pc(C.c_void_p(s), keep, None)
        L.orc_pcpg_solve(C.c_void_p(s))
        res = {k: L.orc_pcpg_get(C.c_void_p(s), k.encode()) for k in ("rnorm", "iteration", "reason")}
        res["iteration"], res["reason"] = int(res["iteration"]), int(res["reason"])
        res["x"] = x
    finally:
        L.orc_pcpg_delete(C.c_void_p(s))
    return res


class Gluing:
    """MATGLUING restatement: leaves (primal dof, lambda index, sign)."""

    def __init__(self, n_x, n_lambda, leaves_row, leaves_root, leaves_sign):
        self.rows, self.roots, self.signs = _i32(leaves_row), _i32(leaves_root), _f64(leaves_sign)
        self.n_x, self.n_lambda = int(n_x), int(n_lambda)
        self.c = OrcGluing(self.n_x, self.n_lambda, len(self.rows), _p(self.rows), _p(self.roots), _p(self.signs))

    def mult(self, lam):  # x = B' lambda   (MatMult_Gluing)
        x = np.empty(self.n_x)
        lib().orc_gluing_mult(C.byref(self.c), _dp(_f64(lam)), _dp(x))
        return x

    def mult_transpose(self, x):  # lambda = B x   (MatMultTranspose_Gluing)
        lam = np.empty(self.n_lambda)
        lib().orc_gluing_mult_transpose(C.byref(self.c), _dp(_f64(x)), _dp(lam))
        return lam


class MatInv:
    """MATINV apply on the reference's iterative path: per-block Jacobi-CG, Moore-Penrose wrapped (orc_matinv_mult)."""

    def __init__(self, K, rowstart, R=None, rtol=1e-10, atol=1e-50, max_it=20000, omp=False, kernel_tol=0.0):
        self.L = lib(omp)
        self.L.orc_matinv_new.restype = C.c_void_p
        self.L.orc_matinv_spmv_count.restype = C.c_longlong
        self.K, self.rowstart = K, _i32(rowstart)
        self.R = _f64(R) if R is not None else None
        kdim = self.R.shape[0] if self.R is not None else 0
        self.h = self.L.orc_matinv_new(C.byref(K.c), len(self.rowstart) - 1, _p(self.rowstart), kdim, _p(self.R), C.c_double(rtol), C.c_double(atol), C.c_int(max_it))
        if kernel_tol:  # the product's rule for loads that lie in the kernel (off by default: the reference has none)
            self.L.orc_matinv_set_kernel_tol(C.c_void_p(self.h), C.c_double(kernel_tol))

    def mult(self, f):
        u = np.empty(self.K.nrows)
        self.L.orc_matinv_mult(C.c_void_p(self.h), _dp(_f64(f)), _dp(u))
        return u

    def spmv_count(self):
        return int(self.L.orc_matinv_spmv_count(C.c_void_p(self.h)))


class FetiOp:
    """Native operators of the FETI dual QP: F = B K^+ B' (which=0) or A_rho = P F P + rho Q (which=1)."""

    def __init__(self, B, Kplus, pf, rho=0.0, which=0, omp=False):
        L = lib(omp)
        L.orc_feti_new.restype = C.c_void_p
        L.orc_feti_fn.restype = C.c_void_p
        self._keep = (B, Kplus, pf)
        self.h = L.orc_feti_new(C.byref(B.c), C.c_void_p(Kplus.h), C.byref(pf.c) if pf is not None else None, C.c_double(rho))
        self.op = Op(B.n_lambda, native=(L.orc_feti_fn(C.c_int(which)), C.c_void_p(self.h), self), omp=omp)


def kkt_box(op, b, x, lb):
    """The four 'r =' lines of QPViewKKT + QPCViewKKT_Box for a lower-bound-only QP
    (src/qp/interface/qp.c:245-369, src/qpc/impls/box/qpcbox.c:333-427, multipliers qp.c:828-893)."""
    normb = float(np.sqrt(np.dot(b, b)))
    llb = op(x) - b  # QPComputeMissingBoxMultipliers: lambda_lb = A x - b
    r0 = (op(x) - b) - llb
    r = [float(np.linalg.norm(r0))]
    r.append(float(np.linalg.norm(np.minimum(x - lb, 0.0))))
    r.append(float(np.linalg.norm(np.minimum(llb, 0.0))))
    d = lb - x
    d = np.where(lb <= -np.inf, -1.0, d)
    r.append(float(abs(np.dot(llb, d))))
    return r, normb
