"""Synthetic QP generators for the PERMON QPS hot path (numpy, host side).

Each generator restates the *problem definition* of one of the reference's tutorials (cited per
function; paths relative to /root/reference) or one of BASELINE.json's configs, and returns plain
CSR arrays (int32 indices, fp64 values) + vectors.  No solver code lives here.
"""
import numpy as np

__all__ = ["ex1", "ex2", "ex3_primal", "jbearing2", "laplace2d_box", "laplace2d_csr", "svm_dual"]


def _fobst(i, n):
    # src/tutorials/ex1.c:42-46
    h = 1.0 / (n - 1)
    return np.sin(4 * np.pi * i * h - np.pi / 6.0) / 2 - 2


def _tridiag_dirichlet(n):
    """CSR of ex1's Hessian: tridiag(-1,2,-1) with identity first/last rows and the couplings to
    them dropped (src/tutorials/ex1.c:74-100)."""
    rowptr = np.zeros(n + 1, dtype=np.int32)
    col, val = [], []
    for i in range(n):
        if i == 0 or i == n - 1:
            col.append(i)
            val.append(1.0)
        else:
            if i != 1:
                col.append(i - 1)
                val.append(-1.0)
            col.append(i)
            val.append(2.0)
            if i != n - 2:
                col.append(i + 1)
                val.append(-1.0)
        rowptr[i + 1] = len(col)
    return rowptr, np.asarray(col, dtype=np.int32), np.asarray(val, dtype=np.float64)


def ex1(n=100):
    """String-contact QP of src/tutorials/ex1.c:58-157: min 1/2 x'Ax - x'b s.t. x >= c.

    Returns dict(n, rowptr, col, val, b, lb, x0).  b_i = -15 h^2 * 2 (ex1.c:98), lb_i = fobst(i) for
    interior i and 0 at both ends (vector c is zero-initialised, ex1.c:71,99), x0 = 0.
    """
    h = 1.0 / (n - 1)
    rowptr, col, val = _tridiag_dirichlet(n)
    b = np.full(n, -15 * h * h * 2)
    b[0] = b[-1] = 0.0
    i = np.arange(n)
    lb = _fobst(i, n)
    lb[0] = lb[-1] = 0.0
    return dict(n=n, rowptr=rowptr, col=col, val=val, b=b, lb=lb, ub=None, x0=np.zeros(n))


def ex2(n=100, infinite=False):
    """src/tutorials/ex2.c:32-140: as ex1 but the bound acts on the first half of the unknowns only,
    either through an index set (is = [0, n/2), lb of length n/2) or through -inf bounds."""
    p = ex1(n)
    i = np.arange(n)
    if infinite:
        lb = np.where(i < n // 2, _fobst(i, n), -np.inf)
        lb[0] = 0.0
        lb[-1] = 0.0  # rows 0 and n-1 are never set (ex2.c:88-103) -> 0
        p.update(lb=lb, is_=None)
    else:
        lb = _fobst(np.arange(n // 2), n)
        lb[0] = 0.0
        p.update(lb=lb, is_=np.arange(n // 2, dtype=np.int32))
    return p


def ex3_primal(n=100):
    """src/tutorials/ex3.c:32-150: ex1's QP with the bound written as the inequality -I x <= -c
    (B = -I, cI = -c, ex3.c:118-120).  Returned in primal form; dualisation is the caller's job
    (QPTDualize, src/qp/interface/qptransform.c:909-1197)."""
    p = ex1(n)
    c = p.pop("lb")
    p.update(BI_diag=-np.ones(n), cI=-c)
    return p


def jbearing2(nx, ny, ecc=0.1, b=10.0):
    """Journal-bearing QP (MINPACK-2 DPJB) of src/tutorials/jbearing2.c: Hessian FormHessian :349-482,
    linear term ComputeB :191-232, QP set-up :511-514 (QPSetRhsPlus => rhs = -B), bounds 0 <= x <= 1000
    (:151-152), x0 = 0.  Natural ordering row = j*nx + i (single-rank DMDA)."""
    hx = 2.0 * (4.0 * np.arctan(1.0)) / (nx + 1.0)
    hy = 2.0 * b / (ny + 1.0)
    hxhy = hx * hy
    hxhx = 1.0 / (hx * hx)
    hyhy = 1.0 / (hy * hy)

    def p(xi):
        t = 1.0 + ecc * np.cos(xi)
        return t * t * t

    n = nx * ny
    rowptr = np.zeros(n + 1, dtype=np.int32)
    col, val = [], []
    six = 6.0
    rows = {}
    for i in range(nx):
        xi = (i + 1) * hx
        trule1 = hxhy * (p(xi) + p(xi + hx) + p(xi)) / six
        trule2 = hxhy * (p(xi) + p(xi - hx) + p(xi)) / six
        trule3 = hxhy * (p(xi) + p(xi + hx) + p(xi + hx)) / six
        trule4 = hxhy * (p(xi) + p(xi - hx) + p(xi - hx)) / six
        trule5 = trule1
        trule6 = trule2
        vdown = -(trule5 + trule2) * hyhy
        vleft = -hxhx * (trule2 + trule4)
        vright = -hxhx * (trule1 + trule3)
        vup = -hyhy * (trule1 + trule6)
        vmiddle = hxhx * (trule1 + trule2 + trule3 + trule4) + hyhy * (trule1 + trule2 + trule5 + trule6)
        for j in range(ny):
            row = j * nx + i
            c, v = [], []
            if j > 0:
                c.append(row - nx)
                v.append(vdown)
            if i > 0:
                c.append(row - 1)
                v.append(vleft)
            c.append(row)
            v.append(vmiddle)
            if i + 1 < nx:
                c.append(row + 1)
                v.append(vright)
            if j + 1 < ny:
                c.append(row + nx)
                v.append(vup)
            rows[row] = (c, v)
    for r in range(n):
        c, v = rows[r]
        col.extend(c)
        val.extend(v)
        rowptr[r + 1] = len(col)
    ehxhy = ecc * hx * hy
    B = np.empty(n)
    for i in range(nx):
        temp = np.sin((i + 1) * hx)
        for j in range(ny):
            B[nx * j + i] = -ehxhy * temp
    return dict(n=n, rowptr=rowptr, col=np.asarray(col, dtype=np.int32), val=np.asarray(val, dtype=np.float64), b=-B,
                lb=np.zeros(n), ub=np.full(n, 1000.0), x0=np.zeros(n))


def laplace2d_csr(nx, ny):
    """5-point Laplacian (4 on the diagonal, -1 off) on an nx x ny interior grid with homogeneous
    Dirichlet boundary eliminated (SPD).  Row r = j*nx + i; columns ascending.  Vectorised: fits the
    10 M-row config of BASELINE.json configs[1] (nx = ny = 3162 -> n = 9 998 244, nnz = 49 978 572)."""
    n = nx * ny
    r = np.arange(n, dtype=np.int64)
    i = r % nx
    j = r // nx
    has = [j > 0, i > 0, np.ones(n, dtype=bool), i < nx - 1, j < ny - 1]
    offs = [-nx, -1, 0, 1, nx]
    vals = [-1.0, -1.0, 4.0, -1.0, -1.0]
    cnt = np.zeros(n, dtype=np.int64)
    for h in has:
        cnt += h
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(cnt, out=rowptr[1:])
    nnz = int(rowptr[-1])
    col = np.empty(nnz, dtype=np.int32)
    val = np.empty(nnz, dtype=np.float64)
    pos = rowptr[:-1].copy()
    for h, o, v in zip(has, offs, vals):
        idx = pos[h]
        col[idx] = (r[h] + o).astype(np.int32)
        val[idx] = v
        pos[h] += 1
    return rowptr.astype(np.int32), col, val


def laplace2d_box(nx, ny, variant="obstacle", seed=20260515):
    """BASELINE.json configs[1]: synthetic SPD 5-pt Laplacian QP with box constraints.

    variant "obstacle": b = -15 h^2 * 2 (ex1's load scaled the same way), lb = sin(4 pi x - pi/6) *
    sin(4 pi y - pi/6)/2 - 2 (2-D analogue of ex1's obstacle), ub = +inf (None).
    variant "twosided": lb = -1, ub = +1, b standard normal (numpy default_rng(seed)) - exercises both bounds.
    """
    n = nx * ny
    rowptr, col, val = laplace2d_csr(nx, ny)
    hx, hy = 1.0 / (nx + 1), 1.0 / (ny + 1)
    r = np.arange(n, dtype=np.int64)
    x = ((r % nx) + 1) * hx
    y = ((r // nx) + 1) * hy
    if variant == "obstacle":
        b = np.full(n, -15.0 * hx * hy * 2)
        lb = np.sin(4 * np.pi * x - np.pi / 6.0) * np.sin(4 * np.pi * y - np.pi / 6.0) / 2 - 2
        ub = None
    elif variant == "twosided":
        rng = np.random.default_rng(seed)
        b = rng.standard_normal(n)
        lb = np.full(n, -1.0)
        ub = np.full(n, 1.0)
    else:
        raise ValueError(variant)
    return dict(n=n, rowptr=rowptr, col=col, val=val, b=b, lb=lb, ub=ub, x0=np.zeros(n))


def svm_dual(N, d=64, C=1.0, seed_x=7, seed_w=8):
    """BASELINE.json configs[4]: PermonSVM-style hinge-loss dual without bias term,
    min 1/2 a'Ha - 1'a, 0 <= a <= C, H = diag(y) X X' diag(y) applied matrix-free.
    X in R^{N x d} i.i.d. N(0,1) (default_rng(seed_x)), w* ~ N(0,1) (default_rng(seed_w)), y = sign(X w* + 0.1 N(0,1))."""
    rng = np.random.default_rng(seed_x)
    X = rng.standard_normal((N, d))
    w = np.random.default_rng(seed_w).standard_normal(d)
    y = np.sign(X @ w + 0.1 * rng.standard_normal(N))
    y[y == 0] = 1.0
    return dict(n=N, d=d, X=X, y=y, b=np.ones(N), lb=np.zeros(N), ub=np.full(N, float(C)), x0=np.zeros(N))


def write_contact_problem(path, f):
    """A CubeFeti contact problem (permon_amd.feti.CubeFeti) in the binary layout examples/contact_tfeti.c reads:
    everything pmh_feti_contact_solve takes -- block-diagonal K, f, B as leaves (equality rows first), c, R, the node boxes."""
    K = f.K.tocsr()
    K.sort_indices()
    nn = f.nel + 1
    with open(path, "wb") as fh:
        np.array([0x504D4831, f.nsub, f.N, K.nnz, f.n_lambda, f.n_eq, f.leaves_row.size, f.kdim], dtype=np.int32).tofile(fh)
        np.array([f.ndof], dtype=np.int32).tofile(fh)
        np.array([nn, nn, nn] * f.nsub, dtype=np.int32).tofile(fh)
        for a in (f.block_rowstart, K.indptr, K.indices, f.leaves_row, f.leaves_root):
            np.ascontiguousarray(a, dtype=np.int32).tofile(fh)
        for a in (K.data, f.f, f.leaves_sign, f.c, f.R):
            np.ascontiguousarray(a, dtype=np.float64).tofile(fh)
