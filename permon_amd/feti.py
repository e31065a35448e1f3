"""Host-side producers of the FETI hot-path inputs for structured cube decompositions (numpy / scipy).

What the reference computes once on the CPU before the QPS loop starts -- the per-subdomain stiffness
blocks, the signed gluing B (values/signs as QPFetiGetBgtSF, src/qp/impls/feti/qpfeti.c:786-821), the
kernel R, G = R'B', e = R'f, the dual right-hand side and bounds (QPTDualize,
src/qp/interface/qptransform.c:1102-1174), the homogenisation (QPTHomogenizeEq :464-518) and the
projected QP (QPTEnforceEqByProjector :272-298) -- restated for Q1 elements on unit cubes so that
BASELINE.json configs[2]/[3] (3-D elasticity TFETI with a rigid obstacle) can be generated at any size.
Everything here is set-up; the iteration runs in libpermonhip.
"""
import numpy as np
import scipy.sparse as sp

__all__ = ["q1_elasticity_element", "q1_poisson_element", "gluing_links", "CubeFeti", "DmdaFeti", "box_mg_hierarchy"]


def _gauss_q1():
    g = 1.0 / np.sqrt(3.0)
    pts = [(a, b, c) for a in (-g, g) for b in (-g, g) for c in (-g, g)]
    nodes = np.array([(a, b, c) for c in (-1, 1) for b in (-1, 1) for a in (-1, 1)], dtype=float)  # x fastest
    return pts, nodes


def _dshape(xi, nodes):
    # dN_a/dxi_j on the reference cube [-1,1]^3
    d = np.empty((8, 3))
    for a in range(8):
        s = nodes[a]
        d[a, 0] = s[0] * (1 + s[1] * xi[1]) * (1 + s[2] * xi[2]) / 8
        d[a, 1] = s[1] * (1 + s[0] * xi[0]) * (1 + s[2] * xi[2]) / 8
        d[a, 2] = s[2] * (1 + s[0] * xi[0]) * (1 + s[1] * xi[1]) / 8
    return d


def q1_poisson_element(h):
    """8x8 Laplace stiffness of a Q1 hexahedron of edge h (2x2x2 Gauss)."""
    pts, nodes = _gauss_q1()
    Ke = np.zeros((8, 8))
    for xi in pts:
        dN = _dshape(xi, nodes) * (2.0 / h)
        Ke += dN @ dN.T * (h / 2.0) ** 3
    return Ke


def q1_elasticity_element(h, E=1.0, nu=0.3):
    """24x24 linear-elasticity stiffness of a Q1 hexahedron of edge h; dof = 3*node + component
    (same element the reference tabulates as elast_3D_emat, src/tutorials/feti/ex71.c:35, computed here
    by quadrature instead of copied)."""
    pts, nodes = _gauss_q1()
    lam = E * nu / ((1 + nu) * (1 - 2 * nu))
    mu = E / (2 * (1 + nu))
    D = np.zeros((6, 6))
    D[:3, :3] = lam
    D[np.arange(3), np.arange(3)] += 2 * mu
    D[3:, 3:] = np.eye(3) * mu
    Ke = np.zeros((24, 24))
    for xi in pts:
        dN = _dshape(xi, nodes) * (2.0 / h)
        Bm = np.zeros((6, 24))
        for a in range(8):
            dx, dy, dz = dN[a]
            Bm[0, 3 * a] = dx
            Bm[1, 3 * a + 1] = dy
            Bm[2, 3 * a + 2] = dz
            Bm[3, 3 * a], Bm[3, 3 * a + 1] = dy, dx
            Bm[4, 3 * a + 1], Bm[4, 3 * a + 2] = dz, dy
            Bm[5, 3 * a], Bm[5, 3 * a + 2] = dz, dx
        Ke += Bm.T @ D @ Bm * (h / 2.0) ** 3
    return 0.5 * (Ke + Ke.T)


def gluing_links(m, gtype="full", scale=True):
    """Links (dual rows) that glue the m copies of one interface dof, copies ordered by rank: a list of
    [(copy, value), ...] rows as QPFetiGetBgtSF assembles them (src/qp/impls/feti/qpfeti.c).

    nonred  m-1 links, copy 0 against copy k (star around the lowest rank, :643-648); full  all m(m-1)/2 pairs
    (i, k>i) (:650-657); both +1 on the lower rank, -1 on the highest rank of the link, times 1/sqrt(m) with
    -SCALE_ON, the default (:786-806).  orth  m-1 orthonormal links, link k couples copies 0..d-1 (value 1/d)
    with copy d = m-1-k (value -1), all divided by sqrt(1/d + 1) (:659-667, :700-716, :807-817)."""
    if m < 2:
        return []
    if gtype == "orth":
        rows = []
        for k in range(m - 1):
            d = m - 1 - k
            x = np.sqrt(1.0 / d + 1.0)
            rows.append([(t, 1.0 / d / x) for t in range(d)] + [(d, -1.0 / x)])
        return rows
    sc = 1.0 / np.sqrt(m) if scale else 1.0
    if gtype == "nonred":
        return [[(0, sc), (k, -sc)] for k in range(1, m)]
    if gtype == "full":
        return [[(i, sc), (k, -sc)] for i in range(m - 1) for k in range(i + 1, m)]
    raise ValueError("unknown FETI gluing type %r" % gtype)  # qpfeti.c:561


def gluing_from_l2g(l2g_list, gtype="full", scale=True, exclude=None):
    """QPFetiGetBgtSF (src/qp/impls/feti/qpfeti.c:465-925) for a decomposition given by the subdomains' local-to-global dof maps:
    returns (leaves_row, leaves_root, leaves_val, n_lambda) with leaves_row indexing the concatenated local numbering.
    Thin wrapper of pmh_feti_gluing_from_l2g (csrc/regularize.hip); exclude: global dofs left out of the gluing."""
    import ctypes as C

    from . import _lib

    L = _lib.load()
    tcode = {"nonred": 0, "full": 1, "orth": 2}
    if gtype not in tcode:
        raise ValueError("unknown FETI gluing type %r" % gtype)  # qpfeti.c:561
    start = np.concatenate([[0], np.cumsum([len(g) for g in l2g_list])]).astype(np.int32)
    cat = np.ascontiguousarray(np.concatenate(l2g_list) if len(l2g_list) else np.zeros(0), dtype=np.int32)
    ex = np.ascontiguousarray(np.unique(exclude), dtype=np.int32) if exclude is not None and len(exclude) else np.zeros(0, dtype=np.int32)
    nl, nleaf = C.c_int(), C.c_int()
    args = (len(l2g_list), start.ctypes.data_as(C.c_void_p), cat.ctypes.data_as(C.c_void_p), tcode[gtype], int(bool(scale)), ex.size, ex.ctypes.data_as(C.c_void_p) if ex.size else None)
    _lib.check(L.pmh_feti_gluing_from_l2g(*args, C.byref(nl), C.byref(nleaf), None, None, None))
    rows, roots, vals = np.zeros(nleaf.value, dtype=np.int32), np.zeros(nleaf.value, dtype=np.int32), np.zeros(nleaf.value)
    _lib.check(L.pmh_feti_gluing_from_l2g(*args, C.byref(nl), C.byref(nleaf), rows.ctypes.data_as(C.c_void_p), roots.ctypes.data_as(C.c_void_p), vals.ctypes.data_as(C.c_void_p)))
    return rows, roots, vals, nl.value


class CubeFeti:
    """TFETI data for sx x sy x sz unit cubes of nel^3 Q1 elements each.

    physics 'elasticity' (3 dof/node, 6 rigid-body modes per cube) or 'poisson' (1 dof/node, 1 mode).
    Dirichlet u = 0 on the global x = 0 face enforced by B (TFETI: every subdomain floats,
    KSPFETISetDirichlet(..., FETI_LOCAL, PETSC_TRUE) as in src/tutorials/feti/ex1.c:89-90);
    gluing 'nonred' | 'full' | 'orth' as gluing_links();
    contact=True adds the rigid obstacle below the global z = 0 face: -u_z <= gap (inequality rows).
    Dual rows are ordered [Dirichlet | gluing | contact]; the first n_eq are equalities.
    """

    def __init__(self, sub=(2, 2, 2), nel=3, physics="elasticity", gluing="full", scale=True, contact=True, gap0=0.0, gap_slope=0.05, load=-1.0, young=None, graded=None):
        """young: None (one material: every subdomain has the SAME stiffness matrix, the congruent case) or one Young's modulus per subdomain --
        a heterogeneous body whose subdomain matrices K_s = E_s K_1 all differ (no two blocks are bit-identical: pmh_csr_block_classes finds nsub
        classes), the general, non-congruent case of the explicit dual operators."""
        self.sub, self.nel, self.physics = tuple(sub), int(nel), physics
        self.ndof = 3 if physics == "elasticity" else 1
        self.kdim = 6 if physics == "elasticity" else 1
        sx, sy, sz = self.sub
        ne = self.nel
        nn1 = ne + 1
        h = 1.0 / ne
        self.nsub = sx * sy * sz
        nloc = nn1 ** 3 * self.ndof
        self.n_i = nloc
        self.N = nloc * self.nsub
        self.block_rowstart = np.arange(self.nsub + 1, dtype=np.int32) * nloc
        Ke = q1_elasticity_element(h) if physics == "elasticity" else q1_poisson_element(h)
        nd = self.ndof

        # one subdomain stiffness (all cubes are congruent)
        ix, iy, iz = np.meshgrid(np.arange(ne), np.arange(ne), np.arange(ne), indexing="ij")
        e0 = (iz.ravel() * nn1 + iy.ravel()) * nn1 + ix.ravel()
        offs = np.array([(c * nn1 + b) * nn1 + a for c in (0, 1) for b in (0, 1) for a in (0, 1)])
        enodes = e0[:, None] + offs[None, :]
        edofs = (enodes[:, :, None] * nd + np.arange(nd)[None, None, :]).reshape(len(e0), 8 * nd)
        rows = np.repeat(edofs, 8 * nd, axis=1).ravel()
        cols = np.tile(edofs, (1, 8 * nd)).ravel()
        vals = np.tile(Ke.ravel(), len(e0))
        Ki = sp.coo_matrix((vals, (rows, cols)), shape=(nloc, nloc)).tocsr()
        Ki.sum_duplicates()
        Ki.sort_indices()
        self.Ki = Ki
        self._K = None
        self.young = None if young is None else np.asarray(young, dtype=np.float64)
        if self.young is not None and self.young.size != self.nsub:
            raise ValueError("young: one modulus per subdomain")
        # graded: {subdomain: E(x, y, z)} -- a modulus that varies from element to element inside the subdomain (x, y, z: the element centre in the unit cube of the subdomain):
        # such a block is congruent to no other one AND invariant under none of the cube's symmetries (the set-up of its explicit operator has nothing to lean on)
        self.graded = dict(graded) if graded else {}
        if self.graded and self.young is None:
            self.young = np.ones(self.nsub)
        self._conn = (rows, cols, Ke, (ix.ravel() + 0.5) * h, (iy.ravel() + 0.5) * h, (iz.ravel() + 0.5) * h) if self.graded else None

        # body force: constant `load` in the last component (z for elasticity), consistent Q1 load vector
        fe = np.zeros(nloc)
        w = np.zeros(nn1 ** 3)
        np.add.at(w, enodes.ravel(), h ** 3 / 8.0)
        fe[(nd - 1)::nd] = load * w
        self.f = np.tile(fe, self.nsub)

        # coordinates and kernel basis (block-wise orthonormal)
        gz, gy, gx = np.meshgrid(np.arange(nn1), np.arange(nn1), np.arange(nn1), indexing="ij")
        lx, ly, lz = gx.ravel() * h, gy.ravel() * h, gz.ravel() * h
        self.R = np.zeros((self.kdim, self.N))
        self.coords = np.zeros((self.nsub, nn1 ** 3, 3))
        for s in range(self.nsub):
            ox, oy, oz = s % sx, (s // sx) % sy, s // (sx * sy)
            X, Y, Z = lx + ox, ly + oy, lz + oz
            self.coords[s] = np.stack([X, Y, Z], axis=1)
            if nd == 1:
                Rs = np.ones((nloc, 1))
            else:
                Rs = np.zeros((nloc, 6))
                Rs[0::3, 0] = 1
                Rs[1::3, 1] = 1
                Rs[2::3, 2] = 1
                Rs[0::3, 3], Rs[1::3, 3] = -Y, X
                Rs[1::3, 4], Rs[2::3, 4] = -Z, Y
                Rs[0::3, 5], Rs[2::3, 5] = Z, -X
            Q, _ = np.linalg.qr(Rs)
            self.R[:, s * nloc:(s + 1) * nloc] = Q.T

        # ---- constraints as leaves (primal dof, dual row, value) --------------------------------------
        rows_l, roots_l, vals_l = [], [], []
        c_rhs = []
        nrow = 0
        GX, GY, GZ = sx * ne + 1, sy * ne + 1, sz * ne + 1
        copies = {}
        for s in range(self.nsub):
            ox, oy, oz = s % sx, (s // sx) % sy, s // (sx * sy)
            gid = ((gz.ravel() + oz * ne) * GY + (gy.ravel() + oy * ne)) * GX + (gx.ravel() + ox * ne)
            for ln, g in enumerate(gid):
                copies.setdefault(int(g), []).append((s, ln))
        # Dirichlet rows on x = 0
        for g in sorted(copies):
            if g % GX == 0:
                for (s, ln) in copies[g]:
                    for c in range(nd):
                        rows_l.append(s * nloc + ln * nd + c)
                        roots_l.append(nrow)
                        vals_l.append(1.0)
                        c_rhs.append(0.0)
                        nrow += 1
        self.n_dirichlet = nrow
        # gluing rows
        for g in sorted(copies):
            cp = copies[g]
            m = len(cp)
            if m < 2:
                continue
            for link in gluing_links(m, gluing, scale):
                for c in range(nd):
                    for t, v in link:
                        sa, la = cp[t]
                        rows_l.append(sa * nloc + la * nd + c)
                        roots_l.append(nrow)
                        vals_l.append(v)
                    c_rhs.append(0.0)
                    nrow += 1
        self.n_eq = nrow
        # contact rows on the global z = 0 face: -u_last <= gap(x,y)
        if contact:
            for g in sorted(copies):
                if g // (GX * GY) == 0:
                    for (s, ln) in copies[g]:
                        x_, y_ = self.coords[s][ln, 0], self.coords[s][ln, 1]
                        rows_l.append(s * nloc + ln * nd + (nd - 1))
                        roots_l.append(nrow)
                        vals_l.append(-1.0)
                        c_rhs.append(gap0 + gap_slope * (x_ + y_))
                        nrow += 1
        self.n_lambda = nrow
        self.n_ineq = nrow - self.n_eq
        self.leaves_row = np.asarray(rows_l, dtype=np.int32)
        self.leaves_root = np.asarray(roots_l, dtype=np.int32)
        self.leaves_sign = np.asarray(vals_l, dtype=np.float64)
        self.c = np.asarray(c_rhs)
        self.B = sp.csr_matrix((self.leaves_sign, (self.leaves_root, self.leaves_row)), shape=(self.n_lambda, self.N))
        # dual bounds: lb(E) = -inf, lb(I) = 0 (qptransform.c:1136-1162)
        self.lb = np.concatenate([np.full(self.n_eq, -np.inf), np.zeros(self.n_ineq)])

    @property
    def K(self):
        """blockdiag(K_i) of all subdomains (built on demand: 158 M non-zeros at configs[2] size)."""
        if self._K is None:
            self._K = csr_block_diag([self.block_K(s) for s in range(self.nsub)])
            self._K.sort_indices()
        return self._K

    def block_K(self, s):
        """Stiffness matrix of subdomain s (E_s K_1 for a heterogeneous body; the one shared matrix object otherwise)."""
        if self.young is None:
            return self.Ki
        if s in self.graded:
            rows, cols, Ke, cx, cy, cz = self._conn
            Ee = float(self.young[s]) * np.asarray(self.graded[s](cx, cy, cz), dtype=np.float64)
            Ks = sp.coo_matrix(((Ee[:, None] * Ke.ravel()[None, :]).ravel(), (rows, cols)), shape=self.Ki.shape).tocsr()
            Ks.sum_duplicates()
            Ks.sort_indices()
            return Ks
        Ks = self.Ki.copy()
        Ks.data = Ks.data * float(self.young[s])
        return Ks

    @property
    def congruent(self):
        return self.young is None or (not self.graded and bool(np.all(self.young == self.young[0])))

    # ---- coarse space -----------------------------------------------------------------------------------
    def kernel_matrix(self):
        """R as a sparse N x (kdim*nsub) block-diagonal matrix."""
        nloc = self.n_i
        blocks = [sp.csr_matrix(self.R[:, s * nloc:(s + 1) * nloc].T) for s in range(self.nsub)]
        return sp.block_diag(blocks, format="csr")

    def coarse(self, orthonormalize=True):
        """G = R'B' (explicit, qptransform.c:838), e = R'f; optionally G <- L^{-1}G, e <- L^{-1}e with GG' = LL'
        (QPTOrthonormalizeEq) so that G has orthonormal rows."""
        Rm = self.kernel_matrix()
        G = (Rm.T @ self.B.T).tocsr()
        e = Rm.T @ self.f
        if orthonormalize:
            GGt = (G @ G.T).toarray()
            L = np.linalg.cholesky(GGt)
            Gd = np.linalg.solve(L, G.toarray())
            Gd[np.abs(Gd) < 1e-300] = 0.0
            G = sp.csr_matrix(Gd)
            e = np.linalg.solve(L, e)
        G.sort_indices()
        return G, e

    def subset(self, blocks):
        """Restriction to the given subdomain blocks (one rank's share): local K, f, R and the leaves whose
        primal dof lives there (dual numbering stays global: lambda is replicated)."""
        blocks = list(blocks)
        nloc = self.n_i
        keep = np.zeros(self.N, dtype=bool)
        newidx = -np.ones(self.N, dtype=np.int64)
        for k, s in enumerate(blocks):
            keep[s * nloc:(s + 1) * nloc] = True
            newidx[s * nloc:(s + 1) * nloc] = np.arange(k * nloc, (k + 1) * nloc)
        sel = keep[self.leaves_row]
        return dict(
            nblocks=len(blocks), block_rowstart=np.arange(len(blocks) + 1, dtype=np.int32) * nloc,
            K=csr_block_diag([self.block_K(s) for s in blocks]), f=self.f[keep], R=self.R[:, keep],
            leaves_row=newidx[self.leaves_row[sel]].astype(np.int32), leaves_root=self.leaves_root[sel], leaves_sign=self.leaves_sign[sel],
            n_x=len(blocks) * nloc, n_lambda=self.n_lambda)


def csr_block_diag(blocks):
    """blockdiag(blocks) of CSR matrices by concatenating their arrays (scipy.sparse.block_diag goes through COO: ~5 s for the 8 x 20 M non-zeros of configs[2], this is ~0.3 s).
    Rows keep their order and sorted indices."""
    blocks = [b.tocsr() for b in blocks]
    nnz = np.cumsum([0] + [b.nnz for b in blocks])
    rows = np.cumsum([0] + [b.shape[0] for b in blocks])
    cols = np.cumsum([0] + [b.shape[1] for b in blocks])
    wide = nnz[-1] > np.iinfo(np.int32).max or cols[-1] > np.iinfo(np.int32).max
    it = np.int64 if wide else np.int32
    indptr = np.concatenate([np.asarray(b.indptr[:-1], dtype=it) + it(nnz[i]) for i, b in enumerate(blocks)] + [np.asarray([nnz[-1]], dtype=it)])
    indices = np.concatenate([np.asarray(b.indices, dtype=it) + it(cols[i]) for i, b in enumerate(blocks)]) if len(blocks) else np.zeros(0, dtype=it)
    data = np.concatenate([b.data for b in blocks]) if len(blocks) else np.zeros(0)
    out = sp.csr_matrix((data, indices, indptr), shape=(int(rows[-1]), int(cols[-1])))
    out.has_sorted_indices = all(b.has_sorted_indices for b in blocks)
    return out


def _dmda_partition3(M, N, P, size):
    """Process grid DMDACreate3d picks for PETSC_DECIDE (the 'squarish' rule of DMSetUp_DA_3D)."""
    n = max(int(0.5 + ((N * N) * size / (P * M)) ** (1.0 / 3.0)), 1)
    while n > 0:
        pm = size // n
        if n * pm == size:
            break
        n -= 1
    n = max(n, 1)
    m = max(int(0.5 + np.sqrt(M * size / (P * n))), 1)
    p = 1
    while m > 0:
        p = size // (m * n)
        if m * n * p == size:
            break
        m -= 1
    if M > P and m < p:
        m, p = p, m
    return m, n, p


def _dmda_node_ranges(M, m):
    """Node range (inclusive) of each of the m element slabs along one direction: rank i owns M/m + (M%m > i)
    nodes and the elements whose upper node it owns (DMDAGetElements)."""
    own = [M // m + (1 if (M % m) > i else 0) for i in range(m)]
    s = np.cumsum([0] + own)
    return [((s[i] - 1 if i > 0 else 0), s[i + 1] - 1) for i in range(m)]


class DmdaFeti:
    """The problem of the reference's DMDA tutorial, src/tutorials/feti/ex71.c, as FETI hot-path inputs:
    Q1 Poisson / elasticity (lambda = mu = 1, the tabulated element matrices :16-126 recomputed by quadrature)
    on [0,cx] x [0,cy] x [0,cz] with unit cells, decomposed the way DMDACreate3d does for `size` ranks, x = 0 face
    fixed IN the subdomain matrices (MatZeroRowsColumnsIS, diagonal 1/multiplicity, :305-349), b = 1 split by
    multiplicity (QPTMatISToBlockDiag, qptransform.c:2095-2113), gluing by QPFetiGetBgtSF (gluing_links).
    Subdomains touching x = 0 are non-singular; the others float (1 or 6 kernel vectors)."""

    def __init__(self, cells=(7, 8, 9), size=6, physics="poisson", gluing="full", scale=True):
        M, N, P = (c + 1 for c in cells)
        self.physics, self.gluing = physics, gluing
        nd = self.ndof = 3 if physics == "elasticity" else 1
        kd = 6 if physics == "elasticity" else 1
        self.procs = _dmda_partition3(M, N, P, size)
        m, n, p = self.procs
        rx, ry, rz = _dmda_node_ranges(M, m), _dmda_node_ranges(N, n), _dmda_node_ranges(P, p)
        Ke = q1_elasticity_element(1.0, E=2.5, nu=0.25) if physics == "elasticity" else q1_poisson_element(1.0)
        lex = [(a, b, c) for c in (0, 1) for b in (0, 1) for a in (0, 1)]
        self.nsub = size
        Ks, gids, Rs, coords = [], [], [], []
        for kz in range(p):
            for jy in range(n):
                for ix in range(m):
                    (x0, x1), (y0, y1), (z0, z1) = rx[ix], ry[jy], rz[kz]
                    nx, ny, nz = x1 - x0 + 1, y1 - y0 + 1, z1 - z0 + 1
                    kk, jj, ii = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
                    ii, jj, kk = ii.ravel(), jj.ravel(), kk.ravel()
                    gids.append(((kk + z0) * N + (jj + y0)) * M + (ii + x0))
                    coords.append(np.stack([ii + x0, jj + y0, kk + z0], axis=1).astype(float))
                    ek, ej, ei = np.meshgrid(np.arange(nz - 1), np.arange(ny - 1), np.arange(nx - 1), indexing="ij")
                    e0 = (ek.ravel() * ny + ej.ravel()) * nx + ei.ravel()
                    offs = np.array([(c * ny + b) * nx + a for (a, b, c) in lex])
                    en = e0[:, None] + offs[None, :]
                    ed = (en[:, :, None] * nd + np.arange(nd)[None, None, :]).reshape(len(e0), 8 * nd)
                    Ki = sp.coo_matrix((np.tile(Ke.ravel(), len(e0)), (np.repeat(ed, 8 * nd, axis=1).ravel(), np.tile(ed, (1, 8 * nd)).ravel())),
                                       shape=(nx * ny * nz * nd,) * 2).tocsr()
                    Ks.append(Ki)
        mult = np.zeros(M * N * P, dtype=np.int64)
        for g in gids:
            mult[g] += 1
        fs = []
        for s, (Ki, g, X) in enumerate(zip(Ks, gids, coords)):
            nl = len(g)
            dn = np.nonzero(g % M == 0)[0]
            if len(dn):
                dd = (dn[:, None] * nd + np.arange(nd)[None, :]).ravel()
                keep = np.ones(nl * nd)
                keep[dd] = 0.0
                Dk = sp.diags(keep)
                fix = np.zeros(nl * nd)
                fix[dd] = 1.0 / np.repeat(mult[g[dn]], nd)
                Ki = (Dk @ Ki @ Dk + sp.diags(fix)).tocsr()
                Rs.append(np.zeros((0, nl * nd)))
            else:
                if nd == 1:
                    R = np.ones((nl, 1))
                else:
                    R = np.zeros((nl * 3, 6))
                    R[0::3, 0] = R[1::3, 1] = R[2::3, 2] = 1.0
                    R[0::3, 3], R[1::3, 3] = -X[:, 1], X[:, 0]
                    R[1::3, 4], R[2::3, 4] = -X[:, 2], X[:, 1]
                    R[0::3, 5], R[2::3, 5] = X[:, 2], -X[:, 0]
                Q, _ = np.linalg.qr(R)
                Rs.append(Q.T[:kd])
            Ki.sum_duplicates()
            Ki.eliminate_zeros()
            Ki.sort_indices()
            Ks[s] = Ki
            fs.append(np.repeat(1.0 / mult[g], nd))
        self.blocks, self.gids, self.Rblocks = Ks, gids, Rs
        self.block_rowstart = np.concatenate([[0], np.cumsum([K.shape[0] for K in Ks])]).astype(np.int32)
        self.N = int(self.block_rowstart[-1])
        self.f = np.concatenate(fs)
        # gluing: QPFetiGetBgtSF on the dof-level local-to-global maps (node-major dofs), pmh_feti_gluing_from_l2g (C++)
        l2g = [(np.asarray(g)[:, None] * nd + np.arange(nd)[None, :]).ravel() for g in gids]
        rows_l, roots_l, vals_l, nrow = gluing_from_l2g(l2g, gluing, scale)
        self.n_lambda = self.n_eq = nrow
        self.leaves_row = np.asarray(rows_l, dtype=np.int32)
        self.leaves_root = np.asarray(roots_l, dtype=np.int32)
        self.leaves_sign = np.asarray(vals_l, dtype=np.float64)
        self.B = sp.csr_matrix((self.leaves_sign, (self.leaves_root, self.leaves_row)), shape=(self.n_lambda, self.N))
        self.c = np.zeros(self.n_lambda)
        self.lb = np.full(self.n_lambda, -np.inf)
        self.kdim = max(R.shape[0] for R in Rs)
        # kernel as a dense (kdim x N) array, zero columns over the non-singular blocks (MatInv / MP projection input)
        self.R = np.zeros((self.kdim, self.N))
        for s, R in enumerate(Rs):
            self.R[:R.shape[0], self.block_rowstart[s]:self.block_rowstart[s + 1]] = R

    @property
    def K(self):
        K = sp.block_diag(self.blocks, format="csr")
        K.sort_indices()
        return K

    def coarse(self):
        """G = R'B' over the floating subdomains only (no zero rows), e = R'f; None, None if nothing floats."""
        rows = []
        for s, R in enumerate(self.Rblocks):
            for k in range(R.shape[0]):
                v = np.zeros(self.N)
                v[self.block_rowstart[s]:self.block_rowstart[s + 1]] = R[k]
                rows.append(v)
        if not rows:
            return None, None
        Rm = sp.csr_matrix(np.array(rows))
        G = (Rm @ self.B.T).tocsr()
        G.sort_indices()
        return G, Rm @ self.f

    def local(self):
        """All blocks on one rank, in the layout FetiDualQP consumes (CubeFeti.subset)."""
        return dict(nblocks=self.nsub, block_rowstart=self.block_rowstart, K=self.K, f=self.f, R=self.R if self.kdim and np.any(self.R) else None,
                    leaves_row=self.leaves_row, leaves_root=self.leaves_root, leaves_sign=self.leaves_sign, n_x=self.N, n_lambda=self.n_lambda)


# ---- multigrid hierarchy for structured box blocks (input of pmh_mg_create) -------------------------------------
def _interp1d(n):
    """Linear interpolation onto n grid nodes from the coarse nodes {0,2,4,...} U {n-1}: (n x nc) CSR, exact for
    linear functions also when the last coarse interval is short (n even)."""
    c = list(range(0, n, 2))
    if c[-1] != n - 1:
        c.append(n - 1)
    if len(c) == n:
        return sp.identity(n, format="csr")
    rows, cols, vals = [], [], []
    for j in range(len(c) - 1):
        a, b = c[j], c[j + 1]
        for i in range(a, b):
            t = (i - a) / float(b - a)
            rows.append(i), cols.append(j), vals.append(1.0 - t)
            if t > 0.0:
                rows.append(i), cols.append(j + 1), vals.append(t)
    rows.append(n - 1), cols.append(len(c) - 1), vals.append(1.0)
    return sp.csr_matrix((vals, (rows, cols)), shape=(n, len(c)))


def _lambda_max_dinv_a(A, its=20, seed=0):
    """Power-method estimate of lambda_max(D^-1 A) (what KSPChebyshev's eigen-estimate provides)."""
    d = A.diagonal()
    dinv = np.where(d != 0.0, 1.0 / np.where(d != 0.0, d, 1.0), 1.0)
    v = np.random.default_rng(seed).standard_normal(A.shape[0])
    lam = 1.0
    for _ in range(its):
        w = dinv * (A @ v)
        lam = np.linalg.norm(w) / max(np.linalg.norm(v), 1e-300)
        v = w / max(np.linalg.norm(w), 1e-300)
    return float(lam)


def box_mg_hierarchy(blocks, dims, ndof, min_nodes=400, max_levels=12):
    """Geometric multigrid hierarchy for a block-diagonal matrix whose blocks are Q1 discretisations on
    nx x ny x nz node boxes (node-major dof numbering, x fastest): trilinear prolongation P_l (x) I_ndof, Galerkin
    coarse operators A_{l+1} = P_l' A_l P_l.  Trilinear interpolation reproduces constants and rigid-body modes, so
    floating blocks stay consistently singular down to the coarsest level, where a dense pseudo-inverse is used.
    Coarsening stops at <= min_nodes nodes per block: a ~1000-dof dense coarse solve costs one small GEMV, whereas
    every further smoothed level costs ~11 latency-bound launches per cycle.
    blocks: list of scipy matrices; dims: list of (nx, ny, nz); congruent blocks (same object) are processed once.
    Returns dict(A=[...], P=[...], lambda_max=[...], coarse_rowstart, coarse_pinv) with block-concatenated matrices."""
    cache = {}
    per_block = []
    for Kb, dm in zip(blocks, dims):
        key = (id(Kb), tuple(dm))
        if key not in cache:
            A, P, lam = [Kb.tocsr()], [], []
            d = tuple(int(v) for v in dm)
            while len(A) < max_levels and d[0] * d[1] * d[2] > min_nodes and max(d) > 2:
                P1 = [_interp1d(n) for n in d]
                Pn = sp.kron(P1[2], sp.kron(P1[1], P1[0], format="csr"), format="csr")
                Pl = sp.kron(Pn, sp.identity(ndof, format="csr"), format="csr") if ndof > 1 else Pn
                Ac = (Pl.T @ A[-1] @ Pl).tocsr()
                Ac = 0.5 * (Ac + Ac.T)
                Ac.sort_indices()
                lam.append(_lambda_max_dinv_a(A[-1]))
                P.append(Pl)
                A.append(Ac.tocsr())
                d = tuple(p.shape[1] for p in P1)
            pinv = np.linalg.pinv(A[-1].toarray(), rcond=1e-10, hermitian=True)
            cache[key] = (A, P, lam, pinv)
        per_block.append(cache[key])
    nlev = min(len(pb[0]) for pb in per_block)
    out = dict(A=[], P=[], lambda_max=[])
    for l in range(nlev):
        # a block with a deeper private hierarchy is cut at the common depth: its level-l operator is re-inverted densely below
        Al = sp.block_diag([pb[0][l] for pb in per_block], format="csr")
        Al.sort_indices()
        out["A"].append(Al)
        if l + 1 < nlev:
            Pl = sp.block_diag([pb[1][l] for pb in per_block], format="csr")
            Pl.sort_indices()
            out["P"].append(Pl)
            out["lambda_max"].append(max(pb[2][l] for pb in per_block))
    sizes = [pb[0][nlev - 1].shape[0] for pb in per_block]
    out["coarse_rowstart"] = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    pinvs = []
    for pb in per_block:
        pinvs.append(pb[3] if len(pb[0]) == nlev else np.linalg.pinv(pb[0][nlev - 1].toarray(), rcond=1e-10, hermitian=True))
    out["coarse_pinv"] = np.concatenate([p.ravel() for p in pinvs]) if pinvs else np.zeros(0)
    return out


# ---- symmetries of a box block (set-up of the class-shared explicit operators by symmetry) -----------------------
def box_symmetries(dims, ndof, K=None, nsample=4000, tol=1e-11, seed=0):
    """Signed dof permutations of a block of nx x ny x nz nodes (x fastest) x ndof dofs per node induced by the signed coordinate
    permutations that map the box onto itself (a cube: the 48 elements of the hyperoctahedral group; ndof = 3: the components
    transform as a vector, ndof = 1: as a scalar).  K (the block's matrix): every GENERATOR (reflections, swaps of equal axes) is
    checked against K on `nsample` random rows (K[g i, g j] s_i s_j = K[i, j]) and dropped if it fails; the group is the closure of
    the surviving generators, so every operation returned is a composition of checked ones.  Returns (perm[nsym, n], sign[nsym, n])
    with operation 0 the identity: dof i goes to perm[g, i] with sign[g, i]."""
    import itertools

    nx, ny, nz = (int(d) for d in dims)
    n = nx * ny * nz * ndof
    nodes = np.arange(nx * ny * nz)
    ijk = np.stack([nodes % nx, (nodes // nx) % ny, nodes // (nx * ny)], axis=1)
    dim = (nx, ny, nz)

    def op(axes, flips):  # new coordinate a = (+-) old coordinate axes[a]
        if any(dim[axes[a]] != dim[a] for a in range(3)):
            return None
        new = np.empty_like(ijk)
        for a in range(3):
            src = ijk[:, axes[a]]
            new[:, a] = (dim[a] - 1 - src) if flips[a] < 0 else src
        node2 = new[:, 0] + nx * (new[:, 1] + ny * new[:, 2])
        perm = np.empty(n, dtype=np.int64)
        sign = np.ones(n, dtype=np.int8)
        if ndof == 3:  # component axes[a] of the old field becomes component a of the new one, with the flip's sign
            for a in range(3):
                perm[nodes * 3 + axes[a]] = node2 * 3 + a
                sign[nodes * 3 + axes[a]] = flips[a]
        else:
            for d in range(ndof):
                perm[nodes * ndof + d] = node2 * ndof + d
        return perm, sign

    gens = [op((0, 1, 2), f) for f in ((-1, 1, 1), (1, -1, 1), (1, 1, -1))] + [op(a, (1, 1, 1)) for a in ((1, 0, 2), (0, 2, 1), (2, 1, 0))]
    gens = [g for g in gens if g is not None]
    if K is not None:
        K = K.tocsr()
        rng = np.random.default_rng(seed)
        rows = rng.choice(n, size=min(nsample, n), replace=False)
        sub = K[rows].tocoo()
        ri, cj, v = rows[sub.row], sub.col, sub.data
        scale = np.abs(v).max() if v.size else 1.0
        ok = []
        for perm, sign in gens:
            w = np.asarray(K[perm[ri], perm[cj]]).ravel() * sign[ri] * sign[cj]
            if np.abs(w - v).max() <= tol * scale:
                ok.append((perm, sign))
        gens = ok
    ident = (np.arange(n, dtype=np.int64), np.ones(n, dtype=np.int8))
    import hashlib

    group, frontier = [ident], [ident]
    key = lambda g: hashlib.blake2b(g[0].tobytes() + g[1].tobytes(), digest_size=16).digest()  # noqa: E731
    seen = {key(ident)}
    while frontier:
        nxt = []
        for (p1, s1), (p2, s2) in itertools.product(frontier, gens):
            pc, sc = p2[p1], (s1 * s2[p1]).astype(np.int8)  # apply (p1, s1) first, then the generator
            k = key((pc, sc))
            if k not in seen:
                seen.add(k)
                group.append((pc, sc))
                nxt.append((pc, sc))
        frontier = nxt
    return np.stack([g[0] for g in group]).astype(np.int32), np.stack([g[1] for g in group]).astype(np.int8)


# ---- irregular decompositions (subdomains that are NOT boxes) ---------------------------------------------------------------
def irregular_partition(n, kind="staircase", w=None):
    """Element -> subdomain map elem_sub[iz, iy, ix] of a (2n)^3-element cube cut into 8 subdomains that are not boxes (input of MeshFeti).
    'staircase': the three cutting planes of the 2 x 2 x 2 decomposition move by one element in a checkerboard of steps w elements wide, so every interface
    is a staircase and no subdomain has a symmetry or a congruent partner; 'lshape': the 2 x 2 x 2 cubes with a quarter-size brick handed from every cube of
    the lower layer to its x / y neighbour and from every cube of the upper layer to the cube below it (L-shaped bodies); 'cubes': the plain 2 x 2 x 2 boxes."""
    n = int(n)
    m = 2 * n
    iz, iy, ix = np.meshgrid(np.arange(m), np.arange(m), np.arange(m), indexing="ij")
    if kind == "cubes":
        return ((ix >= n) + 2 * (iy >= n) + 4 * (iz >= n)).astype(np.int32)
    if kind == "staircase":
        w = max(2, n // 3) if w is None else int(w)
        xc = n + ((iy // w + iz // w) % 2)
        yc = n + ((ix // w + iz // w + 1) % 2)
        zc = n + ((ix // w + iy // w) % 2)
        return _repair_partition(((ix >= xc) + 2 * (iy >= yc) + 4 * (iz >= zc)).astype(np.int32))
    if kind == "lshape":
        a = max(1, n // 2)
        es = ((ix >= n) + 2 * (iy >= n) + 4 * (iz >= n)).astype(np.int32)
        lo = iz < n
        es[lo & (ix >= n) & (ix < n + a) & (iy < a)] = 0          # cube 1 -> cube 0
        es[lo & (iy >= n) & (iy < n + a) & (ix >= m - a)] = 1     # cube 3 -> cube 1
        es[lo & (ix >= n - a) & (ix < n) & (iy >= m - a)] = 3     # cube 2 -> cube 3
        es[lo & (iy >= n - a) & (iy < n) & (ix < a)] = 2          # cube 0 -> cube 2
        up = (iz >= n) & (iz < n + a)
        es[up & (ix < a) & (iy < a)] = 0                          # cube 4 -> cube 0 ... a brick of every upper cube goes down
        es[up & (ix >= m - a) & (iy < a)] = 1
        es[up & (ix < a) & (iy >= m - a)] = 2
        es[up & (ix >= m - a) & (iy >= m - a)] = 3
        return es
    raise ValueError("unknown partition kind %r" % kind)


def _face_components(es, s):
    """Face-connected components of subdomain s of an element map: (element ids, component label per element, number of components)."""
    from scipy.sparse.csgraph import connected_components

    esf = es.ravel()
    eid = np.arange(es.size).reshape(es.shape)
    pairs = [(eid[:, :, :-1].ravel(), eid[:, :, 1:].ravel()), (eid[:, :-1, :].ravel(), eid[:, 1:, :].ravel()), (eid[:-1, :, :].ravel(), eid[1:, :, :].ravel())]
    el = np.nonzero(esf == s)[0]
    loc = -np.ones(es.size, dtype=np.int64)
    loc[el] = np.arange(el.size)
    ii = np.concatenate([loc[a][(esf[a] == s) & (esf[b] == s)] for a, b in pairs])
    jj = np.concatenate([loc[b][(esf[a] == s) & (esf[b] == s)] for a, b in pairs])
    ncomp, lab = connected_components(sp.coo_matrix((np.ones(ii.size), (ii, jj)), shape=(el.size, el.size)), directed=False)
    return el, lab, ncomp


def _repair_partition(es):
    """Hands every element outside the largest face-connected piece of its subdomain to the subdomain most of its face neighbours belong to (a subdomain held
    together by an edge or a vertex would be a mechanism: more than 6 kernel vectors)."""
    es = es.copy()
    shp = es.shape
    for _ in range(8):
        changed = False
        for s in range(int(es.max()) + 1):
            el, lab, ncomp = _face_components(es, s)
            if ncomp <= 1:
                continue
            big = np.argmax(np.bincount(lab))
            for e in el[lab != big]:
                k, j, i = np.unravel_index(e, shp)
                votes = {}
                for dk, dj, di in ((0, 0, 1), (0, 0, -1), (0, 1, 0), (0, -1, 0), (1, 0, 0), (-1, 0, 0)):
                    kk, jj, ii = k + dk, j + dj, i + di
                    if 0 <= kk < shp[0] and 0 <= jj < shp[1] and 0 <= ii < shp[2] and es[kk, jj, ii] != s:
                        votes[int(es[kk, jj, ii])] = votes.get(int(es[kk, jj, ii]), 0) + 1
                if votes:
                    es[k, j, i] = max(sorted(votes), key=lambda t: votes[t])
                    changed = True
        if not changed:
            break
    return es


class MeshFeti:
    """TFETI data for a box of nex x ney x nez Q1 elements of edge h cut into ARBITRARY (face-connected) subdomains: elem_sub[iz, iy, ix] names the subdomain of
    every element.  Same conventions as CubeFeti (Dirichlet u = 0 on the global x = 0 face enforced by B, rigid obstacle under the global z = 0 face as
    inequality rows, dual rows ordered [Dirichlet | gluing | contact]) but nothing is assumed about the subdomains' shape: the local numbering of a subdomain is
    its nodes in ascending global order (node-major dofs), the gluing comes from the subdomains' local-to-global maps through QPFetiGetBgtSF's rules
    (pmh_feti_gluing_from_l2g), the kernel basis from the node coordinates.  This is the decomposition a mesh partitioner hands the reference
    (QPTMatISToBlockDiag + QPFetiSetUp): blocks with no box structure, no symmetry and no congruent partner."""

    def __init__(self, elem_sub, h=None, physics="elasticity", gluing="full", scale=True, contact=True, gap0=0.0, gap_slope=0.05, load=-1.0, young=None):
        es = np.asarray(elem_sub, dtype=np.int64)
        nez, ney, nex = es.shape
        self.elem_sub = es
        self.physics = physics
        nd = self.ndof = 3 if physics == "elasticity" else 1
        self.kdim = 6 if physics == "elasticity" else 1
        h = 2.0 / max(nex, ney, nez) if h is None else float(h)
        self.h = h
        self.nsub = int(es.max()) + 1
        GX, GY, GZ = nex + 1, ney + 1, nez + 1
        Ke = q1_elasticity_element(h) if physics == "elasticity" else q1_poisson_element(h)
        self.young = None if young is None else np.asarray(young, dtype=np.float64)
        ez, ey, ex = np.meshgrid(np.arange(nez), np.arange(ney), np.arange(nex), indexing="ij")
        e0 = (ez.ravel() * GY + ey.ravel()) * GX + ex.ravel()  # lowest global node of every element
        offs = np.array([(c * GY + b) * GX + a for c in (0, 1) for b in (0, 1) for a in (0, 1)])
        esf = es.ravel()
        self.blocks, self.gnodes, self.coords, fs, Rs = [], [], [], [], []
        for s in range(self.nsub):
            el, _, ncomp = _face_components(es, s)  # (a subdomain hanging together by an edge or a vertex only would have more than 6 kernel vectors)
            if el.size == 0:
                raise ValueError("subdomain %d has no element" % s)
            if ncomp != 1:
                raise ValueError("subdomain %d is not face-connected (%d pieces)" % (s, ncomp))
            gn = e0[el][:, None] + offs[None, :]
            nodes = np.unique(gn)
            ln = np.searchsorted(nodes, gn)
            nl = nodes.size
            ed = (ln[:, :, None] * nd + np.arange(nd)[None, None, :]).reshape(el.size, 8 * nd)
            Es = 1.0 if self.young is None else float(self.young[s])
            Ks = sp.coo_matrix((np.tile(Es * Ke.ravel(), el.size), (np.repeat(ed, 8 * nd, axis=1).ravel(), np.tile(ed, (1, 8 * nd)).ravel())), shape=(nl * nd, nl * nd)).tocsr()
            Ks.sum_duplicates()
            Ks.sort_indices()
            X = np.stack([(nodes % GX) * h, ((nodes // GX) % GY) * h, (nodes // (GX * GY)) * h], axis=1)
            wgt = np.zeros(nl)
            np.add.at(wgt, ln.ravel(), h ** 3 / 8.0)
            f = np.zeros(nl * nd)
            f[(nd - 1)::nd] = load * wgt
            if nd == 1:
                Rb = np.ones((nl, 1))
            else:
                Rb = np.zeros((nl * 3, 6))
                Rb[0::3, 0] = Rb[1::3, 1] = Rb[2::3, 2] = 1.0
                Rb[0::3, 3], Rb[1::3, 3] = -X[:, 1], X[:, 0]
                Rb[1::3, 4], Rb[2::3, 4] = -X[:, 2], X[:, 1]
                Rb[0::3, 5], Rb[2::3, 5] = X[:, 2], -X[:, 0]
            Q, _ = np.linalg.qr(Rb)
            self.blocks.append(Ks), self.gnodes.append(nodes), self.coords.append(X), fs.append(f), Rs.append(Q.T)
        self.block_rowstart = np.concatenate([[0], np.cumsum([K.shape[0] for K in self.blocks])]).astype(np.int32)
        self.N = int(self.block_rowstart[-1])
        self.f = np.concatenate(fs)
        self.R = np.zeros((self.kdim, self.N))
        for s, Rb in enumerate(Rs):
            self.R[:, self.block_rowstart[s]:self.block_rowstart[s + 1]] = Rb
        self.l2g = [(g[:, None] * nd + np.arange(nd)[None, :]).ravel() for g in self.gnodes]
        # ---- constraints: [Dirichlet on x = 0 | gluing | contact on z = 0] ----
        rows_l, roots_l, vals_l, c_rhs = [], [], [], []
        nrow = 0
        for s, g in enumerate(self.gnodes):
            dn = np.nonzero(g % GX == 0)[0]
            dd = (dn[:, None] * nd + np.arange(nd)[None, :]).ravel() + self.block_rowstart[s]
            rows_l.append(dd), roots_l.append(nrow + np.arange(dd.size)), vals_l.append(np.ones(dd.size))
            nrow += dd.size
        self.n_dirichlet = nrow
        gr, gt, gv, ng = gluing_from_l2g(self.l2g, gluing, scale)
        rows_l.append(np.asarray(gr)), roots_l.append(np.asarray(gt) + nrow), vals_l.append(np.asarray(gv))
        nrow += ng
        self.n_eq = nrow
        c_rhs.append(np.zeros(nrow))
        if contact:
            for s, g in enumerate(self.gnodes):
                cn = np.nonzero(g // (GX * GY) == 0)[0]
                rows_l.append(cn * nd + (nd - 1) + self.block_rowstart[s]), roots_l.append(nrow + np.arange(cn.size)), vals_l.append(-np.ones(cn.size))
                c_rhs.append(gap0 + gap_slope * (self.coords[s][cn, 0] + self.coords[s][cn, 1]))
                nrow += cn.size
        self.n_lambda = nrow
        self.n_ineq = nrow - self.n_eq
        self.leaves_row = np.concatenate(rows_l).astype(np.int32)
        self.leaves_root = np.concatenate(roots_l).astype(np.int32)
        self.leaves_sign = np.concatenate(vals_l).astype(np.float64)
        self.c = np.concatenate(c_rhs)
        self.B = sp.csr_matrix((self.leaves_sign, (self.leaves_root, self.leaves_row)), shape=(self.n_lambda, self.N))
        self.lb = np.concatenate([np.full(self.n_eq, -np.inf), np.zeros(self.n_ineq)])
        self._K = None

    @property
    def K(self):
        if self._K is None:
            self._K = csr_block_diag(self.blocks)
        return self._K

    def block_K(self, s):
        return self.blocks[s]

    def kernel_matrix(self):
        rs = self.block_rowstart
        return sp.block_diag([sp.csr_matrix(self.R[:, rs[s]:rs[s + 1]].T) for s in range(self.nsub)], format="csr")

    coarse = CubeFeti.coarse

    def subset(self, blocks):
        """One rank's share, in the layout FetiDualQP consumes (CubeFeti.subset)."""
        blocks = list(blocks)
        rs = self.block_rowstart
        keep = np.zeros(self.N, dtype=bool)
        newidx = -np.ones(self.N, dtype=np.int64)
        o = 0
        nrs = [0]
        for s in blocks:
            n = int(rs[s + 1] - rs[s])
            keep[rs[s]:rs[s + 1]] = True
            newidx[rs[s]:rs[s + 1]] = np.arange(o, o + n)
            o += n
            nrs.append(o)
        sel = keep[self.leaves_row]
        return dict(nblocks=len(blocks), block_rowstart=np.asarray(nrs, dtype=np.int32), K=csr_block_diag([self.blocks[s] for s in blocks]), f=self.f[keep], R=self.R[:, keep],
                    leaves_row=newidx[self.leaves_row[sel]].astype(np.int32), leaves_root=self.leaves_root[sel], leaves_sign=self.leaves_sign[sel], n_x=o, n_lambda=self.n_lambda)


# ---- algebraic multigrid hierarchy (smoothed aggregation) for blocks of ANY shape: the scipy restatement of pmh_mg_create_sa ----------------------
def _splitmix_start(n):
    """The fixed start vector of the library's power method (splitmix64 -> [-1, 1)), mgbox.hip lambda_max_dinv_a."""
    with np.errstate(over="ignore"):
        s = np.uint64(0x9E3779B97F4A7C15) * (np.arange(n, dtype=np.uint64) + np.uint64(2))
        z = s.copy()
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) / 4503599627370496.0 - 1.0


def _lambda_max_dinv_a_fixed(A, its=20):
    d = A.diagonal()
    dinv = np.where(d != 0.0, 1.0 / np.where(d != 0.0, d, 1.0), 1.0)
    v = _splitmix_start(A.shape[0])
    lam = 1.0
    for _ in range(its):
        w = dinv * (A @ v)
        nw, nv = np.linalg.norm(w), np.linalg.norm(v)
        lam = nw / max(nv, 1e-300)
        v = w / max(nw, 1e-300)
    return float(lam)


def sa_aggregate(S):
    """Greedy aggregation of Vanek / Mandel / Brezina on the strength graph S (CSR over nodes, no diagonal, weights = coupling strength): phase 1 -- in index
    order, a node all of whose neighbours are still free becomes a root and takes them; phase 2 -- a free node joins the phase-1 aggregate of its strongest
    aggregated neighbour (lowest index on ties); phase 3 -- what is left forms aggregates with its free neighbours.  Returns (aggregate of every node, count)."""
    nn = S.shape[0]
    ip, ix, w = S.indptr, S.indices, S.data
    agg = -np.ones(nn, dtype=np.int64)
    na = 0
    for i in range(nn):
        if agg[i] >= 0:
            continue
        nb = ix[ip[i]:ip[i + 1]]
        if nb.size and np.all(agg[nb] < 0):
            agg[i] = na
            agg[nb] = na
            na += 1
    agg1 = agg.copy()
    for i in range(nn):
        if agg[i] >= 0:
            continue
        nb, ww = ix[ip[i]:ip[i + 1]], w[ip[i]:ip[i + 1]]
        best, bw = -1, -1.0
        for j, x in zip(nb, ww):
            if agg1[j] >= 0 and x > bw * (1.0 + 1e-10):
                best, bw = j, x
        if best >= 0:
            agg[i] = agg1[best]
    for i in range(nn):
        if agg[i] >= 0:
            continue
        agg[i] = na
        for j in ix[ip[i]:ip[i + 1]]:
            if agg[j] < 0:
                agg[j] = na
        na += 1
    return agg, na


def _sa_level(A, Bn, bs, theta, omega):
    """One coarsening step: (P, A_c, B_c, lambda_max(D^-1 A))."""
    n = A.shape[0]
    nn, m = n // bs, Bn.shape[1]
    A = A.tocsr()
    # block Frobenius norms on the node graph
    Ac = A.tocoo()
    S2 = sp.coo_matrix((Ac.data ** 2, (Ac.row // bs, Ac.col // bs)), shape=(nn, nn)).tocsr()
    S2.sum_duplicates()
    S2.sort_indices()
    dg = np.sqrt(S2.diagonal())
    Sc = S2.tocoo()
    sv = np.sqrt(Sc.data)
    keep = (Sc.row != Sc.col) & (sv > theta * np.sqrt(dg[Sc.row] * dg[Sc.col])) & (sv > 0.0)
    S = sp.csr_matrix((sv[keep], (Sc.row[keep], Sc.col[keep])), shape=(nn, nn))
    S.sort_indices()
    agg, na = sa_aggregate(S)
    # tentative prolongation: per aggregate, modified Gram-Schmidt (twice) of the near-kernel restricted to it; a column that is (numerically) dependent there is dropped
    order = np.argsort(agg, kind="stable")
    start = np.concatenate([[0], np.cumsum(np.bincount(agg, minlength=na))])
    rows, cols, vals = [], [], []
    Bc = np.zeros((na * m, m))
    for a in range(na):
        nodes = order[start[a]:start[a + 1]]
        dofs = (nodes[:, None] * bs + np.arange(bs)[None, :]).ravel()
        V = Bn[dofs, :].copy()
        for k in range(m):
            n0 = np.linalg.norm(V[:, k])
            for _ in range(2):
                for j in range(k):
                    V[:, k] -= (V[:, j] @ V[:, k]) * V[:, j]
            nk = np.linalg.norm(V[:, k])
            if n0 == 0.0 or nk <= 1e-8 * n0:
                V[:, k] = 0.0  # dependent on the earlier columns over this aggregate (fewer than 3 non-collinear nodes): a dead coarse dof
            else:
                V[:, k] /= nk
        Bc[a * m:(a + 1) * m, :] = V.T @ Bn[dofs, :]  # B_a = Q (Q' B_a): the coarse near-kernel
        rr, cc = np.meshgrid(dofs, a * m + np.arange(m), indexing="ij")
        nz = V != 0.0
        rows.append(rr[nz]), cols.append(cc[nz]), vals.append(V[nz])
    Pt = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, na * m))
    lam = _lambda_max_dinv_a_fixed(A)
    d = A.diagonal()
    dinv = np.where(d != 0.0, 1.0 / np.where(d != 0.0, d, 1.0), 1.0)
    P = (Pt - sp.diags(omega / lam * dinv) @ (A @ Pt)).tocsr()
    P.sort_indices()
    Acn = (P.T @ A @ P).tocsr()
    Acn = (0.5 * (Acn + Acn.T)).tolil()
    dead = np.nonzero(np.asarray(abs(Pt).sum(axis=0)).ravel() == 0.0)[0]
    for k in dead:
        Acn[k, k] = 1.0
    Acn = Acn.tocsr()
    Acn.sort_indices()
    return P, Acn, Bc, lam


def sa_mg_hierarchy(blocks, nns, ndof=3, max_coarse=1500, max_levels=8, theta=0.0, omega=4.0 / 3.0, singular=None):
    """Smoothed-aggregation hierarchy (Vanek, Mandel, Brezina 1996) for a block-diagonal matrix whose blocks have NO structure to lean on: per level the node graph
    (ndof dofs per node on the fine level, one node per aggregate with m = near-kernel dimension dofs below), greedy aggregates, the tentative prolongation that
    reproduces the near-kernel vectors nns[b] (m x n_b: the rigid-body modes of an elasticity block -- for a floating TFETI block its kernel R_b) exactly,
    one damped-Jacobi smoothing step P = (I - omega / lambda_max D^-1 A) P_t, Galerkin operators.  Same dict as box_mg_hierarchy (input of MatInv.set_pc_mg /
    oracle.mg_host).  singular[b] (default: A_b nns_b' = 0 numerically): the coarsest operator of block b keeps nns as its kernel and gets a pseudo-inverse."""
    per_block = []
    for b, (Kb, Nb) in enumerate(zip(blocks, nns)):
        A, P, lam = [Kb.tocsr()], [], []
        Bn = np.asarray(Nb, dtype=np.float64).T.copy()
        bs = ndof
        sing = (np.abs(Kb @ Bn).max() <= 1e-9 * abs(Kb).max() * max(np.abs(Bn).max(), 1e-300)) if singular is None else bool(singular[b])
        Bl = [Bn]
        per_block.append([A, P, lam, Bl, sing, bs])
    # all blocks get the SAME number of levels: that of the block that needs most
    done = False
    while not done:
        done = True
        if len(per_block[0][0]) >= max_levels:
            break
        if max([pb[0][-1].shape[0] for pb in per_block if len(pb) < 7] + [0]) <= max_coarse:  # (a block whose level cannot be aggregated any further does not ask for more levels)
            break
        for pb in per_block:
            A, P, lam, Bl, sing, bs = pb[:6]
            lvl = None if A[-1].shape[0] // bs <= 8 else _sa_level(A[-1], Bl[-1], bs, theta * (0.5 ** (len(A) - 1)), omega)
            if lvl is not None and lvl[1].shape[0] < 2 * Bl[0].shape[1] and len(pb) < 7:
                pb.append("stalled")
            if lvl is None or lvl[1].shape[0] < 2 * Bl[0].shape[1]:
                # down to a handful of nodes while a larger block still coarsens, or a small dense level that would collapse into ONE aggregate: carried over as it is (P = I)
                P.append(sp.identity(A[-1].shape[0], format="csr")), lam.append(_lambda_max_dinv_a_fixed(A[-1])), A.append(A[-1]), Bl.append(Bl[-1])
                continue
            Pn, Ac, Bc, lm = lvl
            P.append(Pn), A.append(Ac), Bl.append(Bc), lam.append(lm)
            pb[5] = Bl[0].shape[1]  # below the fine level a node is an aggregate with m dofs
        done = False
    nlev = len(per_block[0][0])
    out = dict(A=[], P=[], lambda_max=[])
    for l in range(nlev):
        Al = sp.block_diag([pb[0][l] for pb in per_block], format="csr")
        Al.sort_indices()
        out["A"].append(Al)
        if l + 1 < nlev:
            Pl = sp.block_diag([pb[1][l] for pb in per_block], format="csr")
            Pl.sort_indices()
            out["P"].append(Pl)
            out["lambda_max"].append(max(pb[2][l] for pb in per_block))
    sizes = [pb[0][-1].shape[0] for pb in per_block]
    out["coarse_rowstart"] = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    out["coarse_pinv"] = np.concatenate([np.linalg.pinv(pb[0][-1].toarray(), rcond=1e-10, hermitian=True).ravel() for pb in per_block])
    return out
