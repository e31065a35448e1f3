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

__all__ = ["q1_elasticity_element", "q1_poisson_element", "CubeFeti"]


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


class CubeFeti:
    """TFETI data for sx x sy x sz unit cubes of nel^3 Q1 elements each.

    physics 'elasticity' (3 dof/node, 6 rigid-body modes per cube) or 'poisson' (1 dof/node, 1 mode).
    Dirichlet u = 0 on the global x = 0 face enforced by B (TFETI: every subdomain floats,
    KSPFETISetDirichlet(..., FETI_LOCAL, PETSC_TRUE) as in src/tutorials/feti/ex1.c:89-90);
    gluing 'nonred' or 'full' with -SCALE_ON values +-1/sqrt(multiplicity) (qpfeti.c:789-806);
    contact=True adds the rigid obstacle below the global z = 0 face: -u_z <= gap (inequality rows).
    Dual rows are ordered [Dirichlet | gluing | contact]; the first n_eq are equalities.
    """

    def __init__(self, sub=(2, 2, 2), nel=3, physics="elasticity", gluing="full", scale=True, contact=True, gap0=0.0, gap_slope=0.05, load=-1.0):
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
            sc = 1.0 / np.sqrt(m) if scale else 1.0
            pairs = [(cp[i], cp[i + 1]) for i in range(m - 1)] if gluing == "nonred" else [(cp[i], cp[j]) for i in range(m) for j in range(i + 1, m)]
            for (sa, la), (sb, lb_) in pairs:
                for c in range(nd):
                    rows_l += [sa * nloc + la * nd + c, sb * nloc + lb_ * nd + c]
                    roots_l += [nrow, nrow]
                    vals_l += [sc, -sc]  # the copy on the higher rank carries -1 (qpfeti.c:791-795)
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
            self._K = sp.block_diag([self.Ki] * self.nsub, format="csr")
            self._K.sort_indices()
        return self._K

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
            K=sp.block_diag([self.Ki] * len(blocks), format="csr"), f=self.f[keep], R=self.R[:, keep],
            leaves_row=newidx[self.leaves_row[sel]].astype(np.int32), leaves_root=self.leaves_root[sel], leaves_sign=self.leaves_sign[sel],
            n_x=len(blocks) * nloc, n_lambda=self.n_lambda)
