"""CPU restatement of the product's iterative K^+ (TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this; nothing under permon_amd/ does).

What is restated: the inner KSP of MATINV (src/mat/impls/inv/matinv.c:734-743: KSPSolve per application) as the product runs
it -- block-wise preconditioned CG, every block with its own scalars and stopping test (KSPConvergedDefault, zero initial
guess), preconditioned by a PCMG-style V-cycle: Chebyshev(degree)/Jacobi smoothing on [lo, hi] x lambda_max(D^-1 A) before and
after the coarse correction, Galerkin operators, dense block pseudo-inverses on the coarsest level (permon_amd/csrc/mg.hip), and
the Moore-Penrose wrapping P_R K^- P_R of QPTDualize (src/qp/interface/qptransform.c:1020-1062).  numpy for the vectors; the
sparse products go through `spmv` (scipy by default, the OpenMP CSR kernel of oracle/permon_oracle.c in the CPU baseline).
"""
import numpy as np


def vcycle(H, degree=2, lo=0.1, hi=1.1, spmv=None):
    """x = V(b) for the hierarchy dict of permon_amd.feti.box_mg_hierarchy (A, P, lambda_max, coarse_rowstart, coarse_pinv).
    spmv(tag, level, x): 'A' -> A_l x, 'P' -> P_l x, 'R' -> P_l' x; None = scipy."""
    A, P, lam = H["A"], H["P"], H["lambda_max"]
    nl = len(A)
    dinv = [1.0 / a.diagonal() for a in A]
    Pt = [p.T.tocsr() for p in P]
    crs = H["coarse_rowstart"]
    blocks, o = [], 0
    for b in range(len(crs) - 1):
        m = crs[b + 1] - crs[b]
        blocks.append(np.asarray(H["coarse_pinv"][o:o + m * m]).reshape(m, m))
        o += m * m
    if spmv is None:
        def spmv(tag, l, x):
            return (A[l] if tag == "A" else P[l] if tag == "P" else Pt[l]) @ x

    def smooth(l, b, x):
        a_, b_ = lo * lam[l], hi * lam[l]
        th, de = (a_ + b_) / 2, (b_ - a_) / 2
        sig = th / de
        rho = 1 / sig
        if x is None:
            r = dinv[l] * b
            d = r / th
            x = d.copy()
        else:
            r = dinv[l] * (b - spmv("A", l, x))
            d = r / th
            x = x + d
        for _ in range(1, degree):
            rn = 1 / (2 * sig - rho)
            r = r - dinv[l] * spmv("A", l, d)
            d = rn * rho * d + 2 * rn / de * r
            x = x + d
            rho = rn
        return x

    def V(l, b):
        if l == nl - 1:
            return np.concatenate([blocks[k] @ b[crs[k]:crs[k + 1]] for k in range(len(blocks))])
        x = smooth(l, b, None)
        x = x + spmv("P", l, V(l + 1, spmv("R", l, b - spmv("A", l, x))))
        return smooth(l, b, x)

    return lambda b: V(0, b)


class KplusMG:
    """u = K^+ f: block-wise V-cycle-preconditioned CG (rtol on every block's own residual), Moore-Penrose wrapped with the
    block-wise orthonormal kernel basis R (kdim x n) when given."""

    def __init__(self, K, block_rowstart, H, R=None, rtol=1e-9, max_it=200, degree=2, spmv=None):
        self.K, self.rs, self.R, self.rtol, self.max_it = K.tocsr(), np.asarray(block_rowstart), R, rtol, max_it
        self.V = vcycle(H, degree, spmv=spmv)
        self.spmv = spmv
        self.n_spmv = 0
        self.last_its = 0

    def _proj(self, v):
        if self.R is None:
            return v
        out = v.copy()
        for b in range(len(self.rs) - 1):
            lo, hi = self.rs[b], self.rs[b + 1]
            Rb = self.R[:, lo:hi]
            out[lo:hi] -= Rb.T @ (Rb @ v[lo:hi])
        return out

    def _bdot(self, x, y):
        return np.add.reduceat(x * y, self.rs[:-1])

    def _Kx(self, x):
        self.n_spmv += 1
        return self.spmv("A", 0, x) if self.spmv is not None else self.K @ x

    def __call__(self, f):
        f0 = np.asarray(f, dtype=np.float64)
        f = self._proj(f0)
        nb = len(self.rs) - 1
        sizes = np.diff(self.rs)
        u = np.zeros_like(f)
        r = f.copy()
        z = self.V(r)
        p = z.copy()
        rz = self._bdot(r, z)
        tol = self.rtol * np.sqrt(self._bdot(r, r))
        active = np.sqrt(self._bdot(r, r)) > tol
        if self.R is not None:  # the product's rule (k_cg_init): a load in the kernel leaves only the rounding residue of its projection, which is not in the range of the singular K: u_b = 0
            active &= np.sqrt(self._bdot(r, r)) > 64.0 * np.finfo(float).eps * np.sqrt(self._bdot(f0, f0))
        it = 0
        while it < self.max_it and active.any():
            Ap = self._Kx(p)
            pAp = self._bdot(p, Ap)
            alpha = np.where(active, rz / np.where(pAp != 0, pAp, 1.0), 0.0)
            ae = np.repeat(alpha, sizes)
            u += ae * p
            r -= ae * Ap
            z = self.V(r)
            rzn = self._bdot(r, z)
            rr = self._bdot(r, r)
            beta = np.where(active, rzn / np.where(rz != 0, rz, 1.0), 0.0)
            it += 1
            active = active & (np.sqrt(rr) > tol)
            p = z + np.repeat(np.where(active, beta, 0.0), sizes) * p
            rz = rzn
        self.last_its = it
        return self._proj(u)
