"""feti/ex1.c TEST smalxe_orth: -ne 7 (4 subdomains) -project 0 -qps_smalxe_rho 1e1 -dual_qp_E_orth_type {gs, implicit}: SMALXE on the UNprojected dual QP
(A = F, BE = T G orthonormal rows, no box => the inner solve is unconstrained); golden: 16 outer iterations."""
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import permon_amd as pa  # noqa: E402
from permon_amd.chain import FetiDualQP  # noqa: E402

ctx = pa.Context(0)
ns, ne_l = 4, 7
nl, ng = ne_l + 1, ns * ne_l + 1
h = 1.0 / (ns * ne_l)
N = ns * nl
K = np.zeros((N, N))
b = np.zeros(ng)
l2g = np.zeros(N, dtype=np.int64)
for r in range(ns):
    for i in range(nl):
        l2g[r * nl + i] = r * ne_l + i
    for i in range(ne_l):
        a = r * nl + i
        K[a, a] += 1
        K[a + 1, a + 1] += 1
        K[a, a + 1] -= 1
        K[a + 1, a] -= 1
        v = np.sin((r * ne_l + i + .5) * h * 3.14159) * .5 * h * h
        b[r * ne_l + i] += v
        b[r * ne_l + i + 1] += v
mult = np.bincount(l2g, minlength=ng)
f = b[l2g] / mult[l2g]
rows, roots, vals = [0, N - 1], [0, 1], [1.0, 1.0]
for r in range(ns - 1):
    rows += [r * nl + nl - 1, (r + 1) * nl]
    roots += [2 + r, 2 + r]
    vals += [1 / np.sqrt(2), -1 / np.sqrt(2)]
nlam = 2 + ns - 1
R = np.zeros((1, N))
for r in range(ns):
    R[0, r * nl:(r + 1) * nl] = 1 / np.sqrt(nl)
local = dict(nblocks=ns, block_rowstart=np.arange(ns + 1, dtype=np.int32) * nl, K=sp.csr_matrix(K), f=f, R=R, leaves_row=np.array(rows, dtype=np.int32), leaves_root=np.array(roots, dtype=np.int32),
             leaves_sign=np.array(vals), n_x=N, n_lambda=nlam)
B = sp.csr_matrix((vals, (roots, rows)), shape=(nlam, N))
Rm = np.zeros((N, ns))
for r in range(ns):
    Rm[r * nl:(r + 1) * nl, r] = 1 / np.sqrt(nl)
G = sp.csr_matrix((B @ Rm).T)
e = Rm.T @ f
Qm, Rq = np.linalg.qr(G.toarray().T)  # G0' = Q R  =>  the orthonormal-row matrix is Q' = R^{-T} G0, and the constraint right-hand side R^{-T} e
for form in ("gs", "implicit"):
    q = FetiDualQP(ctx, local, G if form == "implicit" else sp.csr_matrix(Qm.T), e if form == "implicit" else np.linalg.solve(Rq.T, e), np.zeros(nlam), np.full(nlam, -np.inf),
                   orthonormal="implicit" if form == "implicit" else True, kplus_rtol=1e-14, regularize=True, explicit=dict(rtol=1e-14))
    print(form, "chain built; has_box", q.has_box, "A is F:", q.A is q.F)
    qp = pa.QP(ctx)
    qp.SetOperator(q.F)
    qp.SetRhs(q.b_bar)
    lam = ctx.vec(nlam)
    qp.SetInitialVector(lam)
    qp.lb, qp.ub = None, None
    qp.SetEq(q.pf)
    qps = pa.QPS(ctx)
    qps.SetQP(qp)
    left = qps.SetFromOptions("-qps_smalxe_rho 1e1")
    print("type", qps.type, "left", left, "rho_user", qps.smalxe_opts.rho_user if hasattr(qps.smalxe_opts, "rho_user") else "?")
    st = qps.Solve()
    print(form, "outer", st.iteration, "inner", st.inner_iter_accu, "reason", st.reason, "rnorm", st.rnorm, "M1_updates", st.M1_updates, "rho_updates", st.rho_updates, "lam", (lam.to_numpy() + q.lam_tilde.to_numpy()))
ctx.close()
