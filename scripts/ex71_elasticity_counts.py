import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, permon_amd as pa
from permon_amd.feti import DmdaFeti
ctx = pa.Context(0)
prob = DmdaFeti((8, 6, 4), 7, "elasticity")
nd = prob.ndof
l2g = np.concatenate([(np.asarray(g)[:, None] * nd + np.arange(nd)[None, :]).ravel() for g in prob.gids]).astype(np.int32)
import scipy.sparse.linalg as spla
Kb = prob.blocks[1]
lmax = float(spla.eigsh(Kb, k=1, which="LA", return_eigenvectors=False)[0])
print("lambda_max(K_1) =", lmax)
for rho in (0.0, 0.25 * lmax, 0.5 * lmax, 0.75 * lmax, lmax, 1.5 * lmax, 1.0, 2.0, 4.0):
    for lumped in (False, True):
        u, lam, st = pa.KSPFETISolve(ctx, prob.block_rowstart, prob.K, prob.f, l2g, R=prob.R, regularize=True, lumped=lumped, rtol=1e-6, kplus_rtol=1e-14, regularize_rho=rho)
        print("rho %.4f lumped %s its %d rnorm %.3e" % (rho, lumped, st.iteration, st.rnorm), flush=True)
ctx.close()
