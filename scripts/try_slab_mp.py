"""Accuracy of the Moore-Penrose K^+ (block CG on the singular K, projected) on ex71's one-element-thick elasticity slabs, against dense pinv."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import permon_amd as pa
from permon_amd.feti import DmdaFeti

ctx = pa.Context(0)
prob = DmdaFeti((8, 6, 4), 7, "elasticity")
rs = np.asarray(prob.block_rowstart)
Kd = prob.K.toarray()
print("N", prob.N, "blocks", rs)
Kp = np.zeros_like(Kd)
for s in range(len(rs) - 1):
    lo, hi = rs[s], rs[s + 1]
    blk = Kd[lo:hi, lo:hi]
    w = np.linalg.eigvalsh(blk)
    print("block", s, "n", hi - lo, "smallest eig", w[:8], "largest", w[-1])
    Kp[lo:hi, lo:hi] = np.linalg.pinv(blk, rcond=1e-10, hermitian=True)
R = np.asarray(prob.R)
print("R shape", R.shape, "||K R'||", np.linalg.norm(Kd @ R.T), "R R' - I", np.abs(R @ R.T - np.eye(R.shape[0])).max() if R.shape[0] else 0)
K = pa.MatBlockDiag.from_scipy(ctx, prob.block_rowstart, prob.K)
rng = np.random.default_rng(3)
g = rng.standard_normal(prob.N)
for rtol in (1e-10, 1e-12, 1e-14):
    for jac in (True, False):
        M = pa.MatInv(K, rtol=rtol, max_it=40000, jacobi=jac, nullspace=R)
        u = ctx.vec(prob.N)
        M.mult(ctx.vec_from(g), u)
        ref = Kp @ g
        its, tot = M.last_iterations()
        print("rtol %.0e jacobi %s: rel err vs pinv %.3e, its %d" % (rtol, jac, np.linalg.norm(u.to_numpy() - ref) / np.linalg.norm(ref), its))
f = np.asarray(prob.f)
M = pa.MatInv(K, rtol=1e-14, max_it=40000, nullspace=R)
u = ctx.vec(prob.N); M.mult(ctx.vec_from(f), u)
print("K+ f: rel err", np.linalg.norm(u.to_numpy() - Kp @ f) / np.linalg.norm(Kp @ f))
un = u.to_numpy(); ref = Kp @ f
for s in range(len(rs) - 1):
    lo, hi = rs[s], rs[s + 1]
    Rb = R[:, lo:hi]
    fb = f[lo:hi]
    Pf = fb - Rb.T @ (Rb @ fb)
    e = un[lo:hi] - ref[lo:hi]
    print("block %d: ||f|| %.3e ||P_R f|| %.3e  err %.3e (rel to ref %.3e)  kernel part of err %.3e  ||K err|| %.3e" % (s, np.linalg.norm(fb), np.linalg.norm(Pf), np.linalg.norm(e), np.linalg.norm(e) / max(np.linalg.norm(ref[lo:hi]), 1e-300), np.linalg.norm(Rb @ e), np.linalg.norm(Kd[lo:hi, lo:hi] @ e)))
