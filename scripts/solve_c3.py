"""Full SMALXE+MPGP solve of BASELINE configs[2] (or a smaller nel) on one GPU; prints convergence facts and
size-independent checks (dual feasibility, G lambda = e, primal constraint violation after recovery)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import permon_amd as pa
from permon_amd.chain import FetiDualQP

nel = int(sys.argv[1]) if len(sys.argv) > 1 else 43
rtol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-5
pc = sys.argv[3] if len(sys.argv) > 3 else "mg"  # PC of the inner KSP of MATINV: mg (fp16 V-cycle, as bench.py) or jacobi
ctx = pa.Context(0)
t0 = time.time()
f = pa.CubeFeti((2, 2, 2), nel, contact=True)
G, e = f.coarse(orthonormalize=True)
hier = None
if pc == "mg":
    nn = nel + 1
    hier = pa.box_mg_hierarchy([f.Ki] * 8, [(nn, nn, nn)] * 8, 3)
q = FetiDualQP(ctx, f.subset(range(8)), G, e, f.c, f.lb, orthonormal=True, kplus_rtol=1e-9, mg_hierarchy=hier, mg_precision="fp16", bsr3=True)
t1 = time.time()
st = q.solve_smalxe(rtol=rtol)
ctx.sync()
t2 = time.time()
lam = q.dual_solution()
u, Fl_minus_d = q.primal_solution(G)
Ru = f.kernel_matrix()
tight = (np.arange(f.n_lambda) < f.n_eq) | (lam > 1e-8 * np.abs(lam).max())
BR = (f.B @ Ru).toarray()
alpha = np.linalg.lstsq(BR[tight], Fl_minus_d[tight], rcond=None)[0]
uu = u + Ru @ alpha
Bu = f.B @ uu
its, spmv = q.Kplus.last_iterations()
out = dict(kplus_pc=pc, nel=nel, N=f.N, n_lambda=f.n_lambda, setup_s=round(t1 - t0, 1), solve_s=round(t2 - t1, 1), outer=st.iteration, reason=st.reason,
           inner_total=st.inner_iter_accu, inner_nmv=st.inner.nmv, ncg=st.inner.ncg, nexp=st.inner.nexp, nprop=st.inner.nprop,
           M1_hits=st.M1_hits, eta_hits=st.eta_hits, rho_updates=st.rho_updates, normBu=st.normBu, rnorm=st.rnorm, K_spmv_total=spmv,
           min_lambda_I=float(lam[f.n_eq:].min()), active_contacts=int((lam[f.n_eq:] > 1e-8 * np.abs(lam).max()).sum()),
           G_lambda_minus_e=float(np.linalg.norm(G @ lam - e) / max(1.0, np.linalg.norm(e))),
           eq_violation=float(np.abs(Bu[:f.n_eq] - f.c[:f.n_eq]).max() / np.abs(uu).max()),
           ineq_violation=float(max(0.0, (Bu[f.n_eq:] - f.c[f.n_eq:]).max()) / np.abs(uu).max()),
           its_per_s=round(st.inner_iter_accu / (t2 - t1), 3))
print(json.dumps(out))
