"""ex71 TEST 2 (7 elasticity slabs) with the K^+ the reference actually takes there (no kernel supplied by KSPFETI => computed => left generalised inverse, qptransform.c:997-1008):
iteration counts and the KKT report's norms (||d|| = ratio of the dual QP's lines identifies the K^+: golden 17.36)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import permon_amd as pa
from permon_amd.feti import DmdaFeti

ctx = pa.Context(0)
prob = DmdaFeti((8, 6, 4), 7, "elasticity")
nd = prob.ndof
l2g = np.concatenate([(np.asarray(g)[:, None] * nd + np.arange(nd)[None, :]).ravel() for g in prob.gids]).astype(np.int32)
for pc in ("none", "lumped"):
    for extra in ("", "-qpt_dualize_Kplus_left 0", "-qpt_dualize_Kplus_mp"):
        u, lam, st = pa.KSPFETISolve(ctx, prob.block_rowstart, prob.K, prob.f, l2g, R=prob.R, kplus_rtol=1e-14, kplus_max_it=40000,
                                     options="-qps_view_convergence -qp_chain_view_kkt -pde_type Elasticity -dim 3 -qps_rtol 1e-6 -dual_pc_dual_type %s %s" % (pc, extra))
        print("pc %-6s K+ %-24s: %d iterations reason %d rnorm %.4e" % (pc, extra or "left (default)", st.iteration, st.reason, st.rnorm), flush=True)
        print(st.view_text)
