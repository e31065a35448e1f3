"""the residual history of feti/ex71.c TEST 2 (7 elasticity slabs, -qps_rtol 1e-6) with -dual_pc_dual_type lumped and none on the GPU path, for both
K^+ (K_reg^{-1} at two regularisation scales, Moore-Penrose) and through the explicit local dual operators: does the iteration the golden stops at (26 / 66) miss the
threshold rtol ||b|| by rounding or by an operator difference?  PMH_KSP_MONITOR=1 prints -ksp_monitor's lines on stderr.
  gpurun -- python scripts/ex71_lumped_history.py 2> gpurun_out/ex71_hist.txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PMH_KSP_MONITOR"] = "1"
import permon_amd as pa  # noqa: E402
from permon_amd.feti import DmdaFeti  # noqa: E402

ctx = pa.Context(0)
prob = DmdaFeti((8, 6, 4), 7, "elasticity")
nd = prob.ndof
l2g = np.concatenate([(np.asarray(g)[:, None] * nd + np.arange(nd)[None, :]).ravel() for g in prob.gids]).astype(np.int32)
for lumped in (True, False):
    for name, kw in (("K_reg^{-1} rho = lambda_max (default)", dict(regularize=True)), ("K_reg^{-1} rho = 1", dict(regularize=True, regularize_rho=1.0)),
                     ("Moore-Penrose P_R K^- P_R", dict(regularize=False)), ("K_reg^{-1} explicit operators", dict(regularize=True, explicit=True))):
        sys.stderr.write("==== -dual_pc_dual_type %s, K^+ = %s\n" % ("lumped" if lumped else "none", name))
        sys.stderr.flush()
        u, lam, st = pa.KSPFETISolve(ctx, prob.block_rowstart, prob.K, prob.f, l2g, R=prob.R, kplus_rtol=1e-14,
                                     options="-qps_rtol 1e-6 -dual_pc_dual_type %s" % ("lumped" if lumped else "none"), **kw)
        sys.stderr.write("==== -> %d iterations (golden %d), reason %d\n" % (st.iteration, 26 if lumped else 66, st.reason))
ctx.close()
