"""Per-operator times of one F / projector apply on the configs[2] TFETI problem (HIP-event timed, averaged):
finds the small-launch costs that dominate once the subdomain blocks are spread over 8 GPUs.
  python scripts/feti_op_times.py [sim_world=8] [nel=43]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import permon_amd as pa
from permon_amd.chain import FetiDualQP

sim = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nel = int(sys.argv[2]) if len(sys.argv) > 2 else 43
ctx = pa.Context(0)
f = pa.CubeFeti((2, 2, 2), nel, contact=True)
G, e = f.coarse(orthonormalize=True)
per = 8 // sim
local = f.subset(range(per))
nn = nel + 1
hier = pa.box_mg_hierarchy([f.Ki] * per, [(nn, nn, nn)] * per, 3)
q = FetiDualQP(ctx, local, G, e, f.c, f.lb, orthonormal=True, kplus_rtol=1e-9, mg_hierarchy=hier, mg_degree=2, mg_precision="fp16", bsr3=True)
nl, nx = q.n_lambda, local["n_x"]
lam, lam2, x, x2 = ctx.vec_from(np.random.default_rng(1).standard_normal(nl)), ctx.vec(nl), ctx.vec(nx), ctx.vec(nx)
cm = ctx.vec(q.pf.m)


def t(name, fn, reps=20):
    for _ in range(3):
        fn()
    ctx.sync()
    ctx.timer_start()
    for _ in range(reps):
        fn()
    ms = ctx.timer_stop() / reps
    print("%-34s %9.1f us" % (name, ms * 1e3), flush=True)


print("sim-world %d: %d block(s), n_x=%d, n_lambda=%d, m_G=%d" % (sim, per, nx, nl, q.pf.m))
t("B' lambda (gluing mult)", lambda: q.B.mult(lam, x))
t("B x (gluing mult transpose)", lambda: q.B.mult_transpose(x, lam2))
t("G v", lambda: q.pf.ApplyG(lam, cm))
t("Q v", lambda: q.pf.ApplyQ(lam, lam2))
t("P v", lambda: q.pf.ApplyP(lam, lam2))
t("GtG v", lambda: q.pf.ApplyGtG(lam, lam2))
q.B.mult(lam, x)
t("K^+ f", lambda: q.Kplus.mult(x, x2), reps=5)
print("K^+ block CG iterations:", q.Kplus.last_iterations()[0])
t("V-cycle", lambda: q.Kplus.mg.apply(x, x2))
t("K x (blockdiag mult, CSR)", lambda: q.K.mult(x, x2))
t("F lambda", lambda: q.F.mult(lam, lam2), reps=5)
t("P F P lambda", lambda: q.A.mult(lam, lam2), reps=5)
ctx.close()
