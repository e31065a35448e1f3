"""configs[2] with the explicit local dual operators: assembly seconds, GEMV GB/s, F apply and MPGP step times next to the
multigrid-CG K^+, and the full SMALXE solve through both.  usage: python scripts/explicit_c2.py [nel] [fx_rw]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import permon_amd as pa  # noqa: E402
from permon_amd.chain import FetiDualQP  # noqa: E402

nel = int(sys.argv[1]) if len(sys.argv) > 1 else 43
ctx = pa.Context(0)
t0 = time.time()
f = pa.CubeFeti((2, 2, 2), nel, contact=True)
G, e = f.coarse()
loc = f.subset(range(8))
nn = nel + 1
hier = pa.box_mg_hierarchy([f.Ki] * 8, [(nn, nn, nn)] * 8, 3, min_nodes=min(400, nn ** 3 // 8))
print("generate + hierarchy %.1f s" % (time.time() - t0), flush=True)
t0 = time.time()
q = FetiDualQP(ctx, loc, G, e, f.c, f.lb, kplus_rtol=1e-9, mg_hierarchy=hier, mg_precision="fp16", bsr3=True, explicit=dict(rtol=float(os.environ.get("FX_RTOL", "1e-12")), storage=os.environ.get("FX_STORAGE", "sym")))
ctx.sync()
ns, secs = q.E.assemble_stats()
out = {"nel": nel, "n_gamma": q.E.n_gamma.tolist(), "dense_GB": q.E.dense_bytes / 1e9, "assemble_solves": ns, "assemble_seconds": secs, "setup_seconds": time.time() - t0}
print(json.dumps(out), flush=True)
lam = ctx.vec_from(np.random.default_rng(1).standard_normal(f.n_lambda))
y1, y2 = ctx.vec(f.n_lambda), ctx.vec(f.n_lambda)


def timeit(fn, reps):
    fn()
    ctx.sync()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    return (time.perf_counter() - t) / reps * 1e3


q.E.timing_enable(200)
ms_ex = timeit(lambda: q.F.mult(lam, y1), 50)
n, ms, b = q.E.timing_get()
q.E.timing_enable(0)
q.Kplus.attach_explicit(None)
ms_it = timeit(lambda: q.F.mult(lam, y2), 10)
d = np.linalg.norm(y1.to_numpy() - y2.to_numpy()) / np.linalg.norm(y2.to_numpy())
print(json.dumps({"F_apply_ms_explicit": ms_ex, "F_apply_ms_iterative_rtol1e-9": ms_it, "rel_diff": d, "gemv_launches": n, "gemv_avg_ms": ms / max(n, 1), "gemv_GBs": b / (ms / max(n, 1) * 1e-3) / 1e9,
                  "gemv_frac_of_8TBs": b / (ms / max(n, 1) * 1e-3) / 8e12, "first_kernel_avg_ms": q.E.first_kernel_ms / max(n, 1)}), flush=True)
for mode in ("explicit", "iterative"):
    q.Kplus.attach_explicit(q.E if mode == "explicit" else None)
    q.lam.set(0.0)
    qps = q.make_smalxe()
    qps.RunFixed(5)
    q.lam.set(0.0)
    ctx.sync()
    t = time.perf_counter()
    st = qps.RunFixed(50)
    ctx.sync()
    dt = time.perf_counter() - t
    print(json.dumps({"mode": mode, "ms_per_step": dt / 50 * 1e3, "nmv": st.nmv, "ncg": st.ncg, "nexp": st.nexp}), flush=True)
    q.lam.set(0.0)
    ctx.sync()
    t = time.perf_counter()
    s = qps.Solve()
    ctx.sync()
    dt = time.perf_counter() - t
    print(json.dumps({"mode": mode, "solve_seconds": dt, "outer": s.iteration, "inner": s.inner_iter_accu, "reason": s.reason, "nmv": s.inner.nmv, "ncg": s.inner.ncg, "nexp": s.inner.nexp, "nprop": s.inner.nprop,
                      "rnorm": s.rnorm}), flush=True)
    qps.Destroy()
ctx.close()
