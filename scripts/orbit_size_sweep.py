"""Contact TFETI on 2 x 2 x 2 cubes of nel^3 elements for a sweep of odd sizes (row / k / column remainders of the orbit GEMM's tiles): the explicit operator in orbit
storage against the inner-Krylov K^+ -- same SMALXE / MPGP counts, lambda to 1e-7, F lambda to 1e-9.  usage: python scripts/orbit_size_sweep.py [nel ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import permon_amd as pa  # noqa: E402
from permon_amd.chain import FetiDualQP  # noqa: E402

ctx = pa.Context(0)
bad = 0
for nel in [int(a) for a in sys.argv[1:]] or [3, 6, 7, 9, 11, 13, 17, 23]:
    f = pa.CubeFeti((2, 2, 2), nel, contact=True)
    G, e = f.coarse()
    loc = f.subset(range(f.nsub))
    nn = nel + 1
    qi = FetiDualQP(ctx, loc, G, e, f.c, f.lb, kplus_rtol=1e-12)
    qo = FetiDualQP(ctx, loc, G, e, f.c, f.lb, kplus_rtol=1e-12, explicit=dict(rtol=1e-13, storage="class_orbit", symmetry=dict(dims=(nn, nn, nn), ndof=3)))
    lam = np.random.default_rng(nel).standard_normal(f.n_lambda)
    lv, y0, y1 = ctx.vec_from(lam), ctx.vec(f.n_lambda), ctx.vec(f.n_lambda)
    qi.F.mult(lv, y0)
    qo.F.mult(lv, y1)
    dF = np.linalg.norm(y1.to_numpy() - y0.to_numpy()) / np.linalg.norm(y0.to_numpy())
    si, so = qi.solve_smalxe(rtol=1e-6), qo.solve_smalxe(rtol=1e-6)
    li, lo = qi.dual_solution(), qo.dual_solution()
    dl = np.linalg.norm(li - lo) / np.linalg.norm(li)
    ci, co = (si.iteration, si.inner_iter_accu, si.inner.ncg, si.inner.nexp), (so.iteration, so.inner_iter_accu, so.inner.ncg, so.inner.nexp)
    ok = dF <= 1e-9 and dl <= 1e-7 and ci == co and qo.explicit_storage == "class_orbit"
    bad += not ok
    print("nel %2d n_lambda %6d n_c %6d symmetries %2d: |F_orbit l - F_iter l| %.1e, counts %s vs %s, |dlambda| %.1e %s" % (nel, f.n_lambda, qo.E.class_union(0).size, qo.explicit_symmetries, dF, co, ci, dl, "OK" if ok else "MISMATCH"), flush=True)
ctx.close()
sys.exit(1 if bad else 0)
