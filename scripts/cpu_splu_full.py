"""The reference's own K^+ at the headline block size, measured ONCE on the GPU box's host (round 5): MATINV factors K_reg = MatRegularize(K, R) per block
(src/mat/impls/inv/matinv.c:481-580) and applies it by one forward / backward substitution (matinv.c:734-743).  PETSc / MUMPS are absent: scipy's SuperLU stands in.
One 43^3 Q1 elasticity block (255 552 dof).  Writes gpurun_out/r05/splu_43.json; progress lines on stderr (the factorisation runs for many minutes)."""
import json
import os
import resource
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
nel = int(sys.argv[1]) if len(sys.argv) > 1 else 43
out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/r05/splu_%d.json" % nel
os.makedirs(os.path.dirname(out), exist_ok=True)
import scipy.sparse.linalg as spla  # noqa: E402

import permon_amd as pa  # noqa: E402

ctx = pa.Context(0)  # MatRegularize runs its power method on the device
g = pa.CubeFeti((1, 1, 1), nel, contact=False)
Kreg, piv, rho = pa.MatRegularize(ctx, g.Ki, g.R)
n = Kreg.shape[0]
sys.stderr.write("n = %d, nnz = %d; factoring ...\n" % (n, Kreg.nnz))
stop = False


def ticker():
    t0 = time.time()
    while not stop:
        time.sleep(30)
        sys.stderr.write("  ... %.0f s, max RSS %.1f GB\n" % (time.time() - t0, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1048576.0))
        sys.stderr.flush()


threading.Thread(target=ticker, daemon=True).start()
t0 = time.perf_counter()
lu = spla.splu(Kreg.tocsc(), permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
t_fac = time.perf_counter() - t0
rhs = np.random.default_rng(3).standard_normal(n)
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    x = lu.solve(rhs)
    ts.append(time.perf_counter() - t0)
stop = True
res = float(np.linalg.norm(Kreg @ x - rhs) / np.linalg.norm(rhs))
cpu = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.lower().startswith("model name")][:1]
d = dict(nel=nel, n=int(n), nnz=int(Kreg.nnz), factor_seconds=t_fac, solve_seconds=ts, solve_seconds_median=float(np.median(ts)), factor_nnz=int(lu.L.nnz + lu.U.nnz), residual=res,
         max_rss_GB=resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1048576.0, cpu_model=cpu[0] if cpu else None, threads=1,
         solver="scipy.sparse.linalg.splu (SuperLU, MMD_AT_PLUS_A, SymmetricMode, no pivoting) on K_reg = MatRegularize(K, R)")
json.dump(d, open(out, "w"), indent=1)
print(json.dumps(d))
