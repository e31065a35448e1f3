"""Times the vector-SpMV variants (PMH_SPMV_VTUNE = lanes,nt) on Q1-elasticity subdomain blocks K_i (81 nnz/row)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import permon_amd as pa
from permon_amd.feti import CubeFeti
nel = int(sys.argv[1]) if len(sys.argv) > 1 else 43
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
f = CubeFeti((1, 1, 1), nel, contact=False)
K = sp.block_diag([f.Ki] * nb, format="csr"); K.sort_indices()
n = K.shape[0]
print("n", n, "nnz", K.nnz, "nnz/row %.1f" % (K.nnz / n))
ctx = pa.Context(0)
x = ctx.vec_from(np.random.default_rng(0).standard_normal(n)); y = ctx.vec(n)
bytes_ = 12.0 * K.nnz + 20.0 * n
ref = None
cfgs = [("0,0,0,0", "16,0")] + [("1,%d,%d,%d" % (nnzb, mode, nt), "16,0") for nnzb in (1024, 2048, 4096) for mode in (0, 2) for nt in (0, 1)]
for mt, vt in cfgs:
    for _ in (0,):
        os.environ["PMH_SPMV_MTUNE"] = mt
        os.environ["PMH_SPMV_VTUNE"] = vt
        lanes, nt = mt, 0
        A = pa.CsrMat(ctx, n, n, K.indptr, K.indices, K.data)
        for _ in range(3): A.mult(x, y)
        ctx.sync(); ctx.timer_start()
        reps = 20
        for _ in range(reps): A.mult(x, y)
        ms = ctx.timer_stop() / reps
        yy = y.to_numpy()
        if ref is None: ref = yy
        print("mtune=%s nt=%d  %.1f us  %.0f GB/s (%.1f%% of 8 TB/s) maxdiff=%.1e" % (lanes, nt, ms * 1e3, bytes_ / ms / 1e6, bytes_ / ms / 1e6 / 80, np.max(np.abs(yy - ref))), flush=True)
        A.destroy()
