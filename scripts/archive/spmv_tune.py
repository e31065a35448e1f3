"""Times the stream-SpMV variants (PMH_SPMV_TUNE = nnzb,mode,nt) on the configs[1] matrix.  GPU only."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import permon_amd as pa
from permon_amd import problems as P

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 3162
ctx = pa.Context(0)
rp, ci, va = P.laplace2d_csr(grid, grid)
n = grid * grid
x = ctx.vec_from(np.random.default_rng(0).standard_normal(n))
y = ctx.vec(n)
bytes_ = 12.0 * va.size + 20.0 * n
ref = None
for nnzb, mode, nt in [(1024,2,1),(1024,2,3),(1024,0,1),(1024,0,3),(2048,0,1),(2048,0,3),(2048,2,3),(4096,0,3),(4096,2,3)]:
    os.environ["PMH_SPMV_TUNE"] = "%d,%d,%d" % (nnzb, mode, nt)
    try:
        A = pa.CsrMat(ctx, n, n, rp, ci, va)
    except Exception as e:
        print(nnzb, mode, nt, "ERR", e); continue
    for _ in range(5):
        A.mult(x, y)
    ctx.sync()
    ctx.timer_start()
    reps = 40
    for _ in range(reps):
        A.mult(x, y)
    ms = ctx.timer_stop() / reps
    yy = y.to_numpy()
    if ref is None:
        ref = yy
    ok = np.array_equal(yy, ref)
    print("nnzb=%4d mode=%d nt=%d  %.1f us  %.0f GB/s  (%.1f%% of 8 TB/s) same=%s" % (nnzb, mode, nt, ms * 1e3, bytes_ / ms / 1e6, bytes_ / ms / 1e6 / 80.0, ok), flush=True)
    A.destroy()
