"""Times the dense apply kernels of the explicit local dual operators at configs[2] size without the set-up solves (the storage is
filled with a byte pattern).  usage: python scripts/symv_tune.py [sym|full|class|class_sym|class_orbit] [nel] [blocks] [share of N]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import permon_amd as pa  # noqa: E402
from permon_amd._lib import check  # noqa: E402

storage = sys.argv[1] if len(sys.argv) > 1 else "sym"
nel = int(sys.argv[2]) if len(sys.argv) > 2 else 43
nblk = int(sys.argv[3]) if len(sys.argv) > 3 else 8
ctx = pa.Context(0)
f = pa.CubeFeti((2, 2, 2), nel, contact=True)
loc = f.subset(range(nblk))
import scipy.sparse as sp  # noqa: E402

# the explicit operator only needs the block structure: a diagonal stand-in for K keeps the upload small
K = pa.MatBlockDiag.from_scipy(ctx, loc["block_rowstart"], sp.identity(loc["n_x"], format="csr"))
B = pa.MatGluing(ctx, loc["n_x"], f.n_lambda, loc["leaves_row"], loc["leaves_root"], loc["leaves_sign"])
E = pa.MatExplicitDual(B, K, storage=storage, block_class=np.zeros(nblk, dtype=np.int32))  # "class": the cubes are congruent
if len(sys.argv) > 4:  # rank 0's share of an N-GPU run
    E.set_stripe(0, int(sys.argv[4]))
if storage == "class_orbit":  # needs the cube's symmetries before anything is planned
    print("symmetries used:", E.set_box_symmetry(0, (nel + 1,) * 3, 3, f.Ki))
check(ctx.L.pmh_fexplicit_fill_pattern(E.h, 0x3C))
ntot, gs = E.compressed_size()
x, y = ctx.vec_from(np.random.default_rng(0).standard_normal(ntot)), ctx.vec(ntot)
for _ in range(3):
    E.dense_mult(x, y)
E.timing_enable(64)
for _ in range(30):
    E.dense_mult(x, y)
n, ms, b = E.timing_get()
E.refresh_sizes()
b = E.gemv_bytes
if storage == "class_orbit":
    print("orbit GEMM: %.3f ms per dense apply, %.1f TFLOP/s useful of the fp64 matrix peak 78.6" % (ms / n, E.apply_flops() / (ms / n * 1e-3) / 1e12))
print("%s nel=%d blocks=%d n_gamma=%s  stored %.2f GB  avg %.3f ms (first kernel %.3f ms)  %.0f GB/s algorithmic = %.3f of 8 TB/s; first kernel alone on the stored bytes: %.0f GB/s"
      % (storage, nel, nblk, E.n_gamma.tolist(), E.dense_bytes / 1e9, ms / n, E.first_kernel_ms / n, b / (ms / n * 1e-3) / 1e9, b / (ms / n * 1e-3) / 8e12, E.dense_bytes / (E.first_kernel_ms / n * 1e-3) / 1e9))
ctx.close()
