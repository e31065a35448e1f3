"""Round 5: the multi-right-hand-side operator product (csrc/mv.hip) on ONE 43^3-node elasticity block, 8 columns: ms per launch for fp64 / fp32 / fp16 entries, and the
same block through the one-column kernels for comparison (k_bsr3 fp64 on 8 congruent copies = 8 columns: pmh_blockdiag_mult)."""
import ctypes as C
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import permon_amd as pa
from permon_amd._lib import check

nel = int(sys.argv[1]) if len(sys.argv) > 1 else 42
ctx = pa.Context(0)
f = pa.CubeFeti((1, 1, 1), nel, contact=False)
K = f.K.tocsr()
K.sort_indices()
n = K.shape[0]
print("block: %d dofs, %d non-zeros" % (n, K.nnz), flush=True)
Ad = pa.CsrMat(ctx, n, n, K.indptr, K.indices, K.data)
rng = np.random.default_rng(0)
X = rng.standard_normal((n, 8))
xd, yd = ctx.vec_from(X.reshape(-1)), ctx.vec(n * 8)
ref = K @ X
for storage, name in ((0, "fp64"), (1, "fp32"), (2, "fp16")):
    ms = C.c_float(0)
    check(ctx.L.pmh_mv_test_spmv(Ad.h, storage, xd.p, yd.p, 50, C.byref(ms)))
    Y = yd.to_numpy().reshape(n, 8)
    err = np.abs(Y - ref).max() / np.abs(ref).max()
    byts = K.nnz * {0: 8, 1: 4, 2: 2}[storage] * (12.0 / 9 if storage == 2 else 1) + n / 3 * 27 * 4 + 2 * n * 8 * (8 if storage == 0 else 4)
    print("mv %s entries, 8 columns: %.4f ms per product (%.0f GB/s of matrix + vector bytes), max error %.2e" % (name, ms.value, byts / ms.value / 1e6, err), flush=True)
