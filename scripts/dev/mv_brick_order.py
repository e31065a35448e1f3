"""Round 6 probe: does the multi-right-hand-side product gain from a node order with 3-D locality?  One 43^3-node block, K permuted symmetrically so that the nodes of a
b x b x b brick are consecutive (bricks in lexicographic order), timed through pmh_mv_test_spmv as scripts/micro/mv_spmv_time.py does.  b = 1: the generator's order."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.sparse as sp
import permon_amd as pa
from permon_amd._lib import check

nel = int(sys.argv[1]) if len(sys.argv) > 1 else 42
ctx = pa.Context(0)
f = pa.CubeFeti((1, 1, 1), nel, contact=False)
K0 = f.K.tocsr()
n = K0.shape[0]
nn = nel + 1
rng = np.random.default_rng(0)
for b in (1, 2, 4, 8):
    ix, iy, iz = np.meshgrid(np.arange(nn), np.arange(nn), np.arange(nn), indexing="ij")  # node id = ix + nn (iy + nn iz)?  take the generator's: lexicographic in (x fastest)
    node = (ix + nn * (iy + nn * iz)).ravel()
    key = np.lexsort(((ix % b).ravel(), (iy % b).ravel(), (iz % b).ravel(), (ix // b).ravel(), (iy // b).ravel(), (iz // b).ravel()))
    perm_nodes = node[key]  # new position -> old node
    perm = (3 * perm_nodes[:, None] + np.arange(3)[None, :]).ravel()
    Pm = sp.csr_matrix((np.ones(n), (np.arange(n), perm)), shape=(n, n))
    K = (Pm @ K0 @ Pm.T).tocsr()
    K.sort_indices()
    Ad = pa.CsrMat(ctx, n, n, K.indptr, K.indices, K.data)
    X = rng.standard_normal((n, 8))
    xd, yd = ctx.vec_from(X.reshape(-1)), ctx.vec(n * 8)
    ref = K @ X
    out = []
    for storage, name in ((0, "fp64"), (1, "fp32"), (2, "fp16")):
        ms = C.c_float(0)
        check(ctx.L.pmh_mv_test_spmv(Ad.h, storage, xd.p, yd.p, 50, C.byref(ms)))
        Y = yd.to_numpy().reshape(n, 8)
        out.append("%s %.4f ms (err %.1e)" % (name, ms.value, np.abs(Y - ref).max() / np.abs(ref).max()))
    print("brick %d: " % b + ", ".join(out), flush=True)
