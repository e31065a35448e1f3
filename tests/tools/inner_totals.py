"""Round 5, verdict #8: SMALXE inner-iteration totals of the 12 random equality-constrained problems of tests/test_gpu_random_parity.py against the oracle, per product path
(fused chain / separate projector launches).  Prints one line per seed."""
import sys
sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import numpy as np
import permon_amd as pa
from test_gpu_random_parity import _eq_problem
from oracle import oracle

oracle.build()

ctx = pa.Context(0)
for seed in range(12):
    A, G, orth, b, lb = _eq_problem(seed)
    n = A.shape[0]
    pfo = oracle.Qppf(oracle.Csr.from_scipy(G), orthonormal=orth)
    ref = oracle.smalxe(oracle.Op(n, csr=oracle.Csr.from_scipy(A)), b, np.zeros(n), oracle.Box(n, lb=lb), pfo, rtol=1e-7)
    row = []
    for chain, gtf, unf in ((1, 1, False), (0, 1, False), (0, 0, False), (0, 0, True)):
        ctx.L.pmh_set_knob(b"chain", chain)
        ctx.L.pmh_set_knob(b"gt_fusion", gtf)
        Ad = pa.CsrMat(ctx, n, n, A.indptr, A.indices, A.data)
        qp = pa.QP(ctx)
        qp.SetOperator(pa.Op.from_csr(Ad))
        qp.SetRhs(ctx.vec_from(b))
        x = ctx.vec(n)
        qp.SetInitialVector(x)
        qp.SetBox(None, ctx.vec_from(lb), None)
        qp.SetEq(pa.QPPF.from_scipy(ctx, G, orthonormal=orth))
        qps = pa.QPS(ctx)
        qps.SetQP(qp)
        qps.SetType("smalxe")
        qps.SetTolerances(rtol=1e-7)
        if unf:
            qps.MPGPSetUnfused(True)
        st = qps.Solve()
        row.append(st.inner_iter_accu - ref["inner_iter_accu"])
    print(seed, n, G.shape[0], orth, ref["inner_iter_accu"], row, flush=True)
