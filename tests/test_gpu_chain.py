"""The five-launch dual-space chain (permon_amd/csrc/dualchain.hip) of the penalised, projected FETI operator y = rho Q x + P F P x
(MatMult_Penalized src/qp/utils/matpenalized.c:12-22 over P F P, src/qp/interface/qptransform.c:273-284; QPPFApplyQ / P src/qppf/interface/qppf.c:454-575;
MatMult(Transpose)_Gluing src/mat/impls/gluing/gluing.c:47-159): against a dense numpy restatement of the same product, against the round-4 launch sequence
(pmh_set_knob("chain", 0)) and -- through SMALXE with the chain's emitted ||B u|| and in-kernel scalar reductions -- against the CPU oracle."""
import ctypes as C

import numpy as np
import pytest

import permon_amd as pa
from permon_amd._lib import check
from permon_amd.chain import FetiDualQP

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = pa.Context(0)
    yield c
    check(c.L.pmh_set_knob(b"chain", 1))
    c.close()


def _knob(ctx, name):
    v = C.c_int()
    check(ctx.L.pmh_get_knob(name.encode(), C.byref(v)))
    return v.value


KPLUS = {
    "orbit": lambda nn: dict(explicit=dict(rtol=1e-13, storage="class_orbit", symmetry=dict(dims=(nn, nn, nn), ndof=3))),
    "sym": lambda nn: dict(explicit=dict(rtol=1e-13, storage="sym")),
    "class_sym": lambda nn: dict(explicit=dict(rtol=1e-13, storage="class_sym")),
    "iterative": lambda nn: dict(),
}


def _problem(ctx, kplus, sub=(2, 2, 2), nel=3):
    f = pa.CubeFeti(sub, nel, contact=True)
    G0, e0 = f.coarse(orthonormalize=False)
    q = FetiDualQP(ctx, f.subset(range(f.nsub)), G0, e0, f.c, f.lb, orthonormal="implicit", kplus_rtol=1e-13, **KPLUS[kplus](nel + 1))
    return f, G0, q


def _dense(f, G0):
    Kp = np.linalg.pinv(f.Ki.toarray(), rcond=1e-10, hermitian=True)
    B = f.B.toarray()
    F = sum(B[:, s * f.n_i:(s + 1) * f.n_i] @ Kp @ B[:, s * f.n_i:(s + 1) * f.n_i].T for s in range(f.nsub))
    G = np.asarray(G0.todense()) if hasattr(G0, "todense") else np.asarray(G0)
    Q = G.T @ np.linalg.solve(G @ G.T, G)
    return F, Q


@pytest.mark.parametrize("kplus", ["orbit", "sym", "class_sym", "iterative"])
def test_chain_apply_vs_dense_and_round4(ctx, kplus):
    check(ctx.L.pmh_set_knob(b"chain", 1))
    f, G0, q = _problem(ctx, kplus)
    F, Q = _dense(f, G0)
    n = f.n_lambda
    P = np.eye(n) - Q
    rho = 2.5
    Aref = rho * Q + P @ F @ P
    rng = np.random.default_rng(3)
    Ap = pa.MatCreatePenalized(q.A, q.pf, rho)
    check(ctx.L.pmh_set_knob(b"chain", 0))
    Ap0 = pa.MatCreatePenalized(q.A, q.pf, rho)  # the round-4 sequence
    check(ctx.L.pmh_set_knob(b"chain", 1))
    y, y0 = ctx.vec(n), ctx.vec(n)
    for trial in range(3):
        x = rng.standard_normal(n)
        xv = ctx.vec_from(x)
        check(ctx.L.pmh_set_knob(b"chain_applies", 0)), check(ctx.L.pmh_set_knob(b"chain_launches", 0))
        Ap.mult(xv, y)
        assert _knob(ctx, "chain_applies") == 1 and _knob(ctx, "chain_launches") == 4  # emit, gather, scatter, final (+ the middle stage)
        Ap0.mult(xv, y0)
        assert _knob(ctx, "chain_applies") == 1
        ref = Aref @ x
        tol = 1e-9 if kplus == "iterative" else 1e-11
        assert np.linalg.norm(y.to_numpy() - ref) <= tol * np.linalg.norm(ref)
        assert np.linalg.norm(y.to_numpy() - y0.to_numpy()) <= tol * np.linalg.norm(ref)
        y2 = ctx.vec(n)
        Ap.mult(xv, y2)
        assert np.array_equal(y.to_numpy(), y2.to_numpy())  # fixed summation orders: bitwise reproducible, whoever draws the last ticket
        xv.free(), y2.free()
    Ap.destroy(), Ap0.destroy()


@pytest.mark.parametrize("kplus,sub,nel", [("orbit", (2, 2, 2), 3), ("sym", (2, 2, 1), 4), ("iterative", (2, 1, 2), 3), ("orbit", (2, 2, 2), 7)])
def test_chain_smalxe_vs_round4_and_oracle(ctx, oracle, kplus, sub, nel):
    """SMALXE + MPGP on the contact problem: the chain (G0 x / G0 p emitted by the vector kernels, ||B u|| from the emission, the MPGP scalars reduced by the last
    workgroup) takes the steps the round-4 sequence and the CPU oracle take."""
    check(ctx.L.pmh_set_knob(b"chain", 1))
    f, G0, q = _problem(ctx, kplus, sub, nel)
    n = f.n_lambda
    check(ctx.L.pmh_set_knob(b"chain_applies", 0)), check(ctx.L.pmh_set_knob(b"chain_launches", 0))
    q.lam.set(0.0)
    st = q.solve_smalxe()
    applies, launches = _knob(ctx, "chain_applies"), _knob(ctx, "chain_launches")
    lam1 = q.lam.to_numpy()
    inner1 = (st.inner.ncg, st.inner.nexp, st.inner.nprop, st.inner.nmv)
    q.qps.Destroy()
    assert st.reason > 0 and applies >= st.inner.nmv
    # inside the inner solver only the first product of a solve forms G0 x itself: close to three launches per application + two of the middle stage
    assert launches <= 3 * applies + 3 * st.iteration + 8, (launches, applies, st.iteration)
    check(ctx.L.pmh_set_knob(b"chain", 0))
    q.lam.set(0.0)
    st0 = q.solve_smalxe()
    lam0 = q.lam.to_numpy()
    inner0 = (st0.inner.ncg, st0.inner.nexp, st0.inner.nprop, st0.inner.nmv)
    q.qps.Destroy()
    check(ctx.L.pmh_set_knob(b"chain", 1))
    assert (st.reason, st.iteration, st.M1_updates, st.rho_updates) == (st0.reason, st0.iteration, st0.M1_updates, st0.rho_updates)
    assert (st.inner_iter_accu,) + inner1 == (st0.inner_iter_accu,) + inner0  # the same steps, one by one (round 6: equality, as in the random sweep)
    assert np.linalg.norm(lam1 - lam0) <= 1e-4 * np.linalg.norm(lam0)
    if nel > 4:
        return
    # the CPU oracle on the dense restatement of the same QP
    F, Q = _dense(f, G0)
    G = np.asarray(G0.todense()) if hasattr(G0, "todense") else np.asarray(G0)
    L = np.linalg.cholesky(G @ G.T)
    Gon = np.linalg.solve(L, G)  # T G0: orthonormal rows
    import scipy.sparse as sp

    pfo = oracle.Qppf(oracle.Csr.from_scipy(sp.csr_matrix(Gon)), orthonormal=True)
    A_or = oracle.Op(n, fn=lambda x: pfo.P(F @ pfo.P(x)))
    ref = oracle.smalxe(A_or, q.b.to_numpy(), np.zeros(n), oracle.Box(n, lb=q.lb_new.to_numpy()), pfo)
    assert (st.reason, st.iteration, st.M1_updates, st.rho_updates) == (ref["reason"], ref["iteration"], ref["M1_updates"], ref["rho_updates"])
    assert st.inner_iter_accu == ref["inner_iter_accu"]
    assert np.linalg.norm(lam1 - ref["u"]) <= 1e-4 * np.linalg.norm(ref["u"])


def test_chain_with_device_finalised_scalars():
    """MPGP with row-distributed scalars (`distributed` = 1: every reduction finalised on the DEVICE and all-reduced) on a chained operator: the chain leaves its block
    partials per 1024-entry tile for the host, so the operator must NOT take the vector phase into its last kernel for such a caller (PenalizedOp::mult_epi answers
    'unsupported' and MPGP runs its own vector kernels behind the chain's plain product).  On a 1-rank communicator the solve must take exactly the steps of the local mode."""
    import os

    os.environ["PMH_COMM_FORCE"] = "1"
    try:
        c = pa.Context(0)
        check(c.L.pmh_set_knob(b"chain", 1))
        c.comm_init(0, 1, c.comm_unique_id())
        f, G0, q = _problem(c, "orbit")
        res = []
        for dist in (0, 1):
            q.lam.set(0.0)
            st = q.solve_smalxe(inner=dict(distributed=dist))
            res.append(((st.reason, st.iteration, st.inner_iter_accu, st.inner.ncg, st.inner.nexp, st.inner.nprop, st.inner.nmv), q.lam.to_numpy().copy()))
            q.qps.Destroy()
        assert res[0][0] == res[1][0] and res[0][0][0] > 0
        assert np.linalg.norm(res[0][1] - res[1][1]) <= 1e-9 * np.linalg.norm(res[0][1])
        c.close()
    finally:
        os.environ.pop("PMH_COMM_FORCE", None)
