"""KSPFETI driver (pmh_kspfeti_solve, csrc/kspfeti.hip = KSPFETISetUp + KSPSolve_FETI, src/ksp/impls/feti/feti.c:71-156):
the reference's ex71 goldens through ONE C call, and the recovered primal solution against a direct solve of the assembled
(undecomposed) problem -- a check that does not involve the oracle's FETI chain at all."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import permon_amd as pa
from permon_amd.feti import CubeFeti, DmdaFeti

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = pa.Context(0)
    yield c
    c.close()


def _dmda_l2g(prob):
    nd = prob.ndof
    return np.concatenate([(np.asarray(g)[:, None] * nd + np.arange(nd)[None, :]).ravel() for g in prob.gids]).astype(np.int32)


def _assembled(prob, l2g):
    ng = int(l2g.max()) + 1
    Rg = sp.csr_matrix((np.ones(prob.N), (np.arange(prob.N), l2g)), shape=(prob.N, ng))
    return Rg, (Rg.T @ prob.K @ Rg).tocsc(), Rg.T @ prob.f


@pytest.mark.parametrize("explicit", [False, True])
@pytest.mark.parametrize("gtype,its", [("nonred", 16), ("full", 9), ("orth", 9)])
def test_ex71_poisson_goldens_through_the_driver(ctx, goldens, gtype, its, explicit):
    """explicit=True: F applies through the explicit local dual operators (the exact K^+ path, pmh_fexplicit_*): same golden counts."""
    prob = DmdaFeti((7, 8, 9), 6, "poisson", gtype)
    l2g = _dmda_l2g(prob)
    # configured by the reference's own command line (feti/ex71.c:433-438)
    u, lam, st = pa.KSPFETISolve(ctx, prob.block_rowstart, prob.K, prob.f, l2g, explicit=explicit,  # Dirichlet is in K, nothing floats
                                 options="-qps_view_convergence -qp_chain_view_kkt -pde_type Poisson -cells 7,8,9 -dim 3 -feti_gluing_type %s" % gtype)
    assert (st.reason, st.iteration) == (2, its) and st.iteration == goldens["feti_ex71_1_" + gtype]["solves"][0]["iterations"]
    assert st.coarse_dim == 0 and st.n_dirichlet_rows == 0 and st.n_lambda == prob.n_lambda
    if gtype != "nonred":
        k = goldens["feti_ex71_1_" + gtype]["kkt"][0]
        assert abs(st.rnorm - float(k["r"])) <= 6e-3 * float(k["r"]) + 5e-7  # ||F lambda - d|| = 1.41e-04 of the golden
    # primal solution against the direct solve of the assembled problem (stopped at rtol 1e-5 of the dual residual)
    Rg, A, b = _assembled(prob, l2g)
    x = spla.spsolve(A, b)
    assert np.linalg.norm(u - Rg @ x) <= 2e-4 * np.linalg.norm(x)


@pytest.mark.parametrize("kplus,explicit", [("reg", False), ("mp", False), ("left", False), ("reg", True)])
@pytest.mark.parametrize("lumped", [False, True])
def test_ex71_elasticity_floating_slabs(ctx, goldens, kplus, lumped, explicit, monkeypatch, capfd):
    """7 slabs, 6 of them floating (coarse problem of 36); golden counts 66 (none) / 26 (lumped).  KSPFETI hands the reference no kernel, so the golden ran on the LEFT generalised
    inverse K^- P_R with MUMPS' null pivots (qptransform.c:997-1008; the golden's ||d|| = 17.4 rules K_reg^{-1} out: 2 230 there, 8.6 on the left inverse with MatRegularize's
    fixing dofs).  The projected operator P F P is the same for every generalised inverse, the counts are not pinned by it: 64 / 27 on the left inverse and on the Moore-Penrose
    form (which agree with each other, as they must), 66 / 27 on K_reg^{-1}; the residual stalls around the threshold at the stopping iteration
    (profiles/r04_ex71_2_residual_history.txt).  Asserted: the golden count within +-2 / +-1, left == mp.
    Round 5 pins the stall itself: from the -ksp_monitor trace (PMH_KSP_MONITOR) the residual AT THE GOLDEN'S ITERATION is within 1.3 x of the threshold whenever this solve is
    still iterating there (lumped: 2.54e-4 against 2.04e-4 at iteration 26), and the iteration before this solve's last one is above the threshold by less than 3 x.
    Until round 4 the Moore-Penrose form took 66-87 / 29-35 here: the interior slabs' load lies in the kernel altogether, and the block CG iterated on the rounding residue of its
    projection (tests/test_gpu_feti.py::test_matinv_load_in_the_kernel).
    explicit: the same through the explicit local dual operators (K_reg^{-1} on Gamma assembled at rtol 1e-13)."""
    prob = DmdaFeti((8, 6, 4), 7, "elasticity")
    l2g = _dmda_l2g(prob)
    extra = {"reg": " -qpt_dualize_Kplus_left 0", "mp": " -qpt_dualize_Kplus_mp", "left": ""}[kplus]  # left: the default (KSPFETI never supplies a kernel)
    monkeypatch.setenv("PMH_KSP_MONITOR", "1")
    capfd.readouterr()
    u, lam, st = pa.KSPFETISolve(ctx, prob.block_rowstart, prob.K, prob.f, l2g, R=prob.R, kplus_rtol=1e-14 if kplus == "reg" else 1e-13, explicit=explicit,
                                 options="-pde_type Elasticity -dim 3 -qps_rtol 1e-6 -dual_pc_dual_type %s%s" % ("lumped" if lumped else "none", extra))  # feti/ex71.c:442
    gold = goldens["feti_ex71_2_lumped" if lumped else "feti_ex71_2_none"]["solves"][0]["iterations"]
    trace = [ln.split() for ln in capfd.readouterr().err.splitlines() if "KSP Residual norm" in ln]
    hist = {int(t[0]): float(t[4]) for t in trace}
    ttol = float(trace[0][6].rstrip(")"))
    print("ex71_2 %s K+ %s explicit=%s: %d iterations (golden %d)" % ("lumped" if lumped else "none", kplus, explicit, st.iteration, gold))
    assert len(hist) == st.iteration + 1 and abs(ttol - 1e-6 * hist[0]) <= 1e-12 * ttol and hist[st.iteration] <= ttol < hist[st.iteration - 1]
    if gold < st.iteration:
        assert hist[gold] <= (1.35 if explicit else 1.3) * ttol, (hist[gold], ttol)  # the golden stopped where this residual is 1.24 x the threshold (1.32 x through the explicit operators)
    assert hist[st.iteration - 1] <= 3.0 * ttol, (hist[st.iteration - 1], ttol)  # the stall around the threshold: one iteration earlier is already close
    assert st.reason == 2 and st.coarse_dim == 36
    if not lumped:
        assert gold == 66 and st.iteration == (66 if kplus == "reg" else 64)
    else:
        assert gold == 26 and st.iteration == 27
    Rg, A, b = _assembled(prob, l2g)
    x = spla.spsolve(A, b)
    assert np.linalg.norm(u - Rg @ x) <= 1e-4 * np.linalg.norm(x)
    assert np.linalg.norm(prob.B @ u) <= 3e-4  # continuity across the interfaces: the dual residual at the stopping iteration (1.7e-4 ... 2.0e-4)


@pytest.mark.parametrize("gtype", ["full", "orth"])
def test_tfeti_dirichlet_rows_in_B(ctx, gtype):
    """Total FETI: Dirichlet dofs enforced by rows of B (KSPFETISetDirichlet(..., enforce_by_B), feti/ex1.c:89-90), every
    subdomain floats (coarse dimension 6 per cube).  Tight tolerances: u must equal the solution of the assembled problem
    with the Dirichlet dofs eliminated."""
    f = CubeFeti((2, 2, 1), 3, contact=False, gluing=gtype)
    # local-to-global map of the cubes from the generator's own B: rebuild from coordinates
    nn, ne = f.nel + 1, f.nel
    sx, sy, sz = f.sub
    GX, GY = sx * ne + 1, sy * ne + 1
    l2g, dirl = [], []
    for s in range(f.nsub):
        ix, iy, iz = s % sx, (s // sx) % sy, s // (sx * sy)
        for k in range(nn):
            for j in range(nn):
                for i in range(nn):
                    gnode = ((iz * ne + k) * GY + (iy * ne + j)) * GX + (ix * ne + i)
                    for c in range(3):
                        if ix * ne + i == 0:
                            dirl.append(len(l2g))
                        l2g.append(gnode * 3 + c)
    l2g = np.asarray(l2g, dtype=np.int32)
    assert l2g.size == f.N
    # right-hand side split among the copies (QPTMatISToBlockDiag qptransform.c:2095-2113): assembled load / multiplicity
    ng = int(l2g.max()) + 1
    Rg = sp.csr_matrix((np.ones(f.N), (np.arange(f.N), l2g)), shape=(f.N, ng))
    mult = np.asarray(Rg.sum(axis=0)).ravel()
    b = Rg.T @ f.f
    fsplit = (Rg @ (b / mult))
    u, lam, st = pa.KSPFETISolve(ctx, f.block_rowstart, f.K, fsplit, l2g, dirichlet_local=dirl, R=f.R, gluing=gtype, rtol=1e-10, kplus_rtol=1e-13)
    assert st.reason == 2 and st.coarse_dim == 6 * f.nsub and st.n_dirichlet_rows == len(dirl)
    A = (Rg.T @ f.K @ Rg).tocsr()
    free = np.setdiff1d(np.arange(ng), np.unique(l2g[dirl]))
    x = np.zeros(ng)
    x[free] = spla.spsolve(A[free][:, free].tocsc(), b[free])
    assert np.linalg.norm(u - Rg @ x) <= 1e-7 * np.linalg.norm(x)
    # same with the Moore-Penrose K^+ and with the Dirichlet dofs excluded from the gluing
    u2, _, st2 = pa.KSPFETISolve(ctx, f.block_rowstart, f.K, fsplit, l2g, dirichlet_local=dirl, R=f.R, gluing=gtype, regularize=False, exclude_dirichlet=True, rtol=1e-10, kplus_rtol=1e-13)
    assert st2.reason == 2 and st2.n_lambda < st.n_lambda
    assert np.linalg.norm(u2 - Rg @ x) <= 1e-7 * np.linalg.norm(x)


@pytest.mark.parametrize("dir_in_hess", [False, True])
def test_feti_ex1_tutorial_one_iteration(ctx, goldens, dir_in_hess):
    """src/tutorials/feti/ex1.c (-ne 7, 4 subdomains of a 1-D bar, -u'' = sin(pi u)): the goldens ex1_1.out / ex1_2.out say
    'PERMON FETI CONVERGED_RTOL in 1 iteration'.  MATIS input: local element matrices, ASSEMBLED right-hand side split by
    pmh_qpt_matis_split_rhs, Dirichlet ends by rows of B (default) or eliminated in the blocks (-dir_in_hess)."""
    import ctypes as C

    from permon_amd import _lib

    ns, ne_l = 4, 7
    nl, ng = ne_l + 1, ns * ne_l + 1
    h = 1.0 / (ns * ne_l)
    Ke = np.array([[1.0, -1.0], [-1.0, 1.0]])
    blocks, l2g = [], []
    b = np.zeros(ng)
    for r in range(ns):
        Ki = np.zeros((nl, nl))
        for i in range(ne_l):
            Ki[i:i + 2, i:i + 2] += Ke
            v = np.sin((r * ne_l + i + 0.5) * h * 3.14159) * 0.5 * h * h
            b[r * ne_l + i] += v
            b[r * ne_l + i + 1] += v
        blocks.append(Ki)
        l2g.append(r * ne_l + np.arange(nl))
    l2g = np.concatenate(l2g).astype(np.int32)
    N = l2g.size
    f = np.zeros(N)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    _lib.check(ctx.L.pmh_qpt_matis_split_rhs(N, p(l2g), ng, p(b), p(f)))
    rs = np.arange(ns + 1, dtype=np.int32) * nl
    dirl = [0, N - 1]  # global dofs 0 and ng-1 live in the first and the last subdomain only
    R = np.zeros((1, N))
    if dir_in_hess:  # MatZeroRowsColumns with diag = max |diag| (qpfeti.c:296-303): the two end subdomains no longer float
        for blk, i in ((0, 0), (ns - 1, nl - 1)):
            blocks[blk][i, :] = 0.0
            blocks[blk][:, i] = 0.0
            blocks[blk][i, i] = 2.0
        f[dirl] = 0.0
        R[0, nl:N - nl] = 1.0
    else:
        R[0, :] = 1.0
    K = sp.block_diag(blocks, format="csr")
    u, lam, st = pa.KSPFETISolve(ctx, rs, K, f, l2g, dirichlet_local=None if dir_in_hess else dirl, R=R, rtol=1e-5, kplus_rtol=1e-14)
    g = goldens["feti_ex1_2" if dir_in_hess else "feti_ex1_1"]
    assert st.reason == 2 and st.iteration == 1  # "PERMON FETI CONVERGED_RTOL in 1 iteration"
    assert st.coarse_dim == (ns - 2 if dir_in_hess else ns) and st.n_dirichlet_rows == (0 if dir_in_hess else 2)
    assert g["kkt"][-1]["name"] == "||A*x - b||"
    # the solution of the assembled problem with u(0) = u(1) = 0
    A = np.zeros((ng, ng))
    for e in range(ns * ne_l):
        A[e:e + 2, e:e + 2] += Ke
    x = np.zeros(ng)
    x[1:-1] = np.linalg.solve(A[1:-1, 1:-1], b[1:-1])
    xg = np.zeros(ng)
    _lib.check(ctx.L.pmh_qpt_matis_assemble_solution(N, p(l2g), p(np.ascontiguousarray(u)), ng, p(xg)))
    assert np.linalg.norm(xg - x) <= 1e-10 * np.linalg.norm(x)


@pytest.mark.parametrize("orth", ["none", "gs", "gslingen", "cholesky", "implicit"])
def test_unprojected_smalxe_chain_solves_the_same_problem(ctx, orth):
    """-project 0 (SMALXE on the dual QP with its equality constraint kept, qptransform.c:2185 / qps.c:437-441) against the projected CG chain on a 3-D TFETI problem with a
    24-dimensional coarse space: the same primal solution, for every way of orthonormalising G the driver offers (none: the dense (GG')^{-1} sits in the projector and the penalty
    term is rho G'G)."""
    f = CubeFeti((2, 2, 1), 3, contact=False, gluing="full")
    nn, ne = f.nel + 1, f.nel
    sx, sy, sz = f.sub
    GX, GY = sx * ne + 1, sy * ne + 1
    l2g, dirl = [], []
    for s in range(f.nsub):
        ix, iy, iz = s % sx, (s // sx) % sy, s // (sx * sy)
        for k in range(nn):
            for j in range(nn):
                for i in range(nn):
                    gnode = ((iz * ne + k) * GY + (iy * ne + j)) * GX + (ix * ne + i)
                    for c in range(3):
                        if ix * ne + i == 0:
                            dirl.append(len(l2g))
                        l2g.append(gnode * 3 + c)
    l2g = np.asarray(l2g, dtype=np.int32)
    ng = int(l2g.max()) + 1
    Rg = sp.csr_matrix((np.ones(f.N), (np.arange(f.N), l2g)), shape=(f.N, ng))
    mult = np.asarray(Rg.sum(axis=0)).ravel()
    fsplit = Rg @ ((Rg.T @ f.f) / mult)
    u0, _, st0 = pa.KSPFETISolve(ctx, f.block_rowstart, f.K, fsplit, l2g, dirichlet_local=dirl, R=f.R, rtol=1e-9, kplus_rtol=1e-13)
    u1, _, st1 = pa.KSPFETISolve(ctx, f.block_rowstart, f.K, fsplit, l2g, dirichlet_local=dirl, R=f.R, rtol=1e-9, kplus_rtol=1e-13,
                                 options="-project 0 -dual_qp_E_orth_type %s -qps_smalxe_rho 1e1" % orth)
    assert st0.reason > 0 and st1.reason > 0 and st1.smalxe.iteration == st1.iteration >= 1
    assert np.linalg.norm(u1 - u0) <= 1e-6 * np.linalg.norm(u0), (orth, np.linalg.norm(u1 - u0) / np.linalg.norm(u0))


def test_driver_and_chain_release_their_device_memory(ctx):
    """Every object pmh_kspfeti_solve creates (CSR copies, K_reg, MATINV work vectors, gluing, projector, chain) is released:
    repeated solves do not grow the HBM footprint (hipMemGetInfo through pmh_mem_info)."""
    prob = DmdaFeti((8, 6, 4), 7, "elasticity")
    l2g = _dmda_l2g(prob)

    def once():
        pa.KSPFETISolve(ctx, prob.block_rowstart, prob.K, prob.f, l2g, R=prob.R, rtol=1e-6)  # the default: K^- P_R
        pa.KSPFETISolve(ctx, prob.block_rowstart, prob.K, prob.f, l2g, R=prob.R, rtol=1e-6, regularize=True)
        pa.KSPFETISolve(ctx, prob.block_rowstart, prob.K, prob.f, l2g, R=prob.R, rtol=1e-6, regularize=False, lumped=True)
        for orth in ("gs", "implicit"):  # the unprojected chain: SMALXE, the orthonormalised projector, its explicit T G / implicit T
            _, _, st = pa.KSPFETISolve(ctx, prob.block_rowstart, prob.K, prob.f, l2g, R=prob.R, rtol=1e-6, options="-project 0 -dual_qp_E_orth_type %s -qp_chain_view_kkt" % orth)
            assert st.reason > 0 and st.smalxe.iteration == st.iteration
            assert ("not available" in st.view_text) == (orth == "implicit")  # ||BE x - cE|| of the implicitly orthonormalised QP (qp.c:303-318)

    once()
    free0, total = ctx.mem_info()
    assert total > 250e9  # an MI355X: 288 GB
    for _ in range(5):
        once()
    free1, _ = ctx.mem_info()
    assert free0 - free1 <= 8 << 20, (free0, free1)
