"""The FETI chain against the reference's own tutorial goldens (src/tutorials/feti/output/ex71_*.out):
-pde_type Poisson -cells 7,8,9 -dim 3 on 6 ranks needs 16 / 9 / 9 dual CG iterations with -feti_gluing_type
nonred / full / orth, and the 7-rank elasticity case 66 / 26 with -dual_pc_dual_type none / lumped.

The oracle half runs on the CPU (-m "not gpu"); the product half solves the same problems through the C ABI
(gluing -> K^+ -> F -> QPS of type ksp) on the GPU.
The Poisson counts are reproduced exactly.  The elasticity counts are not exactly reproducible outside the
reference's arithmetic: the slab decomposition (7 x 1 x 1, one element thick) is so ill conditioned that a 1e-14
relative perturbation of the right-hand side moves the count by one or two, so those are asserted with a margin.
"""
import numpy as np
import pytest

from permon_amd.feti import DmdaFeti, gluing_links

POISSON = {"nonred": "feti_ex71_1_nonred", "full": "feti_ex71_1_full", "orth": "feti_ex71_1_orth"}
ELAST = {False: "feti_ex71_2_none", True: "feti_ex71_2_lumped"}


def _golden_its(goldens, key):
    s = goldens[key]["solves"][0]
    assert s["reason_name"] == "CONVERGED_RTOL"
    return s["iterations"]


def test_gluing_links_rules():
    # degree 2: one link, the same for every type up to the orth normalisation (which equals 1/sqrt(2))
    for t in ("nonred", "full", "orth"):
        (l,) = gluing_links(2, t)
        assert [c for c, _ in l] == [0, 1] and np.allclose([v for _, v in l], [2 ** -0.5, -(2 ** -0.5)])
    assert len(gluing_links(4, "nonred")) == 3 and len(gluing_links(4, "full")) == 6 and len(gluing_links(4, "orth")) == 3
    assert [[c for c, _ in l] for l in gluing_links(4, "nonred")] == [[0, 1], [0, 2], [0, 3]]  # star around the lowest rank
    assert [[c for c, _ in l] for l in gluing_links(3, "full")] == [[0, 1], [0, 2], [1, 2]]
    # orth rows are orthonormal and annihilate the constant (continuity across all copies)
    for m in (3, 4, 8):
        B = np.zeros((m - 1, m))
        for r, l in enumerate(gluing_links(m, "orth")):
            for c, v in l:
                B[r, c] = v
        assert np.allclose(B @ B.T, np.eye(m - 1)) and np.allclose(B @ np.ones(m), 0.0)
    with pytest.raises(ValueError):
        gluing_links(3, "bogus")


def test_dmda_decomposition_matches_petsc_rules():
    p = DmdaFeti((7, 8, 9), 6, "poisson", "nonred")
    assert p.procs == (1, 2, 3)
    assert [K.shape[0] for K in p.blocks] == [8 * 5 * 4, 8 * 5 * 4, 8 * 5 * 4, 8 * 5 * 4, 8 * 5 * 4, 8 * 5 * 4]
    assert p.B.shape == (240, 960) and p.kdim == 0  # every subdomain touches x = 0: nothing floats
    e = DmdaFeti((8, 6, 4), 7, "elasticity")
    assert e.procs == (7, 1, 1)
    assert [K.shape[0] for K in e.blocks] == [210, 315, 210, 210, 210, 210, 210]
    G, _ = e.coarse()
    assert G.shape == (36, 630)
    K = e.K
    for s in range(1, 7):  # floating slabs: K R = 0
        sl = slice(e.block_rowstart[s], e.block_rowstart[s + 1])
        assert np.abs(K[sl, sl] @ e.Rblocks[s].T).max() < 1e-12


def _oracle_dual(oracle, prob, rtol_k=1e-13, kernel_tol=0.0):
    K = oracle.Csr.from_scipy(prob.K)
    R = prob.R if prob.kdim else None
    Kp = oracle.MatInv(K, prob.block_rowstart, R, rtol=rtol_k, kernel_tol=kernel_tol)
    B = oracle.Gluing(prob.N, prob.n_lambda, prob.leaves_row, prob.leaves_root, prob.leaves_sign)
    F = oracle.FetiOp(B, Kp, None, which=0)
    d = B.mult_transpose(Kp.mult(prob.f))
    return K, Kp, B, F, d


@pytest.mark.parametrize("gtype", ["nonred", "full", "orth"])
def test_oracle_ex71_poisson_iteration_goldens(oracle, goldens, gtype):
    prob = DmdaFeti((7, 8, 9), 6, "poisson", gtype)
    _, _, _, F, d = _oracle_dual(oracle, prob)
    res = oracle.pcpg(F.op, d, np.zeros(prob.n_lambda), None, rtol=1e-5)
    assert res["reason"] == 2  # KSP_CONVERGED_RTOL
    assert res["iteration"] == _golden_its(goldens, POISSON[gtype])
    _check_kkt_lines(goldens, gtype, prob, res["rnorm"], np.linalg.norm(d))
    if gtype != "nonred":
        _, Kp, B, _, _ = _oracle_dual(oracle, prob)
        _check_assembled_line(goldens, gtype, prob, Kp.mult(prob.f - B.mult(res["x"])))


def _check_kkt_lines(goldens, gtype, prob, rnorm, normd):
    """The -qp_chain_view_kkt lines of the golden that the dual solve determines: `r = ||A*x - b||` of the dual QP
    (||F lambda - d|| and its ratio to ||d||) and `r = ||BE*x||` of the primal QP (B u = -(F lambda - d), ratio to ||f||).
    full / orth: reproduced to the printed digits (last digit +-1: the goldens print 3 significant digits of a number that
    went through MUMPS instead of our inner CG).  nonred: the iteration count is reproduced but the final residual is not
    (2.36e-04 here, 1.73e-04 in the golden): CG's residual at the stopping iteration depends on WHICH copy of a degree-4
    node the three non-redundant links are centred on (star around the lowest rank by our reading of qpfeti.c:643-648,
    which then sorts the link ids per rank, :708) -- for nonred that detail of B is therefore NOT pinned, only the count."""
    kkt = goldens[POISSON[gtype]]["kkt"]
    g_dual, g_be = kkt[0], kkt[3]
    assert g_dual["name"] == "||A*x - b||" and g_be["name"] == "||BE*x||"
    if gtype == "nonred":
        assert abs(normd - float(g_dual["r"]) / float(g_dual["r_rel"])) < 0.15  # ||d|| is reproduced (29.2)
        return

    assert _close(rnorm, g_dual["r"]) and _close(rnorm / normd, g_dual["r_rel"])
    assert _close(rnorm, g_be["r"]) and _close(rnorm / np.linalg.norm(prob.f), g_be["r_rel"])


def _close(val, printed, rel=6e-3):
    """Equal to the golden's printed 3 significant digits up to 0.6 % (our dual residual at the stopping iteration is
    1.4148e-04 where the MUMPS-backed reference printed 1.41e-04 / 4.59e-06, i.e. about 1.413e-04)."""
    p = float(printed)
    return abs(val - p) <= rel * p + 0.5e-2 * 10 ** np.floor(np.log10(p))


def _check_assembled_line(goldens, gtype, prob, u):
    """Last KKT line of the golden: r = ||A*x - b|| of the ORIGINAL (assembled, MATIS) QP after QPTPostSolve_QPTMatISToBlockDiag
    (qptransform.c:1905-1982): x is assembled by a reverse scatter with INSERT_VALUES -- one copy of every shared node wins,
    nothing is averaged -- and A, b are the assembled operator / right-hand side.  Averaging the copies would print 6.5e-05."""
    import scipy.sparse as sp

    g = goldens[POISSON[gtype]]["kkt"][6]
    assert g["name"] == "||A*x - b||"
    gids = np.concatenate(prob.gids) if isinstance(prob.gids, (list, tuple)) else np.asarray(prob.gids)
    ng = int(gids.max()) + 1
    Rg = sp.csr_matrix((np.ones(prob.N), (np.arange(prob.N), gids)), shape=(prob.N, ng))
    A, b = (Rg.T @ prob.K @ Rg).tocsr(), Rg.T @ prob.f
    assert _close(np.linalg.norm(b), "%.4g" % (float(g["r"]) / float(g["r_rel"])), rel=1e-2)
    vals = []
    for order in (range(prob.N), range(prob.N - 1, -1, -1)):  # which copy wins is not specified by VecScatter: both bracket the golden
        x = np.zeros(ng)
        for i in order:
            x[gids[i]] = u[i]
        vals.append(np.linalg.norm(A @ x - b))
    assert any(_close(v, g["r"]) and _close(v / np.linalg.norm(b), g["r_rel"]) for v in vals), (vals, g)
    cnt = np.asarray(Rg.sum(axis=0)).ravel()
    assert np.linalg.norm(A @ ((Rg.T @ u) / cnt) - b) < 0.5 * min(vals)  # the averaged vector is a different (better) one


@pytest.mark.parametrize("lumped", [False, True])
def test_oracle_ex71_elasticity_iteration_goldens(oracle, goldens, lumped):
    prob = DmdaFeti((8, 6, 4), 7, "elasticity")
    # the interior slabs' load lies in the kernel of their K_b altogether: the iterative K^+ runs with the rule for such loads switched ON (the oracle's default is the plain
    # KSPCG of the reference's iterative MATINV, which diverges along the kernel there) -- and is checked below against a K^+ that has no such rule, numpy's dense pinv
    K, Kp, B, F, d = _oracle_dual(oracle, prob, kernel_tol=64.0)
    G, e = prob.coarse()
    pf = oracle.Qppf(oracle.Csr.from_scipy(G))
    lam_t = pf.half_Q_transpose(e)
    b = pf.P(d - F.op(lam_t))
    A = oracle.Op(prob.n_lambda, fn=lambda x: pf.P(F.op(x)))  # P F, QPTEnforceEqByProjector for an eq.-only QP
    Ks = prob.K
    pc = (lambda w: pf.P(B.mult_transpose(Ks @ B.mult(w)))) if lumped else None  # P (B K B')
    res = oracle.pcpg(A, b, np.zeros(prob.n_lambda), None, rtol=1e-6, pc=pc)
    assert res["reason"] == 2
    # golden 66 / 26; the oracle's Moore-Penrose K^+ gives 64 / 27 -- what the product gives on the Moore-Penrose AND on the left generalised inverse (tests/test_gpu_kspfeti.py) --
    # since its block CG, like the product's, no longer iterates on the rounding residue of a load that lies in the kernel (67 / 29 before)
    assert res["iteration"] == (27 if lumped else 64) and abs(res["iteration"] - _golden_its(goldens, ELAST[lumped])) <= 2
    # golden KKT line 1 of ex71_2_*: rO/||b|| with ||b|| = ||P b_bar|| = 2.00e-04/9.79e-07 = 1.41e-04/6.90e-07 = 204.3
    assert abs(np.linalg.norm(b) - 204.3) < 0.5
    # INDEPENDENT of the iterative K^+ and of its rule for loads in the kernel: the same chain on the dense Moore-Penrose inverse of every block (numpy SVD).  Same operator
    # to 1e-9, same count -- so the 64 / 27 against the golden's 66 / 26 is not an artefact of the block CG (the golden ran on MUMPS' factorisation with null pivots)
    rs = np.asarray(prob.block_rowstart)
    Kd = prob.K.tocsr()
    pinvs = [np.linalg.pinv(Kd[rs[i]:rs[i + 1], rs[i]:rs[i + 1]].toarray(), rcond=1e-10, hermitian=True) for i in range(len(rs) - 1)]

    def kplus_dense(f):
        return np.concatenate([pinvs[i] @ f[rs[i]:rs[i + 1]] for i in range(len(rs) - 1)])

    Fd = lambda x: B.mult_transpose(kplus_dense(B.mult(x)))  # noqa: E731
    dd = B.mult_transpose(kplus_dense(prob.f))
    assert np.linalg.norm(dd - d) <= 1e-8 * np.linalg.norm(d)
    bd = pf.P(dd - Fd(lam_t))
    Ad = oracle.Op(prob.n_lambda, fn=lambda x: pf.P(Fd(x)))
    resd = oracle.pcpg(Ad, bd, np.zeros(prob.n_lambda), None, rtol=1e-6, pc=pc)
    assert resd["reason"] == 2 and abs(resd["iteration"] - res["iteration"]) <= 1, (resd["iteration"], res["iteration"])
    assert np.linalg.norm(resd["x"] - res["x"]) <= 1e-4 * np.linalg.norm(res["x"])
    # without the rule the plain block CG iterates on the rounding residue of the projected load (what round 3 measured: 67-87 iterations)
    _, _, _, F0, d0 = _oracle_dual(oracle, prob)
    assert np.linalg.norm(d0 - d) > 1e-9 * np.linalg.norm(d)


# ---- product path ----------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ctx():
    import permon_amd as pa

    c = pa.Context(0)
    yield c
    c.close()


def _dual_qp(ctx, prob, kplus_rtol=1e-13):
    from permon_amd.chain import FetiDualQP

    G, e = prob.coarse()
    return FetiDualQP(ctx, prob.local(), G, e, prob.c, prob.lb, orthonormal=False, kplus_rtol=kplus_rtol)


@pytest.mark.gpu
@pytest.mark.parametrize("gtype", ["nonred", "full", "orth"])
def test_gpu_ex71_poisson_iteration_goldens(ctx, oracle, goldens, gtype):
    prob = DmdaFeti((7, 8, 9), 6, "poisson", gtype)
    dq = _dual_qp(ctx, prob)
    st = dq.solve_ksp(rtol=1e-5)
    assert st.reason == 2
    assert st.iteration == _golden_its(goldens, POISSON[gtype])
    _check_kkt_lines(goldens, gtype, prob, st.rnorm, np.linalg.norm(dq.d.to_numpy()))
    u, Fl = dq.primal_solution(None)  # B u - c = -(F lambda - d): the golden's ||BE*x|| equals its dual residual
    assert abs(np.linalg.norm(prob.B @ u) - st.rnorm) <= 1e-3 * st.rnorm
    if gtype != "nonred":
        _check_assembled_line(goldens, gtype, prob, u)
    # and the multipliers agree with the oracle's solve of the same QP
    _, _, _, F, d = _oracle_dual(oracle, prob)
    ref = oracle.pcpg(F.op, d, np.zeros(prob.n_lambda), None, rtol=1e-5)
    lam = dq.dual_solution()
    assert np.linalg.norm(lam - ref["x"]) <= 1e-4 * np.linalg.norm(ref["x"])  # both stopped at rtol 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("lumped", [False, True])
def test_gpu_ex71_elasticity_iteration_goldens(ctx, goldens, lumped):
    prob = DmdaFeti((8, 6, 4), 7, "elasticity")
    dq = _dual_qp(ctx, prob)
    st = dq.solve_ksp(rtol=1e-6, lumped=lumped)
    assert st.reason == 2
    # exactly 64 / 26 on this chain (the Moore-Penrose form P_R K^- P_R; the library's arithmetic is deterministic); pmh_kspfeti_solve's default, the LEFT generalised
    # inverse, takes 64 / 27 (test_gpu_kspfeti.py, = the CPU oracle).  The golden's 66 / 26 ran on MUMPS' null pivots and lies within the +-2 that rounding at the
    # stopping iteration moves this count by (profiles/r04_ex71_2_residual_history.txt)
    assert st.iteration == {False: 64, True: 26}[lumped], st.iteration
    assert abs(st.iteration - _golden_its(goldens, ELAST[lumped])) <= 2
    assert abs(np.linalg.norm(dq.b.to_numpy()) - 204.3) < 0.5
    # solution check: primal residual of the recovered u as the reference's last KKT line (r/||b|| ~ 2e-05)
    u, Fl = dq.primal_solution(None)
    lam = dq.dual_solution()
    G, e = prob.coarse()
    alpha = -np.linalg.solve((G @ G.T).toarray(), G @ Fl)  # G' alpha = d - F lambda
    Rm = np.zeros((G.shape[0], prob.N))
    r0 = 0
    for s, R in enumerate(prob.Rblocks):
        Rm[r0:r0 + R.shape[0], prob.block_rowstart[s]:prob.block_rowstart[s + 1]] = R
        r0 += R.shape[0]
    u = u - Rm.T @ alpha
    res = prob.K @ u - prob.f + prob.B.T @ lam
    assert np.linalg.norm(res) <= 1e-4 * np.linalg.norm(prob.f)
    assert np.linalg.norm(prob.B @ u) <= 1e-3 * np.linalg.norm(u)


def test_nonred_final_residual_is_not_pinned_by_the_gluing():
    """VERDICT r3 next #7(iii).  feti/output/ex71_1_feti_gluing_type-nonred.out ends at ||F lambda - d|| = 1.73e-04 after 16 iterations; this library stops after the same 16
    iterations at 2.36e-04 (GPU) / 1.80e-04 (a host CG on the same B with a sparse direct K^{-1}).  Why that number cannot be pinned:
      (a) the B of pmh_feti_gluing_from_l2g IS the reading of qpfeti.c:643-648 / :786-806 (m - 1 links per multi-node, every link between the LOWEST rank's copy and one other
          copy in rank order, +-1/sqrt(m), the sort of the link ids at :708 a no-op because they are issued in global dof order) -- compared row for row below;
      (b) CG's residual history on that B is reproducible to 4 digits up to iteration 15 and rounding-determined from iteration 16 on: a 1e-14 relative perturbation of f moves the
          16th residual by several percent and the 17th by a factor of 2-3 (the Krylov space of the separated part of the spectrum is exhausted after 15 steps; what is left is
          at the level the rounding of K^{-1} feeds in).  The golden's 1.73e-04 lies inside the spread of those perturbed runs.
    full / orth stop at iteration 9, well before that regime, and are reproduced to the printed digits (test_*_ex71_poisson_iteration_goldens)."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla

    prob = DmdaFeti((7, 8, 9), 6, "poisson", "nonred")
    N, nd = prob.N, prob.ndof
    l2g = np.concatenate([(np.asarray(g)[:, None] * nd + np.arange(nd)[None, :]).ravel() for g in prob.gids])
    copies = {}
    for i, g in enumerate(l2g):  # local dofs are numbered subdomain after subdomain: the copies of a global dof come out in rank order
        copies.setdefault(int(g), []).append(i)
    want = set()
    for c in copies.values():
        m = len(c)
        for k in range(1, m):
            want.add((c[0], c[k], round(1.0 / np.sqrt(m), 12)))
    B = prob.B.tocsr()
    got = set()
    for r in range(B.shape[0]):
        idx, val = B.indices[B.indptr[r]:B.indptr[r + 1]], B.data[B.indptr[r]:B.indptr[r + 1]]
        assert idx.size == 2 and val[0] == -val[1]
        o = np.argsort(idx)
        assert val[o[0]] > 0  # +1 on the lower rank, -1 on the highest rank of the link (qpfeti.c:789-793)
        got.add((int(idx[o[0]]), int(idx[o[1]]), round(abs(float(val[0])), 12)))
    assert got == want and len(want) == 240

    lu = spla.splu(prob.K.tocsc())

    def history(pert, nit=18):
        f = prob.f * (1.0 + pert * np.random.default_rng(1).standard_normal(N))
        F = lambda v: B @ lu.solve(B.T @ v)  # noqa: E731
        d = B @ lu.solve(f)
        x, r = np.zeros(B.shape[0]), d.copy()
        p, rr, h = r.copy(), float(r @ r), []
        for _ in range(nit):
            Ap = F(p)
            a = rr / float(p @ Ap)
            x += a * p
            r -= a * Ap
            rn = float(r @ r)
            p = r + (rn / rr) * p
            rr = rn
            h.append(np.sqrt(rr))
        return np.array(h), float(np.linalg.norm(d))

    runs = [history(p) for p in (0.0, 1e-14, 1e-13, 1e-12, 1e-11)]
    h0, nd0 = runs[0]
    assert abs(nd0 - 29.2) < 0.15
    assert h0[14] > 1e-5 * nd0 >= h0[15]  # 16 iterations at rtol 1e-5, as the golden says
    for h, _ in runs[1:]:
        assert np.allclose(h[:15], h0[:15], rtol=2e-3)  # reproducible up to iteration 15
    r16 = np.array([h[15] for h, _ in runs])
    r17 = np.array([h[16] for h, _ in runs])
    assert r16.max() / r16.min() > 1.02 and r17.max() / r17.min() > 1.5  # rounding-determined from iteration 16 on
    assert 0.9 * r16.min() <= 1.73e-4 <= 1.1 * r16.max()  # the golden's value is one sample of that spread
