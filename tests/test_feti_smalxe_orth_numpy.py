"""CPU pin of feti/output/ex1_smalxe_orth_dual_qp_E_orth_type-{gs,implicit}.out (src/tutorials/feti/ex1.c, TEST block smalxe_orth: 4 ranks, -ne 7, -project 0 -qps_smalxe_rho 1e1):
a dense numpy restatement of the chain the reference builds there, independent of the HIP library and of the C oracle --

  * QPTDualize with NO kernel supplied => the reference computes one and takes the LEFT generalised inverse K^+ = K^- P_R without regularisation (qptransform.c:997-1062);
    K^- = the solve of a factorisation with null-pivot detection: the null-pivot dof (an end dof of every subdomain of this 1-D bar) carries 0;
  * QPTOrthonormalizeEq on G = R'B' (gs: classical Gram-Schmidt of the rows; implicit: T = L^{-1} of G G' = L L'), QPTHomogenizeEq (lambda~ = G'(GG')^{-1} e, b_bar = d - F lambda~);
  * QPSSolve_SMALXE (smalxe.c:893-997) with QPConverged_Inner_SMALXE (:610-692) on an inner CG (QPSSetDefaultType: no box => QPSKSP), rho = 10 lambda_max, M1 = 100 lambda_max,
    eta = 0.1 ||b||, rho doubled in state 3 (smalxe.c:373-398,439-488).

Reproduced: 16 outer iterations and the golden's KKT numbers (1.44e-07, 3.18e-08, 1.11e-08, 1.72e-09 ...).  With K_reg^{-1} (MatRegularize) instead the same solve takes 11 outer iterations --
this is how the reference's choice of K^+ was identified."""
import numpy as np
import pytest


def _problem():
    ns, ne = 4, 7
    nl, ng = ne + 1, ns * ne + 1
    N, h = ns * nl, 1.0 / (ns * ne)
    Kb = np.zeros((nl, nl))
    for i in range(ne):
        Kb[i, i] += 1
        Kb[i + 1, i + 1] += 1
        Kb[i, i + 1] -= 1
        Kb[i + 1, i] -= 1
    b, l2g = np.zeros(ng), np.zeros(N, int)
    for r in range(ns):
        for i in range(nl):
            l2g[r * nl + i] = r * ne + i
        for i in range(ne):
            v = np.sin((r * ne + i + .5) * h * 3.14159) * .5 * h * h  # ex1.c's load (pi as 3.14159 there)
            b[r * ne + i] += v
            b[r * ne + i + 1] += v
    mult = np.bincount(l2g, minlength=ng)
    f = b[l2g] / mult[l2g]  # QPTMatISToBlockDiag: the assembled load divided among the copies
    rows = []
    for d in (0, N - 1):  # Dirichlet rows (KSPFETISetDirichlet, enforced by B)
        e = np.zeros(N)
        e[d] = 1
        rows.append(e)
    for r in range(ns - 1):  # gluing rows, -SCALE_ON: +-1/sqrt(2)
        e = np.zeros(N)
        e[r * nl + nl - 1] = 1 / np.sqrt(2)
        e[(r + 1) * nl] = -1 / np.sqrt(2)
        rows.append(e)
    return ns, nl, N, Kb, np.array(rows), f


def _orth(G0, e0, kind):
    if kind == "gs":
        G, T = G0.copy(), np.eye(G0.shape[0])
        for i in range(G.shape[0]):
            for j in range(i):
                c = G[j] @ G[i]
                G[i] -= c * G[j]
                T[i] -= c * T[j]
            nrm = np.linalg.norm(G[i])
            G[i] /= nrm
            T[i] /= nrm
        return G, T @ e0
    T = np.linalg.inv(np.linalg.cholesky(G0 @ G0.T))
    return T @ G0, T @ e0


def _lambda_max(A, tol=1e-4, its=50):
    """MatGetMaxEigenvalue (permonmatutils.c:442-522): power method from v = 1."""
    v = np.ones(A.shape[0])
    lam = 0.0
    for _ in range(its):
        w = A @ v
        lam_new = (v @ w) / (v @ v)
        v = w / np.linalg.norm(w)
        if abs(lam_new - lam) <= tol * abs(lam_new):
            lam = lam_new
            break
        lam = lam_new
    return lam


def _smalxe(A, b, G, rtol=1e-5):
    n = A.shape[0]
    maxeig = _lambda_max(A)
    M1, rho, eta = 100.0 * maxeig, 10.0 * maxeig, 0.1 * np.linalg.norm(b)
    GtG = G.T @ G
    u, Btmu, Lag_old = np.zeros(n), np.zeros(n), 0.0
    gtol = ttol = rtol * np.linalg.norm(b)
    st = dict(M1=M1, out=0, state=1, rnorm=0.0, normBu=0.0)

    def test(i, gn, uu):  # QPSConverged_Inner_SMALXE
        nb = np.linalg.norm(G @ uu)
        st["normBu"], st["rnorm"] = nb, max(nb, gn)
        if st["rnorm"] <= ttol:
            st["out"] = 2
            return 5
        if gn < min(st["M1"] * nb, eta):
            return 3
        if st["state"] == 3 and i < 1:
            return 0
        if gn <= gtol and not gn > nb:
            st["state"] = 3
            return 2
        return 0

    outer = 0
    for outer in range(100):
        Btmu = Btmu + rho * (GtG @ u)
        if st["out"]:
            break
        b_in, Ar = b - Btmu, A + rho * GtG
        r = b_in - Ar @ u
        p, rr, i = r.copy(), r @ r, 0
        reason = test(0, np.sqrt(rr), u)
        while reason == 0:  # KSPCG, unpreconditioned norm, nonzero initial guess
            Ap = Ar @ p
            a = rr / (p @ Ap)
            u, r = u + a * p, r - a * Ap
            rn, i = r @ r, i + 1
            reason = test(i, np.sqrt(rn), u)
            p, rr = r + (rn / rr) * p, rn
        nb = np.linalg.norm(G @ u)
        Lag = -u @ (b_in - 0.5 * Ar @ u)
        if Lag - (Lag_old + 0.5 * rho * nb * nb) < 0 and reason == 3:
            st["M1"] /= 2.0
        if not np.sqrt(rr) > nb and st["state"] == 3:
            rho *= 2.0
        Lag_old = Lag
    return outer, u, Btmu, b_in, Ar, st


@pytest.mark.parametrize("kind", ["gs", "implicit"])
def test_ex1_smalxe_orth_golden_from_the_left_generalised_inverse(goldens, kind):
    ns, nl, N, Kb, B, f = _problem()
    Rb = np.ones((1, nl)) / np.sqrt(nl)
    PR = np.eye(nl) - Rb.T @ Rb
    keep = list(range(1, nl))  # null pivot = dof 0 of every subdomain
    Km = np.zeros((nl, nl))
    Km[np.ix_(keep, keep)] = np.linalg.inv(Kb[np.ix_(keep, keep)])
    Kplus = np.kron(np.eye(ns), Km @ PR)  # K^- P_R
    R = np.kron(np.eye(ns), Rb.T)
    F, d = B @ Kplus @ B.T, B @ Kplus @ f
    G0, e0 = (B @ R).T, R.T @ f
    assert abs(np.linalg.norm(d) - 4.85e-3) < 0.02e-3  # the golden's ||d|| (1.44e-07 / 2.97e-05): K_reg^{-1} or the Moore-Penrose form give other values
    G, e = _orth(G0, e0, kind)
    lt = G.T @ np.linalg.solve(G @ G.T, e)
    bbar = d - F @ lt
    outer, x, Btmu, b_in, Ar, st = _smalxe(F, bbar, G)
    lam = x + lt
    gold = goldens["feti_ex1_smalxe_orth_" + kind]
    assert gold["text"][-1].strip() == "PERMON FETI CONVERGED_RTOL in 16 iteration" and outer == 16

    def close(a, ref):  # the golden prints 3 significant digits
        return abs(a - float(ref)) <= 0.012 * float(ref)

    kkt = gold["kkt"]
    assert close(np.linalg.norm(Ar @ x - b_in), kkt[0]["r"])                      # penalised QP: 1.44e-07
    assert close(np.linalg.norm(F @ x - bbar + Btmu), kkt[1]["r"]) and close(np.linalg.norm(G @ x), kkt[2]["r"])  # homogenised: 1.44e-07, ||BE x|| 3.18e-08
    assert close(np.linalg.norm(F @ lam - d + Btmu) / np.linalg.norm(d), kkt[3]["r_rel"])  # orthonormalised: 2.97e-05 of ||d||
    assert close(np.linalg.norm(G0 @ lam - e0), "1.11e-08")                         # dual QP: ||G lambda - e||
    r3 = F @ lam - d
    alpha = -np.linalg.solve(G0 @ G0.T, G0 @ r3)
    u = Kplus @ (f - B.T @ lam) - R @ alpha
    Kfull = np.kron(np.eye(ns), Kb)
    assert close(np.linalg.norm(Kfull @ u - f + B.T @ lam), "1.11e-08") and close(np.linalg.norm(B @ u), "1.72e-09")  # decomposed primal QP


def test_ex1_smalxe_orth_with_the_regularised_inverse_takes_11(goldens):
    """The control: the same chain on K_reg^{-1} = (K + rho^2 e_0 e_0')^{-1} (MatRegularize with one fixing dof) needs 11 outer iterations, not the golden's 16."""
    ns, nl, N, Kb, B, f = _problem()
    Rb = np.ones((1, nl)) / np.sqrt(nl)
    Kr = Kb.copy()
    Kr[0, 0] += np.linalg.eigvalsh(Kb)[-1] ** 2  # rho = lambda_max(K) (MatRegularize's scale; 11 iterations for any rho from 1 to 4, 12 at 10)
    Kplus = np.kron(np.eye(ns), np.linalg.inv(Kr))
    R = np.kron(np.eye(ns), Rb.T)
    F, d = B @ Kplus @ B.T, B @ Kplus @ f
    G, e = _orth((B @ R).T, R.T @ f, "gs")
    lt = G.T @ np.linalg.solve(G @ G.T, e)
    outer, *_ = _smalxe(F, d - F @ lt, G)
    assert outer == 11 and abs(np.linalg.norm(d) - 4.85e-3) > 0.1e-3
