"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle and the reference goldens."""
import numpy as np
import pytest

import permon_amd as pa
from permon_amd import problems as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = pa.Context(0)
    yield c
    c.close()


def _solve(ctx, p, opts=None, unfused=False, monitor=False, is_=None, expansion=None):
    A = pa.CsrMat(ctx, p["n"], p["n"], p["rowptr"], p["col"], p["val"])
    op = pa.Op.from_csr(A)
    qp = pa.QP(ctx)
    qp.SetOperator(op)
    qp.SetRhs(ctx.vec_from(p["b"]))
    x = ctx.vec_from(p["x0"])
    qp.SetInitialVector(x)
    lb = ctx.vec_from(p["lb"]) if p.get("lb") is not None else None
    ub = ctx.vec_from(p["ub"]) if p.get("ub") is not None else None
    qp.SetBox(is_, lb, ub)
    qps = pa.QPS(ctx)
    qps.SetQP(qp)
    qps.SetType("mpgp")
    qps.SetTolerances(**(opts or {}))
    if expansion:
        qps.MPGPSetExpansionType(*expansion)
    qps.MPGPSetUnfused(unfused)
    qps.MonitorSet(monitor)
    st = qps.Solve()
    return qps, st, x.to_numpy()


def _counts(st):
    return (st.iteration, st.nmv, st.ncg, st.nexp, st.nprop)


def _gold(g):
    return (g["iterations"], g["nmv"], g["ncg"], g["nexp"], g["nprop"])


@pytest.mark.parametrize("unfused", [False, True])
def test_ex1_default_matches_reference_golden(ctx, goldens, oracle, unfused):
    g = goldens["ex1_1"]
    p = P.ex1(100)
    qps, st, x = _solve(ctx, p, unfused=unfused)
    assert st.reason == g["solves"][0]["reason"]
    assert _counts(st) == _gold(g["solves"][0])
    A = oracle.Csr(p["n"], p["n"], p["rowptr"], p["col"], p["val"])
    op = oracle.Op(p["n"], csr=A)
    ref = oracle.mpgp(op, p["b"], p["x0"], oracle.Box(p["n"], lb=p["lb"]))
    assert np.max(np.abs(x - ref["x"])) <= 1e-12
    # active set bookkeeping: bit-exact
    astol = 10 * np.finfo(float).eps
    assert np.array_equal(np.abs(x - p["lb"]) <= astol, np.abs(ref["x"] - p["lb"]) <= astol)
    r, normb = oracle.kkt_box(op, p["b"], x, p["lb"])
    for val, line in zip(r[2:], g["kkt"][2:]):
        assert "%.2e" % val == line["r"]


@pytest.mark.parametrize("case", ["ex1_opt", "ex1_optapprox", "ex1_bb", "ex1_projcg"])
def test_ex1_expansion_variants(ctx, goldens, case):
    g = goldens[case]
    o = g["args"]["opts"]
    p = P.ex1(100)
    qps, st, x = _solve(ctx, p, expansion=(o["exptype"], o.get("explengthtype", "fixed")))
    assert _counts(st) == _gold(g["solves"][0])


@pytest.mark.parametrize("infinite", [False, True])
def test_ex2_index_set_and_infinite_bounds(ctx, goldens, infinite):
    g = goldens["ex2_1_infinite-%s" % ("true" if infinite else "false")]
    p = P.ex2(100, infinite=infinite)
    qps, st, x = _solve(ctx, p, is_=p["is_"])
    assert _counts(st) == _gold(g["solves"][0])


@pytest.mark.parametrize("case,unfused", [("jbearing2_4", False), ("jbearing2_4", True), ("jbearing2_5", False), ("jbearing2_6", False)])
def test_jbearing_trace(ctx, goldens, case, unfused):
    """Both bounds (0 <= x <= 1000); per-iteration monitor lines against the reference golden:
    step types exact, norms to 1e-9 relative (reduction order differs from PETSc's)."""
    g = goldens[case]
    a = g["args"]
    p = P.jbearing2(a["mx"], a["my"])
    qps, st, x = _solve(ctx, p, opts=a["opts"], unfused=unfused, monitor=True)
    assert _counts(st) == _gold(g["solves"][0])
    steps, gp, gf, gc, alpha = qps.MPGPGetTrace()
    assert steps == "".join(t["step"] for t in g["trace"])
    for t in g["trace"]:
        k = t["it"]
        assert gp[k] == pytest.approx(float(t["gp"]), rel=1e-9)
        assert gf[k] == pytest.approx(float(t["gf"]), rel=1e-9, abs=1e-300)
        assert gc[k] == pytest.approx(float(t["gc"]), rel=1e-9, abs=1e-300)
        assert alpha[k] == pytest.approx(float(t["alpha"]), rel=1e-10)


def test_spmv_bit_exact_vs_oracle(ctx, oracle):
    """STREAM kernel sums each row left to right like MatMult_SeqAIJ => identical bits."""
    rng = np.random.default_rng(1)
    for nx, ny in ((37, 23), (200, 150)):
        rp, ci, va = P.laplace2d_csr(nx, ny)
        n = nx * ny
        va = va * rng.uniform(0.5, 1.5, va.size)
        x = rng.standard_normal(n)
        A = pa.CsrMat(ctx, n, n, rp, ci, va)
        xd, yd = ctx.vec_from(x), ctx.vec(n)
        A.mult(xd, yd)
        ref = oracle.spmv(oracle.Csr(n, n, rp, ci, va), x)
        assert np.array_equal(yd.to_numpy(), ref)
        A.mult_transpose(xd, yd)
        ref = oracle.spmv_transpose(oracle.Csr(n, n, rp, ci, va), x)
        assert np.array_equal(yd.to_numpy(), ref)


def test_spmv_long_rows_vector_kernel(ctx, oracle):
    import scipy.sparse as sp

    rng = np.random.default_rng(2)
    n = 5000
    M = sp.random(n, n, density=0.012, random_state=3, format="csr") + sp.eye(n, format="csr")
    M.sort_indices()
    x = rng.standard_normal(n)
    A = pa.CsrMat(ctx, n, n, M.indptr, M.indices, M.data)
    xd, yd = ctx.vec_from(x), ctx.vec(n)
    A.mult(xd, yd)
    ref = oracle.spmv(oracle.Csr.from_scipy(M), x)
    assert np.max(np.abs(yd.to_numpy() - ref)) <= 1e-13 * np.max(np.abs(ref))


def test_power_method(ctx, oracle):
    p = P.ex1(100)
    A = pa.CsrMat(ctx, p["n"], p["n"], p["rowptr"], p["col"], p["val"])
    lam, its = pa.Op.from_csr(A).max_eigenvalue()
    ref, rits = oracle.max_eigenvalue(oracle.Op(p["n"], csr=oracle.Csr(p["n"], p["n"], p["rowptr"], p["col"], p["val"])))
    assert its == rits and lam == pytest.approx(ref, rel=1e-13)


def test_qpc_kernels_match_oracle(ctx, oracle):
    rng = np.random.default_rng(5)
    n = 10007
    lb = rng.standard_normal(n) - 1.0
    ub = lb + rng.uniform(0.1, 2.0, n)
    lb[::7] = -np.inf
    ub[::5] = np.inf
    x = np.clip(rng.standard_normal(n), lb, ub)
    x[::3] = lb[::3]
    x[1::11] = ub[1::11]
    x = np.where(np.isfinite(x), x, 0.0)
    g, d = rng.standard_normal(n), rng.standard_normal(n)
    box = oracle.Box(n, lb=lb, ub=ub)
    L, h = ctx.L, ctx.h
    xd, gd, dd, lbd, ubd = (ctx.vec_from(a) for a in (x, g, d, lb, ub))
    gf, gc, gr, px = (ctx.vec(n) for _ in range(4))
    astol = 10 * np.finfo(float).eps
    pa._lib.check(L.pmh_qpc_box_grads(h, n, xd.p, gd.p, lbd.p, ubd.p, astol, gf.p, gc.p))
    rgf, rgc = box.grads(x, g)
    assert np.array_equal(gf.to_numpy(), rgf) and np.array_equal(gc.to_numpy(), rgc)
    pa._lib.check(L.pmh_qpc_box_gradreduced(h, n, xd.p, gf.p, lbd.p, ubd.p, 0.37, gr.p))
    assert np.array_equal(gr.to_numpy(), box.gradreduced(x, rgf, 0.37))
    import ctypes as C

    a = C.c_double()
    pa._lib.check(L.pmh_qpc_box_feas(h, n, xd.p, dd.p, lbd.p, ubd.p, C.byref(a)))
    assert a.value == box.feas(x, d)
    y = x + rng.standard_normal(n)
    yd = ctx.vec_from(y)
    pa._lib.check(L.pmh_qpc_box_project(h, n, yd.p, lbd.p, ubd.p, px.p))
    assert np.array_equal(px.to_numpy(), box.project(y))


def test_medium_laplace_fused_vs_oracle(ctx, oracle):
    """250 k unknowns, both bound kinds: converged solution and active set against the oracle."""
    p = P.laplace2d_box(500, 500, variant="twosided")
    qps, st, x = _solve(ctx, p, opts=dict(rtol=1e-8))
    assert st.reason == 2
    A = oracle.Csr(p["n"], p["n"], p["rowptr"], p["col"], p["val"])
    ref = oracle.mpgp(oracle.Op(p["n"], csr=A), p["b"], p["x0"], oracle.Box(p["n"], lb=p["lb"], ub=p["ub"]), rtol=1e-8)
    assert ref["reason"] == 2
    assert np.linalg.norm(x - ref["x"]) <= 1e-6 * np.linalg.norm(ref["x"])
    astol = 10 * np.finfo(float).eps
    act = lambda v: (np.abs(v - p["lb"]) <= astol).astype(int) - (np.abs(v - p["ub"]) <= astol).astype(int)
    assert np.mean(act(x) != act(ref["x"])) < 1e-3


@pytest.mark.parametrize("case", ["ex1_1", "ex1_opt", "ex1_optapprox", "ex1_bb", "ex1_projcg"])
def test_kkt_lines_match_reference_golden(ctx, goldens, case):
    """End to end through the product: solve, then the post-solve KKT report (-qp_chain_view_kkt) must print the
    numbers of the reference's golden file (src/tutorials/output/ex1_*.out)."""
    import re

    g = goldens[case]
    o = g["args"]["opts"]
    p = P.ex1(100)
    qps, st, x = _solve(ctx, p, expansion=(o["exptype"], o.get("explengthtype", "fixed")) if o else None)
    lines = qps.ViewKKT()
    assert len(lines) == len(g["kkt"]) == 4
    for line, ref in zip(lines, g["kkt"]):
        assert line.startswith("r = " + ref["name"])
        m = re.match(r"r = (.*?)\s*= (\S+)\s+rO?/\|\|b\|\| = (\S+)", line)
        if float(ref["r"]) < 1e-15:
            assert float(m.group(2)) < 1e-12  # 0.00e+00 in the golden: rounding-level here as well
        else:
            assert (m.group(2), m.group(3)) == (ref["r"], ref["r_rel"])


def test_kkt_two_sided_vs_numpy(ctx):
    p = P.jbearing2(10, 16)
    ub = np.full(p["n"], 0.05)
    p2 = dict(p, ub=ub)
    qps, st, x = _solve(ctx, p2, opts=dict(rtol=1e-8))
    import scipy.sparse as sp

    A = sp.csr_matrix((p["val"], p["col"], p["rowptr"]), shape=(p["n"], p["n"]))
    r = A @ x - p["b"]
    llb, lub = np.maximum(r, 0), np.maximum(-r, 0)
    normb = np.linalg.norm(p["b"])
    exp = [np.linalg.norm(r - llb + lub), np.linalg.norm(np.minimum(x - p["lb"], 0)), np.linalg.norm(np.minimum(llb, 0)), abs(llb @ (p["lb"] - x)),
           np.linalg.norm(np.maximum(x - ub, 0)), np.linalg.norm(np.minimum(lub, 0)), abs(lub @ (x - ub))]
    lines = qps.ViewKKT()
    assert len(lines) == 7 and "- lambda_lb + lambda_ub" in lines[0]
    for line, e in zip(lines, exp):
        got = float(line.split("=")[2].split()[0])
        assert got == pytest.approx(e, rel=1e-2, abs=1e-15)
    assert exp[3] / normb < 1e-6 and exp[6] / normb < 1e-6  # complementarity at the solution


def test_view_convergence_text_matches_golden(ctx):
    """-qps_view_convergence output, line by line against src/tutorials/output/ex1_1.out (filter: CONVERGED | number)."""
    qps, st, x = _solve(ctx, P.ex1(100))
    assert qps.ViewConvergence() == [
        "last QPSSolve CONVERGED due to CONVERGED_RTOL, KSPReason=2, required 181 iterations",
        "number of Hessian multiplications 200",
        "number of CG steps 156",
        "number of expansion steps 18",
        "number of proportioning steps 7",
    ]


# the args lines of the reference's own TEST blocks (src/tutorials/ex1.c:165-184), fed verbatim to QPSSetFromOptions
_EX1_TEST_BLOCKS = {
    "ex1_1": "-n 100 -qps_view_convergence -qp_chain_view_kkt",
    "ex1_opt": "-n 100 -qps_view_convergence -qp_chain_view_kkt -qps_mpgp_expansion_type gf -qps_mpgp_expansion_length_type opt",
    "ex1_optapprox": "-n 100 -qps_view_convergence -qp_chain_view_kkt -qps_mpgp_expansion_type g -qps_mpgp_expansion_length_type optapprox",
    "ex1_bb": "-n 100 -qps_view_convergence -qp_chain_view_kkt -qps_mpgp_expansion_type gfgr -qps_mpgp_expansion_length_type bb",
    "ex1_projcg": "-n 100 -qps_view_convergence -qp_chain_view_kkt -qps_mpgp_expansion_type projcg",
}


@pytest.mark.parametrize("case", sorted(_EX1_TEST_BLOCKS))
def test_ex1_goldens_from_the_reference_command_lines(ctx, goldens, case):
    """QPSSetFromOptions on the reference's command line -> the golden's -qps_view_convergence text and KKT lines."""
    g = goldens[case]
    p = P.ex1(100)
    A = pa.CsrMat(ctx, p["n"], p["n"], p["rowptr"], p["col"], p["val"])
    qp = pa.QP(ctx)
    qp.SetOperator(pa.Op.from_csr(A))
    qp.SetRhs(ctx.vec_from(p["b"]))
    qp.SetInitialVector(ctx.vec_from(p["x0"]))
    qp.SetBox(None, ctx.vec_from(p["lb"]), None)
    qps = pa.QPS(ctx)
    qps.SetQP(qp)
    left = qps.SetFromOptions(_EX1_TEST_BLOCKS[case])
    assert left == ["-n", "-qp_chain_view_kkt"]  # the example's own keys, not the solver's
    assert qps.type == "mpgp" and qps.view_convergence  # QPSSetDefaultTypeIfNotSpecified: box only -> MPGP (qps.c:445)
    st = qps.Solve()
    assert _counts(st) == _gold(g["solves"][0]) and st.reason == g["solves"][0]["reason"]
    text = qps.ViewConvergence()
    assert text[0].endswith("required %d iterations" % g["solves"][0]["iterations"]) and "CONVERGED_RTOL" in text[0]
    assert text[1:] == ["number of Hessian multiplications %d" % g["solves"][0]["nmv"], "number of CG steps %d" % g["solves"][0]["ncg"],
                        "number of expansion steps %d" % g["solves"][0]["nexp"], "number of proportioning steps %d" % g["solves"][0]["nprop"]]
    kkt = qps.ViewKKT()
    for line, gl in zip(kkt[1:], g["kkt"][1:]):  # r = ||min(x-lb,0)||, ||min(lambda_lb,0)||, |lambda_lb'(lb-x)|
        assert line.split("=")[-2].split()[0] == gl["r"]
