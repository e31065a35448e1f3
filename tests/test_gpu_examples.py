"""examples/ex1.c -- a plain C program over the C ABI (no Python, no PETSc) that follows the reference's first tutorial -- run with
the command lines of the reference's TEST blocks (src/tutorials/ex1.c:165-184); its stdout must equal the reference's golden
output files line for line (what PETSc's test harness diffs: -qps_view_convergence and -qp_chain_view_kkt text)."""
import os

import numpy as np
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "ex1")

CASES = {
    "ex1_1": "-n 100 -qps_view_convergence -qp_chain_view_kkt",
    "ex1_opt": "-n 100 -qps_view_convergence -qp_chain_view_kkt -qps_mpgp_expansion_type gf -qps_mpgp_expansion_length_type opt",
    "ex1_optapprox": "-n 100 -qps_view_convergence -qp_chain_view_kkt -qps_mpgp_expansion_type g -qps_mpgp_expansion_length_type optapprox",
    "ex1_bb": "-n 100 -qps_view_convergence -qp_chain_view_kkt -qps_mpgp_expansion_type gfgr -qps_mpgp_expansion_length_type bb",
    "ex1_projcg": "-n 100 -qps_view_convergence -qp_chain_view_kkt -qps_mpgp_expansion_type projcg",
}


def _build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "-s"])


def test_example_builds_as_plain_c():
    """C99, gcc, only include/permon_hip.h and -lpermonhip: the boundary is a C ABI, not a C++ or Python one."""
    _build()
    assert os.access(EXE, os.X_OK) and os.access(os.path.join(ROOT, "examples", "feti_ex1"), os.X_OK)


# (examples/feti_ex1.c against the WHOLE golden files feti/output/ex1_1.out, ex1_2.out: tests/test_gpu_feti_kkt_text.py)


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(CASES))
def test_example_output_equals_the_reference_golden_file(goldens, case):
    _build()
    out = subprocess.run([EXE] + CASES[case].split(), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    got = [ln.rstrip() for ln in out.stdout.splitlines() if ln.strip()]
    exp = [ln.rstrip() for ln in goldens[case]["text"] if ln.strip()]
    assert got == exp


@pytest.mark.gpu
@pytest.mark.parametrize("explicit,storage", [(1, 1), (1, 2), (1, 3), (1, 4), (0, 1)])
def test_contact_example_reproduces_the_python_chain(tmp_path, explicit, storage):
    """examples/contact_tfeti.c = pmh_feti_contact_solve from plain C (hierarchy built by pmh_mg_create_box, explicit dual operators,
    SMALXE + MPGP, rigid-body recovery, no Python in the solve) against the Python-orchestrated chain on the same problem: identical
    outer / inner / Hessian-multiplication / step-type counts, a feasible solution."""
    import re

    import permon_amd as pa
    from permon_amd import problems as P
    from permon_amd.chain import FetiDualQP

    _build()
    f = pa.CubeFeti((2, 2, 2), 8, contact=True)
    path = str(tmp_path / "contact.bin")
    P.write_contact_problem(path, f)
    out = subprocess.run([os.path.join(ROOT, "examples", "contact_tfeti"), path, str(explicit), "2", str(storage)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    ctx = pa.Context(0)
    G, e = f.coarse(orthonormalize=False)  # orthonormalised implicitly by the library, as pmh_feti_contact_solve does by default
    nn = f.nel + 1
    hier = pa.box_mg_hierarchy([f.Ki] * f.nsub, [(nn, nn, nn)] * f.nsub, 3, min_nodes=400)
    q = FetiDualQP(ctx, f.subset(range(f.nsub)), G, e, f.c, f.lb, orthonormal="implicit", kplus_rtol=1e-9, mg_hierarchy=hier, mg_precision="fp16", bsr3=True, explicit=dict(rtol=1e-12, storage={0: "full", 1: "sym", 2: "class", 3: "class_sym", 4: "class_orbit"}[storage], **({"symmetry": dict(dims=(9, 9, 9), ndof=3)} if storage == 4 else {})) if explicit else None)  # the same storage = the same rounding as the C run
    st = q.solve_smalxe(rtol=1e-5)
    want = q.qps.ViewConvergence()
    # the C program prints the same block (its first line without the reason's name)
    assert lines[0] == "last QPSSolve CONVERGED, KSPReason=%d, required %d iterations" % (st.reason, st.iteration)
    assert lines[1:4] == want[1:4]  # inner iterations, hits, updates
    assert lines[4:8] == want[5:9]  # Hessian multiplications, CG / expansion / proportioning steps
    m = re.search(r"\|\|G lambda - e\|\| = (\S+)\s+max\|B_E u - c_E\| / max\|u\| = (\S+)\s+max\(B_I u - c_I\) / max\|u\| = (\S+)\s+min lambda_I = (\S+)", lines[-1])
    gle, eqv, pen, lmin = (float(v) for v in m.groups())
    assert gle <= 1e-5 and eqv <= 1e-3 and pen <= 1e-3 and lmin >= -1e-12
    act = int(re.search(r"active contact rows (\d+)", lines[-2]).group(1))
    lam = q.dual_solution()
    assert act == int((lam[f.n_eq:] > 1e-8 * np.abs(lam).max()).sum())
    ctx.close()
