"""examples/ex1.c -- a plain C program over the C ABI (no Python, no PETSc) that follows the reference's first tutorial -- run with
the command lines of the reference's TEST blocks (src/tutorials/ex1.c:165-184); its stdout must equal the reference's golden
output files line for line (what PETSc's test harness diffs: -qps_view_convergence and -qp_chain_view_kkt text)."""
import os
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


@pytest.mark.gpu
@pytest.mark.parametrize("args,case", [("-ns 4 -ne 7 -qp_chain_view_kkt -qpt_matis_to_diag_norm", "feti_ex1_1"), ("-ns 4 -ne 7 -qp_chain_view_kkt -qpt_matis_to_diag_norm -dir_in_hess", "feti_ex1_2")])
def test_feti_example_prints_the_golden_solver_line(goldens, args, case):
    """examples/feti_ex1.c with the TEST-block arguments of src/tutorials/feti/ex1.c (4 'ranks'): the solver line of the golden."""
    _build()
    out = subprocess.run([os.path.join(ROOT, "examples", "feti_ex1")] + args.split(), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    want = [ln for ln in goldens[case]["text"] if "PERMON FETI" in ln]
    assert want == ["PERMON FETI CONVERGED_RTOL in 1 iteration"] and out.stdout.splitlines() == want


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(CASES))
def test_example_output_equals_the_reference_golden_file(goldens, case):
    _build()
    out = subprocess.run([EXE] + CASES[case].split(), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    got = [ln.rstrip() for ln in out.stdout.splitlines() if ln.strip()]
    exp = [ln.rstrip() for ln in goldens[case]["text"] if ln.strip()]
    assert got == exp
