"""bench.py keeps the driver's contract: one JSON line with the agreed keys (small problem sizes here; the real sizes are the
defaults).  Also the CPU-baseline leg and the secondary configs[1] block on a reduced grid."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"}
ROOF = {"bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source"}


def _run(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines  # ONE JSON line on stdout
    return json.loads(lines[0])


def test_default_workload_line_small():
    d = _run("--nel", "7", "--steps", "30", "--warmup", "3", "--grid", "300", "--cpu-its", "3", "--cpu-its-feti", "3", "--general-nel", "5", "--cpu-direct-nel", "7", "--c2-steps", "100", "--svm-n", "200000")
    assert KEYS <= set(d) and ROOF <= set(d["roofline"])
    assert d["n_gpus"] == 1 and d["steps"] == 30 and d["warmup"] == 3 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "QPS iterations/s" and d["dtype"] == "f64" and d["data"] == "synthetic" and d["scaling"] == "strong"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] - 1e3) < 1e-6 * 1e3
    r = d["roofline"]
    # congruent cubes: the orbit storage (GEMM on the fp64 matrix instruction, compute-bound) is the default; the HBM-bound storages otherwise
    assert (r["bound"], r["unit"], r["peak"]) in (("hbm", "GB/s", 8000.0), ("mfma", "TFLOP/s", 78.6)) and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # the headline path: explicit local dual operators; the inner-Krylov path and its strict-fp64 variant ride along
    assert d["config"]["kplus"]["path"] == "explicit" and any(k in r["kernel"] for k in ("k_fx_symv", "k_fxs_gemm8", "k_fxs_symm8", "k_fxo_gemm")) and r["launches_timed"] > 0
    assert d["config"]["kplus"]["storage"] == "class_orbit" and r["bound"] == "mfma" and d["config"]["kplus"]["setup_symmetries"] == 48
    st = d["config"]["steps_by_type"]
    assert st["cg"] + st["expansion"] + st["proportioning"] == 30 and st["solves"] >= 1
    for k in ("iterative", "strict_fp64"):
        assert d[k]["value"] > 0 and ROOF <= set(d[k]["roofline"]) and "k_bsr3<double>" in d[k]["roofline"]["kernel"]
    assert d["config"]["rccl_ranks"] is None
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] == "port", d["cpu_baseline"]
    for cb in (d["cpu_baseline"], d["cpu_baseline_iterative"]):  # the reference's direct K^+ (sparse factorisation per block) and the host restatement of the iterative K^+
        assert {"value", "unit", "cores", "kind", "sample"} <= set(cb) and cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1, cb
    assert "splu" in d["cpu_baseline"]["sample"] and d["cpu_baseline"]["growth_exponent_solve"] > 0
    # what makes windows of different length comparable, at the top level
    assert d["applies_per_step"] > 1.0 and d["ms_per_operator_apply"] > 0 and d["time_to_solution_s"] > 0
    assert d["full_solve"]["reason"] > 0 and d["full_solve"]["inner_iterations"] > 0
    # the opt-in extension (A_rho u carried between the inner solve and the outer update) beside the reference's operation sequence: same iterations, fewer products
    rp = d["reuse_products"]
    assert rp["value"] > 0 and rp["rel_diff_lambda"] <= 1e-9 and rp["full_solve"]["inner_iterations"] == rp["reference_sequence"]["inner_iterations"], rp
    assert rp["full_solve"]["hessian_mults"] < rp["reference_sequence"]["hessian_mults"] and "EXTENSION" in rp["what"], rp
    c1 = d["configs1"]
    assert ROOF <= set(c1["roofline"]) and c1["value"] > 0 and c1["cpu_baseline"]["value"] > 0, c1
    assert "whole solve" in c1["timed"] and c1["steps_by_type"]["cg"] + c1["steps_by_type"]["expansion"] + c1["steps_by_type"]["proportioning"] == c1["steps"]
    # the secondary blocks of the driver-run line: general (non-congruent) decomposition, configs[3], configs[4], the one-call contact solve
    g = d["general"]
    assert g["value"] > 0 and g["kplus"]["storage"] == "sym" and g["roofline"]["bound"] == "hbm" and "k_fx_symv" in g["roofline"]["kernel"] and "HETEROGENEOUS" in g["workload"], g
    c3 = d["configs3"]
    assert c3["value"] > 0 and c3["coarse_problem"]["m"] == 384 and c3["coarse_problem"]["GGt_mfma_ms"] > 0 and c3["workload"].startswith("configs[3]"), c3
    c4 = d["configs4"]
    assert c4["value"] > 0 and c4["workload"].startswith("configs[4]") and ROOF - {"traffic_source"} <= set(c4["roofline"]), c4
    cs = d["contact_solve"]
    assert cs["time_to_solution_seconds"] > 0 and cs["outer"] >= 1 and cs["explicit_solves"] > 0, cs


def test_rehearsal_and_other_workloads_small():
    d = _run("--nel", "7", "--steps", "2", "--warmup", "1", "--sim-world", "4", "--no-cpu-baseline", "--no-c2")
    assert "REHEARSAL" in d["config"]["parallelism"] and "iterative" not in d
    d = _run("--nel", "7", "--steps", "4", "--warmup", "1", "--kplus", "iterative", "--no-cpu-baseline", "--no-c2")
    assert d["config"]["kplus"]["path"] == "iterative" and "strict_fp64" in d and "k_bsr3<double>" in d["roofline"]["kernel"]
    d = _run("--workload", "c2", "--grid", "400", "--steps", "20", "--warmup", "2", "--no-cpu-baseline")
    assert KEYS <= set(d) and d["scaling"] == "weak" and d["config"]["workload"].startswith("configs[1]")
    d = _run("--workload", "svm", "--svm-n", "200000", "--steps", "10", "--warmup", "2")
    assert KEYS <= set(d) and d["config"]["workload"].startswith("configs[4]")
    d = _run("--sub", "2,2,1", "--nel", "5", "--dense-coarse", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-c2")
    assert d["config"]["coarse_problem"]["m"] == 24 and d["config"]["coarse_problem"]["GGt_mfma_ms"] > 0


def test_forced_distributed_path_on_one_rank():
    """The N > 1 code path of bench.py on ONE rank (no 8-GPU node is available to the builder): torch.distributed rendezvous over
    RCCL, ncclUniqueId broadcast, pmh_comm_init, and -- with PMH_COMM_FORCE=1 -- every all-reduce of the data path really issued on
    the 1-rank communicator (B u in pmh_gluing_mult_transpose; the grouped scalar all-reduces and w of the SVM workload).  The line
    must carry rccl_ranks and the same checksum, bit for bit, as the local mode."""
    env = dict(os.environ, PMH_BENCH_FORCE_DIST="1", PMH_COMM_FORCE="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29531")

    def run(envv, *args):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True, timeout=900, env=envv)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][-1])

    for args in (("--nel", "7", "--steps", "40", "--warmup", "2", "--no-cpu-baseline", "--no-c2", "--no-iterative"),
                 ("--nel", "7", "--steps", "12", "--warmup", "2", "--no-cpu-baseline", "--no-c2", "--kplus", "iterative", "--no-iterative"),
                 ("--nel", "9", "--steps", "20", "--warmup", "2", "--no-cpu-baseline", "--no-c2", "--sim-world", "2"),  # striped operators + replica solver + the all-reduce
                 ("--workload", "svm", "--svm-n", "200000", "--steps", "10", "--warmup", "2")):
        # (the SVM operator pairs its passes over X on one GPU without a communicator only: the local run takes the separate passes the forced-communicator run takes)
        loc = run(dict(os.environ, PMH_SVM_NO_PAIRING="1"), *args)
        dst = run(env, *args)
        assert loc["config"]["rccl_ranks"] is None and dst["config"]["rccl_ranks"] == 1
        assert loc["config"]["checksum"] == dst["config"]["checksum"], (loc["config"]["checksum"], dst["config"]["checksum"])
        for k in ("cg", "expansion", "proportioning", "hessian_mults"):
            assert loc["config"]["steps_by_type"][k] == dst["config"]["steps_by_type"][k]
