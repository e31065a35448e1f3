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


def _run(*args, env=None):
    """Runs bench.py; returns (the compact line the driver parses, the full object of the details file)."""
    import tempfile
    import time

    det = os.path.join(tempfile.mkdtemp(prefix="pmh_bench_test_"), "details.json")
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--details", det] + list(args), capture_output=True, text=True, timeout=900, env=env)
    wall = time.time() - t0
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines  # ONE JSON line on stdout
    assert len(lines[-1]) < 4096, len(lines[-1])  # the driver could not take round 3's 28 KB line
    c = json.loads(lines[-1])
    assert json.loads(json.dumps(c)) == c
    assert c["ms_per_step"] * c["steps"] * 1e-3 < wall
    with open(det) as fh:
        d = json.load(fh)
    assert c["details"] == det
    # the compact line is a projection of the details object
    for k in ("metric", "unit", "n_gpus", "steps", "warmup", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert c[k] == d[k], k
    assert abs(c["value"] - d["value"]) <= 1e-5 * abs(d["value"])
    assert KEYS <= set(c) and {"bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches_timed"} <= set(c["roofline"])
    assert abs(c["roofline"]["frac"] - d["roofline"]["frac"]) <= 1e-3 and c["roofline"]["frac"] <= 1.0
    assert set(c["config"]) >= {"workload", "parallelism", "rccl_ranks"} and "model" not in c["config"] and len(c["config"]["workload"]) <= 300
    return c, d


def fd_n(d):
    import re

    return int(re.search(r"K_i (\d+) rows / (\d+) nnz", d["config"]["workload"]).group(1))


def fd_nnz(d):
    import re

    return int(re.search(r"K_i (\d+) rows / (\d+) nnz", d["config"]["workload"]).group(2))


def test_default_workload_line_small():
    c, d = _run("--nel", "7", "--steps", "30", "--warmup", "3", "--grid", "300", "--cpu-its", "3", "--cpu-its-feti", "3", "--general-nel", "5", "--cpu-direct-nel", "7", "--c2-steps", "100", "--svm-n", "200000")
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
    # what the matrix cores execute / launch time: the round-2/3 count (skipped k segments still in it) rides along under its own name
    assert r["flops_per_launch"] > 0 and r["frac_legacy_r02"] > 0 and abs(r["frac"] - r["flops_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e12 / 78.6) < 1e-9
    assert 0 < r["avg_launch_ms"] < r["dense_apply_ms"] and r["finishing_kernel_ms"] > 0 and r["frac_dense_apply"] < r["frac"]  # avg_launch_ms = the GEMM kernel alone
    for k in ("iterative", "strict_fp64"):
        # the fp32 / fp16 V-cycle: the 8 congruent blocks run as the 8 columns of one block (k_mv_spmv); strict fp64: the one-column kernel on 8 replicas of one device copy
        kern = d[k]["roofline"]["kernel"]
        assert d[k]["value"] > 0 and ROOF <= set(d[k]["roofline"]) and ("k_bsr3<double>" in kern or "k_mv_spmv<double" in kern)
        assert 0 < d[k]["roofline"]["frac"] <= 1.0 and d[k]["roofline"]["blocks_per_device_copy"] == 8
        assert "k_mv_spmv<double" in kern or d[k]["roofline"]["blockdiag_figure_GBs"] > d[k]["roofline"]["achieved"]
        assert c[k]["value"] > 0 and c[k]["roofline_frac"] <= 1.0
    assert "k_mv_spmv<double" in d["iterative"]["roofline"]["kernel"] and "k_bsr3<double>" in d["strict_fp64"]["roofline"]["kernel"]
    assert d["config"]["rccl_ranks"] is None
    # cpu_baseline = the REFERENCE's algorithm on the host: the sparse direct K^+ (splu per block, solve phase timed; at this test's block size it is factored in the run itself,
    # at the headline's 43^3 the committed one-time measurement is carried over); cpu_baseline_iterative = the host port of the GPU's inner-Krylov K^+, measured in the run
    cb = d["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cb) and cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and "splu" in cb["sample"], cb
    assert c["cpu_baseline"]["value"] > 0 and c["cpu_baseline"]["kind"] == "port"
    ci = d["cpu_baseline_iterative"]
    assert ci["value"] > 0 and ci["kind"] == "port" and "V-cycle" in ci["kplus"] and c["cpu_baseline_iterative"]["value"] > 0, ci
    # the FETI dual SpMV on HBM: distinct K_i, one device copy each, both kernels, fractions of the peak <= 1 and the two kernels agree
    fd = d["feti_dual_spmv"]
    assert fd["csr"]["device_copies"] == 8 and fd["bsr3"]["device_copies"] == 8 and 0 < fd["frac"] <= 1.0 and 0 < fd["bsr3"]["frac"] <= 1.0, fd
    assert fd["bsr3"]["max_rel_diff_vs_csr_kernel"] <= 1e-13 and abs(fd["csr"]["algorithmic_bytes_per_launch"] - (12.0 * 8 * fd_nnz(d) + 20.0 * 8 * fd_n(d))) < 1.0, fd
    assert c["feti_dual_spmv"]["frac"] <= 1.0 and c["feti_dual_spmv"]["kernel"] == "k_spmv_stream"
    kc = d["kplus_cg_product"]
    assert kc["blocks_per_copy"] == 8 and 0 < kc["frac"] <= 1.0 and kc["blockdiag_figure_GBs"] > kc["achieved"], kc
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
    # (round 4: one class per block, every class on the closure of its touched set under the cube's group -> orbit storage with all 48 operations, the GEMM as the dense apply)
    assert g["value"] > 0 and g["kplus"]["storage"] == "class_orbit" and g["kplus"]["setup_symmetries"] == 48 and g["roofline"]["bound"] == "mfma" and "k_fxo_gemm" in g["roofline"]["kernel"], g
    assert "HETEROGENEOUS" in g["workload"] and 0 < g["roofline"]["frac"] <= 1.0, g
    c3 = d["configs3"]
    assert c3["value"] > 0 and c3["coarse_problem"]["m"] == 384 and c3["coarse_problem"]["GGt_mfma_ms"] > 0 and c3["workload"].startswith("configs[3]"), c3
    c4 = d["configs4"]
    assert c4["value"] > 0 and c4["workload"].startswith("configs[4]") and ROOF - {"traffic_source"} <= set(c4["roofline"]), c4
    cs = d["contact_solve"]
    assert cs["time_to_solution_seconds"] > 0 and cs["outer"] >= 1 and cs["explicit_solves"] > 0, cs
    for k in ("general", "configs1", "configs3", "configs4"):
        assert c[k]["value"] > 0 and c[k]["roofline_frac"] is not None, (k, c[k])


def test_rehearsal_and_other_workloads_small():
    c, d = _run("--nel", "7", "--steps", "2", "--warmup", "1", "--sim-world", "4", "--no-cpu-baseline", "--no-c2")
    assert "REHEARSAL" in d["config"]["parallelism"] and "REHEARSAL" in c["config"]["parallelism"] and "iterative" not in d
    c, d = _run("--nel", "7", "--steps", "4", "--warmup", "1", "--kplus", "iterative", "--no-cpu-baseline", "--no-c2")
    assert d["config"]["kplus"]["path"] == "iterative" and "strict_fp64" in d and "k_mv_spmv<double" in d["roofline"]["kernel"] and c["roofline"]["kernel"].startswith("k_mv_spmv<double")
    c, d = _run("--workload", "c2", "--grid", "400", "--steps", "20", "--warmup", "2", "--no-cpu-baseline")
    assert d["scaling"] == "weak" and d["config"]["workload"].startswith("configs[1]") and c["config"]["workload"].startswith("configs[1]")
    c, d = _run("--workload", "svm", "--svm-n", "200000", "--steps", "10", "--warmup", "2")
    assert d["config"]["workload"].startswith("configs[4]")
    c, d = _run("--sub", "2,2,1", "--nel", "5", "--dense-coarse", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-c2")
    assert d["config"]["coarse_problem"]["m"] == 24 and d["config"]["coarse_problem"]["GGt_mfma_ms"] > 0


def test_forced_distributed_path_on_one_rank():
    """The N > 1 code path of bench.py on ONE rank (no 8-GPU node is available to the builder): torch.distributed rendezvous over
    RCCL, ncclUniqueId broadcast, pmh_comm_init, and -- with PMH_COMM_FORCE=1 -- every all-reduce of the data path really issued on
    the 1-rank communicator (B u in pmh_gluing_mult_transpose; the grouped scalar all-reduces and w of the SVM workload).  The line
    must carry rccl_ranks and the same checksum, bit for bit, as the local mode."""
    env = dict(os.environ, PMH_BENCH_FORCE_DIST="1", PMH_COMM_FORCE="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29531")

    def run(envv, *args):
        return _run(*args, env=envv)[1]

    for args in (("--nel", "7", "--steps", "40", "--warmup", "2", "--no-cpu-baseline", "--no-c2", "--no-iterative"),
                 ("--nel", "7", "--steps", "12", "--warmup", "2", "--no-cpu-baseline", "--no-c2", "--kplus", "iterative", "--no-iterative"),
                 ("--nel", "9", "--steps", "20", "--warmup", "2", "--no-cpu-baseline", "--no-c2", "--sim-world", "2"),  # striped operators + replica solver + the all-reduce
                 ("--workload", "svm", "--svm-n", "200000", "--steps", "10", "--warmup", "2")):
        # (round 4: the SVM operator pairs its passes over X with a communicator too -- w and the feasible step length are completed across the ranks between the passes)
        loc = run(dict(os.environ), *args)
        dst = run(env, *args)
        assert loc["config"]["rccl_ranks"] is None and dst["config"]["rccl_ranks"] == 1
        assert loc["config"]["checksum"] == dst["config"]["checksum"], (loc["config"]["checksum"], dst["config"]["checksum"])
        for k in ("cg", "expansion", "proportioning", "hessian_mults"):
            assert loc["config"]["steps_by_type"][k] == dst["config"]["steps_by_type"][k]


def test_two_ranks_on_one_gpu_host_transport():
    """`python bench.py --gpus 2` END TO END on the one-GPU box: the self-launcher starts two ranks, PMH_BENCH_TRANSPORT=host puts both on device 0 with gloo between them and the
    library's collectives on its host transport (RCCL cannot put two ranks on one device).  What this runs that nothing else does: bench.py's own N > 1 orchestration -- the ranks'
    shares of the operator (k-range split of the orbit GEMM; subdomain blocks for the inner-Krylov K^+; samples for the SVM), barriers, max over ranks, rank 0's line -- on top of the
    library's distributed arithmetic.  The two-rank run must take the steps the one-rank run takes and land on the same iterate (to the rounding of the split sums)."""
    env = dict(os.environ, PMH_BENCH_TRANSPORT="host")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    for args, key in ((("--nel", "7", "--steps", "40", "--warmup", "2", "--no-cpu-baseline", "--no-c2", "--no-iterative"), "norm_lambda_child_after_last_step"),
                      (("--nel", "7", "--young", "distinct", "--steps", "40", "--warmup", "2", "--no-cpu-baseline", "--no-c2", "--no-iterative"), "norm_lambda_child_after_last_step"),  # 8 materials: every rank the closed orbit classes of its own 4 blocks
                      (("--nel", "7", "--steps", "12", "--warmup", "2", "--no-cpu-baseline", "--no-c2", "--kplus", "iterative", "--no-iterative"), "norm_lambda_child_after_last_step"),
                      (("--workload", "svm", "--svm-n", "200000", "--steps", "10", "--warmup", "2"), "norm_x_after_last_step"),
                      # round 6: an IRREGULAR partition (8 staircase-bounded subdomains, 4 per rank: algebraic hierarchy, per-block explicit operators / the inner-Krylov K^+ on it) ...
                      (("--partition", "staircase", "--nel", "6", "--steps", "30", "--warmup", "2", "--no-cpu-baseline", "--no-c2", "--no-iterative"), "norm_lambda_child_after_last_step"),
                      (("--partition", "staircase", "--nel", "6", "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--no-c2", "--kplus", "iterative", "--no-iterative"), "norm_lambda_child_after_last_step"),
                      # ... and the explicit headline WITH its inner-Krylov pass: every N > 1 line carries `iterative` next to the headline
                      (("--nel", "7", "--steps", "12", "--warmup", "2", "--no-cpu-baseline", "--no-c2"), "norm_lambda_child_after_last_step")):
        one = _run(*args)[1]
        two = _run("--gpus", "2", *args, env=env)[1]
        if "--no-iterative" not in args and "--workload" not in args:
            assert two["iterative"]["value"] > 0 and "k_bsr3<double>" in two["iterative"]["roofline"]["kernel"], two.get("iterative")  # (4 blocks per rank: the one-column kernel)
        assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["config"]["rccl_ranks"] == 2 and "host" in two["config"]["transport"]
        a, b = float(one["config"]["checksum"][key]), float(two["config"]["checksum"][key])
        assert abs(a - b) <= 1e-9 * abs(a), (args, a, b)  # (the FETI paths reproduce the one-rank iterate bit for bit: replicated dual arithmetic, the split sums added in a fixed order)
        for k in ("cg", "expansion", "proportioning", "hessian_mults"):
            assert one["config"]["steps_by_type"][k] == two["config"]["steps_by_type"][k], (k, one["config"]["steps_by_type"], two["config"]["steps_by_type"])
        assert two["value"] > 0 and two["scaling"] == one["scaling"]


def test_four_ranks_on_one_gpu_host_transport():
    """The same with FOUR ranks (the driver's N = 4; N = 8 exceeds the six processes a box lets share its GPU): the k range of the orbit GEMM in four shares, configs[3]'s shape
    (64 subdomains, dense coarse problem) with 16 blocks per rank.  The split sums are added in rank order by the all-reduce: the iterate agrees with the one-rank run to rounding."""
    env = dict(os.environ, PMH_BENCH_TRANSPORT="host")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    for args in (("--nel", "9", "--steps", "40", "--warmup", "2", "--no-cpu-baseline", "--no-c2", "--no-iterative"),
                 ("--sub", "4,4,4", "--nel", "5", "--dense-coarse", "--steps", "30", "--warmup", "2", "--no-cpu-baseline", "--no-c2", "--no-iterative")):
        one = _run(*args)[1]
        four = _run("--gpus", "4", *args, env=env)[1]
        assert four["n_gpus"] == 4 and four["config"]["rccl_ranks"] == 4 and "host" in four["config"]["transport"]
        a, b = float(one["config"]["checksum"]["norm_lambda_child_after_last_step"]), float(four["config"]["checksum"]["norm_lambda_child_after_last_step"])
        assert abs(a - b) <= 1e-12 * abs(a), (args, a, b)
        for k in ("cg", "expansion", "proportioning", "hessian_mults", "outer"):
            assert one["config"]["steps_by_type"][k] == four["config"]["steps_by_type"][k]
