"""Host-side logic of bench.py (no GPU): the self-launcher that `python bench.py --gpus N` runs for N > 1, and the compact line the driver parses."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_gpus_n_launches_n_ranks_itself():
    """--gpus N without a rendezvous in the environment: N children, RANK = LOCAL_RANK = 0..N-1, WORLD_SIZE = N, one shared 127.0.0.1 port, the launcher itself never
    imports torch or touches a GPU (--dry-launch: the children report their environment and exit)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "7", "--warmup", "2", "--dry-launch"], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_children"] == 4 and len(d["children"]) == 4
    ports = {k["master_port"] for k in d["children"]}
    assert len(ports) == 1 and int(next(iter(ports))) > 0
    for r, k in enumerate(d["children"]):
        assert k["rank"] == str(r) and k["local_rank"] == str(r) and k["world_size"] == "4" and k["master_addr"] == "127.0.0.1"
        assert k["ppid"] == d["launcher_pid"] and k["pid"] != d["launcher_pid"]
        assert k["argv"] == ["--gpus", "4", "--steps", "7", "--warmup", "2", "--dry-launch"]


def test_ranks_share_the_host_threads():
    """Eight ranks on one node must not start eight full thread pools: the launcher hands every child cores // N threads (OMP_NUM_THREADS = PMH_HOST_THREADS: numpy / OpenMP and
    the library's host-side builders), and a rank launched by torch.distributed.run arrives at the same share from LOCAL_WORLD_SIZE."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "OMP_NUM_THREADS", "PMH_HOST_THREADS")}
    cores = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-launch"], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["n_children"] == 8
    share = max(1, cores // 8)
    for k in d["children"]:
        assert k["host_threads"] == share and k["omp_num_threads"] == str(share) and k["pmh_host_threads"] == str(share), k
    assert 8 * share <= max(cores, 8)
    # the driver's launch: torch.distributed.run exports LOCAL_WORLD_SIZE (and OMP_NUM_THREADS=1 unless set); the rank derives its share itself
    env2 = dict(env, RANK="3", LOCAL_RANK="3", WORLD_SIZE="4", LOCAL_WORLD_SIZE="4", MASTER_ADDR="127.0.0.1", MASTER_PORT="29998")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-launch"], capture_output=True, text=True, timeout=60, env=env2)
    assert out.returncode == 0, out.stderr[-2000:]
    k = json.loads(out.stdout.strip().splitlines()[-1])
    assert k["host_threads"] == max(1, cores // 4) and k["pmh_host_threads"] == str(max(1, cores // 4))


def test_launcher_watchdog_ends_hung_ranks():
    """A rank that never finishes must not hang the launcher: PMH_BENCH_DEADLINE_S ends the children by PID and the launcher exits non-zero."""
    import time

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(PMH_BENCH_DEADLINE_S="2", PMH_BENCH_TEST_HANG="1")
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], capture_output=True, text=True, timeout=60, env=env)
    assert out.returncode != 0 and time.time() - t0 < 30
    assert "did not finish" in out.stderr


def test_launched_by_torchrun_env_is_not_relaunched():
    """With WORLD_SIZE in the environment (the driver's torch.distributed.run launch) bench.py is a rank, not a launcher."""
    env = dict(os.environ, RANK="1", LOCAL_RANK="1", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], capture_output=True, text=True, timeout=60, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert "children" not in d and d["rank"] == "1" and d["world_size"] == "2"


def test_failing_rank_fails_the_launcher():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "nonsense"], capture_output=True, text=True, timeout=60, env=env)
    assert out.returncode != 0


def test_compact_line_of_a_full_result_object():
    """The round-3 default run's full object (28 KB: the line the driver could not parse) through compact_line: < 4 KB, the contract's keys, numbers preserved."""
    import bench

    with open(os.path.join(ROOT, "profiles", "r03_bench_default_1gpu.json")) as fh:
        full = json.load(fh)
    assert len(json.dumps(full)) > 20000
    line = bench.compact_line(full, os.path.join(ROOT, "bench_details.json"))
    assert len(line) < 4000 and "\n" not in line
    c = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "details"):
        assert k in c, k
    assert abs(c["value"] - full["value"]) <= 1e-6 * full["value"] and c["steps"] == full["steps"] and c["n_gpus"] == 1
    assert set(c["config"]) >= {"workload", "parallelism", "rccl_ranks"} and len(c["config"]["workload"]) <= 300
    r = c["roofline"]
    assert r["bound"] == "mfma" and r["kernel"] == "k_fxo_gemm / k_fxo_gemm4<NA>" and r["peak"] == 78.6 and r["unit"] == "TFLOP/s" and r["launches_timed"] == full["roofline"]["launches_timed"]
    assert {"value", "unit", "cores", "kind", "extrapolated", "sample"} <= set(c["cpu_baseline"])
    for k in ("configs1", "configs3", "configs4", "general"):
        assert c[k]["value"] > 0 and 0 < c[k]["roofline_frac"]
    assert c["details"] == "bench_details.json"
    # a pathological object (every string 10x longer) still yields a line the driver can take
    fat = json.loads(json.dumps(full))
    fat["config"]["workload"] = fat["config"]["workload"] * 10
    for k in ("configs1", "configs3", "configs4", "general"):
        fat[k]["roofline"]["kernel"] = "k_x " * 400
    assert len(bench.compact_line(fat, "/tmp/x.json")) < 4000


def test_kernel_name_only():
    import bench

    assert bench.kernel_name_only("k_fxo_gemm / k_fxo_gemm4<NA> (row tile 128, or 8 NA): W_c is") == "k_fxo_gemm / k_fxo_gemm4<NA>"
    assert bench.kernel_name_only("k_fxo_gemm16<NI, NWM> (+ k_fxo_fin) (v_mfma_f64_16x16x4_f64; ...") == "k_fxo_gemm16<NI, NWM>"
    assert bench.kernel_name_only("k_bsr3<double>: the fp64 K x") == "k_bsr3<double>"
    assert bench.kernel_name_only("k_svm_x64_p1 + k_svm_x64_grad (paired passes") == "k_svm_x64_p1 + k_svm_x64_grad"
    assert bench.kernel_name_only("k_spmv_stream") == "k_spmv_stream"
