"""SURVEY 8(e) on a one-GPU box: two processes run the library's own distributed arithmetic (sample-sharded SVM MPGP with `distributed` = 1; subdomain-sharded FETI with
the B u all-reduce) with the collectives on the host transport (pmh_comm_set_host_transport over gloo) and must reproduce the single-rank run."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("case", ["svm", "svm separate", "feti_iterative", "feti_explicit"])
def test_world2_library_distributed_arithmetic(case):
    port = str(_free_port())
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_transport_worker.py")] + case.split(), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o[-4000:])
        assert "rank %d ok" % r in o, o[-2000:]
