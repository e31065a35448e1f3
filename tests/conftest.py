import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the ABI / host-logic tests load the in-tree extension: build it (hipcc cross-compiles without a GPU) if a fresh checkout
    # has not run __graft_entry__.build() yet.  Nothing falls back to a CPU path when the build fails: the tests then fail loudly.
    so = os.path.join(ROOT, "permon_amd", "libpermonhip.so")
    if not os.path.exists(so):
        import subprocess

        subprocess.call(["make", "-C", os.path.join(ROOT, "permon_amd", "csrc"), "-j8", "-s", "all"])


@pytest.fixture(scope="session")
def goldens():
    with open(os.path.join(ROOT, "tests", "golden", "reference_goldens.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O

    O.build()
    return O
