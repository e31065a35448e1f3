"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol that
include/permon_hip.h declares; the product fails loudly without a GPU; nothing in the product imports the oracle."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "permon_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pmh_[A-Za-z0-9_]+)\s*\(", txt)) - {"pmh_converged_fn", "pmh_shell_mult_fn"})


def test_build_and_exports():
    import __graft_entry__ as g

    g.build()
    lib = ctypes.CDLL(os.path.join(ROOT, "permon_amd", "libpermonhip.so"))
    syms = _header_symbols()
    assert len(syms) > 80
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    from permon_amd import _lib

    assert sorted(_lib.EXPORTED) == syms  # the ctypes binding covers the whole header, nothing more


def test_no_gpu_fails_loudly():
    """No CPU fallback: without a HIP device pmh_init must return an error, not compute on the host."""
    code = "import permon_amd as pa\ntry:\n    pa.Context(0)\n    print('GPU')\nexcept pa.PermonHipError as e:\n    print('ERR', e)\n"
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, env=dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1"))
    assert out.stdout.startswith("ERR"), out.stdout + out.stderr
    assert "no HIP device" in out.stdout or "no CPU fallback" in out.stdout


def test_product_does_not_touch_oracle():
    """Nothing shipped under permon_amd/ imports, links, loads or executes anything of oracle/."""
    pat = re.compile(r"(import\s+oracle|from\s+oracle|from\s+\.\.?oracle|oracle/|oracle\.py|liborc|\borc_[a-z]+\(|permon_oracle)")
    for dirpath, _, files in os.walk(os.path.join(ROOT, "permon_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".c")) or f == "Makefile":
                src = open(os.path.join(dirpath, f)).read()
                m = pat.search(src)
                assert not m, (dirpath, f, m.group(0))


def test_struct_layouts_match_header():
    """sizeof() of the ctypes mirrors equals the C structs (compiled with gcc from the public header)."""
    from permon_amd import _lib

    src = r'''
#include <stdio.h>
#include "permon_hip.h"
int main(void){ printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(pmh_mpgp_opts), sizeof(pmh_mpgp_stats), sizeof(pmh_smalxe_opts), sizeof(pmh_smalxe_stats), sizeof(pmh_pcpg_stats),
  sizeof(pmh_qps_opts), sizeof(pmh_kspfeti_opts), sizeof(pmh_kspfeti_stats)); return 0; }
'''
    import tempfile

    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = list(map(int, subprocess.check_output([exe]).split()))
    got = [ctypes.sizeof(t) for t in (_lib.MpgpOpts, _lib.MpgpStats, _lib.SmalxeOpts, _lib.SmalxeStats, _lib.PcpgStats, _lib.QpsOpts, _lib.KspFetiOpts, _lib.KspFetiStats)]
    assert got == sizes
