"""The PETSc-side glue TU (permon_amd/csrc/petsc_glue/permonhip_petsc.c, 2 000 lines) has never been through a compiler: PETSc is absent from this image and from the GPU box.
Round 5: `gcc -fsyntax-only` of that TU against PERMON's OWN headers (/root/reference/include: permonqps.h, permon/private/qpsimpl.h:12-24, qpcimpl.h:8-25, permonmatimpl.h,
qppfimpl.h ...) with a PETSc stand-in that declares types, macros and prototypes only (tests/stubs/petsc/petsc_stub.h, written from PETSc's manual pages; nothing is built,
linked or shipped; no reference source is compiled).  It catches typos, undeclared identifiers, wrong arity and argument types against PERMON's prototypes -- the first run
found one: MatMult_Timer is defined in libpermon (src/mat/impls/timer/mattimer.c:5) but declared in none of its headers.  It cannot check PETSc's real struct layouts, macro
expansions, linking or running.  Skipped where /root/reference does not exist (the GPU box)."""
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/include"
GLUE = os.path.join(ROOT, "permon_amd", "csrc", "petsc_glue", "permonhip_petsc.c")
FLAGS = ["gcc", "-fsyntax-only", "-fmax-errors=0", "-std=gnu11", "-Wall", "-Werror=implicit-function-declaration", "-Werror=incompatible-pointer-types", "-Werror=int-conversion",
         "-Werror=return-type", "-Werror=implicit-int", "-Wno-unused", "-Wno-comment", "-I", os.path.join(ROOT, "tests", "stubs", "petsc"), "-I", REF, "-I", os.path.join(ROOT, "include")]

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference's headers are not on this machine")


def test_glue_passes_syntax_and_arity_check_against_permon_headers():
    out = subprocess.run(FLAGS + [GLUE], capture_output=True, text=True, timeout=300)
    errors = [ln for ln in out.stderr.splitlines() if " error: " in ln]
    assert out.returncode == 0 and not errors, "\n".join(errors[:40])


@pytest.mark.parametrize("snippet,needle", [
    ("PetscErrorCode f(QP qp) { Vec lb; PetscCall(QPGetBox(qp, &lb)); return 0; }", "too few arguments"),                     # QPGetBox(QP, IS *, Vec *, Vec *): arity against PERMON's prototype
    ("PetscErrorCode f(QPS qps) { qps->ops->solve = (PetscErrorCode(*)(QP))0; return 0; }", "incompatible pointer"),           # _QPSOps.solve takes a QPS (qpsimpl.h:12-24)
    ("PetscErrorCode f(Mat A, Vec x) { PetscCall(MatMultHip(A, x, x)); return 0; }", "implicit declaration"),                   # an undeclared function
    ("PetscErrorCode f(QPC qpc, Vec x) { PetscReal a; PetscCall(QPCFeas(qpc, x, &a)); return 0; }", "too few arguments"),      # QPCFeas(QPC, Vec, Vec, PetscReal *)
])
def test_the_check_has_teeth(snippet, needle):
    """Negative controls: the same flags and headers reject a wrong arity against PERMON's own prototypes, a wrong op-table slot type and an undeclared function."""
    src = "#include <permon/private/qpsimpl.h>\n#include <permon/private/qpcimpl.h>\n#include <permonmat.h>\n#include <permon/private/qpimpl.h>\n" + snippet + "\n"
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "neg.c")
        open(c, "w").write(src)
        out = subprocess.run(FLAGS + [c], capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and needle in out.stderr, out.stderr[-1500:]


def test_stub_is_declarations_only():
    """The stand-in carries no function bodies (it is never compiled into anything) and says what it is."""
    import re

    txt = open(os.path.join(ROOT, "tests", "stubs", "petsc", "petsc_stub.h")).read()
    assert "STAND-IN" in txt.split("*/")[0]
    code = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    code = "\n".join(ln for ln in code.splitlines() if not ln.lstrip().startswith("#") and not ln.rstrip().endswith("\\"))
    assert not re.search(r"\)\s*\{[^}]*return", code), "a function body in the PETSc stand-in"
