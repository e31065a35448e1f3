"""permon_amd -- MI355X-native implementation of PERMON's QPS hot path (MPGP / SMALXE / PCPG over CSR SpMV,
box-constraint kernels and the FETI dual operator).  All computation happens in libpermonhip.so
(hand-written HIP for gfx950, include/permon_hip.h); this package is the host-side mirror of the
reference's QP / QPS / Mat interface for that path.  There is no CPU fallback."""
from . import problems  # noqa: F401
from ._lib import PermonHipError, load  # noqa: F401
from .core import Context, CsrMat, Op, Vec  # noqa: F401
from .qps import QP, QPS  # noqa: F401
from .mat import MG, MatBlockDiag, MatExplicitDual, csr_block_classes, MatCreateFetiDual, MatCreatePenalized, MatCreateProjected, MatCreateSVMDual, MatExtension, MatGluing, MatInv, MatRegularize, PCDualLumpedOp, QPPF  # noqa: F401,E402
from .feti import CubeFeti, DmdaFeti, box_mg_hierarchy, gluing_from_l2g, gluing_links  # noqa: F401,E402
from .chain import FetiDualQP, FETIContactSolve, KSPFETISolve, regularize_blocks  # noqa: F401,E402
