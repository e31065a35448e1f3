"""Host-side mirror of PERMON's QP / QPS interface for the hot path (same names, argument meaning and
error behaviour as include/permonqp.h, include/permonqps.h), over the C ABI of libpermonhip.

    qp  = QP(ctx);  qp.SetOperator(A); qp.SetRhs(b); qp.SetInitialVector(x); qp.SetBox(None, lb, ub)
    qps = QPS(ctx); qps.SetQP(qp); qps.SetType("mpgp"); qps.SetTolerances(rtol=1e-6); qps.Solve()

reads like src/tutorials/ex1.c:110-146.  Solver failure is not an exception: it is reason < 0 (qps.c:551).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check
from .core import Op, Vec, _ptr

QPSMPGPExpansionTypes = ("std", "projcg", "gf", "g", "gfgr", "ggr")  # mpgp.c:3
QPSMPGPExpansionLengthTypes = ("fixed", "opt", "optapprox", "bb")  # mpgp.c:4

KSP_CONVERGED_RTOL, KSP_CONVERGED_ATOL, KSP_CONVERGED_ITS, KSP_CONVERGED_HAPPY_BREAKDOWN = 2, 3, 4, 7
KSP_DIVERGED_ITS, KSP_DIVERGED_DTOL, KSP_DIVERGED_BREAKDOWN, KSP_DIVERGED_NANORINF = -3, -4, -5, -9


class QP:
    """The slice of the QP object the solvers read (qps->solQP: A, b, x, qpc/box, BE via pf, pc)."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.A = self.b = self.x = self.lb = self.ub = None
        self.pf = None  # QPPF holding BE = G (cE must be homogenised away: QPTHomogenizeEq)
        self.pc = None
        self._keep = []

    def SetOperator(self, A):  # QPSetOperator
        self.A = A

    def SetRhs(self, b):  # QPSetRhs
        self.b = b

    def SetInitialVector(self, x):  # QPSetInitialVector (qp.c:1964): x also receives the solution
        self.x = x

    def SetBox(self, is_, lb, ub):  # QPSetBox (qp.c:1842)
        """is_: None or an int array selecting the constrained sub-vector (lb/ub then have len(is_))."""
        n = self.A.n
        if is_ is not None:
            is_ = np.ascontiguousarray(is_, dtype=np.int32)
            full = []
            for bound, fill in ((lb, -np.inf), (ub, np.inf)):
                if bound is None:
                    full.append(None)
                    continue
                v = Vec(self.ctx, n, zero=False)
                check(self.ctx.L.pmh_qpc_box_expand_is(self.ctx.h, n, is_.size, is_.ctypes.data_as(C.c_void_p), bound.p, fill, v.p))
                full.append(v)
            lb, ub = full
        self.lb, self.ub = lb, ub

    def SetEq(self, pf):  # QPSetEq + QPSetQPPF: BE = G lives in the projector factory
        self.pf = pf

    def GetSolutionVector(self):
        return self.x


class QPS:
    """QPS front end: SetType("mpgp" | "smalxe" | "pcpg" | "ksp"), tolerances, Solve, statistics."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.L = ctx.L
        self.type = None
        self.qp = None
        self.rtol, self.atol, self.divtol, self.max_it = 1e-5, 1e-50, 1e4, 10000  # qps.c:73-76
        self._max_it_set = False
        self.mpgp_opts = _lib.MpgpOpts()
        check(self.L.pmh_mpgp_default_opts(C.byref(self.mpgp_opts)))
        self.smalxe_opts = None
        self.view_convergence = False
        self.h = None
        self.stats = None
        self._cb = None

    # ---- set-up -------------------------------------------------------------------------------------
    def SetQP(self, qp):
        self.qp = qp

    def SetType(self, t):
        if t not in ("mpgp", "smalxe", "pcpg", "ksp"):
            raise ValueError("unknown QPS type %r" % t)  # QPSSetType: PETSC_ERR_ARG_UNKNOWN_TYPE
        self.type = t
        if t == "smalxe" and self.smalxe_opts is None:
            self.smalxe_opts = _lib.SmalxeOpts()
            check(self.L.pmh_smalxe_default_opts(C.byref(self.smalxe_opts)))
            if not self._max_it_set:
                self.max_it = self.smalxe_opts.max_it  # 100, smalxe.c:1203

    def SetDefaultType(self):  # QPSSetDefaultType qps.c:422-455
        if self.qp.pf is not None:
            self.SetType("smalxe")
        elif self.qp.lb is not None or self.qp.ub is not None:
            self.SetType("mpgp")
        else:
            self.SetType("ksp")  # QPSKSP with KSPCG (qps.c:448, qpsksp.c:244)

    def SetTolerances(self, rtol=None, atol=None, divtol=None, max_it=None):  # QPSSetTolerances
        if rtol is not None:
            self.rtol = float(rtol)
        if atol is not None:
            self.atol = float(atol)
        if divtol is not None:
            self.divtol = float(divtol)
        if max_it is not None:
            self.max_it = int(max_it)
            self._max_it_set = True

    def SetFromOptions(self, options="", prefix=""):
        """QPSSetFromOptions (qps.c:860-900) over a PETSc-style option string, e.g. the args of the reference's TEST blocks:
        "-qps_type mpgp -qps_rtol 1e-6 -qps_mpgp_expansion_type gf -qps_mpgp_expansion_length_type opt".  Parsed by
        pmh_qps_set_from_options (C++, csrc/options.hip) with the reference's key names and argument checks; SMALXE's inner
        MPGP reads <prefix>smalxe_qps_*.  Returns the list of keys nobody consumed (PETSc's -options_left)."""
        q = _lib.QpsOpts()
        check(self.L.pmh_qps_default_opts(C.byref(q)))
        q.rtol, q.atol, q.divtol, q.max_it = self.rtol, self.atol, self.divtol, self.max_it
        sm = self.smalxe_opts
        if sm is None:
            sm = _lib.SmalxeOpts()
            check(self.L.pmh_smalxe_default_opts(C.byref(sm)))
        left = C.create_string_buffer(4096)
        check(self.L.pmh_qps_set_from_options(options.encode(), prefix.encode(), C.byref(q), C.byref(self.mpgp_opts), C.byref(sm), left, len(left)))
        t = q.type.decode()
        if t:
            self.smalxe_opts = sm if t == "smalxe" else self.smalxe_opts
            self.SetType(t)
        elif self.type is None and self.qp is not None:  # QPSSetDefaultTypeIfNotSpecified
            if self.qp.pf is not None:
                self.smalxe_opts = sm
            self.SetDefaultType()
        elif self.type == "smalxe":
            self.smalxe_opts = sm
        self.SetTolerances(rtol=q.rtol, atol=q.atol, divtol=q.divtol, max_it=q.max_it if q.max_it_set else None)
        if q.monitor:
            self.MonitorSet(True)
        self.view_convergence = bool(q.view_convergence)
        return [k for k in left.value.decode().split() if k]

    # MPGP options (QPSSetFromOptions_MPGP keys, mpgp.c:723-745)
    def MPGPSetAlpha(self, alpha, direct=False):
        self._mo().alpha_user, self._mo().alpha_direct = float(alpha), int(bool(direct))

    def MPGPSetGamma(self, gamma):
        self._mo().gamma = float(gamma)

    def MPGPSetOperatorMaxEigenvalue(self, maxeig):
        if not (maxeig >= 0 or maxeig == -1.0):
            raise ValueError("Argument must be nonnegative")  # mpgp.c:995
        self._mo().maxeig = float(maxeig)

    def MPGPSetOperatorMaxEigenvalueTolerance(self, tol):
        self._mo().maxeig_tol = float(tol)

    def MPGPSetOperatorMaxEigenvalueIterations(self, numit):
        if not numit > 1:
            raise ValueError("Argument must be > 1")  # mpgp.c:1088
        self._mo().maxeig_iter = int(numit)

    def MPGPSetExpansionType(self, exptype, lengthtype="fixed"):
        self._mo().exptype = QPSMPGPExpansionTypes.index(exptype)
        self._mo().explengthtype = QPSMPGPExpansionLengthTypes.index(lengthtype)

    def MPGPSetFallback(self, fallback=False, fallback2=False):
        self._mo().fallback, self._mo().fallback2 = int(fallback), int(fallback2)

    def MPGPSetUnfused(self, flag=True):
        self._mo().unfused = int(flag)

    def MPGPSetDistributed(self, flag=True):
        """x, b, lb, ub are row-distributed over the communicator's ranks (PETSc MPI Vec layout)."""
        self._mo().distributed = int(flag)

    def SMALXESetReuseProducts(self, flag=True):
        """Extension (pmh_smalxe_set_reuse_products, off by default): A_rho u carried from the inner solve's last gradient -- two operator applications less
        per outer iteration than the reference's sequence.  Call after SetUp."""
        if self.type != "smalxe":
            raise ValueError("SMALXESetReuseProducts is for QPS of type smalxe")
        self.SetUp()
        check(self.L.pmh_smalxe_set_reuse_products(self.h, int(bool(flag))))

    def MonitorSet(self, flag=True):  # QPSMonitorSet(qps, QPSMonitorDefault, ...)
        self._mo().monitor = int(flag)

    def _mo(self):
        return self.smalxe_opts.inner if self.type == "smalxe" else self.mpgp_opts

    # ---- solve --------------------------------------------------------------------------------------
    def SetUp(self):  # QPSSetUp qps.c:198-221
        if self.h is not None:
            return
        qp = self.qp
        if self.type is None:
            self.SetDefaultType()
        h = C.c_void_p()
        if self.type == "mpgp":
            if qp.pf is not None:
                raise ValueError("QPS solver mpgp is not compatible with its attached QP")  # QPSIsQPCompatible_MPGP
            o = self.mpgp_opts
            o.rtol, o.atol, o.divtol, o.max_it = self.rtol, self.atol, self.divtol, self.max_it
            check(self.L.pmh_mpgp_create(self.ctx.h, qp.A.h, qp.b.p, qp.x.p, _ptr(qp.lb), _ptr(qp.ub), C.byref(o), C.byref(h)))
        elif self.type == "smalxe":
            o = self.smalxe_opts
            o.rtol, o.atol, o.divtol, o.max_it = self.rtol, self.atol, self.divtol, self.max_it
            check(self.L.pmh_smalxe_create(self.ctx.h, qp.A.h, qp.b.p, qp.x.p, _ptr(qp.lb), _ptr(qp.ub), qp.pf.h, C.byref(o), C.byref(h)))
        self.h = h

    def Solve(self):  # QPSSolve qps.c:537-555
        self.SetUp()
        qp = self.qp
        if self.type == "mpgp":
            check(self.L.pmh_mpgp_solve(self.h))
            st = _lib.MpgpStats()
            check(self.L.pmh_mpgp_get_stats(self.h, C.byref(st)))
        elif self.type == "smalxe":
            check(self.L.pmh_smalxe_solve(self.h))
            st = _lib.SmalxeStats()
            check(self.L.pmh_smalxe_get_stats(self.h, C.byref(st)))
        elif self.type == "ksp":
            st = _lib.PcpgStats()
            check(self.L.pmh_ksp_cg_solve(self.ctx.h, qp.A.h, qp.b.p, qp.x.p, qp.pc.h if qp.pc is not None else None,
                                          self.rtol, self.atol, self.divtol, self.max_it, C.byref(st)))
        else:
            st = _lib.PcpgStats()
            check(self.L.pmh_pcpg_solve(self.ctx.h, qp.A.h, qp.b.p, qp.x.p, qp.pf.h, qp.pc.h if qp.pc is not None else None,
                                        self.rtol, self.atol, self.divtol, self.max_it, C.byref(st)))
        self.stats = st
        return st

    def _mpgp_handle(self):
        if self.type == "smalxe":  # QPSSMALXEGetInnerQPS
            h = C.c_void_p()
            check(self.L.pmh_smalxe_get_inner(self.h, C.byref(h)))
            return h
        return self.h

    def RunFixed(self, iters):
        """Throughput mode (bench.py): exactly `iters` MPGP iterations (for SMALXE: of its inner MPGP on the
        penalised operator, injected convergence test still evaluated), verdict of the test ignored."""
        self.SetUp()
        h = self._mpgp_handle()
        check(self.L.pmh_mpgp_run_fixed(h, int(iters)))
        st = _lib.MpgpStats()
        check(self.L.pmh_mpgp_get_stats(h, C.byref(st)))
        self.stats = st
        return st

    def RunFixedSolve(self, iters):
        """Throughput mode of a SMALXE solver (bench.py): the REAL solver loop -- outer multiplier / M1 / rho updates and the inner
        stopping rule included -- for exactly `iters` inner MPGP iterations in total; a solve that converges earlier restarts from
        the zero initial guess.  Returns the counts accumulated over the restarts."""
        if self.type != "smalxe":
            raise ValueError("RunFixedSolve is for QPS of type smalxe")
        self.SetUp()
        v = [C.c_int() for _ in range(6)]
        check(self.L.pmh_smalxe_run_fixed(self.h, int(iters), *[C.byref(x) for x in v]))
        keys = ("solves", "outer", "cg", "expansion", "proportioning", "hessian_mults")
        return dict(zip(keys, (x.value for x in v)))

    def ViewKKT(self):
        """The `r = ...` lines of -qp_chain_view_kkt for a box-constrained QP (QPViewKKT qp.c:245-369 + QPCViewKKT_Box
        qpcbox.c:333-427), formatted exactly as the reference prints them."""
        qp = self.qp
        out = (C.c_double * 8)()
        work = Vec(self.ctx, qp.A.n, zero=False)
        check(self.L.pmh_qp_kkt_box(qp.A.h, qp.b.p, qp.x.p, _ptr(qp.lb), _ptr(qp.ub), work.p, out))
        work.free()
        r, normb = list(out)[:7], out[7]
        name = "A*x - b" + (" - lambda_lb" if qp.lb is not None else "") + (" + lambda_ub" if qp.ub is not None else "")
        lines = ["r = ||%s|| = %.2e    rO/||b|| = %.2e" % (name, r[0], r[0] / normb)]
        if qp.lb is not None:
            lines.append("r = ||min(x-lb,0)||      = %.2e    r/||b|| = %.2e" % (r[1], r[1] / normb))
            lines.append("r = ||min(lambda_lb,0)|| = %.2e    r/||b|| = %.2e" % (r[2], r[2] / normb))
            lines.append("r = |lambda_lb'*(lb-x)|  = %.2e    r/||b|| = %.2e" % (r[3], r[3] / normb))
        if qp.ub is not None:
            lines.append("r = ||max(x-ub,0)||      = %.2e    r/||b|| = %.2e" % (r[4], r[4] / normb))
            lines.append("r = ||min(lambda_ub,0)|| = %.2e    r/||b|| = %.2e" % (r[5], r[5] / normb))
            lines.append("r = |lambda_ub'*(x-ub)|  = %.2e    r/||b|| = %.2e" % (r[6], r[6] / normb))
        return lines

    def ViewConvergence(self):
        """The -qps_view_convergence lines (QPSViewConvergence qps.c:968 + _MPGP mpgp.c:751-770 + _SMALXE smalxe.c:1001-1018)
        in the reference's wording, so that golden files can be compared as text."""
        names = {2: "CONVERGED_RTOL", 3: "CONVERGED_ATOL", 4: "CONVERGED_ITS", 7: "CONVERGED_HAPPY_BREAKDOWN", -3: "DIVERGED_ITS",
                 -4: "DIVERGED_DTOL", -5: "DIVERGED_BREAKDOWN", -9: "DIVERGED_NANORINF"}

        def head(st):
            return "last QPSSolve %s due to %s, KSPReason=%d, required %d iterations" % (
                "CONVERGED" if st.reason > 0 else "DIVERGED", names.get(st.reason, str(st.reason)), st.reason, st.iteration)

        def mpgp(st):
            return ["number of Hessian multiplications %d" % st.nmv, "number of CG steps %d" % st.ncg,
                    "number of expansion steps %d" % st.nexp, "number of proportioning steps %d" % st.nprop]

        st = self.stats
        if self.type == "mpgp":
            return [head(st)] + mpgp(st)
        if self.type == "smalxe":
            return [head(st), "Total number of inner iterations %d" % st.inner_iter_accu,
                    "#hits    of M1, eta: %3d, %3d" % (st.M1_hits, st.eta_hits), "#updates of M1, rho: %3d, %3d" % (st.M1_updates, st.rho_updates),
                    head(st.inner)] + mpgp(st.inner)
        return [head(st)]

    def GetIterationNumber(self):
        return self.stats.iteration

    def GetConvergedReason(self):
        return self.stats.reason

    def GetResidualNorm(self):
        return self.stats.rnorm

    def MPGPGetTrace(self):
        """The QPSMonitorDefault_MPGP lines (mpgp.c:21-34) as (steps, gp, gf, gc, alpha)."""
        cap = self.stats.iteration + 2 if self.type == "mpgp" else 1 << 20
        step = C.create_string_buffer(cap + 1)
        arrs = [np.zeros(cap) for _ in range(4)]
        ln = C.c_int()
        check(self.L.pmh_mpgp_get_trace(self.h, cap, step, *[a.ctypes.data_as(_lib.c_double_p) for a in arrs], C.byref(ln)))
        m = min(ln.value, cap)
        return (step.raw[:m].decode(),) + tuple(a[:m] for a in arrs)

    def MPGPGetWork(self, idx):
        p = C.c_void_p()
        check(self.L.pmh_mpgp_get_work(self.h, int(idx), C.byref(p)))
        a = np.empty(self.qp.A.n)
        check(self.L.pmh_memcpy_d2h(self.ctx.h, a.ctypes.data_as(C.c_void_p), p, 8 * a.size))
        return a

    def Destroy(self):
        if self.h is not None:
            if self.type == "mpgp":
                self.L.pmh_mpgp_destroy(self.h)
            elif self.type == "smalxe":
                self.L.pmh_smalxe_destroy(self.h)
            self.h = None
