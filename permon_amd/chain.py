"""The QP transform chain of the (T)FETI path, host side, over device operators:
QPTDualize -> [QPTOrthonormalizeEq] -> QPTHomogenizeEq -> QPTEnforceEqByProjector -> QPS (SMALXE / PCPG).
Each step cites the reference function it follows (src/qp/interface/qptransform.c).  Only set-up happens
here (a handful of operator applications); the iteration itself is pmh_smalxe_solve / pmh_pcpg_solve."""
import ctypes as C

import numpy as np

from ._lib import check
from .core import Op, Vec
from .mat import QPPF, MatBlockDiag, MatCreateFetiDual, MatCreateProjected, MatExplicitDual, MatGluing, MatInv, MatRegularize, PCDualLumpedOp, csr_block_classes
from .qps import QP, QPS


def regularize_blocks(ctx, local, rho=None):
    """MatRegularize on a MATBLOCKDIAG (permonmatregularize.c:241-266 works on the rank's diagonal block; with several
    subdomains per GPU every block is treated as its own 'rank'): returns (blockdiag(K_reg,i) as scipy CSR, pivots per
    block, rho per block).  Congruent blocks (same K_i and R_i objects' values) are regularised once.
    rho: None = MatGetMaxEigenvalue(K_loc, NULL, &rho, 1, 20) per block as the reference does (:254; it stops after 2 or 3
    power iterations from a RAND48 restart depending on the SIGN of a rounding-noise Rayleigh quotient, so it is not
    reproducible to more than its order of magnitude), or a given positive value for every block."""
    import scipy.sparse as sp

    rs = np.asarray(local["block_rowstart"])
    K = local["K"].tocsr()
    R = np.asarray(local["R"]) if local.get("R") is not None else np.zeros((0, K.shape[0]))
    blocks, pivots, rhos, cache = [], [], [], []
    for b in range(len(rs) - 1):
        lo, hi = int(rs[b]), int(rs[b + 1])
        Kb = K[lo:hi, lo:hi].tocsr()
        Rb = R[:, lo:hi]
        Rb = Rb[np.any(Rb != 0.0, axis=1)]  # a block without kernel has zero columns in R (d_loc = 0)
        hit = None
        for (Kc, Rc, out) in cache:
            if Kc.shape == Kb.shape and Kc.nnz == Kb.nnz and Rc.shape == Rb.shape and np.array_equal(Kc.indices, Kb.indices) and np.array_equal(Kc.data, Kb.data) and np.array_equal(Rc, Rb):
                hit = out
                break
        if hit is None:
            hit = MatRegularize(ctx, Kb, Rb, rho=rho)
            cache.append((Kb, Rb, hit))
        blocks.append(hit[0]), pivots.append(hit[1]), rhos.append(hit[2])
    return sp.block_diag(blocks, format="csr"), pivots, rhos


class FetiDualQP:
    """Dual QP of a TFETI problem on this rank's subdomain blocks (lambda replicated on every rank)."""

    def __init__(self, ctx, local, G, e, c, lb, orthonormal=True, kplus_rtol=1e-10, kplus_max_it=20000, jacobi=True, mg_hierarchy=None, mg_degree=2, mg_precision="fp64", bsr3=False,
                 regularize=False, explicit=None, mg_box=None, mg_sa=None):
        """local: dict from CubeFeti.subset(); G, e: coarse matrix / rhs (global, replicated); c: constraint rhs;
        lb: dual lower bound (-inf on equality rows, 0 on inequality rows).
        regularize: the reference's default (-regularize 1, QPTDualize -> MatInvSetRegularizationType(MAT_REG_EXPLICIT),
        qptransform.c:1012): MATINV works on K_reg = MatRegularize(K, R) and K^+ = K_reg^{-1} without projections;
        False: -regularize 0 with the Moore-Penrose wrapping P_R K^- P_R (-qpt_dualize_Kplus_mp, qptransform.c:1020-1062).
        With regularize=True a multigrid hierarchy, if given, must have been built on the regularised blocks.
        explicit: None, or a dict -- F applies through the explicit local dual operators W_b = (K_b^+)[Gamma_b, Gamma_b]
        (pmh_fexplicit, the exact K^+ path): rtol (1e-12) of the set-up solves; min_slots: a rank with fewer blocks than this
        assembles with a replica solver built by solver_factory(nslots) -> MatInv whose slots all hold the class matrix."""
        self.ctx = ctx
        nl = local["n_lambda"]
        self.n_lambda = nl
        self.K = MatBlockDiag.from_scipy(ctx, local["block_rowstart"], local["K"])
        if regularize:
            Kreg = local["Kreg"] if "Kreg" in local else regularize_blocks(ctx, local)[0]
            self.Kreg = MatBlockDiag.from_scipy(ctx, local["block_rowstart"], Kreg)
            self._Kinv_sp = Kreg
            self.Kplus = MatInv(self.Kreg, rtol=kplus_rtol, max_it=kplus_max_it, jacobi=jacobi, nullspace=None)
        else:
            self.Kplus = MatInv(self.K, rtol=kplus_rtol, max_it=kplus_max_it, jacobi=jacobi, nullspace=local["R"])
            self._Kinv_sp = local["K"]
        if bsr3:  # K x of the inner CG on the 3x3-block kernel (BAIJ bs=3)
            self.Kplus.enable_bsr3()
        if mg_hierarchy is not None:  # -mat_inv_pc_type mg: V-cycle PC for the inner CG (feti.box_mg_hierarchy)
            self.Kplus.set_pc_mg(mg_hierarchy, degree=mg_degree, precision=mg_precision)
        elif mg_box is not None:  # the same PC, hierarchy built inside the library (pmh_mg_create_box): dict(dims=[(nx, ny, nz)], ndof, min_nodes)
            self.Kplus.set_pc_mg_box(self._Kinv_sp, mg_box["dims"], mg_box["ndof"], R=None if regularize else local["R"], min_nodes=mg_box.get("min_nodes", 400),
                                     degree=mg_degree, precision=mg_precision)
        elif mg_sa is not None:  # the same PC on blocks of ANY shape, algebraic hierarchy built inside the library (pmh_mg_create_sa): dict(ndof, max_coarse, theta, nns)
            self.Kplus.set_pc_mg_sa(self._Kinv_sp, mg_sa.get("ndof", 3), R=None if regularize else local["R"], nns=mg_sa.get("nns"), max_coarse=mg_sa.get("max_coarse", 1500),
                                    theta=mg_sa.get("theta", 0.08), degree=mg_degree, precision=mg_precision)
        self.B = MatGluing(ctx, local["n_x"], nl, local["leaves_row"], local["leaves_root"], local["leaves_sign"])
        self.E = None
        if explicit is not None:
            self.E = self.assemble_explicit(local, **explicit)
        self.has_box = bool(np.any(np.isfinite(lb)))
        self.f = ctx.vec_from(local["f"])
        # BE = G, cE = e (QPSetEq(child,G,e) qptransform.c:1169); G None: no floating subdomain, no equality constraint
        self.pf = QPPF.from_scipy(ctx, G, orthonormal=orthonormal) if G is not None else None
        if G is not None and orthonormal == "implicit":  # G, e come un-orthonormalised: the constraint is (T G) lambda = T e
            e = self.pf.orth_rhs(e)
        self.e = ctx.vec_from(e) if G is not None else None
        # QPTDualize -> QPTHomogenizeEq -> QPTEnforceEqByProjector on the device (pmh_qpt_feti_chain_create, csrc/feti.hip):
        # F = B K^+ B', d = B K^+ f - c, lambda~ = G'(GG')^{-1} e, b_bar = d - F lambda~, lb <- lb - lambda~, A = P F P | P F, b = P b_bar
        cv = ctx.vec_from(c)
        lbv = ctx.vec_from(np.asarray(lb, dtype=np.float64))  # dual box of QPTDualize (:1136-1162); all -inf = no box => A = P F
        h = C.c_void_p()
        check(ctx.L.pmh_qpt_feti_chain_create(self.B.h, self.Kplus.h, self.f.p, cv.p, self.pf.h if self.pf is not None else None,
                                              self.e.p if self.e is not None else None, lbv.p if self.has_box else None, C.byref(h)))
        self.h = h
        cv.free()
        lbv.free()
        ptr = [C.c_void_p() for _ in range(7)]
        check(ctx.L.pmh_qpt_feti_chain_get(h, *[C.byref(p) for p in ptr]))
        self.F = Op(ctx, ptr[0], nl, keep=[self.B, self.Kplus])
        self.A = self.F if ptr[1].value == ptr[0].value else Op(ctx, ptr[1], nl, keep=[self.F, self.pf])
        self.F.destroy = self.A.destroy = lambda: None  # owned by the chain
        self.d, self.b_bar, self.b, self.lb_new, self.lam_tilde = (Vec.borrowed(ctx, p, nl) if p.value else None for p in ptr[2:7])
        self.tprim = ctx.vec(local["n_x"])
        self.lam = ctx.vec(nl)  # child solution (lambda - lambda~), zero initial guess (qptransform.c:1164-1165)

    def _knob(self, name):
        v = C.c_int(1)
        check(self.ctx.L.pmh_get_knob(name.encode(), C.byref(v)))
        return v.value

    def assemble_explicit(self, local, rtol=1e-12, max_it=0, min_slots=0, solver_factory=None, share_congruent=True, storage="sym", stripe=None, symmetry=None, multi_rhs="auto"):
        """MatInvExplicitly restricted to Gamma (pmh_fexplicit_assemble): the columns come from this rank's own K^+ (one unit
        right-hand side per block and application; congruent blocks share their columns), or from a replica solver when the rank
        has fewer blocks than min_slots and all of them are congruent.  Attaches the result to K^+: every F built on it is explicit.
        stripe = (rank, size, glob): several GPUs and ALL blocks of the decomposition congruent -- the operator spans every block
        (glob: dict n_x, block_rowstart, leaves_row / _root / _sign of the whole decomposition) and this rank assembles and applies
        an even share of 128-row stripes of every W_b instead of its own blocks (pmh_fexplicit_set_stripe).
        symmetry = dict(dims=(nx, ny, nz), ndof=3) ("class_sym", all blocks one class of box-shaped blocks): the signed coordinate permutations of the
        box that leave K invariant (feti.box_symmetries, checked against K) serve the set-up: one K^+ solve per orbit of rows (a cube: 48 x fewer).
        multi_rhs: True / False / "auto" -- the set-up solves 8 columns per block at a time on the multi-right-hand-side K^+ (matinv_mv.hip: every 3 x 3 block of K_b loaded once
        for 8 columns, every launch of the V-cycle serving 8 columns; blocks need no symmetry and no congruence for it); "auto": where that solver applies (fused fp32 V-cycle on
        3 x 3 blocks, no left inverse), else one column per block.
        symmetry["close"] ("class_orbit", several classes): every class's touched set is extended to its closure under the box's operations (mat.box_symmetry_closure: the whole
        boundary of a cube), so that a class of ONE block -- a decomposition into boxes of different materials -- keeps all of them instead of the 2 ... 8 its own faces allow."""
        import scipy.sparse as sp

        rs = np.asarray(local["block_rowstart"])
        nb = len(rs) - 1
        cls = csr_block_classes(rs, self._Kinv_sp) if share_congruent else np.arange(nb, dtype=np.int32)
        one_class = int(cls.max()) == 0
        if storage == "auto":
            # congruent blocks: ONE full matrix per class applied to 8 blocks' vectors per pass ("class": ceil(|c| / 8) 8 n_c^2 bytes) when
            # that moves fewer bytes than one symmetric matrix per block ("sym": sum_b 4 n_b^2)
            src = stripe[2] if stripe is not None else dict(block_rowstart=rs, leaves_row=local["leaves_row"])
            grs = np.asarray(src["block_rowstart"])
            tr = np.unique(np.asarray(src["leaves_row"]))
            blk = np.searchsorted(grs, tr, side="right") - 1
            n_b = np.bincount(blk, minlength=len(grs) - 1).astype(float)
            gcls = np.zeros(len(grs) - 1, dtype=np.int64) if stripe is not None else cls
            b_sym = 4.0 * float(np.sum(n_b ** 2))
            b_cls = 0.0
            for c in np.unique(gcls):
                members = np.nonzero(gcls == c)[0]
                union = np.unique(np.concatenate([tr[blk == m] - grs[m] for m in members])).size
                b_cls += np.ceil(len(members) / 8.0) * 8.0 * float(union) ** 2
            storage = "class_sym" if 0.5 * b_cls < b_sym else "sym"  # the class matrix in symmetric tiles: half of b_cls
        one_class_all = one_class
        want_orbit = storage == "class_orbit" or (storage == "class_sym" and symmetry is not None and symmetry.get("orbit", True) and one_class_all)
        if storage == "class_orbit" and symmetry is None:
            raise ValueError("class_orbit needs box-shaped blocks and symmetry=dict(dims=..., ndof=...)")
        if storage == "class_orbit" and not one_class and stripe is not None:
            raise ValueError("striped class_orbit operators need congruent blocks")
        for attempt in ("class_orbit", "class_sym") if want_orbit else (storage,):
            extra = None
            if attempt == "class_orbit" and symmetry is not None and symmetry.get("close") and stripe is None:
                from .mat import box_symmetry_closure

                tr = np.unique(np.asarray(local["leaves_row"]))
                blk = np.searchsorted(rs, tr, side="right") - 1
                extra = []
                for c in range(int(cls.max()) + 1):
                    members = np.nonzero(cls == c)[0]
                    touched = np.unique(np.concatenate([tr[blk == m] - rs[m] for m in members]))
                    b0 = int(members[0])
                    closure, _ = box_symmetry_closure(symmetry["dims"], symmetry.get("ndof", 3), self._Kinv_sp[rs[b0]:rs[b0 + 1], rs[b0]:rs[b0 + 1]], touched)
                    extra.append(np.setdiff1d(closure, touched))
            E = self._create_explicit(local, attempt, stripe, cls, nb, class_extra=extra)
            self.explicit_symmetries = 1
            if symmetry is not None and attempt in ("class_sym", "class_orbit") and one_class:
                n_i = int(rs[1] - rs[0])
                Kc = self._Kinv_sp[:n_i, :n_i]  # the class matrix (what the solver inverts)
                self.explicit_symmetries = E.set_box_symmetry(0, symmetry["dims"], symmetry.get("ndof", 3), Kc)
            elif symmetry is not None and attempt == "class_orbit":
                # several classes of box-shaped blocks on the same box (e.g. one material per subdomain): every class has its own matrix, hence its own check of the operations
                nsym = []
                for c in np.unique(cls):
                    b0 = int(np.nonzero(cls == c)[0][0])
                    Kc = self._Kinv_sp[rs[b0]:rs[b0 + 1], rs[b0]:rs[b0 + 1]]
                    nsym.append(E.set_box_symmetry(int(c), symmetry["dims"], symmetry.get("ndof", 3), Kc))
                self.explicit_symmetries = int(min(nsym))
            if attempt == "class_orbit" and self.explicit_symmetries < 16 and storage != "class_orbit":
                E.destroy()  # too few operations for the GEMM form to pay: the streaming kernel on the symmetric tiles
                continue
            storage = attempt
            break
        self.explicit_storage = storage
        ngl = len(stripe[2]["block_rowstart"]) - 1 if stripe is not None else nb
        def run(mv):  # (8 columns per block need no replica solver: a rank with one block gets its 8 slots from the multi-right-hand-side K^+)
            if solver_factory is not None and nb < min_slots and one_class and not mv:
                solver = solver_factory(int(min_slots))
                E.assemble(solver, slot_class=np.zeros(solver.K.nblocks, dtype=np.int32), block_class=np.zeros(ngl, dtype=np.int32), rtol=rtol, max_it=max_it)
                self._replica_solver = solver
            elif stripe is not None:
                E.assemble(self.Kplus, slot_class=np.zeros(nb, dtype=np.int32), block_class=np.zeros(ngl, dtype=np.int32), rtol=rtol, max_it=max_it, multi_rhs=mv)
            else:
                E.assemble(self.Kplus, slot_class=cls, block_class=cls, rtol=rtol, max_it=max_it, multi_rhs=mv)

        self.explicit_multi_rhs = False
        if multi_rhs == "auto" and not self._knob("multi_rhs"):
            run(False)  # PMH_NO_MULTI_RHS / pmh_set_knob("multi_rhs", 0): the A/B switch
        elif multi_rhs == "auto":
            from ._lib import PermonHipError

            try:
                run(True)
                self.explicit_multi_rhs = True
            except PermonHipError as ex:
                if getattr(ex, "code", 0) != 4:  # PMH_ERR_SUP: the multi-right-hand-side solver does not apply here
                    raise
                import os
                import sys

                if os.environ.get("PMH_MV_VERBOSE") or os.environ.get("PMH_PROGRESS"):
                    sys.stderr.write("assemble_explicit: one column per block and application (%s)\n" % ex)
                run(False)
        else:
            run(bool(multi_rhs))
            self.explicit_multi_rhs = bool(multi_rhs)
        self.Kplus.attach_explicit(E)
        return E

    def _create_explicit(self, local, storage, stripe, cls, nb, class_extra=None):
        """The pmh_fexplicit object of assemble_explicit for one storage (striped over all blocks of the decomposition, or this rank's blocks)."""
        import scipy.sparse as sp

        if stripe is not None:
            rank, size, glob = stripe
            if int(cls.max()) != 0:
                raise ValueError("striped explicit operators need congruent blocks")
            self._Bglob = MatGluing(self.ctx, glob["n_x"], self.n_lambda, glob["leaves_row"], glob["leaves_root"], glob["leaves_sign"])
            self._Kglob = MatBlockDiag.from_scipy(self.ctx, glob["block_rowstart"], sp.identity(glob["n_x"], format="csr"))  # block structure only
            ngl = len(glob["block_rowstart"]) - 1
            E = MatExplicitDual(self._Bglob, self._Kglob, storage=storage if storage in ("class", "class_sym", "class_orbit") else "sym", block_class=np.zeros(ngl, dtype=np.int32))
            E.set_stripe(rank, size)
        else:
            E = MatExplicitDual(self.B, self.Kreg if hasattr(self, "Kreg") else self.K, storage=storage, block_class=cls, class_extra=class_extra)
        return E

    def make_smalxe(self, rtol=1e-5, max_it=100, inner=None, **smalxe):
        """QPS of type SMALXE on the projected dual QP, set up but not solved."""
        qp = QP(self.ctx)
        qp.SetOperator(self.A)
        qp.SetRhs(self.b)
        qp.SetInitialVector(self.lam)
        qp.lb, qp.ub = self.lb_new, None
        qp.SetEq(self.pf)
        qps = QPS(self.ctx)
        qps.SetQP(qp)
        qps.SetType("smalxe")
        qps.SetTolerances(rtol=rtol, max_it=max_it)
        for k, v in smalxe.items():
            setattr(qps.smalxe_opts, k, v)
        for k, v in (inner or {}).items():
            setattr(qps.smalxe_opts.inner, k, v)
        qps.SetUp()
        self.qps = qps
        return qps

    def solve_smalxe(self, rtol=1e-5, max_it=100, inner=None, **smalxe):
        """QPSSetDefaultType: BE present -> SMALXE with inner MPGP (qps.c:443-444)."""
        return self.make_smalxe(rtol=rtol, max_it=max_it, inner=inner, **smalxe).Solve()

    def solve_pcpg(self, rtol=1e-5, max_it=1000, lumped=False):
        """Equality-only dual QP (no box): projected preconditioned CG (QPSPCPG)."""
        qp = QP(self.ctx)
        qp.SetOperator(self.F)
        qp.SetRhs(self.b_bar)
        qp.SetInitialVector(self.lam)
        qp.SetEq(self.pf)
        qp.pc = PCDualLumpedOp(self.B, self.K) if lumped else None
        qps = QPS(self.ctx)
        qps.SetQP(qp)
        qps.SetType("pcpg")
        qps.SetTolerances(rtol=rtol, max_it=max_it)
        st = qps.Solve()
        self.qps = qps
        return st

    def solve_ksp(self, rtol=1e-5, max_it=10000, lumped=False):
        """The reference's default for a dual QP without box: after QPTEnforceEqByProjector the child QP has operator
        P F, rhs P b_bar, preconditioner P M^{-1} (qptransform.c:272-308) and no constraint left, so QPSSetDefaultType
        picks QPSKSP = CG (qps.c:448); without floating subdomains it is CG on F lambda = d."""
        qp = QP(self.ctx)
        qp.SetOperator(self.A)
        qp.SetRhs(self.b)
        qp.SetInitialVector(self.lam)
        if lumped:
            self._lumped = PCDualLumpedOp(self.B, self.K)
            qp.pc = MatCreateProjected(self._lumped, self.pf, symmetric=False) if self.pf is not None else self._lumped
        qps = QPS(self.ctx)
        qps.SetQP(qp)
        qps.SetType("ksp")
        qps.SetTolerances(rtol=rtol, max_it=max_it)
        st = qps.Solve()
        self.qps = qps
        return st

    def dual_solution(self):
        """lambda = lambda_child + lambda~ (QPTHomogenizeEqPostSolve_Private qptransform.c:423-431)."""
        lam = self.ctx.vec(self.n_lambda)
        check(self.ctx.L.pmh_qpt_feti_chain_post_solve(self.h, self.lam.p, lam.p, None, None))
        out = lam.to_numpy()
        lam.free()
        return out

    def primal_solution(self, G_host, e_host=None):
        """The two device-side pieces of QPTDualizePostSolve_Private (qptransform.c:783-833): returns
        (K^+(f - B' lambda), F lambda - d).  The caller finishes on the host with the small coarse solve:
        u = K^+(f - B' lambda) - R alpha,  G' alpha = d - F lambda, i.e. alpha = -(G G')^{-1} G (F lambda - d)."""
        ctx = self.ctx
        lam, u, Fl = ctx.vec(self.n_lambda), ctx.vec(self.tprim.n), ctx.vec(self.n_lambda)
        check(ctx.L.pmh_qpt_feti_chain_post_solve(self.h, self.lam.p, lam.p, u.p, Fl.p))
        out = (u.to_numpy(), Fl.to_numpy())
        for v in (lam, u, Fl):
            v.free()
        return out


def KSPFETISolve(ctx, block_rowstart, K, f, l2g, dirichlet_local=None, R=None, gluing="full", scale=True, exclude_dirichlet=False, regularize=None, lumped=False,
                 rtol=1e-5, atol=1e-50, divtol=1e4, max_it=10000, kplus_rtol=1e-12, kplus_max_it=20000, options=None, regularize_rho=0.0, explicit=False, kplus_pc="jacobi"):
    """KSPFETI (src/ksp/impls/feti/feti.c:71-156) for a decomposed linear problem, one call into pmh_kspfeti_solve (C++):
    K block-diagonal scipy CSR, f split among copies, l2g global dof of every local dof, dirichlet_local = local dofs enforced
    by B (TFETI) or None, R = (kdim, N) kernel vectors (zero over non-floating blocks) or None.
    regularize: None = the library's default K^+, which is KSPFETI's (the left generalised inverse K^- P_R; K_reg^{-1} with explicit=True); True = K_reg^{-1} (MatRegularize);
    False = the Moore-Penrose form P_R K^- P_R.
    Returns (u, lambda, stats) with stats = (iteration, reason, rnorm, n_lambda, n_dirichlet_rows, coarse_dim)."""
    from . import _lib

    K = K.tocsr()
    K.sort_indices()
    N = K.shape[0]
    rs = np.ascontiguousarray(block_rowstart, dtype=np.int32)
    ip, ci, va = (np.ascontiguousarray(K.indptr, dtype=np.int32), np.ascontiguousarray(K.indices, dtype=np.int32), np.ascontiguousarray(K.data, dtype=np.float64))
    fv = np.ascontiguousarray(f, dtype=np.float64)
    lg = np.ascontiguousarray(l2g, dtype=np.int32)
    dl = np.ascontiguousarray(dirichlet_local if dirichlet_local is not None else [], dtype=np.int32)
    Rm = np.ascontiguousarray(R, dtype=np.float64) if R is not None and np.size(R) else np.zeros((0, N))
    assert fv.size == N and lg.size == N and Rm.shape[1] == N
    o = _lib.KspFetiOpts()
    check(ctx.L.pmh_kspfeti_default_opts(C.byref(o)))
    o.gluing_type, o.scale, o.exclude_dirichlet = {"nonred": 0, "full": 1, "orth": 2}[gluing], int(bool(scale)), int(bool(exclude_dirichlet))
    if regularize is not None:
        o.kplus_left, o.regularize = 0, int(bool(regularize))
    o.lumped_pc, o.regularize_rho = int(bool(lumped)), float(regularize_rho)
    o.kplus_rtol, o.kplus_max_it, o.rtol, o.atol, o.divtol, o.max_it = kplus_rtol, kplus_max_it, rtol, atol, divtol, max_it
    o.explicit_dual = int(bool(explicit))  # F through the explicit local dual operators (pmh_fexplicit_*)
    o.kplus_pc = {"jacobi": 0, "gamg": 1, "mg": 1}[kplus_pc]  # -dual_mat_inv_pc_type: the algebraic V-cycle (pmh_mg_create_sa) as the PC of MATINV's inner KSP
    if options:  # the reference's command line on top of the keyword arguments (pmh_kspfeti_set_from_options)
        left = C.create_string_buffer(2048)
        check(ctx.L.pmh_kspfeti_set_from_options(options.encode(), C.byref(o), left, len(left)))
    st = _lib.KspFetiStats()
    u = np.zeros(N)
    cap = int(dl.size + 8 * N)
    lam = np.zeros(cap)
    p = lambda a: a.ctypes.data_as(C.c_void_p) if a.size else None  # noqa: E731
    view = C.create_string_buffer(8192)  # -qps_view_convergence / -qp_chain_view_kkt / -qpt_matis_to_diag_norm text, if asked for in `options`
    o.view_buf, o.view_cap = C.cast(view, C.c_char_p), len(view)
    check(ctx.L.pmh_kspfeti_solve(ctx.h, rs.size - 1, p(rs), p(ip), p(ci), p(va), p(fv), p(lg), dl.size, p(dl), Rm.shape[0], p(Rm), C.byref(o), p(u), p(lam), cap, C.byref(st)))
    st.view_text = view.value.decode()
    return u, lam[:st.n_lambda].copy(), st


def FETIContactSolve(ctx, f, explicit=True, mg_precision="fp16", rtol=1e-5, kplus_rtol=1e-9, explicit_rtol=1e-12, mg_min_nodes=400, dims=None, explicit_storage=None, explicit_symmetry=True):
    """pmh_feti_contact_solve (contact.hip): the whole contact TFETI solve in ONE library call -- QPTFromOptions / QPTAllInOne
    (qptransform.c:2152-2237) + QPSSolve + the post-solve chain; f: a CubeFeti-like problem (K, f, leaves, c, R, n_eq).
    Returns (u, lambda, stats: _lib.FetiContactStats)."""
    from . import _lib

    K = f.K.tocsr()
    K.sort_indices()
    o, st = _lib.FetiContactOpts(), _lib.FetiContactStats()
    check(ctx.L.pmh_feti_contact_default_opts(C.byref(o)))
    o.smalxe.rtol, o.kplus_rtol, o.explicit_dual, o.explicit_rtol = rtol, kplus_rtol, int(bool(explicit)), explicit_rtol
    o.mg_precision, o.mg_min_nodes = {"fp64": 0, "fp32": 1, "fp16": 2}[mg_precision], int(mg_min_nodes)
    if explicit_storage is not None:
        o.explicit_storage = {"full": 0, "sym": 1, "class": 2, "class_sym": 3, "class_orbit": 4}[explicit_storage]
    o.explicit_symmetry = int(bool(explicit_symmetry))
    if dims is None and hasattr(f, "nel"):
        dims = [(f.nel + 1,) * 3] * f.nsub
    a32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)  # noqa: E731
    a64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
    arrs = dict(rs=a32(f.block_rowstart), ip=a32(K.indptr), ci=a32(K.indices), va=a64(K.data), ff=a64(f.f), lr=a32(f.leaves_row), lo=a32(f.leaves_root), ls=a64(f.leaves_sign),
                cc=a64(f.c), R=a64(f.R), dm=a32(dims) if dims is not None else None)
    p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None  # noqa: E731
    u, lam = np.zeros(f.N), np.zeros(f.n_lambda)
    check(ctx.L.pmh_feti_contact_solve(ctx.h, f.nsub, p(arrs["rs"]), p(arrs["ip"]), p(arrs["ci"]), p(arrs["va"]), p(arrs["ff"]), f.n_lambda, f.n_eq, arrs["lr"].size, p(arrs["lr"]), p(arrs["lo"]),
                                       p(arrs["ls"]), p(arrs["cc"]), arrs["R"].shape[0], p(arrs["R"]), p(arrs["dm"]), f.ndof, C.byref(o), p(u), p(lam), C.byref(st)))
    return u, lam, st
