"""Options front end (pmh_qps_set_from_options, csrc/options.hip): the reference's option keys, value syntax and setter
argument checks (qps.c:860-930, mpgp.c:712-745, smalxe.c:696-766).  Host-only C ABI calls."""
import ctypes as C

import pytest

from permon_amd import _lib


def _parse(opts, prefix="", smalxe=True):
    L = _lib.load()
    q, m, s = _lib.QpsOpts(), _lib.MpgpOpts(), _lib.SmalxeOpts()
    _lib.check(L.pmh_qps_default_opts(C.byref(q)))
    _lib.check(L.pmh_mpgp_default_opts(C.byref(m)))
    _lib.check(L.pmh_smalxe_default_opts(C.byref(s)))
    left = C.create_string_buffer(1024)
    rc = L.pmh_qps_set_from_options(opts.encode(), prefix.encode(), C.byref(q), C.byref(m), C.byref(s) if smalxe else None, left, len(left))
    return rc, q, m, s, left.value.decode().split(), L.pmh_last_error().decode()


def test_reference_test_block_args():
    # src/tutorials/ex1.c:165-184 (TEST blocks)
    rc, q, m, s, left, _ = _parse("-n 100 -qps_view_convergence -qp_chain_view_kkt")
    assert rc == 0 and q.view_convergence == 1 and q.type == b"" and left == ["-n", "-qp_chain_view_kkt"]
    assert (m.exptype, m.explengthtype) == (0, 0) and (q.rtol, q.atol, q.divtol, q.max_it) == (1e-5, 1e-50, 1e4, 10000)
    for args, exp in (("-qps_mpgp_expansion_type gf -qps_mpgp_expansion_length_type opt", (2, 1)), ("-qps_mpgp_expansion_type g -qps_mpgp_expansion_length_type optapprox", (3, 2)),
                      ("-qps_mpgp_expansion_type gfgr -qps_mpgp_expansion_length_type bb", (4, 3)), ("-qps_mpgp_expansion_type projcg", (1, 0)), ("-qps_mpgp_expansion_type GGR", (5, 0))):
        rc, q, m, s, left, _ = _parse(args)
        assert rc == 0 and (m.exptype, m.explengthtype) == exp and not left
    # jbearing2.c:587, feti/ex71.c:442, feti/ex1.c:130
    rc, q, m, s, left, _ = _parse("-tao_gttol 1e-6 -qps_view_convergence -qps_type mpgp -mx 8 -my 12")
    assert rc == 0 and q.type == b"mpgp" and left == ["-tao_gttol", "-mx", "-my"]
    rc, q, m, s, left, _ = _parse("-pde_type Elasticity -dim 3 -qps_rtol 1e-6 -dual_pc_dual_type lumped")
    assert rc == 0 and q.rtol == 1e-6 and left == ["-pde_type", "-dim", "-dual_pc_dual_type"]
    rc, q, m, s, left, _ = _parse("-project 0 -qps_smalxe_rho 1e1 -dual_qp_E_orth_type gs")
    assert rc == 0 and s.rho_user == 10.0 and s.rho_direct == 0 and left == ["-project", "-dual_qp_E_orth_type"]


def test_value_syntax_and_prefixes():
    rc, q, m, s, left, _ = _parse("-qps_max_it 77 -qps_atol 1e-9 -qps_divtol 1e3 -qps_mpgp_gamma 0.5 -qps_mpgp_alpha 1.5 -qps_mpgp_fallback2 -qps_mpgp_fallback true -qps_monitor")
    assert rc == 0 and (q.max_it, q.max_it_set, q.atol, q.divtol, q.monitor) == (77, 1, 1e-9, 1e3, 1)
    assert m.gamma == 0.5 and m.alpha_user == 1.5 and m.alpha_direct == 0
    assert m.fallback2 == 1 and m.fallback == 0  # mpgp.c:743: fallback2 switches fallback off
    rc, q, m, s, left, _ = _parse("-qps_mpgp_alpha 0.01 -qps_mpgp_alpha_direct")
    assert rc == 0 and m.alpha_user == 0.01 and m.alpha_direct == 1
    rc, q, m, s, left, _ = _parse("-qps_mpgp_maxeig -1 -qps_mpgp_maxeig_tol 1e-3 -qps_mpgp_maxeig_iter 30 -qps_mpgp_alpha_reset false")  # "-1" is a value, not a key
    assert rc == 0 and m.maxeig == -1.0 and m.maxeig_tol == 1e-3 and m.maxeig_iter == 30 and m.resetalpha == 0
    # SMALXE keys and its inner solver's prefix (smalxe.c:500-502)
    rc, q, m, s, left, _ = _parse("-qps_type smalxe -qps_smalxe_M1 5 -qps_smalxe_M1_direct 1 -qps_smalxe_eta 0.2 -qps_smalxe_rho_update 2 -qps_smalxe_maxeig_inject 0 "
                                  "-smalxe_qps_mpgp_gamma 2 -smalxe_qps_max_it 500 -smalxe_qps_mpgp_expansion_type gf -qps_smalxe_rtol_E 1e-3")
    assert rc == 0 and q.type == b"smalxe" and not left
    assert (s.M1_user, s.M1_direct, s.eta_user, s.rho_update, s.inject_maxeig, s.inject_maxeig_set, s.rtol_E) == (5.0, 1, 0.2, 2.0, 0, 1, 1e-3)
    assert (s.inner.gamma, s.inner.max_it, s.inner.exptype) == (2.0, 500, 2) and m.gamma == 1.0
    # an object prefix: only prefixed keys are this solver's
    rc, q, m, s, left, _ = _parse("-dual_qps_rtol 1e-7 -qps_rtol 1e-3 -dual_smalxe_qps_mpgp_gamma 3", prefix="dual_")
    assert rc == 0 and q.rtol == 1e-7 and s.inner.gamma == 3.0 and left == ["-qps_rtol"]


@pytest.mark.parametrize("opts,msg", [
    ("-qps_mpgp_maxeig_iter 1", "Argument must be > 1"),  # mpgp.c:1088
    ("-qps_mpgp_maxeig -3", "Argument must be nonnegative"),  # mpgp.c:995
    ("-qps_smalxe_rho 0", "Argument must be positive"),  # smalxe.c:1315
    ("-qps_smalxe_rho_update 0.5", "Argument must be >= 1"),  # smalxe.c:1361
    ("-qps_rtol 1.5", "must be non-negative and less than 1.0"),  # qps.c:913
    ("-qps_max_it -5", "must be non-negative"),  # qps.c:925
    ("-qps_type tao", "Unable to find requested QPS type"),  # qps.c:394 (TAO wrapper: out of scope)
    ("-qps_mpgp_expansion_type fancy", "unknown value"),
    ("-qps_mpgp_gamma", "needs a value"),
    ("-qps_mpgp_fallback maybe", "unknown logical value"),
    ("stray -qps_rtol 1e-3", "expected an option key"),
])
def test_argument_checks_of_the_reference_setters(opts, msg):
    rc, q, m, s, left, err = _parse(opts)
    assert rc != 0 and msg in err


def test_feti_driver_options():
    """pmh_kspfeti_set_from_options: the keys of QPFetiSetUp / QPFetiGetBgtSF / QPTFromOptions / QPTDualize / PCDUAL
    (qpfeti.c:340-341,757-758, qptransform.c:1019,2231, pcdual.c:170) as the ex71 TEST blocks use them (feti/ex71.c:438-442)."""
    L = _lib.load()

    def parse(opts):
        o = _lib.KspFetiOpts()
        _lib.check(L.pmh_kspfeti_default_opts(C.byref(o)))
        left = C.create_string_buffer(512)
        rc = L.pmh_kspfeti_set_from_options(opts.encode(), C.byref(o), left, len(left))
        return rc, o, left.value.decode().split()

    rc, o, left = parse("-pde_type Poisson -cells 7,8,9 -dim 3 -feti_gluing_type orth -qps_view_convergence -qp_chain_view_kkt")
    assert rc == 0 and o.gluing_type == 2 and (o.scale, o.regularize, o.lumped_pc, o.rtol) == (1, 1, 0, 1e-5)
    assert left == ["-pde_type", "-cells", "-dim"] and (o.view_convergence, o.view_kkt, o.matis_to_diag_norm) == (1, 1, 0)  # the two view keys configure the post-solve report
    rc, o, left = parse("-pde_type Elasticity -dim 3 -qps_rtol 1e-6 -dual_pc_dual_type lumped")
    assert rc == 0 and o.lumped_pc == 1 and o.rtol == 1e-6 and o.gluing_type == 1
    rc, o, left = parse("-feti_gluing_type NONRED -SCALE_ON 0 -feti_gluing_exclude_dirichlet -regularize false -dual_mat_inv_ksp_rtol 1e-10 -qps_max_it 50")
    assert rc == 0 and (o.gluing_type, o.scale, o.exclude_dirichlet, o.regularize, o.kplus_rtol, o.max_it) == (0, 0, 1, 0, 1e-10, 50) and not left
    rc, o, left = parse("-qpt_dualize_Kplus_mp")
    assert rc == 0 and o.regularize == 0 and o.kplus_left == 0  # the Moore-Penrose form wins over the left inverse (qptransform.c:1018-1019)
    # feti/ex1.c's TEST block smalxe_orth: the unprojected dual QP, SMALXE's rho, the orthonormalisation of G (qptransform.c:2228, smalxe.c:716, qptransform.c:653)
    rc, o, left = parse("-project 0 -qps_smalxe_rho 1e1 -dual_qp_E_orth_type implicit -qpt_dualize_Kplus_left -smalxe_qps_max_it 77")
    assert rc == 0 and not left and (o.project, o.E_orth_type, o.kplus_left, o.regularize) == (0, 4, 1, 0) and o.smalxe.rho_user == 10.0 and o.smalxe.inner.max_it == 77
    rc, o, left = parse("-dual_qp_E_orth_type gs")
    assert rc == 0 and (o.project, o.E_orth_type, o.kplus_left) == (1, 1, 1) and o.smalxe.rho_user == 1.1  # defaults: projected, KSPFETI's left inverse, smalxe.c:1190
    rc, o, left = parse("-qpt_dualize_Kplus_left 0")
    assert rc == 0 and (o.kplus_left, o.regularize) == (0, 1)  # MatRegularize
    assert parse("-dual_qp_E_orth_type cholesky")[1].E_orth_type == 3 and parse("-dual_qp_E_orth_type gslingen")[1].E_orth_type == 2
    assert parse("-dual_qp_E_orth_type inexact")[0] != 0  # the one MatOrthType this library does not build: said, not ignored
    assert parse("-feti_gluing_type sideways")[0] != 0 and parse("-qps_rtol 2")[0] != 0
    # -qpt_dualize_Kplus_mp wins over -qpt_dualize_Kplus_left in EITHER order (qptransform.c:1018-1019 reads _left only if !true_mp)
    for opts in ("-qpt_dualize_Kplus_mp -qpt_dualize_Kplus_left 1", "-qpt_dualize_Kplus_left 1 -qpt_dualize_Kplus_mp"):
        rc, o, left = parse(opts)
        assert rc == 0 and (o.kplus_left, o.regularize) == (0, 0), opts
    # an explicit -qps_max_it 10000 is "given" (it must reach SMALXE with -project 0 instead of falling to its default of 100): tracked, not compared with the default value
    assert parse("-project 0 -qps_max_it 10000")[1].max_it_set == 1 and parse("-project 0")[1].max_it_set == 0
