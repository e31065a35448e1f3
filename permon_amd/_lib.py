"""ctypes binding of libpermonhip.so (include/permon_hip.h).  Fails loudly when the HIP extension is
missing or no MI355X is visible: there is no CPU fallback anywhere in permon_amd."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpermonhip.so")

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)
vp = C.c_void_p


class PermonHipError(RuntimeError):
    pass


class MpgpOpts(C.Structure):
    _fields_ = [
        ("rtol", C.c_double), ("atol", C.c_double), ("divtol", C.c_double), ("max_it", C.c_int),
        ("alpha_user", C.c_double), ("alpha_direct", C.c_int), ("gamma", C.c_double),
        ("maxeig", C.c_double), ("maxeig_tol", C.c_double), ("maxeig_iter", C.c_int),
        ("bchop_tol", C.c_double), ("astol", C.c_double),
        ("exptype", C.c_int), ("explengthtype", C.c_int),
        ("resetalpha", C.c_int), ("fallback", C.c_int), ("fallback2", C.c_int),
        ("monitor", C.c_int), ("unfused", C.c_int), ("distributed", C.c_int),
    ]


class MpgpStats(C.Structure):
    _fields_ = [
        ("iteration", C.c_int), ("reason", C.c_int),
        ("rnorm", C.c_double), ("gfnorm", C.c_double), ("gcnorm", C.c_double), ("alpha", C.c_double), ("maxeig", C.c_double),
        ("nmv", C.c_int), ("ncg", C.c_int), ("nexp", C.c_int), ("nprop", C.c_int), ("nfinc", C.c_int), ("nfall", C.c_int),
        ("norm_rhs", C.c_double), ("ttol", C.c_double),
        ("current_step_type", C.c_char),
    ]


class SmalxeOpts(C.Structure):
    _fields_ = [
        ("rtol", C.c_double), ("atol", C.c_double), ("divtol", C.c_double), ("max_it", C.c_int),
        ("M1_user", C.c_double), ("M1_direct", C.c_int), ("M1_update", C.c_double),
        ("rtol_E", C.c_double),
        ("rho_user", C.c_double), ("rho_direct", C.c_int), ("rho_update", C.c_double), ("rho_update_late", C.c_double),
        ("eta_user", C.c_double), ("eta_direct", C.c_int),
        ("update_threshold", C.c_double),
        ("maxeig", C.c_double), ("maxeig_tol", C.c_double), ("maxeig_iter", C.c_int),
        ("inject_maxeig", C.c_int), ("inject_maxeig_set", C.c_int),
        ("inner_iter_min", C.c_int), ("inner_no_gtol_stop", C.c_int),
        ("be_implicit", C.c_int), ("lag_enabled", C.c_int), ("lag_offset", C.c_int), ("lag_start", C.c_int), ("lag_step", C.c_int), ("lag_end", C.c_int),
        ("lag_lower", C.c_double), ("lag_upper", C.c_double), ("knoll", C.c_int),
        ("inner", MpgpOpts),
    ]


class SmalxeStats(C.Structure):
    _fields_ = [
        ("iteration", C.c_int), ("reason", C.c_int), ("inner_iter_accu", C.c_int), ("state", C.c_int),
        ("M1_hits", C.c_int), ("eta_hits", C.c_int), ("M1_updates", C.c_int), ("rho_updates", C.c_int),
        ("M1", C.c_double), ("rho", C.c_double), ("eta", C.c_double), ("normBu", C.c_double), ("enorm", C.c_double),
        ("rnorm", C.c_double), ("maxeig", C.c_double),
        ("inner", MpgpStats),
    ]


class QpsOpts(C.Structure):
    _fields_ = [
        ("type", C.c_char * 16), ("rtol", C.c_double), ("atol", C.c_double), ("divtol", C.c_double), ("max_it", C.c_int), ("max_it_set", C.c_int),
        ("monitor", C.c_int), ("monitor_cost", C.c_int), ("view", C.c_int), ("view_convergence", C.c_int), ("auto_post_solve", C.c_int),
    ]


class KspFetiOpts(C.Structure):
    _fields_ = [("gluing_type", C.c_int), ("scale", C.c_int), ("exclude_dirichlet", C.c_int), ("regularize", C.c_int), ("kplus_left", C.c_int), ("project", C.c_int),
                ("E_orth_type", C.c_int), ("lumped_pc", C.c_int), ("regularize_rho", C.c_double),
                ("kplus_rtol", C.c_double), ("kplus_max_it", C.c_int), ("rtol", C.c_double), ("atol", C.c_double), ("divtol", C.c_double), ("max_it", C.c_int), ("max_it_set", C.c_int),
                ("explicit_dual", C.c_int), ("explicit_rtol", C.c_double), ("view_convergence", C.c_int), ("view_kkt", C.c_int), ("matis_to_diag_norm", C.c_int),
                ("view_buf", C.c_char_p), ("view_cap", C.c_int), ("smalxe", SmalxeOpts), ("kplus_pc", C.c_int), ("kplus_pc_ndof", C.c_int)]


class KspFetiStats(C.Structure):
    _fields_ = [("iteration", C.c_int), ("reason", C.c_int), ("rnorm", C.c_double), ("n_lambda", C.c_int), ("n_dirichlet_rows", C.c_int), ("coarse_dim", C.c_int),
                ("smalxe", SmalxeStats)]


class FetiContactOpts(C.Structure):
    _fields_ = [("smalxe", SmalxeOpts), ("kplus_rtol", C.c_double), ("kplus_max_it", C.c_int), ("mg", C.c_int), ("mg_min_nodes", C.c_int), ("mg_degree", C.c_int), ("mg_precision", C.c_int),
                ("bsr3", C.c_int), ("explicit_dual", C.c_int), ("explicit_rtol", C.c_double), ("explicit_storage", C.c_int), ("orthonormalize", C.c_int), ("explicit_symmetry", C.c_int)]


class FetiContactStats(C.Structure):
    _fields_ = [("smalxe", SmalxeStats), ("n_lambda", C.c_int), ("n_eq", C.c_int), ("coarse_dim", C.c_int), ("n_active", C.c_int), ("explicit_solves", C.c_int),
                ("setup_seconds", C.c_double), ("solve_seconds", C.c_double), ("explicit_seconds", C.c_double), ("norm_Glambda_minus_e", C.c_double), ("explicit_symmetries", C.c_int)]


class PcpgStats(C.Structure):
    _fields_ = [("iteration", C.c_int), ("reason", C.c_int), ("rnorm", C.c_double)]


SHELL_MULT_FN = C.CFUNCTYPE(C.c_int, vp, vp, vp)
CONVERGED_FN = C.CFUNCTYPE(C.c_int, vp, C.c_int, C.c_double, c_int_p)

_PROTOS = {
    "pmh_init": [C.c_int, C.POINTER(vp)],
    "pmh_finalize": [vp],
    "pmh_comm_timing_enable": [vp, C.c_int],
    "pmh_comm_timing_get": [vp, c_int_p, c_double_p, c_double_p],
    "pmh_set_knob": [C.c_char_p, C.c_int],
    "pmh_get_knob": [C.c_char_p, c_int_p],
    "pmh_device_name": [vp, C.c_char_p, C.c_size_t],
    "pmh_sync": [vp],
    "pmh_mem_info": [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)],
    "pmh_malloc": [vp, C.c_size_t, C.POINTER(vp)],
    "pmh_free": [vp, vp],
    "pmh_memcpy_h2d": [vp, vp, vp, C.c_size_t],
    "pmh_memcpy_d2h": [vp, vp, vp, C.c_size_t],
    "pmh_memcpy_d2d": [vp, vp, vp, C.c_size_t],
    "pmh_memset": [vp, vp, C.c_int, C.c_size_t],
    "pmh_timer_start": [vp],
    "pmh_timer_stop": [vp, c_double_p],
    "pmh_comm_unique_id": [vp],
    "pmh_comm_init": [vp, C.c_int, C.c_int, vp],
    "pmh_comm_rank": [vp, c_int_p, c_int_p],
    "pmh_comm_allreduce_sum": [vp, vp, C.c_size_t],
    "pmh_comm_allreduce_min": [vp, vp, C.c_size_t],
    "pmh_comm_barrier": [vp],
    "pmh_comm_set_host_transport": [vp, C.c_int, C.c_int, vp, vp],
    "pmh_csr_create": [vp, C.c_int, C.c_int, vp, vp, vp, C.POINTER(vp)],
    "pmh_csr_destroy": [vp],
    "pmh_csr_sizes": [vp, c_int_p, c_int_p, C.POINTER(C.c_longlong)],
    "pmh_csr_mult": [vp, vp, vp],
    "pmh_csr_mult_add": [vp, vp, vp, vp],
    "pmh_csr_mult_transpose": [vp, vp, vp],
    "pmh_csr_mult_transpose_add": [vp, vp, vp, vp],
    "pmh_csr_algorithmic_bytes": [vp, c_double_p],
    "pmh_csr_timing_enable": [vp, C.c_int],
    "pmh_csr_timing_get": [vp, C.c_int, c_int_p, c_double_p],
    "pmh_op_create_csr": [vp, C.POINTER(vp)],
    "pmh_op_create_shell": [vp, C.c_int, SHELL_MULT_FN, vp, C.POINTER(vp)],
    "pmh_op_destroy": [vp],
    "pmh_op_size": [vp, c_int_p],
    "pmh_op_mult": [vp, vp, vp],
    "pmh_op_mult_transpose": [vp, vp, vp],
    "pmh_op_penalized_mult_add": [vp, vp, vp, vp],
    "pmh_op_penalized_mult_transpose_add": [vp, vp, vp, vp],
    "pmh_op_max_eigenvalue": [vp, C.c_double, C.c_int, c_double_p, c_int_p],
    "pmh_qpc_box_project": [vp, C.c_int, vp, vp, vp, vp],
    "pmh_qpc_box_feas": [vp, C.c_int, vp, vp, vp, vp, c_double_p],
    "pmh_qpc_box_grads": [vp, C.c_int, vp, vp, vp, vp, C.c_double, vp, vp],
    "pmh_qpc_box_gradreduced": [vp, C.c_int, vp, vp, vp, vp, C.c_double, vp],
    "pmh_qpc_box_expand_is": [vp, C.c_int, C.c_int, vp, vp, C.c_double, vp],
    "pmh_qp_kkt_box": [vp, vp, vp, vp, vp, vp, c_double_p],
    "pmh_vec_axpy": [vp, C.c_int, vp, C.c_double, vp],
    "pmh_vec_aypx": [vp, C.c_int, vp, C.c_double, vp],
    "pmh_vec_waxpy": [vp, C.c_int, vp, C.c_double, vp, vp],
    "pmh_vec_scale": [vp, C.c_int, vp, C.c_double],
    "pmh_vec_set": [vp, C.c_int, vp, C.c_double],
    "pmh_vec_copy": [vp, C.c_int, vp, vp],
    "pmh_vec_dot": [vp, C.c_int, vp, vp, c_double_p],
    "pmh_vec_norm2": [vp, C.c_int, vp, c_double_p],
    "pmh_mpgp_default_opts": [C.POINTER(MpgpOpts)],
    "pmh_mpgp_create": [vp, vp, vp, vp, vp, vp, C.POINTER(MpgpOpts), C.POINTER(vp)],
    "pmh_mpgp_destroy": [vp],
    "pmh_mpgp_solve": [vp],
    "pmh_mpgp_get_stats": [vp, C.POINTER(MpgpStats)],
    "pmh_mpgp_set_convergence_test": [vp, CONVERGED_FN, vp],
    "pmh_mpgp_set_tolerances": [vp, C.c_double, C.c_double, C.c_double, C.c_int],
    "pmh_mpgp_get_trace": [vp, C.c_int, C.c_char_p, c_double_p, c_double_p, c_double_p, c_double_p, c_int_p],
    "pmh_mpgp_get_work": [vp, C.c_int, C.POINTER(vp)],
    "pmh_mpgp_set_operator_max_eigenvalue": [vp, C.c_double],
    "pmh_mpgp_update_max_eigenvalue": [vp, C.c_double],
    "pmh_mpgp_get_current_step_type": [vp, C.c_char_p],
    "pmh_mpgp_reset_statistics": [vp],
    "pmh_mpgp_run_fixed": [vp, C.c_int],
    "pmh_mpgp_get_tolerances": [vp, c_double_p, c_double_p, c_double_p, c_int_p],
    "pmh_qppf_create": [vp, vp, C.c_int, C.POINTER(vp)],
    "pmh_qppf_destroy": [vp],
    "pmh_qppf_orth_rhs": [vp, vp, vp],
    "pmh_qppf_setup_stats": [vp, c_double_p, c_double_p, c_double_p],
    "pmh_qppf_apply_Q": [vp, vp, vp],
    "pmh_qppf_apply_P": [vp, vp, vp],
    "pmh_qppf_apply_GtG": [vp, vp, vp],
    "pmh_qppf_apply_CP": [vp, vp, vp],
    "pmh_qppf_apply_halfQ": [vp, vp, vp],
    "pmh_qppf_apply_halfQ_transpose": [vp, vp, vp],
    "pmh_qppf_apply_G": [vp, vp, vp],
    "pmh_op_create_penalized": [vp, vp, C.c_double, C.POINTER(vp)],
    "pmh_op_penalized_set_penalty": [vp, C.c_double],
    "pmh_op_penalized_get_penalty": [vp, c_double_p],
    "pmh_op_create_projected": [vp, vp, C.c_int, C.POINTER(vp)],
    "pmh_gluing_create": [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.POINTER(vp)],
    "pmh_gluing_destroy": [vp],
    "pmh_gluing_mult": [vp, vp, vp],
    "pmh_gluing_mult_transpose": [vp, vp, vp],
    "pmh_gluing_mult_add": [vp, vp, vp, vp],
    "pmh_gluing_mult_transpose_add": [vp, vp, vp, vp],
    "pmh_extension_create": [vp, C.c_int, C.c_int, vp, vp, vp, C.POINTER(vp)],
    "pmh_extension_destroy": [vp],
    "pmh_extension_mult": [vp, vp, vp],
    "pmh_extension_mult_transpose": [vp, vp, vp],
    "pmh_extension_mult_add": [vp, vp, vp, vp],
    "pmh_extension_mult_transpose_add": [vp, vp, vp, vp],
    "pmh_blockdiag_create": [vp, C.c_int, vp, vp, C.POINTER(vp)],
    "pmh_blockdiag_destroy": [vp],
    "pmh_blockdiag_mult": [vp, vp, vp],
    "pmh_blockdiag_mult_transpose": [vp, vp, vp],
    "pmh_blockdiag_mult_add": [vp, vp, vp, vp],
    "pmh_blockdiag_mult_transpose_add": [vp, vp, vp, vp],
    "pmh_blockdiag_enable_bsr3": [vp, C.c_int],
    "pmh_blockdiag_timing_enable": [vp, C.c_int],
    "pmh_blockdiag_timing_get": [vp, c_int_p, c_double_p, c_double_p, c_double_p, c_int_p],
    "pmh_matinv_create": [vp, C.c_double, C.c_double, C.c_int, C.c_int, C.POINTER(vp)],
    "pmh_matinv_destroy": [vp],
    "pmh_matinv_set_nullspace": [vp, C.c_int, vp],
    "pmh_matinv_set_left_inverse": [vp, C.c_int, vp],
    "pmh_matinv_set_kernel_load_tolerance": [vp, C.c_double],
    "pmh_matinv_mult": [vp, vp, vp],
    "pmh_matinv_last_iterations": [vp, c_int_p, C.POINTER(C.c_longlong)],
    "pmh_mat_regularize_pivots": [C.c_int, C.c_int, vp, vp],
    "pmh_mat_regularization_Q": [C.c_int, C.c_int, vp, vp, vp, vp],
    "pmh_mat_regularize_csr": [C.c_int, vp, vp, vp, C.c_int, vp, C.c_double, vp, vp, vp, vp, C.POINTER(C.c_longlong)],
    "pmh_feti_gluing_from_l2g": [C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, vp, c_int_p, c_int_p, vp, vp, vp],
    "pmh_op_create_feti_dual": [vp, vp, C.POINTER(vp)],
    "pmh_pc_dual_lumped_apply": [vp, vp, vp, vp],
    "pmh_qpt_feti_chain_create": [vp, vp, vp, vp, vp, vp, vp, C.POINTER(vp)],
    "pmh_qpt_feti_chain_get": [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)],
    "pmh_qpt_feti_chain_post_solve": [vp, vp, vp, vp, vp],
    "pmh_qpt_feti_chain_destroy": [vp],
    "pmh_qpt_feti_chain_kkt": [vp, vp, vp, vp, vp, vp],
    "pmh_op_create_svm_dual": [vp, C.c_int, C.c_int, vp, vp, C.POINTER(vp)],
    "pmh_op_svm_dual_passes": [vp, C.POINTER(C.c_longlong)],
    "pmh_smalxe_default_opts": [C.POINTER(SmalxeOpts)],
    "pmh_smalxe_create": [vp, vp, vp, vp, vp, vp, vp, C.POINTER(SmalxeOpts), C.POINTER(vp)],
    "pmh_smalxe_destroy": [vp],
    "pmh_smalxe_solve": [vp],
    "pmh_smalxe_set_reuse_products": [vp, C.c_int],
    "pmh_smalxe_get_stats": [vp, C.POINTER(SmalxeStats)],
    "pmh_smalxe_get_inner": [vp, C.POINTER(vp)],
    "pmh_smalxe_run_fixed": [vp, C.c_int, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p],
    "pmh_smalxe_reset": [vp],
    "pmh_smalxe_set_inner_max_it": [vp, C.c_int],
    "pmh_smalxe_get_inner_max_it": [vp, c_int_p],
    "pmh_smalxe_get_solution": [vp, C.POINTER(vp), C.POINTER(vp), c_int_p],
    "pmh_smalxe_get_penalized": [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)],
    "pmh_pcpg_solve": [vp, vp, vp, vp, vp, vp, C.c_double, C.c_double, C.c_double, C.c_int, C.POINTER(PcpgStats)],
    "pmh_mg_create": [vp, C.c_int, vp, vp, C.c_int, vp, C.c_double, C.c_double, C.c_int, vp, vp, C.c_int, C.POINTER(vp)],
    "pmh_mg_create_box": [vp, vp, C.c_int, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)],
    "pmh_sa_aggregate": [C.c_int, C.c_int, vp, vp, vp, C.c_double, vp, C.POINTER(C.c_int)],
    "pmh_sa_hierarchy_host": [C.c_int, C.c_int, vp, vp, vp, C.c_int, vp, C.c_int, C.c_double, C.POINTER(C.c_int), vp, vp],
    "pmh_mg_create_sa": [vp, vp, C.c_int, vp, C.c_int, vp, vp, vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_double, C.c_int, C.c_int, C.POINTER(vp)],
    "pmh_mg_timing_enable": [vp, C.c_int],
    "pmh_mg_timing_get": [vp, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)],
    "pmh_matinv_enable_bsr3": [vp],
    "pmh_matinv_bsr3_replicas": [vp, c_int_p],
    "pmh_mv_test_spmv": [vp, C.c_int, vp, vp, C.c_int, C.POINTER(C.c_float)],
    "pmh_matinv_mult_multi": [vp, vp, vp, c_int_p],
    "pmh_matinv_multi_rhs_active": [vp, c_int_p],
    "pmh_fexplicit_assemble_auto": [vp, vp, vp, vp, C.c_double, C.c_int, c_int_p],
    "pmh_matinv_timing_enable": [vp, C.c_int],
    "pmh_matinv_timing_get": [vp, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)],
    "pmh_mg_apply": [vp, vp, vp],
    "pmh_mg_stats": [vp, C.POINTER(C.c_longlong)],
    "pmh_mg_destroy": [vp],
    "pmh_matinv_set_pc_mg": [vp, vp],
    "pmh_qps_default_opts": [C.POINTER(QpsOpts)],
    "pmh_qps_set_from_options": [C.c_char_p, C.c_char_p, C.POINTER(QpsOpts), C.POINTER(MpgpOpts), C.POINTER(SmalxeOpts), C.c_char_p, C.c_int],
    "pmh_qpt_matis_split_rhs": [C.c_int, vp, C.c_int, vp, vp],
    "pmh_qpt_matis_to_blockdiag": [C.c_int, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, c_int_p, vp],
    "pmh_qpt_matis_assemble_solution": [C.c_int, vp, vp, C.c_int, vp],
    "pmh_kspfeti_default_opts": [C.POINTER(KspFetiOpts)],
    "pmh_kspfeti_set_from_options": [C.c_char_p, C.POINTER(KspFetiOpts), C.c_char_p, C.c_int],
    "pmh_kspfeti_solve": [vp, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int, vp, C.c_int, vp, C.POINTER(KspFetiOpts), vp, vp, C.c_int, C.POINTER(KspFetiStats)],
    "pmh_fexplicit_create": [vp, vp, C.c_int, C.POINTER(vp)],
    "pmh_fexplicit_create_shared": [vp, vp, vp, C.POINTER(vp)],
    "pmh_fexplicit_create_shared_sym": [vp, vp, vp, C.POINTER(vp)],
    "pmh_fexplicit_create_shared_orbit": [vp, vp, vp, C.POINTER(vp)],
    "pmh_fexplicit_create_shared_orbit_union": [vp, vp, vp, vp, vp, C.POINTER(vp)],
    "pmh_fexplicit_apply_flops": [vp, c_double_p],
    "pmh_fexplicit_apply_flops_detail": [vp, c_double_p, c_double_p],
    "pmh_fexplicit_class_sym_plan": [C.c_int, C.c_int, vp, vp],
    "pmh_fexplicit_orbit_row_tile": [C.c_int, vp, vp],
    "pmh_box_symmetries": [vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp],
    "pmh_box_symmetry_closure": [vp, C.c_int, vp, vp, vp, C.c_int, vp, c_int_p, vp, c_int_p],
    "pmh_fexplicit_set_box_symmetry": [vp, C.c_int, vp, C.c_int, vp, vp, vp, vp],
    "pmh_fexplicit_class_union": [vp, C.c_int, vp, vp],
    "pmh_fexplicit_set_class_symmetry": [vp, C.c_int, C.c_int, vp, vp],
    "pmh_fexplicit_destroy": [vp],
    "pmh_fexplicit_sizes": [vp, c_int_p, vp, C.POINTER(C.c_longlong), c_double_p],
    "pmh_fexplicit_set_stripe": [vp, C.c_int, C.c_int],
    "pmh_fexplicit_stripe_owner": [C.c_int, vp, C.c_int, vp],
    "pmh_fexplicit_stripe_bytes": [C.c_int, vp, C.c_int, c_double_p],
    "pmh_fexplicit_assemble": [vp, vp, C.c_int, vp, vp, C.c_double, C.c_int],
    "pmh_fexplicit_fill_pattern": [vp, C.c_int],
    "pmh_fexplicit_assemble_stats": [vp, C.POINTER(C.c_longlong), c_double_p],
    "pmh_fexplicit_get_block": [vp, C.c_int, vp, vp],
    "pmh_fexplicit_mult": [vp, vp, vp],
    "pmh_fexplicit_compressed_size": [vp, c_int_p, vp],
    "pmh_fexplicit_dense_mult": [vp, vp, vp],
    "pmh_fexplicit_timing_enable": [vp, C.c_int, C.c_int],
    "pmh_fexplicit_timing_get": [vp, c_int_p, c_double_p, c_double_p],
    "pmh_matinv_attach_explicit": [vp, vp],
    "pmh_matinv_set_tolerances": [vp, C.c_double, C.c_double, C.c_int],
    "pmh_matinv_get_tolerances": [vp, c_double_p, c_double_p, c_int_p],
    "pmh_csr_block_classes": [C.c_int, vp, vp, vp, vp, vp, c_int_p],
    "pmh_feti_contact_default_opts": [C.POINTER(FetiContactOpts)],
    "pmh_feti_contact_solve": [vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, C.c_int, C.POINTER(FetiContactOpts), vp, vp, C.POINTER(FetiContactStats)],
    "pmh_ksp_cg_solve": [vp, vp, vp, vp, vp, C.c_double, C.c_double, C.c_double, C.c_int, C.POINTER(PcpgStats)],
}

#: every symbol include/permon_hip.h declares
EXPORTED = sorted(list(_PROTOS) + ["pmh_last_error", "pmh_stream"])

_lib = None


def load(strict=True):
    """Load libpermonhip.so (built by __graft_entry__.build() / make -C permon_amd/csrc)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PermonHipError("libpermonhip.so is missing (%s): build it with `python -c 'import __graft_entry__ as g; g.build()'`; "
                             "permon_amd has no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    L.pmh_last_error.restype = C.c_char_p
    L.pmh_stream.restype = vp
    L.pmh_stream.argtypes = [vp]
    missing = [name for name in _PROTOS if not hasattr(L, name)]
    if missing and strict:
        raise PermonHipError("libpermonhip.so does not export: %s (stale build?)" % ", ".join(missing))
    for name, args in _PROTOS.items():
        if name in missing:
            continue
        f = getattr(L, name)
        f.argtypes = args
        f.restype = C.c_int
    _lib = L
    return L


def check(rc):
    if rc != 0:
        ex = PermonHipError("libpermonhip error %d: %s" % (rc, load().pmh_last_error().decode(errors="replace")))
        ex.code = int(rc)
        raise ex
