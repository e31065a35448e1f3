// Handle structs of the FETI Mats shared by feti.hip and fexplicit.hip (internal; public ABI: include/permon_hip.h)
#pragma once
#include <vector>

#include "pmh_internal.h"

struct pmh_fexplicit_s;

// ---- MATGLUING -----------------------------------------------------------------------------------------------------
struct pmh_gluing_s {
  pmh_ctx ctx;
  int     n_x, n_lambda, n_leaves;
  pmh_csr B, Bt;
  double *d_tmp; // n_lambda scratch of mult_transpose_add on several GPUs (lazy)
  std::vector<int>    h_row, h_root; // host copy of the leaves (set-up consumers: the explicit local dual operators)
  std::vector<double> h_sign;
};

// ---- MATBLOCKDIAG ---------------------------------------------------------------------------------------------------
struct pmh_blockdiag_s {
  pmh_ctx          ctx;
  int              nblocks, n;
  std::vector<int> rowstart;
  int             *d_rowstart;
  pmh_csr          K;
  pmh_bsr3         Kb = nullptr; // optional 3x3-block device copy for MatMult_BlockDiag itself (pmh_blockdiag_enable_bsr3)
};

// ---- MATINV: block-wise CG -------------------------------------------------------------------------------------------
struct pmh_matinv_s {
  pmh_blockdiag K;
  pmh_ctx       ctx;
  int           n, nblocks, wgs; // wgs = workgroups per block in the segmented kernels
  double        rtol, atol;
  int           max_it, jacobi;
  double       *dinv, *r, *z, *p, *Ap;
  double       *d_part, *d_partB; // [2][nblocks*wgs] each
  double       *d_bs;             // [2 parities][nblocks][rz, tol]
  int          *d_bi;             // [2 parities][nblocks][active, its]
  int          *d_nactive, *d_done, *h_nactive;
  int           last_max_its;
  long long     total_spmv;
  // Moore-Penrose variant (QPTDualize true_mp path, qptransform.c:1006-1062): K^+ := P_R K^- P_R with
  // P_R = I - R R', R = block-wise orthonormal kernel basis stored as kdim columns of length n
  int     kdim;
  double *d_R, *d_coef, *d_fproj, *d_kpart;
  double *d_fnorm2 = nullptr; // ||f_b||^2 of the last right-hand side before its projection (k_cg_init: loads in the kernel)
  double  kernel_tol = 64.0;  // ||P_R f_b|| <= kernel_tol eps ||f_b||: the block's load lies in the kernel, u_b = 0 (pmh_matinv_set_kernel_load_tolerance)
  // left generalised inverse (QPTDualize's -qpt_dualize_Kplus_left, qptransform.c:1006-1062: K^+ := K^- P_R): no projection of the result, and the right-hand side entries of
  // the fixing dofs (the null pivots of the reference's factorisation; identity rows / columns of the K handed in) are zeroed
  int  left = 0, nfix = 0;
  int *d_fix = nullptr;
  pmh_mg  mg; // optional V-cycle preconditioner (pmh_matinv_set_pc_mg); NULL: Jacobi / none
  pmh_bsr3 Kb; // optional 3x3-block copy of K for the CG's own product (pmh_matinv_enable_bsr3)
  pmh_fexplicit_s *E; // optional explicit K^+ on the dofs B touches (pmh_matinv_attach_explicit): F applies through it
};

// explicit local dual operators (fexplicit.hip)
bool pmh_fexplicit_matches(pmh_fexplicit_s *E, pmh_gluing B);
int  pmh_fexplicit_apply(pmh_fexplicit_s *E, const double *lambda, double *y);
int  pmh_fexplicit_stages(pmh_fexplicit_s *E, pmh_csr *gather, double **mid_in, pmh_csr *scatter, const double **mid_out); // the sparse stages around the dense one (FetiDualOp::stages)
int  pmh_fexplicit_mid(pmh_fexplicit_s *E);                                                                                // the dense stage alone: mid_in -> mid_out
