// Handle structs of the FETI Mats shared by feti.hip and fexplicit.hip (internal; public ABI: include/permon_hip.h)
#pragma once
#include <algorithm>
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
  // 8 congruent blocks with a fused fp32 V-cycle: pmh_matinv_mult runs them as the 8 columns of ONE block on the multi-right-hand-side kernels (matinv_mv.hip; knob "kplus_mv").
  // mvc_state: 0 not tried yet, 1 in use, -1 does not apply; reset by whatever changes the solver (V-cycle, kernel, left inverse, block copy)
  struct pmh_matinv_mv_s *mvc = nullptr;
  int                     mvc_state = 0;
};

// explicit local dual operators (fexplicit.hip)
bool pmh_fexplicit_matches(pmh_fexplicit_s *E, pmh_gluing B);
int  pmh_fexplicit_apply(pmh_fexplicit_s *E, const double *lambda, double *y);
int  pmh_fexplicit_stages(pmh_fexplicit_s *E, pmh_csr *gather, double **mid_in, pmh_csr *scatter, const double **mid_out); // the sparse stages around the dense one (FetiDualOp::stages)
int  pmh_fexplicit_mid(pmh_fexplicit_s *E);                                                                                // the dense stage alone: mid_in -> mid_out

// ---- what the set-up loops of the explicit operators (fexplicit.hip, fshared.hip) need from a K^+ solver: one COLUMN per slot ----------------------------------
// nslots == solver->nblocks: slot s = block s of the one-column solver (pmh_matinv_mult).  nslots == PMH_MV_R * solver->nblocks: the multi-right-hand-side solver
// (matinv_mv.hip), slot s = column s % R of block s / R; its interleaved result is laid out column by column before the rows are extracted.
#include "mv_internal.h"
struct pmh_asm_solver {
  pmh_matinv    M = nullptr;
  pmh_matinv_mv V = nullptr;
  double       *umv = nullptr; // mv: the interleaved result (the caller's `sol` receives the columns one after the other)
  int open(pmh_matinv solver, int nslots)
  {
    M = solver;
    if (nslots == solver->nblocks) return PMH_SUCCESS;
    if (nslots != solver->nblocks * PMH_MV_R) return pmh_set_error(PMH_ERR_ARG, "pmh_fexplicit_assemble: %d slots for a solver of %d blocks (one per block, or %d per block)", nslots, solver->nblocks, PMH_MV_R);
    int rc = pmh_matinv_mv_create(solver, &V);
    if (rc == PMH_EPI_UNSUPPORTED)
      return pmh_set_error(PMH_ERR_SUP, "pmh_fexplicit_assemble: the multi-right-hand-side K^+ does not apply to this solver: %s", pmh_mv_why());
    if (rc) return rc;
    return pmh_malloc(M->ctx, sizeof(double) * len(), (void **)&umv);
  }
  void close()
  {
    if (V) pmh_matinv_mv_destroy(V), V = nullptr;
    if (umv && M) pmh_free(M->ctx, umv), umv = nullptr;
  }
  int    R() const { return V ? PMH_MV_R : 1; }
  size_t len() const { return (size_t)std::max(1, M->n) * R(); }
  int    rows(int s) const { return M->K->rowstart[s / R() + 1] - M->K->rowstart[s / R()]; }
  int    rhs_index(int s, int rel) const { return V ? (M->K->rowstart[s / PMH_MV_R] + rel) * PMH_MV_R + s % PMH_MV_R : M->K->rowstart[s] + rel; }
  const double *sol_of(const double *sol, int s) const { return V ? sol + (size_t)(s % PMH_MV_R) * M->n + M->K->rowstart[s / PMH_MV_R] : sol + M->K->rowstart[s]; }
  int solve(const double *rhs, double *sol)
  {
    if (!V) return pmh_matinv_mult(M, rhs, sol);
    PMH_CHK(pmh_matinv_mv_mult(V, rhs, umv));
    return pmh_matinv_mv_to_columns(V, umv, sol);
  }
  bool hit_the_limit() const { return (V ? pmh_matinv_mv_last_iterations(V) : M->last_max_its) >= M->max_it; }
};
