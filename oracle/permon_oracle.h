/*
 * permon_oracle.h -- CPU restatement of PERMON's QPS hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle: a plain-C, single-threaded (optionally OpenMP) restatement of the
 * reference algorithm with the reference's own (unfused) operation order.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call it; the product
 * (permon_amd/, libpermonhip.so) never does.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 * The arithmetic of SpMV / BLAS-1 lives in PETSc (un-vendored third party, >= 3.17, see
 * include/permonpetscretro.h:5); it is restated here from its published semantics
 * (MatMult_SeqAIJ: row loop, left-to-right sum from 0.0; VecDot/VecNorm/VecAXPY/...).
 * Pinned against the reference's golden outputs src/tutorials/output/{ex1_*,ex2_*,ex3_*,jbearing2_*}.out
 * (see tests/golden/ and tests/test_oracle_golden.py).
 */
#ifndef PERMON_ORACLE_H
#define PERMON_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* KSPConvergedReason values used by the reference (PETSc petscksp.h) */
#define ORC_CONVERGED_ITERATING 0
#define ORC_CONVERGED_RTOL 2
#define ORC_CONVERGED_ATOL 3
#define ORC_CONVERGED_ITS 4
#define ORC_CONVERGED_HAPPY_BREAKDOWN 7
#define ORC_DIVERGED_ITS (-3)
#define ORC_DIVERGED_DTOL (-4)
#define ORC_DIVERGED_BREAKDOWN (-5)
#define ORC_DIVERGED_NANORINF (-9)

#define ORC_DECIDE (-1.0) /* PETSC_DECIDE */

/* generic linear operator y = A x (callback so Python / composed operators can be plugged in) */
typedef void (*orc_mult_fn)(void *ctx, const double *x, double *y);
typedef struct {
  orc_mult_fn mult;
  void       *ctx;
  int         n; /* square, n x n */
} orc_op;

/* CSR matrix, int32 indices, fp64 values (PETSc SeqAIJ layout) */
typedef struct {
  int           nrows, ncols;
  const int    *rowptr;
  const int    *col;
  const double *val;
} orc_csr;

void orc_csr_mult(void *csr, const double *x, double *y);           /* MatMult_SeqAIJ */
void orc_csr_mult_transpose(void *csr, const double *x, double *y); /* MatMultTranspose_SeqAIJ */

/* box constraints: src/qpc/impls/box/qpcboximpl.h:5-10 + qpc->is, qpc->astol (qpcimpl.h:27-34) */
typedef struct {
  int           n;   /* full vector length */
  int           nis; /* length of lb/ub (== n when is == NULL) */
  const int    *is;  /* optional index set selecting the constrained sub-vector */
  const double *lb;  /* may be NULL */
  const double *ub;  /* may be NULL */
  double        astol;
} orc_box;

void   orc_box_project(const orc_box *qpc, const double *x, double *Px);
double orc_box_feas(const orc_box *qpc, const double *x, const double *d);
void   orc_box_grads(const orc_box *qpc, const double *x, const double *g, double *gf, double *gc);
void   orc_box_gradreduced(const orc_box *qpc, const double *x, const double *gf, double alpha, double *gr);

/* MatGetMaxEigenvalue, src/mat/interface/permonmatutils.c:442-522 */
double orc_max_eigenvalue(const orc_op *A, double tol, int maxits, int *its_out);

enum { ORC_EXP_STD = 0, ORC_EXP_PROJCG, ORC_EXP_GF, ORC_EXP_G, ORC_EXP_GFGR, ORC_EXP_GGR };
enum { ORC_EXPLEN_FIXED = 0, ORC_EXPLEN_OPT, ORC_EXPLEN_OPTAPPROX, ORC_EXPLEN_BB };

struct orc_qps_s;
typedef void (*orc_converged_fn)(struct orc_qps_s *qps, void *ctx);

/* QPS + QPS_MPGP state: include/permon/private/qpsimpl.h:26-70, src/qps/impls/mpgp/mpgpimpl.h:5-38 */
typedef struct orc_qps_s {
  /* problem */
  const orc_op  *A;
  const double  *b;
  double        *x;
  const orc_box *qpc;
  /* QPS tolerances (qps.c:73-76) */
  double rtol, atol, divtol;
  int    max_it;
  /* MPGP parameters (mpgp.c:827-843) */
  double alpha_user;
  int    alpha_direct;
  double gamma;
  double maxeig, maxeig_tol;
  int    maxeig_iter;
  double bchop_tol;
  int    exptype, explengthtype;
  int    resetalpha, fallback, fallback2;
  /* convergence test (default: QPSConvergedDefault) */
  orc_converged_fn converged;
  void            *converged_ctx;
  /* QPSConvergedDefaultCtx (qpsimpl.h:73-76) */
  double norm_rhs, ttol, norm_rhs_div;
  int    cvg_setup_called;
  /* state / results */
  int    setupcalled;
  double alpha;
  double rnorm, gfnorm, gcnorm;
  int    iteration, reason;
  int    nmv, ncg, nexp, nprop, nfinc, nfall;
  char   currentStepType;
  int    expproject;
  /* monitor trace (QPSMonitorDefault_MPGP, mpgp.c:21-34); filled when trace_cap > 0 */
  int     trace_cap, trace_len;
  char   *trace_step;
  double *trace_rnorm, *trace_gfnorm, *trace_gcnorm, *trace_alpha;
  /* work vectors */
  double *work[10];
} orc_qps;

void orc_qps_init(orc_qps *qps); /* defaults of QPSCreate + QPSCreate_MPGP */
int  orc_mpgp_setup(orc_qps *qps);
int  orc_mpgp_solve(orc_qps *qps);
void orc_qps_free(orc_qps *qps);
void orc_converged_default(orc_qps *qps, void *ctx);

/* QPComputeObjective (qp.c:913-927), QPComputeObjectiveFromGradient (qp.c:981-996) */
double orc_objective(const orc_op *A, const double *b, const double *x, double *work);
double orc_objective_from_gradient(int n, const double *b, const double *x, const double *g);

/* ---- projector Q = G'(GG')^{-1}G, P = I - Q (src/qppf/interface/qppf.c:454-645) ----------------- */
typedef struct {
  int           m, n;      /* G is m x n */
  const orc_csr *G;        /* explicit G */
  const double *GGt_chol;  /* m x m lower Cholesky factor of G G' (row-major), NULL if G has orthonormal rows */
  double       *G_left, *Gt_right; /* length m work */
} orc_qppf;

void orc_qppf_apply_Q(const orc_qppf *pf, const double *v, double *Qv);
void orc_qppf_apply_P(const orc_qppf *pf, const double *v, double *Pv);
void orc_qppf_apply_GtG(const orc_qppf *pf, int orthonormal, const double *v, double *y);
void orc_qppf_apply_halfQ(const orc_qppf *pf, const double *x, double *y);           /* y = (GG')^{-1} G x */
void orc_qppf_apply_halfQ_transpose(const orc_qppf *pf, const double *x, double *y); /* y = G'(GG')^{-1} x */
int  orc_dense_cholesky(int m, double *a); /* in place lower factor, row-major; returns 0 on success */
void orc_dense_chol_solve(int m, const double *l, const double *rhs, double *sol);

/* MatMult_Penalized y = rho*(BtB x) + A x  (src/qp/utils/matpenalized.c:12-22) */
typedef struct {
  const orc_op   *A;
  const orc_qppf *pf;
  int             orthonormal; /* pf->G_has_orthonormal_rows_{ex,im}plicitly => BtB == Q */
  double          rho;
} orc_penalized;
void orc_penalized_mult(void *ctx, const double *x, double *y);

/* composite P*A*P (qptransform.c:278-283) and P*A (:273-277) */
typedef struct {
  const orc_op   *A;
  const orc_qppf *pf;
  double         *w1, *w2;
} orc_pap;
void orc_pap_mult(void *ctx, const double *x, double *y);
void orc_pa_mult(void *ctx, const double *x, double *y);

/* QPS_SMALXE: src/qps/impls/smalxe/smalxeimpl.h:13-67 */
typedef struct {
  /* outer problem: min 1/2 u'Au - u'b  s.t. G u = 0, box */
  const orc_op   *A;
  const double   *b;
  double         *u;
  const orc_box  *qpc;
  const orc_qppf *pf;
  int             G_orthonormal;
  /* outer tolerances */
  double rtol, atol, divtol;
  int    max_it;
  /* parameters (smalxe.c:1159-1207) */
  double M1_user;
  int    M1_direct;
  double M1_update;
  double rtol_E;
  double rho_user;
  int    rho_direct;
  double rho_update, rho_update_late;
  double eta_user;
  int    eta_direct;
  double update_threshold;
  double maxeig, maxeig_tol;
  int    maxeig_iter;
  int    inject_maxeig, inject_maxeig_set;
  int    inner_iter_min, inner_no_gtol_stop;
  int    inner_max_it;
  /* ||Bu|| update variant (smalxe.c:878-886): 0 = _SMALXE (:247-261, BE has a mult slot), 1 = _SMALXEON (:265-285, only B'B),
     2 = _Lag_SMALXEON (:289-370, -qps_smalxe_norm_update_lag); lag parameters :1190-1200 (norm_update_lag_offset is zero-initialised) */
  int    norm_update, lag_offset, Jstart, Jstep, Jend;
  double lower, upper;
  int    knoll; /* -qps_smalxe_knoll smalxe.c:938-943 */
  /* the function-static state of the lagged update (:291-293) */
  double lag_normBu0;
  int    lag_II, lag_J, lag_neval, lag_niter;
  /* state / results */
  double M1, M1_initial, eta, rho;
  int    M1_updates, M1_hits, eta_hits, rho_updates;
  int    state, inner_iter_accu;
  double normBu, normBu_old, enorm;
  double rnorm;
  int    iteration, reason;
  orc_qps inner;
  /* private */
  orc_penalized pen;
  orc_op        A_inner;
  double       *Btmu, *b_inner, *BtBu, *Bu, *xwork;
  double        gtol, ttol_outer, norm_rhs_outer, MNormBu;
  double        outer_norm_rhs, outer_ttol, outer_norm_rhs_div;
  int           outer_cvg_setup;
} orc_smalxe;

void orc_smalxe_init(orc_smalxe *s);
int  orc_smalxe_setup(orc_smalxe *s);
int  orc_smalxe_solve(orc_smalxe *s);
void orc_smalxe_free(orc_smalxe *s);

/* QPSSolve_PCPG  src/qps/impls/pcpg/pcpg.c:51-134 */
typedef void (*orc_pc_fn)(void *ctx, const double *x, double *y);
typedef struct {
  const orc_op   *A;
  const double   *b;
  double         *x;
  const orc_qppf *pf;
  orc_pc_fn       pc; /* NULL => none */
  void           *pc_ctx;
  double          rtol, atol, divtol;
  int             max_it;
  double          rnorm;
  int             iteration, reason;
} orc_pcpg;
int orc_pcpg_solve(orc_pcpg *s);

/* MatMult_Gluing / MatMultTranspose_Gluing, src/mat/impls/gluing/gluing.c:47-81,125-159
   (single-process restatement: SF root of leaf i = leaves_root[i]) */
typedef struct {
  int           n_x;      /* primal length */
  int           n_lambda; /* dual length */
  int           n_leaves;
  const int    *leaves_row;  /* primal dof of leaf i */
  const int    *leaves_root; /* lambda index of leaf i */
  const double *leaves_sign;
} orc_gluing;
void orc_gluing_mult(const orc_gluing *B, const double *lambda, double *x);           /* x = B' lambda */
void orc_gluing_mult_transpose(const orc_gluing *B, const double *x, double *lambda); /* lambda = B x */

/* MATINV apply on the reference's iterative path (KSPCG + PCJACOBI per block, matinv.c:535-540) wrapped as the
   Moore-Penrose inverse P_R K^- P_R (QPTDualize -qpt_dualize_Kplus_mp, qptransform.c:1006-1062); blocks = contiguous
   row ranges of the block-diagonal CSR K; R = kdim columns of length n (column-major), block-wise orthonormal */
typedef struct {
  const orc_csr *K;
  int            nblocks;
  const int     *rowstart;
  int            kdim;
  const double  *R;
  double         rtol, atol;
  int            max_it;
  long long      spmv_count;
  int            last_max_its;
  double         kernel_tol; /* 0 (default): the plain KSPCG of the reference's iterative MATINV.  c > 0: a block with ||P_R f_b|| <= c eps ||f_b|| is given u_b = 0 -- the
                                product's rule for loads in the kernel (pmh_matinv_set_kernel_load_tolerance); tests switch it on explicitly and check it against a dense pinv */
} orc_matinv;
void orc_matinv_mult(orc_matinv *M, const double *f, double *u);

/* F = B K^+ B' (qptransform.c:1103-1128) and the SMALXE inner operator A_rho = P F P + rho Q as native callbacks */
typedef struct {
  const orc_gluing *B;
  orc_matinv       *Kplus;
  const orc_qppf   *pf;
  double            rho;
  double           *t1, *t2, *w1, *w2; /* work: n_x, n_x, n_lambda, n_lambda */
} orc_feti;
void orc_feti_dual_mult(void *ctx, const double *x, double *y);     /* y = F x */
void orc_feti_penalized_mult(void *ctx, const double *x, double *y); /* y = P F P x + rho Q x */

/* MatRegularize (src/mat/interface/permonmatregularize.c): fixing-DOF regularisation of a singular block, see the .c file */
void orc_regularize_pivots(int p, int d, const double *R /* p x d column-major */, int *pivots /* d, ascending */);
int  orc_regularization_Q(int p, int d, const double *R, const int *pivots, double *Q /* d x d */, int *keep /* d x d */);
int  orc_regularize_csr(const orc_csr *K, int d, const double *R, double rho, int *pivots, int *rowptr_out, int *col_out, double *val_out);

/* unfused reference-order CG step timing helper for the CPU baseline: runs `iters` MPGP iterations
   without convergence test on a fixed problem; returns elapsed seconds */
double orc_now(void);

#ifdef __cplusplus
}
#endif
#endif
