/*
 * permon_hip.h -- C ABI of libpermonhip: an MI355X (gfx950) implementation of PERMON's QPS hot path.
 *
 * Drop-in boundary.  PERMON reaches its numerical kernels through three op tables and a handful of
 * Mat implementations; every entry point below names the reference slot it replaces
 * (file:line under /root/reference).  INTEGRATION.md shows the PETSc-side glue that binds them.
 *
 *   - all vector arguments are DEVICE pointers to fp64 (PetscScalar = double, real build);
 *     index arrays handed to *_create functions are HOST pointers (copied to the device once);
 *   - every function returns 0 (PETSC_SUCCESS) or a non-zero pmh error code; pmh_last_error()
 *     returns the message.  Solver failure is NOT an error: it is reason < 0 (qps.c:551);
 *   - no function falls back to a CPU path: without a usable HIP device pmh_init() fails.
 *   - one pmh context per process = one GPU (one MPI rank <-> one GPU, matblockdiag.c:787-788).
 */
#ifndef PERMON_HIP_H
#define PERMON_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- errors ------------------------------------------------------------------------------------ */
#define PMH_SUCCESS 0
#define PMH_ERR_HIP 1       /* HIP runtime error (message has the hipError string) */
#define PMH_ERR_ARG 2       /* bad argument (PETSC_ERR_ARG_*) */
#define PMH_ERR_STATE 3     /* wrong state (PETSC_ERR_ARG_WRONGSTATE / PETSC_ERR_ORDER) */
#define PMH_ERR_SUP 4       /* unsupported (PETSC_ERR_SUP) */
#define PMH_ERR_NODEVICE 5  /* no gfx950 device / extension unusable */
#define PMH_ERR_COMM 6      /* RCCL error */
const char *pmh_last_error(void);

/* KSPConvergedReason values the path produces (PETSc petscksp.h; qps.c:675-714, smalxe.c:610-692) */
#define PMH_CONVERGED_ITERATING 0
#define PMH_CONVERGED_RTOL 2
#define PMH_CONVERGED_ATOL 3
#define PMH_CONVERGED_ITS 4
#define PMH_CONVERGED_HAPPY_BREAKDOWN 7
#define PMH_DIVERGED_ITS (-3)
#define PMH_DIVERGED_DTOL (-4)
#define PMH_DIVERGED_BREAKDOWN (-5)
#define PMH_DIVERGED_NANORINF (-9)
#define PMH_DECIDE (-1.0) /* PETSC_DECIDE */

/* ---- context, memory, multi-GPU communicator ----------------------------------------------------- */
typedef struct pmh_ctx_s *pmh_ctx;

int pmh_init(int device, pmh_ctx *ctx);     /* PermonInitialize's device part; fails loudly without a GPU */
/* Process-wide run-time switches (the role of PETSc's options database for the library's own A/B switches; the environment variable PMH_<NAME> gives the initial value).
 * "chain" (PMH_NO_CHAIN unset = 1): the penalised, projected FETI operator as the five-launch dual-space chain (dualchain.hip); 0 = the round-4 launch sequence.
 * Takes effect for operators created afterwards.  "chain_applies" / "chain_launches": counters of the chain's applications and of its own kernel launches (the middle
 * stage's -- GEMM + finishing launch, or the inner Krylov solve -- not included); set to reset.  "host_threads": threads of the host-side set-up builders (3x3-block
 * copies, multigrid hierarchy, class detection); initial value PMH_HOST_THREADS, else OMP_NUM_THREADS, else min(16, the CPUs the process may run on) -- with several ranks
 * per node every rank must get its share.  "svm_pairing" (PMH_SVM_NO_PAIRING unset = 1): the SVM dual's paired passes over X inside MPGP; 0 = every Hessian application as its
 * own two passes (may change between two solves; every rank of a job must set it alike).  "gt_fusion" (PMH_NO_GT_FUSION), "smalxe_prefetch" (PMH_SMALXE_NO_PREFETCH),
 * "mg_d0_fusion" (PMH_MG_NO_D0_FUSION): A/B switches of fusions on per-product paths, 1 = fused (the default).  Unknown name: PMH_ERR_ARG. */
int pmh_set_knob(const char *name, int value);
int pmh_get_knob(const char *name, int *value);
int pmh_finalize(pmh_ctx ctx);
int pmh_mem_info(pmh_ctx ctx, size_t *free_bytes, size_t *total_bytes); /* HBM of the context's device */
int pmh_device_name(pmh_ctx ctx, char *buf, size_t len);
int pmh_sync(pmh_ctx ctx);                  /* hipStreamSynchronize of the context's compute stream */
void *pmh_stream(pmh_ctx ctx);              /* the hipStream_t kernels are launched on */

int pmh_malloc(pmh_ctx ctx, size_t bytes, void **dptr);
int pmh_free(pmh_ctx ctx, void *dptr);
int pmh_memcpy_h2d(pmh_ctx ctx, void *dst, const void *src, size_t bytes);
int pmh_memcpy_d2h(pmh_ctx ctx, void *dst, const void *src, size_t bytes);
int pmh_memcpy_d2d(pmh_ctx ctx, void *dst, const void *src, size_t bytes);
int pmh_memset(pmh_ctx ctx, void *dst, int value, size_t bytes);

/* timing on the compute stream with HIP events (bench.py's roofline leg) */
int pmh_timer_start(pmh_ctx ctx);
int pmh_timer_stop(pmh_ctx ctx, double *milliseconds);

/* RCCL communicator over xGMI: replaces the MPI_Allreduce / PetscSF / VecScatter call sites of SURVEY 2.4 */
#define PMH_UNIQUE_ID_BYTES 128
int pmh_comm_unique_id(void *id128);                                     /* rank 0, then broadcast out of band */
int pmh_comm_init(pmh_ctx ctx, int rank, int size, const void *id128);  /* collective */
int pmh_comm_rank(pmh_ctx ctx, int *rank, int *size);
int pmh_comm_allreduce_sum(pmh_ctx ctx, double *dbuf, size_t count);    /* in place, device buffer */
int pmh_comm_allreduce_min(pmh_ctx ctx, double *dbuf, size_t count);
int pmh_comm_barrier(pmh_ctx ctx);
/* HIP-event pairs (on the launch stream) around the next max_events vector all-reduces of the data path -- the sum of the ranks' B u that ends MatMultTranspose_Gluing
 * (PetscSFReduce, gluing.c:144-147); 0 switches the timing off.  _get: how many were timed, their total milliseconds and bytes; resets the counters. */
int pmh_comm_timing_enable(pmh_ctx ctx, int max_events);
int pmh_comm_timing_get(pmh_ctx ctx, int *count, double *total_ms, double *bytes);
/* Host-staged transport instead of RCCL: every collective of the data path (B u in pmh_gluing_mult_transpose, the SVM w, the grouped MPGP scalars, the barrier) copies its
 * device buffer to pinned host memory and calls fn -- an IN-PLACE all-reduce over the ranks (op: PMH_COMM_SUM / PMH_COMM_MIN; PMH_COMM_BARRIER with count 0), returning 0 on
 * success -- then copies the result back, in stream order.  What the PETSc glue passes when the ranks cannot form an RCCL communicator: MPI_Allreduce(MPI_IN_PLACE, buf, count,
 * MPI_DOUBLE, MPI_SUM / MPI_MIN, comm) (replaces the reference's MPI reductions: gluing.c:144-147, qpc.c:521, VecDot / VecNorm).  fn == NULL removes it.  Not together with pmh_comm_init. */
enum { PMH_COMM_SUM = 0, PMH_COMM_MIN = 1, PMH_COMM_BARRIER = 2 };
typedef int (*pmh_comm_host_fn)(void *user, int op, double *host_buf, size_t count);
int pmh_comm_set_host_transport(pmh_ctx ctx, int rank, int size, pmh_comm_host_fn fn, void *user);

/* ---- Mat: CSR (PETSc SeqAIJ role) ---------------------------------------------------------------- */
/* replaces MatMult(A,..) at mpgp.c:500,537,578,606,624, mpgp.c:250, permonmatutils.c:487,
   matblockdiag.c:197 (local block), extension.c:485,519 (condensed block) */
typedef struct pmh_csr_s *pmh_csr;
int pmh_csr_create(pmh_ctx ctx, int nrows, int ncols, const int *rowptr, const int *col, const double *val, pmh_csr *A);
int pmh_csr_destroy(pmh_csr A);
int pmh_csr_sizes(pmh_csr A, int *nrows, int *ncols, long long *nnz);
int pmh_csr_mult(pmh_csr A, const double *x, double *y);                       /* y = A x       */
int pmh_csr_mult_add(pmh_csr A, const double *x, const double *y1, double *y);  /* y = y1 + A x  */
int pmh_csr_mult_transpose(pmh_csr A, const double *x, double *y);             /* y = A' x      */
int pmh_csr_mult_transpose_add(pmh_csr A, const double *x, const double *y1, double *y); /* y = y1 + A' x */
int pmh_csr_algorithmic_bytes(pmh_csr A, double *bytes);                        /* 12 nnz + 20 nrows */
/* per-launch kernel timing with HIP events on the launch stream (bench.py's roofline leg): every SpMV launch of A
   is bracketed by an event pair while enabled; `epilogue` selects 0 plain, 1 mult-add, 2 fused "-b", 3 fused MPGP phase P1 */
int pmh_csr_timing_enable(pmh_csr A, int max_launches);
int pmh_csr_timing_get(pmh_csr A, int epilogue, int *launches, double *total_ms);

/* ---- generic operator (PETSc Mat with a mult slot) ------------------------------------------------ */
typedef struct pmh_op_s *pmh_op;
typedef int (*pmh_shell_mult_fn)(void *user, const double *x_dev, double *y_dev);
int pmh_op_create_csr(pmh_csr A, pmh_op *op);                           /* borrows A */
int pmh_op_create_shell(pmh_ctx ctx, int n, pmh_shell_mult_fn f, void *user, pmh_op *op); /* MatCreateShellPermon, shell.c:5-31 */
int pmh_op_destroy(pmh_op op);
int pmh_op_size(pmh_op op, int *n);
int pmh_op_mult(pmh_op op, const double *x, double *y);
int pmh_op_mult_transpose(pmh_op op, const double *x, double *y); /* MatMultTranspose; PMH_ERR_SUP for a shell without the slot */
/* MatGetMaxEigenvalue, src/mat/interface/permonmatutils.c:442-522 (tol/maxits: PMH_DECIDE -> 1e-4 / 50) */
int pmh_op_max_eigenvalue(pmh_op op, double tol, int maxits, double *lambda, int *its);

/* ---- QPC box: struct _QPCOps slots (include/permon/private/qpcimpl.h:8-25) ------------------------- */
/* lb / ub may be NULL (no bound of that kind).  n = local length of the constrained (sub)vector. */
int pmh_qpc_box_project(pmh_ctx ctx, int n, const double *x, const double *lb, const double *ub, double *Px);     /* QPCProject_Box qpcbox.c:290-305 */
int pmh_qpc_box_feas(pmh_ctx ctx, int n, const double *x, const double *d, const double *lb, const double *ub, double *alpha_host); /* QPCFeas_Box qpcbox.c:104-146; the MIN all-reduce of qpc.c:521 has no counterpart: constrained vectors are replicated */
int pmh_qpc_box_grads(pmh_ctx ctx, int n, const double *x, const double *g, const double *lb, const double *ub, double astol, double *gf, double *gc); /* QPCGrads qpc.c:540-569 + QPCGrads_Box qpcbox.c:21-64 */
int pmh_qpc_box_gradreduced(pmh_ctx ctx, int n, const double *x, const double *gf, const double *lb, const double *ub, double alpha, double *gr); /* QPCGradReduced qpc.c:589-615 + _Box qpcbox.c:68-100 */
/* expands an index-set restricted bound (qpc->is, qpc.c:416-437) to a full-length bound with -inf/+inf elsewhere */
int pmh_qpc_box_expand_is(pmh_ctx ctx, int n, int nis, const int *is_host, const double *bound_sub, double fill, double *bound_full);

/* ---- post-solve: KKT residuals of a box-constrained QP -------------------------------------------------------- */
/* QPComputeMissingBoxMultipliers (qp.c:828-893: lambda_lb = A x - b, lambda_ub = -(A x - b), both clipped at 0 when both
   bounds exist) + the `r = ...` lines of QPViewKKT (qp.c:245-369) and QPCViewKKT_Box (qpcbox.c:333-427).
   out[0] = ||A x - b - lambda_lb + lambda_ub||, out[1] = ||min(x-lb,0)||, out[2] = ||min(lambda_lb,0)||,
   out[3] = |lambda_lb'(lb-x)|, out[4] = ||max(x-ub,0)||, out[5] = ||min(lambda_ub,0)||, out[6] = |lambda_ub'(x-ub)|,
   out[7] = ||b||  (entries of an absent bound are 0).  work: device scratch of length n. */
int pmh_qp_kkt_box(pmh_op A, const double *b, const double *x, const double *lb, const double *ub, double *work, double out_host[8]);

/* ---- Vec kernels used by the path (PETSc VecAXPY/AYPX/WAXPY/Dot/Norm/Copy/Set/Scale) --------------- */
int pmh_vec_axpy(pmh_ctx ctx, int n, double *y, double a, const double *x);                  /* y += a x */
int pmh_vec_aypx(pmh_ctx ctx, int n, double *y, double a, const double *x);                  /* y = x + a y */
int pmh_vec_waxpy(pmh_ctx ctx, int n, double *w, double a, const double *x, const double *y); /* w = a x + y */
int pmh_vec_scale(pmh_ctx ctx, int n, double *x, double a);
int pmh_vec_set(pmh_ctx ctx, int n, double *x, double a);
int pmh_vec_copy(pmh_ctx ctx, int n, const double *x, double *y);
int pmh_vec_dot(pmh_ctx ctx, int n, const double *x, const double *y, double *result_host);
int pmh_vec_norm2(pmh_ctx ctx, int n, const double *x, double *result_host);

/* ---- QPS MPGP: struct _QPSOps.solve / .setup  (src/qps/impls/mpgp/mpgp.c:359-650) ------------------- */
enum { PMH_EXP_STD = 0, PMH_EXP_PROJCG, PMH_EXP_GF, PMH_EXP_G, PMH_EXP_GFGR, PMH_EXP_GGR };           /* QPSMPGPExpansionTypes, mpgp.c:3 */
enum { PMH_EXPLEN_FIXED = 0, PMH_EXPLEN_OPT, PMH_EXPLEN_OPTAPPROX, PMH_EXPLEN_BB };                    /* QPSMPGPExpansionLengthTypes, mpgp.c:4 */

typedef struct {
  /* QPS tolerances: QPSCreate qps.c:73-76 */
  double rtol, atol, divtol;
  int    max_it;
  /* QPS_MPGP: QPSCreate_MPGP mpgp.c:827-843 */
  double alpha_user;   /* PMH_DECIDE -> 2.0 */
  int    alpha_direct; /* QPS_ARG_DIRECT (1) / QPS_ARG_MULTIPLE (0) */
  double gamma;
  double maxeig;       /* PMH_DECIDE -> power method */
  double maxeig_tol;
  int    maxeig_iter;
  double bchop_tol;
  double astol;        /* qpc->astol, qpc.c:28: 10*eps */
  int    exptype, explengthtype;
  int    resetalpha, fallback, fallback2;
  int    monitor;      /* record the QPSMonitorDefault_MPGP trace (mpgp.c:21-34) */
  int    unfused;      /* 1: run the literal one-kernel-per-PETSc-call sequence (every variant supports it);
                          0: std expansion + fixed length without fallback takes the fused kernels */
  int    distributed;  /* 1: x, b, lb, ub are ROW-DISTRIBUTED over the ranks of the communicator (PETSc MPI Vec layout):
                          every dot / norm / min is completed with an RCCL all-reduce of the device scalars, replacing
                          VecDot's and QPCFeas' MPI_Allreduce.  0: vectors are rank-local or replicated */
} pmh_mpgp_opts;

typedef struct {
  int    iteration, reason;
  double rnorm, gfnorm, gcnorm, alpha, maxeig;
  int    nmv, ncg, nexp, nprop, nfinc, nfall;
  double norm_rhs, ttol;
  char   current_step_type;
} pmh_mpgp_stats;

typedef struct pmh_mpgp_s *pmh_mpgp;
int pmh_mpgp_default_opts(pmh_mpgp_opts *o);
/* QPSSetup_MPGP: work vectors, bound chop, power method, alpha. lb/ub full-length device vectors or NULL */
int pmh_mpgp_create(pmh_ctx ctx, pmh_op A, const double *b, double *x, const double *lb, const double *ub, const pmh_mpgp_opts *o, pmh_mpgp *s);
int pmh_mpgp_destroy(pmh_mpgp s);
int pmh_mpgp_solve(pmh_mpgp s);                                     /* QPSSolve_MPGP mpgp.c:438-650 */
int pmh_mpgp_get_stats(pmh_mpgp s, pmh_mpgp_stats *st);
/* qps->convergencetest (qps.c:675; called at mpgp.c:531 every iteration with rnorm / iteration current).
   The callback sets *reason (0 = keep iterating); NULL restores QPSConvergedDefault.  This is the hook
   SMALXE uses to inject QPSConverged_Inner_SMALXE (smalxe.c:874-875). */
typedef int (*pmh_converged_fn)(void *user, int iteration, double rnorm, int *reason);
int pmh_mpgp_set_convergence_test(pmh_mpgp s, pmh_converged_fn f, void *user);
int pmh_mpgp_set_tolerances(pmh_mpgp s, double rtol, double atol, double divtol, int max_it); /* QPSSetTolerances */
/* monitor trace: per iteration step type, ||gP||, ||gf||, ||gc||, alpha (arrays of length >= iteration+1) */
int pmh_mpgp_get_trace(pmh_mpgp s, int cap, char *step, double *gp, double *gf, double *gc, double *alpha, int *len);
int pmh_mpgp_get_work(pmh_mpgp s, int idx, const double **dptr);   /* work[0..6] = gP,gf,gc,g,p,Ap,gr (mpgp.c:6-17) */
/* composed methods SMALXE needs from its inner solver (mpgp.c:858-869) */
int pmh_mpgp_set_operator_max_eigenvalue(pmh_mpgp s, double maxeig);   /* "QPSMPGPSetOperatorMaxEigenvalue_MPGP_C" mpgp.c:107-115 */
int pmh_mpgp_update_max_eigenvalue(pmh_mpgp s, double maxeig_update);  /* "QPSMPGPUpdateMaxEigenvalue_MPGP_C"      mpgp.c:119-143 */
int pmh_mpgp_get_current_step_type(pmh_mpgp s, char *step);            /* "QPSMPGPGetCurrentStepType_MPGP_C"       mpgp.c:38-45  */
int pmh_mpgp_reset_statistics(pmh_mpgp s);                             /* QPSResetStatistics_MPGP mpgp.c:654-664 */
int pmh_mpgp_get_tolerances(pmh_mpgp s, double *rtol, double *atol, double *divtol, int *max_it); /* QPSGetTolerances; any pointer may be NULL */

/* ---- QPPF projector factory (src/qppf/interface/qppf.c) ----------------------------------------------- */
typedef struct pmh_qppf_s *pmh_qppf;
/* G: m x n CSR (device); GG' is formed and Cholesky-factored on the host once (QPPFSetUpGGt/GGtinv_Private
   qppf.c:213-333, MATINV monolithic direct solve), the factor is applied redundantly on the device.
   orthonormal == 1 <=> G_has_orthonormal_rows (GGtinv = NULL, qppf.c:225-229).
   orthonormal == 2: G is NOT orthonormal and is orthonormalised IMPLICITLY (QPTOrthonormalizeEq with -qp_E_orth_form implicit, the reference's default
   form, qptransform.c:647; MatOrthRows_Implicit_Default permonmatorth.c:176-205): the object behaves exactly as one created with the explicit
   T G (GG' = L L', T = L^{-1}; orthonormal rows) but keeps G as sparse as it came and applies the small dense T / T'T inside the finishing launch
   of G v.  The right-hand side of the constraint transforms with pmh_qppf_orth_rhs (e = T e0, host vectors of length m). */
int pmh_qppf_create(pmh_ctx ctx, pmh_csr G, int orthonormal, pmh_qppf *pf);
int pmh_qppf_orth_rhs(pmh_qppf pf, const double *e0_host, double *e_host);
int pmh_qppf_destroy(pmh_qppf pf);
/* set-up cost of the coarse problem: GG' assembly on the matrix cores (ms, flops = 2 Mp^2 n) and the host Cholesky + inverse (ms) */
int pmh_qppf_setup_stats(pmh_qppf pf, double *ggt_mfma_ms, double *ggt_flops, double *host_inverse_ms);
int pmh_qppf_apply_Q(pmh_qppf pf, const double *v, double *Qv);     /* QPPFApplyQ   qppf.c:454-503 */
int pmh_qppf_apply_P(pmh_qppf pf, const double *v, double *Pv);     /* QPPFApplyP   qppf.c:563-575 */
int pmh_qppf_apply_GtG(pmh_qppf pf, const double *v, double *y);    /* QPPFApplyGtG qppf.c:580-605 */
int pmh_qppf_apply_CP(pmh_qppf pf, const double *x, double *y);     /* QPPFApplyCP  qppf.c:610-645: y = (GG')^{-1} x, length m */
int pmh_qppf_apply_halfQ(pmh_qppf pf, const double *x, double *y);  /* QPPFApplyHalfQ qppf.c:507-527 */
int pmh_qppf_apply_halfQ_transpose(pmh_qppf pf, const double *x, double *y); /* qppf.c:531-559 */
int pmh_qppf_apply_G(pmh_qppf pf, const double *v, double *Gv);     /* MatMult(cp->G,..) qppf.c:475 */

/* ---- composed operators of the QP transform chain ------------------------------------------------------ */
int pmh_op_create_penalized(pmh_op A, pmh_qppf pf, double rho, pmh_op *op);   /* MatCreatePenalized matpenalized.c:212-243; mult :12-22 */
/* the other three op slots MatCreatePenalized registers (matpenalized.c:232-235): MatMultTranspose_Penalized :26-36 (through
   pmh_op_mult_transpose), MatMultAdd_Penalized :40-57 (y = x2 + A_rho x; x2 may be y), MatMultTransposeAdd_Penalized :61-78 */
int pmh_op_penalized_mult_add(pmh_op op, const double *x, const double *x2, double *y);
int pmh_op_penalized_mult_transpose_add(pmh_op op, const double *x, const double *x2, double *y);
int pmh_op_penalized_set_penalty(pmh_op op, double rho);                      /* MatPenalizedSetPenalty */
int pmh_op_penalized_get_penalty(pmh_op op, double *rho);
int pmh_op_create_projected(pmh_op A, pmh_qppf pf, int symmetric, pmh_op *op); /* P*A*P (symmetric) or P*A: qptransform.c:273-284 */

/* ---- FETI Mats ------------------------------------------------------------------------------------------ */
/* MATGLUING (src/mat/impls/gluing/gluing.c): leaves (local primal dof, lambda index, sign).
   mult: x = B' lambda (gluing.c:47-81); mult_transpose: lambda = B x (gluing.c:125-159); in a multi-GPU
   run lambda is replicated and mult_transpose ends with an RCCL all-reduce (replaces PetscSFReduce :144-147). */
typedef struct pmh_gluing_s *pmh_gluing;
int pmh_gluing_create(pmh_ctx ctx, int n_x, int n_lambda, int n_leaves, const int *leaves_row, const int *leaves_root, const double *leaves_sign, pmh_gluing *B);
int pmh_gluing_destroy(pmh_gluing B);
int pmh_gluing_mult(pmh_gluing B, const double *lambda, double *x);
int pmh_gluing_mult_transpose(pmh_gluing B, const double *x, double *lambda);
int pmh_gluing_mult_add(pmh_gluing B, const double *lambda, const double *x1, double *x);                 /* MatMultAdd_Gluing gluing.c:85-123 */
int pmh_gluing_mult_transpose_add(pmh_gluing B, const double *x, const double *lambda1, double *lambda); /* MatMultTransposeAdd_Gluing :163-199 */

/* MATEXTENSION (src/mat/impls/extension/extension.c, the reference's DEFAULT gluing matrix type, qpfeti.c:825-831):
   TA = scatter(ris) * A * gather(cis) with a condensed local CSR A (n_ris x n_cis).
   mult (extension.c:476-489):           r = 0; r[ris] += A * c[cis]
   mult_transpose (extension.c:510-523): c = 0; c[cis] += A' * r[ris]
   ris / cis are host index arrays (each without repeats: they are index sets); n_r, n_c are the lengths of r and c. */
typedef struct pmh_extension_s *pmh_extension;
int pmh_extension_create(pmh_ctx ctx, int n_r, int n_c, pmh_csr A, const int *ris, const int *cis, pmh_extension *TA);
int pmh_extension_destroy(pmh_extension TA);
int pmh_extension_mult(pmh_extension TA, const double *c, double *r);
int pmh_extension_mult_transpose(pmh_extension TA, const double *r, double *c);
int pmh_extension_mult_add(pmh_extension TA, const double *c, const double *r1, double *r);           /* MatMultAdd_Extension extension.c:493-506 */
int pmh_extension_mult_transpose_add(pmh_extension TA, const double *r, const double *c1, double *c); /* MatMultTransposeAdd_Extension :527-540 */

/* MATBLOCKDIAG (src/mat/impls/blockdiag/matblockdiag.c:190-201): the rank's sequential blocks.
   Several subdomains per GPU (BASELINE configs[3]) are stored as ONE concatenated CSR + block offsets. */
typedef struct pmh_blockdiag_s *pmh_blockdiag;
int pmh_blockdiag_create(pmh_ctx ctx, int nblocks, const int *block_rowstart /* nblocks+1 */, pmh_csr Kcat, pmh_blockdiag *K);
int pmh_blockdiag_destroy(pmh_blockdiag K);
int pmh_blockdiag_mult(pmh_blockdiag K, const double *x, double *y);
int pmh_blockdiag_mult_transpose(pmh_blockdiag K, const double *x, double *y);                        /* matblockdiag.c:205-216 */
int pmh_blockdiag_mult_add(pmh_blockdiag K, const double *x, const double *y1, double *y);            /* :220-233, y1 may be y */
int pmh_blockdiag_mult_transpose_add(pmh_blockdiag K, const double *x, const double *y1, double *y);  /* :237-250 */
/* MatMult_BlockDiag on a 3x3-block device copy (the role MATSEQBAIJ bs = 3 plays for 3-dof elasticity blocks: 8.44 instead of 12 bytes per non-zero).
 * share != 0: congruent blocks (checked entry by entry) share ONE device copy -- the product then reads most of K from the XCDs' L2; share == 0: one copy per block,
 * every byte streamed from HBM.  PMH_ERR_SUP if K has no 3x3 block structure.  The transpose / add forms stay on the CSR kernel. */
int pmh_blockdiag_enable_bsr3(pmh_blockdiag K, int share);
/* HIP-event pairs around the launches of pmh_blockdiag_mult.  csr_bytes = 12 nnz + 20 n (SURVEY 8d's figure of the product); hbm_bytes = what the kernel in use has to
 * move from HBM per launch (its stored format; a shared copy once); device_copies = matrix copies on the device (1 when shared, else nblocks). */
int pmh_blockdiag_timing_enable(pmh_blockdiag K, int max_launches);
int pmh_blockdiag_timing_get(pmh_blockdiag K, int *launches, double *total_ms, double *csr_bytes, double *hbm_bytes, int *device_copies);

/* MATINV apply (src/mat/impls/inv/matinv.c:734-743) on the iterative path the reference takes for a
   non-factorisable inner matrix (KSPCG + PCNONE/PCJACOBI per block, matinv.c:535-540): block-wise
   (regularised) CG, every block with its own scalars.  rtol/max_it as KSP; regularisation is the
   caller's (MatRegularize is set-up code). */
typedef struct pmh_matinv_s *pmh_matinv;
int pmh_matinv_create(pmh_blockdiag K, double rtol, double atol, int max_it, int jacobi, pmh_matinv *Kplus);
int pmh_matinv_destroy(pmh_matinv Kplus);
/* MatInvSetNullSpace + Moore-Penrose wrapping P_R K^- P_R (QPTDualize -qpt_dualize_Kplus_mp, qptransform.c:1006-1062).
   R_host: kdim (<= 8) columns of length n, column-major; rows of block b hold that block's orthonormal kernel basis */
int pmh_matinv_set_nullspace(pmh_matinv Kplus, int kdim, const double *R_host);
int pmh_matinv_mult(pmh_matinv Kplus, const double *f, double *u);
/* -qpt_dualize_Kplus_left (QPTDualize qptransform.c:997-1062): K^+ := K^- P_R, what the reference takes when it had to compute the kernel itself.  K^- = the solve of a
   factorisation with null-pivot detection: the fixing dofs (ascending local indices; identity rows / columns in the K given to pmh_matinv_create) carry 0 and their
   equations are dropped.  Needs pmh_matinv_set_nullspace; the result is not projected.  nfix = 0: back to P_R K^- P_R. */
int pmh_matinv_set_left_inverse(pmh_matinv Kplus, int nfix, const int *fix_dofs_host);
/* DEVIATION from the reference, which factorises (matinv.c:481-580) and has no such case: a block whose load lies in the kernel of K_b altogether leaves the block CG only the
 * rounding residue of P_R f_b, which is not in the range of the singular K_b.  ||P_R f_b|| <= c eps ||f_b|| (default c = 64): the block's image is taken as zero (what the
 * Moore-Penrose inverse gives for a load in the kernel); c = 0 switches the rule off.  No other block's threshold is touched. */
int pmh_matinv_set_kernel_load_tolerance(pmh_matinv Kplus, double c);
int pmh_matinv_last_iterations(pmh_matinv Kplus, int *max_block_its, long long *total_spmv);

/* MatRegularize (src/mat/interface/permonmatregularize.c:198-287), the set-up step of the reference's default FETI path
   (-regularize 1 -> MAT_REG_EXPLICIT, qptransform.c:2215,2231; MATINV then factors / iterates on K_reg, matinv.c:449-459).
   Host routines (they run once per block on host CSR / dense R; the per-iteration work stays in pmh_matinv_mult):
   _pivots = MatRegularize_GetPivots_Private (:6-116): the d "fixing" DOFs picked from the p x d column-major kernel basis
             R_loc by complete pivoting from the last column backwards -- index bookkeeping, reproduced exactly;
   _Q      = MatRegularize_GetRegularization_Private (:118-160): RI (RI'RI)^{-1} RI' filtered at 10 eps (d x d row-major);
   _csr    = K_reg = K + rho (rho Q) on the union pattern (rho enters twice, :256,265, kept); rho is the caller's
             pmh_op_max_eigenvalue(K, 1.0, 20, ...) (:254).  Output arrays hold rowptr[n] + d*d entries. */
int pmh_mat_regularize_pivots(int p, int d, const double *R_host, int *pivots_out /* d, ascending */);
int pmh_mat_regularization_Q(int p, int d, const double *R_host, const int *pivots, double *Q_out, int *keep_out);
int pmh_mat_regularize_csr(int n, const int *rowptr, const int *col, const double *val, int d, const double *R_host, double rho, int *pivots_out, int *rowptr_out, int *col_out,
                           double *val_out, long long *nnz_out);

/* QPFetiGetBgtSF (src/qp/impls/feti/qpfeti.c:465-925): the signed gluing matrix of a decomposition from the subdomains' local-to-
   global dof maps (concatenated; l2g_start[nsub+1]).  type 0 nonred / 1 full / 2 orth (FetiGluingType), scale = -SCALE_ON
   (qpfeti.c:757-758), exclude = sorted global dofs left out (-feti_gluing_exclude_dirichlet).  Copies are ordered by subdomain
   index, links by ascending global dof; the leaves come out grouped by link in the summation order of MatMultTranspose_Gluing and
   feed pmh_gluing_create directly (leaves_row = position in the concatenated local numbering).  Host routine (integer set-up,
   runs once).  leaves_* NULL: counts only. */
int pmh_feti_gluing_from_l2g(int nsub, const int *l2g_start, const int *l2g, int type, int scale, int n_exclude, const int *exclude, int *n_lambda, int *n_leaves, int *leaves_row,
                             int *leaves_root, double *leaves_val);

/* F = B K^+ B' (QPTDualize qptransform.c:1103-1128; MatCreateProd matprod.c:42-48) */
int pmh_op_create_feti_dual(pmh_gluing B, pmh_matinv Kplus, pmh_op *F);
/* PCApply_Dual lumped: y = B K B' x (src/pc/impls/dual/pcdual.c:63-78) */
int pmh_pc_dual_lumped_apply(pmh_gluing B, pmh_blockdiag K, const double *x, double *y);

/* ---- explicit local dual operators: the exact K^+ path of F ------------------------------------------------
 * The reference forms an inverse explicitly column by column with its inner KSP (MatInvExplicitly_Inv, src/mat/impls/inv/
 * matinv.c:670-730, _Private :640-665: one KSPSolve per column of the identity); it applies K^+ inside F = B K^+ B'
 * (qptransform.c:1103-1128) by a per-block factorisation (matinv.c:435-590).  pmh_fexplicit is that explicit inverse restricted
 * to what F can see: for every block b of the rank the dense symmetric W_b = (K_b^+)[Gamma_b, Gamma_b], Gamma_b = the primal
 * dofs of the block that B touches.  F lambda = Bhat blockdiag(W_b) Bhat' lambda: two CSR launches + ONE dense fp64 GEMV that
 * streams 8 n_Gamma_b^2 bytes per block (HBM roofline; SURVEY 8d "dense path"), instead of an inner Krylov solve.
 * _assemble: the columns come from K^+ applications of `solver`, a MATINV whose nslots blocks are solved at once, one unit
 * right-hand side each, at tolerance rtol (1e-12: F exact to that).  block_class[b] / slot_class[s] name classes of identical
 * matrices (pmh_csr_block_classes): blocks of one class share their columns and any slot of the class may produce them; NULL, NULL
 * = slot s solves for block s (solver = the operator's own K^+).  Set-up cost: one K^+ application per
 * ceil(|union of the class's Gamma| / slots of the class). */
typedef struct pmh_fexplicit_s *pmh_fexplicit;
#define PMH_FX_FULL 0 /* W_b as a full row-major matrix: one GEMV, 8 n^2 bytes per block and apply */
#define PMH_FX_SYM 1  /* W_b = W_b' kept as its lower block-triangle (bands of 32 rows): a deterministic two-launch SYMV, 4 n^2 bytes */
#define PMH_FX_CLASS 2 /* pmh_fexplicit_create_shared: congruent blocks share ONE full matrix W_c = (K^+)[U_c, U_c] per class (U_c = union of their
                          Gamma_b) applied to the blocks' vectors together, 8 right-hand sides per pass: 8 n_c^2 bytes for the whole class */
int pmh_fexplicit_create(pmh_gluing B, pmh_blockdiag K, int storage, pmh_fexplicit *E); /* finds Gamma_b, allocates the dense blocks (zero) */
#define PMH_FX_CLASS_SYM 3 /* pmh_fexplicit_create_shared_sym: the same W_c kept as its lower block-triangle in 16 x 16 tiles (4 n_c^2 bytes for the
                              whole class); both products of a tile with the 8 right-hand sides run on the fp64 matrix instruction (4x4x4_4b) */
#define PMH_FX_CLASS_ORBIT 4 /* pmh_fexplicit_create_shared_orbit: W_c is invariant under the class's symmetries (pmh_fexplicit_set_class_symmetry /
                                _set_box_symmetry, REQUIRED before the assembly), so only the rows of the orbit representatives are kept (a cube: 1 / 48 of
                                the rows) and F's dense part becomes a GEMM over (representatives) x (operations x 8 right-hand sides) on the fp64 matrix
                                instruction: 48 flop per stored byte instead of 4 -- compute-bound instead of HBM-bound */
int pmh_fexplicit_create_shared(pmh_gluing B, pmh_blockdiag K, const int *block_class, pmh_fexplicit *E); /* storage PMH_FX_CLASS */
int pmh_fexplicit_create_shared_orbit(pmh_gluing B, pmh_blockdiag K, const int *block_class, pmh_fexplicit *E); /* storage PMH_FX_CLASS_ORBIT */
/* the same with the touched set of class c extended by the block-relative dofs extra_rel[extra_ptr[c] .. extra_ptr[c + 1]) -- e.g. its closure under the block's symmetries
   (pmh_box_symmetry_closure), which lets a class of ONE box-shaped block (non-congruent decompositions: one material per subdomain) keep all the box's operations */
int pmh_fexplicit_create_shared_orbit_union(pmh_gluing B, pmh_blockdiag K, const int *block_class, const int *extra_ptr, const int *extra_rel, pmh_fexplicit *E);
int pmh_fexplicit_apply_flops(pmh_fexplicit E, double *flops); /* PMH_FX_CLASS_ORBIT: useful flops of the dense apply = (rows of the row tiles) x (the columns their blocks need) x 2 n_c (the roofline of that storage is the fp64 MFMA peak); 0 otherwise */
int pmh_fexplicit_apply_flops_detail(pmh_fexplicit E, double *issued /* padded tiles, what the matrix cores execute */, double *dense /* every (representative, operation, block): the count without the pruning by the blocks' touched dofs */);
int pmh_fexplicit_create_shared_sym(pmh_gluing B, pmh_blockdiag K, const int *block_class, pmh_fexplicit *E); /* storage PMH_FX_CLASS_SYM */
int pmh_fexplicit_destroy(pmh_fexplicit E);
int pmh_fexplicit_sizes(pmh_fexplicit E, int *nblocks, int *n_gamma /* [nblocks] or NULL */, long long *dense_bytes, double *gemv_algorithmic_bytes);
/* several GPUs, congruent blocks: E built over ALL blocks (B = the global gluing, K = the global block structure); this rank
   assembles and applies the 128-row stripes idx = rank (mod size) of the size-ordered list -- an even share of the dense bytes; the
   all-reduce that ends B u completes F lambda (PMH_FX_CLASS: a contiguous range of the rows of W_c; PMH_FX_CLASS_SYM: whole 256-row super
   bands dealt in snake order).  Not for PMH_FX_FULL; before the assembly. */
int pmh_fexplicit_set_stripe(pmh_fexplicit E, int rank, int size);
int pmh_fexplicit_stripe_owner(int nblocks, const int *n_gamma, int size, int *owner_out); /* host: owner rank of every 128-row stripe, block after block */
int pmh_fexplicit_stripe_bytes(int nblocks, const int *n_gamma, int size, double *bytes_per_rank); /* host: dense bytes per rank under that rule */
/* set-up by symmetry (PMH_FX_CLASS_SYM): nsym signed permutations of the class's touched dofs U_c (pmh_fexplicit_class_union: their indices relative to the
   block start, ascending = the row numbering of W_c) under which K_c, hence K_c^+, is invariant -- posmap[g * n_c + c] = position in U_c of the image of the
   c-th touched dof, sign[g * n_c + c] = +-1, operation 0 the identity.  W[g p][g c] = sign_g[p] sign_g[c] W[p][c]: ONE K^+ solve per ORBIT of rows (a cube
   of identical Q1 elasticity elements: the 48 signed coordinate permutations, 48 x fewer set-up solves).  The caller vouches for the invariance; the
   assembly re-solves a batch of symmetry-filled rows directly and fails if they differ.  Before pmh_fexplicit_assemble. */
/* host helper: the signed dof permutations of a box of dims[0] x dims[1] x dims[2] nodes (x fastest, ndof dofs per node; ndof = 3: vector components)
   induced by the signed coordinate permutations that map the box onto itself and leave the block's matrix (CSR rowptr / col / val, n rows; NULL: not
   checked) invariant -- generators checked on nsample rows, the group is their closure (<= 48 operations, 0 = identity).  perm, sign: [48 * n]. */
int pmh_box_symmetries(const int *dims, int ndof, const int *rowptr, const int *col, const double *val, int nsample, int *nsym, int *perm, signed char *sign);
/* sorted union of the images of the dofs in_rel under those operations (out_rel: capacity nx ny nz ndof, may be NULL to get the count); host */
int pmh_box_symmetry_closure(const int *dims, int ndof, const int *rowptr, const int *col, const double *val, int n_in, const int *in_rel, int *n_out, int *out_rel, int *nsym);
/* the two together for box-shaped blocks: symmetries of the box checked against the CSR of one block of the class (column indices relative to the block),
   restricted to those that map the class's touched dofs onto themselves, handed to pmh_fexplicit_set_class_symmetry; nsym_used: how many (1 = none) */
int pmh_fexplicit_set_box_symmetry(pmh_fexplicit E, int cls, const int *dims, int ndof, const int *rowptr, const int *col, const double *val, int *nsym_used);
int pmh_fexplicit_class_union(pmh_fexplicit E, int cls, int *n_c, int *urel_out /* [n_c] or NULL */);
int pmh_fexplicit_set_class_symmetry(pmh_fexplicit E, int cls, int nsym, const int *posmap, const signed char *sign);
int pmh_fexplicit_class_sym_plan(int n_c, int size, int *megaband_owner /* [ceil(ceil(n_c / 256) / 4)] or NULL */, double *bytes_per_rank /* [size] or NULL */); /* host: PMH_FX_CLASS_SYM's rule */
int pmh_fexplicit_orbit_row_tile(int n_representatives, int *row_tile /* 128, 120, 112, 104 or 96 */, int *padded_rows); /* host: PMH_FX_CLASS_ORBIT's rule for the row tile of its GEMM */
int pmh_fexplicit_assemble(pmh_fexplicit E, pmh_matinv solver, int nslots, const int *slot_class, const int *block_class, double rtol, int max_it);
/* nslots == 8 x (blocks of the solver): the multi-right-hand-side K^+ (csrc/matinv_mv.hip) solves 8 columns per block and application, slot s = column s % 8 of block s / 8
   (slot_class per slot; PMH_ERR_SUP where that solver does not apply).  pmh_fexplicit_assemble_auto tries exactly that and falls back to one column per block;
   slot_class there names the class of every BLOCK of the solver.  Column-blocked set-up of the reference: MatInvExplicitly_Inv, src/mat/impls/inv/matinv.c:640-730. */
int pmh_fexplicit_assemble_auto(pmh_fexplicit E, pmh_matinv solver, const int *slot_class, const int *block_class, double rtol, int max_it, int *used_multi_rhs /* or NULL */);
int pmh_fexplicit_fill_pattern(pmh_fexplicit E, int byte); /* tuning helper: byte pattern instead of the assembly (not F afterwards) */
int pmh_fexplicit_assemble_stats(pmh_fexplicit E, long long *n_solves, double *seconds);
int pmh_fexplicit_get_block(pmh_fexplicit E, int b, double *W_host /* n_Gamma_b^2 row-major or NULL */, int *gamma_host /* or NULL */);
int pmh_fexplicit_mult(pmh_fexplicit E, const double *lambda, double *y);              /* y = F lambda (MatMult of the product) */
int pmh_fexplicit_compressed_size(pmh_fexplicit E, int *ntot, int *gstart /* [nblocks+1] or NULL */);
int pmh_fexplicit_dense_mult(pmh_fexplicit E, const double *xhat, double *yhat);        /* the dense kernel alone, compressed vectors */
int pmh_fexplicit_timing_enable(pmh_fexplicit E, int max_launches, int stride);         /* HIP-event pairs around the GEMV launches */
int pmh_fexplicit_timing_get(pmh_fexplicit E, int *launches, double *total_ms, double *first_kernel_ms /* SYM: k_fx_symv alone; CLASS_ORBIT: the GEMM kernel alone (total - first = the finishing kernel); or NULL */);
/* F = B K^+ B' built on this MATINV (pmh_op_create_feti_dual, the FETI chain) applies through E from now on (E built from the
   same B; NULL detaches).  K^+ f for a general f (d = B K^+ f - c, primal recovery) stays on the inner KSP. */
int pmh_matinv_attach_explicit(pmh_matinv Kplus, pmh_fexplicit E);
int pmh_matinv_set_tolerances(pmh_matinv Kplus, double rtol, double atol, int max_it);  /* KSPSetTolerances of MatInvGetKSP's KSP */
int pmh_matinv_get_tolerances(pmh_matinv Kplus, double *rtol, double *atol, int *max_it);
/* classes of bit-identical diagonal blocks of a block-diagonal host CSR (congruent subdomains): block_class[b] in [0, nclasses) */
int pmh_csr_block_classes(int nblocks, const int *block_rowstart, const int *rowptr, const int *col, const double *val, int *block_class, int *nclasses);

/* ---- QP transform chain of the (T)FETI path, data part (src/qp/interface/qptransform.c) -------------------------------
   QPTDualize (:1102-1174: F = B K^+ B', d = B K^+ f - c) -> QPTHomogenizeEq (:437-527: lambda~ = G'(GG')^{-1} e,
   b_bar = d - F lambda~, lb <- lb - lambda~) -> QPTEnforceEqByProjector (:215-316: A = P F P with a box, P F without;
   b = P b_bar).  f: n_x, c / lb: n_lambda (lb NULL = no box, -inf on equality rows otherwise), e: rows of G; pf NULL = no
   equality constraint (no floating subdomain).  The handles stay owned by the caller, the chain owns what it creates. */
typedef struct pmh_feti_chain_s *pmh_feti_chain;
int pmh_qpt_feti_chain_create(pmh_gluing B, pmh_matinv Kplus, const double *f, const double *c, pmh_qppf pf, const double *e, const double *lb, pmh_feti_chain *ch);
int pmh_qpt_feti_chain_get(pmh_feti_chain ch, pmh_op *F, pmh_op *A, double **d, double **b_bar, double **b, double **lb_new, double **lambda_tilde); /* borrowed */
/* QPTHomogenizeEqPostSolve_Private (:423-431) + the operator part of QPTDualizePostSolve_Private (:783-833):
   lambda = lambda_child + lambda~; u0 = K^+(f - B' lambda); r = F lambda - d (u0, r optional) */
int pmh_qpt_feti_chain_post_solve(pmh_feti_chain ch, const double *lambda_child, double *lambda, double *u0, double *r);
int pmh_qpt_feti_chain_destroy(pmh_feti_chain ch);
/* the numbers of -qp_chain_view_kkt (QPViewKKT qp.c:245-369 on every QP of the chain, qpchain.c:247-268) for the QPs this chain object stands for; linear chain (no dual box).
 * K: the block-diagonal stiffness of the primal QP; x_child: the solved vector of the last QP; lambda = x_child + lambda~; u: the recovered primal solution */
typedef struct {
  int    has_coarse;                             /* the projected and the homogenised QP exist (floating subdomains) */
  double proj_r, proj_normb;                     /* ||P F x - P b_bar||, ||P b_bar|| */
  double hom_r, hom_be, hom_normb;               /* ||F x - b_bar + (B'lambda)|| (0 by construction of the missing multiplier), ||G x||, ||b_bar|| */
  double dual_r, dual_be, dual_normb;            /* ||F lambda - d + (B'lambda)||, ||G lambda - e||, ||d|| */
  double prim_r, prim_be, prim_normb;            /* ||K u - f + B' lambda||, ||B u||, ||f|| */
  double prim_r_zeroed_operator;                 /* ||B' lambda - f||: the same line once QPTPostSolve_QPTMatISToBlockDiag has left K zeroed (-qpt_matis_to_diag_norm with Dirichlet by B) */
} pmh_feti_chain_kkt;
int pmh_qpt_feti_chain_kkt(pmh_feti_chain chain, pmh_blockdiag K, const double *x_child, const double *lambda, const double *u, pmh_feti_chain_kkt *out);

/* ---- dense-row SVM dual Hessian (BASELINE configs[4]) ---------------------------------------------------- */
/* H = diag(y) X X' diag(y), X: n_local x d row-major in HBM (d <= 256), applied as two GEMV passes; with a
   communicator the samples are sharded by rows and w = X'(y o a) is all-reduced (d doubles) between the passes */
int pmh_op_create_svm_dual(pmh_ctx ctx, int n_local, int d, const double *X_dev, const double *y_dev, pmh_op *op);
/* how many times the operator has streamed X since it was created (2 per plain application; inside pmh_mpgp_solve on one GPU with d = 64 the second pass of an
   application also does the first pass of the next one wherever the MPGP step allows it -- svm.hip, "paired passes" -- so a run of expansion steps costs 2 passes
   per step instead of 4): what a bandwidth figure for this operator has to be computed from */
int pmh_op_svm_dual_passes(pmh_op op, long long *passes);

/* ---- QPS SMALXE (src/qps/impls/smalxe/smalxe.c) -------------------------------------------------------- */
typedef struct {
  double rtol, atol, divtol;
  int    max_it;             /* outer, default 100 (smalxe.c:1203) */
  double M1_user;  int M1_direct;  double M1_update;
  double rtol_E;
  double rho_user; int rho_direct; double rho_update, rho_update_late;
  double eta_user; int eta_direct;
  double update_threshold;
  double maxeig, maxeig_tol; int maxeig_iter;
  int    inject_maxeig, inject_maxeig_set;
  int    inner_iter_min, inner_no_gtol_stop;
  /* ||Bu|| update (smalxe.c:878-886): be_implicit = BE has no mult slot, only B'B is available -> QPSSMALXEUpdateNormBu_SMALXEON
     (:265-285, ||Bu|| = sqrt(u'B'Bu), BtBu reused by the lambda update), with lag_enabled (-qps_smalxe_norm_update_lag) the lagged
     variant (:289-370; offset / Jstart / Jstep / Jend / lower / upper as :757-762, defaults :1190-1200) */
  int    be_implicit, lag_enabled, lag_offset, lag_start, lag_step, lag_end;
  double lag_lower, lag_upper;
  int    knoll;              /* -qps_smalxe_knoll (smalxe.c:764,938-943): u0 = P b */
  pmh_mpgp_opts inner;       /* inner MPGP (prefix smalxe_) */
} pmh_smalxe_opts;

typedef struct {
  int    iteration, reason, inner_iter_accu, state;
  int    M1_hits, eta_hits, M1_updates, rho_updates;
  double M1, rho, eta, normBu, enorm, rnorm, maxeig;
  pmh_mpgp_stats inner;
} pmh_smalxe_stats;

typedef struct pmh_smalxe_s *pmh_smalxe;
int pmh_smalxe_default_opts(pmh_smalxe_opts *o);
/* QPSSetUp_SMALXE smalxe.c:772-888: A (outer Hessian, e.g. P F P), b, u, box, pf with BE = G, cE = 0 */
int pmh_smalxe_create(pmh_ctx ctx, pmh_op A, const double *b, double *u, const double *lb, const double *ub, pmh_qppf pf, const pmh_smalxe_opts *o, pmh_smalxe *s);
int pmh_smalxe_destroy(pmh_smalxe s);
int pmh_smalxe_solve(pmh_smalxe s);                                  /* QPSSolve_SMALXE smalxe.c:893-997 */
/* Extension, off by default: carry A_rho u from the last gradient of an inner solve into the Lagrangian (smalxe.c:982 forms it by QPComputeObjective's own MatMult)
   and into the first gradient of the next inner solve (mpgp.c:500 starts every solve with MatMult) -- g' = g + rho_new B'B u, B'B u being at hand for the multiplier
   update: two operator applications less per outer iteration, the same iterates up to the rounding of g's recurrence over the CG steps of the inner solve.  The
   number of Hessian multiplications reported then differs from the reference's for the same solve. */
int pmh_smalxe_set_reuse_products(pmh_smalxe s, int on);
int pmh_smalxe_get_stats(pmh_smalxe s, pmh_smalxe_stats *st);
int pmh_smalxe_reset(pmh_smalxe s);                                  /* QPSReset: the state machine of the inner convergence test back to 1 (a solve from a fresh initial guess) */
int pmh_smalxe_set_inner_max_it(pmh_smalxe s, int max_it);           /* the inner QPS's iteration limit, summed over the outer iterations (smalxe.c:626-631: DIVERGED_ITS / outer BREAKDOWN beyond it) */
int pmh_smalxe_get_inner_max_it(pmh_smalxe s, int *max_it);
int pmh_smalxe_get_solution(pmh_smalxe s, pmh_ctx *ctx, double **u, int *n); /* QPGetSolutionVector: the caller's device vector (+ context, length); any pointer may be NULL */
int pmh_smalxe_get_penalized(pmh_smalxe s, pmh_op *A_rho, double **b_inner, double **Bt_mu); /* the penalised child QP (A + rho B'B, b - B'mu) and B'mu (borrowed; any may be NULL) */
int pmh_smalxe_get_inner(pmh_smalxe s, pmh_mpgp *inner);             /* QPSSMALXEGetInnerQPS smalxe.c:492-507 (borrowed) */

/* ---- options front end (QPSSetFromOptions qps.c:860-900, _MPGP mpgp.c:712-745, _SMALXE smalxe.c:696-766) ----------
 * The reference is configured from PETSc's options database; this parses the same keys out of a PETSc-style option string
 * (command line / permonrc contents) into the option structs, with the argument checks of the reference's setters.
 * prefix: options prefix of the QPS object ("" for the top solver); SMALXE's inner MPGP reads <prefix>smalxe_qps_* (smalxe.c:500-502).
 * unknown: optional buffer for the keys nobody consumed (PETSc's -options_left), space separated. */
typedef struct {
  char   type[16];           /* -qps_type: "mpgp" | "smalxe" | "pcpg" | "ksp"; "" = QPSSetDefaultType decides (qps.c:422-455) */
  double rtol, atol, divtol; /* -qps_rtol / -qps_atol / -qps_divtol (QPSSetTolerances qps.c:905-930) */
  int    max_it, max_it_set; /* -qps_max_it */
  int    monitor, monitor_cost, view, view_convergence, auto_post_solve;
} pmh_qps_opts;
int pmh_qps_default_opts(pmh_qps_opts *q);
int pmh_qps_set_from_options(const char *options, const char *prefix, pmh_qps_opts *q, pmh_mpgp_opts *m, pmh_smalxe_opts *s /* or NULL */, char *unknown, int unknown_cap);

/* ---- PC for the inner KSP of MATINV: multigrid V-cycle (PCMG semantics) ----------------------------------------
 * The reference's iterative MATINV applies K^+ with a PETSc KSP whose PC is chosen by -mat_inv_pc_type
 * (src/mat/impls/inv/matinv.c, MatInvGetKSP / MatInvSetUp).  pmh_mg is that PC on the device for PCMG-like set-ups:
 * A[0] is the fine operator (the same CSR the MATBLOCKDIAG holds), A[l+1] = P[l]' A[l] P[l] are handed over by the
 * caller, P[l] is n_l x n_{l+1}.  Smoother: Chebyshev of the given degree on D^-1 A over
 * [lo_frac, hi_frac] x lambda_max[l] (PETSc's PCMG/GAMG default is 0.1, 1.1), applied before and after the coarse
 * correction; the coarsest level is solved block-wise by the dense (pseudo-)inverses in coarse_pinv (row-major blocks
 * of sizes coarse_rowstart[b+1]-coarse_rowstart[b], concatenated).  The CSR handles stay owned by the caller.
 * precision: PMH_MG_FP64, or PMH_MG_FP32 = the cycle (operators, vectors, coarse inverses) in single precision -- it only
 * preconditions the fp64 CG; needs 3x3-block operators (elasticity, PETSc's BAIJ bs=3 case) on every smoothed level. */
#define PMH_MG_FP64 0
#define PMH_MG_FP32 1
#define PMH_MG_FP16 2 /* as FP32, the fine-level operator stored as (scaled) fp16 entries */
typedef struct pmh_mg_s *pmh_mg;
int pmh_mg_create(pmh_ctx ctx, int nlevels, const pmh_csr *A, const pmh_csr *P, int degree, const double *lambda_max, double lo_frac, double hi_frac, int nb_coarse, const int *coarse_rowstart,
                  const double *coarse_pinv_host, int precision, pmh_mg *mg);
/* The same PC with the hierarchy built HERE (host C++, runs once) for a block-diagonal A_fine whose blocks are Q1 discretisations on
 * boxes of dims[3 b .. 3 b + 2] nodes (x fastest, node-major dofs, ndof per node): trilinear prolongations (x) I_ndof (exact for rigid-body
 * modes), Galerkin operators, lambda_max(D^-1 A) by the power method, dense pseudo-inverses of the coarsest blocks
 * (A + Q Q')^{-1} - Q Q' with Q the kernel basis injected from R_host (kdim x n, zero over non-singular blocks; NULL if none floats).
 * Coarsening stops at <= min_nodes nodes per block; congruent blocks are processed once.  rowptr / col / val: host copy of A_fine. */
int pmh_mg_create_box(pmh_ctx ctx, pmh_csr A_fine, int nblocks, const int *block_rowstart, const int *dims, int ndof, const int *rowptr, const int *col, const double *val, int kdim,
                      const double *R_host, int min_nodes, int degree, int precision, pmh_mg *mg);
/* The same PC with an ALGEBRAIC hierarchy built HERE (host C++, runs once) for blocks of ANY shape -- subdomains that are not boxes, what a mesh partitioner hands
 * the reference, whose MATINV factorises any block (src/mat/impls/inv/matinv.c:481-580) and whose iterative path takes -mat_inv_pc_type gamg: smoothed aggregation
 * (Vanek, Mandel, Brezina 1996).  Per level: node graph (ndof dofs per node on the fine level, one node per aggregate below), couplings kept where the Frobenius norm
 * of their block exceeds theta * 0.5^level * sqrt(|A_ii| |A_jj|), greedy aggregates, a tentative prolongation that reproduces the block's near-kernel exactly, one
 * damped-Jacobi smoothing step of it, Galerkin operators; coarsening stops when every block has <= max_coarse dofs (every block is coarsened equally often);
 * congruent blocks are processed once.  Near-kernel of a block: its kernel vectors where R_host (kdim x n) is non-zero over it (a floating block: the rigid-body
 * modes; such a block stays consistently singular down the hierarchy and its coarsest operator gets a pseudo-inverse), else the rows of nns_host (nns x n; NULL or
 * zero over the block: the ndof translations).  A near-kernel of 3 or 6 vectors keeps 3 x 3 blocks on every level (PMH_MG_FP32 / PMH_MG_FP16 and the
 * multi-right-hand-side solver apply).  rowptr / col / val: host copy of A_fine. */
int pmh_mg_create_sa(pmh_ctx ctx, pmh_csr A_fine, int nblocks, const int *block_rowstart, int ndof, const int *rowptr, const int *col, const double *val, int kdim, const double *R_host,
                     int nns, const double *nns_host, int max_coarse, double theta, int degree, int precision, pmh_mg *mg);
/* host routine (tests, diagnostics): the aggregates pmh_mg_create_sa forms on ONE level of one block -- A: n x n host CSR with bs dofs per node; agg_out[n / bs] */
int pmh_sa_aggregate(int n, int bs, const int *rowptr, const int *col, const double *val, double theta, int *agg_out, int *n_agg);
/* host routine (tests, sanitizer runs; no device): the whole hierarchy pmh_mg_create_sa builds for ONE block (R: kdim x n kernel vectors; kdim = 0: a non-singular block, near-kernel = the
   ndof translations).  level_rows[0 .. *nlevels): rows per level (room for 16); defect[0] = max_l max|A_l B_l| / max|A_l|, defect[1] = max_l max|P_l B_{l+1} - B_l| (both 0 for kdim = 0),
   defect[2] = max|A_c A_c^+ A_c - A_c| / max|A_c| of the coarsest operator and its dense (pseudo-)inverse */
int pmh_sa_hierarchy_host(int n, int ndof, const int *rowptr, const int *col, const double *val, int kdim, const double *R, int max_coarse, double theta, int *nlevels, int *level_rows, double *defect);
int pmh_mg_apply(pmh_mg mg, const double *b_dev, double *x_dev); /* x = V(b), zero initial guess (PCApply) */
int pmh_mg_stats(pmh_mg mg, long long *fine_spmv);
int pmh_mg_timing_enable(pmh_mg mg, int max_launches); /* HIP-event pairs around the fine-level operator launches */
int pmh_mg_timing_get(pmh_mg mg, int *launches, double *total_ms, double *bytes_per_launch);
int pmh_mg_destroy(pmh_mg mg);
int pmh_matinv_set_pc_mg(pmh_matinv Kplus, pmh_mg mg); /* NULL: back to Jacobi / none */
/* MATINV's own K x product on the 3x3-block kernel (PETSc MATSEQBAIJ role: 8.44 instead of 12 bytes per non-zero);
 * returns PMH_ERR_SUP if K has no 3x3 block structure.  Timing as pmh_csr_timing_*. */
int pmh_matinv_enable_bsr3(pmh_matinv Kplus);
int pmh_matinv_bsr3_replicas(pmh_matinv Kplus, int *nrep); /* congruent blocks sharing the one device copy of the 3x3-block operator (1: a copy per block / no such copy) */
int pmh_matinv_timing_enable(pmh_matinv Kplus, int max_launches);
int pmh_matinv_timing_get(pmh_matinv Kplus, int *launches, double *total_ms, double *bytes_per_launch);

/* ---- QPS PCPG (src/qps/impls/pcpg/pcpg.c:51-134) -------------------------------------------------------- */
typedef struct {
  int    iteration, reason;
  double rnorm;
} pmh_pcpg_stats;
/* pc: NULL (none) or an operator applied as z = M^{-1} w (PCApply), e.g. the lumped dual preconditioner */
int pmh_pcpg_solve(pmh_ctx ctx, pmh_op A, const double *b, double *x, pmh_qppf pf, pmh_op pc, double rtol, double atol, double divtol, int max_it, pmh_pcpg_stats *st);

/* ---- QPS KSP (src/qps/impls/ksp/qpsksp.c:127-143): the solver QPSSetDefaultType picks for a QP without box or
 * equality constraints (qps.c:437-449), e.g. the projected dual of a linear TFETI problem or the FETI-1 dual.
 * The KSP is the one QPSCreate_KSP sets up (qpsksp.c:244-250): CG, unpreconditioned residual norm, x as initial
 * guess, pc NULL = PCNONE; stopping test QPSKSPConverged_KSP -> QPSConvergedDefault. */
int pmh_ksp_cg_solve(pmh_ctx ctx, pmh_op A, const double *b, double *x, pmh_op pc, double rtol, double atol, double divtol, int max_it, pmh_pcpg_stats *st);

/* ---- KSPFETI (src/ksp/impls/feti/feti.c): KSPFETISetUp (:71-94) + KSPSolve_FETI (:144-156) for a decomposed linear problem ----
 * Input = what the reference holds after QPTMatISToBlockDiag (qptransform.c:2007-2150): the block-diagonal K (host CSR, blocks
 * given by block_rowstart), f split among the copies of shared dofs, l2g (global dof of every local dof), the Dirichlet dofs to be
 * enforced by B (local indices, KSPFETISetDirichlet(..., FETI_LOCAL, PETSC_TRUE); none if they are eliminated in K), and the
 * kernel vectors R (kdim rows of length N, zero over non-floating blocks; orthonormalised internally).
 * Builds B = [B_d; B_g] (QPFetiAssembleDirichlet, QPFetiGetBgtSF), K^+ (MatRegularize + MATINV, or the Moore-Penrose wrapping),
 * G = R'B', the dual QP chain, solves it with the QPS the reference's QPSSetDefaultType picks (CG on P F, optional lumped PC) and
 * recovers u.  One call per solve; everything it creates is released before it returns. */
typedef struct {
  int    gluing_type;        /* -feti_gluing_type: 0 nonred, 1 full (default, qpfeti.c:322), 2 orth */
  int    scale;              /* -SCALE_ON (default 1) */
  int    exclude_dirichlet;  /* -feti_gluing_exclude_dirichlet (default 0) */
  int    regularize;         /* -regularize (default 1, qptransform.c:2215); 0: -qpt_dualize_Kplus_mp */
  int    kplus_left;         /* -qpt_dualize_Kplus_left (DEFAULT 1): K^+ = K^- P_R with the fixing dofs of MatRegularize as null pivots (pmh_matinv_set_left_inverse).  The reference switches to it,
                                and to -regularize 0, by itself whenever the QP came without a kernel and it computed one (qptransform.c:997-1008) -- always the case under KSPFETI, which
                                never sets one (feti.c:71-94; feti/ex1.c, ex71.c).  0: `regularize` decides (K_reg^{-1} or the Moore-Penrose form).  Ignored with explicit_dual (K^- P_R is not symmetric) */
  int    project;            /* -project (default 1, set by -feti: QPTEnforceEqByProjector).  0: the equality constraint G lambda = e stays in the dual QP, which is homogenised and
                                handed to QPS SMALXE (QPSSetDefaultType qps.c:437-441; QPTEnforceEqByPenalty inside SMALXE) -- `smalxe` below configures it */
  int    E_orth_type;        /* -dual_qp_E_orth_type (QPTOrthonormalizeEqFromOptions qptransform.c:643-660; MatOrthTypes): 0 none, 1 gs / 2 gslingen (explicit T G, T e), 3 cholesky / 4 implicit (G stays sparse, the projector carries T = L^{-1}) */
  int    lumped_pc;          /* -dual_pc_dual_type lumped (default none) */
  double regularize_rho;     /* > 0: the rho of MatRegularize for every block; 0 (default): the reference's power-method estimate */
  double kplus_rtol; int kplus_max_it; /* inner KSP of MATINV */
  double rtol, atol, divtol; int max_it; /* -qps_rtol ... of the dual solve (qps.c:73-76) */
  int    max_it_set;         /* 1: max_it was given (-qps_max_it, or by the caller): with project == 0 it replaces QPSCreate_SMALXE's own default of 100 (smalxe.c:1203) whatever its value */
  int    explicit_dual; double explicit_rtol; /* 1: F applies through the explicit local dual operators (pmh_fexplicit_*), assembled at explicit_rtol (default 0 / 1e-13) */
  /* the reference's post-solve report, QPChainPostSolve (src/qp/interface/qpchain.c:198-275): */
  int    view_convergence;   /* -qps_view_convergence: "  last QPSSolve CONVERGED due to ..., KSPReason=.., required .. iterations" (qps.c:1188-1230) */
  int    view_kkt;           /* -qp_chain_view_kkt: the `r = ...` lines of QPViewKKT (qp.c:245-369) for EVERY QP of the chain, last to first: projected dual, homogenised dual, dual
                                (twice: QPTScale always adds a child sharing all vectors, qptransform.c:1459), decomposed primal (twice), assembled original */
  int    matis_to_diag_norm; /* -qpt_matis_to_diag_norm: the "Dirichlet in Hess: .., r = ||Ax-b|| = .." line of QPTPostSolve_QPTMatISToBlockDiag (qptransform.c:1954-1979), and its
                                side effect on the two QPs printed after it when the Dirichlet dofs are enforced by B (see pmh_kspfeti_solve in kspfeti.hip) */
  char  *view_buf; int view_cap; /* the text (full PETSc viewer lines, '\n'-separated, NUL-terminated, truncated to view_cap) goes here; NULL: to stdout */
  pmh_smalxe_opts smalxe;    /* -qps_smalxe_* / -qps_smalxe_qps_* of the SMALXE solve taken with project == 0 (its outer tolerances are rtol / atol / divtol / max_it above) */
  int    kplus_pc;           /* -dual_mat_inv_pc_type: 0 jacobi (default: what the golden tests pin), 1 gamg = the algebraic V-cycle built inside the library (pmh_mg_create_sa) on the
                                matrix MATINV inverts, near-kernel = the kernel vectors R -- subdomains of any shape, any of the three generalised inverses */
  int    kplus_pc_ndof;      /* dofs per node of the aggregation's node graph; 0: 3 where the blocks come with 6 kernel vectors and 3 | their sizes, else 1 */
} pmh_kspfeti_opts;
typedef struct {
  int    iteration, reason;  /* CG iterations on P F; with project == 0 SMALXE's outer iterations */
  double rnorm;              /* ||P (F lambda - d)|| at exit; with project == 0 SMALXE's max(||G x||, ||g||) */
  int    n_lambda, n_dirichlet_rows, coarse_dim;
  pmh_smalxe_stats smalxe;   /* filled with project == 0 */
} pmh_kspfeti_stats;
/* vector part of QPTMatISToBlockDiag (qptransform.c:2095-2113: assembled rhs -> copies, interface values divided by their
   multiplicity) and of its post-solve (:1945-1949: INSERT_VALUES assembly of the solution, no averaging); host routines */
int pmh_qpt_matis_split_rhs(int N, const int *l2g, int n_global, const double *b_global, double *f_local);
int pmh_qpt_matis_assemble_solution(int N, const int *l2g, const double *u_local, int n_global, double *x_global);
/* matrix side of QPTMatISToBlockDiag (qptransform.c:2007-2150): MATIS = per-subdomain local matrices + l2g  ->  the MATBLOCKDIAG of the child QP
   (MatCreateBlockDiag(comm, matis->A)) in the concatenated local numbering, matis->counter (dof multiplicities: D = 1/counter), the interface
   flags of the local dofs and i2g = the sorted global interface dofs (QPFetiSetInterfaceToGlobalMapping); host routine.  loc_rowptr: the
   subdomains' row pointers one after the other (n_s + 1 entries each, each starting at 0); col / val / counter / is_interface / i2g may be NULL */
int pmh_qpt_matis_to_blockdiag(int nsub, const int *l2g_start, const int *l2g, int n_global, const int *loc_rowptr, const int *loc_col, const double *loc_val, int *block_rowstart,
                               int *rowptr, int *col, double *val, int *counter, int *is_interface, int *n_i2g, int *i2g);
int pmh_kspfeti_default_opts(pmh_kspfeti_opts *o);
/* the options-database keys of the FETI chain (-feti_gluing_type, -feti_gluing_exclude_dirichlet, -SCALE_ON, -regularize,
   -qpt_dualize_Kplus_mp, -dual_pc_dual_type, -qps_rtol/-qps_atol/-qps_divtol/-qps_max_it, -dual_mat_inv_ksp_rtol/_max_it, -dual_mat_inv_pc_type jacobi | gamg) */
int pmh_kspfeti_set_from_options(const char *options, pmh_kspfeti_opts *o, char *unknown, int unknown_cap);
int pmh_kspfeti_solve(pmh_ctx ctx, int nsub, const int *block_rowstart, const int *rowptr, const int *col, const double *val, const double *f, const int *l2g, int n_dir,
                      const int *dir_local, int kdim, const double *R, const pmh_kspfeti_opts *o, double *u_host, double *lambda_host /* or NULL */, int lambda_cap,
                      pmh_kspfeti_stats *st);

/* ---- TFETI contact problem in one call (QPTFromOptions / QPTAllInOne, src/qp/interface/qptransform.c:2152-2237: dualize ->
 * orthonormalize -> homogenize -> project; QPSSetDefaultType -> SMALXE + MPGP; the post-solve chain) ----------------------------------
 * Input = the decomposed QP: block-diagonal K (host CSR), f, the constraint matrix B as leaves (local primal dof, dual row, value) with
 * the n_eq equality rows (gluing, Dirichlet) first and the inequality rows (B_I u <= c_I, e.g. non-penetration) after them, c, the
 * kernel vectors R (kdim x N; every block floats in TFETI), optionally the blocks' node boxes (dims: nsub x 3, x fastest, node-major
 * dofs) for the multigrid PC of K^+.  K^+ = P_R K^- P_R (-regularize 0 -qpt_dualize_Kplus_mp); with explicit_dual the explicit local dual
 * operators carry F (exact, one dense SYMV per apply).  Output: u (N, host), lambda (n_lambda, host, optional), the solver's statistics. */
typedef struct {
  pmh_smalxe_opts smalxe;           /* outer tolerances + SMALXE / inner MPGP parameters (pmh_smalxe_default_opts) */
  double kplus_rtol; int kplus_max_it;
  int    mg, mg_min_nodes, mg_degree, mg_precision; /* multigrid PC of the inner KSP: the box hierarchy when dims != NULL (pmh_mg_create_box), else the algebraic one
                                                       (pmh_mg_create_sa on the kernel vectors; max_coarse = 3 mg_min_nodes): blocks of any shape */
  int    bsr3;                      /* K x of the inner CG on the 3x3-block kernel when ndof == 3 */
  int    explicit_dual; double explicit_rtol; int explicit_storage; /* pmh_fexplicit_* (PMH_FX_SYM / PMH_FX_FULL / PMH_FX_CLASS / PMH_FX_CLASS_SYM / PMH_FX_CLASS_ORBIT) */
  int    orthonormalize;            /* QPTOrthonormalizeEq: 1 G <- L^{-1} G formed explicitly, 2 implicitly (G stays sparse, -qp_E_orth_form implicit), 0 none */
  int    explicit_symmetry;         /* PMH_FX_CLASS_SYM / _ORBIT with dims != NULL: the symmetries of the box (pmh_fexplicit_set_box_symmetry) serve the set-up / the storage */
} pmh_feti_contact_opts;
typedef struct {
  pmh_smalxe_stats smalxe;
  int    n_lambda, n_eq, coarse_dim, n_active, explicit_solves;
  double setup_seconds, solve_seconds, explicit_seconds, norm_Glambda_minus_e;
  int    explicit_symmetries;       /* operations used by the set-up by symmetry (0 / 1: none) */
} pmh_feti_contact_stats;
int pmh_feti_contact_default_opts(pmh_feti_contact_opts *o);
int pmh_feti_contact_solve(pmh_ctx ctx, int nsub, const int *block_rowstart, const int *rowptr, const int *col, const double *val, const double *f, int n_lambda, int n_eq, int n_leaves,
                           const int *leaves_row, const int *leaves_root, const double *leaves_val, const double *c, int kdim, const double *R, const int *dims /* or NULL */, int ndof,
                           const pmh_feti_contact_opts *o, double *u_host, double *lambda_host /* or NULL */, pmh_feti_contact_stats *st);

/* ---- bench-only shims (csrc/bench_shims.hip: written over the public hooks above, nothing inside the solvers) --------------------------------
 * pmh_mpgp_run_fixed: exactly `iters` MPGP iterations (rtol = atol = 0, max_it = iters - 1 around pmh_mpgp_solve; the test is evaluated every iteration);
 * pmh_smalxe_run_fixed: the real SMALXE loop for exactly `inner_iters` inner iterations in total (inner iteration limit = what is left of the budget; a solve
 * that converges earlier restarts from u = 0 after pmh_smalxe_reset; counts accumulate over the restarts). */
int pmh_mpgp_run_fixed(pmh_mpgp s, int iters);
int pmh_smalxe_run_fixed(pmh_smalxe s, int inner_iters, int *solves, int *outer_iters, int *ncg, int *nexp, int *nprop, int *nmv);

/* ---- multi-right-hand-side K^+ (csrc/mv.hip, mg_mv.hip, matinv_mv.hip): PMH_MV_R = 8 columns per block solved together on interleaved multivectors
 * V[(dof) * 8 + column] -- the column-blocked set-up of the explicit inverse (MatInvExplicitly_Inv, src/mat/impls/inv/matinv.c:640-730: MatMatSolve on blocks of
 * right-hand sides) for blocks with NO symmetry and NO congruence: every 3 x 3 block of K_b is loaded once for the 8 columns.
 * pmh_mv_test_spmv: the operator product alone (measurement / test entry): Y = A X for a resident CSR of 3 x 3 blocks and 8 columns, entries kept in `storage`
 * (0 fp64, 1 fp32, 2 fp16 with fp32 vectors); x, y: device, 8 n doubles; the average launch time of `repeats` products through HIP events. */
int pmh_mv_test_spmv(pmh_csr A, int storage, const double *x, double *y, int repeats, float *ms_per_launch);
/* U = K^+ F for 8 columns per block at once: F, U device arrays of 8 n doubles, entry (dof i, column r) at i * 8 + r; K, the V-cycle (pmh_matinv_set_pc_mg), the kernel
 * basis (pmh_matinv_set_nullspace: K^+ = P_R K^- P_R) and the tolerances are the solver's own; every (block, column) pair converges by its own test.  PMH_ERR_SUP where
 * the multi-right-hand-side kernels do not apply (K without regular 3 x 3 blocks, a V-cycle other than the fused fp32 one, the left generalised inverse). */
int pmh_matinv_mult_multi(pmh_matinv M, const double *F, double *U, int *max_iterations /* or NULL */);
/* pmh_matinv_mult (MatMult_Inv, matinv.c:734-743) itself takes these kernels when the solver has exactly 8 congruent blocks (verified entry by entry by pmh_matinv_enable_bsr3)
   and the fused fp32 V-cycle: the 8 blocks' vectors are the 8 columns of ONE block.  *active: whether the next application will (knob "kplus_mv" / PMH_NO_KPLUS_MV for the A/B). */
int pmh_matinv_multi_rhs_active(pmh_matinv M, int *active);

#ifdef __cplusplus
}
#endif
#endif /* PERMON_HIP_H */
