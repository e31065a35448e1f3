/*
 * permon_oracle.c -- CPU restatement of PERMON's QPS hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * See permon_oracle.h.  The operation ORDER of the reference is kept on purpose (one pass per
 * PETSc Vec/Mat call, separate dot products), so iteration counts, step-type sequences and
 * monitor traces can be compared with the reference's golden outputs, and so that timing this
 * code is a fair stand-in for "the reference's PETSc CPU path" (BASELINE.md section 2).
 * Citations are file:line under /root/reference.
 */
#include "permon_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define ORC_EPS DBL_EPSILON /* PETSC_MACHINE_EPSILON for real double */

/* ------------------------------------------------------------------------------------------ */
/* PETSc Vec primitives restated (third-party semantics; each is one memory pass)              */
/* ------------------------------------------------------------------------------------------ */
static double v_dot(int n, const double *x, const double *y)
{
  double s = 0.0;
  int    i;
#if defined(_OPENMP)
#pragma omp parallel for reduction(+ : s) schedule(static)
#endif
  for (i = 0; i < n; i++) s += x[i] * y[i];
  return s;
}
static double v_norm2(int n, const double *x) { return sqrt(v_dot(n, x, x)); }
static void   v_copy(int n, const double *x, double *y)
{
  if (x != y) memcpy(y, x, (size_t)n * sizeof(double));
}
static void v_set(int n, double *x, double a)
{
  int i;
#if defined(_OPENMP)
#pragma omp parallel for schedule(static)
#endif
  for (i = 0; i < n; i++) x[i] = a;
}
static void v_axpy(int n, double *y, double a, const double *x) /* y += a x */
{
  int i;
#if defined(_OPENMP)
#pragma omp parallel for schedule(static)
#endif
  for (i = 0; i < n; i++) y[i] += a * x[i];
}
static void v_aypx(int n, double *y, double a, const double *x) /* y = x + a y */
{
  int i;
#if defined(_OPENMP)
#pragma omp parallel for schedule(static)
#endif
  for (i = 0; i < n; i++) y[i] = x[i] + a * y[i];
}
static void v_waxpy(int n, double *w, double a, const double *x, const double *y) /* w = a x + y */
{
  int i;
#if defined(_OPENMP)
#pragma omp parallel for schedule(static)
#endif
  for (i = 0; i < n; i++) w[i] = a * x[i] + y[i];
}
static void v_scale(int n, double *x, double a)
{
  int i;
#if defined(_OPENMP)
#pragma omp parallel for schedule(static)
#endif
  for (i = 0; i < n; i++) x[i] *= a;
}

double orc_now(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ------------------------------------------------------------------------------------------ */
/* MatMult_SeqAIJ (PETSc): y_i = sum_j a_ij x_j, left to right from 0.0.                        */
/* Call sites: mpgp.c:500,537,578,606,624; mpgp.c:250; permonmatutils.c:487                     */
/* ------------------------------------------------------------------------------------------ */
void orc_csr_mult(void *ctx, const double *x, double *y)
{
  const orc_csr *A = (const orc_csr *)ctx;
  int            i;
#if defined(_OPENMP)
#pragma omp parallel for schedule(static)
#endif
  for (i = 0; i < A->nrows; i++) {
    double s = 0.0;
    int    j;
    for (j = A->rowptr[i]; j < A->rowptr[i + 1]; j++) s += A->val[j] * x[A->col[j]];
    y[i] = s;
  }
}

void orc_csr_mult_transpose(void *ctx, const double *x, double *y)
{
  const orc_csr *A = (const orc_csr *)ctx;
  int            i, j;
  for (i = 0; i < A->ncols; i++) y[i] = 0.0;
  for (i = 0; i < A->nrows; i++)
    for (j = A->rowptr[i]; j < A->rowptr[i + 1]; j++) y[A->col[j]] += A->val[j] * x[i];
}

/* ------------------------------------------------------------------------------------------ */
/* QPC box.  Wrappers src/qpc/interface/qpc.c:466-615 (copy/zero, IS sub-vector), kernels       */
/* src/qpc/impls/box/qpcbox.c:21-146,290-305.                                                   */
/* ------------------------------------------------------------------------------------------ */
#define IDX(qpc, k) ((qpc)->is ? (qpc)->is[k] : (k))

/* QPCProject (qpc.c:466-491) + QPCProject_Box (qpcbox.c:290-305) */
void orc_box_project(const orc_box *qpc, const double *x, double *Px)
{
  int k;
  v_copy(qpc->n, x, Px); /* qpc.c:479 */
  for (k = 0; k < qpc->nis; k++) {
    int    i = IDX(qpc, k);
    double v = x[i];
    if (qpc->lb) {
      v = (v > qpc->lb[k]) ? v : qpc->lb[k];               /* VecPointwiseMax(Px,x,lb) */
      if (qpc->ub) v = (v < qpc->ub[k]) ? v : qpc->ub[k];  /* VecPointwiseMin(Px,Px,ub) */
    } else if (qpc->ub) {
      v = (v < qpc->ub[k]) ? v : qpc->ub[k];
    }
    Px[i] = v;
  }
}

/* QPCFeas (qpc.c:503-527) + QPCFeas_Box (qpcbox.c:104-146); single rank => the Allreduce(MIN) is the identity */
double orc_box_feas(const orc_box *qpc, const double *x, const double *d)
{
  double alpha_temp = INFINITY, alpha_i;
  int    k;
  for (k = 0; k < qpc->nis; k++) {
    int i = IDX(qpc, k);
    if (d[i] > 0. && qpc->lb && qpc->lb[k] > -INFINITY) {
      alpha_i = x[i] - qpc->lb[k];
      alpha_i = alpha_i / d[i];
      if (alpha_i < alpha_temp) alpha_temp = alpha_i;
    }
    if (d[i] < 0. && qpc->ub && qpc->ub[k] < INFINITY) {
      alpha_i = x[i] - qpc->ub[k];
      alpha_i = alpha_i / d[i];
      if (alpha_i < alpha_temp) alpha_temp = alpha_i;
    }
  }
  return alpha_temp;
}

/* QPCGrads (qpc.c:540-569: gf=g, gc=0 first) + QPCGrads_Box (qpcbox.c:21-64) */
void orc_box_grads(const orc_box *qpc, const double *x, const double *g, double *gf, double *gc)
{
  int k;
  v_copy(qpc->n, g, gf);
  v_set(qpc->n, gc, 0.0);
  for (k = 0; k < qpc->nis; k++) {
    int i = IDX(qpc, k);
    if (qpc->lb && fabs(x[i] - qpc->lb[k]) <= qpc->astol) {
      gf[i] = 0.0;
      gc[i] = (g[i] < 0.0) ? g[i] : 0.0; /* PetscMin(g,0) */
    } else if (qpc->ub && fabs(x[i] - qpc->ub[k]) <= qpc->astol) {
      gf[i] = 0.0;
      gc[i] = (g[i] > 0.0) ? g[i] : 0.0; /* PetscMax(g,0) */
    } else {
      gf[i] = g[i];
    }
  }
}

/* QPCGradReduced (qpc.c:589-615: gr=gf first) + QPCGradReduced_Box (qpcbox.c:68-100) */
void orc_box_gradreduced(const orc_box *qpc, const double *x, const double *gf, double alpha, double *gr)
{
  int k;
  v_copy(qpc->n, gf, gr);
  for (k = 0; k < qpc->nis; k++) {
    int i = IDX(qpc, k);
    if (qpc->lb && gf[i] > 0.0) {
      double t = (x[i] - qpc->lb[k]) / alpha;
      gr[i]    = (gf[i] < t) ? gf[i] : t; /* PetscMin(a,b) = a<b?a:b */
    } else if (qpc->ub && gf[i] < 0.0) {
      double t = (x[i] - qpc->ub[k]) / alpha;
      gr[i]    = (gf[i] < t) ? t : gf[i]; /* PetscMax(a,b) = a<b?b:a */
    }
  }
}

/* ------------------------------------------------------------------------------------------ */
/* MatGetMaxEigenvalue, src/mat/interface/permonmatutils.c:442-522 (v = 1 start, tol 1e-4,       */
/* <= 50 its, v = Av/sqrt(v'v) quirk kept).  The lambda < eps random-restart branch (:491-499)   */
/* replaces A*v by PETSc's RAND48 stream: PetscRandomCreate seeds 0x12345678 (+76543*rank, rank   */
/* 0 here), PetscRandomSeed_Rand48 = srand48(seed), VecSetRandom_Seq = drand48() per entry in     */
/* index order (PETSc src/sys/classes/random/impls/rand48/rand48.c, un-vendored: restated from    */
/* its published source; no golden of the reference exercises this branch => parity unpinned).    */
/* lambda keeps its (< eps) value in that iteration, exactly as in the reference.                 */
/* ------------------------------------------------------------------------------------------ */
double orc_max_eigenvalue(const orc_op *A, double tol, int maxits, int *its_out)
{
  int     n = A->n, i;
  double *v = (double *)malloc((size_t)n * sizeof(double));
  double *Av = (double *)malloc((size_t)n * sizeof(double));
  double  lambda = 0.0, lambda0, err, relerr, vAv, vv;
  int     rand_seeded = 0;
  if (tol == ORC_DECIDE || tol == -2.0) tol = 1e-4;
  if (maxits == -1 || maxits == -2) maxits = 50;
  v_set(n, v, 1.0);
  for (i = 1; i <= maxits; i++) {
    lambda0 = lambda;
    A->mult(A->ctx, v, Av);
    vAv    = v_dot(n, v, Av); /* VecMDot(v,2,{Av,v}) */
    vv     = v_dot(n, v, v);
    lambda = vAv / vv;
    if (lambda < ORC_EPS) { /* :491-499 */
      int k;
      if (!rand_seeded) {
        srand48(0x12345678L);
        rand_seeded = 1;
      }
      for (k = 0; k < n; k++) Av[k] = drand48();
      vAv = v_dot(n, v, Av); /* VecDot(v,Av,&vAv_vv[0]): lambda itself is NOT recomputed */
    }
    err    = fabs(lambda - lambda0);
    relerr = err / fabs(lambda);
    if (relerr < tol) break;
    v_copy(n, Av, v);
    v_scale(n, v, 1.0 / sqrt(vv));
  }
  (void)vAv;
  if (its_out) *its_out = i;
  free(v);
  free(Av);
  return lambda;
}

/* ------------------------------------------------------------------------------------------ */
/* QPS base + MPGP                                                                              */
/* ------------------------------------------------------------------------------------------ */
/* QPSCreate defaults qps.c:73-76; QPSCreate_MPGP defaults mpgp.c:827-843 */
void orc_qps_init(orc_qps *qps)
{
  memset(qps, 0, sizeof(*qps));
  qps->rtol          = 1e-5;
  qps->atol          = 1e-50;
  qps->divtol        = 1e4;
  qps->max_it        = 10000;
  qps->alpha_user    = ORC_DECIDE;
  qps->alpha_direct  = 0;
  qps->gamma         = 1.0;
  qps->maxeig        = ORC_DECIDE;
  qps->maxeig_tol    = ORC_DECIDE;
  qps->maxeig_iter   = -1;
  qps->bchop_tol     = 0.0;
  qps->exptype       = ORC_EXP_STD;
  qps->explengthtype = ORC_EXPLEN_FIXED;
  qps->expproject    = 1;
  qps->converged     = orc_converged_default;
  qps->currentStepType = ' ';
}

void orc_qps_free(orc_qps *qps)
{
  int i;
  for (i = 0; i < 10; i++) {
    free(qps->work[i]);
    qps->work[i] = NULL;
  }
}

/* QPSConvergedDefault qps.c:675-714 + QPSConvergedDefaultSetUp :718-731 */
void orc_converged_default(orc_qps *qps, void *ctx)
{
  int    i     = qps->iteration;
  double rnorm = qps->rnorm;
  (void)ctx;
  qps->reason = ORC_CONVERGED_ITERATING;
  if (!qps->cvg_setup_called) {
    qps->norm_rhs         = v_norm2(qps->A->n, qps->b);
    qps->ttol             = fmax(qps->rtol * qps->norm_rhs, qps->atol);
    qps->norm_rhs_div     = qps->norm_rhs;
    qps->cvg_setup_called = 1;
  }
  if (i > qps->max_it) { /* strict, qps.c:688 */
    qps->reason = ORC_DIVERGED_ITS;
    return;
  }
  if (isnan(rnorm) || isinf(rnorm)) {
    qps->reason = ORC_DIVERGED_NANORINF;
  } else if (rnorm <= qps->ttol) {
    if (rnorm < qps->atol) qps->reason = ORC_CONVERGED_ATOL;
    else qps->reason = ORC_CONVERGED_RTOL;
  } else if (rnorm >= qps->divtol * qps->norm_rhs_div) {
    qps->reason = ORC_DIVERGED_DTOL;
  }
}

static double *orc_vec(int n) { return (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double)); }

/* VecFilter(v,tol) (PETSc): |v_i| < tol -> 0 */
static void v_filter(int n, double *v, double tol)
{
  int i;
  for (i = 0; i < n; i++)
    if (fabs(v[i]) < tol) v[i] = 0.0;
}

/* QPSSetup_MPGP mpgp.c:359-428 */
int orc_mpgp_setup(orc_qps *qps)
{
  int n = qps->A->n, nwork, i;
  if (qps->setupcalled) return 0;
  if (qps->fallback || qps->fallback2) nwork = (qps->explengthtype != ORC_EXPLEN_BB) ? 9 : 10;
  else if (qps->explengthtype == ORC_EXPLEN_BB) nwork = 9;
  else nwork = 7;
  for (i = 0; i < nwork; i++)
    if (!qps->work[i]) qps->work[i] = orc_vec(n);
  if (qps->bchop_tol) { /* mpgp.c:379-382 (bounds are chopped in place in the reference) */
    if (qps->qpc->lb) v_filter(qps->qpc->nis, (double *)qps->qpc->lb, qps->bchop_tol);
    if (qps->qpc->ub) v_filter(qps->qpc->nis, (double *)qps->qpc->ub, qps->bchop_tol);
  }
  qps->expproject = 1;
  if (qps->exptype == ORC_EXP_STD && qps->explengthtype == ORC_EXPLEN_FIXED) qps->expproject = 0; /* mpgp.c:388 */
  if (!qps->alpha_direct) { /* QPS_ARG_MULTIPLE, mpgp.c:417-422 */
    if (qps->maxeig == ORC_DECIDE) qps->maxeig = orc_max_eigenvalue(qps->A, qps->maxeig_tol, qps->maxeig_iter, NULL);
    if (qps->alpha_user == ORC_DECIDE) qps->alpha_user = 2.0;
    qps->alpha = qps->alpha_user / qps->maxeig;
  } else {
    qps->alpha = qps->alpha_user;
  }
  qps->setupcalled = 1;
  return 0;
}

/* vector selection of QPSSetup_MPGP mpgp.c:384-414 */
static double *exp_direction(orc_qps *qps)
{
  switch (qps->exptype) {
  case ORC_EXP_STD: return qps->work[6];
  case ORC_EXP_GF: return qps->work[1];
  case ORC_EXP_G: return qps->work[3];
  case ORC_EXP_GFGR: return qps->work[1];
  case ORC_EXP_GGR: return qps->work[3];
  case ORC_EXP_PROJCG: return qps->work[1];
  }
  return NULL;
}
static double *exp_lengthvec(orc_qps *qps)
{
  switch (qps->exptype) {
  case ORC_EXP_STD: return qps->work[6];
  case ORC_EXP_GF: return qps->work[1];
  case ORC_EXP_G: return qps->work[3];
  case ORC_EXP_GFGR: return qps->work[6];
  case ORC_EXP_GGR: return qps->work[6];
  case ORC_EXP_PROJCG: return qps->work[1];
  }
  return NULL;
}

/* MPGPGrads mpgp.c:198-223 */
static void mpgp_grads(orc_qps *qps, const double *x, const double *g)
{
  double *gP = qps->work[0], *gf = qps->work[1], *gc = qps->work[2], *gr = qps->work[6];
  orc_box_grads(qps->qpc, x, g, gf, gc);
  orc_box_gradreduced(qps->qpc, x, gf, qps->alpha, gr);
  v_waxpy(qps->A->n, gP, 1.0, gf, gc);
}

/* MPGPExpansionLength mpgp.c:233-287 */
static void mpgp_expansion_length(orc_qps *qps, double *xold, double *explengthvecold)
{
  int     n  = qps->A->n;
  double *lv = exp_lengthvec(qps);
  double  dots[2];
  switch (qps->explengthtype) {
  case ORC_EXPLEN_FIXED: break;
  case ORC_EXPLEN_OPT: {
    double *g = qps->work[3], *Ap = qps->work[5];
    qps->A->mult(qps->A->ctx, lv, Ap);
    qps->nmv++;
    dots[0] = v_dot(n, lv, g);
    dots[1] = v_dot(n, lv, Ap);
    if (dots[1] == .0 && qps->resetalpha) qps->alpha = qps->alpha / qps->maxeig;
    else qps->alpha = qps->alpha_user * (dots[0] / dots[1]);
  } break;
  case ORC_EXPLEN_OPTAPPROX: {
    double *g = qps->work[3];
    if (g != lv) {
      dots[0]    = v_dot(n, lv, g);
      dots[1]    = v_dot(n, lv, lv);
      qps->alpha = qps->alpha_user * (dots[0] / dots[1]);
    } else {
      qps->alpha = qps->alpha_user;
    }
    qps->alpha = qps->alpha / qps->maxeig;
  } break;
  case ORC_EXPLEN_BB: {
    double *v0 = explengthvecold, *v1 = xold;
    v_aypx(n, v0, -1.0, lv);     /* s_k */
    v_aypx(n, v1, -1.0, qps->x); /* y_k */
    dots[0] = v_dot(n, v0, v0);
    dots[1] = v_dot(n, v0, v1);
    if (dots[1] == .0 && qps->resetalpha) qps->alpha = qps->alpha / qps->maxeig;
    else qps->alpha = qps->alpha_user * (dots[0] / dots[1]);
  } break;
  }
}

/* MPGPExpansion_Std mpgp.c:299-323 */
static void mpgp_expansion_std(orc_qps *qps, double afeas, double *xold, double *explengthvecold)
{
  int     n = qps->A->n;
  double *g = qps->work[3], *p = qps->work[4], *Ap = qps->work[5];
  v_axpy(n, qps->x, -afeas, p);
  v_axpy(n, g, -afeas, Ap);
  mpgp_grads(qps, qps->x, g);
  mpgp_expansion_length(qps, xold, explengthvecold);
  v_axpy(n, qps->x, -qps->alpha, exp_direction(qps));
}

static void mpgp_trace(orc_qps *qps)
{
  int k = qps->trace_len;
  if (k < qps->trace_cap) {
    qps->trace_step[k]   = qps->currentStepType;
    qps->trace_rnorm[k]  = qps->rnorm;
    qps->trace_gfnorm[k] = qps->gfnorm;
    qps->trace_gcnorm[k] = qps->gcnorm;
    qps->trace_alpha[k]  = qps->alpha;
  }
  qps->trace_len = k + 1;
}

/* QPSSolve_MPGP mpgp.c:438-650 */
int orc_mpgp_solve(orc_qps *qps)
{
  const orc_op *A = qps->A;
  int           n = A->n;
  double       *x = qps->x;
  const double *b = qps->b;
  double       *gP, *gf, *gc, *g, *p, *Ap, *gold = NULL, *xold = NULL, *explengthvecold = NULL;
  double        gamma2, acg, bcg, afeas, pAp, gcTgc, gfTgf, f, fold;
  int           nmv = 0, ncg = 0, nprop = 0, nexp = 0, nfinc = 0, nfall = 0;
  int           fallback = qps->fallback;

  orc_mpgp_setup(qps);
  gP = qps->work[0];
  gf = qps->work[1];
  gc = qps->work[2];
  g  = qps->work[3];
  p  = qps->work[4];
  Ap = qps->work[5];
  if (qps->explengthtype == ORC_EXPLEN_BB) {
    explengthvecold = qps->work[7];
    xold            = qps->work[8];
    if (qps->fallback || qps->fallback2) gold = qps->work[9];
  } else if (qps->fallback || qps->fallback2) {
    xold = qps->work[7];
    gold = qps->work[8];
  }
  gamma2 = qps->gamma * qps->gamma;

  orc_box_project(qps->qpc, x, x); /* mpgp.c:497 */
  A->mult(A->ctx, x, g);           /* :500 */
  nmv++;
  v_axpy(n, g, -1.0, b);
  mpgp_grads(qps, x, g);
  v_copy(n, gf, p);

  qps->currentStepType = ' ';
  qps->iteration       = 0;
  qps->trace_len       = 0;
  while (1) {
    qps->rnorm = v_norm2(n, gP);
    gcTgc      = v_dot(n, gc, gc);
    gfTgf      = v_dot(n, gf, gf);
    qps->gfnorm = sqrt(gfTgf);
    qps->gcnorm = sqrt(gcTgc);
    if (qps->trace_cap) mpgp_trace(qps);

    qps->converged(qps, qps->converged_ctx);
    if (qps->reason != ORC_CONVERGED_ITERATING) break;

    if (gcTgc <= gamma2 * gfTgf) {
      A->mult(A->ctx, p, Ap);
      nmv++;
      pAp   = v_dot(n, p, Ap);
      acg   = v_dot(n, g, p);
      acg   = acg / pAp;
      afeas = orc_box_feas(qps->qpc, x, p);
      if (acg <= afeas) {
        ncg++;
        qps->currentStepType = 'c';
        v_axpy(n, x, -acg, p);
        v_axpy(n, g, -acg, Ap);
        mpgp_grads(qps, x, g);
        bcg = v_dot(n, Ap, gf);
        bcg = bcg / pAp;
        v_aypx(n, p, -bcg, gf);
      } else {
        nexp++;
        qps->currentStepType = 'e';
        if (qps->explengthtype == ORC_EXPLEN_BB || fallback || qps->fallback2) {
          v_copy(n, x, xold);
          if (qps->explengthtype == ORC_EXPLEN_BB) v_copy(n, exp_lengthvec(qps), explengthvecold);
        }
        if (qps->exptype == ORC_EXP_PROJCG) v_axpy(n, x, -acg, p); /* MPGPExpansion_ProjCG mpgp.c:335-349 */
        else mpgp_expansion_std(qps, afeas, xold, explengthvecold);
        if (qps->expproject) orc_box_project(qps->qpc, x, x);

        if (fallback || qps->fallback2) v_copy(n, g, gold);
        A->mult(A->ctx, x, g);
        nmv++;
        v_axpy(n, g, -1.0, b);

        if (fallback || qps->fallback2) {
          fold = orc_objective_from_gradient(n, b, xold, gold);
          f    = orc_objective_from_gradient(n, b, x, g);
          if (f > fold) {
            nfinc++;
            if (qps->fallback2) {
              mpgp_grads(qps, x, g);
              gcTgc    = v_dot(n, gc, gc);
              gfTgf    = v_dot(n, gf, gf);
              fallback = (gcTgc <= gamma2 * gfTgf) ? 0 : 1;
            }
            if (fallback) {
              nfall++;
              qps->currentStepType = 'f';
              v_copy(n, xold, x);
              v_copy(n, gold, g);
              if (qps->fallback2) mpgp_grads(qps, xold, gold);
              { /* MPGPExpansion_Std with the std vectors is what mpgp.c:604 calls; exptype stays */
                mpgp_expansion_std(qps, afeas, xold, explengthvecold);
              }
              orc_box_project(qps->qpc, x, x);
              A->mult(A->ctx, x, g);
              nmv++;
              v_axpy(n, g, -1.0, b);
            }
          }
        }
        mpgp_grads(qps, x, g);
        v_copy(n, gf, p);
      }
    } else {
      nprop++;
      qps->currentStepType = 'p';
      v_copy(n, gc, p);
      A->mult(A->ctx, p, Ap);
      nmv++;
      pAp = v_dot(n, p, Ap);
      acg = v_dot(n, g, p);
      acg = acg / pAp;
      v_axpy(n, x, -acg, p);
      v_axpy(n, g, -acg, Ap);
      mpgp_grads(qps, x, g);
      v_copy(n, gf, p);
    }
    qps->iteration++;
  }
  qps->fallback = fallback;
  qps->ncg += ncg;
  qps->nexp += nexp;
  qps->nmv += nmv;
  qps->nprop += nprop;
  qps->nfinc += nfinc;
  qps->nfall += nfall;
  return 0;
}

/* QPComputeObjective qp.c:913-927: f = -x'(b - 1/2 A x) */
double orc_objective(const orc_op *A, const double *b, const double *x, double *work)
{
  A->mult(A->ctx, x, work);
  v_aypx(A->n, work, -0.5, b);
  return -v_dot(A->n, x, work);
}

/* QPComputeObjectiveFromGradient qp.c:981-996: f = 1/2 x'(g - b) */
double orc_objective_from_gradient(int n, const double *b, const double *x, const double *g)
{
  double s = 0.0;
  int    i;
  for (i = 0; i < n; i++) s += x[i] * (-1.0 * b[i] + g[i]); /* VecWAXPY(xwork,-1,b,g); VecDot */
  return .5 * s;
}

/* ------------------------------------------------------------------------------------------ */
/* dense Cholesky for the coarse problem (GG')^{-1}: the reference uses MATINV -> KSPPREONLY +   */
/* PCCHOLESKY/LU (matinv.c:481-580), i.e. a direct solve; restated as a dense LL' solve.          */
/* ------------------------------------------------------------------------------------------ */
int orc_dense_cholesky(int m, double *a)
{
  int i, j, k;
  for (j = 0; j < m; j++) {
    double d = a[j * m + j];
    for (k = 0; k < j; k++) d -= a[j * m + k] * a[j * m + k];
    if (d <= 0.0) return 1;
    d            = sqrt(d);
    a[j * m + j] = d;
    for (i = j + 1; i < m; i++) {
      double s = a[i * m + j];
      for (k = 0; k < j; k++) s -= a[i * m + k] * a[j * m + k];
      a[i * m + j] = s / d;
    }
    for (i = 0; i < j; i++) a[i * m + j] = 0.0;
  }
  return 0;
}

void orc_dense_chol_solve(int m, const double *l, const double *rhs, double *sol)
{
  int i, k;
  for (i = 0; i < m; i++) {
    double s = rhs[i];
    for (k = 0; k < i; k++) s -= l[i * m + k] * sol[k];
    sol[i] = s / l[i * m + i];
  }
  for (i = m - 1; i >= 0; i--) {
    double s = sol[i];
    for (k = i + 1; k < m; k++) s -= l[k * m + i] * sol[k];
    sol[i] = s / l[i * m + i];
  }
}

/* QPPFApplyQ qppf.c:454-503 (the (v,state)->Qv cache only saves work, values are identical) */
void orc_qppf_apply_Q(const orc_qppf *pf, const double *v, double *Qv)
{
  orc_csr_mult((void *)pf->G, v, pf->G_left);
  if (pf->GGt_chol) {
    orc_dense_chol_solve(pf->m, pf->GGt_chol, pf->G_left, pf->Gt_right); /* QPPFApplyCP qppf.c:610-645 */
    orc_csr_mult_transpose((void *)pf->G, pf->Gt_right, Qv);
  } else {
    orc_csr_mult_transpose((void *)pf->G, pf->G_left, Qv);
  }
}

/* QPPFApplyP qppf.c:563-575 */
void orc_qppf_apply_P(const orc_qppf *pf, const double *v, double *Pv)
{
  if (v == Pv) {
    double *t = orc_vec(pf->n);
    orc_qppf_apply_Q(pf, v, t);
    v_aypx(pf->n, t, -1.0, v);
    v_copy(pf->n, t, Pv);
    free(t);
    return;
  }
  orc_qppf_apply_Q(pf, v, Pv);
  v_aypx(pf->n, Pv, -1.0, v);
}

/* QPPFApplyGtG qppf.c:580-605 */
void orc_qppf_apply_GtG(const orc_qppf *pf, int orthonormal, const double *v, double *y)
{
  if (orthonormal) {
    orc_qppf_apply_Q(pf, v, y);
    return;
  }
  orc_csr_mult((void *)pf->G, v, pf->G_left);
  orc_csr_mult_transpose((void *)pf->G, pf->G_left, y);
}

/* QPPFApplyHalfQ qppf.c:507-527 */
void orc_qppf_apply_halfQ(const orc_qppf *pf, const double *x, double *y)
{
  orc_csr_mult((void *)pf->G, x, pf->G_left);
  if (pf->GGt_chol) orc_dense_chol_solve(pf->m, pf->GGt_chol, pf->G_left, y);
  else v_copy(pf->m, pf->G_left, y);
}

/* QPPFApplyHalfQTranspose qppf.c:531-559 */
void orc_qppf_apply_halfQ_transpose(const orc_qppf *pf, const double *x, double *y)
{
  if (pf->GGt_chol) {
    orc_dense_chol_solve(pf->m, pf->GGt_chol, x, pf->Gt_right);
    orc_csr_mult_transpose((void *)pf->G, pf->Gt_right, y);
  } else {
    orc_csr_mult_transpose((void *)pf->G, x, y);
  }
}

/* MatMult_Penalized matpenalized.c:12-22 */
void orc_penalized_mult(void *ctx, const double *x, double *y)
{
  orc_penalized *P = (orc_penalized *)ctx;
  int            n = P->A->n, i;
  double        *t = orc_vec(n);
  orc_qppf_apply_GtG(P->pf, P->orthonormal, x, y);
  v_scale(n, y, P->rho);
  P->A->mult(P->A->ctx, x, t); /* MatMultAdd(A,x,y,y) */
  for (i = 0; i < n; i++) y[i] = y[i] + t[i];
  free(t);
}

/* MatCreateProd(P,A,P): y = P(A(P x)) qptransform.c:278-283, matprod.c:42-48 */
void orc_pap_mult(void *ctx, const double *x, double *y)
{
  orc_pap *c = (orc_pap *)ctx;
  orc_qppf_apply_P(c->pf, x, c->w1);
  c->A->mult(c->A->ctx, c->w1, c->w2);
  orc_qppf_apply_P(c->pf, c->w2, y);
}

/* MatCreateProd(P,A): y = P(A x) qptransform.c:273-277 */
void orc_pa_mult(void *ctx, const double *x, double *y)
{
  orc_pap *c = (orc_pap *)ctx;
  c->A->mult(c->A->ctx, x, c->w2);
  orc_qppf_apply_P(c->pf, c->w2, y);
}

/* ------------------------------------------------------------------------------------------ */
/* SMALXE  src/qps/impls/smalxe/smalxe.c                                                        */
/* ------------------------------------------------------------------------------------------ */
/* QPSCreate_SMALXE defaults smalxe.c:1159-1207 */
void orc_smalxe_init(orc_smalxe *s)
{
  memset(s, 0, sizeof(*s));
  s->rtol               = 1e-5;
  s->atol               = 1e-50;
  s->divtol             = 1e4;
  s->max_it             = 100;
  s->M1_user            = 1e2;
  s->M1_direct          = 0;
  s->M1_update          = 2.0;
  s->rtol_E             = 1e-0;
  s->rho_user           = 1.1;
  s->rho_direct         = 0;
  s->rho_update         = 1.0;
  s->rho_update_late    = 2.0;
  s->eta_user           = 1e-1;
  s->eta_direct         = 0;
  s->update_threshold   = 0.0;
  s->maxeig             = ORC_DECIDE;
  s->maxeig_tol         = ORC_DECIDE;
  s->maxeig_iter        = -1;
  s->inject_maxeig      = 0;
  s->inject_maxeig_set  = 0;
  s->inner_iter_min     = 1;
  s->inner_no_gtol_stop = 0;
  s->inner_max_it       = 10000;
  s->norm_update        = 0;
  s->lag_offset         = 0; /* never assigned in QPSCreate_SMALXE: PetscNew zero */
  s->Jstart             = 10;
  s->Jstep              = 5;
  s->Jend               = 20;
  s->lower              = 0.1;
  s->upper              = 1.1;
  s->knoll              = 0;
  s->state              = 1;
  s->normBu = s->normBu_old = s->enorm = NAN;
  orc_qps_init(&s->inner);
}

/* QPSSMALXEUpdateNormBu_SMALXE smalxe.c:247-261 (cE == NULL after homogenisation) */
static void smalxe_update_normBu_std(orc_smalxe *s, const double *u, double *normBu, double *enorm)
{
  orc_csr_mult((void *)s->pf->G, u, s->Bu);
  *normBu = v_norm2(s->pf->m, s->Bu);
  *enorm  = *normBu / s->rtol_E;
}

/* QPSSMALXEUpdateNormBu_SMALXEON smalxe.c:265-285: ||Bu|| = sqrt(u'B'Bu) from the penalised term alone */
static void smalxe_update_normBu_on(orc_smalxe *s, const double *u, double *normBu, double *enorm)
{
  double dot;
  orc_qppf_apply_GtG(s->pf, s->G_orthonormal, u, s->BtBu);
  dot     = v_dot(s->A->n, u, s->BtBu);
  *normBu = sqrt(dot);
  *enorm  = *normBu / s->rtol_E;
}

/* QPSSMALXEUpdateNormBu_Lag_SMALXEON smalxe.c:289-370: the exact norm every J-th inner iteration only, J growing from Jstart
   to Jend by Jstep while the norm stays within [lower, upper) of the last exact one */
static void smalxe_update_normBu_lag(orc_smalxe *s, const double *u, double *normBu, double *enorm)
{
  double normBu_approx, normBu_exact, enorm_exact, rdiff;
  if (s->inner.iteration <= s->lag_offset) {
    smalxe_update_normBu_on(s, u, &normBu_exact, &enorm_exact);
    s->lag_neval++;
    s->lag_normBu0 = normBu_exact;
    normBu_approx  = s->lag_normBu0;
    s->lag_J       = s->Jstart;
    s->lag_II      = 0;
  } else {
    if (s->lag_II == 0) {
      smalxe_update_normBu_on(s, u, &normBu_exact, &enorm_exact);
      s->lag_neval++;
      rdiff = fabs(normBu_exact / s->lag_normBu0);
      if (rdiff >= s->upper) {
        s->lag_II = 0;
        s->lag_J  = s->Jstart;
      } else if (rdiff < s->lower) {
        s->lag_II = 0;
        s->lag_J  = s->Jstart;
      } else {
        s->lag_II++;
      }
      s->lag_normBu0 = normBu_exact;
    } else {
      s->lag_II++;
    }
    normBu_approx = s->lag_normBu0;
  }
  s->lag_niter++;
  if (s->lag_II == s->lag_J) {
    s->lag_II = 0;
    if (s->lag_J < s->Jend) s->lag_J += s->Jstep;
  }
  *normBu = normBu_approx;
  *enorm  = *normBu / s->rtol_E;
}

/* smalxe->updateNormBu, chosen in QPSSetUp_SMALXE smalxe.c:878-886 */
static void smalxe_update_normBu(orc_smalxe *s, const double *u, double *normBu, double *enorm)
{
  if (s->norm_update == 2) smalxe_update_normBu_lag(s, u, normBu, enorm);
  else if (s->norm_update == 1) smalxe_update_normBu_on(s, u, normBu, enorm);
  else smalxe_update_normBu_std(s, u, normBu, enorm);
}

/* outer QPSConvergedDefault (qps.c:675-714) evaluated on the outer solver's fields */
static int smalxe_outer_converged(orc_smalxe *s)
{
  int reason = ORC_CONVERGED_ITERATING;
  if (!s->outer_cvg_setup) {
    s->outer_norm_rhs     = v_norm2(s->A->n, s->b);
    s->outer_ttol         = fmax(s->rtol * s->outer_norm_rhs, s->atol);
    s->outer_norm_rhs_div = s->outer_norm_rhs;
    s->outer_cvg_setup    = 1;
  }
  if (s->iteration > s->max_it) return ORC_DIVERGED_ITS;
  if (isnan(s->rnorm) || isinf(s->rnorm)) reason = ORC_DIVERGED_NANORINF;
  else if (s->rnorm <= s->outer_ttol) reason = (s->rnorm < s->atol) ? ORC_CONVERGED_ATOL : ORC_CONVERGED_RTOL;
  else if (s->rnorm >= s->divtol * s->outer_norm_rhs_div) reason = ORC_DIVERGED_DTOL;
  return reason;
}

/* QPSConverged_Inner_SMALXE smalxe.c:610-692 */
static void smalxe_inner_converged(orc_qps *in, void *ctx)
{
  orc_smalxe *s     = (orc_smalxe *)ctx;
  int         i     = in->iteration;
  double      gnorm = in->rnorm;
  in->reason        = ORC_CONVERGED_ITERATING;

  smalxe_update_normBu(s, in->x, &s->normBu, &s->enorm);
  s->rnorm   = fmax(s->enorm, gnorm);
  s->MNormBu = s->M1 * s->normBu;
  in->atol   = fmin(s->MNormBu, s->eta);

  if (i > in->max_it - s->inner_iter_accu) {
    in->reason = ORC_DIVERGED_ITS;
    s->reason  = ORC_DIVERGED_BREAKDOWN;
    return;
  }
  if (isnan(gnorm) || isinf(gnorm)) {
    in->reason = ORC_DIVERGED_NANORINF;
    s->reason  = ORC_DIVERGED_BREAKDOWN;
    return;
  }
  s->reason = smalxe_outer_converged(s);
  if (s->reason) {
    in->reason = (s->reason > 0) ? ORC_CONVERGED_HAPPY_BREAKDOWN : ORC_DIVERGED_BREAKDOWN;
    return;
  }
  if (gnorm < in->atol) {
    in->reason = ORC_CONVERGED_ATOL;
    if (s->MNormBu < s->eta) s->M1_hits++;
    else s->eta_hits++;
    return;
  }
  if (s->state == 3 && (i < s->inner_iter_min || s->inner_no_gtol_stop)) return;
  if (gnorm <= s->gtol) {
    if (in->rnorm > s->enorm) {
      /* skipping gtol criterion because G > E */
    } else {
      if (s->inner_no_gtol_stop < 2) in->reason = ORC_CONVERGED_RTOL;
      if (s->state != 3) s->state = 3;
    }
  }
}

/* QPSSetUp_SMALXE smalxe.c:772-888 */
int orc_smalxe_setup(orc_smalxe *s)
{
  int      n = s->A->n;
  double   maxeig_inner;
  orc_qps *in = &s->inner;

  s->BtBu    = orc_vec(n);
  s->Btmu    = orc_vec(n);
  s->b_inner = orc_vec(n);
  s->xwork   = orc_vec(n);
  s->Bu      = orc_vec(s->pf->m);

  s->eta = s->eta_user;
  if (!s->eta_direct) s->eta *= v_norm2(n, s->b);
  s->M1_initial = s->M1_user;
  if (!s->M1_direct) {
    if (s->maxeig == ORC_DECIDE) s->maxeig = orc_max_eigenvalue(s->A, s->maxeig_tol, s->maxeig_iter, NULL);
    s->M1_initial *= s->maxeig;
  }
  if (!s->rho_direct) {
    if (s->maxeig == ORC_DECIDE) s->maxeig = orc_max_eigenvalue(s->A, s->maxeig_tol, s->maxeig_iter, NULL);
    s->rho = s->rho_user * s->maxeig;
  } else {
    s->rho = s->rho_user;
  }
  /* QPTEnforceEqByPenalty(qp, rho, PETSC_TRUE) qptransform.c:329-410 => A_rho shell */
  s->pen.A           = s->A;
  s->pen.pf          = s->pf;
  s->pen.orthonormal = s->G_orthonormal;
  s->pen.rho         = s->rho;
  s->A_inner.mult    = orc_penalized_mult;
  s->A_inner.ctx     = &s->pen;
  s->A_inner.n       = n;
  v_copy(n, s->b, s->b_inner);

  in->A      = &s->A_inner;
  in->b      = s->b_inner;
  in->x      = s->u;
  in->qpc    = s->qpc;
  in->max_it = s->inner_max_it;

  maxeig_inner = fmax(s->rho, s->maxeig);
  if (!s->inject_maxeig_set) s->inject_maxeig = s->G_orthonormal; /* QPPFGetGHasOrthonormalRows smalxe.c:866 */
  if (s->inject_maxeig) in->maxeig = maxeig_inner;
  orc_mpgp_setup(in);
  in->converged     = smalxe_inner_converged;
  in->converged_ctx = s;
  return 0;
}

/* QPSSMALXEUpdate_SMALXE smalxe.c:439-488 + ...UpdateRho :373-398 */
static void smalxe_update(orc_smalxe *s, double Lag_old, double Lag, double rho)
{
  double t, t2, rho_update;
  int    flag;
  t    = 0.5 * rho * s->normBu * s->normBu;
  t2   = Lag - (Lag_old + t);
  flag = (t2 < s->update_threshold);
  if (flag && s->M1_update != 1.0) {
    if (s->inner.reason == ORC_CONVERGED_ATOL) {
      s->M1 = s->M1 / s->M1_update;
      s->M1_updates++;
    }
  }
  if (s->inner.rnorm > s->enorm) return;
  rho_update = s->rho_update;
  if (s->state == 3) {
    rho_update = s->rho_update_late;
    flag       = 1;
  }
  if (!flag || rho_update == 1.0) return;
  s->pen.rho *= rho_update; /* MatPenalizedUpdatePenalty */
  /* QPSMPGPUpdateMaxEigenvalue mpgp.c:119-143 */
  s->inner.maxeig = s->inner.maxeig * rho_update;
  if (!s->inner.alpha_direct) s->inner.alpha = s->inner.alpha / rho_update;
  s->rho_updates++;
}

/* QPSSolve_SMALXE smalxe.c:893-997 */
int orc_smalxe_solve(orc_smalxe *s)
{
  int      n  = s->A->n, i, maxits = s->max_it;
  orc_qps *in = &s->inner;
  double   Lag, Lag_old, rho;

  s->M1 = s->M1_initial;
  rho   = s->pen.rho;
  v_set(n, s->Btmu, 0.0);
  if (s->knoll) orc_qppf_apply_P(s->pf, s->b, s->u); /* the Knoll trick smalxe.c:938-943: u = P b */
  Lag_old = orc_objective(&s->A_inner, s->b_inner, s->u, s->xwork);
  smalxe_update_normBu(s, s->u, &s->normBu_old, &s->enorm);

  s->iteration       = 0;
  s->inner_iter_accu = 0;
  s->reason          = ORC_CONVERGED_ITERATING;
  in->ncg = in->nexp = in->nmv = in->nprop = 0; /* QPSResetStatistics */

  for (i = 0; i < maxits; i++) {
    /* QPSSMALXEUpdateLambda_SMALXE smalxe.c:402-435 */
    orc_qppf_apply_GtG(s->pf, s->G_orthonormal, s->u, s->BtBu);
    v_axpy(n, s->Btmu, rho, s->BtBu);
    if (s->reason) break;
    v_waxpy(n, s->b_inner, -1.0, s->Btmu, s->b);
    in->divtol = s->divtol;
    /* QPSConvergedSetUp_Inner_SMALXE smalxe.c:537-557 */
    s->norm_rhs_outer     = v_norm2(n, s->b);
    s->gtol               = s->rtol * s->norm_rhs_outer;
    s->ttol_outer         = fmax(s->rtol * s->norm_rhs_outer, s->atol);
    s->outer_norm_rhs_div = v_norm2(n, s->b_inner);
    orc_mpgp_solve(in);
    s->inner_iter_accu += in->iteration;
    s->iteration = i + 1;
    smalxe_update_normBu(s, s->u, &s->normBu, &s->enorm);
    rho = s->pen.rho;
    Lag = orc_objective(&s->A_inner, s->b_inner, s->u, s->xwork);
    smalxe_update(s, Lag_old, Lag, rho);
    Lag_old       = Lag;
    s->normBu_old = s->normBu;
  }
  if (i == maxits && !s->reason) s->reason = ORC_DIVERGED_ITS;
  return 0;
}

void orc_smalxe_free(orc_smalxe *s)
{
  free(s->BtBu);
  free(s->Btmu);
  free(s->b_inner);
  free(s->xwork);
  free(s->Bu);
  orc_qps_free(&s->inner);
}

/* ------------------------------------------------------------------------------------------ */
/* QPSSolve_PCPG src/qps/impls/pcpg/pcpg.c:51-134                                               */
/* ------------------------------------------------------------------------------------------ */
int orc_pcpg_solve(orc_pcpg *s)
{
  int     n = s->A->n;
  double *p = orc_vec(n), *r = orc_vec(n), *w = orc_vec(n), *z = orc_vec(n), *yb = orc_vec(n), *Ap = orc_vec(n), *y;
  double  alpha, alpha1, beta, beta1 = 0, beta2, norm_rhs, ttol;

  norm_rhs = v_norm2(n, s->b);
  ttol     = fmax(s->rtol * norm_rhs, s->atol);
  s->A->mult(s->A->ctx, s->x, r);
  v_aypx(n, r, -1.0, s->b);
  s->iteration = 0;
  do {
    if (s->pf) orc_qppf_apply_P(s->pf, r, w);
    else v_copy(n, r, w); /* no eq. constraints: KSPCG as set up by QPSKSP (qpsksp.c:244-250), unpreconditioned norm */
    s->rnorm = v_norm2(n, w);
    /* QPSConvergedDefault */
    s->reason = ORC_CONVERGED_ITERATING;
    if (s->iteration > s->max_it) s->reason = ORC_DIVERGED_ITS;
    else if (isnan(s->rnorm) || isinf(s->rnorm)) s->reason = ORC_DIVERGED_NANORINF;
    else if (s->rnorm <= ttol) s->reason = (s->rnorm < s->atol) ? ORC_CONVERGED_ATOL : ORC_CONVERGED_RTOL;
    else if (s->rnorm >= s->divtol * norm_rhs) s->reason = ORC_DIVERGED_DTOL;
    if (s->reason) break;
    if (!s->pc) {
      y = w;
    } else {
      s->pc(s->pc_ctx, w, z);
      if (s->pf) orc_qppf_apply_P(s->pf, z, yb);
      else v_copy(n, z, yb);
      y = yb;
    }
    beta2 = beta1;
    beta1 = v_dot(n, y, w);
    if (!s->iteration) {
      beta = 0;
      v_copy(n, y, p);
    } else {
      beta = beta1 / beta2;
      v_aypx(n, p, beta, y);
    }
    s->A->mult(s->A->ctx, p, Ap);
    alpha1 = v_dot(n, p, Ap);
    alpha  = beta1 / alpha1;
    v_axpy(n, s->x, alpha, p);
    v_axpy(n, r, -alpha, Ap);
    s->iteration++;
  } while (s->iteration < s->max_it);
  (void)beta;
  free(p);
  free(r);
  free(w);
  free(z);
  free(yb);
  free(Ap);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* MATGLUING src/mat/impls/gluing/gluing.c:47-81 (x = B' lambda) and :125-159 (lambda = B x)     */
/* ------------------------------------------------------------------------------------------ */
void orc_gluing_mult(const orc_gluing *B, const double *lambda, double *x)
{
  int i;
  for (i = 0; i < B->n_x; i++) x[i] = 0.0;
  for (i = 0; i < B->n_leaves; i++) x[B->leaves_row[i]] += lambda[B->leaves_root[i]] * B->leaves_sign[i];
}

void orc_gluing_mult_transpose(const orc_gluing *B, const double *x, double *lambda)
{
  int i;
  for (i = 0; i < B->n_lambda; i++) lambda[i] = 0.0;
  for (i = 0; i < B->n_leaves; i++) lambda[B->leaves_root[i]] += x[B->leaves_row[i]] * B->leaves_sign[i];
}

/* ------------------------------------------------------------------------------------------ */
/* MatMult_Inv matinv.c:734-743 on the KSPCG/PCJACOBI per-block path (matinv.c:535-540)         */
/* ------------------------------------------------------------------------------------------ */
static void matinv_project(const orc_matinv *M, const double *v, double *out)
{
  int b, k, i;
  v_copy(M->K->nrows, v, out);
  for (b = 0; b < M->nblocks; b++) {
    int lo = M->rowstart[b], hi = M->rowstart[b + 1];
    for (k = 0; k < M->kdim; k++) {
      const double *Rk = M->R + (size_t)k * M->K->nrows;
      double        c  = v_dot(hi - lo, Rk + lo, v + lo);
      for (i = lo; i < hi; i++) out[i] -= c * Rk[i];
    }
  }
}

void orc_matinv_mult(orc_matinv *M, const double *f, double *u)
{
  int     n = M->K->nrows, nb = M->nblocks, b, i, it, nactive = 0, mx = 0;
  double *r = orc_vec(n), *z = orc_vec(n), *p = orc_vec(n), *Ap = orc_vec(n), *dinv = orc_vec(n), *t = orc_vec(n);
  double *rz = orc_vec(nb), *tol = orc_vec(nb);
  int    *active = (int *)calloc((size_t)nb, sizeof(int));
  for (i = 0; i < n; i++) {
    int    k;
    double d = 1.0;
    for (k = M->K->rowptr[i]; k < M->K->rowptr[i + 1]; k++)
      if (M->K->col[k] == i) d = M->K->val[k];
    dinv[i] = 1.0 / d;
  }
  if (M->kdim) matinv_project(M, f, r);
  else v_copy(n, f, r);
  for (i = 0; i < n; i++) {
    u[i] = 0.0;
    z[i] = dinv[i] * r[i];
    p[i] = z[i];
  }
  for (b = 0; b < nb; b++) {
    int    lo = M->rowstart[b], m = M->rowstart[b + 1] - lo;
    double rr = v_dot(m, r + lo, r + lo);
    rz[b]     = v_dot(m, r + lo, z + lo);
    tol[b]    = fmax(M->rtol * sqrt(rr), M->atol);
    active[b] = sqrt(rr) > tol[b];
    /* OPTION (off by default; not in the reference, whose K^+ is a factorisation): a load that lies in the kernel of K leaves only the rounding residue of its projection,
       which is NOT in the range of the singular K -- CG on it diverges along the kernel (feti/ex71.c's interior slabs under a uniform body force).  With kernel_tol = c such a
       block gets u_b = 0, as in the product (k_cg_init) */
    if (M->kdim && M->kernel_tol > 0.0 && sqrt(rr) <= M->kernel_tol * 2.220446049250313e-16 * sqrt(v_dot(m, f + lo, f + lo))) active[b] = 0;
    nactive += active[b];
  }
  for (it = 0; it < M->max_it && nactive; it++) {
    orc_csr_mult((void *)M->K, p, Ap);
    M->spmv_count++;
    for (b = 0; b < nb; b++) {
      int    lo = M->rowstart[b], m = M->rowstart[b + 1] - lo;
      double alpha, rzn, rr, beta;
      if (!active[b]) continue;
      alpha = rz[b] / v_dot(m, p + lo, Ap + lo);
      v_axpy(m, u + lo, alpha, p + lo);
      v_axpy(m, r + lo, -alpha, Ap + lo);
      for (i = lo; i < lo + m; i++) z[i] = dinv[i] * r[i];
      rzn   = v_dot(m, r + lo, z + lo);
      rr    = v_dot(m, r + lo, r + lo);
      beta  = rzn / rz[b];
      rz[b] = rzn;
      mx    = it + 1;
      if (sqrt(rr) <= tol[b]) {
        active[b] = 0;
        nactive--;
      } else {
        v_aypx(m, p + lo, beta, z + lo);
      }
    }
  }
  M->last_max_its = mx;
  if (M->kdim) {
    matinv_project(M, u, t);
    v_copy(n, t, u);
  }
  free(r), free(z), free(p), free(Ap), free(dinv), free(t), free(rz), free(tol), free(active);
}

void orc_feti_dual_mult(void *ctx, const double *x, double *y)
{
  orc_feti *F = (orc_feti *)ctx;
  orc_gluing_mult(F->B, x, F->t1);
  orc_matinv_mult(F->Kplus, F->t1, F->t2);
  orc_gluing_mult_transpose(F->B, F->t2, y);
}

void orc_feti_penalized_mult(void *ctx, const double *x, double *y)
{
  orc_feti *F = (orc_feti *)ctx;
  int       n = F->B->n_lambda, i;
  orc_qppf_apply_Q(F->pf, x, y); /* Q x (orthonormal G: BtB = Q) */
  for (i = 0; i < n; i++) F->w1[i] = x[i] - y[i]; /* P x */
  orc_feti_dual_mult(ctx, F->w1, F->w2);
  orc_qppf_apply_P(F->pf, F->w2, F->w1);
  for (i = 0; i < n; i++) y[i] = F->rho * y[i] + F->w1[i];
}

/* ------------------------------------------------------------------------------------------ */
/* MatRegularize, src/mat/interface/permonmatregularize.c (TFETI "fixing DOFs" regularisation   */
/* of a singular subdomain stiffness matrix; default of QPTFromOptions: -regularize 1,          */
/* qptransform.c:2215,2231 -> MAT_REG_EXPLICIT).                                                */
/* ------------------------------------------------------------------------------------------ */
/* MatRegularize_GetPivots_Private :6-116.  R: p x d column-major (R_loc, the kernel basis of the block).
   Complete pivoting from the last column / row backwards; the search is column-outer, row-inner with a strict
   '>' (first maximum wins); rows are swapped only in columns 0..J, columns only in rows 0..II; the remaining
   columns are combined so that row II vanishes (columns whose entry is < eps are skipped).  pivots = the last d
   entries of the row permutation, sorted ascending (ISSort). */
void orc_regularize_pivots(int p, int d, const double *R, int *pivots)
{
  double *W = (double *)malloc(sizeof(double) * (size_t)p * (size_t)(d > 0 ? d : 1));
  int    *perm = (int *)malloc(sizeof(int) * (size_t)(p > 0 ? p : 1));
  int     i, j, J, II, ipivot = 0, jpivot = 0, t;
  memcpy(W, R, sizeof(double) * (size_t)p * (size_t)d);
  for (i = 0; i < p; i++) perm[i] = i;
#define RW(i, j) W[(size_t)(j) * (size_t)p + (size_t)(i)]
  for (J = d - 1, II = p - 1; J >= 0; J--, II--) {
    double vpivot = 0.0;
    for (j = 0; j <= J; j++)
      for (i = 0; i <= II; i++)
        if (fabs(RW(i, j)) > fabs(vpivot)) {
          ipivot = i;
          jpivot = j;
          vpivot = RW(i, j);
        }
    for (j = 0; j <= J; j++) { /* swap rows ipivot and II */
      double a = RW(ipivot, j);
      RW(ipivot, j) = RW(II, j);
      RW(II, j)     = a;
    }
    t            = perm[ipivot];
    perm[ipivot] = perm[II];
    perm[II]     = t;
    for (i = 0; i <= II; i++) { /* swap columns jpivot and J */
      double a = RW(i, jpivot);
      RW(i, jpivot) = RW(i, J);
      RW(i, J)      = a;
    }
    for (j = 0; j <= J - 1; j++) { /* v2 = -(vpivot/v2(II)) * v2 + v1, v1 = the pivot column (now column J) */
      double alpha;
      if (fabs(RW(II, j)) < ORC_EPS) continue;
      alpha = -vpivot / RW(II, j);
      for (i = 0; i <= II; i++) {
        double v = RW(i, j);
        v *= alpha;
        v += RW(i, J);
        RW(i, j) = v;
      }
    }
  }
#undef RW
  for (i = 0; i < d; i++) pivots[i] = perm[p - d + i];
  for (i = 1; i < d; i++) { /* ISSort */
    int v = pivots[i];
    for (j = i - 1; j >= 0 && pivots[j] > v; j--) pivots[j + 1] = pivots[j];
    pivots[j + 1] = v;
  }
  free(W), free(perm);
}

/* MatRegularize_GetRegularization_Private :118-160: Q_loc_condensed = RI (RI' RI)^{-1} RI' with RI = R(pivots,:),
   then MatFilterZeros(.,10 eps) (entries with |q| <= 10 eps are dropped, permonmatutils.c:547).  Q: d x d row-major,
   dropped entries are exact zeros and keep[] (d*d) flags the stored ones.  (RI'RI)^{-1} is the reference's
   MatInvExplicitly of a dense SPD matrix (a PETSc factorisation, third party): restated with a Cholesky solve, so
   values agree to rounding only.  Returns non-zero if RI'RI is not positive definite. */
int orc_regularization_Q(int p, int d, const double *R, const int *pivots, double *Q, int *keep)
{
  double *RI = (double *)malloc(sizeof(double) * (size_t)(d * d + 1)), *M = (double *)malloc(sizeof(double) * (size_t)(d * d + 1));
  double *Minv = (double *)malloc(sizeof(double) * (size_t)(d * d + 1)), *T = (double *)malloc(sizeof(double) * (size_t)(d * d + 1));
  double *e = (double *)malloc(sizeof(double) * (size_t)(d + 1)), *c = (double *)malloc(sizeof(double) * (size_t)(d + 1));
  int     i, j, k, rc;
  for (i = 0; i < d; i++)
    for (j = 0; j < d; j++) RI[i * d + j] = R[(size_t)j * (size_t)p + (size_t)pivots[i]];
  for (i = 0; i < d; i++) /* RItRI = RIt * RI */
    for (j = 0; j < d; j++) {
      double s = 0.0;
      for (k = 0; k < d; k++) s += RI[k * d + i] * RI[k * d + j];
      M[i * d + j] = s;
    }
  rc = orc_dense_cholesky(d, M);
  if (!rc) {
    for (j = 0; j < d; j++) { /* explicit inverse, column by column */
      for (i = 0; i < d; i++) e[i] = (i == j) ? 1.0 : 0.0;
      orc_dense_chol_solve(d, M, e, c);
      for (i = 0; i < d; i++) Minv[i * d + j] = c[i];
    }
    for (i = 0; i < d; i++) /* RI_invRItRI = RI * invRItRI */
      for (j = 0; j < d; j++) {
        double s = 0.0;
        for (k = 0; k < d; k++) s += RI[i * d + k] * Minv[k * d + j];
        T[i * d + j] = s;
      }
    for (i = 0; i < d; i++) /* Q = RI_invRItRI * RIt */
      for (j = 0; j < d; j++) {
        double s = 0.0;
        for (k = 0; k < d; k++) s += T[i * d + k] * RI[j * d + k];
        keep[i * d + j] = fabs(s) > 10.0 * ORC_EPS;
        Q[i * d + j]    = keep[i * d + j] ? s : 0.0;
      }
  }
  free(RI), free(M), free(Minv), free(T), free(e), free(c);
  return rc;
}

/* MatRegularize :198-287, MAT_REG_EXPLICIT: Kreg_loc = K_loc + rho * (rho * Q_loc) -- MatScale(Q_loc,rho) followed by
   MatAXPY(Kreg_loc,rho,Q_loc,DIFFERENT_NONZERO_PATTERN): the regulariser is scaled by rho TWICE (:256,265; kept).
   rho is the caller's MatGetMaxEigenvalue(K_loc,NULL,&rho,1,20) (:254).  Output CSR = union pattern, sorted columns;
   arrays sized nnz(K) + d*d.  Returns the number of stored entries, or -1 if RI'RI is not positive definite. */
int orc_regularize_csr(const orc_csr *K, int d, const double *R, double rho, int *pivots, int *rowptr_out, int *col_out, double *val_out)
{
  int     n = K->nrows, i, j, k, nnz = 0;
  double *Q    = (double *)malloc(sizeof(double) * (size_t)(d * d + 1));
  int    *keep = (int *)malloc(sizeof(int) * (size_t)(d * d + 1));
  int    *prow = (int *)malloc(sizeof(int) * (size_t)(n + 1)); /* row -> index in pivots or -1 */
  for (i = 0; i < n; i++) prow[i] = -1;
  if (d > 0) {
    orc_regularize_pivots(n, d, R, pivots);
    if (orc_regularization_Q(n, d, R, pivots, Q, keep)) {
      free(Q), free(keep), free(prow);
      return -1;
    }
    for (i = 0; i < d; i++) prow[pivots[i]] = i;
  }
  rowptr_out[0] = 0;
  for (i = 0; i < n; i++) {
    int a = K->rowptr[i], b = K->rowptr[i + 1], pi = prow[i];
    j = 0; /* merge the sorted row of K with the (sorted) pivot columns of row pi of Q */
    for (k = a; k < b || (pi >= 0 && j < d);) {
      int cq = (pi >= 0 && j < d) ? pivots[j] : 0x7fffffff;
      int ck = (k < b) ? K->col[k] : 0x7fffffff;
      if (pi >= 0 && j < d && !keep[pi * d + j]) {
        j++;
        continue;
      }
      if (ck < cq) {
        col_out[nnz] = ck, val_out[nnz] = K->val[k], k++;
      } else if (ck == cq) {
        col_out[nnz] = ck, val_out[nnz] = K->val[k] + rho * (Q[pi * d + j] * rho), k++, j++;
      } else {
        col_out[nnz] = cq, val_out[nnz] = rho * (Q[pi * d + j] * rho), j++;
      }
      nnz++;
    }
    rowptr_out[i + 1] = nnz;
  }
  free(Q), free(keep), free(prow);
  return nnz;
}
