/*
 * orc_api.c -- flat, string-keyed accessors over the oracle's structs so the Python test harness
 * (oracle/oracle.py, ctypes) never has to mirror the C struct layouts.  TEST INFRASTRUCTURE ONLY.
 */
#include "permon_oracle.h"

#include <stdlib.h>
#include <string.h>

orc_qps *orc_qps_new(void)
{
  orc_qps *q = (orc_qps *)malloc(sizeof(orc_qps));
  orc_qps_init(q);
  return q;
}

void orc_qps_delete(orc_qps *q)
{
  orc_qps_free(q);
  free(q->trace_step);
  free(q->trace_rnorm);
  free(q->trace_gfnorm);
  free(q->trace_gcnorm);
  free(q->trace_alpha);
  free(q);
}

void orc_qps_set_problem(orc_qps *q, const orc_op *A, const double *b, double *x, const orc_box *qpc)
{
  q->A   = A;
  q->b   = b;
  q->x   = x;
  q->qpc = qpc;
}

void orc_qps_enable_trace(orc_qps *q, int cap)
{
  q->trace_cap    = cap;
  q->trace_step   = (char *)calloc((size_t)cap + 1, 1);
  q->trace_rnorm  = (double *)calloc((size_t)cap, sizeof(double));
  q->trace_gfnorm = (double *)calloc((size_t)cap, sizeof(double));
  q->trace_gcnorm = (double *)calloc((size_t)cap, sizeof(double));
  q->trace_alpha  = (double *)calloc((size_t)cap, sizeof(double));
}

const char   *orc_qps_trace_steps(orc_qps *q) { return q->trace_step; }
const double *orc_qps_trace_array(orc_qps *q, int which)
{
  switch (which) {
  case 0: return q->trace_rnorm;
  case 1: return q->trace_gfnorm;
  case 2: return q->trace_gcnorm;
  case 3: return q->trace_alpha;
  }
  return NULL;
}
const double *orc_qps_work(orc_qps *q, int i) { return q->work[i]; }

#define QSETD(name) \
  if (!strcmp(key, #name)) { \
    q->name = v; \
    return 0; \
  }
#define QSETI(name) \
  if (!strcmp(key, #name)) { \
    q->name = (int)v; \
    return 0; \
  }
int orc_qps_set(orc_qps *q, const char *key, double v)
{
  QSETD(rtol) QSETD(atol) QSETD(divtol) QSETI(max_it) QSETD(alpha_user) QSETI(alpha_direct) QSETD(gamma) QSETD(maxeig) QSETD(maxeig_tol)
    QSETI(maxeig_iter) QSETD(bchop_tol) QSETI(exptype) QSETI(explengthtype) QSETI(resetalpha) QSETI(fallback) QSETI(fallback2) return 1;
}

#define QGET(name) \
  if (!strcmp(key, #name)) return (double)q->name;
double orc_qps_get(orc_qps *q, const char *key)
{
  QGET(rtol) QGET(atol) QGET(divtol) QGET(max_it) QGET(alpha_user) QGET(alpha) QGET(gamma) QGET(maxeig) QGET(rnorm) QGET(gfnorm) QGET(gcnorm)
    QGET(iteration) QGET(reason) QGET(nmv) QGET(ncg) QGET(nexp) QGET(nprop) QGET(nfinc) QGET(nfall) QGET(trace_len) QGET(norm_rhs) QGET(ttol)
      QGET(expproject) return -12345.678;
}

orc_smalxe *orc_smalxe_new(void)
{
  orc_smalxe *s = (orc_smalxe *)malloc(sizeof(orc_smalxe));
  orc_smalxe_init(s);
  return s;
}

void orc_smalxe_delete(orc_smalxe *s)
{
  orc_smalxe_free(s);
  free(s);
}

void orc_smalxe_set_problem(orc_smalxe *s, const orc_op *A, const double *b, double *u, const orc_box *qpc, const orc_qppf *pf, int G_orthonormal)
{
  s->A             = A;
  s->b             = b;
  s->u             = u;
  s->qpc           = qpc;
  s->pf            = pf;
  s->G_orthonormal = G_orthonormal;
}

orc_qps *orc_smalxe_inner(orc_smalxe *s) { return &s->inner; }

#define SSETD(name) \
  if (!strcmp(key, #name)) { \
    s->name = v; \
    return 0; \
  }
#define SSETI(name) \
  if (!strcmp(key, #name)) { \
    s->name = (int)v; \
    return 0; \
  }
int orc_smalxe_set(orc_smalxe *s, const char *key, double v)
{
  SSETD(rtol) SSETD(atol) SSETD(divtol) SSETI(max_it) SSETD(M1_user) SSETI(M1_direct) SSETD(M1_update) SSETD(rtol_E) SSETD(rho_user) SSETI(rho_direct)
    SSETD(rho_update) SSETD(rho_update_late) SSETD(eta_user) SSETI(eta_direct) SSETD(update_threshold) SSETD(maxeig) SSETD(maxeig_tol) SSETI(maxeig_iter)
      SSETI(inject_maxeig) SSETI(inject_maxeig_set) SSETI(inner_iter_min) SSETI(inner_no_gtol_stop) SSETI(inner_max_it) SSETI(norm_update) SSETI(lag_offset) SSETI(Jstart) SSETI(Jstep) SSETI(Jend) SSETD(lower) SSETD(upper) SSETI(knoll) return 1;
}

#define SGET(name) \
  if (!strcmp(key, #name)) return (double)s->name;
double orc_smalxe_get(orc_smalxe *s, const char *key)
{
  SGET(M1) SGET(M1_initial) SGET(eta) SGET(rho) SGET(M1_updates) SGET(M1_hits) SGET(eta_hits) SGET(rho_updates) SGET(state) SGET(inner_iter_accu)
    SGET(normBu) SGET(enorm) SGET(rnorm) SGET(iteration) SGET(reason) SGET(maxeig) SGET(gtol) SGET(lag_neval) SGET(lag_niter) if (!strcmp(key, "rho_current")) return s->pen.rho;
  return -12345.678;
}

orc_pcpg *orc_pcpg_new(const orc_op *A, const double *b, double *x, const orc_qppf *pf, double rtol, double atol, double divtol, int max_it)
{
  orc_pcpg *s = (orc_pcpg *)calloc(1, sizeof(orc_pcpg));
  s->A        = A;
  s->b        = b;
  s->x        = x;
  s->pf       = pf;
  s->rtol     = rtol;
  s->atol     = atol;
  s->divtol   = divtol;
  s->max_it   = max_it;
  return s;
}
void   orc_pcpg_set_pc(orc_pcpg *s, orc_pc_fn pc, void *ctx) { s->pc = pc, s->pc_ctx = ctx; }
double orc_pcpg_get(orc_pcpg *s, const char *key)
{
  SGET(rnorm) SGET(iteration) SGET(reason) return -12345.678;
}
void orc_pcpg_delete(orc_pcpg *s) { free(s); }

/* CPU-baseline helper: time `reps` full MPGP solves capped at `max_it` iterations each on a CSR
   operator; returns seconds of the solve phase only (power method excluded, done in setup). */
double orc_time_mpgp(orc_qps *q, const double *x0, int reps, int *iters_out)
{
  int    n = q->A->n, r, its = 0;
  double t0, t = 0.0;
  orc_mpgp_setup(q);
  for (r = 0; r < reps; r++) {
    memcpy(q->x, x0, (size_t)n * sizeof(double));
    q->cvg_setup_called = 0;
    t0                  = orc_now();
    orc_mpgp_solve(q);
    t += orc_now() - t0;
    its += q->iteration;
  }
  if (iters_out) *iters_out = its;
  return t;
}

/* time `reps` CSR SpMVs */
double orc_time_spmv(const orc_csr *A, const double *x, double *y, int reps)
{
  int    r;
  double t0 = orc_now();
  for (r = 0; r < reps; r++) orc_csr_mult((void *)A, x, y);
  return orc_now() - t0;
}

orc_matinv *orc_matinv_new(const orc_csr *K, int nblocks, const int *rowstart, int kdim, const double *R, double rtol, double atol, int max_it)
{
  orc_matinv *M = (orc_matinv *)calloc(1, sizeof(orc_matinv));
  M->K = K, M->nblocks = nblocks, M->rowstart = rowstart, M->kdim = kdim, M->R = R, M->rtol = rtol, M->atol = atol, M->max_it = max_it;
  return M;
}
void      orc_matinv_set_kernel_tol(orc_matinv *M, double c) { M->kernel_tol = c; }
long long orc_matinv_spmv_count(orc_matinv *M) { return M->spmv_count; }
int       orc_matinv_last_its(orc_matinv *M) { return M->last_max_its; }
void      orc_matinv_delete(orc_matinv *M) { free(M); }

orc_feti *orc_feti_new(const orc_gluing *B, orc_matinv *Kplus, const orc_qppf *pf, double rho)
{
  orc_feti *F = (orc_feti *)calloc(1, sizeof(orc_feti));
  F->B = B, F->Kplus = Kplus, F->pf = pf, F->rho = rho;
  F->t1 = (double *)calloc((size_t)B->n_x + 1, sizeof(double));
  F->t2 = (double *)calloc((size_t)B->n_x + 1, sizeof(double));
  F->w1 = (double *)calloc((size_t)B->n_lambda + 1, sizeof(double));
  F->w2 = (double *)calloc((size_t)B->n_lambda + 1, sizeof(double));
  return F;
}
void orc_feti_delete(orc_feti *F)
{
  free(F->t1), free(F->t2), free(F->w1), free(F->w2), free(F);
}
orc_mult_fn orc_feti_fn(int which) { return which ? orc_feti_penalized_mult : orc_feti_dual_mult; }
