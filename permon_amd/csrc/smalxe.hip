// QPS SMALXE (src/qps/impls/smalxe/smalxe.c) and QPS PCPG (src/qps/impls/pcpg/pcpg.c) on gfx950.
// Outer loops are host logic over device-resident vectors; all arithmetic is in the kernels of vec.hip,
// spmv.hip, qppf.hip and the inner MPGP driver (mpgp.hip).
#include <cmath>

#include "pmh_internal.h"

struct pmh_smalxe_s {
  pmh_ctx         ctx;
  pmh_op          A;
  const double   *b;
  double         *u;
  const double   *lb, *ub;
  pmh_qppf        pf;
  pmh_smalxe_opts o;
  int             n;
  // QPS_SMALXE state (smalxeimpl.h:13-67)
  double M1, M1_initial, eta, maxeig;
  int    M1_updates, M1_hits, eta_hits, rho_updates;
  int    state, inner_iter_accu;
  double normBu, normBu_old, enorm;
  double rnorm;
  int    iteration, reason;
  // inner solver + penalised QP
  pmh_op   A_inner;
  pmh_mpgp inner;
  double  *Btmu, *b_inner, *BtBu, *Bu, *xwork;
  int      normBu_prefetched;
  // QPSConvergedCtx_Inner_SMALXE + outer QPSConvergedDefaultCtx
  double gtol, ttol_outer, norm_rhs_outer, MNormBu;
  double outer_norm_rhs, outer_ttol, outer_norm_rhs_div;
  int    outer_cvg_setup;
  double inner_atol;
  int    inner_reason, inner_max_it;
  double inner_rnorm;
  // ||Bu|| update variants (smalxe.c:265-370): BtBu reuse flag (QPSWorkVecStateChanged, smalxe.c:421-430), the inner solver's
  // current iteration (qps_inner->iteration as the lagged update reads it) and the function-static state of the lagged update
  int    BtBu_valid, inner_it_now;
  double lag_normBu0;
  int    lag_II, lag_J, lag_neval, lag_niter;
  // pmh_smalxe_set_reuse_products: A_rho u is carried from the inner solve's last gradient into the Lagrangian and into the next inner solve's first gradient
  int reuse;
  int normBu_final_valid; // normBu / enorm are those of the current u (set by the inner convergence test)
};

// QPSCreate_SMALXE defaults smalxe.c:1159-1207
extern "C" int pmh_smalxe_default_opts(pmh_smalxe_opts *o)
{
  PMH_ARG(o);
  memset(o, 0, sizeof(*o));
  o->rtol               = 1e-5;
  o->atol               = 1e-50;
  o->divtol             = 1e4;
  o->max_it             = 100;
  o->M1_user            = 1e2;
  o->M1_direct          = 0;
  o->M1_update          = 2.0;
  o->rtol_E             = 1e-0;
  o->rho_user           = 1.1;
  o->rho_direct         = 0;
  o->rho_update         = 1.0;
  o->rho_update_late    = 2.0;
  o->eta_user           = 1e-1;
  o->eta_direct         = 0;
  o->update_threshold   = 0.0;
  o->maxeig             = PMH_DECIDE;
  o->maxeig_tol         = PMH_DECIDE;
  o->maxeig_iter        = -1;
  o->inject_maxeig      = 0;
  o->inject_maxeig_set  = 0;
  o->inner_iter_min     = 1;
  o->inner_no_gtol_stop = 0;
  o->be_implicit        = 0;   // BE has a mult slot (smalxe.c:878-879)
  o->lag_enabled        = 0;   // smalxe.c:1192
  o->lag_offset         = 0;   // norm_update_lag_offset is never assigned in QPSCreate_SMALXE (PetscNew zero)
  o->lag_start          = 10;  // Jstart, Jstep, Jend, lower, upper: smalxe.c:1196-1200
  o->lag_step           = 5;
  o->lag_end            = 20;
  o->lag_lower          = 0.1;
  o->lag_upper          = 1.1;
  o->knoll              = 0;   // smalxe.c:1202
  return pmh_mpgp_default_opts(&o->inner);
}

// QPSSMALXEUpdateNormBu_SMALXE smalxe.c:247-261 (cE is homogenised away before SMALXE, smalxe.c:782-787)
static int update_normBu_std(pmh_smalxe s, const double *u, double *normBu, double *enorm)
{
  if (s->pf->m > 0 && s->normBu_prefetched && u == s->u) {
    // prefetch_normBu enqueued B u and its squared norm before the inner solver waited for this step's scalars: same kernels, same value, no second round trip
    *normBu              = sqrt(s->ctx->h_scal[PMH_SLOT_NORMBU2]);
    s->normBu_prefetched = 0;
  } else if (s->pf->m > 0) {
    PMH_CHK(pmh_qppf_apply_G(s->pf, u, s->Bu));
    PMH_CHK(pmh_vec_norm2(s->ctx, s->pf->m, s->Bu, normBu));
  } else {
    *normBu = 0.0;
  }
  *enorm = *normBu / s->o.rtol_E;
  return PMH_SUCCESS;
}

// pre-test hook of the inner MPGP (pmh_mpgp_set_pre_test_hook): the standard ||B u|| of the inner convergence test, enqueued behind the step
static int prefetch_normBu(void *user)
{
  pmh_smalxe s = (pmh_smalxe)user;
  if (s->o.be_implicit || s->pf->m == 0 || !pmh_knobs().smalxe_prefetch) return PMH_SUCCESS;
  // the kernel that wrote u emitted G0 u; its last workgroup left T G0 u and the squared norm in place (dualchain.hip)
  if (pmh_op_penalized_normG_ready(s->A_inner, s->u)) {
    s->normBu_prefetched = 1;
    return PMH_SUCCESS;
  }
  // it rode on the speculative A_rho p of this iteration (arm_normBu below): already in the scalar slot, same bits
  if (pmh_op_penalized_take_aux_done(s->A_inner)) {
    s->normBu_prefetched = 1;
    return PMH_SUCCESS;
  }
  PMH_CHK(pmh_qppf_apply_G_norm2(s->pf, s->u, s->Bu, PMH_SLOT_NORMBU2));
  s->normBu_prefetched = 1;
  return PMH_SUCCESS;
}

// pre-P1 hook (pmh_mpgp_set_pre_p1_hook): the iterate u is final when the inner MPGP enqueues the next A_rho p; G0 u then shares the pass over G0 with the
// projector's G0 p and T G0 u / its squared norm are finished inside the projector's kernel (the one-launch projector form with m <= 64 only -- the condition
// under which pmh_qppf_apply_G_norm2 takes the same two kernels on its own)
static int arm_normBu(void *user)
{
  pmh_smalxe s = (pmh_smalxe)user;
  if (s->o.be_implicit || s->pf->m == 0 || !(s->pf->implicit_orth && s->pf->m <= 64) || !pmh_knobs().smalxe_prefetch) return PMH_SUCCESS;
  return pmh_op_penalized_arm_aux_normG(s->A_inner, s->u, s->Bu, PMH_SLOT_NORMBU2);
}

// QPSSMALXEUpdateNormBu_SMALXEON smalxe.c:265-285: only the penalised term B'B is available; ||Bu|| = sqrt(u'B'Bu).
// BtBu stays valid for this u (QPSWorkVecStateUpdate :278-279) and is reused by the next lambda update.
static int update_normBu_on(pmh_smalxe s, const double *u, double *normBu, double *enorm)
{
  double dot;
  PMH_CHK(pmh_qppf_apply_GtG(s->pf, u, s->BtBu));
  s->BtBu_valid = (u == s->u);
  PMH_CHK(pmh_vec_dot(s->ctx, s->n, u, s->BtBu, &dot));
  *normBu = sqrt(dot);
  *enorm  = *normBu / s->o.rtol_E;
  return PMH_SUCCESS;
}

// QPSSMALXEUpdateNormBu_Lag_SMALXEON smalxe.c:289-370 (-qps_smalxe_norm_update_lag): the exact norm every J-th inner
// iteration only; J grows from lag_start to lag_end by lag_step while the norm stays within [lower, upper) of the last exact one
static int update_normBu_lag(pmh_smalxe s, const double *u, double *normBu, double *enorm)
{
  double normBu_approx, normBu_exact, enorm_exact;
  if (s->inner_it_now <= s->o.lag_offset) {
    PMH_CHK(update_normBu_on(s, u, &normBu_exact, &enorm_exact));
    s->lag_neval++;
    s->lag_normBu0 = normBu_exact;
    normBu_approx  = s->lag_normBu0;
    s->lag_J       = s->o.lag_start;
    s->lag_II      = 0;
  } else {
    if (s->lag_II == 0) {
      PMH_CHK(update_normBu_on(s, u, &normBu_exact, &enorm_exact));
      s->lag_neval++;
      const double rdiff = fabs(normBu_exact / s->lag_normBu0);
      if (rdiff >= s->o.lag_upper || rdiff < s->o.lag_lower) {
        s->lag_II = 0;
        s->lag_J  = s->o.lag_start;
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
    if (s->lag_J < s->o.lag_end) s->lag_J += s->o.lag_step;
  }
  *normBu = normBu_approx;
  *enorm  = *normBu / s->o.rtol_E;
  return PMH_SUCCESS;
}

// smalxe->updateNormBu as QPSSetUp_SMALXE picks it (smalxe.c:878-886)
static int update_normBu(pmh_smalxe s, const double *u, double *normBu, double *enorm)
{
  if (!s->o.be_implicit || s->pf->m == 0) return update_normBu_std(s, u, normBu, enorm);
  return s->o.lag_enabled ? update_normBu_lag(s, u, normBu, enorm) : update_normBu_on(s, u, normBu, enorm);
}

// the outer solver's QPSConvergedDefault (qps.c:675-714)
static int outer_converged(pmh_smalxe s, int *reason)
{
  if (!s->outer_cvg_setup) {
    PMH_CHK(pmh_vec_norm2(s->ctx, s->n, s->b, &s->outer_norm_rhs));
    s->outer_ttol         = fmax(s->o.rtol * s->outer_norm_rhs, s->o.atol);
    s->outer_norm_rhs_div = s->outer_norm_rhs;
    s->outer_cvg_setup    = 1;
  }
  *reason = PMH_CONVERGED_ITERATING;
  if (s->iteration > s->o.max_it) {
    *reason = PMH_DIVERGED_ITS;
    return PMH_SUCCESS;
  }
  if (std::isnan(s->rnorm) || std::isinf(s->rnorm)) *reason = PMH_DIVERGED_NANORINF;
  else if (s->rnorm <= s->outer_ttol) *reason = (s->rnorm < s->o.atol) ? PMH_CONVERGED_ATOL : PMH_CONVERGED_RTOL;
  else if (s->rnorm >= s->o.divtol * s->outer_norm_rhs_div) *reason = PMH_DIVERGED_DTOL;
  return PMH_SUCCESS;
}

// QPSConverged_Inner_SMALXE smalxe.c:610-692, injected into the inner MPGP (smalxe.c:874-875)
static int inner_converged(void *user, int i, double gnorm, int *reason)
{
  pmh_smalxe s = (pmh_smalxe)user;
  *reason      = PMH_CONVERGED_ITERATING;
  s->inner_rnorm = gnorm;
  s->inner_it_now = i;
  s->BtBu_valid   = 0; // the inner solver has moved u
  if (update_normBu(s, s->u, &s->normBu, &s->enorm)) return 1;
  s->normBu_final_valid = 1; // (normBu / enorm are those of the current u; the next step of the inner solver is followed by another test)
  s->rnorm      = fmax(s->enorm, gnorm);
  s->MNormBu    = s->M1 * s->normBu;
  s->inner_atol = fmin(s->MNormBu, s->eta);
  {
    // how far the inner solve is from its end (the inner MPGP does not enqueue the next A_rho p ahead of a test that is likely to end the solve): the larger of
    // the two thresholds the norm has to fall below; the iteration budget of the throughput mode and the iteration limit end it for certain
    const bool   gtol_on = !(s->state == 3 && (i < s->o.inner_iter_min || s->o.inner_no_gtol_stop));
    const double thr     = fmax(s->inner_atol, gtol_on ? s->gtol : 0.0);
    double       margin  = gnorm / fmax(thr, 1e-300);
    if (i + 1 > s->inner_max_it - s->inner_iter_accu) margin = 1e-300;
    (void)pmh_mpgp_set_convergence_margin(s->inner, margin);
  }

  if (i > s->inner_max_it - s->inner_iter_accu) {
    *reason   = PMH_DIVERGED_ITS;
    s->reason = PMH_DIVERGED_BREAKDOWN;
    return 0;
  }
  if (std::isnan(gnorm) || std::isinf(gnorm)) {
    *reason   = PMH_DIVERGED_NANORINF;
    s->reason = PMH_DIVERGED_BREAKDOWN;
    return 0;
  }
  if (outer_converged(s, &s->reason)) return 1;
  if (s->reason) {
    *reason = (s->reason > 0) ? PMH_CONVERGED_HAPPY_BREAKDOWN : PMH_DIVERGED_BREAKDOWN;
    return 0;
  }
  if (gnorm < s->inner_atol) {
    *reason = PMH_CONVERGED_ATOL;
    if (s->MNormBu < s->eta) s->M1_hits++;
    else s->eta_hits++;
    return 0;
  }
  if (s->state == 3 && (i < s->o.inner_iter_min || s->o.inner_no_gtol_stop)) return 0;
  if (gnorm <= s->gtol) {
    if (gnorm > s->enorm) {
      // skipping gtol criterion because G > E (smalxe.c:675-676)
    } else {
      if (s->o.inner_no_gtol_stop < 2) *reason = PMH_CONVERGED_RTOL;
      if (s->state != 3) s->state = 3;
    }
  }
  return 0;
}

// QPSSetUp_SMALXE smalxe.c:772-888
extern "C" int pmh_smalxe_create(pmh_ctx ctx, pmh_op A, const double *b, double *u, const double *lb, const double *ub, pmh_qppf pf, const pmh_smalxe_opts *o, pmh_smalxe *out)
{
  PMH_ARG(ctx && A && b && u && pf && o && out);
  PMH_ARG(pf->n == A->n);
  pmh_smalxe s = new pmh_smalxe_s();
  memset((void *)s, 0, sizeof(*s));
  s->ctx = ctx, s->A = A, s->b = b, s->u = u, s->lb = lb, s->ub = ub, s->pf = pf, s->o = *o, s->n = A->n;
  s->state  = 1;
  s->normBu = s->normBu_old = s->enorm = NAN;
  s->reuse      = 0;
  const int n = s->n;
  PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)n, (void **)&s->BtBu));
  PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)n, (void **)&s->Btmu));
  PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)n, (void **)&s->b_inner));
  PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)n, (void **)&s->xwork));
  PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)(pf->m ? pf->m : 1), (void **)&s->Bu));
  PMH_CHK(pmh_memset(ctx, s->Btmu, 0, sizeof(double) * (size_t)n));

  s->eta = s->o.eta_user;
  if (!s->o.eta_direct) {
    double normb;
    PMH_CHK(pmh_vec_norm2(ctx, n, b, &normb));
    s->eta *= normb;
  }
  s->maxeig     = s->o.maxeig;
  s->M1_initial = s->o.M1_user;
  if (!s->o.M1_direct) {
    if (s->maxeig == PMH_DECIDE) PMH_CHK(pmh_op_max_eigenvalue(A, s->o.maxeig_tol, s->o.maxeig_iter, &s->maxeig, nullptr));
    s->M1_initial *= s->maxeig;
  }
  double rho;
  if (!s->o.rho_direct) {
    if (s->maxeig == PMH_DECIDE) PMH_CHK(pmh_op_max_eigenvalue(A, s->o.maxeig_tol, s->o.maxeig_iter, &s->maxeig, nullptr));
    rho = s->o.rho_user * s->maxeig;
  } else {
    rho = s->o.rho_user;
  }
  // QPTEnforceEqByPenalty(qp, rho, PETSC_TRUE) qptransform.c:329-410: A_rho = A + rho*BE'*BE
  PMH_CHK(pmh_op_create_penalized(A, pf, rho, &s->A_inner));
  PMH_CHK(pmh_vec_copy(ctx, n, b, s->b_inner));

  // inner MPGP inherits max(rho, maxeig) when G has orthonormal rows (smalxe.c:864-868)
  pmh_mpgp_opts io       = s->o.inner;
  s->inner_max_it        = io.max_it;
  const double maxeig_in = fmax(rho, s->maxeig);
  int          inject    = s->o.inject_maxeig_set ? s->o.inject_maxeig : pf->orthonormal;
  if (inject) io.maxeig = maxeig_in;
  PMH_CHK(pmh_mpgp_create(ctx, s->A_inner, s->b_inner, u, lb, ub, &io, &s->inner));
  PMH_CHK(pmh_mpgp_set_convergence_test(s->inner, inner_converged, s));
  PMH_CHK(pmh_mpgp_set_pre_test_hook(s->inner, prefetch_normBu, s));
  PMH_CHK(pmh_mpgp_set_pre_p1_hook(s->inner, arm_normBu, s));
  if (!s->o.be_implicit && pf->m > 0 && pf->implicit_orth && pf->m <= 64) PMH_CHK(pmh_op_penalized_set_normG_target(s->A_inner, s->Bu, PMH_SLOT_NORMBU2));
  *out = s;
  return PMH_SUCCESS;
}

// Extension (off by default; the reference forms both quantities by a product of their own, smalxe.c:982 QPComputeObjective and mpgp.c:500 at the start of
// every inner solve): carry A_rho u from the last gradient of the inner solve -- two operator applications less per outer iteration, the same numbers up to the
// rounding of g's recurrence over the inner CG steps.  The count of Hessian multiplications then differs from the reference's.
extern "C" int pmh_smalxe_set_reuse_products(pmh_smalxe s, int on)
{
  PMH_ARG(s);
  s->reuse = on ? 1 : 0;
  return PMH_SUCCESS;
}

extern "C" int pmh_smalxe_destroy(pmh_smalxe s)
{
  if (!s) return PMH_SUCCESS;
  pmh_mpgp_destroy(s->inner);
  pmh_op_destroy(s->A_inner);
  pmh_free(s->ctx, s->BtBu);
  pmh_free(s->ctx, s->Btmu);
  pmh_free(s->ctx, s->b_inner);
  pmh_free(s->ctx, s->xwork);
  pmh_free(s->ctx, s->Bu);
  delete s;
  return PMH_SUCCESS;
}

// QPComputeObjective qp.c:913-927 on the penalised QP: f = -u'(b_inner - 1/2 A_rho u)
static int objective(pmh_smalxe s, double *f)
{
  PMH_CHK(s->A_inner->mult(s->u, s->xwork));
  PMH_CHK(pmh_vec_aypx(s->ctx, s->n, s->xwork, -0.5, s->b_inner));
  double dot;
  PMH_CHK(pmh_vec_dot(s->ctx, s->n, s->u, s->xwork, &dot));
  *f = -dot;
  return PMH_SUCCESS;
}

// QPSSMALXEUpdate_SMALXE smalxe.c:439-488 + QPSSMALXEUpdateRho_SMALXE :373-398
static int smalxe_update(pmh_smalxe s, double Lag_old, double Lag, double rho)
{
  double t    = 0.5 * rho * s->normBu * s->normBu;
  double t2   = Lag - (Lag_old + t);
  int    flag = (t2 < s->o.update_threshold);
  if (flag && s->o.M1_update != 1.0) {
    if (s->inner_reason == PMH_CONVERGED_ATOL) {
      s->M1 = s->M1 / s->o.M1_update;
      s->M1_updates++;
    }
  }
  if (s->inner_rnorm > s->enorm) return PMH_SUCCESS;
  double rho_update = s->o.rho_update;
  if (s->state == 3) {
    rho_update = s->o.rho_update_late;
    flag       = 1;
  }
  if (!flag || rho_update == 1.0) return PMH_SUCCESS;
  double r;
  PMH_CHK(pmh_op_penalized_get_penalty(s->A_inner, &r));
  PMH_CHK(pmh_op_penalized_set_penalty(s->A_inner, r * rho_update)); // MatPenalizedUpdatePenalty
  PMH_CHK(pmh_mpgp_update_max_eigenvalue(s->inner, rho_update));
  s->rho_updates++;
  return PMH_SUCCESS;
}

// QPSSolve_SMALXE smalxe.c:893-997
extern "C" int pmh_smalxe_solve(pmh_smalxe s)
{
  PMH_ARG(s);
  pmh_ctx ctx = s->ctx;
  const int n = s->n, maxits = s->o.max_it;
  double  Lag, Lag_old, rho;
  int     i;

  s->M1 = s->M1_initial;
  PMH_CHK(pmh_op_penalized_get_penalty(s->A_inner, &rho));
  PMH_CHK(pmh_memset(ctx, s->Btmu, 0, sizeof(double) * (size_t)n));
  s->BtBu_valid = 0;
  if (s->o.knoll) PMH_CHK(pmh_qppf_apply_P(s->pf, s->b, s->u)); // the Knoll trick smalxe.c:938-943: projected rhs as initial guess
  PMH_CHK(objective(s, &Lag_old));
  PMH_CHK(update_normBu(s, s->u, &s->normBu_old, &s->enorm));
  s->iteration       = 0;
  s->inner_iter_accu = 0;
  s->reason          = PMH_CONVERGED_ITERATING;
  PMH_CHK(pmh_mpgp_reset_statistics(s->inner));

  for (i = 0; i < maxits; i++) {
    // QPSSMALXEUpdateLambda_SMALXE smalxe.c:402-435: Btmu += rho * BtB u
    if (!s->BtBu_valid) PMH_CHK(pmh_qppf_apply_GtG(s->pf, s->u, s->BtBu)); // "BtBu reused" otherwise (smalxe.c:421-430)
    s->BtBu_valid = 1;
    PMH_CHK(pmh_vec_axpy(ctx, n, s->Btmu, rho, s->BtBu));
    if (s->reason) break;
    PMH_CHK(pmh_vec_waxpy(ctx, n, s->b_inner, -1.0, s->Btmu, s->b)); // b_inner = b - Btmu
    // QPSConvergedSetUp_Inner_SMALXE smalxe.c:537-557
    // (b does not change over the outer iterations: the same value, one host round trip per outer iteration less)
    if (i == 0) PMH_CHK(pmh_vec_norm2(ctx, n, s->b, &s->norm_rhs_outer));
    s->gtol       = s->o.rtol * s->norm_rhs_outer;
    s->ttol_outer = fmax(s->o.rtol * s->norm_rhs_outer, s->o.atol);
    PMH_CHK(pmh_vec_norm2(ctx, n, s->b_inner, &s->outer_norm_rhs_div));
    PMH_CHK(pmh_mpgp_set_tolerances(s->inner, s->o.inner.rtol, s->o.inner.atol, s->o.divtol, s->inner_max_it));
    s->normBu_prefetched = 0;
    s->normBu_final_valid = 0;
    if (s->reuse && i > 0) {
      // the inner solver still holds g = A_rho u - b_inner of the solve that just ended.  Since then b_inner lost rho_old B'B u (the multiplier update above)
      // and A_rho gained (rho_new - rho_old) B'B (smalxe_update): the gradient the next solve starts from is g + rho_new B'B u, with B'B u = BtBu already at
      // hand -- no product with F
      double  rho_now, *g = nullptr;
      PMH_CHK(pmh_op_penalized_get_penalty(s->A_inner, &rho_now));
      PMH_CHK(pmh_mpgp_set_gradient_valid(s->inner, 1, &g));
      if (g) PMH_CHK(pmh_vec_axpy(ctx, n, g, rho_now, s->BtBu));
    }
    PMH_CHK(pmh_mpgp_solve(s->inner));
    pmh_mpgp_stats st;
    PMH_CHK(pmh_mpgp_get_stats(s->inner, &st));
    s->inner_reason = st.reason;
    s->inner_rnorm  = st.rnorm;
    s->inner_iter_accu += st.iteration;
    s->inner_it_now = st.iteration;
    s->BtBu_valid   = 0;
    s->iteration = i + 1;
    // QPSSMALXEUpdateNormBu after the inner solve (smalxe.c:977): the inner solver's last convergence test evaluated ||B u|| for this very u (inner_converged
    // calls the same function on s->u and nothing has moved u since) -- the value is at hand, a second evaluation would cost two launches and a host round trip
    // for the same bits.  Not with the lagged update (its in-solve value may be the approximate one) or a caller-supplied B'B-only path
    if (!(s->normBu_final_valid && !s->o.be_implicit && !s->o.lag_enabled)) PMH_CHK(update_normBu(s, s->u, &s->normBu, &s->enorm));
    PMH_CHK(pmh_op_penalized_get_penalty(s->A_inner, &rho));
    if (s->reuse) { // f = -u'(b_inner - 1/2 A_rho u) with A_rho u = g + b_inner: -1/2 u'(b_inner - g)
      double *g = nullptr, dot;
      PMH_CHK(pmh_mpgp_set_gradient_valid(s->inner, 0, &g));
      PMH_ARG(g);
      PMH_CHK(pmh_vec_waxpy(ctx, n, s->xwork, -1.0, g, s->b_inner));
      PMH_CHK(pmh_vec_dot(ctx, n, s->u, s->xwork, &dot));
      Lag = -0.5 * dot;
    } else
    PMH_CHK(objective(s, &Lag));
    PMH_CHK(smalxe_update(s, Lag_old, Lag, rho));
    Lag_old       = Lag;
    s->normBu_old = s->normBu;
  }
  if (i == maxits && !s->reason) s->reason = PMH_DIVERGED_ITS;
  return PMH_SUCCESS;
}

// QPSReset for a solver object that is to solve again from a fresh initial guess: the state machine of QPSConverged_Inner_SMALXE back to 1 (as after QPSCreate,
// smalxe.c:1149)
extern "C" int pmh_smalxe_reset(pmh_smalxe s)
{
  PMH_ARG(s);
  s->state = 1;
  return PMH_SUCCESS;
}

// the limit of the inner solver's iterations summed over the outer iterations (the inner QPS's max_it: QPSConverged_Inner_SMALXE smalxe.c:626-631 ends the
// solve with DIVERGED_ITS / outer DIVERGED_BREAKDOWN once inner iteration i > max_it - accumulated)
extern "C" int pmh_smalxe_set_inner_max_it(pmh_smalxe s, int max_it)
{
  PMH_ARG(s && max_it >= 0);
  s->inner_max_it = max_it;
  return PMH_SUCCESS;
}

// QPGetSolutionVector of the QP this solver works on (the caller's device vector handed to pmh_smalxe_create), with its context and length
extern "C" int pmh_smalxe_get_solution(pmh_smalxe s, pmh_ctx *ctx, double **u, int *n)
{
  PMH_ARG(s);
  if (ctx) *ctx = s->ctx;
  if (u) *u = s->u;
  if (n) *n = s->n;
  return PMH_SUCCESS;
}

// the QP SMALXE hands to its inner solver -- the child QPTEnforceEqByPenalty adds (smalxe.c:822-838): Hessian A + rho B'B with the current rho, right-hand side
// b - B'mu of the last outer iteration -- and B'mu itself, which QPSSolve_SMALXE leaves as the parent's Bt_lambda (what -qp_chain_view_kkt reports them from)
extern "C" int pmh_smalxe_get_penalized(pmh_smalxe s, pmh_op *A_rho, double **b_inner, double **Bt_mu)
{
  PMH_ARG(s);
  if (A_rho) *A_rho = s->A_inner;
  if (b_inner) *b_inner = s->b_inner;
  if (Bt_mu) *Bt_mu = s->Btmu;
  return PMH_SUCCESS;
}

extern "C" int pmh_smalxe_get_inner_max_it(pmh_smalxe s, int *max_it)
{
  PMH_ARG(s && max_it);
  *max_it = s->inner_max_it;
  return PMH_SUCCESS;
}

extern "C" int pmh_smalxe_get_inner(pmh_smalxe s, pmh_mpgp *inner)
{
  PMH_ARG(s && inner);
  *inner = s->inner;
  return PMH_SUCCESS;
}

extern "C" int pmh_smalxe_get_stats(pmh_smalxe s, pmh_smalxe_stats *st)
{
  PMH_ARG(s && st);
  memset(st, 0, sizeof(*st));
  st->iteration = s->iteration, st->reason = s->reason, st->inner_iter_accu = s->inner_iter_accu, st->state = s->state;
  st->M1_hits = s->M1_hits, st->eta_hits = s->eta_hits, st->M1_updates = s->M1_updates, st->rho_updates = s->rho_updates;
  st->M1 = s->M1, st->eta = s->eta, st->normBu = s->normBu, st->enorm = s->enorm, st->rnorm = s->rnorm, st->maxeig = s->maxeig;
  pmh_op_penalized_get_penalty(s->A_inner, &st->rho);
  return pmh_mpgp_get_stats(s->inner, &st->inner);
}

// --------------------------------------------------------------------------------------------------------------------
// QPSSolve_PCPG src/qps/impls/pcpg/pcpg.c:51-134: r = b - A x; loop { w = P r; test ||w||; z = M^-1 w; y = P z;
// beta; p; Ap; alpha = (y,w)/(p,Ap); x += alpha p; r -= alpha Ap }
// --------------------------------------------------------------------------------------------------------------------
extern "C" int pmh_pcpg_solve(pmh_ctx ctx, pmh_op A, const double *b, double *x, pmh_qppf pf, pmh_op pc, double rtol, double atol, double divtol, int max_it, pmh_pcpg_stats *st)
{
  PMH_ARG(ctx && A && b && x && st); // pf == NULL: no equality constraints, P = I (the QPSKSP/KSPCG path)
  const int n = A->n;
  double   *p, *r, *w, *z, *yb, *Ap, *y;
  PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)n, (void **)&p));
  PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)n, (void **)&r));
  PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)n, (void **)&w));
  PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)n, (void **)&z));
  PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)n, (void **)&yb));
  PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)n, (void **)&Ap));
  double alpha, alpha1, beta, beta1 = 0, beta2, norm_rhs, ttol;
  int    rc = PMH_SUCCESS;
  const bool monitor = getenv("PMH_KSP_MONITOR") != nullptr;
#define PC_CHK(call) \
  if ((rc = (call))) break;
  st->iteration = 0;
  st->reason    = 0;
  do {
    PC_CHK(pmh_vec_norm2(ctx, n, b, &norm_rhs));
    ttol = fmax(rtol * norm_rhs, atol);
    PC_CHK(A->mult(x, r));
    PC_CHK(pmh_vec_aypx(ctx, n, r, -1.0, b));
    do {
      if (pf) {
        PC_CHK(pmh_qppf_apply_P(pf, r, w));
      } else {
        PC_CHK(pmh_vec_copy(ctx, n, r, w));
      }
      PC_CHK(pmh_vec_norm2(ctx, n, w, &st->rnorm));
      // -ksp_monitor's line (PMH_KSP_MONITOR=1)
      if (monitor) fprintf(stderr, "%3d KSP Residual norm %.12e (threshold %.12e)\n", st->iteration, st->rnorm, ttol);
      st->reason = PMH_CONVERGED_ITERATING; // QPSConvergedDefault
      if (st->iteration > max_it) st->reason = PMH_DIVERGED_ITS;
      else if (std::isnan(st->rnorm) || std::isinf(st->rnorm)) st->reason = PMH_DIVERGED_NANORINF;
      else if (st->rnorm <= ttol) st->reason = (st->rnorm < atol) ? PMH_CONVERGED_ATOL : PMH_CONVERGED_RTOL;
      else if (st->rnorm >= divtol * norm_rhs) st->reason = PMH_DIVERGED_DTOL;
      if (st->reason) break;
      if (!pc) {
        y = w;
      } else {
        PC_CHK(pc->mult(w, z));
        if (pf) {
          PC_CHK(pmh_qppf_apply_P(pf, z, yb));
          y = yb;
        } else {
          y = z;
        }
      }
      beta2 = beta1;
      PC_CHK(pmh_vec_dot(ctx, n, y, w, &beta1));
      if (!st->iteration) {
        beta = 0;
        PC_CHK(pmh_vec_copy(ctx, n, y, p));
      } else {
        beta = beta1 / beta2;
        PC_CHK(pmh_vec_aypx(ctx, n, p, beta, y));
      }
      PC_CHK(A->mult(p, Ap));
      PC_CHK(pmh_vec_dot(ctx, n, p, Ap, &alpha1));
      alpha = beta1 / alpha1;
      PC_CHK(pmh_vec_axpy(ctx, n, x, alpha, p));
      PC_CHK(pmh_vec_axpy(ctx, n, r, -alpha, Ap));
      st->iteration++;
    } while (st->iteration < max_it);
  } while (0);
  (void)beta;
  pmh_free(ctx, p);
  pmh_free(ctx, r);
  pmh_free(ctx, w);
  pmh_free(ctx, z);
  pmh_free(ctx, yb);
  pmh_free(ctx, Ap);
  return rc;
}

// QPSSolve_KSP src/qps/impls/ksp/qpsksp.c:127-143 with the KSP that QPSCreate_KSP configures (:244-250: KSPCG,
// KSP_NORM_UNPRECONDITIONED, nonzero initial guess, PCNONE unless the QP carries a PC) and QPSKSPConverged_KSP ->
// QPSConvergedDefault as the stopping test: this is PCPG with P = I, so the same driver serves it.
extern "C" int pmh_ksp_cg_solve(pmh_ctx ctx, pmh_op A, const double *b, double *x, pmh_op pc, double rtol, double atol, double divtol, int max_it, pmh_pcpg_stats *st)
{
  return pmh_pcpg_solve(ctx, A, b, x, nullptr, pc, rtol, atol, divtol, max_it, st);
}
