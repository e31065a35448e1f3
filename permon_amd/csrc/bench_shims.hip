// Bench-only entry points (bench.py, tests): throughput modes that run a solver for an exact number of iterations.  Nothing here reaches into a solver: both are written over
// the public hooks the reference has too -- QPSSetTolerances / QPSGetTolerances (qps.c) and the inner solver's iteration limit of SMALXE (smalxe.c:626-631) -- so the solvers'
// convergence paths carry no benchmark branches.
#include <cmath>

#include "pmh_internal.h"

// exactly `iters` iterations of the MPGP loop: tolerances that no residual satisfies (rtol = atol = 0, no divergence bound) and max_it = iters - 1, i.e. QPSConvergedDefault
// (qps.c:675-714) ends the solve with DIVERGED_ITS at iteration `iters` (strict `it > max_it`); the convergence test is evaluated every iteration as in a real solve.  With an
// injected convergence test (SMALXE's inner solver) the tolerances are not consulted: use pmh_smalxe_run_fixed there.
extern "C" int pmh_mpgp_run_fixed(pmh_mpgp s, int iters)
{
  PMH_ARG(s && iters >= 0);
  double rtol, atol, divtol;
  int    max_it;
  PMH_CHK(pmh_mpgp_get_tolerances(s, &rtol, &atol, &divtol, &max_it));
  PMH_CHK(pmh_mpgp_set_tolerances(s, 0.0, 0.0, INFINITY, iters - 1));
  const int rc = pmh_mpgp_solve(s);
  PMH_CHK(pmh_mpgp_set_tolerances(s, rtol, atol, divtol, max_it));
  return rc;
}

// the REAL SMALXE loop (outer updates of the multipliers, M1, rho and the inner stopping rule included) for exactly `inner_iters` inner MPGP iterations in total: the inner
// iteration limit is set to what is left of the budget (minus one: the limit is strict), a solve that converges earlier is restarted from u = 0 with the state machine reset
// (pmh_smalxe_reset), the last one ends by that limit.  Counts accumulate over the restarts.
extern "C" int pmh_smalxe_run_fixed(pmh_smalxe s, int inner_iters, int *solves, int *outer_iters, int *ncg, int *nexp, int *nprop, int *nmv)
{
  PMH_ARG(s && inner_iters >= 0);
  int limit0 = 0;
  PMH_CHK(pmh_smalxe_get_inner_max_it(s, &limit0));
  pmh_mpgp inner = nullptr;
  PMH_CHK(pmh_smalxe_get_inner(s, &inner));
  long long left = inner_iters;
  int       ns = 0, no = 0, cg = 0, ex = 0, pr = 0, mv = 0, rc = PMH_SUCCESS;
  while (left > 0) {
    double *u = nullptr;
    int     n = 0;
    pmh_ctx ctx = nullptr;
    if ((rc = pmh_smalxe_get_solution(s, &ctx, &u, &n))) break;
    if ((rc = pmh_vec_set(ctx, n, u, 0.0))) break;
    if ((rc = pmh_smalxe_reset(s))) break;
    if ((rc = pmh_smalxe_set_inner_max_it(s, (int)(left - 1)))) break;
    if ((rc = pmh_smalxe_solve(s))) break;
    pmh_smalxe_stats st;
    if ((rc = pmh_smalxe_get_stats(s, &st))) break;
    cg += st.inner.ncg, ex += st.inner.nexp, pr += st.inner.nprop, mv += st.inner.nmv;
    ns++, no += st.iteration;
    if (st.inner_iter_accu <= 0) {
      rc = pmh_set_error(PMH_ERR_STATE, "pmh_smalxe_run_fixed: the solve made no inner iteration");
      break;
    }
    left -= st.inner_iter_accu;
  }
  (void)pmh_smalxe_set_inner_max_it(s, limit0);
  if (rc) return rc;
  if (solves) *solves = ns;
  if (outer_iters) *outer_iters = no;
  if (ncg) *ncg = cg;
  if (nexp) *nexp = ex;
  if (nprop) *nprop = pr;
  if (nmv) *nmv = mv;
  return PMH_SUCCESS;
}
