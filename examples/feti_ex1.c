/*
 * feti_ex1.c -- the reference's (T)FETI tutorial (src/tutorials/feti/ex1.c: -u'' = sin(pi u) on [0,1], u(0) = u(1) = 0, one
 * subdomain per "rank", -ne elements per subdomain) as a plain C program over the C ABI: element matrices per subdomain
 * (the MATIS input), the ASSEMBLED right-hand side, the local-to-global map, the Dirichlet ends.  Where the reference calls
 * KSPSetType(ksp, KSPFETI) / KSPFETISetDirichlet / KSPSolve, this calls pmh_qpt_matis_split_rhs, pmh_kspfeti_solve and
 * pmh_qpt_matis_assemble_solution.
 *   ./feti_ex1 -ns 4 -ne 7 -qp_chain_view_kkt -qpt_matis_to_diag_norm [-dir_in_hess] [-feti_gluing_type full] [-project 0 -qps_smalxe_rho 1e1 -dual_qp_E_orth_type gs|implicit] ...
 * prints what the reference's test harness keeps of the tutorial's output (src/tutorials/feti/output/ex1_1.out / ex1_2.out, 4 ranks, -ne 7): the `r = ...` lines of
 * every QP of the chain (written by pmh_kspfeti_solve, as QPChainPostSolve writes them inside KSPSolve) and "PERMON FETI CONVERGED_RTOL in 1 iteration".
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "permon_hip.h"

#define CHK(call) \
  do { \
    int rc_ = (call); \
    if (rc_) { \
      fprintf(stderr, "%s:%d libpermonhip error %d: %s\n", __FILE__, __LINE__, rc_, pmh_last_error()); \
      return 1; \
    } \
  } while (0)

int main(int argc, char **argv)
{
  int    ns = 4, ne_l = 3, dir_in_hess = 0, i, r;
  char   opts[2048] = "";
  size_t len = 0;
  for (i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "-ns") && i + 1 < argc) ns = atoi(argv[i + 1]); /* stands for mpirun -n */
    if (!strcmp(argv[i], "-ne") && i + 1 < argc) ne_l = atoi(argv[i + 1]);
    if (!strcmp(argv[i], "-dir_in_hess")) dir_in_hess = 1;
    len += (size_t)snprintf(opts + len, sizeof(opts) - len, "%s ", argv[i]);
    if (len >= sizeof(opts)) return 2;
  }
  if (ns < 2 || ne_l < 1) return 2;
  const int    nl = ne_l + 1, ng = ns * ne_l + 1, N = ns * nl;
  const double h = 1.0 / (ns * ne_l);

  /* MATIS input: per subdomain a tridiagonal element-sum matrix; l2g; assembled load vector */
  int    *rs = (int *)malloc(sizeof(int) * (size_t)(ns + 1)), *rowptr = (int *)malloc(sizeof(int) * (size_t)(N + 1));
  int    *col = (int *)malloc(sizeof(int) * (size_t)(3 * N)), *l2g = (int *)malloc(sizeof(int) * (size_t)N);
  double *val = (double *)malloc(sizeof(double) * (size_t)(3 * N)), *b = (double *)calloc((size_t)ng, sizeof(double));
  double *f = (double *)malloc(sizeof(double) * (size_t)N), *R = (double *)calloc((size_t)N, sizeof(double));
  double *u = (double *)malloc(sizeof(double) * (size_t)N), *x = (double *)malloc(sizeof(double) * (size_t)ng);
  int     nz = 0, dir[2], n_dir = 0;
  for (r = 0; r < ns; r++) {
    rs[r] = r * nl;
    for (i = 0; i < nl; i++) {
      const int row = r * nl + i, fixed = dir_in_hess && ((r == 0 && i == 0) || (r == ns - 1 && i == nl - 1));
      const int nbr_fixed_lo = dir_in_hess && r == 0 && i == 1, nbr_fixed_hi = dir_in_hess && r == ns - 1 && i == nl - 2;
      l2g[row]    = r * ne_l + i;
      rowptr[row] = nz;
      if (fixed) { /* MatZeroRowsColumns with the largest diagonal entry (qpfeti.c:296-303) */
        col[nz] = row, val[nz++] = 2.0;
        continue;
      }
      if (i > 0 && !nbr_fixed_lo) col[nz] = row - 1, val[nz++] = -1.0;
      col[nz] = row, val[nz++] = (i == 0 || i == nl - 1) ? 1.0 : 2.0;
      if (i < nl - 1 && !nbr_fixed_hi) col[nz] = row + 1, val[nz++] = -1.0;
    }
    for (i = 0; i < ne_l; i++) {
      const double v = sin((r * ne_l + i + .5) * h * 3.14159) * .5 * h * h;
      b[r * ne_l + i] += v, b[r * ne_l + i + 1] += v;
    }
  }
  rs[ns] = N, rowptr[N] = nz;
  CHK(pmh_qpt_matis_split_rhs(N, l2g, ng, b, f));
  for (i = 0; i < N; i++) R[i] = 1.0; /* kernel of a floating 1-D bar: constants */
  if (dir_in_hess) {
    f[0] = f[N - 1] = 0.0;
    for (i = 0; i < nl; i++) R[i] = R[N - 1 - i] = 0.0; /* the two end subdomains are fixed */
  } else {
    dir[0] = 0, dir[1] = N - 1, n_dir = 2; /* KSPFETISetDirichlet(ksp, dirichletIS, FETI_GLOBAL_UNDECOMPOSED, PETSC_TRUE) */
  }

  pmh_ctx           ctx;
  pmh_kspfeti_opts  o;
  pmh_kspfeti_stats st;
  char              left[512];
  CHK(pmh_init(0, &ctx));
  CHK(pmh_kspfeti_default_opts(&o));
  /* The tutorial hands the reference no kernel of K: QPTDualize computes one and, having done so, switches to the left generalised inverse K^- P_R without regularisation
     (qptransform.c:997-1008).  This library cannot compute kernels (it has no direct solver), so R is given above; K^- P_R is pmh_kspfeti_default_opts' K^+ (kplus_left = 1). */
  CHK(pmh_kspfeti_set_from_options(opts, &o, left, (int)sizeof(left)));
  CHK(pmh_kspfeti_solve(ctx, ns, rs, rowptr, col, val, f, l2g, n_dir, dir, 1, R, &o, u, NULL, 0, &st));
  CHK(pmh_qpt_matis_assemble_solution(N, l2g, u, ng, x));
  printf("PERMON FETI %s in %d iteration\n", st.reason == 2 ? "CONVERGED_RTOL" : (st.reason == 3 ? "CONVERGED_ATOL" : "DIVERGED"), st.iteration);

  /* not in the reference's output: the discrete equations A x = b of the assembled problem */
  double res = 0.0, nb = 0.0;
  for (i = 1; i < ng - 1; i++) {
    const double ri = 2.0 * x[i] - x[i - 1] - x[i + 1] - b[i];
    res += ri * ri, nb += b[i] * b[i];
  }
  fprintf(stderr, "dual dimension %d (Dirichlet rows %d), coarse dimension %d, ||A x - b|| / ||b|| = %.2e, |x(0)| + |x(1)| = %.1e\n", st.n_lambda, st.n_dirichlet_rows, st.coarse_dim,
          sqrt(res / nb), fabs(x[0]) + fabs(x[ng - 1]));
  pmh_finalize(ctx);
  free(rs), free(rowptr), free(col), free(l2g), free(val), free(b), free(f), free(R), free(u), free(x);
  return sqrt(res / nb) < 1e-4 ? 0 : 3;
}
