/*
 * ex1.c -- the reference's first tutorial (src/tutorials/ex1.c: 1-D string above an obstacle, box-constrained QP
 *   min 1/2 x'Ax - x'b  s.t.  c <= x,  A = tridiag(-1,2,-1) with Dirichlet ends, b = -30 h^2, c = sin(4 pi i h - pi/6)/2 - 2)
 * as a plain C program over the C ABI of libpermonhip (include/permon_hip.h): what a PETSc-free caller writes instead of
 * QPCreate / QPSetOperator / QPSetBox / QPSCreate / QPSSetFromOptions / QPSSolve.  It takes the reference's command line
 *   ./ex1 -n 100 -qps_view_convergence -qp_chain_view_kkt [-qps_mpgp_expansion_type gf -qps_mpgp_expansion_length_type opt ...]
 * and prints what the reference prints for those options, so its output can be diffed against src/tutorials/output/ex1_*.out
 * (tests/test_gpu_examples.py does exactly that).
 * Build:  gcc -std=c99 -O2 -Iinclude examples/ex1.c -o examples/ex1 -Lpermon_amd -lpermonhip -Wl,-rpath,$PWD/permon_amd -lm
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

static const char *reason_name(int r)
{
  switch (r) {
  case 2: return "CONVERGED_RTOL";
  case 3: return "CONVERGED_ATOL";
  case 4: return "CONVERGED_ITS";
  case 7: return "CONVERGED_HAPPY_BREAKDOWN";
  case -3: return "DIVERGED_ITS";
  case -4: return "DIVERGED_DTOL";
  case -5: return "DIVERGED_BREAKDOWN";
  case -9: return "DIVERGED_NANORINF";
  }
  return "UNKNOWN";
}

int main(int argc, char **argv)
{
  int    n = 10, view_kkt = 0, i;
  char   opts[4096] = "", left[1024];
  size_t len = 0;
  for (i = 1; i < argc; i++) { /* the example's own keys; everything goes to the options front end as well */
    if (!strcmp(argv[i], "-n") && i + 1 < argc) n = atoi(argv[i + 1]);
    if (!strcmp(argv[i], "-qp_chain_view_kkt")) view_kkt = 1;
    len += (size_t)snprintf(opts + len, sizeof(opts) - len, "%s ", argv[i]);
    if (len >= sizeof(opts)) return 2;
  }
  if (n < 3) return 2;

  /* problem data on the host */
  const double pi = 3.14159265358979323846, h = 1.0 / (n - 1);
  int         *rowptr = (int *)malloc(sizeof(int) * (size_t)(n + 1)), *col = (int *)malloc(sizeof(int) * (size_t)(3 * n));
  double      *val = (double *)malloc(sizeof(double) * (size_t)(3 * n)), *b = (double *)calloc((size_t)n, sizeof(double));
  double      *c = (double *)calloc((size_t)n, sizeof(double));
  int          nz = 0;
  for (i = 0; i < n; i++) {
    rowptr[i] = nz;
    if (i == 0 || i == n - 1) { /* Dirichlet ends: identity rows, b = 0, c = 0 */
      col[nz] = i, val[nz++] = 1.0;
      continue;
    }
    if (i != 1) col[nz] = i - 1, val[nz++] = -1.0; /* the coupling to the Dirichlet ends is dropped */
    col[nz] = i, val[nz++] = 2.0;
    if (i != n - 2) col[nz] = i + 1, val[nz++] = -1.0;
    b[i] = -15 * h * h * 2;
    c[i] = sin(4 * pi * i * h - pi / 6.) / 2 - 2;
  }
  rowptr[n] = nz;

  /* device objects */
  pmh_ctx ctx;
  pmh_csr A;
  pmh_op  op;
  double *d_b, *d_x, *d_c, *d_work;
  CHK(pmh_init(0, &ctx));
  CHK(pmh_csr_create(ctx, n, n, rowptr, col, val, &A));
  CHK(pmh_op_create_csr(A, &op));
  CHK(pmh_malloc(ctx, sizeof(double) * (size_t)n, (void **)&d_b));
  CHK(pmh_malloc(ctx, sizeof(double) * (size_t)n, (void **)&d_x));
  CHK(pmh_malloc(ctx, sizeof(double) * (size_t)n, (void **)&d_c));
  CHK(pmh_malloc(ctx, sizeof(double) * (size_t)n, (void **)&d_work));
  CHK(pmh_memcpy_h2d(ctx, d_b, b, sizeof(double) * (size_t)n));
  CHK(pmh_memcpy_h2d(ctx, d_c, c, sizeof(double) * (size_t)n));
  CHK(pmh_memset(ctx, d_x, 0, sizeof(double) * (size_t)n)); /* QPSetInitialVector(qp, x): x = 0 */

  /* QPSSetFromOptions: box constraints only => QPSSetDefaultType picks MPGP (qps.c:445) */
  pmh_qps_opts  q;
  pmh_mpgp_opts m;
  CHK(pmh_qps_default_opts(&q));
  CHK(pmh_mpgp_default_opts(&m));
  CHK(pmh_qps_set_from_options(opts, "", &q, &m, NULL, left, (int)sizeof(left)));
  if (q.type[0] && strcmp(q.type, "mpgp")) {
    fprintf(stderr, "this example solves a box-constrained QP: -qps_type %s is not compatible with it\n", q.type);
    return 1;
  }
  m.rtol = q.rtol, m.atol = q.atol, m.divtol = q.divtol, m.max_it = q.max_it;

  /* QPSSolve */
  pmh_mpgp       s;
  pmh_mpgp_stats st;
  CHK(pmh_mpgp_create(ctx, op, d_b, d_x, d_c, NULL, &m, &s));
  CHK(pmh_mpgp_solve(s));
  CHK(pmh_mpgp_get_stats(s, &st));
  if (st.reason <= 0) printf("QPS did not converge!\n");

  if (q.view_convergence) { /* QPSViewConvergence (qps.c:968) + QPSViewConvergence_MPGP (mpgp.c:751-770) */
    printf("  last QPSSolve %s due to %s, KSPReason=%d, required %d iterations\n", st.reason > 0 ? "CONVERGED" : "DIVERGED", reason_name(st.reason), st.reason, st.iteration);
    printf("    number of Hessian multiplications %d\n", st.nmv);
    printf("    number of CG steps %d\n", st.ncg);
    printf("    number of expansion steps %d\n", st.nexp);
    printf("    number of proportioning steps %d\n", st.nprop);
  }
  if (view_kkt) { /* QPViewKKT (qp.c:245-369) + QPCViewKKT_Box (qpcbox.c:333-427), lower bound only */
    double r[8];
    CHK(pmh_qp_kkt_box(op, d_b, d_x, d_c, NULL, d_work, r));
    printf("r = ||A*x - b - lambda_lb|| = %.2e    rO/||b|| = %.2e\n", r[0], r[0] / r[7]);
    printf("r = ||min(x-lb,0)||      = %.2e    r/||b|| = %.2e\n", r[1], r[1] / r[7]);
    printf("r = ||min(lambda_lb,0)|| = %.2e    r/||b|| = %.2e\n", r[2], r[2] / r[7]);
    printf("r = |lambda_lb'*(lb-x)|  = %.2e    r/||b|| = %.2e\n", r[3], r[3] / r[7]);
  }

  pmh_mpgp_destroy(s);
  pmh_op_destroy(op);
  pmh_csr_destroy(A);
  pmh_free(ctx, d_b), pmh_free(ctx, d_x), pmh_free(ctx, d_c), pmh_free(ctx, d_work);
  pmh_finalize(ctx);
  free(rowptr), free(col), free(val), free(b), free(c);
  return 0;
}
