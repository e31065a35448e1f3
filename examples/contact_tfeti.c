/* A TFETI contact problem solved from plain C over the C ABI of libpermonhip (include/permon_hip.h): pmh_feti_contact_solve is the
 * counterpart of the reference's QPTFromOptions -> QPSSolve -> QPChainPostSolve for a decomposed QP with inequality rows
 * (src/qp/interface/qptransform.c:2152-2237).  The problem comes from a file written by permon_amd.problems.write_contact_problem
 * (the generator of BASELINE configs[2]: cubes of Q1 elasticity elements, TFETI gluing, rigid obstacle):
 *   int32 header[8] = {magic 0x504D4831, nsub, N, nnz, n_lambda, n_eq, n_leaves, kdim}, int32 ndof, int32 dims[3 nsub],
 *   int32 block_rowstart[nsub+1], rowptr[N+1], col[nnz], leaves_row[n_leaves], leaves_root[n_leaves]; then float64 val[nnz], f[N],
 *   leaves_val[n_leaves], c[n_lambda], R[kdim N].
 * usage: contact_tfeti problem.bin [explicit=1] [mg_precision 0|1|2] [explicit_storage 0|1|2|3|4]      prints the -qps_view_convergence block and checks of the solution */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "permon_hip.h"

#define CHECK(call) \
  do { \
    int rc_ = (call); \
    if (rc_) { \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, pmh_last_error()); \
      return 1; \
    } \
  } while (0)

static void *rd(FILE *fh, size_t bytes)
{
  void *p = malloc(bytes ? bytes : 1);
  if (!p || fread(p, 1, bytes, fh) != bytes) {
    fprintf(stderr, "short read\n");
    exit(2);
  }
  return p;
}

int main(int argc, char **argv)
{
  if (argc < 2) {
    fprintf(stderr, "usage: %s problem.bin [explicit=1] [mg_precision=2]\n", argv[0]);
    return 2;
  }
  FILE *fh = fopen(argv[1], "rb");
  if (!fh) {
    perror(argv[1]);
    return 2;
  }
  int *h = (int *)rd(fh, 8 * sizeof(int));
  if (h[0] != 0x504D4831) {
    fprintf(stderr, "bad magic\n");
    return 2;
  }
  const int nsub = h[1], N = h[2], nnz = h[3], nl = h[4], neq = h[5], nleaf = h[6], kdim = h[7];
  int      *ndof = (int *)rd(fh, sizeof(int)), *dims = (int *)rd(fh, sizeof(int) * 3 * nsub), *rs = (int *)rd(fh, sizeof(int) * (nsub + 1));
  int      *rp = (int *)rd(fh, sizeof(int) * (N + 1)), *ci = (int *)rd(fh, sizeof(int) * nnz), *lrow = (int *)rd(fh, sizeof(int) * nleaf), *lroot = (int *)rd(fh, sizeof(int) * nleaf);
  double   *va = (double *)rd(fh, sizeof(double) * nnz), *f = (double *)rd(fh, sizeof(double) * N), *lval = (double *)rd(fh, sizeof(double) * nleaf);
  double   *c = (double *)rd(fh, sizeof(double) * nl), *R = (double *)rd(fh, sizeof(double) * (size_t)kdim * N);
  fclose(fh);

  pmh_ctx ctx;
  CHECK(pmh_init(0, &ctx));
  pmh_feti_contact_opts  o;
  pmh_feti_contact_stats st;
  CHECK(pmh_feti_contact_default_opts(&o));
  if (argc > 2) o.explicit_dual = atoi(argv[2]);
  if (argc > 3) o.mg_precision = atoi(argv[3]);
  if (argc > 4) o.explicit_storage = atoi(argv[4]); /* 1 PMH_FX_SYM (default), 0 PMH_FX_FULL, 2 PMH_FX_CLASS, 3 PMH_FX_CLASS_SYM, 4 PMH_FX_CLASS_ORBIT */
  double *u = (double *)malloc(sizeof(double) * N), *lam = (double *)malloc(sizeof(double) * nl);
  CHECK(pmh_feti_contact_solve(ctx, nsub, rs, rp, ci, va, f, nl, neq, nleaf, lrow, lroot, lval, c, kdim, R, dims, *ndof, &o, u, lam, &st));

  /* the reference's -qps_view_convergence block for SMALXE (QPSViewConvergence_SMALXE smalxe.c:1001-1018) */
  const pmh_smalxe_stats *s = &st.smalxe;
  printf("last QPSSolve %s, KSPReason=%d, required %d iterations\n", s->reason > 0 ? "CONVERGED" : "DIVERGED", s->reason, s->iteration);
  printf("Total number of inner iterations %d\n", s->inner_iter_accu);
  printf("#hits    of M1, eta: %3d, %3d\n", s->M1_hits, s->eta_hits);
  printf("#updates of M1, rho: %3d, %3d\n", s->M1_updates, s->rho_updates);
  printf("number of Hessian multiplications %d\n", s->inner.nmv);
  printf("number of CG steps %d\n", s->inner.ncg);
  printf("number of expansion steps %d\n", s->inner.nexp);
  printf("number of proportioning steps %d\n", s->inner.nprop);
  /* checks of the solution on the host: B u <= c on the inequality rows, B u = c on the equality rows, lambda_I >= 0 */
  double *Bu = (double *)calloc(nl, sizeof(double)), umax = 0.0, eqv = 0.0, pen = 0.0, lmin = 0.0;
  for (int q = 0; q < nleaf; q++) Bu[lroot[q]] += lval[q] * u[lrow[q]];
  for (int i = 0; i < N; i++) umax = fmax(umax, fabs(u[i]));
  for (int q = 0; q < nl; q++) {
    if (q < neq) eqv = fmax(eqv, fabs(Bu[q] - c[q]));
    else pen = fmax(pen, Bu[q] - c[q]), lmin = fmin(lmin, lam[q]);
  }
  printf("n_lambda %d  coarse_dim %d  active contact rows %d  explicit set-up solves %d (symmetries used: %d)\n", st.n_lambda, st.coarse_dim, st.n_active, st.explicit_solves, st.explicit_symmetries);
  printf("||G lambda - e|| = %.2e   max|B_E u - c_E| / max|u| = %.2e   max(B_I u - c_I) / max|u| = %.2e   min lambda_I = %.2e\n", st.norm_Glambda_minus_e, eqv / umax, pen / umax, lmin);
  fprintf(stderr, "set-up %.2f s (explicit operators %.2f s), solve %.3f s\n", st.setup_seconds, st.explicit_seconds, st.solve_seconds);
  CHECK(pmh_finalize(ctx));
  return 0;
}
