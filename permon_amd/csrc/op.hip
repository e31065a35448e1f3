// Generic operators (the PETSc Mat "mult slot" of the QP chain) and MatGetMaxEigenvalue.
#include "pmh_internal.h"
#include "reduce.h"

struct CsrOp : pmh_op_s {
  pmh_csr A;
  int     mult(const double *x, double *y) override { return pmh_csr_mult(A, x, y); }
  int     mult_transpose(const double *x, double *y) override { return pmh_csr_mult_transpose(A, x, y); }
  pmh_csr as_csr() override { return A; }
};

struct ShellOp : pmh_op_s {
  pmh_shell_mult_fn f;
  void             *user;
  int               mult(const double *x, double *y) override
  {
    int rc = f(user, x, y);
    if (rc) return pmh_set_error(PMH_ERR_ARG, "shell operator callback returned %d", rc);
    return PMH_SUCCESS;
  }
};

extern "C" int pmh_op_create_csr(pmh_csr A, pmh_op *op)
{
  PMH_ARG(A && op);
  PMH_ARG(A->nrows == A->ncols);
  CsrOp *o = new CsrOp();
  o->ctx   = A->ctx;
  o->n     = A->nrows;
  o->A     = A;
  *op      = o;
  return PMH_SUCCESS;
}

extern "C" int pmh_op_create_shell(pmh_ctx ctx, int n, pmh_shell_mult_fn f, void *user, pmh_op *op)
{
  PMH_ARG(ctx && f && op && n >= 0);
  ShellOp *o = new ShellOp();
  o->ctx     = ctx;
  o->n       = n;
  o->f       = f;
  o->user    = user;
  *op        = o;
  return PMH_SUCCESS;
}

extern "C" int pmh_op_destroy(pmh_op op)
{
  delete op;
  return PMH_SUCCESS;
}

extern "C" int pmh_op_size(pmh_op op, int *n)
{
  PMH_ARG(op && n);
  *n = op->n;
  return PMH_SUCCESS;
}

extern "C" int pmh_op_mult(pmh_op op, const double *x, double *y)
{
  PMH_ARG(op);
  return op->mult(x, y);
}

extern "C" int pmh_op_mult_transpose(pmh_op op, const double *x, double *y)
{
  PMH_ARG(op);
  return op->mult_transpose(x, y);
}

// MatGetMaxEigenvalue, src/mat/interface/permonmatutils.c:442-522: v = 1; <= maxits power iterations
// lambda = (v,Av)/(v,v); stop at |dlambda|/|lambda| < tol; v = Av/sqrt(v,v) (the reference's normalisation,
// kept).  The two dots are one fused pass; the loop is host driven (it runs once per solver set-up).
extern "C" int pmh_op_max_eigenvalue(pmh_op op, double tol, int maxits, double *lambda_out, int *its_out)
{
  PMH_ARG(op && lambda_out);
  pmh_ctx ctx = op->ctx;
  int     n   = op->n;
  if (tol == PMH_DECIDE || tol == -2.0) tol = 1e-4;
  if (maxits == -1 || maxits == -2) maxits = 50;
  double *v = nullptr, *Av = nullptr;
  PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)n, (void **)&v));
  PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)n, (void **)&Av));
  PMH_CHK(pmh_vec_set(ctx, n, v, 1.0));
  double lambda = 0.0, lambda0, vAv, vv;
  int    i, rc = PMH_SUCCESS;
  bool   seeded = false;
  for (i = 1; i <= maxits; i++) {
    lambda0 = lambda;
    if ((rc = op->mult(v, Av))) break;
    if ((rc = pmh_vec_dot(ctx, n, v, Av, &vAv))) break;
    if ((rc = pmh_vec_dot(ctx, n, v, v, &vv))) break;
    lambda = vAv / vv;
    if (lambda < 2.220446049250313e-16) {
      // permonmatutils.c:491-499: v hit the null space of A (e.g. v = 1 and a floating elasticity block): A v is replaced by a
      // random vector and only v'Av is recomputed -- lambda keeps its value in this iteration.  PETSc's RAND48 stream:
      // PetscRandomCreate seeds 0x12345678 (+ 76543 rank), srand48(seed), one drand48() per entry in index order.
      if (!seeded) srand48(0x12345678L), seeded = true;
      std::vector<double> h((size_t)n);
      for (int k = 0; k < n; k++) h[k] = drand48();
      if ((rc = pmh_memcpy_h2d(ctx, Av, h.data(), sizeof(double) * (size_t)n))) break;
      if ((rc = pmh_vec_dot(ctx, n, v, Av, &vAv))) break;
    }
    double err = fabs(lambda - lambda0), relerr = err / fabs(lambda);
    if (relerr < tol) break;
    if ((rc = pmh_vec_copy(ctx, n, Av, v))) break;
    if ((rc = pmh_vec_scale(ctx, n, v, 1.0 / sqrt(vv)))) break;
  }
  pmh_free(ctx, v);
  pmh_free(ctx, Av);
  if (rc) return rc;
  *lambda_out = lambda;
  if (its_out) *its_out = i;
  return PMH_SUCCESS;
}

// ---- post-solve KKT residuals (QPViewKKT qp.c:245-369, QPCViewKKT_Box qpcbox.c:333-427, multipliers qp.c:828-893) ----
__global__ __launch_bounds__(PMH_BLOCK) void k_kkt_box(long long n, const double *__restrict__ Ax, const double *__restrict__ b, const double *__restrict__ x, const double *__restrict__ lb, const double *__restrict__ ub, double *__restrict__ partials, int ld)
{
  __shared__ double lds[PMH_BLOCK / 64];
  double            acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (long long i = (long long)blockIdx.x * PMH_BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * PMH_BLOCK) {
    const double bi = b[i], xi = x[i], r = Ax[i] - bi; // QPComputeLagrangianGradient without the box part
    double       llb = 0.0, lub = 0.0;
    if (lb) llb = r;
    if (ub) lub = -r;
    if (lb && ub) { // qp.c:874-881: VecPointwiseMax(.,0)
      llb = (llb > 0.0) ? llb : 0.0;
      lub = (lub > 0.0) ? lub : 0.0;
    }
    const double lg = r - llb + lub;
    acc[0] += lg * lg;
    if (lb) {
      const double l = lb[i], d0 = xi - l, m0 = (d0 < 0.0) ? d0 : 0.0, m1 = (llb < 0.0) ? llb : 0.0;
      acc[1] += m0 * m0;
      acc[2] += m1 * m1;
      acc[3] += llb * ((l <= -INFINITY) ? -1.0 : (l - xi)); // qpcbox.c:371-379
    }
    if (ub) {
      const double u = ub[i], d0 = xi - u, m0 = (d0 > 0.0) ? d0 : 0.0, m1 = (lub < 0.0) ? lub : 0.0;
      acc[4] += m0 * m0;
      acc[5] += m1 * m1;
      acc[6] += lub * ((u >= INFINITY) ? 1.0 : (xi - u)); // qpcbox.c:411-419
    }
    acc[7] += bi * bi;
  }
#pragma unroll
  for (int k = 0; k < 8; k++) {
    double r = pmh_block_reduce<PMH_RED_SUM>(acc[k], lds);
    if (threadIdx.x == 0) partials[(size_t)k * ld + blockIdx.x] = r;
  }
}

extern "C" int pmh_qp_kkt_box(pmh_op A, const double *b, const double *x, const double *lb, const double *ub, double *work, double out_host[8])
{
  PMH_ARG(A && b && x && work && out_host);
  pmh_ctx   ctx = A->ctx;
  const int n   = A->n;
  PMH_CHK(A->mult(x, work));
  const int ops[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int nb     = n > 0 ? pmh_vec_grid(n) : 0;
  if (n > 0) {
    hipLaunchKernelGGL(k_kkt_box, dim3(nb), dim3(PMH_BLOCK), 0, ctx->stream, (long long)n, (const double *)work, b, x, lb, ub, ctx->d_partials, ctx->partials_cap);
    PMH_HIP(hipGetLastError());
  }
  PMH_CHK(pmh_finalize_partials(ctx, ctx->d_partials, ctx->partials_cap, nb, 8, ops, 32));
  PMH_CHK(pmh_sync(ctx));
  const double *s = ctx->h_scal + 32;
  out_host[0] = sqrt(s[0]), out_host[1] = sqrt(s[1]), out_host[2] = sqrt(s[2]), out_host[3] = fabs(s[3]);
  out_host[4] = sqrt(s[4]), out_host[5] = sqrt(s[5]), out_host[6] = fabs(s[6]), out_host[7] = sqrt(s[7]);
  return PMH_SUCCESS;
}
