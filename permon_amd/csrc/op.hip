// Generic operators (the PETSc Mat "mult slot" of the QP chain) and MatGetMaxEigenvalue.
#include "pmh_internal.h"

struct CsrOp : pmh_op_s {
  pmh_csr A;
  int     mult(const double *x, double *y) override { return pmh_csr_mult(A, x, y); }
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
  for (i = 1; i <= maxits; i++) {
    lambda0 = lambda;
    if ((rc = op->mult(v, Av))) break;
    if ((rc = pmh_vec_dot(ctx, n, v, Av, &vAv))) break;
    if ((rc = pmh_vec_dot(ctx, n, v, v, &vv))) break;
    lambda = vAv / vv;
    if (lambda < 2.220446049250313e-16) { // permonmatutils.c:491: null-space hit; the RAND48 restart is not restated
      rc = pmh_set_error(PMH_ERR_SUP, "pmh_op_max_eigenvalue: hit the null space of A (lambda=%g) at iteration %d", lambda, i);
      break;
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
