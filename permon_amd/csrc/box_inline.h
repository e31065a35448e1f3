// The box predicates of src/qpc/impls/box/qpcbox.c restated per element, shared by every kernel that folds a gradient split, a reduced gradient or a feasible step
// length into another pass (mpgp.hip, qppf.hip, svm.hip).  Two forms of the same arithmetic: on the bound ARRAYS (a null pointer = no bound on that side) and on the
// bound VALUES (-inf / +inf = no bound: every comparison then falls as with the null pointer), for kernels that load a row's scalars once, ahead of their use.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>

// QPCGrads_Box qpcbox.c:41-55 (lower bound wins ties, `else if`)
static __device__ __forceinline__ void pmh_box_split(double xi, double gi, const double *lb, const double *ub, long long i, double astol, double &gf, double &gc)
{
  gf = gi;
  gc = 0.0;
  if (lb && fabs(xi - lb[i]) <= astol) {
    gf = 0.0;
    gc = (gi < 0.0) ? gi : 0.0;
  } else if (ub && fabs(xi - ub[i]) <= astol) {
    gf = 0.0;
    gc = (gi > 0.0) ? gi : 0.0;
  }
}
static __device__ __forceinline__ void pmh_box_split_v(double xi, double gi, double l, double u, double astol, double &gf, double &gc)
{
  gf = gi;
  gc = 0.0;
  if (fabs(xi - l) <= astol) {
    gf = 0.0;
    gc = (gi < 0.0) ? gi : 0.0;
  } else if (fabs(xi - u) <= astol) {
    gf = 0.0;
    gc = (gi > 0.0) ? gi : 0.0;
  }
}

// QPCGradReduced_Box qpcbox.c:86-92
static __device__ __forceinline__ double pmh_box_reduced(double xi, double gf, const double *lb, const double *ub, long long i, double alpha)
{
  double r = gf;
  if (lb && gf > 0.0) {
    double t = (xi - lb[i]) / alpha;
    r        = (gf < t) ? gf : t;
  } else if (ub && gf < 0.0) {
    double t = (xi - ub[i]) / alpha;
    r        = (gf < t) ? t : gf;
  }
  return r;
}
static __device__ __forceinline__ double pmh_box_reduced_v(double xi, double gf, double l, double u, bool has_l, bool has_u, double alpha)
{
  double r = gf;
  if (has_l && gf > 0.0) {
    double t = (xi - l) / alpha;
    r        = (gf < t) ? gf : t;
  } else if (has_u && gf < 0.0) {
    double t = (xi - u) / alpha;
    r        = (gf < t) ? t : gf;
  }
  return r;
}

// QPCFeas_Box (qpcbox.c:290-305), one entry: the step length along -p that keeps x on the box, folded into a running minimum
static __device__ __forceinline__ double pmh_box_feas_v(double m, double xi, double pi, double l, double u)
{
  if (pi > 0. && l > -INFINITY) m = fmin(m, (xi - l) / pi);
  if (pi < 0. && u < INFINITY) m = fmin(m, (xi - u) / pi);
  return m;
}
