// Multigrid V-cycle preconditioner for the block CG inside MATINV.
// The reference's iterative MATINV path is a PETSc KSP whose PC is whatever -mat_inv_pc_type names
// (src/mat/impls/inv/matinv.c: MatInvGetKSP / MatInvSetUp); for the 3-D elasticity blocks of BASELINE configs[2] the
// Jacobi default needs ~1000 CG iterations per K^+ application.  This is the PCMG equivalent on the device:
// Galerkin hierarchy A_{l+1} = P_l' A_l P_l handed over as CSR (built by the caller), Chebyshev/Jacobi smoothing with
// PETSc's eigenvalue window [lo, hi] x lambda_max(D^-1 A) (KSPCHEBYSHEV as PCMG/PCGAMG configure it), and block-wise
// dense pseudo-inverses on the coarsest level (floating subdomains stay singular down the hierarchy: the prolongation
// reproduces the rigid-body modes).  Pre- and post-smoother are the same polynomial, so the cycle is symmetric positive
// (semi-)definite and valid inside CG.  Every kernel is HBM bound; the SpMVs are the tuned pmh_csr kernels.
#include "pmh_internal.h"

struct mg_level {
  pmh_csr A, P;                  // P: n_l x n_{l+1} (NULL on the coarsest level)
  int     n;
  double *dinv, *x, *b, *r, *d, *t; // x, b are borrowed on level 0
  double  theta, delta;
  std::vector<double> c1, c2;    // Chebyshev recurrence coefficients of steps 1..degree-1
};

struct pmh_mg_s {
  pmh_ctx               ctx;
  int                   nlevels, degree;
  std::vector<mg_level> L;
  int                   nb_coarse;
  int                  *d_crs;   // coarse block row starts [nb_coarse+1]
  long long            *d_cofs;  // offsets of the dense blocks [nb_coarse]
  double               *d_cpinv; // concatenated dense pseudo-inverses, row-major
  const int            *halt;
  long long             fine_spmv; // fine-level SpMVs issued (statistics)
};

__global__ __launch_bounds__(PMH_BLOCK) void k_mg_dinv(int n, const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val, double *__restrict__ dinv)
{
  for (int i = blockIdx.x * PMH_BLOCK + threadIdx.x; i < n; i += gridDim.x * PMH_BLOCK) {
    double d = 0.0;
    for (int k = rowptr[i]; k < rowptr[i + 1]; k++)
      if (col[k] == i) d = val[k];
    dinv[i] = (d != 0.0) ? 1.0 / d : 1.0;
  }
}

// first Chebyshev step from a zero guess: r = D^-1 b, d = r/theta, x = d
__global__ __launch_bounds__(PMH_BLOCK) void k_cheb_first_zero(int n, const int *__restrict__ halt, const double *__restrict__ dinv, const double *__restrict__ b, double itheta, double *__restrict__ r, double *__restrict__ d, double *__restrict__ x)
{
  if (halt && *halt) return;
  for (int i = blockIdx.x * PMH_BLOCK + threadIdx.x; i < n; i += gridDim.x * PMH_BLOCK) {
    const double ri = dinv[i] * b[i], di = ri * itheta;
    r[i] = ri;
    d[i] = di;
    x[i] = di;
  }
}

// first step from the current x (t = A x): r = D^-1 (b - t), d = r/theta, x += d
__global__ __launch_bounds__(PMH_BLOCK) void k_cheb_first(int n, const int *__restrict__ halt, const double *__restrict__ dinv, const double *__restrict__ b, const double *__restrict__ t, double itheta, double *__restrict__ r, double *__restrict__ d, double *__restrict__ x)
{
  if (halt && *halt) return;
  for (int i = blockIdx.x * PMH_BLOCK + threadIdx.x; i < n; i += gridDim.x * PMH_BLOCK) {
    const double ri = dinv[i] * (b[i] - t[i]), di = ri * itheta;
    r[i] = ri;
    d[i] = di;
    x[i] += di;
  }
}

// later steps (t = A d): r -= D^-1 t, d = c1 d + c2 r, x += d
__global__ __launch_bounds__(PMH_BLOCK) void k_cheb_step(int n, const int *__restrict__ halt, const double *__restrict__ dinv, const double *__restrict__ t, double c1, double c2, double *__restrict__ r, double *__restrict__ d, double *__restrict__ x)
{
  if (halt && *halt) return;
  for (int i = blockIdx.x * PMH_BLOCK + threadIdx.x; i < n; i += gridDim.x * PMH_BLOCK) {
    const double ri = r[i] - dinv[i] * t[i], di = c1 * d[i] + c2 * ri;
    r[i] = ri;
    d[i] = di;
    x[i] += di;
  }
}

// x += t
__global__ __launch_bounds__(PMH_BLOCK) void k_mg_add(int n, const int *__restrict__ halt, const double *__restrict__ t, double *__restrict__ x)
{
  if (halt && *halt) return;
  for (int i = blockIdx.x * PMH_BLOCK + threadIdx.x; i < n; i += gridDim.x * PMH_BLOCK) x[i] += t[i];
}

// coarsest level: x_b = pinv_b b_b, one wavefront per row, lanes stride the row of the dense block (fixed order)
__global__ __launch_bounds__(PMH_BLOCK) void k_mg_coarse(int nb, int n, const int *__restrict__ halt, const int *__restrict__ rs, const long long *__restrict__ ofs, const double *__restrict__ pinv, const double *__restrict__ b, double *__restrict__ x)
{
  if (halt && *halt) return;
  const int lane = threadIdx.x & 63;
  const int row  = blockIdx.x * (PMH_BLOCK / 64) + (threadIdx.x >> 6);
  if (row >= n) return;
  int lo = 0, hi = nb; // block of this row: rs[lo] <= row < rs[lo+1]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (rs[mid] <= row) lo = mid;
    else hi = mid;
  }
  const int     r0 = rs[lo], m = rs[lo + 1] - r0;
  const double *a  = pinv + ofs[lo] + (size_t)(row - r0) * m;
  double        s  = 0.0;
  for (int j = lane; j < m; j += 64) s += a[j] * b[r0 + j];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if (lane == 0) x[row] = s;
}

static inline dim3 mg_grid(int n)
{
  long long g = ((long long)n + PMH_BLOCK - 1) / PMH_BLOCK;
  return dim3((unsigned)(g < 1 ? 1 : (g > PMH_MAX_VEC_BLOCKS ? PMH_MAX_VEC_BLOCKS : g)));
}

static int mg_spmv(pmh_mg mg, int l, pmh_csr A, const double *x, double *y)
{
  pmh_spmv_epi epi;
  memset(&epi, 0, sizeof(epi));
  epi.kind = PMH_EPI_NONE;
  epi.halt = mg->halt;
  if (l == 0) mg->fine_spmv++;
  return pmh_csr_spmv_launch(A, x, y, epi);
}

// degree-k Chebyshev/Jacobi smoothing of A x = b on level l; zero: x is taken as 0 on entry
static int mg_smooth(pmh_mg mg, int l, const double *b, double *x, bool zero)
{
  mg_level   &Lv = mg->L[l];
  hipStream_t st = mg->ctx->stream;
  const dim3  g  = mg_grid(Lv.n);
  if (zero) {
    hipLaunchKernelGGL(k_cheb_first_zero, g, dim3(PMH_BLOCK), 0, st, Lv.n, mg->halt, (const double *)Lv.dinv, b, 1.0 / Lv.theta, Lv.r, Lv.d, x);
  } else {
    PMH_CHK(mg_spmv(mg, l, Lv.A, x, Lv.t));
    hipLaunchKernelGGL(k_cheb_first, g, dim3(PMH_BLOCK), 0, st, Lv.n, mg->halt, (const double *)Lv.dinv, b, (const double *)Lv.t, 1.0 / Lv.theta, Lv.r, Lv.d, x);
  }
  for (int j = 1; j < mg->degree; j++) {
    PMH_CHK(mg_spmv(mg, l, Lv.A, Lv.d, Lv.t));
    hipLaunchKernelGGL(k_cheb_step, g, dim3(PMH_BLOCK), 0, st, Lv.n, mg->halt, (const double *)Lv.dinv, (const double *)Lv.t, Lv.c1[j], Lv.c2[j], Lv.r, Lv.d, x);
  }
  PMH_HIP(hipGetLastError());
  return PMH_SUCCESS;
}

static int mg_cycle(pmh_mg mg, int l, const double *b, double *x)
{
  mg_level   &Lv = mg->L[l];
  hipStream_t st = mg->ctx->stream;
  if (l == mg->nlevels - 1) {
    hipLaunchKernelGGL(k_mg_coarse, dim3((Lv.n + 3) / 4), dim3(PMH_BLOCK), 0, st, mg->nb_coarse, Lv.n, mg->halt, (const int *)mg->d_crs, (const long long *)mg->d_cofs, (const double *)mg->d_cpinv, b, x);
    PMH_HIP(hipGetLastError());
    return PMH_SUCCESS;
  }
  mg_level &Lc = mg->L[l + 1];
  PMH_CHK(mg_smooth(mg, l, b, x, true));
  // residual and restriction: b_{l+1} = P'(b - A x)
  pmh_spmv_epi epi;
  memset(&epi, 0, sizeof(epi));
  epi.kind = PMH_EPI_SUB; // t = A x - b
  epi.y1   = b;
  epi.halt = mg->halt;
  if (l == 0) mg->fine_spmv++;
  PMH_CHK(pmh_csr_spmv_launch(Lv.A, x, Lv.t, epi));
  PMH_CHK(pmh_csr_mult_transpose(Lv.P, Lv.t, Lc.b)); // = -P'(b - A x): the sign is undone when the correction is added
  PMH_CHK(mg_cycle(mg, l + 1, Lc.b, Lc.x));
  // x -= P x_{l+1}  (x_{l+1} solves A_{l+1} x_{l+1} = -restricted residual)
  PMH_CHK(pmh_csr_mult(Lv.P, Lc.x, Lv.t));
  PMH_CHK(pmh_vec_axpy(mg->ctx, Lv.n, x, -1.0, Lv.t));
  return mg_smooth(mg, l, b, x, false);
}

int pmh_mg_apply_halt(pmh_mg mg, const double *b, double *x, const int *halt)
{
  mg->halt = halt;
  int rc   = mg_cycle(mg, 0, b, x);
  mg->halt = nullptr;
  return rc;
}

extern "C" int pmh_mg_apply(pmh_mg mg, const double *b, double *x)
{
  PMH_ARG(mg && b && x && b != x);
  return pmh_mg_apply_halt(mg, b, x, nullptr);
}

extern "C" int pmh_mg_create(pmh_ctx ctx, int nlevels, const pmh_csr *A, const pmh_csr *P, int degree, const double *lambda_max, double lo_frac, double hi_frac, int nb_coarse, const int *coarse_rowstart,
                             const double *coarse_pinv_host, pmh_mg *out)
{
  PMH_ARG(ctx && out && A && nlevels >= 1 && degree >= 1 && nb_coarse >= 1 && coarse_rowstart && coarse_pinv_host);
  PMH_ARG(nlevels == 1 || (P && lambda_max));
  PMH_ARG(hi_frac > lo_frac && lo_frac > 0.0);
  for (int l = 0; l < nlevels; l++) {
    PMH_ARG(A[l] && A[l]->nrows == A[l]->ncols);
    if (l + 1 < nlevels) PMH_ARG(P[l] && P[l]->nrows == A[l]->nrows && P[l]->ncols == A[l + 1]->nrows && lambda_max[l] > 0.0);
  }
  PMH_ARG(coarse_rowstart[0] == 0 && coarse_rowstart[nb_coarse] == A[nlevels - 1]->nrows);
  pmh_mg mg   = new pmh_mg_s();
  mg->ctx     = ctx;
  mg->nlevels = nlevels;
  mg->degree  = degree;
  mg->halt    = nullptr;
  mg->fine_spmv = 0;
  mg->L.resize(nlevels);
  for (int l = 0; l < nlevels; l++) {
    mg_level &Lv = mg->L[l];
    Lv.A = A[l], Lv.P = (l + 1 < nlevels) ? P[l] : nullptr, Lv.n = A[l]->nrows;
    Lv.dinv = Lv.x = Lv.b = Lv.r = Lv.d = Lv.t = nullptr;
    const size_t nbytes = sizeof(double) * (size_t)(Lv.n ? Lv.n : 1);
    if (l > 0) {
      PMH_CHK(pmh_malloc(ctx, nbytes, (void **)&Lv.x));
      PMH_CHK(pmh_malloc(ctx, nbytes, (void **)&Lv.b));
    }
    if (l + 1 < nlevels) {
      PMH_CHK(pmh_malloc(ctx, nbytes, (void **)&Lv.dinv));
      PMH_CHK(pmh_malloc(ctx, nbytes, (void **)&Lv.r));
      PMH_CHK(pmh_malloc(ctx, nbytes, (void **)&Lv.d));
      PMH_CHK(pmh_malloc(ctx, nbytes, (void **)&Lv.t));
      if (Lv.n > 0) {
        hipLaunchKernelGGL(k_mg_dinv, mg_grid(Lv.n), dim3(PMH_BLOCK), 0, ctx->stream, Lv.n, (const int *)A[l]->d_rowptr, (const int *)A[l]->d_col, (const double *)A[l]->d_val, Lv.dinv);
        PMH_HIP(hipGetLastError());
      }
      // KSPChebyshev recurrence on the window [lo, hi] x lambda_max
      const double a = lo_frac * lambda_max[l], b = hi_frac * lambda_max[l];
      Lv.theta = 0.5 * (a + b), Lv.delta = 0.5 * (b - a);
      const double sigma = Lv.theta / Lv.delta;
      double       rho   = 1.0 / sigma;
      Lv.c1.assign(degree, 0.0), Lv.c2.assign(degree, 0.0);
      for (int j = 1; j < degree; j++) {
        const double rho_new = 1.0 / (2.0 * sigma - rho);
        Lv.c1[j] = rho_new * rho, Lv.c2[j] = 2.0 * rho_new / Lv.delta;
        rho = rho_new;
      }
    }
  }
  mg->nb_coarse = nb_coarse;
  std::vector<long long> ofs(nb_coarse);
  long long              tot = 0;
  for (int b = 0; b < nb_coarse; b++) {
    const long long m = coarse_rowstart[b + 1] - coarse_rowstart[b];
    PMH_ARG(m >= 0);
    ofs[b] = tot, tot += m * m;
  }
  PMH_CHK(pmh_malloc(ctx, sizeof(int) * (size_t)(nb_coarse + 1), (void **)&mg->d_crs));
  PMH_CHK(pmh_malloc(ctx, sizeof(long long) * (size_t)nb_coarse, (void **)&mg->d_cofs));
  PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)(tot ? tot : 1), (void **)&mg->d_cpinv));
  PMH_CHK(pmh_memcpy_h2d(ctx, mg->d_crs, coarse_rowstart, sizeof(int) * (size_t)(nb_coarse + 1)));
  PMH_CHK(pmh_memcpy_h2d(ctx, mg->d_cofs, ofs.data(), sizeof(long long) * (size_t)nb_coarse));
  PMH_CHK(pmh_memcpy_h2d(ctx, mg->d_cpinv, coarse_pinv_host, sizeof(double) * (size_t)tot));
  *out = mg;
  return PMH_SUCCESS;
}

extern "C" int pmh_mg_destroy(pmh_mg mg)
{
  if (!mg) return PMH_SUCCESS;
  pmh_ctx ctx = mg->ctx;
  for (auto &Lv : mg->L) {
    pmh_free(ctx, Lv.dinv);
    pmh_free(ctx, Lv.r);
    pmh_free(ctx, Lv.d);
    pmh_free(ctx, Lv.t);
    pmh_free(ctx, Lv.x);
    pmh_free(ctx, Lv.b);
  }
  pmh_free(ctx, mg->d_crs);
  pmh_free(ctx, mg->d_cofs);
  pmh_free(ctx, mg->d_cpinv);
  delete mg;
  return PMH_SUCCESS;
}

extern "C" int pmh_mg_stats(pmh_mg mg, long long *fine_spmv)
{
  PMH_ARG(mg);
  if (fine_spmv) *fine_spmv = mg->fine_spmv;
  return PMH_SUCCESS;
}
