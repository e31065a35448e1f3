// Dense-row Hessian of the PermonSVM-style hinge-loss dual (BASELINE.json configs[4], SURVEY 8d "C5"):
//   H = diag(y) X X' diag(y),  X in R^{N x d} row-major (one sample per row), applied matrix-free as two GEMV
//   passes over X:  w = X'(y o a)  (d-vector),  (H a)_i = y_i (x_i . w).
// PermonSVM is a separate repository (README.md:12) with no code or tests in the reference tree, so this operator
// is "parity unpinned" beyond the MPGP solver that calls it; the oracle side of its test is a numpy restatement.
// HBM-bound: algorithmic bytes per apply 2*8*N*d + 40*N (X read twice; a, y read, Ha written, y read again).
// Samples shard over GPUs by rows; the only exchange is the all-reduce of w (d doubles) between the passes.
#include "pmh_internal.h"
#include "reduce.h"

#define SVM_KMAX 4 // d <= 64 * SVM_KMAX

struct SvmDualOp : pmh_op_s {
  int           d;
  const double *X, *y;
  double       *w, *part; // w: d; part: [nblocks][d]
  int           nblocks;
  int           mult(const double *a, double *Ha) override;
  ~SvmDualOp() override
  {
    pmh_free(ctx, w);
    pmh_free(ctx, part);
  }
};

// pass 1: per-workgroup partial of w = sum_i (y_i a_i) x_i ; one wavefront per row, lane j owns columns j, j+64, ...
__global__ __launch_bounds__(PMH_BLOCK) void k_svm_xt(int n, int d, const double *__restrict__ X, const double *__restrict__ y, const double *__restrict__ a, double *__restrict__ part)
{
  __shared__ double lds[PMH_BLOCK / 64][64 * SVM_KMAX];
  const int         lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long   gw = (long long)blockIdx.x * (PMH_BLOCK / 64) + wave, nw = (long long)gridDim.x * (PMH_BLOCK / 64);
  double            acc[SVM_KMAX];
#pragma unroll
  for (int k = 0; k < SVM_KMAX; k++) acc[k] = 0.0;
  for (long long i = gw; i < n; i += nw) {
    const double  s  = y[i] * a[i];
    const double *xr = X + (size_t)i * d;
#pragma unroll
    for (int k = 0; k < SVM_KMAX; k++) {
      const int c = lane + 64 * k;
      if (c < d) acc[k] += s * __builtin_nontemporal_load(&xr[c]);
    }
  }
#pragma unroll
  for (int k = 0; k < SVM_KMAX; k++) lds[wave][lane + 64 * k] = acc[k];
  __syncthreads();
  for (int c = threadIdx.x; c < d; c += PMH_BLOCK) {
    double v = lds[0][c];
#pragma unroll
    for (int wv = 1; wv < PMH_BLOCK / 64; wv++) v += lds[wv][c];
    part[(size_t)blockIdx.x * d + c] = v;
  }
}

// w[c] = sum over workgroups of part[b][c], one wavefront per column, fixed order
__global__ __launch_bounds__(PMH_BLOCK) void k_svm_colsum(int nblocks, int d, const double *__restrict__ part, double *__restrict__ w)
{
  const int lane = threadIdx.x & 63, c = blockIdx.x * (PMH_BLOCK / 64) + (threadIdx.x >> 6);
  if (c >= d) return;
  double v = 0.0;
  for (int b = lane; b < nblocks; b += 64) v += part[(size_t)b * d + c];
  v = pmh_wave_sum(v);
  if (lane == 0) w[c] = v;
}

// pass 2: (H a)_i = y_i (x_i . w)
__global__ __launch_bounds__(PMH_BLOCK) void k_svm_x(int n, int d, const double *__restrict__ X, const double *__restrict__ y, const double *__restrict__ w, double *__restrict__ Ha)
{
  const int       lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long gw = (long long)blockIdx.x * (PMH_BLOCK / 64) + wave, nw = (long long)gridDim.x * (PMH_BLOCK / 64);
  double          wr[SVM_KMAX];
#pragma unroll
  for (int k = 0; k < SVM_KMAX; k++) wr[k] = (lane + 64 * k < d) ? w[lane + 64 * k] : 0.0;
  for (long long i = gw; i < n; i += nw) {
    const double *xr = X + (size_t)i * d;
    double        s  = 0.0;
#pragma unroll
    for (int k = 0; k < SVM_KMAX; k++) {
      const int c = lane + 64 * k;
      if (c < d) s += __builtin_nontemporal_load(&xr[c]) * wr[k];
    }
    s = pmh_wave_sum(s);
    if (lane == 0) Ha[i] = y[i] * s;
  }
}

// ---- d == 64 fast path: 16-byte loads, two rows per wave-instruction (lanes 0-31 row r, lanes 32-63 row r+1), 4-fold unroll ----
typedef double dbl2 __attribute__((ext_vector_type(2))); // native 16-byte vector: accepted by the non-temporal builtins
template <int SVM_UNR>
__global__ __launch_bounds__(PMH_BLOCK) void k_svm_xt64(int n, const double *__restrict__ X, const double *__restrict__ y, const double *__restrict__ a, double *__restrict__ part)
{
  __shared__ double lds[PMH_BLOCK / 64][64];
  const int         lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l2 = lane & 31;
  const long long   gw = (long long)blockIdx.x * (PMH_BLOCK / 64) + wave, nw = (long long)gridDim.x * (PMH_BLOCK / 64);
  double            a0 = 0.0, a1 = 0.0;
  for (long long r0 = gw * 2 * SVM_UNR; r0 < n; r0 += nw * 2 * SVM_UNR) {
    dbl2   v[SVM_UNR];
    double s[SVM_UNR];
#pragma unroll
    for (int u = 0; u < SVM_UNR; u++) {
      const long long i = r0 + 2 * u + half;
      const bool      ok = i < n;
      v[u] = ok ? __builtin_nontemporal_load((const dbl2 *)(X + (size_t)i * 64) + l2) : dbl2{0.0, 0.0};
      s[u] = ok ? y[i] * a[i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < SVM_UNR; u++) {
      a0 += s[u] * v[u].x;
      a1 += s[u] * v[u].y;
    }
  }
  // lanes l and l+32 hold the same two columns (2*l2, 2*l2+1) of different rows: fold, then across the 4 waves in order
  a0 += __shfl_down(a0, 32, 64);
  a1 += __shfl_down(a1, 32, 64);
  if (half == 0) {
    lds[wave][2 * l2]     = a0;
    lds[wave][2 * l2 + 1] = a1;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    double v = lds[0][threadIdx.x];
#pragma unroll
    for (int wv = 1; wv < PMH_BLOCK / 64; wv++) v += lds[wv][threadIdx.x];
    part[(size_t)blockIdx.x * 64 + threadIdx.x] = v;
  }
}

template <int SVM_UNR>
__global__ __launch_bounds__(PMH_BLOCK) void k_svm_x64(int n, const double *__restrict__ X, const double *__restrict__ y, const double *__restrict__ w, double *__restrict__ Ha)
{
  const int       lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l2 = lane & 31;
  const long long gw = (long long)blockIdx.x * (PMH_BLOCK / 64) + wave, nw = (long long)gridDim.x * (PMH_BLOCK / 64);
  const dbl2      wr = ((const dbl2 *)w)[l2];
  for (long long r0 = gw * 2 * SVM_UNR; r0 < n; r0 += nw * 2 * SVM_UNR) {
    dbl2 v[SVM_UNR];
#pragma unroll
    for (int u = 0; u < SVM_UNR; u++) {
      const long long i = r0 + 2 * u + half;
      v[u] = (i < n) ? __builtin_nontemporal_load((const dbl2 *)(X + (size_t)i * 64) + l2) : dbl2{0.0, 0.0};
    }
#pragma unroll
    for (int u = 0; u < SVM_UNR; u++) {
      const long long i = r0 + 2 * u + half;
      double          s = v[u].x * wr.x + v[u].y * wr.y;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) s += __shfl_down(s, o, 32);
      if (l2 == 0 && i < n) Ha[i] = y[i] * s;
    }
  }
}

int SvmDualOp::mult(const double *a, double *Ha)
{
  if (d == 64 && n > 0) {
    // rows in flight per wave-instruction group: 2 x UNR rows of 512 B (16-byte loads, UNR of them outstanding per lane).  UNR decides which wave visits which rows,
    // i.e. the summation order of pass 1 (last-digit differences between UNR values; fixed for a given UNR).  Measured 4 / 8 / 12 / 16 on configs[4]: 464 / 452-488 / 433 / 487
    // iterations per second -- inside the run-to-run spread of the box (the two passes already stream X at the box's copy rate): 4 stays
    static const int unr = getenv("PMH_SVM_UNR") ? atoi(getenv("PMH_SVM_UNR")) : 4;
#define SVM_GO(U)                                                                                                                          \
  do {                                                                                                                                     \
    hipLaunchKernelGGL(k_svm_xt64<U>, dim3(nblocks), dim3(PMH_BLOCK), 0, ctx->stream, n, X, y, a, part);                                   \
    hipLaunchKernelGGL(k_svm_colsum, dim3((d + 3) / 4), dim3(PMH_BLOCK), 0, ctx->stream, nblocks, d, (const double *)part, w);            \
    PMH_HIP(hipGetLastError());                                                                                                            \
    PMH_CHK(pmh_comm_allreduce_sum(ctx, w, (size_t)d));                                                                                    \
    hipLaunchKernelGGL(k_svm_x64<U>, dim3(nblocks), dim3(PMH_BLOCK), 0, ctx->stream, n, X, y, (const double *)w, Ha);                      \
  } while (0)
    if (unr >= 16) SVM_GO(16);
    else if (unr >= 12) SVM_GO(12);
    else if (unr >= 8) SVM_GO(8);
    else SVM_GO(4);
#undef SVM_GO
    PMH_HIP(hipGetLastError());
    return PMH_SUCCESS;
  }
  if (n == 0) return PMH_SUCCESS;
  hipLaunchKernelGGL(k_svm_xt, dim3(nblocks), dim3(PMH_BLOCK), 0, ctx->stream, n, d, X, y, a, part);
  hipLaunchKernelGGL(k_svm_colsum, dim3((d + 3) / 4), dim3(PMH_BLOCK), 0, ctx->stream, nblocks, d, (const double *)part, w);
  PMH_HIP(hipGetLastError());
  PMH_CHK(pmh_comm_allreduce_sum(ctx, w, (size_t)d)); // samples sharded over GPUs: the one exchange step
  hipLaunchKernelGGL(k_svm_x, dim3(nblocks), dim3(PMH_BLOCK), 0, ctx->stream, n, d, X, y, (const double *)w, Ha);
  PMH_HIP(hipGetLastError());
  return PMH_SUCCESS;
}

extern "C" int pmh_op_create_svm_dual(pmh_ctx ctx, int n_local, int d, const double *X_dev, const double *y_dev, pmh_op *op)
{
  PMH_ARG(ctx && op && n_local >= 0 && d >= 1 && d <= 64 * SVM_KMAX && X_dev && y_dev);
  SvmDualOp *o = new SvmDualOp();
  o->ctx       = ctx;
  o->n         = n_local;
  o->d         = d;
  o->X         = X_dev;
  o->y         = y_dev;
  long long nb = ((long long)n_local + 4 * 16 - 1) / (4 * 16); // >= 16 rows per wavefront
  o->nblocks   = (int)(nb < 1 ? 1 : (nb > PMH_MAX_VEC_BLOCKS ? PMH_MAX_VEC_BLOCKS : nb));
  PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)d, (void **)&o->w));
  PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)o->nblocks * d, (void **)&o->part));
  *op = o;
  return PMH_SUCCESS;
}
