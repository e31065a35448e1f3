// Vec and QPC-box kernels of the PERMON QPS path for gfx950: HBM-bound streaming kernels,
// one coalesced pass each, deterministic two-stage reductions.
#include <algorithm>
#include <cstdlib>

#include "pmh_internal.h"
#include "reduce.h"

int pmh_vec_grid(int n)
{
  const int epb = 256; // elements per workgroup of the streaming Vec kernels (grid capped at PMH_MAX_VEC_BLOCKS): one element per thread up to 524 288 entries -- dual vectors of ~1e5 entries ran on 50 workgroups with 2048 each, 4-5 us above the launch floor for the five-operand kernels
  long long b = ((long long)n + epb - 1) / epb;
  if (b < 1) b = 1;
  if (b > PMH_MAX_VEC_BLOCKS) b = PMH_MAX_VEC_BLOCKS;
  return (int)b;
}

#define GRID_STRIDE(i, n) for (long long i = (long long)blockIdx.x * PMH_BLOCK + threadIdx.x; i < (n); i += (long long)gridDim.x * PMH_BLOCK)

// ---- finalise block partials -----------------------------------------------------------------------------
struct pmh_ops8 {
  int op[PMH_MAX_RED];
  int slot[PMH_MAX_RED]; // destination scalar slot of every quantity
};

#define PMH_FIN_THREADS 1024
__global__ __launch_bounds__(PMH_FIN_THREADS) void k_finalize(const double *__restrict__ partials, int ld, int nblocks, int K, pmh_ops8 ops, double *__restrict__ d_scal, double *__restrict__ h_scal, const int *__restrict__ halt, int *__restrict__ post_inc)
{
  __shared__ double lds[PMH_MAX_RED][PMH_FIN_THREADS / 64];
  if (halt && *halt) return; // speculative chain stopped: keep the scalars of the last valid state
  const int         lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double            v[PMH_MAX_RED];
  // all K strided partial sums are accumulated together: their loads overlap (one latency, not K)
#pragma unroll
  for (int k = 0; k < PMH_MAX_RED; k++) v[k] = (ops.op[k] == PMH_RED_SUM) ? 0.0 : INFINITY;
  for (int i = threadIdx.x; i < nblocks; i += PMH_FIN_THREADS) {
#pragma unroll
    for (int k = 0; k < PMH_MAX_RED; k++)
      if (k < K) {
        const double p = partials[(size_t)k * ld + i];
        v[k]           = (ops.op[k] == PMH_RED_SUM) ? (v[k] + p) : fmin(v[k], p);
      }
  }
#pragma unroll
  for (int k = 0; k < PMH_MAX_RED; k++)
    if (k < K) {
      v[k] = (ops.op[k] == PMH_RED_SUM) ? pmh_wave_sum(v[k]) : pmh_wave_min(v[k]);
      if (lane == 0) lds[k][wave] = v[k];
    }
  __syncthreads();
  if (threadIdx.x < K) {
    const int k = threadIdx.x;
    double    r = lds[k][0];
    for (int w = 1; w < PMH_FIN_THREADS / 64; w++) r = (ops.op[k] == PMH_RED_SUM) ? (r + lds[k][w]) : fmin(r, lds[k][w]);
    d_scal[ops.slot[k]] = r;
    h_scal[ops.slot[k]] = r;
  }
  if (post_inc && threadIdx.x == 0) { // device-side iteration / CG-step counters of the speculative chain
    post_inc[0]++;
    post_inc[1]++;
  }
}

int pmh_finalize_partials(pmh_ctx ctx, const double *partials, int ld, int nblocks, int K, const int *ops, int scal_base, const int *halt, int *post_inc)
{
  pmh_ops8 o;
  for (int k = 0; k < PMH_MAX_RED; k++) o.op[k] = (k < K) ? ops[k] : 0, o.slot[k] = scal_base + k;
  hipLaunchKernelGGL(k_finalize, dim3(1), dim3(PMH_FIN_THREADS), 0, ctx->stream, partials, ld, nblocks, K, o, ctx->d_scal, ctx->h_scal, halt, post_inc);
  PMH_HIP(hipGetLastError());
  if (ctx->dist_scalars && pmh_comm_on(ctx)) {
    // row-distributed vectors: complete the reductions across ranks (VecDot / VecNorm / QPCFeas MPI_Allreduce)
    // as ONE grouped exchange over the K device scalars, then refresh the pinned host mirror
    if (halt) return pmh_set_error(PMH_ERR_STATE, "the speculative device-side chain is not available with row-distributed vectors");
    PMH_CHK(pmh_comm_allreduce_scalars(ctx, ctx->d_scal + scal_base, K, ops));
    PMH_HIP(hipMemcpyAsync(ctx->h_scal + scal_base, ctx->d_scal + scal_base, sizeof(double) * K, hipMemcpyDeviceToHost, ctx->stream));
  }
  return PMH_SUCCESS;
}

// K <= PMH_MAX_RED quantities (rows of `partials`) into arbitrary scalar slots: every quantity is reduced exactly as pmh_finalize_partials reduces it (the
// reduction of a row does not depend on the others), so finalising two groups in one launch changes no bit.  Not for row-distributed vectors.
int pmh_finalize_partials_slots(pmh_ctx ctx, const double *partials, int ld, int nblocks, int K, const int *ops, const int *slots)
{
  PMH_ARG(K >= 1 && K <= PMH_MAX_RED);
  if (ctx->dist_scalars && pmh_comm_on(ctx)) return pmh_set_error(PMH_ERR_STATE, "pmh_finalize_partials_slots: not with row-distributed vectors");
  pmh_ops8 o;
  for (int k = 0; k < PMH_MAX_RED; k++) o.op[k] = (k < K) ? ops[k] : 0, o.slot[k] = (k < K) ? slots[k] : 0;
  hipLaunchKernelGGL(k_finalize, dim3(1), dim3(PMH_FIN_THREADS), 0, ctx->stream, partials, ld, nblocks, K, o, ctx->d_scal, ctx->h_scal, (const int *)nullptr, (int *)nullptr);
  PMH_HIP(hipGetLastError());
  return PMH_SUCCESS;
}

// ---- BLAS-1 ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PMH_BLOCK) void k_axpy(long long n, double *__restrict__ y, double a, const double *__restrict__ x)
{
  GRID_STRIDE(i, n) y[i] += a * x[i];
}
__global__ __launch_bounds__(PMH_BLOCK) void k_aypx(long long n, double *__restrict__ y, double a, const double *__restrict__ x)
{
  GRID_STRIDE(i, n) y[i] = x[i] + a * y[i];
}
__global__ __launch_bounds__(PMH_BLOCK) void k_waxpy(long long n, double *w, double a, const double *x, const double *y)
{
  GRID_STRIDE(i, n) w[i] = a * x[i] + y[i];
}
__global__ __launch_bounds__(PMH_BLOCK) void k_scale(long long n, double *x, double a)
{
  GRID_STRIDE(i, n) x[i] *= a;
}
__global__ __launch_bounds__(PMH_BLOCK) void k_set(long long n, double *x, double a)
{
  GRID_STRIDE(i, n) x[i] = a;
}
__global__ __launch_bounds__(PMH_BLOCK) void k_dot(long long n, const double *__restrict__ x, const double *__restrict__ y, double *__restrict__ partials)
{
  __shared__ double lds[PMH_BLOCK / 64];
  double            s = 0.0;
  GRID_STRIDE(i, n) s += x[i] * y[i];
  s = pmh_block_reduce<PMH_RED_SUM>(s, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

#define LAUNCH_VEC(kern, n, ...) \
  do { \
    if ((n) > 0) hipLaunchKernelGGL(kern, dim3(pmh_vec_grid(n)), dim3(PMH_BLOCK), 0, ctx->stream, (long long)(n), __VA_ARGS__); \
    PMH_HIP(hipGetLastError()); \
  } while (0)

extern "C" int pmh_vec_axpy(pmh_ctx ctx, int n, double *y, double a, const double *x)
{
  PMH_ARG(ctx && n >= 0);
  LAUNCH_VEC(k_axpy, n, y, a, x);
  return PMH_SUCCESS;
}
extern "C" int pmh_vec_aypx(pmh_ctx ctx, int n, double *y, double a, const double *x)
{
  PMH_ARG(ctx && n >= 0);
  LAUNCH_VEC(k_aypx, n, y, a, x);
  return PMH_SUCCESS;
}
extern "C" int pmh_vec_waxpy(pmh_ctx ctx, int n, double *w, double a, const double *x, const double *y)
{
  PMH_ARG(ctx && n >= 0);
  LAUNCH_VEC(k_waxpy, n, w, a, x, y);
  return PMH_SUCCESS;
}
extern "C" int pmh_vec_scale(pmh_ctx ctx, int n, double *x, double a)
{
  PMH_ARG(ctx && n >= 0);
  LAUNCH_VEC(k_scale, n, x, a);
  return PMH_SUCCESS;
}
extern "C" int pmh_vec_set(pmh_ctx ctx, int n, double *x, double a)
{
  PMH_ARG(ctx && n >= 0);
  LAUNCH_VEC(k_set, n, x, a);
  return PMH_SUCCESS;
}
extern "C" int pmh_vec_copy(pmh_ctx ctx, int n, const double *x, double *y)
{
  PMH_ARG(ctx && n >= 0);
  return pmh_memcpy_d2d(ctx, y, x, sizeof(double) * (size_t)n);
}

int pmh_k_dot_partials(pmh_ctx ctx, int n, const double *x, const double *y, int slot)
{
  const int ops[1] = {PMH_RED_SUM};
  int       nb     = pmh_vec_grid(n);
  if (n > 0) {
    hipLaunchKernelGGL(k_dot, dim3(nb), dim3(PMH_BLOCK), 0, ctx->stream, (long long)n, x, y, ctx->d_partials);
    PMH_HIP(hipGetLastError());
  } else {
    nb = 0;
  }
  PMH_CHK(pmh_finalize_partials(ctx, ctx->d_partials, ctx->partials_cap, nb, 1, ops, slot));
  return PMH_SUCCESS;
}

// Vectors at this tier are either rank-local (primal blocks) or REPLICATED on every GPU (dual space): a dot product
// is therefore complete locally and identical on every rank -- no MPI_Allreduce counterpart.
extern "C" int pmh_vec_dot(pmh_ctx ctx, int n, const double *x, const double *y, double *result_host)
{
  PMH_ARG(ctx && n >= 0 && result_host);
  PMH_CHK(pmh_k_dot_partials(ctx, n, x, y, 0));
  return pmh_host_scalar(ctx, 0, result_host);
}

extern "C" int pmh_vec_norm2(pmh_ctx ctx, int n, const double *x, double *result_host)
{
  double s;
  PMH_CHK(pmh_vec_dot(ctx, n, x, x, &s));
  *result_host = sqrt(s);
  return PMH_SUCCESS;
}

// ---- QPC box (src/qpc/impls/box/qpcbox.c) ----------------------------------------------------------------------
// QPCProject_Box qpcbox.c:290-305 (+ the VecCopy of qpc.c:479)
__global__ __launch_bounds__(PMH_BLOCK) void k_box_project(long long n, const double *x, const double *lb, const double *ub, double *Px)
{
  GRID_STRIDE(i, n)
  {
    double v = x[i];
    if (lb) {
      double l = lb[i];
      v        = (v > l) ? v : l;
      if (ub) {
        double u = ub[i];
        v        = (v < u) ? v : u;
      }
    } else if (ub) {
      double u = ub[i];
      v        = (v < u) ? v : u;
    }
    Px[i] = v;
  }
}

// QPCFeas_Box qpcbox.c:104-146
__global__ __launch_bounds__(PMH_BLOCK) void k_box_feas(long long n, const double *__restrict__ x, const double *__restrict__ d, const double *__restrict__ lb, const double *__restrict__ ub, double *__restrict__ partials)
{
  __shared__ double lds[PMH_BLOCK / 64];
  double            a = INFINITY;
  GRID_STRIDE(i, n)
  {
    double di = d[i], xi = x[i];
    if (di > 0. && lb) {
      double l = lb[i];
      if (l > -INFINITY) a = fmin(a, (xi - l) / di);
    }
    if (di < 0. && ub) {
      double u = ub[i];
      if (u < INFINITY) a = fmin(a, (xi - u) / di);
    }
  }
  a = pmh_block_reduce<PMH_RED_MIN>(a, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = a;
}

// QPCGrads (gf=g, gc=0: qpc.c:551-552) + QPCGrads_Box qpcbox.c:41-55
__global__ __launch_bounds__(PMH_BLOCK) void k_box_grads(long long n, const double *__restrict__ x, const double *__restrict__ g, const double *__restrict__ lb, const double *__restrict__ ub, double astol, double *__restrict__ gf, double *__restrict__ gc)
{
  GRID_STRIDE(i, n)
  {
    double xi = x[i], gi = g[i], f = gi, c = 0.0;
    if (lb && fabs(xi - lb[i]) <= astol) {
      f = 0.0;
      c = (gi < 0.0) ? gi : 0.0;
    } else if (ub && fabs(xi - ub[i]) <= astol) {
      f = 0.0;
      c = (gi > 0.0) ? gi : 0.0;
    }
    gf[i] = f;
    gc[i] = c;
  }
}

// QPCGradReduced (gr=gf: qpc.c:600) + QPCGradReduced_Box qpcbox.c:86-92
__global__ __launch_bounds__(PMH_BLOCK) void k_box_gradreduced(long long n, const double *__restrict__ x, const double *__restrict__ gf, const double *__restrict__ lb, const double *__restrict__ ub, double alpha, double *__restrict__ gr)
{
  GRID_STRIDE(i, n)
  {
    double f = gf[i], r = f;
    if (lb && f > 0.0) {
      double t = (x[i] - lb[i]) / alpha;
      r        = (f < t) ? f : t;
    } else if (ub && f < 0.0) {
      double t = (x[i] - ub[i]) / alpha;
      r        = (f < t) ? t : f;
    }
    gr[i] = r;
  }
}

extern "C" int pmh_qpc_box_project(pmh_ctx ctx, int n, const double *x, const double *lb, const double *ub, double *Px)
{
  PMH_ARG(ctx && n >= 0);
  LAUNCH_VEC(k_box_project, n, x, lb, ub, Px);
  return PMH_SUCCESS;
}

extern "C" int pmh_qpc_box_feas(pmh_ctx ctx, int n, const double *x, const double *d, const double *lb, const double *ub, double *alpha_host)
{
  PMH_ARG(ctx && n >= 0 && alpha_host);
  const int ops[1] = {PMH_RED_MIN};
  int       nb     = pmh_vec_grid(n);
  if (n > 0) {
    hipLaunchKernelGGL(k_box_feas, dim3(nb), dim3(PMH_BLOCK), 0, ctx->stream, (long long)n, x, d, lb, ub, ctx->d_partials);
    PMH_HIP(hipGetLastError());
  } else {
    nb = 0;
  }
  PMH_CHK(pmh_finalize_partials(ctx, ctx->d_partials, ctx->partials_cap, nb, 1, ops, 0));
  return pmh_host_scalar(ctx, 0, alpha_host);
}

extern "C" int pmh_qpc_box_grads(pmh_ctx ctx, int n, const double *x, const double *g, const double *lb, const double *ub, double astol, double *gf, double *gc)
{
  PMH_ARG(ctx && n >= 0);
  LAUNCH_VEC(k_box_grads, n, x, g, lb, ub, astol, gf, gc);
  return PMH_SUCCESS;
}

extern "C" int pmh_qpc_box_gradreduced(pmh_ctx ctx, int n, const double *x, const double *gf, const double *lb, const double *ub, double alpha, double *gr)
{
  PMH_ARG(ctx && n >= 0);
  LAUNCH_VEC(k_box_gradreduced, n, x, gf, lb, ub, alpha, gr);
  return PMH_SUCCESS;
}

__global__ __launch_bounds__(PMH_BLOCK) void k_scatter_is(long long nis, const int *is, const double *sub, double *full)
{
  GRID_STRIDE(k, nis) full[is[k]] = sub[k];
}

// qpc->is (qpc.c:416-437): the kernels act on the IS sub-vector; equivalently (ex2's two goldens are
// identical, output/ex2_1_infinite-{false,true}.out) the bound is +-inf outside the index set.
extern "C" int pmh_qpc_box_expand_is(pmh_ctx ctx, int n, int nis, const int *is_host, const double *bound_sub, double fill, double *bound_full)
{
  PMH_ARG(ctx && n >= 0 && nis >= 0 && nis <= n);
  for (int k = 0; k < nis; k++) PMH_ARG(is_host[k] >= 0 && is_host[k] < n);
  int *d_is = nullptr;
  PMH_HIP(hipMalloc((void **)&d_is, sizeof(int) * (size_t)(nis ? nis : 1)));
  PMH_CHK(pmh_memcpy_h2d(ctx, d_is, is_host, sizeof(int) * (size_t)nis));
  LAUNCH_VEC(k_set, n, bound_full, fill);
  LAUNCH_VEC(k_scatter_is, nis, (const int *)d_is, bound_sub, bound_full);
  PMH_HIP(hipStreamSynchronize(ctx->stream));
  PMH_HIP(hipFree(d_is));
  return PMH_SUCCESS;
}
