// Dense-row Hessian of the PermonSVM-style hinge-loss dual (BASELINE.json configs[4], SURVEY 8d "C5"):
//   H = diag(y) X X' diag(y),  X in R^{N x d} row-major (one sample per row), applied matrix-free as two GEMV
//   passes over X:  w = X'(y o a)  (d-vector),  (H a)_i = y_i (x_i . w).
// PermonSVM is a separate repository (README.md:12) with no code or tests in the reference tree, so this operator
// is "parity unpinned" beyond the MPGP solver that calls it; the oracle side of its test is a numpy restatement.
// HBM-bound: algorithmic bytes per apply 2*8*N*d + 40*N (X read twice; a, y read, Ha written, y read again).
// Samples shard over GPUs by rows; the only exchange is the all-reduce of w (d doubles) between the passes.
#include "pmh_internal.h"
#include "reduce.h"
#include "box_inline.h"

#define SVM_KMAX 4 // d <= 64 * SVM_KMAX
// a launch that streams X once (counted: pmh_op_svm_dual_passes)
#define SVM_PASS(...)                 \
  do {                                \
    npass++;                          \
    hipLaunchKernelGGL(__VA_ARGS__);  \
  } while (0)

struct SvmDualOp : pmh_op_s {
  int           d;
  const double *X, *y;
  double       *w, *part; // w: d; part: [nblocks][d]
  int           nblocks;
  int           mult(const double *a, double *Ha) override;
  // paired passes (d == 64, one GPU): see the block before k_svm_x64_grad
  int           mult_epi(const double *in, double *out, const pmh_vec_epi &e) override;
  int           spec_expansion_ready() override { return next_is == NEXT_XSPEC; }
  enum { NEXT_NONE = 0, NEXT_P, NEXT_XSPEC };
  // what part_next holds the partial sums of X'(y o v) for: the p of the last gradient split / the prepared expansion iterate
  int           next_is = NEXT_NONE;
  const double *next_p = nullptr;
  double       *part_next = nullptr, *feas_part = nullptr, *d_afeas = nullptr, *x_spec = nullptr;
  int           grid_epi = 0;
  long long     npass = 0; // passes over X so far
  ~SvmDualOp() override
  {
    pmh_free(ctx, w);
    pmh_free(ctx, part);
    if (part_next) pmh_free(ctx, part_next), pmh_free(ctx, feas_part), pmh_free(ctx, d_afeas), pmh_free(ctx, x_spec);
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

// ---- paired passes -------------------------------------------------------------------------------------------------------------------------
// One application of H streams X twice (w = X'(y o a), then y o (X w)); an MPGP expansion step applies H twice (Ap = H p, then g = H x+ - b): four passes over
// X, 2.56 GB each for configs[4], and the kernels already run at the box's streaming rate.  But the row that pass 2 of one application has in registers is the
// row pass 1 of the NEXT application needs, and what that next application multiplies is an elementwise function of this pass's result:
//   * the gradient pass g_i = y_i (x_i . w) - b_i knows gf_i, hence p_i = gf_i, hence (y_i p_i) x_i: it accumulates X'(y o p) for the P1 that follows, and the
//     feasible step length QPCFeas(x, p) (which needs no Ap);
//   * the P1 pass  (Ap)_i = y_i (x_i . w)  knows, with that afeas and the fixed alpha, the iterate an expansion step would produce,
//     x+_i = k_expansion_std(x_i, g_i, p_i, (Ap)_i): it stores it and accumulates X'(y o x+) for the gradient that follows IF the host then chooses the
//       expansion.
// A run of expansion steps costs two passes over X per step instead of four; a CG or proportioning step discards the prepared sums and pays the usual passes.
// The driver says what is fresh (pmh_vec_epi::p_fresh / spec_alpha / x_from_spec); the partial sums of the MPGP reductions go to the same rows of the context's
// partials as the separate Vec kernels write, one entry per workgroup of pmh_vec_grid(n) (the elements a workgroup sums are other ones: same values to
// rounding). the two column sums a lane holds (columns 2 l2, 2 l2 + 1 of the rows its half of the wave visited) -> part[workgroup][64], as k_svm_xt64
static __device__ __forceinline__ void svm_fold_cols(double a0, double a1, double (*lds)[64], double *__restrict__ part)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l2 = lane & 31;
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
// the row's dot product in the first lane of its half-wave (the tree of k_svm_x64)
static __device__ __forceinline__ double svm_row_dot(dbl2 v, dbl2 wr)
{
  double s = v.x * wr.x + v.y * wr.y;
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) s += __shfl_down(s, o, 32);
  return s;
}

struct svm_grad_args {
  const double *b, *x_in, *lb, *ub;
  double       *x_out, *g, *gf, *p, *partials, *feas_part, *part_next;
  double        astol;
  int           ld, prow;
};
// pass 2 of g = H x - b with the gradient split, p = gf, the partial sums of (0, |gP|^2, |gc|^2, |gf|^2), QPCFeas(x, p) and X'(y o p)
#define SVM_EU 4
__global__ __launch_bounds__(PMH_BLOCK) void k_svm_x64_grad(int n, const double *__restrict__ X, const double *__restrict__ y, const double *__restrict__ w, svm_grad_args a)
{
  __shared__ double lds[PMH_BLOCK / 64][64];
  __shared__ double red[PMH_BLOCK / 64];
  const int         lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l2 = lane & 31;
  const long long   gw = (long long)blockIdx.x * (PMH_BLOCK / 64) + wave, nw = (long long)gridDim.x * (PMH_BLOCK / 64);
  const dbl2        wr = ((const dbl2 *)w)[l2];
  double            a0 = 0.0, a1 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0, m = INFINITY;
  for (long long r0 = gw * 2 * SVM_EU; r0 < n; r0 += nw * 2 * SVM_EU) {
    dbl2 v[SVM_EU];
#pragma unroll
    for (int u = 0; u < SVM_EU; u++) {
      const long long i = r0 + 2 * u + half;
      v[u] = (i < n) ? __builtin_nontemporal_load((const dbl2 *)(X + (size_t)i * 64) + l2) : dbl2{0.0, 0.0};
    }
    // the rows' dot products land in the first lane of each half-wave; lane j < 2 SVM_EU takes row r0 + j (u = j >> 1, half = j & 1): ONE coalesced load per
    // vector for the 2 SVM_EU rows (a load per row costs the address unit a whole instruction each: measured 2 x the time of the plain pass), asked for before
    // the dot products so that they travel with the rows of X, the elementwise work once
    const long long i   = r0 + lane;
    const bool      act = lane < 2 * SVM_EU && i < n;
    double          yi = 0.0, xi = 0.0, bi = 0.0, li = -INFINITY, ui = INFINITY;
    if (act) {
      yi = y[i], xi = a.x_in[i], bi = a.b[i];
      if (a.lb) li = a.lb[i];
      if (a.ub) ui = a.ub[i];
    }
    double su[SVM_EU], sm = 0.0;
#pragma unroll
    for (int u = 0; u < SVM_EU; u++) su[u] = svm_row_dot(v[u], wr);
#pragma unroll
    for (int u = 0; u < SVM_EU; u++) {
      const double q = __shfl(su[u], (lane & 1) << 5, 64);
      if ((lane >> 1) == u) sm = q;
    }
    double t = 0.0; // y_i p_i: the row's weight in X'(y o p)
    if (act) {
      const double gi = yi * sm - bi;
      double       f, c;
      pmh_box_split_v(xi, gi, li, ui, a.astol, f, c);
      a.g[i] = gi, a.gf[i] = f, a.p[i] = f;
      if (a.x_out) a.x_out[i] = xi;
      const double gPi = f + c;
      acc1 += gPi * gPi, acc2 += c * c, acc3 += f * f;
      m = pmh_box_feas_v(m, xi, f, li, ui);
      t = yi * f;
    }
#pragma unroll
    for (int u = 0; u < SVM_EU; u++) {
      const double tu = __shfl(t, 2 * u + half, 64);
      a0 += tu * v[u].x, a1 += tu * v[u].y;
    }
  }
  svm_fold_cols(a0, a1, lds, a.part_next);
  const double z  = pmh_block_reduce<PMH_RED_SUM>(0.0, red);
  const double r1 = pmh_block_reduce<PMH_RED_SUM>(acc1, red), r2 = pmh_block_reduce<PMH_RED_SUM>(acc2, red), r3 = pmh_block_reduce<PMH_RED_SUM>(acc3, red);
  const double rm = pmh_block_reduce<PMH_RED_MIN>(m, red);
  if (threadIdx.x == 0) {
    double *pp = a.partials + (size_t)a.prow * a.ld + blockIdx.x;
    pp[0] = z, pp[a.ld] = r1, pp[2 * (size_t)a.ld] = r2, pp[3 * (size_t)a.ld] = r3;
    a.feas_part[blockIdx.x] = rm;
  }
}

struct svm_p1_args {
  const double *p, *g, *x, *lb, *ub, *afeas;
  double       *Ap, *partials, *x_spec, *part_next;
  double        alpha, astol;
  int           ld, prow;
};
// pass 2 of Ap = H p with the partial sums of p'Ap, g'p, QPCFeas(x, p); SPEC: + the iterate of the expansion step and X'(y o x+)
template <int SPEC>
__global__ __launch_bounds__(PMH_BLOCK) void k_svm_x64_p1(int n, const double *__restrict__ X, const double *__restrict__ y, const double *__restrict__ w, svm_p1_args a)
{
  __shared__ double lds[PMH_BLOCK / 64][64];
  __shared__ double red[PMH_BLOCK / 64];
  const int         lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l2 = lane & 31;
  const long long   gw = (long long)blockIdx.x * (PMH_BLOCK / 64) + wave, nw = (long long)gridDim.x * (PMH_BLOCK / 64);
  const dbl2        wr = ((const dbl2 *)w)[l2];
  const double      maf = SPEC ? -(*a.afeas) : 0.0, mal = -a.alpha;
  double            a0 = 0.0, a1 = 0.0, s0 = 0.0, s1 = 0.0, m = INFINITY;
  for (long long r0 = gw * 2 * SVM_EU; r0 < n; r0 += nw * 2 * SVM_EU) {
    dbl2 v[SVM_EU];
#pragma unroll
    for (int u = 0; u < SVM_EU; u++) {
      const long long i = r0 + 2 * u + half;
      v[u] = (i < n) ? __builtin_nontemporal_load((const dbl2 *)(X + (size_t)i * 64) + l2) : dbl2{0.0, 0.0};
    }
    // (as in k_svm_x64_grad: lane j < 2 SVM_EU takes row r0 + j, its scalars asked for up front)
    const long long i   = r0 + lane;
    const bool      act = lane < 2 * SVM_EU && i < n;
    double          yi = 0.0, pi = 0.0, gi = 0.0, xi = 0.0, li = -INFINITY, ui = INFINITY;
    if (act) {
      yi = y[i], pi = a.p[i], gi = a.g[i], xi = a.x[i];
      if (a.lb) li = a.lb[i];
      if (a.ub) ui = a.ub[i];
    }
    double su[SVM_EU], sm = 0.0;
#pragma unroll
    for (int u = 0; u < SVM_EU; u++) su[u] = svm_row_dot(v[u], wr);
#pragma unroll
    for (int u = 0; u < SVM_EU; u++) {
      const double q = __shfl(su[u], (lane & 1) << 5, 64);
      if ((lane >> 1) == u) sm = q;
    }
    double t = 0.0; // y_i x+_i: the row's weight in X'(y o x+)
    if (act) {
      const double api = yi * sm;
      a.Ap[i] = api;
      s0 += pi * api, s1 += gi * pi;
      m = pmh_box_feas_v(m, xi, pi, li, ui);
      if (SPEC) { // k_expansion_std (mpgp.hip) on this entry
        const double xs = xi + maf * pi, gs = gi + maf * api;
        double       f, c;
        pmh_box_split_v(xs, gs, li, ui, a.astol, f, c);
        const double r = pmh_box_reduced_v(xs, f, li, ui, a.lb != nullptr, a.ub != nullptr, a.alpha), xn = xs + mal * r;
        a.x_spec[i] = xn;
        t = yi * xn;
      }
    }
    if (SPEC) {
#pragma unroll
      for (int u = 0; u < SVM_EU; u++) {
        const double tu = __shfl(t, 2 * u + half, 64);
        a0 += tu * v[u].x, a1 += tu * v[u].y;
      }
    }
  }
  if (SPEC) svm_fold_cols(a0, a1, lds, a.part_next);
  const double r0s = pmh_block_reduce<PMH_RED_SUM>(s0, red), r1s = pmh_block_reduce<PMH_RED_SUM>(s1, red), rm = pmh_block_reduce<PMH_RED_MIN>(m, red);
  if (threadIdx.x == 0) {
    double *pp = a.partials + (size_t)a.prow * a.ld + blockIdx.x;
    pp[0] = r0s, pp[a.ld] = r1s, pp[2 * (size_t)a.ld] = rm;
  }
}

// w[c] = sum over workgroups of part[b][c] (as k_svm_colsum) and, in the last workgroup, afeas = min over workgroups of feas_part (exact: a min has no order)
__global__ __launch_bounds__(PMH_BLOCK) void k_svm_colsum_feas(int nblocks, const double *__restrict__ part, double *__restrict__ w, const double *__restrict__ feas_part, double *__restrict__ afeas)
{
  __shared__ double red[PMH_BLOCK / 64];
  if (blockIdx.x == gridDim.x - 1) {
    double m = INFINITY;
    for (int b = threadIdx.x; b < nblocks; b += PMH_BLOCK) m = fmin(m, feas_part[b]);
    m = pmh_block_reduce<PMH_RED_MIN>(m, red);
    if (threadIdx.x == 0) *afeas = m;
    return;
  }
  const int lane = threadIdx.x & 63, c = blockIdx.x * (PMH_BLOCK / 64) + (threadIdx.x >> 6);
  if (c >= 64) return;
  double v = 0.0;
  for (int b = lane; b < nblocks; b += 64) v += part[(size_t)b * 64 + c];
  v = pmh_wave_sum(v);
  if (lane == 0) w[c] = v;
}

int SvmDualOp::mult_epi(const double *in, double *out, const pmh_vec_epi &e)
{
  // (the switch may change between two solves of one process: pmh_set_knob("svm_pairing"), initial value from PMH_SVM_NO_PAIRING -- no getenv on the
  // per-product path.  It is process-wide state that every rank of a job must set alike: ranks that disagree would issue different sequences of collectives)
  if (n <= 0 && pmh_comm_on(ctx)) return pmh_set_error(PMH_ERR_ARG, "SVM dual operator: this rank holds no samples; with a communicator every rank needs at least one row (an empty shard would skip the collectives the other ranks issue)");
  if (!pmh_knobs().svm_pairing || d != 64 || n <= 0) return PMH_EPI_UNSUPPORTED;
  // several GPUs (samples sharded by rows): the 64 column sums w and, where the next pass uses it, the feasible step length afeas are completed across the
  // ranks between the passes -- the same exchange step as the lone application's, one (+ one 8-byte MIN) per pass
  if (!part_next) {
    grid_epi = pmh_vec_grid(n); // one partial sum per workgroup, where pmh_finalize_partials expects them
    PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)grid_epi * 64, (void **)&part_next));
    PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)grid_epi, (void **)&feas_part));
    PMH_CHK(pmh_malloc(ctx, sizeof(double), (void **)&d_afeas));
    PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)n, (void **)&x_spec));
  }
  const int have = next_is;
  next_is        = NEXT_NONE;
  if (e.kind == PMH_VEPI_GRAD_SPLIT) {
    const bool spec = e.x_from_spec && have == NEXT_XSPEC;
    if (e.x_from_spec && !spec) return pmh_set_error(PMH_ERR_STATE, "SVM dual operator: the driver asks for the prepared expansion step, none is prepared");
    // (the min it also takes is not used here)
    if (spec) hipLaunchKernelGGL(k_svm_colsum_feas, dim3(17), dim3(PMH_BLOCK), 0, ctx->stream, grid_epi, (const double *)part_next, w, (const double *)feas_part, d_afeas);
    else {
      SVM_PASS(k_svm_xt64<4>, dim3(nblocks), dim3(PMH_BLOCK), 0, ctx->stream, n, X, y, in, part);
      hipLaunchKernelGGL(k_svm_colsum, dim3(16), dim3(PMH_BLOCK), 0, ctx->stream, nblocks, d, (const double *)part, w);
    }
    PMH_HIP(hipGetLastError());
    PMH_CHK(pmh_comm_allreduce_sum(ctx, w, 64));
    svm_grad_args a;
    a.b = e.b, a.x_in = spec ? (const double *)x_spec : in, a.lb = e.lb, a.ub = e.ub, a.x_out = spec ? e.x_out : nullptr, a.g = out, a.gf = e.gf, a.p = e.p;
    a.partials = e.partials, a.feas_part = feas_part, a.part_next = part_next, a.astol = e.astol, a.ld = e.ld, a.prow = e.prow;
    if (spec && !e.x_out) return pmh_set_error(PMH_ERR_ARG, "SVM dual operator: x_from_spec needs x_out");
    SVM_PASS(k_svm_x64_grad, dim3(grid_epi), dim3(PMH_BLOCK), 0, ctx->stream, n, X, y, (const double *)w, a);
    PMH_HIP(hipGetLastError());
    next_is = NEXT_P, next_p = e.p;
    return PMH_SUCCESS;
  }
  if (e.kind == PMH_VEPI_P1) {
    const bool paired = e.p_fresh && have == NEXT_P && next_p == in;
    if (paired) hipLaunchKernelGGL(k_svm_colsum_feas, dim3(17), dim3(PMH_BLOCK), 0, ctx->stream, grid_epi, (const double *)part_next, w, (const double *)feas_part, d_afeas);
    else {
      SVM_PASS(k_svm_xt64<4>, dim3(nblocks), dim3(PMH_BLOCK), 0, ctx->stream, n, X, y, in, part);
      hipLaunchKernelGGL(k_svm_colsum, dim3(16), dim3(PMH_BLOCK), 0, ctx->stream, nblocks, d, (const double *)part, w);
    }
    PMH_HIP(hipGetLastError());
    PMH_CHK(pmh_comm_allreduce_sum(ctx, w, 64));
    if (paired) PMH_CHK(pmh_comm_allreduce_min(ctx, d_afeas, 1)); // the P1 pass forms the expansion iterate with it (k_svm_x64_p1<1>)
    svm_p1_args a;
    a.p = in, a.g = e.g, a.x = e.xx, a.lb = e.lb, a.ub = e.ub, a.afeas = d_afeas, a.Ap = out, a.partials = e.partials, a.x_spec = x_spec, a.part_next = part_next;
    a.alpha = e.spec_alpha, a.astol = e.astol, a.ld = e.ld, a.prow = e.prow;
    const bool spec = paired && e.spec_alpha > 0.0; // afeas is known before this pass only when the gradient pass computed it
    if (spec) SVM_PASS(k_svm_x64_p1<1>, dim3(grid_epi), dim3(PMH_BLOCK), 0, ctx->stream, n, X, y, (const double *)w, a);
    else SVM_PASS(k_svm_x64_p1<0>, dim3(grid_epi), dim3(PMH_BLOCK), 0, ctx->stream, n, X, y, (const double *)w, a);
    PMH_HIP(hipGetLastError());
    if (spec) next_is = NEXT_XSPEC;
    return PMH_SUCCESS;
  }
  return PMH_EPI_UNSUPPORTED;
}

int SvmDualOp::mult(const double *a, double *Ha)
{
  next_is = NEXT_NONE; // (whatever was prepared belonged to the MPGP driver's vectors)
  if (n == 0 && pmh_comm_on(ctx)) return pmh_set_error(PMH_ERR_ARG, "SVM dual operator: this rank holds no samples; with a communicator every rank needs at least one row");
  if (d == 64 && n > 0) {
    // rows in flight per wave-instruction group: 2 x UNR rows of 512 B (16-byte loads, UNR of them outstanding per lane).  UNR decides which wave visits which
    // rows, i.e. the summation order of pass 1 (last-digit differences between UNR values; fixed for a given UNR).  Measured 4 / 8 / 12 / 16 on configs[4]: 464
    // / 452-488 / 433 / 487 iterations per second -- inside the run-to-run spread of the box (the two passes already stream X at the box's copy rate): 4 stays
    const int unr = 4; // rows in flight per wave of the two passes (4 / 8 / 12 / 16 measured: inside the run-to-run spread)
#define SVM_GO(U)                                                                                                                          \
  do {                                                                                                                                     \
    SVM_PASS(k_svm_xt64<U>, dim3(nblocks), dim3(PMH_BLOCK), 0, ctx->stream, n, X, y, a, part);                                   \
    hipLaunchKernelGGL(k_svm_colsum, dim3((d + 3) / 4), dim3(PMH_BLOCK), 0, ctx->stream, nblocks, d, (const double *)part, w);            \
    PMH_HIP(hipGetLastError());                                                                                                            \
    PMH_CHK(pmh_comm_allreduce_sum(ctx, w, (size_t)d));                                                                                    \
    SVM_PASS(k_svm_x64<U>, dim3(nblocks), dim3(PMH_BLOCK), 0, ctx->stream, n, X, y, (const double *)w, Ha);                      \
  } while (0)
    if (unr >= 16) SVM_GO(16);
    else if (unr >= 12) SVM_GO(12);
    else if (unr >= 8) SVM_GO(8);
    else SVM_GO(4);
#undef SVM_GO
    PMH_HIP(hipGetLastError());
    return PMH_SUCCESS;
  }
  if (n == 0) {
    if (pmh_comm_on(ctx)) return pmh_set_error(PMH_ERR_ARG, "SVM dual operator: this rank holds no samples; with a communicator every rank needs at least one row");
    return PMH_SUCCESS;
  }
  SVM_PASS(k_svm_xt, dim3(nblocks), dim3(PMH_BLOCK), 0, ctx->stream, n, d, X, y, a, part);
  hipLaunchKernelGGL(k_svm_colsum, dim3((d + 3) / 4), dim3(PMH_BLOCK), 0, ctx->stream, nblocks, d, (const double *)part, w);
  PMH_HIP(hipGetLastError());
  PMH_CHK(pmh_comm_allreduce_sum(ctx, w, (size_t)d)); // samples sharded over GPUs: the one exchange step
  SVM_PASS(k_svm_x, dim3(nblocks), dim3(PMH_BLOCK), 0, ctx->stream, n, d, X, y, (const double *)w, Ha);
  PMH_HIP(hipGetLastError());
  return PMH_SUCCESS;
}

extern "C" int pmh_op_svm_dual_passes(pmh_op op, long long *passes)
{
  SvmDualOp *o = dynamic_cast<SvmDualOp *>(op);
  PMH_ARG(o && passes);
  *passes = o->npass;
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
