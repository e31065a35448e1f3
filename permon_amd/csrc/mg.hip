// Multigrid V-cycle preconditioner for the block CG inside MATINV.
// The reference's iterative MATINV path is a PETSc KSP whose PC is whatever -mat_inv_pc_type names
// (src/mat/impls/inv/matinv.c: MatInvGetKSP / MatInvSetUp); for the 3-D elasticity blocks of BASELINE configs[2] the
// Jacobi default needs ~450-1000 CG iterations per K^+ application.  This is the PCMG equivalent on the device:
// Galerkin hierarchy A_{l+1} = P_l' A_l P_l handed over as CSR (built by the caller), Chebyshev/Jacobi smoothing with
// PETSc's eigenvalue window [lo, hi] x lambda_max(D^-1 A) (KSPCHEBYSHEV as PCMG/PCGAMG configure it), and block-wise
// dense pseudo-inverses on the coarsest level (floating subdomains stay singular down the hierarchy: the prolongation
// reproduces the rigid-body modes).  Pre- and post-smoother are the same polynomial, so the cycle is symmetric positive
// (semi-)definite and valid inside CG.
// Every kernel is HBM bound, so the cycle is built to move few bytes: level operators with 3x3 block structure run on the
// block kernel of bsr.hip (one column index per block), and with precision = PMH_MG_FP32 the whole cycle (operators and
// vectors) is single precision -- it only preconditions the fp64 CG, whose residual and solution stay fp64.
#include <chrono>
#include <thread>
#include <algorithm>
#include <cstdio>

#include "pmh_internal.h"
#include "mg_internal.h"

// position of the LAST entry (row i, column i) of a CSR row, -1 if there is none: 8 lanes walk the row together (one thread per row read 81 entries one after the
// other: 2 ms for the 2 M rows of configs[2])
static __device__ __forceinline__ int mg_diag_pos8(const int *__restrict__ rowptr, const int *__restrict__ col, int i, int lane8)
{
  int best = -1;
  for (int k = rowptr[i] + lane8; k < rowptr[i + 1]; k += 8)
    if (col[k] == i) best = k;
  best = max(best, __shfl_xor(best, 4, 8));
  best = max(best, __shfl_xor(best, 2, 8));
  best = max(best, __shfl_xor(best, 1, 8));
  return best;
}
template <typename TV>
__global__ __launch_bounds__(PMH_BLOCK) void k_mg_dinv(int n, const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val, TV *__restrict__ dinv)
{
  const int lane8 = threadIdx.x & 7;
  for (int i0 = (blockIdx.x * PMH_BLOCK + threadIdx.x) >> 3; i0 < ((n + 7) & ~7); i0 += (gridDim.x * PMH_BLOCK) >> 3) { // uniform trip count within every group of 8 lanes
    const int    i = min(i0, n - 1);
    const int    k = mg_diag_pos8(rowptr, col, i, lane8);
    const double d = (k >= 0) ? val[k] : 0.0;
    if (lane8 == 0 && i0 < n) dinv[i] = (TV)((d != 0.0) ? 1.0 / d : 1.0);
  }
}

// first Chebyshev step from a zero guess: r = D^-1 b, d = r/theta, x = d
template <typename TV>
__global__ __launch_bounds__(PMH_BLOCK) void k_cheb_first_zero(int n, const int *__restrict__ halt, const TV *__restrict__ dinv, const TV *__restrict__ b, TV itheta, TV *__restrict__ r, TV *__restrict__ d, TV *__restrict__ x)
{
  if (halt && *halt) return;
  for (int i = blockIdx.x * PMH_BLOCK + threadIdx.x; i < n; i += gridDim.x * PMH_BLOCK) {
    const TV ri = dinv[i] * b[i], di = ri * itheta;
    r[i] = ri;
    d[i] = di;
    x[i] = di;
  }
}

// fused cycle: d0 = D^-1 b / theta (the first pre-smoothing direction); on the fine level of the fp32 cycle b arrives in
// fp64 and its fp32 copy is produced on the way
template <typename TV, typename TB>
__global__ __launch_bounds__(PMH_BLOCK) void k_cheb_d0(int n, const int *__restrict__ halt, const TV *__restrict__ dinv, const TB *__restrict__ b, TV itheta, TV *__restrict__ d, TV *__restrict__ bcopy)
{
  if (halt && *halt) return;
  for (int i = blockIdx.x * PMH_BLOCK + threadIdx.x; i < n; i += gridDim.x * PMH_BLOCK) {
    const TV bi = (TV)b[i];
    d[i] = dinv[i] * bi * itheta;
    if (bcopy) bcopy[i] = bi;
  }
}

// first step from the current x (t = A x): r = D^-1 (b - t), d = r/theta, x += d
template <typename TV>
__global__ __launch_bounds__(PMH_BLOCK) void k_cheb_first(int n, const int *__restrict__ halt, const TV *__restrict__ dinv, const TV *__restrict__ b, const TV *__restrict__ t, TV itheta, TV *__restrict__ r, TV *__restrict__ d, TV *__restrict__ x)
{
  if (halt && *halt) return;
  for (int i = blockIdx.x * PMH_BLOCK + threadIdx.x; i < n; i += gridDim.x * PMH_BLOCK) {
    const TV ri = dinv[i] * (b[i] - t[i]), di = ri * itheta;
    r[i] = ri;
    d[i] = di;
    x[i] += di;
  }
}

// later steps (t = A d): r -= D^-1 t, d = c1 d + c2 r, x += d
template <typename TV>
__global__ __launch_bounds__(PMH_BLOCK) void k_cheb_step(int n, const int *__restrict__ halt, const TV *__restrict__ dinv, const TV *__restrict__ t, TV c1, TV c2, TV *__restrict__ r, TV *__restrict__ d, TV *__restrict__ x)
{
  if (halt && *halt) return;
  for (int i = blockIdx.x * PMH_BLOCK + threadIdx.x; i < n; i += gridDim.x * PMH_BLOCK) {
    const TV ri = r[i] - dinv[i] * t[i], di = c1 * d[i] + c2 * ri;
    r[i] = ri;
    d[i] = di;
    x[i] += di;
  }
}

// restriction b_c = R t with R = P' as CSR (<= 27 entries per row): 8 lanes per coarse row, shuffle-tree sum (fixed order).
// dinv_c != NULL: the first smoothing direction of the coarse level, d_c = D_c^-1 b_c / theta_c, is written on the way
// (saves the k_cheb_d0 launch of every smoothed coarse level: these levels are launch-latency bound).
template <typename TV>
__global__ __launch_bounds__(PMH_BLOCK) void k_mg_restrict(int nc, const int *__restrict__ halt, const int *__restrict__ rowptr, const int *__restrict__ col, const TV *__restrict__ val, const TV *__restrict__ t, TV *__restrict__ bc,
                                                         const TV *__restrict__ dinv_c, TV itheta_c, TV *__restrict__ d_c)
{
  const int hlt  = halt ? *halt : 0; // fetched together with the first row pointers: one round trip less on a latency-bound launch
  const int lane = threadIdx.x & 7;
  bool      first = true;
  for (int i0 = blockIdx.x * (PMH_BLOCK / 8); i0 < nc; i0 += gridDim.x * (PMH_BLOCK / 8)) { // uniform trip count per workgroup
    const int i = i0 + (threadIdx.x >> 3);
    TV        s = (TV)0;
    const int k0 = (i < nc) ? rowptr[i] : 0, k1 = (i < nc) ? rowptr[i + 1] : 0;
    if (first) {
      if (hlt) return;
      first = false;
    }
    for (int k = k0 + lane; k < k1; k += 8) s += (TV)val[k] * t[col[k]];
    s += __shfl_down(s, 4, 8);
    s += __shfl_down(s, 2, 8);
    s += __shfl_down(s, 1, 8);
    if (i < nc && lane == 0) {
      bc[i] = s;
      if (dinv_c) d_c[i] = dinv_c[i] * s * itheta_c;
    }
  }
}

// coarse-grid correction x -= P x_c (<= 8 entries per row of P), one thread per fine row
template <typename TV>
__global__ __launch_bounds__(PMH_BLOCK) void k_mg_prolong_sub(int n, const int *__restrict__ halt, const int *__restrict__ rowptr, const int *__restrict__ col, const TV *__restrict__ val, const TV *__restrict__ xc, TV *__restrict__ x)
{
  const int hlt   = halt ? *halt : 0; // as in k_mg_restrict
  bool      first = true;
  for (int i = blockIdx.x * PMH_BLOCK + threadIdx.x; i < n; i += gridDim.x * PMH_BLOCK) {
    TV        s  = (TV)0;
    const int k0 = rowptr[i], k1 = rowptr[i + 1];
    const TV  xi = x[i];
    if (first) {
      if (hlt) return;
      first = false;
    }
    for (int k = k0; k < k1; k++) s += (TV)val[k] * xc[col[k]];
    x[i] = xi - s;
  }
  if (first && hlt) return;
}

// The same two transfers for prolongations with LONG rows (smoothed aggregation, mgsa.hip: ~ 30 entries per row of P, hundreds per row of P'): a wavefront per coarse
// row / 8 lanes per fine row, sums by butterfly steps in a fixed order.  Chosen at pmh_mg_create by the average row length of P.
template <typename TV>
__global__ __launch_bounds__(PMH_BLOCK) void k_mg_restrict_w(int nc, const int *__restrict__ halt, const int *__restrict__ rowptr, const int *__restrict__ col, const TV *__restrict__ val, const TV *__restrict__ t, TV *__restrict__ bc,
                                                           const TV *__restrict__ dinv_c, TV itheta_c, TV *__restrict__ d_c)
{
  if (halt && *halt) return;
  const int lane = threadIdx.x & 63;
  for (int i = blockIdx.x * (PMH_BLOCK / 64) + (threadIdx.x >> 6); i < nc; i += gridDim.x * (PMH_BLOCK / 64)) {
    TV s = (TV)0;
    for (int k = rowptr[i] + lane; k < rowptr[i + 1]; k += 64) s += (TV)val[k] * t[col[k]];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) {
      bc[i] = s;
      if (dinv_c) d_c[i] = dinv_c[i] * s * itheta_c;
    }
  }
}
template <typename TV>
__global__ __launch_bounds__(PMH_BLOCK) void k_mg_prolong_sub8(int n, const int *__restrict__ halt, const int *__restrict__ rowptr, const int *__restrict__ col, const TV *__restrict__ val, const TV *__restrict__ xc, TV *__restrict__ x)
{
  if (halt && *halt) return;
  const int lane = threadIdx.x & 7;
  for (int i0 = blockIdx.x * (PMH_BLOCK / 8); i0 < n; i0 += gridDim.x * (PMH_BLOCK / 8)) { // uniform trip count per workgroup
    const int i = i0 + (threadIdx.x >> 3);
    TV        s = (TV)0;
    if (i < n)
      for (int k = rowptr[i] + lane; k < rowptr[i + 1]; k += 8) s += (TV)val[k] * xc[col[k]];
    s += __shfl_xor(s, 4, 8);
    s += __shfl_xor(s, 2, 8);
    s += __shfl_xor(s, 1, 8);
    if (i < n && lane == 0) x[i] -= s;
  }
}

// P = P_node (x) I_3 (nodal prolongation of a 3-dof-per-node problem, e.g. trilinear interpolation of elasticity blocks): one
// node-level CSR entry serves the three components, so the transfer operators move a third of the index / value bytes.
// Detected at pmh_mg_create from the entries of P; any other P runs on the scalar kernels above.
template <typename TV>
__global__ __launch_bounds__(PMH_BLOCK) void k_mg_restrict3(int ncn, const int *__restrict__ halt, const int *__restrict__ rowptr, const int *__restrict__ col, const TV *__restrict__ val, const TV *__restrict__ t, TV *__restrict__ bc,
                                                          const TV *__restrict__ dinv_c, TV itheta_c, TV *__restrict__ d_c)
{
  const int hlt  = halt ? *halt : 0;
  const int lane = threadIdx.x & 7;
  bool      first = true;
  for (int i0 = blockIdx.x * (PMH_BLOCK / 8); i0 < ncn; i0 += gridDim.x * (PMH_BLOCK / 8)) { // uniform trip count per workgroup
    const int i  = i0 + (threadIdx.x >> 3);
    TV        s0 = (TV)0, s1 = (TV)0, s2 = (TV)0;
    const int k0 = (i < ncn) ? rowptr[i] : 0, k1 = (i < ncn) ? rowptr[i + 1] : 0;
    if (first) {
      if (hlt) return;
      first = false;
    }
    for (int k = k0 + lane; k < k1; k += 8) {
      const TV  w = val[k];
      const TV *p = t + 3 * (size_t)col[k];
      s0 += w * p[0], s1 += w * p[1], s2 += w * p[2];
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) s0 += __shfl_down(s0, o, 8), s1 += __shfl_down(s1, o, 8), s2 += __shfl_down(s2, o, 8);
    if (i < ncn && lane == 0) {
      TV *o = bc + 3 * (size_t)i;
      o[0] = s0, o[1] = s1, o[2] = s2;
      if (dinv_c) {
        const TV *di = dinv_c + 3 * (size_t)i;
        TV       *dd = d_c + 3 * (size_t)i;
        dd[0] = di[0] * s0 * itheta_c, dd[1] = di[1] * s1 * itheta_c, dd[2] = di[2] * s2 * itheta_c;
      }
    }
  }
}

template <typename TV>
__global__ __launch_bounds__(PMH_BLOCK) void k_mg_prolong_sub3(int nn, const int *__restrict__ halt, const int *__restrict__ rowptr, const int *__restrict__ col, const TV *__restrict__ val, const TV *__restrict__ xc, TV *__restrict__ x)
{
  const int hlt   = halt ? *halt : 0;
  bool      first = true;
  for (int i = blockIdx.x * PMH_BLOCK + threadIdx.x; i < nn; i += gridDim.x * PMH_BLOCK) {
    const int k0 = rowptr[i], k1 = rowptr[i + 1];
    TV       *xi = x + 3 * (size_t)i;
    const TV  a0 = xi[0], a1 = xi[1], a2 = xi[2];
    if (first) {
      if (hlt) return;
      first = false;
    }
    TV s0 = (TV)0, s1 = (TV)0, s2 = (TV)0;
    for (int k = k0; k < k1; k++) {
      const TV  w = val[k];
      const TV *p = xc + 3 * (size_t)col[k];
      s0 += w * p[0], s1 += w * p[1], s2 += w * p[2];
    }
    xi[0] = a0 - s0, xi[1] = a1 - s1, xi[2] = a2 - s2;
  }
  if (first && hlt) return;
}

// coarsest level: x_b = pinv_b b_b, one wavefront per row, lanes stride the row of the dense block (fixed order).
// TP = storage type of the pseudo-inverse: TV, or _Float16 (PMH_MG_FP16: entries / scale, fp32 arithmetic) -- with one or two
// blocks per GPU the hierarchy stops at a ~5000-dof level whose dense solve is a pure HBM stream (2 B per entry).
template <typename TV, typename TP>
__global__ __launch_bounds__(PMH_BLOCK) void k_mg_coarse(int nb, int n, const int *__restrict__ halt, const int *__restrict__ rs, const long long *__restrict__ ofs, const TP *__restrict__ pinv, TV scale, const TV *__restrict__ b, TV *__restrict__ x)
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
  const int r0 = rs[lo], m = rs[lo + 1] - r0;
  const TP *a  = pinv + ofs[lo] + (size_t)(row - r0) * m;
  const TV *bb = b + r0;
  TV        s0 = (TV)0, s1 = (TV)0, s2 = (TV)0, s3 = (TV)0;
  int       j  = lane;
  for (; j + 192 < m; j += 256) { // four independent streams per lane
    s0 += (TV)__builtin_nontemporal_load(&a[j]) * bb[j];
    s1 += (TV)__builtin_nontemporal_load(&a[j + 64]) * bb[j + 64];
    s2 += (TV)__builtin_nontemporal_load(&a[j + 128]) * bb[j + 128];
    s3 += (TV)__builtin_nontemporal_load(&a[j + 192]) * bb[j + 192];
  }
  for (; j < m; j += 64) s0 += (TV)a[j] * bb[j];
  TV s = (s0 + s1) + (s2 + s3);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if (lane == 0) x[row] = (sizeof(TP) == 2) ? s * scale : s;
}

template <typename TA, typename TB>
__global__ __launch_bounds__(PMH_BLOCK) void k_mg_convert(int n, const int *__restrict__ halt, const TA *__restrict__ a, TB *__restrict__ b)
{
  if (halt && *halt) return;
  for (int i = blockIdx.x * PMH_BLOCK + threadIdx.x; i < n; i += gridDim.x * PMH_BLOCK) b[i] = (TB)a[i];
}

static inline dim3 mg_grid(int n)
{
  long long g = ((long long)n + PMH_BLOCK - 1) / PMH_BLOCK;
  return dim3((unsigned)(g < 1 ? 1 : (g > PMH_MAX_VEC_BLOCKS ? PMH_MAX_VEC_BLOCKS : g)));
}

// y = A_l x (NONE) or A_l x - y1 (SUB)
static int mg_spmv(pmh_mg mg, int l, const void *x, void *y, int epi_kind, const void *y1)
{
  mg_level &Lv = mg->L[l];
  if (l == 0) mg->fine_spmv++;
  if (Lv.Ab) {
    if (mg->is_float) return pmh_bsr3_spmv_f32(Lv.Ab, (const float *)x, (float *)y, epi_kind, (const float *)y1, mg->halt);
    return pmh_bsr3_spmv_f64(Lv.Ab, (const double *)x, (double *)y, epi_kind, (const double *)y1, mg->halt);
  }
  pmh_spmv_epi epi;
  memset(&epi, 0, sizeof(epi));
  epi.kind = epi_kind;
  epi.y1   = (const double *)y1;
  epi.halt = mg->halt;
  return pmh_csr_spmv_launch(Lv.A, (const double *)x, (double *)y, epi);
}

// degree-k Chebyshev/Jacobi smoothing of A x = b on level l; zero: x is taken as 0 on entry
template <typename TV>
static int mg_smooth(pmh_mg mg, int l, const TV *b, TV *x, bool zero)
{
  mg_level   &Lv = mg->L[l];
  hipStream_t st = mg->ctx->stream;
  const dim3  g  = mg_grid(Lv.n), blk(PMH_BLOCK);
  TV         *r = (TV *)Lv.r, *d = (TV *)Lv.d, *t = (TV *)Lv.t;
  const TV   *dinv = (const TV *)Lv.dinv;
  if (zero) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cheb_first_zero<TV>), g, blk, 0, st, Lv.n, mg->halt, dinv, b, (TV)(1.0 / Lv.theta), r, d, x);
  } else {
    PMH_CHK(mg_spmv(mg, l, x, t, PMH_EPI_NONE, nullptr));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cheb_first<TV>), g, blk, 0, st, Lv.n, mg->halt, dinv, b, (const TV *)t, (TV)(1.0 / Lv.theta), r, d, x);
  }
  for (int j = 1; j < mg->degree; j++) {
    PMH_CHK(mg_spmv(mg, l, d, t, PMH_EPI_NONE, nullptr));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cheb_step<TV>), g, blk, 0, st, Lv.n, mg->halt, dinv, (const TV *)t, (TV)Lv.c1[j], (TV)Lv.c2[j], r, d, x);
  }
  PMH_HIP(hipGetLastError());
  return PMH_SUCCESS;
}

template <typename TV> static int bsr_epi_launch(pmh_bsr3 B, const TV *x, TV *y, int epi, const pmh_bsr3_epi<TV> &e, const int *halt);
template <> int bsr_epi_launch<double>(pmh_bsr3 B, const double *x, double *y, int epi, const pmh_bsr3_epi<double> &e, const int *halt) { return pmh_bsr3_spmv_epi_f64(B, x, y, epi, e, halt); }
template <> int bsr_epi_launch<float>(pmh_bsr3 B, const float *x, float *y, int epi, const pmh_bsr3_epi<float> &e, const int *halt) { return pmh_bsr3_spmv_epi_f32(B, x, y, epi, e, halt); }

template <typename TV> static int mg_cycle(pmh_mg mg, int l, const TV *b, TV *x, const double *b64 = nullptr, double *z64 = nullptr, bool d0_ready = false);

// does level l run mg_level_fused (its first smoothing direction can then be produced by the kernel that makes its b)?
static inline bool mg_fused_level(pmh_mg mg, int l) { return l < mg->nlevels - 1 && mg->fused && mg->L[l].Ab; }

// b_c = P' t (+ the coarse level's d0) and x -= P x_c: node-level kernels when P = P_node (x) I_3, scalar CSR kernels otherwise
template <typename TV>
static void mg_restrict(pmh_mg mg, int l, const TV *t, bool with_d0)
{
  mg_level   &Lv = mg->L[l], &Lc = mg->L[l + 1];
  hipStream_t st = mg->ctx->stream;
  const TV   *dv = with_d0 ? (const TV *)Lc.dinv : (const TV *)nullptr;
  const TV    it = with_d0 ? (TV)(1.0 / Lc.theta) : (TV)0;
  TV         *dc = with_d0 ? (TV *)Lc.d : (TV *)nullptr;
  if (Lv.rn_rowptr) {
    const int ncn = Lc.n / 3;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mg_restrict3<TV>), mg_grid(8 * ncn), dim3(PMH_BLOCK), 0, st, ncn, mg->halt, (const int *)Lv.rn_rowptr, (const int *)Lv.rn_col, (const TV *)Lv.rn_val, t, (TV *)Lc.b, dv, it, dc);
  } else if (Lv.long_rows) {
    pmh_csr R = Lv.P->transpose;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mg_restrict_w<TV>), mg_grid(64LL * Lc.n > 0x7fffffff ? 0x7fffffff : 64 * Lc.n), dim3(PMH_BLOCK), 0, st, Lc.n, mg->halt, (const int *)R->d_rowptr, (const int *)R->d_col, (const TV *)Lv.rv, t,
                       (TV *)Lc.b, dv, it, dc);
  } else {
    pmh_csr R = Lv.P->transpose;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mg_restrict<TV>), mg_grid(8 * (long long)Lc.n > 0x7fffffff ? 0x7fffffff : 8 * Lc.n), dim3(PMH_BLOCK), 0, st, Lc.n, mg->halt, (const int *)R->d_rowptr, (const int *)R->d_col, (const TV *)Lv.rv, t,
                       (TV *)Lc.b, dv, it, dc);
  }
}

template <typename TV>
static void mg_prolong_sub(pmh_mg mg, int l, TV *x)
{
  mg_level   &Lv = mg->L[l], &Lc = mg->L[l + 1];
  hipStream_t st = mg->ctx->stream;
  if (Lv.pn_rowptr)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mg_prolong_sub3<TV>), mg_grid(Lv.n / 3), dim3(PMH_BLOCK), 0, st, Lv.n / 3, mg->halt, (const int *)Lv.pn_rowptr, (const int *)Lv.pn_col, (const TV *)Lv.pn_val, (const TV *)Lc.x, x);
  else if (Lv.long_rows)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mg_prolong_sub8<TV>), mg_grid(8LL * Lv.n > 0x7fffffff ? 0x7fffffff : 8 * Lv.n), dim3(PMH_BLOCK), 0, st, Lv.n, mg->halt, (const int *)Lv.P->d_rowptr, (const int *)Lv.P->d_col, (const TV *)Lv.pv, (const TV *)Lc.x, x);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mg_prolong_sub<TV>), mg_grid(Lv.n), dim3(PMH_BLOCK), 0, st, Lv.n, mg->halt, (const int *)Lv.P->d_rowptr, (const int *)Lv.P->d_col, (const TV *)Lv.pv, (const TV *)Lc.x, x);
}

// One smoothed level with the degree-2 Chebyshev steps finished inside the operator kernel (7 launches instead of 10):
//   d0 = D^-1 b/theta | xa = (1+c1) d0 + c2 D^-1 (b - A d0) | t = A xa - b | b_c = P't | ... | xa -= P x_c |
//   r, d, x = xa + d from A xa | x += c1 d + c2 (r - D^-1 A d)
// b64 / z64: fp64 input / output of the fp32 cycle's fine level (the conversions ride on the first and last kernel).
template <typename TV>
static int mg_level_fused(pmh_mg mg, int l, const TV *b, TV *x, const double *b64, double *z64, bool d0_ready)
{
  mg_level   &Lv = mg->L[l], &Lc = mg->L[l + 1];
  hipStream_t st = mg->ctx->stream;
  const dim3  g = mg_grid(Lv.n), blk(PMH_BLOCK);
  TV         *r = (TV *)Lv.r, *d = (TV *)Lv.d, *t = (TV *)Lv.t, *xa = (TV *)Lv.xa;
  const TV   *dinv = (const TV *)Lv.dinv;
  const TV    itheta = (TV)(1.0 / Lv.theta), c1 = (TV)Lv.c1[1], c2 = (TV)Lv.c2[1];
  if (d0_ready) { // d (and the cycle-precision copy of b) were written by the producer of b
  } else if (b64) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cheb_d0<TV, double>), g, blk, 0, st, Lv.n, mg->halt, dinv, b64, itheta, d, (TV *)b);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cheb_d0<TV, TV>), g, blk, 0, st, Lv.n, mg->halt, dinv, b, itheta, d, (TV *)nullptr);
  pmh_bsr3_epi<TV> e;
  memset(&e, 0, sizeof(e));
  e.y1 = b, e.dinv = dinv, e.r = r, e.d = d;
  e.c0 = (TV)1 + c1, e.c1 = c1, e.c2 = c2;
  if (l == 0) mg->fine_spmv += 4;
  PMH_CHK(bsr_epi_launch<TV>(Lv.Ab, d, xa, PMH_BSR_EPI_PRE, e, mg->halt));
  PMH_CHK(bsr_epi_launch<TV>(Lv.Ab, xa, t, PMH_EPI_SUB, e, mg->halt));
  const bool cf = mg_fused_level(mg, l + 1); // the coarse level's d0 rides on the restriction
  mg_restrict<TV>(mg, l, (const TV *)t, cf);
  PMH_CHK(mg_cycle<TV>(mg, l + 1, (const TV *)Lc.b, (TV *)Lc.x, nullptr, nullptr, cf));
  mg_prolong_sub<TV>(mg, l, xa);
  PMH_HIP(hipGetLastError());
  e.c0 = itheta;
  PMH_CHK(bsr_epi_launch<TV>(Lv.Ab, xa, x, PMH_BSR_EPI_POST1, e, mg->halt));
  e.z64 = z64;
  return bsr_epi_launch<TV>(Lv.Ab, d, x, PMH_BSR_EPI_POST2, e, mg->halt);
}

template <typename TV>
static int mg_cycle(pmh_mg mg, int l, const TV *b, TV *x, const double *b64, double *z64, bool d0_ready)
{
  mg_level   &Lv = mg->L[l];
  hipStream_t st = mg->ctx->stream;
  const dim3  blk(PMH_BLOCK);
  if (mg_fused_level(mg, l)) return mg_level_fused<TV>(mg, l, b, x, b64, z64, d0_ready);
  if (l == mg->nlevels - 1) {
    if (mg->cp_half) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mg_coarse<TV, _Float16>), dim3((Lv.n + 3) / 4), blk, 0, st, mg->nb_coarse, Lv.n, mg->halt, (const int *)mg->d_crs, (const long long *)mg->d_cofs, (const _Float16 *)mg->d_cpinv, (TV)mg->cp_scale, b, x);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mg_coarse<TV, TV>), dim3((Lv.n + 3) / 4), blk, 0, st, mg->nb_coarse, Lv.n, mg->halt, (const int *)mg->d_crs, (const long long *)mg->d_cofs, (const TV *)mg->d_cpinv, (TV)1, b, x);
    PMH_HIP(hipGetLastError());
    return PMH_SUCCESS;
  }
  mg_level &Lc = mg->L[l + 1];
  PMH_CHK(mg_smooth<TV>(mg, l, b, x, true));
  // t = A x - b; b_{l+1} = P' t = -P'(b - A x); the coarse solve is linear, so the sign is undone by subtracting P x_{l+1}
  PMH_CHK(mg_spmv(mg, l, x, Lv.t, PMH_EPI_SUB, b));
  mg_restrict<TV>(mg, l, (const TV *)Lv.t, false);
  PMH_CHK(mg_cycle<TV>(mg, l + 1, (const TV *)Lc.b, (TV *)Lc.x));
  mg_prolong_sub<TV>(mg, l, x);
  PMH_HIP(hipGetLastError());
  return mg_smooth<TV>(mg, l, b, x, false);
}

static int mg_apply_body(pmh_mg mg, const double *b, double *x, const int *halt, bool d0_ready = false);

// The caller's kernel that produces the residual b can write the fine level's first smoothing direction itself
// (d0 = D^-1 b / theta in fp32, plus the fp32 copy of b): slots and constants for it; returns 0 if the cycle has no such slot.
int pmh_mg_fine_d0_slots(pmh_mg mg, const float **dinv, float *itheta, float **d0, float **b32)
{
  if (!(mg->is_float && mg->fused && mg->nlevels > 1 && mg_fused_level(mg, 0))) return 0;
  mg_level &L0 = mg->L[0];
  *dinv = (const float *)L0.dinv, *itheta = (float)(1.0 / L0.theta), *d0 = (float *)L0.d, *b32 = (float *)L0.b;
  return 1;
}

// (A hipGraph replay of the cycle -- captured once per (b, x, halt) triple -- was opt-in until round 6: slower or equal in every configuration measured, and rocprofv3's
// kernel tracing crashes on graph launches on this ROCm; docs/LAB_NOTEBOOK.md.)
int pmh_mg_apply_halt(pmh_mg mg, const double *b, double *x, const int *halt, bool d0_ready) { return mg_apply_body(mg, b, x, halt, d0_ready); }

static int mg_apply_body(pmh_mg mg, const double *b, double *x, const int *halt, bool d0_ready)
{
  mg->halt = halt;
  int rc;
  if (mg->is_float && mg->fused && mg->nlevels > 1) {
    mg_level &L0 = mg->L[0];
    rc = mg_cycle<float>(mg, 0, (const float *)L0.b, (float *)L0.x, b, x, d0_ready); // conversions fused into the first / last kernel
  } else if (mg->is_float) {
    mg_level   &L0 = mg->L[0];
    hipStream_t st = mg->ctx->stream;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mg_convert<double, float>), mg_grid(L0.n), dim3(PMH_BLOCK), 0, st, L0.n, halt, b, (float *)L0.b);
    rc = mg_cycle<float>(mg, 0, (const float *)L0.b, (float *)L0.x);
    if (!rc) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mg_convert<float, double>), mg_grid(L0.n), dim3(PMH_BLOCK), 0, st, L0.n, halt, (const float *)L0.x, x);
  } else {
    rc = mg_cycle<double>(mg, 0, b, x);
  }
  mg->halt = nullptr;
  return rc;
}

extern "C" int pmh_mg_apply(pmh_mg mg, const double *b, double *x)
{
  PMH_ARG(mg && b && x && b != x);
  return pmh_mg_apply_halt(mg, b, x, nullptr);
}

// P = P_node (x) I_3 ?  Rows 3i+c of P must hold the columns 3j+c with the same value for c = 0,1,2.  If so, node-level CSR
// copies of P and of P' (values in the cycle's precision) are built for the three-components-per-entry kernels.
static int mg_build_nodal_transfer(pmh_mg mg, int l, int fl)
{
  mg_level &Lv = mg->L[l];
  pmh_csr   P  = Lv.P;
  pmh_ctx   ctx = mg->ctx;
  if (P->nrows % 3 || P->ncols % 3 || P->nrows == 0 || P->nnz % 3) return PMH_SUCCESS;
  const int           n = P->nrows, nn = n / 3, ncn = P->ncols / 3;
  std::vector<int>    rp_own, ci_own;
  std::vector<double> va_own;
  const int          *rp = P->h_rowptr, *ci = P->h_col; // the builder's host copy where it lent one (pmh_mg_create_box): no download of what was just uploaded
  const double       *va = P->h_val;
  if (!(rp && ci && va)) {
    rp_own.resize((size_t)n + 1), ci_own.resize((size_t)P->nnz), va_own.resize((size_t)P->nnz);
    PMH_CHK(pmh_memcpy_d2h(ctx, rp_own.data(), P->d_rowptr, sizeof(int) * rp_own.size()));
    PMH_CHK(pmh_memcpy_d2h(ctx, ci_own.data(), P->d_col, sizeof(int) * ci_own.size()));
    PMH_CHK(pmh_memcpy_d2h(ctx, va_own.data(), P->d_val, sizeof(double) * va_own.size()));
    rp = rp_own.data(), ci = ci_own.data(), va = va_own.data();
  }
  // node rows: counts, prefix, then the entries -- host threads over ranges of nodes (one thread took 0.05 s for the fine level of configs[2])
  std::vector<int> nrp((size_t)nn + 1, 0);
  for (int i = 0; i < nn; i++) nrp[i + 1] = nrp[i] + (rp[3 * i + 1] - rp[3 * i]);
  std::vector<int>    nci((size_t)nrp[nn]);
  std::vector<double> nva((size_t)nrp[nn]);
  {
    const int         nt = std::max(1, std::min(pmh_host_threads(), nn / 4096 + 1));
    std::vector<char> bad(nt, 0);
    auto work = [&](int t) {
      for (int i = (int)((long long)nn * t / nt); i < (int)((long long)nn * (t + 1) / nt); i++) {
        const int k0 = rp[3 * i], m = rp[3 * i + 1] - k0;
        if (rp[3 * i + 2] - rp[3 * i + 1] != m || rp[3 * i + 3] - rp[3 * i + 2] != m) { bad[t] = 1; return; }
        for (int q = 0; q < m; q++) {
          const int    j = ci[k0 + q];
          const double w = va[k0 + q];
          if (j % 3 != 0) { bad[t] = 1; return; }
          for (int c = 1; c < 3; c++)
            if (ci[rp[3 * i + c] + q] != j + c || va[rp[3 * i + c] + q] != w) { bad[t] = 1; return; }
          nci[(size_t)nrp[i] + q] = j / 3, nva[(size_t)nrp[i] + q] = w;
        }
      }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++) th.emplace_back(work, t);
    for (auto &x : th) x.join();
    for (char b : bad)
      if (b) return PMH_SUCCESS;
  }
  // node-level transpose (counting sort: columns of every row of P' ascending, the order of the scalar P')
  std::vector<int>    trp((size_t)ncn + 1, 0), tci(nci.size());
  std::vector<double> tva(nva.size());
  for (int c : nci) trp[c + 1]++;
  for (int j = 0; j < ncn; j++) trp[j + 1] += trp[j];
  {
    std::vector<int> pos(trp.begin(), trp.end() - 1);
    for (int i = 0; i < nn; i++)
      for (int k = nrp[i]; k < nrp[i + 1]; k++) {
        const int p = pos[nci[k]]++;
        tci[p] = i, tva[p] = nva[k];
      }
  }
  auto upload = [&](const std::vector<int> &r, const std::vector<int> &c, const std::vector<double> &v, int **dr, int **dc, void **dv) -> int {
    PMH_CHK(pmh_malloc(ctx, sizeof(int) * r.size(), (void **)dr));
    PMH_CHK(pmh_malloc(ctx, sizeof(int) * (c.size() ? c.size() : 1), (void **)dc));
    PMH_CHK(pmh_memcpy_h2d(ctx, *dr, r.data(), sizeof(int) * r.size()));
    if (!c.empty()) PMH_CHK(pmh_memcpy_h2d(ctx, *dc, c.data(), sizeof(int) * c.size()));
    if (fl) {
      std::vector<float> vf(v.begin(), v.end());
      PMH_CHK(pmh_malloc(ctx, sizeof(float) * (vf.size() ? vf.size() : 1), dv));
      if (!vf.empty()) PMH_CHK(pmh_memcpy_h2d(ctx, *dv, vf.data(), sizeof(float) * vf.size()));
    } else {
      PMH_CHK(pmh_malloc(ctx, sizeof(double) * (v.size() ? v.size() : 1), dv));
      if (!v.empty()) PMH_CHK(pmh_memcpy_h2d(ctx, *dv, v.data(), sizeof(double) * v.size()));
    }
    return PMH_SUCCESS;
  };
  PMH_CHK(upload(nrp, nci, nva, &Lv.pn_rowptr, &Lv.pn_col, &Lv.pn_val));
  PMH_CHK(upload(trp, tci, tva, &Lv.rn_rowptr, &Lv.rn_col, &Lv.rn_val));
  return PMH_SUCCESS;
}

extern "C" int pmh_mg_create(pmh_ctx ctx, int nlevels, const pmh_csr *A, const pmh_csr *P, int degree, const double *lambda_max, double lo_frac, double hi_frac, int nb_coarse, const int *coarse_rowstart,
                             const double *coarse_pinv_host, int precision, pmh_mg *out)
{
  PMH_ARG(ctx && out && A && nlevels >= 1 && degree >= 1 && nb_coarse >= 1 && coarse_rowstart && coarse_pinv_host);
  PMH_ARG(nlevels == 1 || (P && lambda_max));
  PMH_ARG(hi_frac > lo_frac && lo_frac > 0.0);
  PMH_ARG(precision == PMH_MG_FP64 || precision == PMH_MG_FP32 || precision == PMH_MG_FP16);
  for (int l = 0; l < nlevels; l++) {
    PMH_ARG(A[l] && A[l]->nrows == A[l]->ncols);
    if (l + 1 < nlevels) PMH_ARG(P[l] && P[l]->nrows == A[l]->nrows && P[l]->ncols == A[l + 1]->nrows && lambda_max[l] > 0.0);
  }
  PMH_ARG(coarse_rowstart[0] == 0 && coarse_rowstart[nb_coarse] == A[nlevels - 1]->nrows);
  const int    fl = precision != PMH_MG_FP64;
  const size_t w  = fl ? sizeof(float) : sizeof(double);
  pmh_mg mg     = new pmh_mg_s();
  mg->ctx       = ctx;
  mg->nlevels   = nlevels;
  mg->degree    = degree;
  mg->is_float  = fl;
  mg->halt      = nullptr;
  mg->fine_spmv = 0;
  mg->timing_on = 0;
  mg->fused = (degree == 2);
  if (const char *e = getenv("PMH_MG_FUSED")) mg->fused = mg->fused && atoi(e); // testing knob: 0 = separate smoothing kernels
  mg->L.resize(nlevels);
  const bool no_bsr = getenv("PMH_MG_NO_BSR") != nullptr; // testing knob: keep the CSR kernels (fp64 only)
  const bool verbose = getenv("PMH_CONTACT_TIMING") != nullptr;
  auto       tnow    = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double     tlast   = tnow();
  auto       stage   = [&](const char *what, int l) {
    if (!verbose) return;
    (void)pmh_sync(ctx);
    const double t = tnow();
    fprintf(stderr, "    pmh_mg_create: level %d %-44s %.3f s\n", l, what, t - tlast);
    tlast = t;
  };
  for (int l = 0; l < nlevels; l++) {
    mg_level &Lv = mg->L[l];
    Lv.A = A[l], Lv.P = (l + 1 < nlevels) ? P[l] : nullptr, Lv.n = A[l]->nrows, Lv.Ab = nullptr;
    Lv.dinv = Lv.x = Lv.b = Lv.r = Lv.d = Lv.t = Lv.xa = nullptr;
    Lv.pv = Lv.rv = nullptr, Lv.pv_owned = false;
    Lv.pn_rowptr = Lv.pn_col = Lv.rn_rowptr = Lv.rn_col = nullptr, Lv.pn_val = Lv.rn_val = nullptr;
    const size_t nbytes = w * (size_t)(Lv.n ? Lv.n : 1);
    if (l > 0 || fl) {
      PMH_CHK(pmh_malloc(ctx, nbytes, &Lv.x));
      PMH_CHK(pmh_malloc(ctx, nbytes, &Lv.b));
    }
    if (l + 1 < nlevels) {
      // FP16: the fine-level operator (almost all of the cycle's bytes) stores fp16 entries; arithmetic and vectors stay fp32
      const int f16_levels = 2; // fp16 entries on the two finest levels (measured: 22.2 -> 20.6 ms per step, same CG count; a third level gains nothing)
      const int storage = !fl ? PMH_BSR_F64 : ((precision == PMH_MG_FP16 && l < f16_levels) ? PMH_BSR_F16 : PMH_BSR_F32);
      // coarser levels use 512-block tiles: twice the workgroups on operators that are too small to fill the chip, and a kernel
      // instantiation of their own, so that profiler averages of the fine-level operator are not mixed with the coarse launches
      const int coarse_tile = 512;
      // congruent blocks (every cube of a structured decomposition): nb_coarse is the number of diagonal blocks on every level; pmh_bsr3_from_csr keeps ONE device copy of the
      // block when all of them turn out to be bit-identical (it compares them), applied to the nb_coarse vector segments
      if (!no_bsr || fl) PMH_CHK(pmh_bsr3_from_csr(A[l], storage, &Lv.Ab, l == 0 ? 0 : coarse_tile, (nb_coarse > 1 && Lv.n % nb_coarse == 0) ? nb_coarse : 1));
      stage("3x3-block operator (pmh_bsr3_from_csr)", l);
      if (fl && !Lv.Ab) {
        pmh_mg_destroy(mg);
        return pmh_set_error(PMH_ERR_SUP, "pmh_mg_create: PMH_MG_FP32/FP16 needs 3x3-block operators on every smoothed level (level %d of size %d is not)", l, Lv.n);
      }
      Lv.long_rows = P[l]->nrows > 0 && (double)P[l]->nnz / P[l]->nrows > 12.0;
      PMH_CHK(pmh_csr_ensure_transpose(P[l]));
      stage("transpose of the prolongation", l);
      Lv.pv = P[l]->d_val, Lv.rv = P[l]->transpose->d_val;
      if (fl && P[l]->nnz > 0) {
        const int nz = (int)P[l]->nnz;
        PMH_CHK(pmh_malloc(ctx, sizeof(float) * (size_t)nz, &Lv.pv));
        PMH_CHK(pmh_malloc(ctx, sizeof(float) * (size_t)nz, &Lv.rv));
        Lv.pv_owned = true;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mg_convert<double, float>), mg_grid(nz), dim3(PMH_BLOCK), 0, ctx->stream, nz, (const int *)nullptr, (const double *)P[l]->d_val, (float *)Lv.pv);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mg_convert<double, float>), mg_grid(nz), dim3(PMH_BLOCK), 0, ctx->stream, nz, (const int *)nullptr, (const double *)P[l]->transpose->d_val, (float *)Lv.rv);
        PMH_HIP(hipGetLastError());
      }
      PMH_CHK(pmh_malloc(ctx, nbytes, &Lv.dinv));
      PMH_CHK(pmh_malloc(ctx, nbytes, &Lv.r));
      PMH_CHK(pmh_malloc(ctx, nbytes, &Lv.d));
      PMH_CHK(pmh_malloc(ctx, nbytes, &Lv.t));
      PMH_CHK(pmh_malloc(ctx, nbytes, &Lv.xa));
      if (Lv.n > 0) {
        if (fl) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mg_dinv<float>), mg_grid((int)std::min<long long>(8LL * Lv.n, 0x7fffff00LL)), dim3(PMH_BLOCK), 0, ctx->stream, Lv.n, (const int *)A[l]->d_rowptr, (const int *)A[l]->d_col, (const double *)A[l]->d_val, (float *)Lv.dinv);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mg_dinv<double>), mg_grid((int)std::min<long long>(8LL * Lv.n, 0x7fffff00LL)), dim3(PMH_BLOCK), 0, ctx->stream, Lv.n, (const int *)A[l]->d_rowptr, (const int *)A[l]->d_col, (const double *)A[l]->d_val, (double *)Lv.dinv);
        PMH_HIP(hipGetLastError());
      }
      stage("value conversion, diagonal, work vectors", l);
      PMH_CHK(mg_build_nodal_transfer(mg, l, fl));
      stage("node-wise transfer operators", l);
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
  mg->nb_coarse  = nb_coarse;
  mg->coarse_m16 = 1;
  std::vector<long long> ofs(nb_coarse);
  long long              tot = 0;
  for (int b = 0; b < nb_coarse; b++) {
    const long long m = coarse_rowstart[b + 1] - coarse_rowstart[b];
    PMH_ARG(m >= 0);
    if (m % 16 || m < 512) mg->coarse_m16 = 0;
    ofs[b] = tot, tot += m * m;
  }
  PMH_CHK(pmh_malloc(ctx, sizeof(int) * (size_t)(nb_coarse + 1), (void **)&mg->d_crs));
  PMH_CHK(pmh_malloc(ctx, sizeof(long long) * (size_t)nb_coarse, (void **)&mg->d_cofs));
  mg->cp_half  = (precision == PMH_MG_FP16 && nlevels > 1) ? 1 : 0;
  mg->cp_scale = 1.0;
  PMH_CHK(pmh_malloc(ctx, (mg->cp_half ? 2 : w) * (size_t)(tot ? tot : 1), &mg->d_cpinv));
  PMH_CHK(pmh_memcpy_h2d(ctx, mg->d_crs, coarse_rowstart, sizeof(int) * (size_t)(nb_coarse + 1)));
  PMH_CHK(pmh_memcpy_h2d(ctx, mg->d_cofs, ofs.data(), sizeof(long long) * (size_t)nb_coarse));
  if (mg->cp_half) {
    // power-of-two scale that brings the largest entry to [1, 2) (as the fp16 fine-level operator, bsr.hip)
    double amax = 0.0;
    for (long long i = 0; i < tot; i++) amax = std::max(amax, fabs(coarse_pinv_host[i]));
    int ex = 0;
    if (amax > 0.0) frexp(amax, &ex);
    mg->cp_scale = ldexp(1.0, ex - 1);
    std::vector<_Float16> ph((size_t)tot);
    for (long long i = 0; i < tot; i++) ph[i] = (_Float16)(float)(coarse_pinv_host[i] / mg->cp_scale);
    PMH_CHK(pmh_memcpy_h2d(ctx, mg->d_cpinv, ph.data(), sizeof(_Float16) * (size_t)tot));
  } else if (fl) {
    std::vector<float> pf((size_t)tot);
    for (long long i = 0; i < tot; i++) pf[i] = (float)coarse_pinv_host[i];
    PMH_CHK(pmh_memcpy_h2d(ctx, mg->d_cpinv, pf.data(), sizeof(float) * (size_t)tot));
  } else {
    PMH_CHK(pmh_memcpy_h2d(ctx, mg->d_cpinv, coarse_pinv_host, sizeof(double) * (size_t)tot));
  }
  *out = mg;
  return PMH_SUCCESS;
}

int pmh_mg_adopt_csr(pmh_mg mg, pmh_csr A)
{
  PMH_ARG(mg && A);
  mg->owned.push_back(A);
  return PMH_SUCCESS;
}

extern "C" int pmh_mg_destroy(pmh_mg mg)
{
  if (!mg) return PMH_SUCCESS;
  pmh_ctx ctx = mg->ctx;
  for (pmh_csr a : mg->owned) pmh_csr_destroy(a);
  for (auto &Lv : mg->L) {
    pmh_free(ctx, Lv.dinv);
    pmh_free(ctx, Lv.r);
    pmh_free(ctx, Lv.d);
    pmh_free(ctx, Lv.t);
    pmh_free(ctx, Lv.xa);
    pmh_free(ctx, Lv.x);
    pmh_free(ctx, Lv.b);
    if (Lv.pv_owned) pmh_free(ctx, Lv.pv), pmh_free(ctx, Lv.rv);
    pmh_free(ctx, Lv.pn_rowptr), pmh_free(ctx, Lv.pn_col), pmh_free(ctx, Lv.pn_val), pmh_free(ctx, Lv.rn_rowptr), pmh_free(ctx, Lv.rn_col), pmh_free(ctx, Lv.rn_val);
    pmh_bsr3_destroy(Lv.Ab);
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

// HIP-event timing of the fine-level operator launches of the cycle (block kernel only; the CSR kernel has its own)
extern "C" int pmh_mg_timing_enable(pmh_mg mg, int max_launches)
{
  PMH_ARG(mg);
  if (!mg->L[0].Ab) return max_launches ? pmh_set_error(PMH_ERR_SUP, "pmh_mg_timing_enable: the fine level runs on the CSR kernel; use pmh_csr_timing_enable") : PMH_SUCCESS;
  mg->timing_on = max_launches > 0;
  return pmh_bsr3_timing_enable(mg->L[0].Ab, max_launches);
}

extern "C" int pmh_mg_timing_get(pmh_mg mg, int *launches, double *total_ms, double *bytes_per_launch)
{
  PMH_ARG(mg && mg->L[0].Ab);
  // algorithmic bytes per launch = the operator product (matrix, column indices, x gather, y) + the operands of the smoothing
  // step fused into it, averaged over the timed launches
  double ex = 0.0;
  PMH_CHK(pmh_bsr3_timing_get(mg->L[0].Ab, launches, total_ms, &ex));
  if (bytes_per_launch) *bytes_per_launch = pmh_bsr3_bytes(mg->L[0].Ab) + (*launches ? ex / *launches : 0.0);
  return PMH_SUCCESS;
}
