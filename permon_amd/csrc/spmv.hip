// CSR SpMV for gfx950 (fp64 values, int32 indices) -- the MatMult_SeqAIJ role of the PERMON QPS path.
//
// Two kernels, chosen per matrix at pmh_csr_create:
//  * STREAM (short rows, e.g. the 5-point Laplacian of BASELINE configs[1]): row-blocked.  A workgroup
//    owns a contiguous block of rows whose non-zeros fit PMH_NNZ_PER_BLOCK; it streams val/col with
//    fully coalesced loads, stages the products val*x[col] in LDS, then one lane per row sums that row's
//    products from LDS LEFT TO RIGHT -- the same order as PETSc's MatMult_SeqAIJ row loop, so y is
//    bit-identical to the CPU path.  Rows longer than a block are reduced by the whole workgroup.
//  * VECTOR (long rows, e.g. 81 nnz/row Q1 elasticity blocks K_i): LPR lanes of a 64-wide wavefront
//    per row, shuffle-tree reduction.
// Both map workgroups to rows XCD-aware: the dispatcher deals workgroups round-robin over the 8 XCDs, so
// workgroup b is given the row block (b%8)*chunk + b/8; each XCD then walks a contiguous slab of rows and
// the gathered x entries of neighbouring row blocks hit that XCD's own L2 (placement only affects speed).
// Epilogues fuse the row-local follow-up work of the MPGP iteration into the SpMV (SURVEY 8d phase P1).
//
// Algorithmic bytes per SpMV: 12*nnz + 20*nrows (8 B value + 4 B column per non-zero; 4 B row pointer,
// 8 B y write, 8 B compulsory x read per row).
#include <algorithm>
#include <climits>

#include <atomic>

#include "pmh_internal.h"
#include "reduce.h"

#define PMH_MAX_ROWS_PER_BLOCK 1024 // bounds the per-row phase when rows are (nearly) empty, e.g. B' of MATGLUING

struct EpiArgs {
  const double *y1;
  const double *g, *xx, *lb, *ub;
  const int    *halt;
};

// streamed-once operands (y, y1, g, x, lb, ub) optionally bypass the caches with non-temporal accesses
template <bool NT>
__device__ __forceinline__ double ldg(const double *p)
{
  return NT ? __builtin_nontemporal_load(p) : *p;
}
template <bool NT>
__device__ __forceinline__ void stg(double *p, double v)
{
  if (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

template <int EPI, bool NT = false>
__device__ __forceinline__ void epi_row(int r, double sum, const double *__restrict__ xin, double *__restrict__ y, const EpiArgs &a, double &s0, double &s1, double &m)
{
  if (EPI == PMH_EPI_NONE) {
    stg<NT>(&y[r], sum);
  } else if (EPI == PMH_EPI_ADD) {
    stg<NT>(&y[r], ldg<NT>(&a.y1[r]) + sum);
  } else if (EPI == PMH_EPI_SUB) {
    stg<NT>(&y[r], sum - ldg<NT>(&a.y1[r]));
  } else { // PMH_EPI_MPGP: Ap = A p with p'Ap, g'p and QPCFeas_Box(x,p) accumulated (mpgp.c:537-544)
    stg<NT>(&y[r], sum);
    double p = xin[r];
    s0 += p * sum;
    s1 += ldg<NT>(&a.g[r]) * p;
    if (p > 0. && a.lb) {
      double l = ldg<NT>(&a.lb[r]);
      if (l > -INFINITY) m = fmin(m, (ldg<NT>(&a.xx[r]) - l) / p);
    }
    if (p < 0. && a.ub) {
      double u = ldg<NT>(&a.ub[r]);
      if (u < INFINITY) m = fmin(m, (ldg<NT>(&a.xx[r]) - u) / p);
    }
  }
}

template <int EPI>
__device__ __forceinline__ void epi_finish(double s0, double s1, double m, double *lds, double *__restrict__ part, int ld, int slot)
{
  if (EPI == PMH_EPI_MPGP) {
    s0 = pmh_block_reduce<PMH_RED_SUM>(s0, lds);
    s1 = pmh_block_reduce<PMH_RED_SUM>(s1, lds);
    m  = pmh_block_reduce<PMH_RED_MIN>(m, lds);
    if (threadIdx.x == 0) {
      part[slot]          = s0;
      part[ld + slot]     = s1;
      part[2 * ld + slot] = m;
    }
  }
}

__device__ __forceinline__ int xcd_remap(int bid, int nlaunch)
{
  const int chunk = nlaunch >> 3;
  return (bid & 7) * chunk + (bid >> 3);
}

// Row-block stream kernel.  MODE 0: one row block per workgroup (grid = #row blocks, XCD-remapped).
// MODE 1/2: persistent grid of 8*W workgroups (W <= 256 per XCD, all resident); XCD x owns the contiguous
// slab of row blocks [x*chunk,(x+1)*chunk) and its W workgroups walk it with stride W, so one XCD works on a
// window of W consecutive row blocks at a time (x stays in its L2) and only 8*W partials reach the finalise
// kernel.  MODE 1 double-buffers the LDS tile (one barrier per row block, 2 tiles of LDS), MODE 2 keeps one
// tile (two barriers).  The next row block's val/col stream is issued before the current per-row phase.
// NT: val/col are read exactly once per SpMV -> non-temporal loads keep them from evicting x out of L2.
// C16: the column indices of the stream come as 16-bit offsets from the row block's smallest column (col16 / cbase, built when every row block
// spans < 65 536 columns: banded matrices such as the 5-point Laplacian of configs[1]): 10 instead of 12 bytes per non-zero.
template <int EPI, int NNZB, int MODE, bool NT, bool VL2, int RL, bool C16 = false>
__global__ __launch_bounds__(PMH_BLOCK) void k_spmv_stream(const int *__restrict__ rowblocks, int nrb, int chunk, const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y, EpiArgs a, double *__restrict__ part, int ld,
                                                          const unsigned short *__restrict__ col16 = nullptr, const int *__restrict__ cbase = nullptr)
{
  if (a.halt && *a.halt) return; // uniform: every workgroup reads the same flag
  constexpr int     ITEMS = NNZB / PMH_BLOCK;
  constexpr int     NBUF  = (MODE == 1) ? 2 : 1;
  __shared__ double prod[NBUF][NNZB];
  __shared__ double red[PMH_BLOCK / 64];
  const int         tid = threadIdx.x;
  int               b, end, W;
  if (MODE == 0) {
    b   = xcd_remap(blockIdx.x, gridDim.x);
    end = (b < nrb) ? b + 1 : b;
    W   = 1;
  } else {
    W             = gridDim.x >> 3;
    const int xcd = blockIdx.x & 7;
    b             = xcd * chunk + (blockIdx.x >> 3);
    end           = min(nrb, (xcd + 1) * chunk);
  }
  double acc0 = 0.0, acc1 = 0.0, amin = INFINITY;
  int    c[ITEMS];
  double v[ITEMS];
  int    r0 = 0, r1 = 0, s0 = 0, s1 = 0;

  typedef double dbl2 __attribute__((ext_vector_type(2)));
  typedef int    int2v __attribute__((ext_vector_type(2)));
  auto prefetch = [&](int bb) {
    r0 = rowblocks[bb], r1 = rowblocks[bb + 1];
    s0 = rowptr[r0], s1 = rowptr[r1];
    if (s1 - s0 <= NNZB - (VL2 ? 1 : 0)) {
      if (VL2) {
        // 16-byte val / 8-byte col loads: start at the even index below s0 (arrays are padded by 2 entries)
        const int s0a = s0 & ~1;
#pragma unroll
        for (int j = 0; j < ITEMS / 2; j++) {
          const int k  = s0a + 2 * (tid + j * PMH_BLOCK);
          dbl2      vv = {0.0, 0.0};
          int2v     cc = {-1, -1};
          if (k < s1) {
            vv = NT ? __builtin_nontemporal_load((const dbl2 *)(val + k)) : *(const dbl2 *)(val + k);
            cc = NT ? __builtin_nontemporal_load((const int2v *)(col + k)) : *(const int2v *)(col + k);
          }
          c[2 * j]     = (k >= s0 && k < s1) ? cc.x : -1;
          c[2 * j + 1] = (k + 1 < s1) ? cc.y : -1;
          v[2 * j]     = vv.x;
          v[2 * j + 1] = vv.y;
        }
      } else {
        const int cb = C16 ? cbase[bb] : 0;
#pragma unroll
        for (int j = 0; j < ITEMS; j++) {
          const int k = s0 + tid + j * PMH_BLOCK;
          if (C16) {
            c[j] = (k < s1) ? cb + (int)(NT ? __builtin_nontemporal_load(&col16[k]) : col16[k]) : -1;
            v[j] = (k < s1) ? (NT ? __builtin_nontemporal_load(&val[k]) : val[k]) : 0.0;
          } else if (NT) {
            c[j] = (k < s1) ? __builtin_nontemporal_load(&col[k]) : -1;
            v[j] = (k < s1) ? __builtin_nontemporal_load(&val[k]) : 0.0;
          } else {
            c[j] = (k < s1) ? col[k] : -1;
            v[j] = (k < s1) ? val[k] : 0.0;
          }
        }
      }
    }
  };

  if (b < end) prefetch(b);
  int buf = 0;
  while (b < end) {
    const int cr0 = r0, cr1 = r1, cs0 = s0, cs1 = s1;
    const int nb  = b + W;
    if (cs1 - cs0 <= NNZB - (VL2 ? 1 : 0)) {
      double *pr = prod[buf];
      if (MODE == 2) __syncthreads(); // previous row phase done before the tile is overwritten
      if (VL2) {
        const int sh = (cs0 & ~1) - cs0; // 0 or -1
#pragma unroll
        for (int j = 0; j < ITEMS / 2; j++) {
          const int k = sh + 2 * (tid + j * PMH_BLOCK);
          if (c[2 * j] >= 0) pr[k] = v[2 * j] * x[c[2 * j]];
          if (c[2 * j + 1] >= 0) pr[k + 1] = v[2 * j + 1] * x[c[2 * j + 1]];
        }
      } else {
#pragma unroll
        for (int j = 0; j < ITEMS; j++)
          if (c[j] >= 0) pr[tid + j * PMH_BLOCK] = v[j] * x[c[j]];
      }
      // short rows: the lane's first row of this block and the operands of its epilogue (row pointers, p, g, x, lb / ub, y1) are
      // requested BEFORE the barrier, so that their latency runs under the barrier and the LDS phase instead of after it
      int    fr = cr0 + tid, fk0 = 0, fk1 = 0;
      double fp = 0.0, fg = 0.0, fx = 0.0, fl = -INFINITY, fu = INFINITY, fy1 = 0.0;
      if (RL == 1 && fr < cr1) {
        fk0 = rowptr[fr] - cs0, fk1 = rowptr[fr + 1] - cs0;
        if (EPI == PMH_EPI_MPGP) {
          fp = x[fr], fg = a.g[fr], fx = a.xx[fr];
          if (a.lb) fl = a.lb[fr];
          if (a.ub) fu = a.ub[fr];
        } else if (EPI == PMH_EPI_ADD || EPI == PMH_EPI_SUB) {
          fy1 = a.y1[fr];
        }
      }
      if (MODE != 0 && nb < end) prefetch(nb); // next block's stream in flight during this block's row phase
      __syncthreads();
      if (RL == 1) {
        // one lane per row, left-to-right sum (bit-identical to MatMult_SeqAIJ)
        if (fr < cr1) {
          double sum = 0.0;
          for (int k = fk0; k < fk1; k++) sum += pr[k];
          if (EPI == PMH_EPI_NONE) {
            y[fr] = sum;
          } else if (EPI == PMH_EPI_ADD) {
            y[fr] = fy1 + sum;
          } else if (EPI == PMH_EPI_SUB) {
            y[fr] = sum - fy1;
          } else { // PMH_EPI_MPGP: same arithmetic as epi_row
            y[fr] = sum;
            acc0 += fp * sum;
            acc1 += fg * fp;
            if (fp > 0. && fl > -INFINITY) amin = fmin(amin, (fx - fl) / fp);
            if (fp < 0. && fu < INFINITY) amin = fmin(amin, (fx - fu) / fp);
          }
        }
        for (int r = fr + PMH_BLOCK; r < cr1; r += PMH_BLOCK) { // (nearly) empty rows: more rows than lanes in a block
          const int k0 = rowptr[r] - cs0, k1 = rowptr[r + 1] - cs0;
          double    sum = 0.0;
          for (int k = k0; k < k1; k++) sum += pr[k];
          epi_row<EPI, false>(r, sum, x, y, a, acc0, acc1, amin);
        }
      } else {
        // medium rows (e.g. 81 nnz/row elasticity blocks): RL lanes per row over the LDS tile + shuffle tree
        const int sub = tid / RL, lane = tid % RL;
        for (int rb = cr0; rb < cr1; rb += PMH_BLOCK / RL) { // uniform trip count for the shuffles
          const int r   = rb + sub;
          double    sum = 0.0;
          if (r < cr1) {
            const int k0 = rowptr[r] - cs0, k1 = rowptr[r + 1] - cs0;
            for (int k = k0 + lane; k < k1; k += RL) sum += pr[k];
          }
#pragma unroll
          for (int o = RL / 2; o > 0; o >>= 1) sum += __shfl_down(sum, o, RL);
          if (r < cr1 && lane == 0) epi_row<EPI, false>(r, sum, x, y, a, acc0, acc1, amin);
        }
      }
      if (MODE == 1) buf ^= 1;
    } else {
      // a single row longer than the LDS tile: the whole workgroup strides over it
      double sum = 0.0;
      for (int k = cs0 + tid; k < cs1; k += PMH_BLOCK) sum += val[k] * x[col[k]];
      sum = pmh_block_reduce<PMH_RED_SUM>(sum, red);
      if (tid == 0) epi_row<EPI>(cr0, sum, x, y, a, acc0, acc1, amin);
      if (MODE != 0 && nb < end) prefetch(nb);
    }
    b = nb;
  }
  if (MODE == 0) {
    const int lb_ = xcd_remap(blockIdx.x, gridDim.x);
    if (lb_ < nrb) epi_finish<EPI>(acc0, acc1, amin, red, part, ld, lb_);
  } else {
    epi_finish<EPI>(acc0, acc1, amin, red, part, ld, blockIdx.x);
  }
}

// Uniformly short rows (the 3-5 non-zeros per row of configs[1]'s 5-point Laplacian): a device-private slot-major copy -- per block of 256 rows, slot k of
// row r at [block][k][r], columns as 16-bit offsets from the block's smallest column where that fits -- lets one thread own one row with fully coalesced
// loads and NO LDS staging or barrier; the row is summed left to right over its own entries only (bit-identical to MatMult_SeqAIJ and to the stream
// kernel), padded slots are loaded but not added.  Same persistent grid and XCD slabs as the stream kernel (mode 2), same epilogues.
template <int EPI, bool C16, int W>
__global__ __launch_bounds__(PMH_BLOCK) void k_spmv_ell(int nrows, int nrb, int chunk, const int *__restrict__ rowptr, const double *__restrict__ ev, const unsigned short *__restrict__ ec16,
                                                        const int *__restrict__ ecol, const int *__restrict__ ecbase, const double *__restrict__ x, double *__restrict__ y, EpiArgs a,
                                                        double *__restrict__ part, int ld)
{
  __shared__ double red[PMH_BLOCK / 64];
  if (a.halt && *a.halt) return;
  const int Wg = gridDim.x >> 3, xcd = blockIdx.x & 7, end = min(nrb, (xcd + 1) * chunk), tid = threadIdx.x;
  double    acc0 = 0.0, acc1 = 0.0, amin = INFINITY;
  for (int b = xcd * chunk + (blockIdx.x >> 3); b < end; b += Wg) {
    const int r = b * PMH_BLOCK + tid;
    if (r < nrows) {
      const int       cnt = rowptr[r + 1] - rowptr[r], cb = C16 ? ecbase[b] : 0;
      const long long o   = (long long)b * W * PMH_BLOCK + tid;
      double          v[W];
      int             c[W];
#pragma unroll
      for (int k = 0; k < W; k++) {
        v[k] = __builtin_nontemporal_load(&ev[o + (long long)k * PMH_BLOCK]);
        c[k] = C16 ? cb + (int)__builtin_nontemporal_load(&ec16[o + (long long)k * PMH_BLOCK]) : __builtin_nontemporal_load(&ecol[o + (long long)k * PMH_BLOCK]);
      }
      double xv[W];
#pragma unroll
      for (int k = 0; k < W; k++) xv[k] = x[c[k]]; // padded slots point at the row's first column
      double sum = 0.0;
#pragma unroll
      for (int k = 0; k < W; k++)
        if (k < cnt) sum += v[k] * xv[k];
      epi_row<EPI, false>(r, sum, x, y, a, acc0, acc1, amin);
    }
  }
  epi_finish<EPI>(acc0, acc1, amin, red, part, ld, blockIdx.x);
}

template <int EPI, int LPR, bool NT>
__global__ __launch_bounds__(PMH_BLOCK) void k_spmv_vector(int nrows, int nblk, int nlaunch, const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y, EpiArgs a, double *__restrict__ part, int ld)
{
  __shared__ double red[PMH_BLOCK / 64];
  if (a.halt && *a.halt) return;
  const int         b = xcd_remap(blockIdx.x, nlaunch);
  if (b >= nblk) return;
  constexpr int RPB  = PMH_BLOCK / LPR;
  const int     sub  = threadIdx.x / LPR, lane = threadIdx.x % LPR;
  const int     r    = b * RPB + sub;
  double        acc0 = 0.0, acc1 = 0.0, amin = INFINITY;
  double        sum = 0.0;
  if (r < nrows) {
    const int k0 = rowptr[r], k1 = rowptr[r + 1];
    for (int k = k0 + lane; k < k1; k += LPR) sum += (NT ? __builtin_nontemporal_load(&val[k]) : val[k]) * x[NT ? __builtin_nontemporal_load(&col[k]) : col[k]];
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) sum += __shfl_down(sum, o, LPR);
  if (r < nrows && lane == 0) epi_row<EPI>(r, sum, x, y, a, acc0, acc1, amin);
  epi_finish<EPI>(acc0, acc1, amin, red, part, ld, b);
}

// Very long rows (G = R'B' of the FETI coarse problem: 6 rows per subdomain, ~10^4 non-zeros each; QPPFApplyQ/P/GtG and the
// ||G u|| of the SMALXE convergence test apply it ~5 times per MPGP step).  One workgroup per row leaves 48 workgroups
// striding 25 000 non-zeros each (182 us measured); here every row is cut into chunks of PMH_LONG_CHUNK non-zeros, one
// workgroup per chunk (fixed-tree block reduction), and a second small kernel adds the chunk sums of a row in chunk order
// and applies the epilogue -- deterministic, ~2 x 5 us.
#define PMH_LONG_CHUNK 4096
template <bool NT>
__global__ __launch_bounds__(PMH_BLOCK) void k_spmv_long_part(const int *__restrict__ chunks, const int *__restrict__ col, const double *__restrict__ val, const double *__restrict__ x, const int *__restrict__ halt, double *__restrict__ part)
{
  __shared__ double red[PMH_BLOCK / 64];
  if (halt && *halt) return;
  const int k0 = chunks[3 * blockIdx.x + 1], k1 = chunks[3 * blockIdx.x + 2];
  double    s[4] = {0.0, 0.0, 0.0, 0.0};
  int       k    = k0 + (int)threadIdx.x;
  if (k1 - k0 == 16 * PMH_BLOCK) { // a full chunk of 4096 (PMH_LONG_CHUNK): all 16 entries of the thread in flight at once, added in the order of the loop below
    double v[16];
    int    c[16];
#pragma unroll
    for (int e = 0; e < 16; e++) {
      v[e] = NT ? __builtin_nontemporal_load(&val[k + e * PMH_BLOCK]) : val[k + e * PMH_BLOCK];
      c[e] = NT ? __builtin_nontemporal_load(&col[k + e * PMH_BLOCK]) : col[k + e * PMH_BLOCK];
    }
    double xv[16];
#pragma unroll
    for (int e = 0; e < 16; e++) xv[e] = x[c[e]];
#pragma unroll
    for (int e = 0; e < 16; e++) s[e & 3] += v[e] * xv[e];
    k = k1;
  }
  for (; k + 3 * PMH_BLOCK < k1; k += 4 * PMH_BLOCK) {
#pragma unroll
    for (int j = 0; j < 4; j++) s[j] += (NT ? __builtin_nontemporal_load(&val[k + j * PMH_BLOCK]) : val[k + j * PMH_BLOCK]) * x[NT ? __builtin_nontemporal_load(&col[k + j * PMH_BLOCK]) : col[k + j * PMH_BLOCK]];
  }
#pragma unroll
  for (int j = 0; j < 3; j++) { // at most three strides are left
    const int kk = k + j * PMH_BLOCK;
    if (kk < k1) s[j] += val[kk] * x[col[kk]];
  }
  const double t = pmh_block_reduce<PMH_RED_SUM>((s[0] + s[1]) + (s[2] + s[3]), red);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// The chunk sums of A x and A x2 in ONE pass over A (SMALXE's G0 u next to the projector's G0 p): every sum exactly as k_spmv_long_part takes it
template <bool NT>
__global__ __launch_bounds__(PMH_BLOCK) void k_spmv_long_part2(const int *__restrict__ chunks, const int *__restrict__ col, const double *__restrict__ val, const double *__restrict__ x, const double *__restrict__ x2,
                                                               double *__restrict__ part, double *__restrict__ part2)
{
  __shared__ double red[PMH_BLOCK / 64];
  const int k0 = chunks[3 * blockIdx.x + 1], k1 = chunks[3 * blockIdx.x + 2];
  double    s[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
  int       k    = k0 + (int)threadIdx.x;
  if (k1 - k0 == 16 * PMH_BLOCK) {
    double v[16];
    int    c[16];
#pragma unroll
    for (int e = 0; e < 16; e++) {
      v[e] = NT ? __builtin_nontemporal_load(&val[k + e * PMH_BLOCK]) : val[k + e * PMH_BLOCK];
      c[e] = NT ? __builtin_nontemporal_load(&col[k + e * PMH_BLOCK]) : col[k + e * PMH_BLOCK];
    }
    double xv[16], xw[16];
#pragma unroll
    for (int e = 0; e < 16; e++) xv[e] = x[c[e]], xw[e] = x2[c[e]];
#pragma unroll
    for (int e = 0; e < 16; e++) s[e & 3] += v[e] * xv[e], s2[e & 3] += v[e] * xw[e];
    k = k1;
  }
  for (; k + 3 * PMH_BLOCK < k1; k += 4 * PMH_BLOCK) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const double vv = NT ? __builtin_nontemporal_load(&val[k + j * PMH_BLOCK]) : val[k + j * PMH_BLOCK];
      const int    cc = NT ? __builtin_nontemporal_load(&col[k + j * PMH_BLOCK]) : col[k + j * PMH_BLOCK];
      s[j] += vv * x[cc], s2[j] += vv * x2[cc];
    }
  }
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const int kk = k + j * PMH_BLOCK;
    if (kk < k1) s[j] += val[kk] * x[col[kk]], s2[j] += val[kk] * x2[col[kk]];
  }
  const double t = pmh_block_reduce<PMH_RED_SUM>((s[0] + s[1]) + (s[2] + s[3]), red);
  const double t2 = pmh_block_reduce<PMH_RED_SUM>((s2[0] + s2[1]) + (s2[2] + s2[3]), red);
  if (threadIdx.x == 0) part[blockIdx.x] = t, part2[blockIdx.x] = t2;
}

template <int EPI>
__global__ __launch_bounds__(PMH_BLOCK) void k_spmv_long_fin(int nrows, const int *__restrict__ lrow, const double *__restrict__ part, const double *__restrict__ x, double *__restrict__ y, EpiArgs a)
{
  if (a.halt && *a.halt) return;
  const int r = blockIdx.x * PMH_BLOCK + threadIdx.x;
  if (r >= nrows) return;
  double sum = 0.0, d0 = 0.0, d1 = 0.0, dm = 0.0;
  for (int c = lrow[r]; c < lrow[r + 1]; c++) sum += part[c];
  epi_row<EPI>(r, sum, x, y, a, d0, d1, dm);
}

// y = M (A x) for a matrix with few rows (the m = 48 ... 384 rows of G) and a small dense m x m matrix M handed over TRANSPOSED (Mt[c * m + r] =
// M[r][c]: lane r reads contiguously): the chunk sums of k_spmv_long_part are added per row in chunk order into LDS, then every row of M is
// applied by one thread, left to right -- the finishing launch of the long-row product and the dense product in ONE launch of one workgroup
// (the implicitly orthonormalised G = T G0 of qppf.hip: G0 keeps its sparsity, T or T'T is applied here).  lrow == nullptr: `part` already
// holds A x (the product of a short-row matrix).
// norm_d / norm_h != nullptr (m <= 64 only): also ||y||^2, summed exactly as k_dot + k_finalize sum a vector of this length (one block: products in
// wave 0, the fixed shuffle tree, zeros from the other waves) -- the same bits, two launches less (SMALXE's ||B u|| of every inner iteration)
__global__ __launch_bounds__(PMH_BLOCK) void k_rows_then_dense(int m, const int *__restrict__ lrow, const double *__restrict__ part, const double *__restrict__ Mt, double *__restrict__ y,
                                                               double *__restrict__ norm_d = nullptr, double *__restrict__ norm_h = nullptr)
{
  extern __shared__ double t0[];
  __shared__ double        red[PMH_BLOCK / 64];
  // the sums keep their order, their loads travel in batches (a plain loop compiles to load - wait - add per entry: this kernel is made of latencies)
  for (int r = threadIdx.x; r < m; r += PMH_BLOCK) {
    double sum = 0.0;
    if (lrow) {
      const int c0 = lrow[r], c1 = lrow[r + 1];
      for (int c = c0; c < c1; c += 8) {
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = (c + j < c1) ? part[c + j] : 0.0;
#pragma unroll
        for (int j = 0; j < 8; j++)
          if (c + j < c1) sum += v[j];
      }
    } else sum = part[r];
    t0[r] = sum;
  }
  __syncthreads();
  double sq = 0.0;
  for (int r = threadIdx.x; r < m; r += PMH_BLOCK) {
    double s = 0.0;
    for (int c = 0; c < m; c += 16) {
      double v[16];
#pragma unroll
      for (int j = 0; j < 16; j++) v[j] = (c + j < m) ? Mt[(size_t)(c + j) * m + r] : 0.0;
#pragma unroll
      for (int j = 0; j < 16; j++)
        if (c + j < m) s += v[j] * t0[c + j];
    }
    y[r] = s;
    sq += s * s;
  }
  if (norm_d) { // uniform
    sq = pmh_block_reduce<PMH_RED_SUM>(sq, red);
    if (threadIdx.x == 0) *norm_d = sq, *norm_h = sq;
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------
static int build_rowblocks(int nrows, const int *rowptr, int nnzb, std::vector<int> &rb)
{
  rb.clear();
  rb.push_back(0);
  int r = 0;
  while (r < nrows) {
    int start = r;
    int base  = rowptr[r];
    while (r < nrows && r - start < PMH_MAX_ROWS_PER_BLOCK && rowptr[r + 1] - base <= nnzb) r++;
    if (r == start) r++; // single row longer than a tile
    rb.push_back(r);
  }
  return (int)rb.size() - 1;
}

extern "C" int pmh_csr_create(pmh_ctx ctx, int nrows, int ncols, const int *rowptr, const int *col, const double *val, pmh_csr *out)
{
  PMH_ARG(ctx && out && nrows >= 0 && ncols >= 0 && rowptr);
  PMH_ARG(rowptr[0] == 0);
  for (int i = 0; i < nrows; i++)
    if (rowptr[i + 1] < rowptr[i]) return pmh_set_error(PMH_ERR_ARG, "pmh_csr_create: rowptr not monotone at row %d", i);
  const long long nnz = rowptr[nrows];
  PMH_ARG(nnz == 0 || (col && val));
  for (long long k = 0; k < nnz; k++)
    if (col[k] < 0 || col[k] >= ncols) return pmh_set_error(PMH_ERR_ARG, "pmh_csr_create: column index %d out of range [0,%d) at nnz %lld", col[k], ncols, k);
  PMH_HIP(hipSetDevice(ctx->device));
  pmh_csr A = new pmh_csr_s();
  {
    static std::atomic<unsigned long long> next_uid{1};
    A->uid = next_uid++;
  }
  memset(A, 0, sizeof(*A));
  A->ctx   = ctx;
  A->nrows = nrows;
  A->ncols = ncols;
  A->nnz   = nnz;
  PMH_HIP(hipMalloc((void **)&A->d_rowptr, sizeof(int) * ((size_t)nrows + 1)));
  PMH_HIP(hipMalloc((void **)&A->d_col, sizeof(int) * (size_t)(nnz + 2)));    // +2: the vector-load variant may touch one entry past the end
  PMH_HIP(hipMalloc((void **)&A->d_val, sizeof(double) * (size_t)(nnz + 2)));
  PMH_HIP(hipMemsetAsync(A->d_col + nnz, 0, sizeof(int) * 2, ctx->stream));
  PMH_HIP(hipMemsetAsync(A->d_val + nnz, 0, sizeof(double) * 2, ctx->stream));
  PMH_CHK(pmh_memcpy_h2d(ctx, A->d_rowptr, rowptr, sizeof(int) * ((size_t)nrows + 1)));
  PMH_CHK(pmh_memcpy_h2d(ctx, A->d_col, col, sizeof(int) * (size_t)nnz));
  PMH_CHK(pmh_memcpy_h2d(ctx, A->d_val, val, sizeof(double) * (size_t)nnz));

  const double avg = nrows ? (double)nnz / nrows : 0.0;
  int medium_stream = 1, m_nnzb = 2048, m_mode = 0, m_nt = 3; // measured best on 81 nnz/row K_i (round 4 sweep at 43^3 x 8 distinct blocks: nt 3 = non-temporal 16-byte value / 8-byte column loads 0.391 ms, nt 1 0.406; 1024 / 4096 entries per row block 0.41 / 0.43)
  const bool medium = avg > 24.0 && avg <= 256.0 && medium_stream;
  if (avg <= 24.0 || avg > 1024.0 || medium) { // short rows: LDS-staged row blocks; very long rows (G of the coarse problem): one workgroup per row
    A->kind = PMH_SPMV_STREAM;
    // tile (entries per row block), mode (0 one row block per workgroup / 2 persistent grid), non-temporal mask: chosen from the sweeps of rounds 1-4 (profiles/r01_spmv_tune_*.txt;
    // the tuning knobs PMH_SPMV_TUNE / _MTUNE / _VTUNE went at the end of round 6)
    A->st_nnzb = 1024, A->st_mode = 2, A->st_nt = 1, A->st_rl = 1;
    if (medium) A->st_nnzb = m_nnzb, A->st_mode = m_mode, A->st_nt = m_nt, A->st_rl = 8;
    std::vector<int> rb;
    A->n_rowblocks = build_rowblocks(nrows, rowptr, A->st_nnzb - 1, rb); // -1: room for the aligned-down start of the 16-byte load variant
    PMH_HIP(hipMalloc((void **)&A->d_rowblocks, sizeof(int) * rb.size()));
    PMH_CHK(pmh_memcpy_h2d(ctx, A->d_rowblocks, rb.data(), sizeof(int) * rb.size()));
    if (A->st_rl == 1 && !(A->st_nt & 2) && A->st_nnzb <= 2048) {
      // 16-bit column offsets per row block where every block spans < 65 536 columns (device-private copy next to the int32 indices)
      std::vector<int>            cb((size_t)A->n_rowblocks, 0);
      std::vector<unsigned short> c16((size_t)nnz + 8, 0);
      bool                        ok = true;
      for (int b = 0; b < A->n_rowblocks && ok; b++) {
        const int k0 = rowptr[rb[b]], k1 = rowptr[rb[b + 1]];
        int       lo = INT_MAX, hi = -1;
        for (int k = k0; k < k1; k++) lo = std::min(lo, col[k]), hi = std::max(hi, col[k]);
        if (k1 > k0 && hi - lo > 65535) ok = false;
        cb[b] = k1 > k0 ? lo : 0;
        for (int k = k0; k < k1 && ok; k++) c16[k] = (unsigned short)(col[k] - lo);
      }
      if (ok && A->n_rowblocks > 0) {
        PMH_HIP(hipMalloc((void **)&A->d_col16, sizeof(unsigned short) * c16.size()));
        PMH_HIP(hipMalloc((void **)&A->d_cbase, sizeof(int) * cb.size()));
        PMH_CHK(pmh_memcpy_h2d(ctx, A->d_col16, c16.data(), sizeof(unsigned short) * c16.size()));
        PMH_CHK(pmh_memcpy_h2d(ctx, A->d_cbase, cb.data(), sizeof(int) * cb.size()));
      }
    }
    if (A->st_rl == 1 && A->st_mode == 2 && nrows > 0) {
      // slot-major copy for uniformly short rows (k_spmv_ell): <= 8 non-zeros per row and at most 25 % padding
      int wmax = 0;
      for (int r = 0; r < nrows; r++) wmax = std::max(wmax, rowptr[r + 1] - rowptr[r]);
      if (wmax >= 1 && wmax <= 8 && (double)wmax * nrows <= 1.25 * (double)nnz) {
        const int nrb_e = (nrows + PMH_BLOCK - 1) / PMH_BLOCK;
        std::vector<double>         ev((size_t)nrb_e * wmax * PMH_BLOCK, 0.0);
        std::vector<int>            ec((size_t)nrb_e * wmax * PMH_BLOCK, 0), cb((size_t)nrb_e, 0);
        bool                        fit16 = true;
        for (int b = 0; b < nrb_e; b++) {
          int lo = INT_MAX, hi = -1;
          for (int r = b * PMH_BLOCK; r < std::min(nrows, (b + 1) * PMH_BLOCK); r++)
            for (int k = rowptr[r]; k < rowptr[r + 1]; k++) lo = std::min(lo, col[k]), hi = std::max(hi, col[k]);
          cb[b] = hi >= 0 ? lo : 0;
          if (hi >= 0 && hi - lo > 65535) fit16 = false;
          for (int r = b * PMH_BLOCK; r < std::min(nrows, (b + 1) * PMH_BLOCK); r++) {
            const int k0 = rowptr[r], cnt = rowptr[r + 1] - k0;
            for (int k = 0; k < wmax; k++) {
              const size_t o = ((size_t)b * wmax + k) * PMH_BLOCK + (r - b * PMH_BLOCK);
              ev[o] = k < cnt ? val[k0 + k] : 0.0;
              ec[o] = k < cnt ? col[k0 + k] : (cnt ? col[k0] : cb[b]);
            }
          }
        }
        PMH_HIP(hipMalloc((void **)&A->d_ell_val, sizeof(double) * ev.size()));
        PMH_CHK(pmh_memcpy_h2d(ctx, A->d_ell_val, ev.data(), sizeof(double) * ev.size()));
        PMH_HIP(hipMalloc((void **)&A->d_ell_cbase, sizeof(int) * cb.size()));
        PMH_CHK(pmh_memcpy_h2d(ctx, A->d_ell_cbase, cb.data(), sizeof(int) * cb.size()));
        if (fit16) {
          std::vector<unsigned short> e16(ec.size());
          for (int b = 0; b < nrb_e; b++)
            for (size_t o = (size_t)b * wmax * PMH_BLOCK; o < (size_t)(b + 1) * wmax * PMH_BLOCK; o++) e16[o] = (unsigned short)(ec[o] - cb[b]);
          // rows past the end of the last block keep offset 0 - base: clamp
          PMH_HIP(hipMalloc((void **)&A->d_ell_c16, sizeof(unsigned short) * e16.size()));
          PMH_CHK(pmh_memcpy_h2d(ctx, A->d_ell_c16, e16.data(), sizeof(unsigned short) * e16.size()));
        } else {
          PMH_HIP(hipMalloc((void **)&A->d_ell_col, sizeof(int) * ec.size()));
          PMH_CHK(pmh_memcpy_h2d(ctx, A->d_ell_col, ec.data(), sizeof(int) * ec.size()));
        }
        A->ell_w = wmax, A->ell_nrb = nrb_e;
      }
    }
    const int chunk = (A->n_rowblocks + 7) / 8; // row blocks per XCD
    if (A->st_mode == 0) {
      A->n_launch_blocks = 8 * (chunk > 0 ? chunk : 1);
    } else {
      const int wcap     = (A->st_mode == 1 || A->st_nnzb == 4096) ? 128 : 256; // resident workgroups per XCD (LDS bound)
      const int W        = chunk < wcap ? (chunk > 0 ? chunk : 1) : wcap;
      A->n_launch_blocks = 8 * W;
    }
  } else {
    A->kind          = PMH_SPMV_VECTOR;
    A->lanes_per_row = (avg <= 48.0) ? 8 : (avg <= 160.0 ? 16 : (avg <= 512.0 ? 32 : 64));
    A->st_nt         = 1;
    const int rpb    = PMH_BLOCK / A->lanes_per_row;
    A->n_rowblocks   = (nrows + rpb - 1) / rpb;
    A->n_launch_blocks = ((A->n_rowblocks + 7) / 8) * 8;
  }
  PMH_HIP(hipMalloc((void **)&A->d_blockpart, sizeof(double) * 3 * (size_t)(A->n_launch_blocks ? A->n_launch_blocks : 8)));
  if (avg > 1024.0 && !getenv("PMH_SPMV_NO_LONG")) { // chunk table of the long-row kernels (plain / ADD / SUB epilogues)
    std::vector<int> ch, lrow((size_t)nrows + 1, 0);
    const int        long_chunk = PMH_LONG_CHUNK;
    for (int r = 0; r < nrows; r++) {
      for (int k = rowptr[r]; k < rowptr[r + 1]; k += long_chunk) {
        ch.push_back(r), ch.push_back(k), ch.push_back(std::min(k + long_chunk, rowptr[r + 1]));
      }
      lrow[r + 1] = (int)(ch.size() / 3);
    }
    A->l_nchunks = (int)(ch.size() / 3);
    if (A->l_nchunks) {
      PMH_HIP(hipMalloc((void **)&A->d_lchunks, sizeof(int) * ch.size()));
      PMH_HIP(hipMalloc((void **)&A->d_lrow, sizeof(int) * lrow.size()));
      PMH_HIP(hipMalloc((void **)&A->d_lpart, sizeof(double) * (size_t)A->l_nchunks));
      PMH_CHK(pmh_memcpy_h2d(ctx, A->d_lchunks, ch.data(), sizeof(int) * ch.size()));
      PMH_CHK(pmh_memcpy_h2d(ctx, A->d_lrow, lrow.data(), sizeof(int) * lrow.size()));
    }
  }
  *out = A;
  return PMH_SUCCESS;
}

extern "C" int pmh_csr_destroy(pmh_csr A)
{
  if (!A) return PMH_SUCCESS;
  hipStreamSynchronize(A->ctx->stream);
  if (A->transpose) pmh_csr_destroy(A->transpose);
  if (A->ev) pmh_csr_timing_enable(A, 0);
  hipFree(A->d_rowptr);
  hipFree(A->d_col);
  hipFree(A->d_val);
  hipFree(A->d_rowblocks);
  if (A->d_col16) hipFree(A->d_col16), hipFree(A->d_cbase);
  if (A->d_ell_val) hipFree(A->d_ell_val), hipFree(A->d_ell_cbase);
  if (A->d_ell_c16) hipFree(A->d_ell_c16);
  if (A->d_ell_col) hipFree(A->d_ell_col);
  hipFree(A->d_blockpart);
  if (A->d_lchunks) hipFree(A->d_lchunks);
  if (A->d_lrow) hipFree(A->d_lrow);
  if (A->d_lpart) hipFree(A->d_lpart);
  if (A->d_lpart2) hipFree(A->d_lpart2);
  delete A;
  return PMH_SUCCESS;
}

extern "C" int pmh_csr_sizes(pmh_csr A, int *nrows, int *ncols, long long *nnz)
{
  PMH_ARG(A);
  if (nrows) *nrows = A->nrows;
  if (ncols) *ncols = A->ncols;
  if (nnz) *nnz = A->nnz;
  return PMH_SUCCESS;
}

extern "C" int pmh_csr_algorithmic_bytes(pmh_csr A, double *bytes)
{
  PMH_ARG(A && bytes);
  *bytes = 12.0 * (double)A->nnz + 20.0 * (double)A->nrows;
  return PMH_SUCCESS;
}

template <int EPI>
static int launch(pmh_csr A, const double *x, double *y, const EpiArgs &a)
{
  pmh_ctx   ctx = A->ctx;
  const int nl  = A->n_launch_blocks;
  if (A->nrows == 0) return PMH_SUCCESS;
  if (A->l_nchunks && EPI != PMH_EPI_MPGP) {
    const bool long_nt = true; // (non-temporal matrix stream of the long-row kernels)
    if (long_nt) hipLaunchKernelGGL(k_spmv_long_part<true>, dim3(A->l_nchunks), dim3(PMH_BLOCK), 0, ctx->stream, (const int *)A->d_lchunks, (const int *)A->d_col, (const double *)A->d_val, x, a.halt, A->d_lpart);
    else hipLaunchKernelGGL(k_spmv_long_part<false>, dim3(A->l_nchunks), dim3(PMH_BLOCK), 0, ctx->stream, (const int *)A->d_lchunks, (const int *)A->d_col, (const double *)A->d_val, x, a.halt, A->d_lpart);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_spmv_long_fin<EPI>), dim3((A->nrows + PMH_BLOCK - 1) / PMH_BLOCK), dim3(PMH_BLOCK), 0, ctx->stream, A->nrows, (const int *)A->d_lrow, (const double *)A->d_lpart, x, y, a);
    PMH_HIP(hipGetLastError());
    return PMH_SUCCESS;
  }
  if (A->kind == PMH_SPMV_STREAM && A->d_ell_val) {
    const int chunk = (A->ell_nrb + 7) / 8;
#define ELL_LAUNCH(WW) \
  do { \
    if (A->d_ell_c16) \
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_spmv_ell<EPI, true, WW>), dim3(nl), dim3(PMH_BLOCK), 0, ctx->stream, A->nrows, A->ell_nrb, chunk, (const int *)A->d_rowptr, (const double *)A->d_ell_val, \
                         (const unsigned short *)A->d_ell_c16, (const int *)nullptr, (const int *)A->d_ell_cbase, x, y, a, A->d_blockpart, nl); \
    else \
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_spmv_ell<EPI, false, WW>), dim3(nl), dim3(PMH_BLOCK), 0, ctx->stream, A->nrows, A->ell_nrb, chunk, (const int *)A->d_rowptr, (const double *)A->d_ell_val, \
                         (const unsigned short *)nullptr, (const int *)A->d_ell_col, (const int *)A->d_ell_cbase, x, y, a, A->d_blockpart, nl); \
  } while (0)
    switch (A->ell_w) {
    case 1: ELL_LAUNCH(1); break;
    case 2: ELL_LAUNCH(2); break;
    case 3: ELL_LAUNCH(3); break;
    case 4: ELL_LAUNCH(4); break;
    case 5: ELL_LAUNCH(5); break;
    case 6: ELL_LAUNCH(6); break;
    case 7: ELL_LAUNCH(7); break;
    default: ELL_LAUNCH(8); break;
    }
    PMH_HIP(hipGetLastError());
    return PMH_SUCCESS;
  }
  if (A->kind == PMH_SPMV_STREAM) {
#define ST_LAUNCH(NNZB, MODE, NT) \
  do { \
    if constexpr (RLV == 1 && (((NT)&2) == 0) && NNZB <= 2048) { \
      if (A->d_col16) { \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_spmv_stream<EPI, NNZB, MODE, ((NT)&1) != 0, false, 1, true>), dim3(nl), dim3(PMH_BLOCK), 0, ctx->stream, A->d_rowblocks, A->n_rowblocks, (A->n_rowblocks + 7) / 8, A->d_rowptr, A->d_col, A->d_val, x, y, a, A->d_blockpart, nl, \
                           (const unsigned short *)A->d_col16, (const int *)A->d_cbase); \
        break; \
      } \
    } \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_spmv_stream<EPI, NNZB, MODE, ((NT)&1) != 0, ((NT)&2) != 0, RLV>), dim3(nl), dim3(PMH_BLOCK), 0, ctx->stream, A->d_rowblocks, A->n_rowblocks, (A->n_rowblocks + 7) / 8, A->d_rowptr, A->d_col, A->d_val, x, y, a, A->d_blockpart, nl, \
                       (const unsigned short *)nullptr, (const int *)nullptr); \
  } while (0)
#define ST_MODE(NNZB) \
  if (A->st_rl == 8) { \
    constexpr int RLV = 8; \
    ST_MODE_(NNZB) \
  } else { \
    constexpr int RLV = 1; \
    ST_MODE_(NNZB) \
  }
#define ST_MODE_(NNZB) \
  switch (A->st_mode * 4 + (A->st_nt & 3)) { \
  case 0: ST_LAUNCH(NNZB, 0, 0); break; \
  case 1: ST_LAUNCH(NNZB, 0, 1); break; \
  case 3: ST_LAUNCH(NNZB, 0, 3); break; \
  case 8: ST_LAUNCH(NNZB, 2, 0); break; \
  case 9: ST_LAUNCH(NNZB, 2, 1); break; \
  case 11: ST_LAUNCH(NNZB, 2, 3); break; \
  default: return pmh_set_error(PMH_ERR_ARG, "PMH_SPMV_TUNE: unsupported mode/nt combination"); \
  }
    if (A->st_nnzb == 512) {
      ST_MODE(512)
    } else if (A->st_nnzb == 1024) {
      ST_MODE(1024)
    } else if (A->st_nnzb == 2048) {
      ST_MODE(2048)
    } else {
      ST_MODE(4096)
    }
  } else {
#define VEC_CASE(L) \
  case L: \
    if (A->st_nt) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_spmv_vector<EPI, L, true>), dim3(nl), dim3(PMH_BLOCK), 0, ctx->stream, A->nrows, A->n_rowblocks, nl, A->d_rowptr, A->d_col, A->d_val, x, y, a, A->d_blockpart, nl); \
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_spmv_vector<EPI, L, false>), dim3(nl), dim3(PMH_BLOCK), 0, ctx->stream, A->nrows, A->n_rowblocks, nl, A->d_rowptr, A->d_col, A->d_val, x, y, a, A->d_blockpart, nl); \
    break;
    switch (A->lanes_per_row) {
      VEC_CASE(8)
      VEC_CASE(16)
      VEC_CASE(32)
      VEC_CASE(64)
    default: return pmh_set_error(PMH_ERR_STATE, "bad lanes_per_row %d", A->lanes_per_row);
    }
  }
  PMH_HIP(hipGetLastError());
  return PMH_SUCCESS;
}

static int spmv_dispatch(pmh_csr A, const double *x, double *y, const pmh_spmv_epi &e);

int pmh_csr_spmv_launch(pmh_csr A, const double *x, double *y, const pmh_spmv_epi &e)
{
  const bool timed = A->ev && (size_t)(2 * A->ev_used + 1) < A->ev->size();
  if (timed) {
    PMH_HIP(hipEventRecord((*A->ev)[2 * A->ev_used], A->ctx->stream));
    A->ev_pending = 1;
  }
  int rc = spmv_dispatch(A, x, y, e);
  if (timed) {
    if (A->ev_pending) PMH_HIP(hipEventRecord((*A->ev)[2 * A->ev_used + 1], A->ctx->stream));
    A->ev_pending                = 0;
    (*A->ev_kind)[A->ev_used++] = e.kind;
  }
  return rc;
}

extern "C" int pmh_csr_timing_enable(pmh_csr A, int max_launches)
{
  PMH_ARG(A && max_launches >= 0);
  PMH_HIP(hipStreamSynchronize(A->ctx->stream));
  if (A->ev) {
    for (hipEvent_t e : *A->ev) hipEventDestroy(e);
    delete A->ev;
    delete A->ev_kind;
    A->ev = nullptr, A->ev_kind = nullptr;
  }
  A->ev_used = 0;
  if (max_launches > 0) {
    A->ev      = new std::vector<hipEvent_t>(2 * (size_t)max_launches);
    A->ev_kind = new std::vector<int>((size_t)max_launches, -1);
    for (auto &e : *A->ev) PMH_HIP(hipEventCreate(&e));
  }
  return PMH_SUCCESS;
}

extern "C" int pmh_csr_timing_get(pmh_csr A, int epilogue, int *launches, double *total_ms)
{
  PMH_ARG(A && launches && total_ms);
  *launches = 0, *total_ms = 0.0;
  if (!A->ev) return PMH_SUCCESS;
  PMH_HIP(hipStreamSynchronize(A->ctx->stream));
  std::vector<float> d;
  float              mx = 0.f;
  for (int i = 0; i < A->ev_used; i++) {
    if ((*A->ev_kind)[i] != epilogue) continue;
    float ms = 0.f;
    PMH_HIP(hipEventElapsedTime(&ms, (*A->ev)[2 * i], (*A->ev)[2 * i + 1]));
    d.push_back(ms);
  }
  if (!d.empty()) {
    std::vector<float> srt(d);
    std::sort(srt.begin(), srt.end());
    mx = srt[(size_t)(0.98 * (double)(srt.size() - 1))];
  }
  // launches of a halted speculative chain return at once (no work, no bytes): they are not SpMVs and are
  // left out of the average (anything below a quarter of the 98th-percentile launch -- on an expansion-heavy stretch most launches of a
  // speculative batch are such no-ops; not of the longest one: the first launch of a kernel in a process pays the code-object load)
  for (float ms : d)
    if (ms >= 0.25f * mx) {
      *total_ms += ms;
      (*launches)++;
    }
  return PMH_SUCCESS;
}

static int spmv_dispatch(pmh_csr A, const double *x, double *y, const pmh_spmv_epi &e)
{
  EpiArgs a;
  a.y1 = e.y1;
  a.g  = e.g;
  a.xx = e.xx;
  a.lb = e.lb;
  a.ub = e.ub;
  a.halt = e.halt;
  switch (e.kind) {
  case PMH_EPI_NONE: return launch<PMH_EPI_NONE>(A, x, y, a);
  case PMH_EPI_ADD: return launch<PMH_EPI_ADD>(A, x, y, a);
  case PMH_EPI_SUB: return launch<PMH_EPI_SUB>(A, x, y, a);
  case PMH_EPI_MPGP: {
    PMH_CHK(launch<PMH_EPI_MPGP>(A, x, y, a));
    if (A->ev_pending) { // close the timing bracket before the (separately launched) finalise kernel
      PMH_HIP(hipEventRecord((*A->ev)[2 * A->ev_used + 1], A->ctx->stream));
      A->ev_pending = 0;
    }
    const int ops[3] = {PMH_RED_SUM, PMH_RED_SUM, PMH_RED_MIN};
    return pmh_finalize_partials(A->ctx, A->d_blockpart, A->n_launch_blocks, (A->kind == PMH_SPMV_STREAM && A->st_mode != 0) ? A->n_launch_blocks : A->n_rowblocks, 3, ops, e.scal_base, e.halt);
  }
  }
  return pmh_set_error(PMH_ERR_ARG, "unknown SpMV epilogue %d", e.kind);
}

extern "C" int pmh_csr_mult(pmh_csr A, const double *x, double *y)
{
  PMH_ARG(A && (x || !A->ncols) && (y || !A->nrows));
  PMH_ARG((const void *)x != (const void *)y);
  pmh_spmv_epi e;
  memset(&e, 0, sizeof(e));
  e.kind = PMH_EPI_NONE;
  return pmh_csr_spmv_launch(A, x, y, e);
}

// the chunk sums of A x alone (long-row matrices): the consumer adds them per row in chunk order itself (qppf.hip folds that and the dense
// T'T product into the G0' kernel of the projector)
int pmh_csr_mult_partials(pmh_csr A, const double *x, const int **lrow, const double **part)
{
  PMH_ARG(A && x && lrow && part && A->l_nchunks > 0);
  const bool long_nt = true; // (non-temporal matrix stream of the long-row kernels)
  if (long_nt) hipLaunchKernelGGL(k_spmv_long_part<true>, dim3(A->l_nchunks), dim3(PMH_BLOCK), 0, A->ctx->stream, (const int *)A->d_lchunks, (const int *)A->d_col, (const double *)A->d_val, x, (const int *)nullptr, A->d_lpart);
  else hipLaunchKernelGGL(k_spmv_long_part<false>, dim3(A->l_nchunks), dim3(PMH_BLOCK), 0, A->ctx->stream, (const int *)A->d_lchunks, (const int *)A->d_col, (const double *)A->d_val, x, (const int *)nullptr, A->d_lpart);
  PMH_HIP(hipGetLastError());
  *lrow = A->d_lrow, *part = A->d_lpart;
  return PMH_SUCCESS;
}

int pmh_csr_mult_partials2(pmh_csr A, const double *x, const double *x2, const int **lrow, const double **part, const double **part2)
{
  PMH_ARG(A && x && x2 && lrow && part && part2 && A->l_nchunks > 0);
  if (!A->d_lpart2) PMH_CHK(pmh_malloc(A->ctx, sizeof(double) * (size_t)A->l_nchunks, (void **)&A->d_lpart2));
  const bool long_nt = true; // (non-temporal matrix stream of the long-row kernels)
  if (long_nt) hipLaunchKernelGGL(k_spmv_long_part2<true>, dim3(A->l_nchunks), dim3(PMH_BLOCK), 0, A->ctx->stream, (const int *)A->d_lchunks, (const int *)A->d_col, (const double *)A->d_val, x, x2, A->d_lpart, A->d_lpart2);
  else hipLaunchKernelGGL(k_spmv_long_part2<false>, dim3(A->l_nchunks), dim3(PMH_BLOCK), 0, A->ctx->stream, (const int *)A->d_lchunks, (const int *)A->d_col, (const double *)A->d_val, x, x2, A->d_lpart, A->d_lpart2);
  PMH_HIP(hipGetLastError());
  *lrow = A->d_lrow, *part = A->d_lpart, *part2 = A->d_lpart2;
  return PMH_SUCCESS;
}

// y = M (A x), M m x m dense given transposed on the device (see k_rows_then_dense); tmp: m doubles of device scratch (short-row matrices)
int pmh_csr_mult_then_dense(pmh_csr A, const double *x, const double *Mt, double *tmp, double *y, int norm_slot)
{
  PMH_ARG(A && x && Mt && tmp && y && A->nrows >= 1 && A->nrows <= 4096 && (norm_slot < 0 || (A->nrows <= 64 && norm_slot < PMH_NSCAL)));
  double *nd = norm_slot >= 0 ? A->ctx->d_scal + norm_slot : nullptr, *nh = norm_slot >= 0 ? A->ctx->h_scal + norm_slot : nullptr;
  pmh_ctx      ctx = A->ctx;
  const size_t lds = sizeof(double) * (size_t)A->nrows;
  if (A->l_nchunks) {
    const bool long_nt = true; // (non-temporal matrix stream of the long-row kernels)
    if (long_nt) hipLaunchKernelGGL(k_spmv_long_part<true>, dim3(A->l_nchunks), dim3(PMH_BLOCK), 0, ctx->stream, (const int *)A->d_lchunks, (const int *)A->d_col, (const double *)A->d_val, x, (const int *)nullptr, A->d_lpart);
    else hipLaunchKernelGGL(k_spmv_long_part<false>, dim3(A->l_nchunks), dim3(PMH_BLOCK), 0, ctx->stream, (const int *)A->d_lchunks, (const int *)A->d_col, (const double *)A->d_val, x, (const int *)nullptr, A->d_lpart);
    hipLaunchKernelGGL(k_rows_then_dense, dim3(1), dim3(PMH_BLOCK), lds, ctx->stream, A->nrows, (const int *)A->d_lrow, (const double *)A->d_lpart, Mt, y, nd, nh);
  } else {
    PMH_CHK(pmh_csr_mult(A, x, tmp));
    hipLaunchKernelGGL(k_rows_then_dense, dim3(1), dim3(PMH_BLOCK), lds, ctx->stream, A->nrows, (const int *)nullptr, (const double *)tmp, Mt, y, nd, nh);
  }
  PMH_HIP(hipGetLastError());
  return PMH_SUCCESS;
}

extern "C" int pmh_csr_mult_add(pmh_csr A, const double *x, const double *y1, double *y)
{
  PMH_ARG(A && y1);
  PMH_ARG((const void *)x != (const void *)y);
  pmh_spmv_epi e;
  memset(&e, 0, sizeof(e));
  e.kind = PMH_EPI_ADD;
  e.y1   = y1;
  return pmh_csr_spmv_launch(A, x, y, e);
}

// A' is built once on the host (counting sort; columns of each row of A' ascending = PETSc MatTranspose layout)
static int build_transpose(pmh_csr A)
{
  pmh_ctx             ctx = A->ctx;
  std::vector<int>    rp((size_t)A->nrows + 1), ci((size_t)A->nnz);
  std::vector<double> va((size_t)A->nnz);
  PMH_CHK(pmh_memcpy_d2h(ctx, rp.data(), A->d_rowptr, sizeof(int) * rp.size()));
  PMH_CHK(pmh_memcpy_d2h(ctx, ci.data(), A->d_col, sizeof(int) * ci.size()));
  PMH_CHK(pmh_memcpy_d2h(ctx, va.data(), A->d_val, sizeof(double) * va.size()));
  std::vector<int> trp((size_t)A->ncols + 1, 0), tci((size_t)A->nnz);
  std::vector<double> tva((size_t)A->nnz);
  for (long long k = 0; k < A->nnz; k++) trp[ci[k] + 1]++;
  for (int j = 0; j < A->ncols; j++) trp[j + 1] += trp[j];
  std::vector<int> pos(trp.begin(), trp.end() - 1);
  for (int i = 0; i < A->nrows; i++)
    for (int k = rp[i]; k < rp[i + 1]; k++) {
      int p  = pos[ci[k]]++;
      tci[p] = i;
      tva[p] = va[k];
    }
  return pmh_csr_create(ctx, A->ncols, A->nrows, trp.data(), tci.data(), tva.data(), &A->transpose);
}

// A' handed over by the caller (who built it anyway): A owns it from here on.  At must be the CSR transpose of A with ascending column indices inside every row
// (what build_transpose produces), so that products with it sum in the same order
int pmh_csr_adopt_transpose(pmh_csr A, pmh_csr At)
{
  PMH_ARG(A && At && At->nrows == A->ncols && At->ncols == A->nrows && At->nnz == A->nnz);
  if (A->transpose) pmh_csr_destroy(A->transpose);
  A->transpose = At;
  return PMH_SUCCESS;
}

int pmh_csr_ensure_transpose(pmh_csr A)
{
  PMH_ARG(A);
  if (!A->transpose) PMH_CHK(build_transpose(A));
  return PMH_SUCCESS;
}

extern "C" int pmh_csr_mult_transpose(pmh_csr A, const double *x, double *y)
{
  PMH_ARG(A);
  if (!A->transpose) PMH_CHK(build_transpose(A));
  return pmh_csr_mult(A->transpose, x, y);
}

// MatMultTransposeAdd: y = y1 + A' x (y1 may be y)
extern "C" int pmh_csr_mult_transpose_add(pmh_csr A, const double *x, const double *y1, double *y)
{
  PMH_ARG(A);
  if (!A->transpose) PMH_CHK(build_transpose(A));
  return pmh_csr_mult_add(A->transpose, x, y1, y);
}
