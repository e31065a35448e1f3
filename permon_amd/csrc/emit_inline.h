// Coarse-space emission and in-kernel finalisation (device side), shared by the fused dual-space chain (dualchain.hip) and the MPGP vector kernels (mpgp.hip).
//
// The projector Q = G0' S G0 (src/qppf/interface/qppf.c:454-503) needs a = G0 v for every vector v an operator application starts from.  G0 (m = 6 x #subdomains rows,
// each ~10^4 entries long) is cut into (row, tile of 1024 columns) SEGMENTS: the kernel that WRITES v -- workgroups of 1024 threads, workgroup b owns the entries
// [1024 b, 1024 b + 1024), one per thread -- keeps its fresh values in LDS and sums its own segments (one of the 16 waves per segment, fixed tree), so G0 v needs no launch of its own.  The last workgroup to finish
// (a ticket: one atomic counter, control only) adds the segment sums per row in segment order, applies the small dense S = T'T, and -- for SMALXE -- T and ||T a||^2
// (smalxe.c:247-261).  The same ticket workgroup reduces the block partials of the MPGP scalars (the role of k_finalize).  Every sum has a fixed order: the ticket decides
// WHO adds, never in which order.
//
// What a ticket costs (scripts/micro/ticket.hip, 400 workgroups): the textbook form -- __threadfence, acq_rel atomic -- 31 us, nearly all of it the L2 write-back of the
// release fence in every workgroup.  Here the partial sums are the ONLY data that cross workgroups inside the kernel, so they alone are stored and loaded at agent scope
// (sc1: past the XCD's L2), the counter is a relaxed atomic behind an s_waitcnt: 1.4 us per 100 workgroups -- hence tiles of 1024 entries, not 256; and 16 waves per
// tile, because a segment is a chain of dependent latencies (descriptor -> entries -> LDS -> tree): with 4 waves the ~12 segments of a wave took 20 us.
#pragma once
#include <hip/hip_runtime.h>

#include "pmh_internal.h"

#define PMH_EMIT_TILE 1024              // threads = entries per workgroup of an emitting kernel
#define PMH_EMIT_NW (PMH_EMIT_TILE / 64) // its waves

static __device__ __forceinline__ void   pmh_st_agent(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static __device__ __forceinline__ double pmh_ld_agent(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int CTRL>
static __device__ __forceinline__ double pmh_dpp_mov(double v)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo     = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  hi     = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// the sum of the 64 lanes' values in EVERY lane: butterflies inside a row of 16 lanes on the DPP cross-bar (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror:
// both partners of an exchange add the same two numbers), then the four row sums through scalar registers, (r0 + r1) + (r2 + r3).  MIN: the same exchanges with fmin.
template <int OP = PMH_RED_SUM>
static __device__ __forceinline__ double pmh_wave_all(double v)
{
#define PMH_COMB(a, b) ((OP == PMH_RED_SUM) ? ((a) + (b)) : fmin((a), (b)))
  v = PMH_COMB(v, pmh_dpp_mov<0xB1>(v));
  v = PMH_COMB(v, pmh_dpp_mov<0x4E>(v));
  v = PMH_COMB(v, pmh_dpp_mov<0x141>(v));
  v = PMH_COMB(v, pmh_dpp_mov<0x140>(v));
  const int    lo = __double2loint(v), hi = __double2hiint(v);
  const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
  const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
  const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
  const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
  return PMH_COMB(PMH_COMB(r0, r1), PMH_COMB(r2, r3));
#undef PMH_COMB
}

// a workgroup's partial sum of K quantities: wave tree, the four waves in order; thread 0 stores row k at partials[k ld + blockIdx.x] at agent scope (read by the ticket workgroup)
template <int K>
static __device__ __forceinline__ void pmh_block_partials_agent(const double (&v)[K], const int (&op)[K], double *__restrict__ partials, int ld)
{
  __shared__ double bp[K][PMH_EMIT_NW];
  const int         lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int k = 0; k < K; k++) {
    const double s = (op[k] == PMH_RED_SUM) ? pmh_wave_all<PMH_RED_SUM>(v[k]) : pmh_wave_all<PMH_RED_MIN>(v[k]);
    if (lane == 0) bp[k][wave] = s;
  }
  __syncthreads();
  if (threadIdx.x < K) {
    const int k = threadIdx.x;
    double    r = bp[k][0];
    for (int w = 1; w < nw; w++) r = (op[k] == PMH_RED_SUM) ? (r + bp[k][w]) : fmin(r, bp[k][w]);
    pmh_st_agent(&partials[(size_t)k * ld + blockIdx.x], r);
  }
}

// block partials -> scalars by the calling workgroup: quantity k strided over the threads in index order, the wave tree, the waves in order
static __device__ __forceinline__ void pmh_fin_in_kernel(const pmh_fin_desc &f, double (*red)[PMH_EMIT_NW])
{
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nt = blockDim.x, nw = nt >> 6;
  double    v[PMH_MAX_RED];
#pragma unroll
  for (int k = 0; k < PMH_MAX_RED; k++) v[k] = (f.op[k] == PMH_RED_SUM) ? 0.0 : INFINITY;
  for (int i = t; i < f.nblocks; i += nt) {
#pragma unroll
    for (int k = 0; k < PMH_MAX_RED; k++)
      if (k < f.K) {
        const double p = pmh_ld_agent(&f.partials[(size_t)k * f.ld + i]);
        v[k]           = (f.op[k] == PMH_RED_SUM) ? (v[k] + p) : fmin(v[k], p);
      }
  }
#pragma unroll
  for (int k = 0; k < PMH_MAX_RED; k++)
    if (k < f.K) {
      const double s = (f.op[k] == PMH_RED_SUM) ? pmh_wave_all<PMH_RED_SUM>(v[k]) : pmh_wave_all<PMH_RED_MIN>(v[k]);
      if (lane == 0) red[k][wave] = s;
    }
  __syncthreads();
  if (t < f.K) {
    double r = red[t][0];
    for (int w = 1; w < nw; w++) r = (f.op[t] == PMH_RED_SUM) ? (r + red[t][w]) : fmin(r, red[t][w]);
    f.d_scal[f.slot[t]] = r;
    f.h_scal[f.slot[t]] = r;
  }
  __syncthreads();
}

// a = G0 v from the segment sums, c = S a, T a and its squared norm where asked for -- by the calling (last) workgroup, in ONE round of memory latency: wave w owns the
// rows r = w + 16 q of G0 (q < 4: m <= 64).  It requests everything it will need at once -- the rows' segment sums (agent scope: they come from other XCDs; a row has at
// most one per tile, <= 512) and, lane = column, its rows of S and T' -- adds the sums per row in segment order (lane-strided, wave tree), and forms its share of the two
// small products, sum_q S[r_q][lane] a[r_q].  The 16 shares are added in wave order.
static __device__ __forceinline__ void pmh_emit_finish(const pmh_emit_tab &tab, const pmh_emit_out &o, double (*ps)[64], double (*pt)[64])
{
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, m = tab.m;
  double    pv[4][8], sv[4], tv[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int  r  = wave + PMH_EMIT_NW * q;
    const bool in = r < m;
    const int  c0 = in ? tab.lrow[r] : 0, c1 = in ? tab.lrow[r + 1] : 0;
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int c = c0 + lane + 64 * u;
      pv[q][u]    = (c < c1) ? pmh_ld_agent(&o.part[c]) : 0.0;
    }
    sv[q] = (in && o.S && lane < m) ? o.S[(size_t)r * m + lane] : 0.0;
    tv[q] = (in && o.Tt && lane < m) ? o.Tt[(size_t)r * m + lane] : 0.0;
  }
  double s = 0.0, tt = 0.0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int r = wave + PMH_EMIT_NW * q;
    if (r >= m) break; // uniform
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < 8; u++) acc += pv[q][u];
    acc = pmh_wave_all<PMH_RED_SUM>(acc);
    if (lane == 0 && o.coarse) o.coarse[r] = acc;
    s += sv[q] * acc, tt += tv[q] * acc;
  }
  ps[wave][lane] = s, pt[wave][lane] = tt;
  __syncthreads();
  if (wave == 0) {
    double c = 0.0, y = 0.0;
#pragma unroll
    for (int w = 0; w < PMH_EMIT_NW; w++) c += ps[w][lane], y += pt[w][lane];
    if (o.S && lane < m) o.coarse_c[lane] = c;
    if (o.Tt) {
      if (o.y2 && lane < m) o.y2[lane] = y;
      const double sq = pmh_wave_all<PMH_RED_SUM>(lane < m ? y * y : 0.0);
      if (lane == 0) *o.norm_d = sq, *o.norm_h = sq;
    }
  }
  __syncthreads();
}

// The tail of a kernel of PMH_EMIT_TILE-thread workgroups whose thread t of workgroup b has written entry 1024 b + t of up to two vectors (v0 -> target 0, v1 -> target 1;
// a target with part == nullptr is off; entries past the end of the vector: 0) and, optionally, block partials of scalar reductions (fin.K > 0: stored with
// pmh_block_partials_agent / pmh_st_agent).  Every thread of every workgroup must call it (barriers inside).
static __device__ __forceinline__ void pmh_emit_tail(const pmh_emit_args &ea, const pmh_fin_desc &fin, double v0, double v1)
{
  __shared__ double xs[2][PMH_EMIT_TILE];
  __shared__ double ps[PMH_EMIT_NW][64], pt[PMH_EMIT_NW][64];
  __shared__ double red[PMH_MAX_RED][PMH_EMIT_NW];
  __shared__ int    last;
  const int         t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool        e0 = ea.o[0].part != nullptr, e1 = ea.o[1].part != nullptr;
  if (!e0 && !e1 && fin.K == 0) return;
  if (e0 || e1) {
    xs[0][t] = v0, xs[1][t] = v1;
    __syncthreads();
    // the tile's segments have fixed places in the table (64 per tile, absent ones empty): a wave takes the segments wave, wave + 16, ... -- all four with every
    // load in flight together (a segment is a chain of latencies: descriptor -> entries -> LDS -> tree)
    const int base = (int)blockIdx.x * PMH_EMIT_TILE, s1 = 64;
    for (int sb = wave; sb < s1; sb += 4 * PMH_EMIT_NW) {
      int k0[4], k1[4], pp[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int s = (int)blockIdx.x * 64 + sb + q * PMH_EMIT_NW;
        k0[q] = ea.tab.seg[3 * s], k1[q] = ea.tab.seg[3 * s + 1], pp[q] = ea.tab.seg[3 * s + 2];
      }
      double gv[4][4];
      int    gc[4][4];
#pragma unroll
      for (int q = 0; q < 4; q++)
#pragma unroll
        for (int u = 0; u < 4; u++) { // the first 256 entries of the segment (most segments are shorter)
          const int k = k0[q] + lane + 64 * u;
          gv[q][u]    = (k < k1[q]) ? ea.tab.gval[k] : 0.0;
          gc[q][u]    = (k < k1[q]) ? ea.tab.gcol[k] - base : 0;
        }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (k1[q] <= k0[q]) continue; // uniform: no such segment
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (k0[q] + lane + 64 * u < k1[q]) a0 += gv[q][u] * xs[0][gc[q][u]], a1 += gv[q][u] * xs[1][gc[q][u]];
        for (int k = k0[q] + lane + 256; k < k1[q]; k += 64) { // the rest of a long segment
          const double v = ea.tab.gval[k];
          const int    c = ea.tab.gcol[k] - base;
          a0 += v * xs[0][c], a1 += v * xs[1][c];
        }
        if (e0) {
          a0 = pmh_wave_all<PMH_RED_SUM>(a0);
          if (lane == 0) pmh_st_agent(&ea.o[0].part[pp[q]], a0);
        }
        if (e1) {
          a1 = pmh_wave_all<PMH_RED_SUM>(a1);
          if (lane == 0) pmh_st_agent(&ea.o[1].part[pp[q]], a1);
        }
      }
    }
  }
  // ticket: this workgroup's agent-scope stores have completed, count it in; the last one reads the others' partial sums at agent scope
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (t == 0) {
    const unsigned k = __hip_atomic_fetch_add(ea.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last             = (k == gridDim.x - 1) ? 1 : 0;
    if (last) __hip_atomic_store(ea.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // everybody has drawn: ready for the next launch
  }
  __syncthreads();
  if (!last) return;
  if (fin.K > 0) pmh_fin_in_kernel(fin, red);
  if (e0) pmh_emit_finish(ea.tab, ea.o[0], ps, pt);
  if (e1) pmh_emit_finish(ea.tab, ea.o[1], ps, pt);
}
