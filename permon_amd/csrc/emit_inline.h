// Coarse-space emission (device side), shared by the fused dual-space chain (dualchain.hip) and the MPGP vector kernels (mpgp.hip).
//
// The projector Q = G0' S G0 (src/qppf/interface/qppf.c:454-503) needs a = G0 v for every vector v an operator application starts from.  G0 (m = 6 x
// #subdomains rows, each ~10^4 entries long) is cut into (row, tile of 1024 columns) SEGMENTS.  The kernel that WRITES v -- workgroups of 1024 threads, thread
// t of workgroup b owns entry 1024 b + t -- keeps its fresh values in LDS and sums its own segments (one of the 16 waves per segment, fixed tree): G0 v needs
// no launch of its own.  The kernel that READS a = G0 v (the next one in the stream) adds the segment sums per row in segment order itself, every workgroup for
// itself: ~2500 numbers out of L2.
//
// Built, measured and dropped on the way here (profiles/r05_ticket_micro.txt, docs/LAB_NOTEBOOK.md): a "last workgroup done" ticket that finishes a = G0 v and
// the scalar reductions inside the producing kernel.  The textbook ticket (__threadfence + acq_rel atomic) costs 31 us per 400 workgroups -- the release fence
// writes back the L2 in every workgroup; with agent-scope stores / loads of the partial sums only and a relaxed counter it is 1.4 us per 100 workgroups, but
// the chain store -> counter -> load across XCDs still adds ~10 us of fabric round trips to a 3 us kernel.  The consumer-side sum costs one L2 round trip that
// overlaps with the consumer's own loads.
#pragma once
#include <hip/hip_runtime.h>

#include "pmh_internal.h"

#define PMH_EMIT_TILE 1024               // threads = entries per workgroup of an emitting kernel
#define PMH_EMIT_NW (PMH_EMIT_TILE / 64) // its waves

template <int CTRL>
static __device__ __forceinline__ double pmh_dpp_mov(double v)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo     = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  hi     = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// the sum of the 64 lanes' values in EVERY lane: butterflies inside a row of 16 lanes on the DPP cross-bar (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror,
// row_mirror: both partners of an exchange add the same two numbers), then the four row sums through scalar registers, (r0 + r1) + (r2 + r3).  MIN: the same
// exchanges with fmin.
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

// a workgroup's partial sums of K quantities: wave tree, the waves in order; row k goes to partials[k ld + blockIdx.x] on the device AND in the pinned host
// copy (the host adds the block sums itself after its next wait: no finalising launch; device consumers add them in their preamble, pmh_sum_block_partials)
template <int K>
static __device__ __forceinline__ void pmh_block_partials(const double (&v)[K], const int (&op)[K], double *__restrict__ partials,
                        double *__restrict__ h_partials, int ld)
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
    partials[(size_t)k * ld + blockIdx.x] = r;
    if (h_partials) h_partials[(size_t)k * ld + blockIdx.x] = r;
  }
}

// the sum of one row of block partials (nblocks <= 512) in every lane of the calling wave: lane l adds its entries l, l + 64, ... in that order, then the wave
// tree
static __device__ __forceinline__ double pmh_sum_block_partials(const double *__restrict__ row, int nblocks)
{
  const int lane = threadIdx.x & 63;
  double    p[8];
#pragma unroll
  for (int u = 0; u < 8; u++) p[u] = (lane + 64 * u < nblocks) ? row[lane + 64 * u] : 0.0;
  double acc = 0.0;
#pragma unroll
  for (int u = 0; u < 8; u++) acc += p[u];
  return pmh_wave_all<PMH_RED_SUM>(acc);
}

// Emission by a kernel of PMH_EMIT_TILE-thread workgroups whose thread t of workgroup b writes entry 1024 b + t of up to two vectors (v0 -> target 0, v1 ->
// target 1; a target with part == nullptr is off; entries past the end of the vector: 0): the segment sums of G0 v, in two halves.  pmh_emit_prefetch at the
// START of the kernel requests what does not depend on the values -- the tile's segment descriptors (fixed places in the table: 64 per tile, absent ones empty;
// a wave takes the segments wave, wave + 16, ...) and the first 256 entries of each of the wave's four segments -- so that these latencies run under the
// kernel's own work; pmh_emit_tail at the END puts the fresh values into LDS and sums.  Every thread of every workgroup must call both (barrier inside the
// tail).
struct pmh_emit_regs {
  int    k0[4], k1[4], pp[4], gc[4][4];
  double gv[4][4];
};
static __device__ __forceinline__ void pmh_emit_prefetch(const pmh_emit_args &ea, pmh_emit_regs &R)
{
  if (ea.o[0].part == nullptr && ea.o[1].part == nullptr) return;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), base = (int)blockIdx.x * PMH_EMIT_TILE;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int s = (int)blockIdx.x * 64 + wave + q * PMH_EMIT_NW;
    R.k0[q] = ea.tab.seg[3 * s], R.k1[q] = ea.tab.seg[3 * s + 1], R.pp[q] = ea.tab.seg[3 * s + 2];
  }
#pragma unroll
  for (int q = 0; q < 4; q++)
#pragma unroll
    for (int u = 0; u < 4; u++) { // the first 256 entries of the segment (most segments are shorter)
      const int k = R.k0[q] + lane + 64 * u;
      R.gv[q][u]  = (k < R.k1[q]) ? ea.tab.gval[k] : 0.0;
      R.gc[q][u]  = (k < R.k1[q]) ? ea.tab.gcol[k] - base : 0;
    }
}
static __device__ __forceinline__ void pmh_emit_tail(const pmh_emit_args &ea, const pmh_emit_regs &R, double v0, double v1)
{
  __shared__ double xs[2][PMH_EMIT_TILE];
  const int         t = threadIdx.x, lane = t & 63, base = (int)blockIdx.x * PMH_EMIT_TILE;
  const bool        e0 = ea.o[0].part != nullptr, e1 = ea.o[1].part != nullptr;
  if (!e0 && !e1) return;
  xs[0][t] = v0, xs[1][t] = v1;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; q++) {
    if (R.k1[q] <= R.k0[q]) continue; // uniform: no such segment
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (R.k0[q] + lane + 64 * u < R.k1[q]) a0 += R.gv[q][u] * xs[0][R.gc[q][u]], a1 += R.gv[q][u] * xs[1][R.gc[q][u]];
    for (int k = R.k0[q] + lane + 256; k < R.k1[q]; k += 64) { // the rest of a long segment
      const double v = ea.tab.gval[k];
      const int    c = ea.tab.gcol[k] - base;
      a0 += v * xs[0][c], a1 += v * xs[1][c];
    }
    if (e0) {
      a0 = pmh_wave_all<PMH_RED_SUM>(a0);
      if (lane == 0) ea.o[0].part[R.pp[q]] = a0;
    }
    if (e1) {
      a1 = pmh_wave_all<PMH_RED_SUM>(a1);
      if (lane == 0) ea.o[1].part[R.pp[q]] = a1;
    }
  }
}

// Consumer side: a = G0 v from the segment sums and the share of the calling wave in M a for up to two m x m matrices given by rows (lane = column).  Wave w of
// nw owns the rows r = w + nw q of G0 (q < RPW: 4 for 16 waves, 16 for 4 waves; m <= 64); a row has at most one segment sum per tile (<= 512), added
// lane-strided in segment order, then the wave tree.  Returns, per lane (= column), s1 = sum_q M1[r_q][lane] a[r_q] and s2 likewise for M2 (0 where the pointer
// is null); a[r] also goes to a_out[r] (LDS or global; may be null).  The caller adds the waves' shares in wave order.  NU: 64 NU >= the number of tiles (2
// serves up to 131 072 dual entries with a quarter of the registers of 8).
template <int RPW, int NU = 8>
static __device__ __forceinline__ void pmh_coarse_share(const pmh_emit_tab &tab, const double *__restrict__ part, const double *__restrict__ M1,
                        const double *__restrict__ M2, double *a_out,
                                                       double &s1, double &s2)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6, m = tab.m;
  s1 = 0.0, s2 = 0.0;
#pragma unroll 1
  for (int qb = 0; qb < RPW; qb += 4) { // four rows at a time: their loads in flight together
    if (wave + nw * qb >= m) break;     // uniform
    double pv[4][NU], m1[4], m2[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int  r  = wave + nw * (qb + q);
      const bool in = r < m;
      const int  c0 = in ? tab.lrow[r] : 0, c1 = in ? tab.lrow[r + 1] : 0;
#pragma unroll
      for (int u = 0; u < NU; u++) {
        const int c = c0 + lane + 64 * u;
        pv[q][u]    = (c < c1) ? part[c] : 0.0;
      }
      m1[q] = (in && M1 && lane < m) ? M1[(size_t)r * m + lane] : 0.0;
      m2[q] = (in && M2 && lane < m) ? M2[(size_t)r * m + lane] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int r = wave + nw * (qb + q);
      if (r >= m) break; // uniform
      double acc = 0.0;
#pragma unroll
      for (int u = 0; u < NU; u++) acc += pv[q][u];
      acc = pmh_wave_all<PMH_RED_SUM>(acc);
      if (a_out && lane == 0) a_out[r] = acc;
      s1 += m1[q] * acc, s2 += m2[q] * acc;
    }
  }
}
