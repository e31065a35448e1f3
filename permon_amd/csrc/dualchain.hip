// The dual-space chain of one application of the penalised, projected FETI operator
//
//     y = rho Q x + P F P x,   Q = G0' S G0,  P = I - Q,  F = Bs (middle) Bg'        (S = T'T = (G0 G0')^{-1}: implicit orthonormalisation)
//
// (MatMult_Penalized src/qp/utils/matpenalized.c:12-22 over MatCreateProd(P, F, P) src/qp/interface/qptransform.c:273-284 with QPPFApplyQ / QPPFApplyP
// src/qppf/interface/qppf.c:454-503,563-575 and MatMult(Transpose)_Gluing src/mat/impls/gluing/gluing.c:47-159) in FIVE launches, two of them the middle
// stage's:
//
//     [the kernel that wrote x left the segment sums of G0 x behind (emit_inline.h; k_dc_emit where nobody did)]
//     k_dc_gather    a = G0 x from the segment sums, c = S a;  mid_in = Bg' (x - G0' c): the projection is recomputed per gathered entry, P x is never stored
//     middle stage   mid_out = W mid_in               (orbit GEMM + its finishing launch, or the inner Krylov solve)
//     k_dc_scatter   w = Bs mid_out and the segment sums of G0 w, stored behind w ([w; sums] is ONE all-reduce on several GPUs)
//     k_dc_final     d = G0 w from the sums, e = S d;  y = rho G0' c + (w - G0' e);  the MPGP vector phase (mpgp.c:537-544 / :578-615) with its block partials
//                    left for the host's next wait, and the segment sums of G0 p for the p = gf it wrote
//
// Round 4 took 9 launches + a finalising one for the same product (k_spmv_long_part, k_gt_fused1d<0>, k_spmv_ell, GEMM, fin, k_spmv_stream, k_spmv_long_part,
// k_gt_fused1d<EPI>, k_finalize): at 5-9 us each they were a quarter of the one-GPU step and what bounded the share of one of eight GPUs.
// Determinism: every sum has a fixed order (lane-strided in segment order -> wave tree -> waves in order); no atomics anywhere.
#include <algorithm>
#include <cmath>

#include "dualchain.h"
#include "emit_inline.h"
#include "box_inline.h"
#include "reduce.h"

struct pmh_dualchain_s {
  pmh_ctx  ctx;
  pmh_qppf pf;
  pmh_op   F;
  int      n, m, nwg, nseg, W; // nwg: tiles of PMH_EMIT_TILE dual entries; W: slots per row of the G0' copy
  pmh_csr  gather = nullptr, scatter = nullptr;
  double  *mid_in = nullptr;
  const double *mid_out = nullptr;
  int     *d_seg = nullptr, *d_lrow = nullptr;
  double        *d_ev = nullptr; // G0' as a slot copy: [n][W] values and one-byte columns
  unsigned char *d_ec = nullptr;
  // the rows of the gather matrix that have entries (built for the gather matrix / mid_in pair in use: the other rows of mid_in are zeroed once)
  pmh_csr  list_for = nullptr;
  unsigned long long list_uid = 0, sc_uid = 0; // (the matrices' uids: an operator rebuilt at the same address is another matrix)
  double  *zeroed   = nullptr;
  int      nlist = 0, *d_reci = nullptr; // per listed row 8 ints (row, partner row or -1, first entry, one past the last, the first three columns, 0) ...
  double  *d_recd = nullptr;            // ... and 4 doubles (the first three values, 0)
  // the scatter matrix with its first two entries per row at fixed places ([n][2] columns / values, entry count per row; longer rows finish from the CSR)
  pmh_csr  sc_for = nullptr;
  int     *d_sc2 = nullptr;
  double  *d_sv2 = nullptr;
  unsigned char *d_scnt = nullptr;
  double  *part[3] = {nullptr, nullptr, nullptr}; // segment sums: iterate, direction, scratch
  double  *c_cur   = nullptr;                     // [64] c = S G0 x of the application under way (written by k_dc_gather, read by k_dc_final)
  double  *w       = nullptr;                     // [n + nseg]: Bs mid_out, then the segment sums of G0 w
  // SMALXE's ||B u||: T G0 u and its squared norm ride on the gather kernel of the application that follows an emission of the iterate
  double  *normGu    = nullptr;
  int      norm_slot = -1;
  const double *x_emitted = nullptr; // the iterate whose segment sums target 0 holds
  bool     norm_done = false;        // ... and whether its norm has been formed (enqueued)
  const double *norm_ptr = nullptr;  // the vector whose T G0 u / squared norm are in normGu / the slot (as far as the stream has got)
  int      launches = 0;
};

// ---- kernels ---------------------------------------------------------------------------------------------------------------------
// All of them are made of memory latencies, not of bytes (the dual space has ~10^5 entries): what counts is the number of DEPENDENT load levels and that every
// wave is resident at once.  G0' is therefore kept as a fixed-width slot copy (W = 8 or 16 entries per row, one byte per column: m <= 64) addressed by the row
// number alone, the gather visits only the rows of Bg' that have entries (the others are zero once and for all) through one record per row, the scatter matrix
// keeps its first two entries per row at fixed places.

// G0 x for a vector nobody emitted (the first product of a solve, plain MatMult callers)
__global__ __launch_bounds__(PMH_EMIT_TILE) void k_dc_emit(int n, const double *__restrict__ x, pmh_emit_args ea)
{
  pmh_emit_regs R;
  pmh_emit_prefetch(ea, R);
  const int r = blockIdx.x * PMH_EMIT_TILE + threadIdx.x;
  pmh_emit_tail(ea, R, r < n ? x[r] : 0.0, 0.0);
}

// T G0 u and ||T G0 u||^2 from the segment sums of the iterate, on their own (one workgroup): where no application follows the emission before the host waits
template <int NU>
__global__ __launch_bounds__(PMH_EMIT_TILE) void k_dc_norm(pmh_emit_tab tab, const double *__restrict__ part, const double *__restrict__ Tt,
                        double *__restrict__ y2, double *__restrict__ norm_d,
                                                         double *__restrict__ norm_h)
{
  __shared__ double pt[PMH_EMIT_NW][64];
  const int         lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = tab.m;
  double            s1, s2;
  pmh_coarse_share<4, NU>(tab, part, Tt, nullptr, nullptr, s1, s2);
  pt[wave][lane] = s1;
  __syncthreads();
  if (wave == 0) {
    double y = 0.0;
#pragma unroll
    for (int w = 0; w < PMH_EMIT_NW; w++) y += pt[w][lane];
    if (lane < m) y2[lane] = y;
    const double sq = pmh_wave_all<PMH_RED_SUM>(lane < m ? y * y : 0.0);
    if (lane == 0) *norm_d = sq, *norm_h = sq;
  }
}

// (G0' c)_j over the slot copy: left to right as the CSR row (padded slots hold 0 x c[0])
template <int W>
static __device__ __forceinline__ void dc_gt_load(const double *__restrict__ ev, const unsigned char *__restrict__ ec, int j, double (&v)[W],
                        unsigned char (&c)[W])
{
  typedef double dbl2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int e = 0; e < W; e += 2) {
    const dbl2 d = *(const dbl2 *)(ev + (size_t)j * W + e);
    v[e] = d.x, v[e + 1] = d.y;
  }
#pragma unroll
  for (int h = 0; h < W / 8; h++) {
    const unsigned long long cc = *(const unsigned long long *)(ec + (size_t)j * W + 8 * h);
#pragma unroll
    for (int e = 0; e < 8; e++) c[8 * h + e] = (unsigned char)(cc >> (8 * e));
  }
}

// a = G0 x from the segment sums, c = S a (every workgroup for itself, overlapping its record loads; workgroup 0 leaves c for k_dc_final, the last workgroup
// forms SMALXE's T G0 u and ||T G0 u||^2 where an emission of the iterate is waiting for it).  Then mid_in[r] = sum_k Bg'[r][k] (x - G0' c)[col k] for the rows
// of Bg' that have entries, each sum left to right as the plain loops take it: row r as MatMult_SeqAIJ sums it, (P x)_j = -1 (G0' c)_j + x_j as QPPFApplyP's
// VecAYPX forms it (qppf.c:563-575).  One record per listed row: the row, its first three entries (more: the rest from the CSR) and -- where another row holds
// the same entries with the opposite signs (the +x / -x copies of the orbit GEMM's multivector) -- that partner row, which gets the negated sum (exactly what
// its own left-to-right sum would be).
struct dc_norm_args {
  const double *part, *Tt; // the iterate's segment sums, T' (nullptr: nothing to do)
  double       *y2, *norm_d, *norm_h;
};
template <int W, int NU>
__global__ __launch_bounds__(PMH_EMIT_TILE) void k_dc_gather(int nlist, const int *__restrict__ reci, const double *__restrict__ recd,
                        const int *__restrict__ bcol, const double *__restrict__ bval,
                                                        const double *__restrict__ ev, const unsigned char *__restrict__ ec, pmh_emit_tab tab, const double *__restrict__ part,
                                                        const double *__restrict__ S, double *__restrict__ c_out, dc_norm_args na, const double *__restrict__ x, double *__restrict__ mid)
{
  typedef int    int4v __attribute__((ext_vector_type(4)));
  typedef double dbl2 __attribute__((ext_vector_type(2)));
  __shared__ double cs[64], ps[PMH_EMIT_NW][64], pt[PMH_EMIT_NW][64];
  const int         t = threadIdx.x, lane = t & 63, wave = t >> 6, i = blockIdx.x * PMH_EMIT_TILE + t, m = tab.m;
  int4v             ra = {0, -1, 0, 0}, rb = {0, 0, 0, 0};
  dbl2              da = {0.0, 0.0}, db = {0.0, 0.0};
  if (i < nlist) {
    ra = *(const int4v *)(reci + (size_t)8 * i), rb = *(const int4v *)(reci + (size_t)8 * i + 4);
    da = *(const dbl2 *)(recd + (size_t)4 * i), db = *(const dbl2 *)(recd + (size_t)4 * i + 2);
  }
  const int    r = ra.x, r2 = ra.y, k0 = ra.z, k1 = ra.w, cnt = (i < nlist) ? k1 - k0 : 0;
  const int    j[3]  = {rb.x, rb.y, rb.z};
  const double bv[3] = {da.x, da.y, db.x};
  double        xv[3], v0[W];
  unsigned char c0[W];
#pragma unroll
  for (int u = 0; u < 3; u++) xv[u] = (u < cnt) ? x[j[u]] : 0.0;
#pragma unroll
  for (int e = 0; e < W; e++) v0[e] = 0.0, c0[e] = 0;
  if (cnt > 0) dc_gt_load<W>(ev, ec, j[0], v0, c0); // most rows have ONE entry: its row of G0' travels with the coarse sums
  // uniform: the extra workgroup forms SMALXE's T G0 u and ||T G0 u||^2 (it has no rows of its own: i >= nlist)
  if (na.part && blockIdx.x == gridDim.x - 1) {
    double s1, s2;
    pmh_coarse_share<4, NU>(tab, na.part, na.Tt, nullptr, nullptr, s1, s2);
    pt[wave][lane] = s1;
    __syncthreads();
    if (wave == 0) {
      double y = 0.0;
#pragma unroll
      for (int w = 0; w < PMH_EMIT_NW; w++) y += pt[w][lane];
      if (lane < m) na.y2[lane] = y;
      const double sq = pmh_wave_all<PMH_RED_SUM>(lane < m ? y * y : 0.0);
      if (lane == 0) *na.norm_d = sq, *na.norm_h = sq;
    }
    return;
  }
  {
    double s1, s2;
    pmh_coarse_share<4, NU>(tab, part, S, nullptr, nullptr, s1, s2);
    ps[wave][lane] = s1;
    __syncthreads();
    if (wave == 0) {
      double cc = 0.0;
#pragma unroll
      for (int w = 0; w < PMH_EMIT_NW; w++) cc += ps[w][lane];
      cs[lane] = cc;
      if (blockIdx.x == 0 && lane < m) c_out[lane] = cc;
    }
    __syncthreads();
  }
  if (i >= nlist) return;
  double sum = 0.0;
  {
    double q = 0.0;
#pragma unroll
    for (int e = 0; e < W; e++) q += v0[e] * cs[c0[e]];
    sum += bv[0] * (-1.0 * q + xv[0]);
  }
#pragma unroll
  for (int u = 1; u < 3; u++)
    if (u < cnt) {
      double        vv[W];
      unsigned char cc[W];
      dc_gt_load<W>(ev, ec, j[u], vv, cc);
      double q = 0.0;
#pragma unroll
      for (int e = 0; e < W; e++) q += vv[e] * cs[cc[e]];
      sum += bv[u] * (-1.0 * q + xv[u]);
    }
  for (int k = k0 + 3; k < k1; k++) { // rows with more than three entries (corner dofs of a redundant gluing)
    const int     jj = bcol[k];
    double        vv[W];
    unsigned char cc[W];
    dc_gt_load<W>(ev, ec, jj, vv, cc);
    double q = 0.0;
#pragma unroll
    for (int e = 0; e < W; e++) q += vv[e] * cs[cc[e]];
    sum += bval[k] * (-1.0 * q + x[jj]);
  }
  mid[r] = sum;
  if (r2 >= 0) mid[r2] = -sum;
}

// w = Bs mid_out (row j left to right: MatMultTranspose_Gluing's accumulation order, gluing.c:142-150) and the segment sums of G0 w, stored behind w
__global__ __launch_bounds__(PMH_EMIT_TILE) void k_dc_scatter(int n, const unsigned char *__restrict__ scnt, const int *__restrict__ sc2,
                        const double *__restrict__ sv2, const int *__restrict__ srp,
                                                            const int *__restrict__ scol, const double *__restrict__ sval, const double *__restrict__ Y, double *__restrict__ w, pmh_emit_args ea)
{
  pmh_emit_regs R;
  pmh_emit_prefetch(ea, R);
  const int r   = blockIdx.x * PMH_EMIT_TILE + threadIdx.x;
  double    sum = 0.0;
  if (r < n) {
    typedef int    int2v __attribute__((ext_vector_type(2)));
    typedef double dbl2 __attribute__((ext_vector_type(2)));
    const int   cnt = scnt[r];
    const int2v c   = *(const int2v *)(sc2 + (size_t)2 * r);
    const dbl2  v   = *(const dbl2 *)(sv2 + (size_t)2 * r);
    const double y0 = cnt > 0 ? Y[c.x] : 0.0, y1 = cnt > 1 ? Y[c.y] : 0.0;
    if (cnt > 0) sum += v.x * y0;
    if (cnt > 1) sum += v.y * y1;
    if (cnt > 2) { // (rows of a gluing with more than two leaves)
      const int k0 = srp[r], k1 = srp[r + 1];
      for (int kk = k0 + 2; kk < k1; kk++) sum += sval[kk] * Y[scol[kk]];
    }
    w[r] = sum;
  }
  pmh_emit_tail(ea, R, sum, 0.0);
}

// d = G0 w from the segment sums behind w, e = S d; out = rho (G0' c)_j + (w_j - (G0' e)_j) -- VecAYPX, VecScale, VecAXPY of matpenalized.c:12-22 per entry --
// and the MPGP vector phase: its block partials go to the device rows and to the pinned host copy, the segment sums of G0 p to the direction's target
template <int EPI, int W, int NU>
__global__ __launch_bounds__(PMH_EMIT_TILE) void k_dc_final(int n, const double *__restrict__ ev, const unsigned char *__restrict__ ec, pmh_emit_tab tab,
                        const double *__restrict__ S,
                                                          const double *__restrict__ cin, const double *__restrict__ w, double rho, double *__restrict__ y, pmh_vec_epi epi,
                                                          const double *__restrict__ pin, pmh_emit_args ea)
{
  __shared__ double cs[64], es[64], ps[PMH_EMIT_NW][64];
  const int         t = threadIdx.x, lane = t & 63, wave = t >> 6, m = tab.m, r = blockIdx.x * PMH_EMIT_TILE + t;
  // the row's own operands travel with the coarse sums
  double        v[W];
  unsigned char c[W];
  double        wr = 0.0, pi = 0.0, gq = 0.0, xq = 0.0, lq = -INFINITY, uq = INFINITY, bq = 0.0;
  pmh_emit_regs R;
  if (EPI == PMH_VEPI_GRAD_SPLIT) pmh_emit_prefetch(ea, R);
#pragma unroll
  for (int e = 0; e < W; e++) v[e] = 0.0, c[e] = 0;
  if (r < n) {
    dc_gt_load<W>(ev, ec, r, v, c);
    wr = w[r];
    if (EPI != 0) pi = pin[r];
    if (EPI == PMH_VEPI_P1) {
      gq = epi.g[r], xq = epi.xx[r];
      if (epi.lb) lq = epi.lb[r];
      if (epi.ub) uq = epi.ub[r];
    }
    if (EPI == PMH_VEPI_GRAD_SPLIT) bq = epi.b[r];
  }
  if (t < m) cs[t] = cin[t];
  {
    double s1, s2;
    pmh_coarse_share<4, NU>(tab, w + n, S, nullptr, nullptr, s1, s2);
    ps[wave][lane] = s1;
    __syncthreads();
    if (wave == 0) {
      double ee = 0.0;
#pragma unroll
      for (int q = 0; q < PMH_EMIT_NW; q++) ee += ps[q][lane];
      es[lane] = ee;
    }
    __syncthreads();
  }
  double pv = 0.0, s0 = 0.0, s1 = 0.0, mn = INFINITY, acc[4] = {0.0, 0.0, 0.0, 0.0};
  if (r < n) {
    double sumc = 0.0, sume = 0.0;
#pragma unroll
    for (int e = 0; e < W; e++) sumc += v[e] * cs[c[e]], sume += v[e] * es[c[e]];
    const double tt  = wr + -1.0 * sume;
    const double out = sumc * rho + tt;
    if (EPI != PMH_VEPI_GRAD_SPLIT) y[r] = out;
    if (EPI == PMH_VEPI_P1) { // k_p1_dots (mpgp.hip): p'Ap, g'p, QPCFeas -- out is Ap[r], pin the operator's input p
      s0 += pi * out;
      s1 += gq * pi;
      mn = pmh_box_feas_v(mn, xq, pi, lq, uq);
    }
    if (EPI == PMH_VEPI_GRAD_SPLIT) { // g = A x - b, the gradient split, p = gf, the partials of (0, |gP|^2, |gc|^2, |gf|^2) -- pin is the iterate
      double gi = out;
      gi += -1.0 * bq;
      y[r] = gi;
      double f, cc;
      pmh_box_split(pi, gi, epi.lb, epi.ub, r, epi.astol, f, cc);
      epi.gf[r]        = f;
      epi.p[r]         = f;
      pv               = f;
      const double gPi = f + cc;
      acc[1] += gPi * gPi;
      acc[2] += cc * cc;
      acc[3] += f * f;
    }
  }
  if (EPI == PMH_VEPI_P1) {
    const double q[3]  = {s0, s1, mn};
    const int    op[3] = {PMH_RED_SUM, PMH_RED_SUM, PMH_RED_MIN};
    pmh_block_partials<3>(q, op, epi.partials + (size_t)epi.prow * epi.ld, epi.h_partials ? epi.h_partials + (size_t)epi.prow * epi.ld : nullptr, epi.ld);
  }
  if (EPI == PMH_VEPI_GRAD_SPLIT) {
    const int op[4] = {PMH_RED_SUM, PMH_RED_SUM, PMH_RED_SUM, PMH_RED_SUM};
    pmh_block_partials<4>(acc, op, epi.partials + (size_t)epi.prow * epi.ld, epi.h_partials ? epi.h_partials + (size_t)epi.prow * epi.ld : nullptr, epi.ld);
    pmh_emit_tail(ea, R, 0.0, pv);
  }
}

// ---- set-up ------------------------------------------------------------------------------------------------------------------------
static bool dc_off() { return pmh_knobs().chain == 0; } // A/B (pmh_set_knob("chain", 0), PMH_NO_CHAIN): the round-4 launch sequence

void pmh_dc_destroy(pmh_dualchain dc);
int pmh_dc_create(pmh_qppf pf, pmh_op F, pmh_dualchain *out)
{
  *out = nullptr;
  if (dc_off() || !pf || !F || !pf->implicit_orth || pf->m < 1 || pf->m > 64) return PMH_EPI_UNSUPPORTED;
  const int n = pf->n, m = pf->m;
  if (n < 1 || n > PMH_MAX_VEC_BLOCKS * PMH_BLOCK) return PMH_EPI_UNSUPPORTED;
  pmh_csr       gather = nullptr, scatter = nullptr;
  double       *mid_in = nullptr;
  const double *mid_out = nullptr;
  if (F->stages(&gather, &mid_in, &scatter, &mid_out) == PMH_EPI_UNSUPPORTED) return PMH_EPI_UNSUPPORTED;
  if (!gather || !scatter || !mid_in || !mid_out || gather->ncols != n || scatter->nrows != n) return PMH_EPI_UNSUPPORTED;
  pmh_ctx ctx = pf->ctx;
  // G0 in (row, tile of PMH_EMIT_TILE columns) segments
  const pmh_csr       G = pf->G;
  std::vector<int>    rp((size_t)m + 1), ci((size_t)G->nnz);
  std::vector<double> va((size_t)G->nnz);
  PMH_CHK(pmh_memcpy_d2h(ctx, rp.data(), G->d_rowptr, sizeof(int) * rp.size()));
  PMH_CHK(pmh_memcpy_d2h(ctx, ci.data(), G->d_col, sizeof(int) * ci.size()));
  PMH_CHK(pmh_memcpy_d2h(ctx, va.data(), G->d_val, sizeof(double) * va.size()));
  const int                     nwg = (n + PMH_EMIT_TILE - 1) / PMH_EMIT_TILE;
  std::vector<std::vector<int>> of_block((size_t)nwg); // (k0, k1, index inside the row, row) per tile, rows ascending
  std::vector<int>              lrow((size_t)m + 1, 0), cnt_col((size_t)n, 0);
  for (int r = 0; r < m; r++)
    for (int k = rp[r] + 1; k < rp[r + 1]; k++)
      if (ci[k] <= ci[k - 1]) return PMH_EPI_UNSUPPORTED; // columns must ascend inside a row (no duplicates): a segment then has at most PMH_EMIT_TILE entries
  for (int r = 0; r < m; r++) {
    int k = rp[r], cnt = 0;
    while (k < rp[r + 1]) {
      const int b = ci[k] / PMH_EMIT_TILE;
      int       e = k + 1;
      while (e < rp[r + 1] && ci[e] / PMH_EMIT_TILE == b) e++;
      of_block[b].push_back(k), of_block[b].push_back(e), of_block[b].push_back(cnt), of_block[b].push_back(r);
      cnt++, k = e;
    }
    lrow[r + 1] = cnt; // counts; prefix sums below
    for (int kk = rp[r]; kk < rp[r + 1]; kk++) cnt_col[ci[kk]]++;
  }
  for (int r = 0; r < m; r++) lrow[r + 1] += lrow[r];
  std::vector<int> seg((size_t)nwg * 64 * 3, 0); // fixed places: tile b, slot i (its i-th segment; absent: k0 = k1 = 0)
  int              nseg = 0;
  for (int b = 0; b < nwg; b++) {
    const std::vector<int> &v = of_block[b];
    for (size_t i = 0; i < v.size(); i += 4, nseg++) {
      const size_t o = ((size_t)b * 64 + i / 4) * 3;
      seg[o] = v[i], seg[o + 1] = v[i + 1], seg[o + 2] = lrow[v[i + 3]] + v[i + 2];
    }
  }
  if (nseg != lrow[m]) return pmh_set_error(PMH_ERR_STATE, "pmh_dc_create: segment count mismatch");
  // G0' as a slot copy: row j = the entries of column j of G0, rows ascending (= the CSR row of the transpose, whose order every sum keeps)
  int wmax = 0;
  for (int j = 0; j < n; j++) wmax = std::max(wmax, cnt_col[j]);
  if (wmax > 16) return PMH_EPI_UNSUPPORTED;
  const int                  W = wmax <= 8 ? 8 : 16;
  std::vector<double>        ev((size_t)n * W, 0.0);
  std::vector<unsigned char> ec((size_t)n * W, 0);
  std::fill(cnt_col.begin(), cnt_col.end(), 0);
  for (int r = 0; r < m; r++)
    for (int k = rp[r]; k < rp[r + 1]; k++) {
      const int j = ci[k], e = cnt_col[j]++;
      ev[(size_t)j * W + e] = va[k], ec[(size_t)j * W + e] = (unsigned char)r;
    }
  pmh_dualchain dc = new pmh_dualchain_s();
#define DC_CHK(call) \
  do { \
    if (int rc_ = (call)) { \
      pmh_dc_destroy(dc); \
      return rc_; \
    } \
  } while (0)
  dc->ctx = ctx, dc->pf = pf, dc->F = F, dc->n = n, dc->m = m, dc->nwg = nwg, dc->nseg = nseg, dc->W = W;
  DC_CHK(pmh_malloc(ctx, sizeof(int) * (seg.size() + 3), (void **)&dc->d_seg));
  DC_CHK(pmh_malloc(ctx, sizeof(int) * lrow.size(), (void **)&dc->d_lrow));
  DC_CHK(pmh_memcpy_h2d(ctx, dc->d_seg, seg.data(), sizeof(int) * seg.size()));
  DC_CHK(pmh_memcpy_h2d(ctx, dc->d_lrow, lrow.data(), sizeof(int) * lrow.size()));
  DC_CHK(pmh_malloc(ctx, sizeof(double) * ev.size(), (void **)&dc->d_ev));
  DC_CHK(pmh_malloc(ctx, ec.size(), (void **)&dc->d_ec));
  DC_CHK(pmh_memcpy_h2d(ctx, dc->d_ev, ev.data(), sizeof(double) * ev.size()));
  DC_CHK(pmh_memcpy_h2d(ctx, dc->d_ec, ec.data(), ec.size()));
  for (int s = 0; s < 3; s++) {
    DC_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)(nseg + 1), (void **)&dc->part[s]));
    DC_CHK(pmh_memset(ctx, dc->part[s], 0, sizeof(double) * (size_t)(nseg + 1)));
  }
  DC_CHK(pmh_malloc(ctx, sizeof(double) * 64, (void **)&dc->c_cur));
  DC_CHK(pmh_memset(ctx, dc->c_cur, 0, sizeof(double) * 64));
  DC_CHK(pmh_malloc(ctx, sizeof(double) * ((size_t)n + nseg + 1), (void **)&dc->w));
  DC_CHK(pmh_memset(ctx, dc->w, 0, sizeof(double) * ((size_t)n + nseg + 1)));
#undef DC_CHK
  *out = dc;
  return PMH_SUCCESS;
}

// the records of the gather matrix's rows that have entries (the rest of mid_in is zeroed here, once: nothing but an SpMV with the same matrix ever writes
// there -- zeros again) and the fixed-place copy of the scatter matrix
static int dc_prepare_stages(pmh_dualchain dc)
{
  pmh_ctx ctx = dc->ctx;
  if (!(dc->list_for == dc->gather && dc->list_uid == dc->gather->uid && dc->zeroed == dc->mid_in)) {
    const pmh_csr       Bg = dc->gather;
    std::vector<int>    rp((size_t)Bg->nrows + 1), ci((size_t)Bg->nnz);
    std::vector<double> va((size_t)Bg->nnz);
    PMH_CHK(pmh_memcpy_d2h(ctx, rp.data(), Bg->d_rowptr, sizeof(int) * rp.size()));
    PMH_CHK(pmh_memcpy_d2h(ctx, ci.data(), Bg->d_col, sizeof(int) * ci.size()));
    PMH_CHK(pmh_memcpy_d2h(ctx, va.data(), Bg->d_val, sizeof(double) * va.size()));
    // pairs of rows with the same columns and opposite values: hash on (columns, |values|), verify entry by entry
    auto key = [&](int r) {
      unsigned long long h = 1469598103934665603ULL;
      for (int k = rp[r]; k < rp[r + 1]; k++) {
        unsigned long long b;
        const double       a = fabs(va[k]);
        memcpy(&b, &a, 8);
        h = (h ^ (unsigned long long)(unsigned)ci[k]) * 1099511628211ULL;
        h = (h ^ b) * 1099511628211ULL;
      }
      return h;
    };
    auto opposite = [&](int r, int q) {
      if (rp[r + 1] - rp[r] != rp[q + 1] - rp[q]) return false;
      for (int k = rp[r], l = rp[q]; k < rp[r + 1]; k++, l++)
        if (ci[k] != ci[l] || va[k] != -va[l]) return false;
      return true;
    };
    std::vector<int> partner((size_t)Bg->nrows, -1); // -1 none, -2 is somebody's partner
    {
      std::vector<std::pair<unsigned long long, int>> keyed;
      for (int r = 0; r < Bg->nrows; r++)
        if (rp[r + 1] > rp[r]) keyed.push_back({key(r), r});
      std::sort(keyed.begin(), keyed.end());
      for (size_t a = 0; a < keyed.size(); a++) {
        const int r = keyed[a].second;
        if (partner[r] != -1) continue;
        for (size_t b = a + 1; b < keyed.size() && keyed[b].first == keyed[a].first; b++) {
          const int q = keyed[b].second;
          if (partner[q] == -1 && opposite(r, q)) {
            partner[r] = q, partner[q] = -2;
            break;
          }
        }
      }
    }
    std::vector<int>    reci;
    std::vector<double> recd;
    for (int r = 0; r < Bg->nrows; r++) {
      if (rp[r + 1] == rp[r] || partner[r] == -2) continue;
      const int k0 = rp[r], k1 = rp[r + 1];
      int       j[3] = {0, 0, 0};
      double    b[3] = {0.0, 0.0, 0.0};
      for (int u = 0; u < 3 && k0 + u < k1; u++) j[u] = ci[k0 + u], b[u] = va[k0 + u];
      const int ints[8] = {r, partner[r] >= 0 ? partner[r] : -1, k0, k1, j[0], j[1], j[2], 0};
      reci.insert(reci.end(), ints, ints + 8);
      recd.push_back(b[0]), recd.push_back(b[1]), recd.push_back(b[2]), recd.push_back(0.0);
    }
    if (dc->d_reci) {
      PMH_CHK(pmh_free(ctx, dc->d_reci));
      PMH_CHK(pmh_free(ctx, dc->d_recd));
    }
    dc->d_reci = nullptr, dc->d_recd = nullptr;
    PMH_CHK(pmh_malloc(ctx, sizeof(int) * (reci.size() + 8), (void **)&dc->d_reci));
    PMH_CHK(pmh_malloc(ctx, sizeof(double) * (recd.size() + 4), (void **)&dc->d_recd));
    PMH_CHK(pmh_memcpy_h2d(ctx, dc->d_reci, reci.data(), sizeof(int) * reci.size()));
    PMH_CHK(pmh_memcpy_h2d(ctx, dc->d_recd, recd.data(), sizeof(double) * recd.size()));
    PMH_CHK(pmh_memset(ctx, dc->mid_in, 0, sizeof(double) * (size_t)Bg->nrows));
    dc->nlist = (int)(reci.size() / 8), dc->list_for = Bg, dc->list_uid = Bg->uid, dc->zeroed = dc->mid_in;
  }
  if (dc->sc_for != dc->scatter || dc->sc_uid != dc->scatter->uid) {
    const pmh_csr       Bs = dc->scatter;
    const int           n  = dc->n;
    std::vector<int>    rp((size_t)n + 1), ci((size_t)Bs->nnz);
    std::vector<double> va((size_t)Bs->nnz);
    PMH_CHK(pmh_memcpy_d2h(ctx, rp.data(), Bs->d_rowptr, sizeof(int) * rp.size()));
    PMH_CHK(pmh_memcpy_d2h(ctx, ci.data(), Bs->d_col, sizeof(int) * ci.size()));
    PMH_CHK(pmh_memcpy_d2h(ctx, va.data(), Bs->d_val, sizeof(double) * va.size()));
    std::vector<int>           sc2((size_t)2 * n, 0);
    std::vector<double>        sv2((size_t)2 * n, 0.0);
    std::vector<unsigned char> cnt((size_t)n, 0);
    for (int r = 0; r < n; r++) {
      const int c = rp[r + 1] - rp[r];
      cnt[r]      = (unsigned char)std::min(c, 255);
      for (int u = 0; u < 2 && u < c; u++) sc2[(size_t)2 * r + u] = ci[rp[r] + u], sv2[(size_t)2 * r + u] = va[rp[r] + u];
    }
    if (!dc->d_sc2) {
      PMH_CHK(pmh_malloc(ctx, sizeof(int) * sc2.size(), (void **)&dc->d_sc2));
      PMH_CHK(pmh_malloc(ctx, sizeof(double) * sv2.size(), (void **)&dc->d_sv2));
      PMH_CHK(pmh_malloc(ctx, cnt.size(), (void **)&dc->d_scnt));
    }
    PMH_CHK(pmh_memcpy_h2d(ctx, dc->d_sc2, sc2.data(), sizeof(int) * sc2.size()));
    PMH_CHK(pmh_memcpy_h2d(ctx, dc->d_sv2, sv2.data(), sizeof(double) * sv2.size()));
    PMH_CHK(pmh_memcpy_h2d(ctx, dc->d_scnt, cnt.data(), cnt.size()));
    dc->sc_for = Bs, dc->sc_uid = Bs->uid;
  }
  return PMH_SUCCESS;
}

void pmh_dc_destroy(pmh_dualchain dc)
{
  if (!dc) return;
  pmh_free(dc->ctx, dc->d_seg), pmh_free(dc->ctx, dc->d_lrow), pmh_free(dc->ctx, dc->d_ev), pmh_free(dc->ctx, dc->d_ec);
  if (dc->d_reci) pmh_free(dc->ctx, dc->d_reci), pmh_free(dc->ctx, dc->d_recd);
  if (dc->d_sc2) pmh_free(dc->ctx, dc->d_sc2), pmh_free(dc->ctx, dc->d_sv2), pmh_free(dc->ctx, dc->d_scnt);
  for (int s = 0; s < 3; s++) pmh_free(dc->ctx, dc->part[s]);
  pmh_free(dc->ctx, dc->c_cur), pmh_free(dc->ctx, dc->w);
  delete dc;
}

static pmh_emit_tab dc_tab(pmh_dualchain dc)
{
  pmh_emit_tab t;
  t.seg = dc->d_seg, t.gcol = dc->pf->G->d_col, t.gval = dc->pf->G->d_val, t.lrow = dc->d_lrow, t.m = dc->m, t.nwg = dc->nwg;
  return t;
}

int pmh_dc_emit_begin(pmh_dualchain dc, const double *x, const double *p, pmh_emit_args *ea)
{
  memset(ea, 0, sizeof(*ea));
  ea->tab = dc_tab(dc);
  if (x) {
    ea->o[0].part = dc->part[0];
    dc->x_emitted = x, dc->norm_done = false, dc->norm_ptr = nullptr;
  }
  if (p) ea->o[1].part = dc->part[1];
  return PMH_SUCCESS;
}

void pmh_dc_invalidate(pmh_dualchain dc) { dc->x_emitted = nullptr, dc->norm_ptr = nullptr, dc->norm_done = false; }

int pmh_dc_set_norm_target(pmh_dualchain dc, double *Gu, int slot)
{
  PMH_ARG(dc && Gu && slot >= 0 && slot < PMH_NSCAL);
  dc->normGu = Gu, dc->norm_slot = slot, dc->norm_ptr = nullptr, dc->norm_done = false;
  return PMH_SUCCESS;
}

// T G0 u and its squared norm are (as far as the stream has got) in place for this u; where the iterate's segment sums are there but no application has
// followed, one small launch forms them
bool pmh_dc_norm_ready(pmh_dualchain dc, const double *u)
{
  if (!dc || !dc->normGu) return false;
  if (dc->norm_ptr == u) return true;
  if (dc->x_emitted == u && !dc->norm_done) {
    if (dc->nwg <= 128)
      hipLaunchKernelGGL(k_dc_norm<2>, dim3(1), dim3(PMH_EMIT_TILE), 0, dc->ctx->stream, dc_tab(dc), (const double *)dc->part[0], (const double *)dc->pf->d_Tt,
                         dc->normGu, dc->ctx->d_scal + dc->norm_slot,
                         dc->ctx->h_scal + dc->norm_slot);
    else
      hipLaunchKernelGGL(k_dc_norm<8>, dim3(1), dim3(PMH_EMIT_TILE), 0, dc->ctx->stream, dc_tab(dc), (const double *)dc->part[0], (const double *)dc->pf->d_Tt,
                         dc->normGu, dc->ctx->d_scal + dc->norm_slot,
                         dc->ctx->h_scal + dc->norm_slot);
    if (hipGetLastError() != hipSuccess) return false;
    dc->norm_done = true, dc->norm_ptr = u;
    return true;
  }
  return false;
}
int pmh_dc_last_launches(pmh_dualchain dc) { return dc ? dc->launches : 0; }

// ---- one application ---------------------------------------------------------------------------------------------------------------
int pmh_dc_apply(pmh_dualchain dc, const double *x, double *y, double rho, const pmh_vec_epi *epi)
{
  pmh_ctx     ctx = dc->ctx;
  hipStream_t st  = ctx->stream;
  const int   n = dc->n;
  const dim3  vgrid((unsigned)dc->nwg), eblk(PMH_EMIT_TILE);
  const int   kind = epi ? epi->kind : 0;
  dc->launches     = 0;
  const pmh_emit_tab tab = dc_tab(dc);
  // 1. the segment sums of G0 x: left behind by the kernel that wrote x, or formed here
  int slot = (epi && epi->in_slot > 0) ? epi->in_slot - 1 : -1;
  if (slot < 0) {
    slot = (kind == PMH_VEPI_GRAD_SPLIT) ? 0 : (kind == PMH_VEPI_P1 ? 1 : 2);
    pmh_emit_args ea;
    memset(&ea, 0, sizeof(ea));
    ea.tab = tab, ea.o[0].part = dc->part[slot];
    if (slot == 0) dc->x_emitted = x, dc->norm_done = false, dc->norm_ptr = nullptr;
    hipLaunchKernelGGL(k_dc_emit, vgrid, eblk, 0, st, n, x, ea);
    dc->launches++;
  }
  // 2. gather with the projection folded in
  // (the stages are asked for at every application: explicit local dual operators may be attached to K^+ after the first product)
  PMH_CHK(dc->F->stages(&dc->gather, &dc->mid_in, &dc->scatter, &dc->mid_out));
  const pmh_csr Bg = dc->gather, Bs = dc->scatter;
  if (Bg->ncols != n || Bs->nrows != n) return pmh_set_error(PMH_ERR_STATE, "pmh_dc_apply: the operator's stages changed their dual dimension");
  PMH_CHK(dc_prepare_stages(dc));
  {
    dc_norm_args na;
    memset(&na, 0, sizeof(na));
    if (dc->normGu && dc->x_emitted && !dc->norm_done) { // SMALXE's ||B u|| for the iterate whose segment sums are waiting
      na.part = dc->part[0], na.Tt = dc->pf->d_Tt, na.y2 = dc->normGu, na.norm_d = ctx->d_scal + dc->norm_slot, na.norm_h = ctx->h_scal + dc->norm_slot;
      dc->norm_done = true, dc->norm_ptr = dc->x_emitted;
    }
    const dim3 ggrid((unsigned)(std::max(1, (dc->nlist + PMH_EMIT_TILE - 1) / PMH_EMIT_TILE) + (na.part ? 1 : 0))); // + the workgroup that forms ||B u||
#define DC_GATHER(WW, NU)                                                                                                                                                                       \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_dc_gather<WW, NU>), ggrid, eblk, 0, st, dc->nlist, (const int *)dc->d_reci, (const double *)dc->d_recd, (const int *)Bg->d_col, (const double *)Bg->d_val, \
                     (const double *)dc->d_ev, (const unsigned char *)dc->d_ec, tab, (const double *)dc->part[slot], (const double *)dc->pf->d_S, dc->c_cur, na, x, dc->mid_in)
    if (dc->W == 8 && dc->nwg <= 128) DC_GATHER(8, 2);
    else if (dc->W == 8) DC_GATHER(8, 8);
    else if (dc->nwg <= 128) DC_GATHER(16, 2);
    else DC_GATHER(16, 8);
#undef DC_GATHER
    dc->launches++;
  }
  PMH_HIP(hipGetLastError());
  // 3. the middle stage
  PMH_CHK(dc->F->mid_apply());
  // 4. scatter + the segment sums of G0 w
  {
    pmh_emit_args ea;
    memset(&ea, 0, sizeof(ea));
    ea.tab = tab, ea.o[0].part = dc->w + n;
    hipLaunchKernelGGL(k_dc_scatter, vgrid, eblk, 0, st, n, (const unsigned char *)dc->d_scnt, (const int *)dc->d_sc2, (const double *)dc->d_sv2,
                       (const int *)Bs->d_rowptr, (const int *)Bs->d_col,
                       (const double *)Bs->d_val, dc->mid_out, dc->w, ea);
    dc->launches++;
  }
  PMH_HIP(hipGetLastError());
  // the ranks' shares of B u and of the sums of G0 (B u) in one exchange (PetscSFReduce, gluing.c:144-147)
  PMH_CHK(pmh_comm_allreduce_sum(ctx, dc->w, (size_t)n + (size_t)dc->nseg));
  // 5. the second projection, the penalty term and the vector phase
  {
    pmh_emit_args ea;
    memset(&ea, 0, sizeof(ea));
    ea.tab = tab;
    pmh_vec_epi e;
    memset(&e, 0, sizeof(e));
    if (epi) e = *epi;
    e.h_partials = nullptr;
    if (epi && epi->hosted) e.h_partials = ctx->h_partials, *epi->hosted = dc->nwg;
    if (kind == PMH_VEPI_GRAD_SPLIT && epi->emitted_p) {
      ea.o[1].part    = dc->part[1];
      *epi->emitted_p = 1;
    }
#define DC_FINAL(EPI, WW, NU)                                                                                                                                                        \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_dc_final<EPI, WW, NU>), vgrid, eblk, 0, st, n, (const double *)dc->d_ev, (const unsigned char *)dc->d_ec, tab, (const double *)dc->pf->d_S, \
                     (const double *)dc->c_cur, (const double *)dc->w, rho, y, e, x, ea)
#define DC_FINAL_W(EPI)                                                                                                                                                              \
  do {                                                                                                                                                                               \
    if (dc->W == 8 && dc->nwg <= 128) DC_FINAL(EPI, 8, 2);                                                                                                                           \
    else if (dc->W == 8) DC_FINAL(EPI, 8, 8);                                                                                                                                        \
    else if (dc->nwg <= 128) DC_FINAL(EPI, 16, 2);                                                                                                                                   \
    else DC_FINAL(EPI, 16, 8);                                                                                                                                                       \
  } while (0)
    if (kind == PMH_VEPI_P1) DC_FINAL_W(PMH_VEPI_P1);
    else if (kind == PMH_VEPI_GRAD_SPLIT) DC_FINAL_W(PMH_VEPI_GRAD_SPLIT);
    else DC_FINAL_W(0);
#undef DC_FINAL_W
#undef DC_FINAL
    dc->launches++;
  }
  PMH_HIP(hipGetLastError());
  pmh_knobs().chain_applies++, pmh_knobs().chain_launches += dc->launches;
  return PMH_SUCCESS;
}
