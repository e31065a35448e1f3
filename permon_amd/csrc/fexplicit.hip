// Explicit local dual operators on gfx950: the exact K^+ path of the FETI dual operator.
//
// The reference applies K^+ by a per-block factorisation (matinv.c:435-590) and can form an inverse explicitly, column by
// column with its inner KSP (MatInvExplicitly_Inv, matinv.c:670-730: KSPSolve on the columns of the identity).  F = B K^+ B'
// (qptransform.c:1103-1128) only ever sees the entries of K_b^+ on Gamma_b = the primal dofs of block b that B touches
// (interface, Dirichlet and contact dofs), so the same construction restricted to those columns and rows,
//
//     W_b = (K_b^+)[Gamma_b, Gamma_b]        (dense, symmetric, n_Gamma_b ~ 17-25 k for a 44^3-node elasticity cube),
//
// gives F = sum_b Bhat_b W_b Bhat_b' EXACTLY (to the tolerance of the set-up solves, 1e-12), with Bhat the gluing over the
// compressed numbering [Gamma_0 | Gamma_1 | ...].  One F apply is then three launches -- Bhat' lambda (CSR), one dense GEMV over
// all blocks of the rank, Bhat (CSR, + the all-reduce on several GPUs) -- instead of ~340 launches of a 12-iteration
// multigrid-CG, and it streams 8 n_Gamma^2 bytes per block once: a pure HBM-bandwidth kernel (SURVEY 8d "dense path").
//
// Layout in HBM: W_b row-major with leading dimension ld_b (n_Gamma_b rounded up to 2, rows 16-byte aligned), all blocks of the
// rank in one allocation (configs[2]: 28 GB for the 8 blocks, ~5 GB the largest; far inside 288 GB).  The compressed vectors pad every
// block to an even offset so a lane's 16-byte load never straddles two blocks.
//
// Assembly.  Columns come from K^+ applications of a `solver` (a pmh_matinv whose blocks are "slots"): each application solves
// one unit right-hand side per slot at once.  Blocks with identical matrices form a class and share their columns (the 8
// congruent cubes of configs[2] need the 33 k boundary columns once, spread over 8 slots, instead of 8 x 17-25 k); a rank that
// owns a single block hands over a solver with several replica slots of it.  Rows are written (row j = column j, K^+ symmetric).
#include <algorithm>
#include <thread>
#include <chrono>
#include <cmath>
#include <map>

#include "feti_internal.h"
#include "fshared.h"
#include "pmh_internal.h"
#include "reduce.h"

typedef double dbl2 __attribute__((ext_vector_type(2)));

struct pmh_fexplicit_s {
  pmh_ctx       ctx;
  pmh_gluing    B;    // the gluing it was built from (borrowed)
  pmh_blockdiag K;    // block structure (borrowed)
  pmh_gluing    Bhat; // gluing over the compressed numbering (owned)
  int           nb, ntot; // blocks, padded length of the compressed vectors
  std::vector<int> gstart; // [nb+1] padded (even) offsets of the blocks in the compressed numbering
  std::vector<int> ngam;   // [nb] n_Gamma_b
  std::vector<int> gamma;  // [sum ngam] primal dof (rank-local concatenated numbering) of every compressed dof, ascending per block
  std::vector<size_t> goff; // [nb+1] offsets into gamma
  std::vector<int> ld;
  std::vector<double *> W; // device pointer of every block inside the single allocation Wbase
  double  *Wbase;
  std::vector<long long> woff; // [nb] offset (doubles) of block b in Wbase
  long long *d_woff;
  int     *d_gamma_rel;    // [sum ngam] gamma relative to its block's first row
  int     *d_ld, *d_gstart, *d_ngam;
  int     *d_wg_block, *d_wg_row0;
  int      nwg, rw;        // GEMV launch table: workgroup -> (block, first row); rows per wave
  fx_shared *sh;           // PMH_FX_CLASS: congruent blocks share one full matrix per class (fshared.hip); everything dense lives there
  int      storage;        // PMH_FX_FULL: n x ld row-major; PMH_FX_SYM: lower block-triangle in bands of 32 rows (see k_fx_symv)
  double  *partial, *ydir; // SYM: transposed-product partials [block][super band][npad] and the direct products [block][segment][npad]
  long long *d_poff, *d_doff; // SYM: offset of block b in `partial` / in `ydir`
  int     *d_sw_block, *d_sw_band, *d_sw_seg; // SYM launch table: workgroup -> (block, super band, column segment)
  int     *d_fw_block, *d_fw_col0; // SYM second launch: workgroup -> (block, first of its 128 columns)
  int     *d_own_ptr, *d_own_list; // SYM: per block the ascending list of the super bands this rank owns
  int      nsw, nfw, seg; // seg: tile columns per workgroup of k_fx_symv
  double   sym_bytes;      // SYM: algorithmic bytes of one apply
  double  *xh, *yh;        // compressed work vectors
  int      assembled;
  int      stripe_rank, stripe_size; // > 0: this rank applies / assembles only the super bands idx % size == rank (see pmh_fexplicit_set_stripe)
  std::vector<std::vector<char>> owned; // [block][super band]
  long long n_solves;
  double   assemble_seconds;
  std::vector<hipEvent_t> ev, ev_mid; // pairs around the dense launch(es); SYM: a third event between k_fx_symv and k_fx_symv_fin
  int      ev_used, ev_on, ev_seen, ev_stride;
};

// ---- kernels ---------------------------------------------------------------------------------------------------------

// y_b = W_b x_b for every block of the rank in one launch.  A workgroup owns 4*RW consecutive rows of one block (wave w the rows
// row0 + w*RW ...), every lane streams its 16-byte column pairs of the RW rows (non-temporal: each byte is used once) against the
// matching pair of x (served by L2: x_b is 0.2 MB), per-lane partial sums in column order, one shuffle tree per row at the end.
// Fixed summation order => bitwise reproducible; -ffp-contract=off keeps the two products of a pair separate.
template <int RW>
__global__ __launch_bounds__(PMH_BLOCK) void k_fx_gemv(const int *__restrict__ wg_block, const int *__restrict__ wg_row0, const int *__restrict__ gstart, const int *__restrict__ ngam, const int *__restrict__ ldv,
                                                      const long long *__restrict__ woff, const double *__restrict__ Wbase, const double *__restrict__ xh, double *__restrict__ yh)
{
  const int b = wg_block[blockIdx.x], lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = ngam[b], ld = ldv[b];
  const int row0 = wg_row0[blockIdx.x] + wave * RW;
  if (row0 >= n) return;
  const double *__restrict__ A = Wbase + woff[b]; // one allocation for all blocks: the row pointers stay in the global address space
  const double *__restrict__ x = xh + gstart[b];
  const double *rp[RW];
#pragma unroll
  for (int r = 0; r < RW; r++) rp[r] = A + (size_t)min(row0 + r, n - 1) * ld; // rows past the end re-read the last row, result dropped
  double acc[RW];
#pragma unroll
  for (int r = 0; r < RW; r++) acc[r] = 0.0;
  int c = lane * 2;
  // two 128-column chunks per trip: 2*RW 16-byte loads in flight per lane before the first use
  for (; c + 128 < ld; c += 256) {
    const dbl2 x0 = *(const dbl2 *)(x + c), x1 = *(const dbl2 *)(x + c + 128);
    dbl2       a0[RW], a1[RW];
#pragma unroll
    for (int r = 0; r < RW; r++) a0[r] = __builtin_nontemporal_load((const dbl2 *)(rp[r] + c));
#pragma unroll
    for (int r = 0; r < RW; r++) a1[r] = __builtin_nontemporal_load((const dbl2 *)(rp[r] + c + 128));
#pragma unroll
    for (int r = 0; r < RW; r++) {
      acc[r] += a0[r].x * x0.x;
      acc[r] += a0[r].y * x0.y;
    }
#pragma unroll
    for (int r = 0; r < RW; r++) {
      acc[r] += a1[r].x * x1.x;
      acc[r] += a1[r].y * x1.y;
    }
  }
  if (c < ld) {
    const dbl2 x0 = *(const dbl2 *)(x + c);
#pragma unroll
    for (int r = 0; r < RW; r++) {
      const dbl2 a = __builtin_nontemporal_load((const dbl2 *)(rp[r] + c));
      acc[r] += a.x * x0.x;
      acc[r] += a.y * x0.y;
    }
  }
#pragma unroll
  for (int r = 0; r < RW; r++) {
    const double s = pmh_wave_sum(acc[r]);
    if (lane == 0 && row0 + r < n) yh[gstart[b] + row0 + r] = s;
  }
}

// Symmetric storage (PMH_FX_SYM): W_b = W_b' is kept as its lower block-triangle in BANDS of FX_RB = 32 rows.  Band k holds the
// rows 32k .. 32k+31 and the columns 0 .. 32(k+1)-1 (the diagonal 32 x 32 tile in full), cut into TILES of 32 rows x 128 columns
// that are stored contiguously (32 KB each, tile-major, row-major inside a tile; the last tile of a band is zero padded): one trip
// of a wave reads one contiguous 32 KB tile, whatever the band's length -- no large strides, no channel aliasing between the rows.
// Band k has ceil((k+1)/4) tiles and starts at tile T(k) = (q+1)(2q+r), k = 4q+r.  n is padded to a multiple of 128 (zero rows /
// columns), which removes every bounds check.  One workgroup per band; wave w owns the tiles w, w+4, ... and streams ALL 32 rows
// of a tile (32 16-byte non-temporal loads per lane in flight):
//   direct     y_band[r] += a[r][c] x[c]      per-lane partial sums in column order, one shuffle tree per row at the end of the band,
//                                             the four waves' results combined through LDS in wave order;
//   transposed z[c]      += a[r][c] x_band[r] for the columns left of the diagonal tile: the lane owns column pair c for all 32 rows,
//                                             so z is complete for the band without any exchange and goes to partial[band][c].
// k_fx_symv_fin then adds, for every c, the direct product and the partials of the bands below c's band in band order.  Every
// matrix byte is read once (half of the GEMV's bytes), the partials add 2 x n^2/64 x 8 bytes (6 %); fixed summation order.
#define FX_RB 32
#define FX_TC 128                 // columns per tile
#define FX_TILE (FX_RB * FX_TC)   // doubles per tile
__host__ __device__ __forceinline__ long long fx_band_tiles(int k) { return (long long)(k / 4 + 1) * (2 * (k / 4) + (k & 3)); } // tiles before band k
__host__ __device__ __forceinline__ long long fx_band_off(int k) { return (long long)FX_TILE * fx_band_tiles(k); }
// doubles of the SYM storage of a block padded to npad (multiple of 128) rows
__host__ __device__ __forceinline__ long long fx_sym_size(int npad) { return fx_band_off(npad / FX_RB); }

// Work decomposition: a workgroup owns (super band s = 4 bands = 128 rows) x (segment j = FX_SEG tile columns = 2048 columns);
// its wave w streams band 4s+w through the segment, one 32 KB tile per trip, the four waves in step.  Per tile column the four
// transposed partial sums z_w (the lane owns its column pair for the 32 rows of the band) are added in wave order through LDS and
// written ONCE per 128 rows (partial[s][c]: n^2/256 entries, 1.6 % of the matrix bytes -- written per band they cost 22 % of the
// kernel time, measured); the direct sums stay in registers across the segment and end in ydseg[j][row].  Every workgroup moves
// the same 2 MB (except the last segment of a super band), so the grid balances at any block count.
#define FX_SEG_MIN 2
#define FX_SEG_MAX 16 // tile columns per workgroup: 16, or fewer when the rank's share is small (see fx_pick_seg)
template <int VAR>
__global__ __launch_bounds__(PMH_BLOCK) void k_fx_symv(const int *__restrict__ sw_block, const int *__restrict__ sw_sband, const int *__restrict__ sw_seg, const int *__restrict__ gstart, const int *__restrict__ ldv,
                                                      const long long *__restrict__ woff, const long long *__restrict__ poff, const long long *__restrict__ doff, const double *__restrict__ Wbase,
                                                      const double *__restrict__ xh, double *__restrict__ ydseg, double *__restrict__ partial, int FX_SEG)
{
  __shared__ dbl2 zred[2][PMH_BLOCK / 64][64];
  const int b = __builtin_amdgcn_readfirstlane(sw_block[blockIdx.x]), sb = __builtin_amdgcn_readfirstlane(sw_sband[blockIdx.x]), seg = __builtin_amdgcn_readfirstlane(sw_seg[blockIdx.x]);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int npad = ldv[b], k = 4 * sb + wave, row0 = k * FX_RB;
  const int t0 = seg * FX_SEG, t1 = min(t0 + FX_SEG, sb + 1); // every band of the super band has sb + 1 tiles
  const double *__restrict__ A = Wbase + woff[b] + fx_band_off(k);
  const double *__restrict__ x = xh + gstart[b];
  double *__restrict__ prow    = partial + poff[b] + (long long)sb * npad;
  double xr[FX_RB]; // x over the band's rows: uniform across the wave (scalar loads)
#pragma unroll
  for (int r = 0; r < FX_RB; r++) xr[r] = x[row0 + r];
  double acc[FX_RB];
#pragma unroll
  for (int r = 0; r < FX_RB; r++) acc[r] = 0.0;
  for (int t = t0; t < t1; t++) {
    const int     c  = t * FX_TC + lane * 2;
    const dbl2    xc = *(const dbl2 *)(x + c);
    const double *ap = A + (long long)t * FX_TILE + lane * 2;
    dbl2          z  = {0.0, 0.0};
#pragma unroll
    for (int h = 0; h < FX_RB; h += 16) {
      dbl2 a[16];
#pragma unroll
      for (int r = 0; r < 16; r++) a[r] = (VAR == 1) ? *(const dbl2 *)(ap + (h + r) * FX_TC) : __builtin_nontemporal_load((const dbl2 *)(ap + (h + r) * FX_TC));
#pragma unroll
      for (int r = 0; r < 16; r++) {
        acc[h + r] = __builtin_fma(a[r].x, xc.x, acc[h + r]); // fused multiply-adds: the dense product has no reference summation order to
        acc[h + r] = __builtin_fma(a[r].y, xc.y, acc[h + r]); // reproduce (W is K^+ data), the order stays fixed
        z.x        = __builtin_fma(a[r].x, xr[h + r], z.x);
        z.y        = __builtin_fma(a[r].y, xr[h + r], z.y);
      }
    }
    if (c >= row0) z = dbl2{0.0, 0.0}; // columns inside the band's diagonal tile are covered by the tile's own rows (direct product)
    const int q = t & 1;               // double-buffered: one barrier per tile column
    zred[q][wave][lane] = z;
    __syncthreads();
    if (wave == (t & 3)) {
      dbl2 zs = zred[q][0][lane];
      zs += zred[q][1][lane];
      zs += zred[q][2][lane];
      zs += zred[q][3][lane];
      *(dbl2 *)(prow + c) = zs;
    }
  }
  double *__restrict__ yd = ydseg + doff[b] + (long long)seg * npad + row0;
#pragma unroll
  for (int r = 0; r < FX_RB; r++) {
    const double sm = pmh_wave_sum(acc[r]);
    if (lane == 0) yd[r] = sm;
  }
}

// y[c] = sum over the segments j of ydseg[j][c] (segment order) + sum over the OWNED super bands s >= c / 128 of partial[s][c] (wave w
// takes every 4th of them in ascending order, eight 16-byte loads in flight per lane; the four waves' sums are added in wave order).
// One workgroup per 128 columns of a block.
__global__ __launch_bounds__(PMH_BLOCK) void k_fx_symv_fin(const int *__restrict__ fw_block, const int *__restrict__ fw_col0, const int *__restrict__ gstart, const int *__restrict__ ldv, const long long *__restrict__ poff,
                                                          const long long *__restrict__ doff, const int *__restrict__ own_ptr, const int *__restrict__ own_list, const double *__restrict__ ydseg,
                                                          const double *__restrict__ partial, double *__restrict__ yh, int FX_SEG)
{
  __shared__ dbl2 red[PMH_BLOCK / 64][64];
  const int b = fw_block[blockIdx.x], lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int npad = ldv[b];
  const int c    = fw_col0[blockIdx.x] + lane * 2; // npad is a multiple of 128: always inside the block
  const int sc   = c / FX_TC;                      // the column's own super band: its diagonal tile column holds the in-super-band transposed terms
  dbl2      s    = {0.0, 0.0};
  {
    // the super bands this rank owns (all of them without striping), ascending; start at the first one >= sc
    const int *own = own_list + own_ptr[b];
    const int  cnt = own_ptr[b + 1] - own_ptr[b];
    int        lo = 0, hi = cnt;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (own[mid] < sc) lo = mid + 1;
      else hi = mid;
    }
    const double *p = partial + poff[b] + c;
    for (int q = lo + wave; q < cnt; q += 32) { // wave w takes the list positions lo + w (mod 4), eight loads in flight
      dbl2 v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = (q + 4 * u < cnt) ? *(const dbl2 *)(p + (long long)own[q + 4 * u] * npad) : dbl2{0.0, 0.0};
#pragma unroll
      for (int u = 0; u < 8; u++) s += v[u];
    }
  }
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0) {
    dbl2          t  = red[0][lane];
    t += red[1][lane];
    t += red[2][lane];
    t += red[3][lane];
    const int     nseg = sc / FX_SEG + 1; // segments of the row's super band (sc + 1 tile columns)
    const double *yd   = ydseg + doff[b] + c;
    for (int j = 0; j < nseg; j++) t += *(const dbl2 *)(yd + (long long)j * npad);
    *(dbl2 *)(yh + gstart[b] + c) = t;
  }
}

// unit right-hand sides of one assembly batch: rhs[idx[s]] = val for the slots of the batch (idx < 0: slot idle)
__global__ void k_fx_set_entries(int m, const int *__restrict__ idx, double val, double *__restrict__ rhs)
{
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < m && idx[s] >= 0) rhs[idx[s]] = val;
}

// one row of W_b: wrow[i] = u[gamma_rel[i]] (u = the slot's solution), i < n; the pad entry stays 0
__global__ __launch_bounds__(PMH_BLOCK) void k_fx_extract(int n, const int *__restrict__ gamma_rel, const double *__restrict__ u, double *__restrict__ wrow)
{
  for (int i = blockIdx.x * PMH_BLOCK + threadIdx.x; i < n; i += gridDim.x * PMH_BLOCK) wrow[i] = u[gamma_rel[i]];
}

// per block the ascending list of owned super bands (E->owned) -> device
static int fx_upload_owned(pmh_fexplicit E)
{
  std::vector<int> ptr(E->nb + 1, 0), lst;
  for (int b = 0; b < E->nb; b++) {
    for (int sb = 0; sb < (int)E->owned[b].size(); sb++)
      if (E->owned[b][sb]) lst.push_back(sb);
    ptr[b + 1] = (int)lst.size();
  }
  int tot = 0;
  for (int b = 0; b < E->nb; b++) tot += (int)E->owned[b].size();
  if (!E->d_own_ptr) {
    PMH_CHK(pmh_malloc(E->ctx, sizeof(int) * (E->nb + 1), (void **)&E->d_own_ptr));
    PMH_CHK(pmh_malloc(E->ctx, sizeof(int) * (size_t)std::max(1, tot), (void **)&E->d_own_list));
  }
  lst.push_back(0);
  PMH_CHK(pmh_memcpy_h2d(E->ctx, E->d_own_ptr, ptr.data(), sizeof(int) * (E->nb + 1)));
  return pmh_memcpy_h2d(E->ctx, E->d_own_list, lst.data(), sizeof(int) * std::max<size_t>(1, lst.size() - 1));
}

// launch table of k_fx_symv for the super bands this rank owns, largest first; picks the segment length (tile columns per workgroup):
// 16 unless that leaves fewer than ~6 rounds of workgroups on the chip (256 CUs x 2 resident), then 8, 4 or 2 -- a rank's 1/8 share
// of configs[2] would otherwise run 2.2 rounds of equal 2 MB workgroups, i.e. a third of the last round idle.  Also the algorithmic
// bytes of one apply of this rank's share.
static int fx_build_launch(pmh_fexplicit E)
{
  const int nb = E->nb;
  int       maxsb = 0;
  for (int b = 0; b < nb; b++) maxsb = std::max(maxsb, E->ld[b] / FX_TC);
  int seg = FX_SEG_MAX;
  for (; seg > FX_SEG_MIN; seg /= 2) {
      long long n = 0;
      for (int b = 0; b < nb; b++)
        for (int sb = 0; sb < E->ld[b] / FX_TC; sb++)
          if (E->owned[b][sb]) n += (sb + seg) / seg;
      if (n >= 6 * 512) break;
    }
  E->seg = seg;
  std::vector<int> swb, swk, swj;
  for (int sb = maxsb - 1; sb >= 0; sb--)
    for (int b = 0; b < nb; b++)
      if (sb < E->ld[b] / FX_TC && E->owned[b][sb])
        for (int j = 0; j * seg < sb + 1; j++) swb.push_back(b), swk.push_back(sb), swj.push_back(j);
  E->nsw = (int)swb.size();
  swb.push_back(0), swk.push_back(0), swj.push_back(0);
  PMH_CHK(pmh_memcpy_h2d(E->ctx, E->d_sw_block, swb.data(), sizeof(int) * swb.size()));
  PMH_CHK(pmh_memcpy_h2d(E->ctx, E->d_sw_band, swk.data(), sizeof(int) * swk.size()));
  PMH_CHK(pmh_memcpy_h2d(E->ctx, E->d_sw_seg, swj.data(), sizeof(int) * swj.size()));
  PMH_CHK(fx_upload_owned(E));
  E->sym_bytes = 0.0;
  for (int b = 0; b < nb; b++) {
    const int nsb = E->ld[b] / FX_TC;
    for (int sb = 0; sb < nsb; sb++)
      // its tiles, its partial row and its direct sums written + read back
      if (E->owned[b][sb]) E->sym_bytes += 8.0 * 4.0 * FX_TILE * (sb + 1) + 2.0 * 8.0 * FX_TC * (sb + 1) + 2.0 * 8.0 * 128.0 * ((sb + seg) / seg);
    E->sym_bytes += 16.0 * E->ld[b]; // x read, y written
  }
  return PMH_SUCCESS;
}

// SYM: the part of row j = 32 k + r that lies in its band, into the band's tiles: element (r, i) at (i / 128) tiles + r * 128 + i % 128
__global__ __launch_bounds__(PMH_BLOCK) void k_fx_extract_tiled(int n, const int *__restrict__ gamma_rel, const double *__restrict__ u, double *__restrict__ band, int r)
{
  for (int i = blockIdx.x * PMH_BLOCK + threadIdx.x; i < n; i += gridDim.x * PMH_BLOCK) band[(long long)(i / FX_TC) * FX_TILE + r * FX_TC + (i % FX_TC)] = u[gamma_rel[i]];
}

// ---- create / destroy ---------------------------------------------------------------------------------------------------

static int fx_create(pmh_gluing B, pmh_blockdiag K, int storage, const int *block_class, pmh_fexplicit *out, const int *extra_ptr = nullptr, const int *extra_rel = nullptr);

extern "C" int pmh_fexplicit_create(pmh_gluing B, pmh_blockdiag K, int storage, pmh_fexplicit *out)
{
  PMH_ARG(storage == PMH_FX_FULL || storage == PMH_FX_SYM);
  return fx_create(B, K, storage, nullptr, out);
}

// congruent blocks (block_class from pmh_csr_block_classes) share ONE full dense matrix per class, applied to the blocks' vectors
// together, eight right-hand sides per pass (fshared.hip)
extern "C" int pmh_fexplicit_create_shared(pmh_gluing B, pmh_blockdiag K, const int *block_class, pmh_fexplicit *out)
{
  PMH_ARG(block_class);
  return fx_create(B, K, PMH_FX_CLASS, block_class, out);
}

// the same with W_c = W_c' kept as its lower block-triangle in 16 x 16 tiles (half the bytes; fp64 MFMA kernel k_fxs_symm8)
extern "C" int pmh_fexplicit_create_shared_sym(pmh_gluing B, pmh_blockdiag K, const int *block_class, pmh_fexplicit *out)
{
  PMH_ARG(block_class);
  return fx_create(B, K, PMH_FX_CLASS_SYM, block_class, out);
}

// only the rows of W_c of the orbit representatives under the class's symmetries (set them before the assembly): the apply is a GEMM on the fp64
// matrix instruction (fshared.hip, FXO section)
extern "C" int pmh_fexplicit_create_shared_orbit(pmh_gluing B, pmh_blockdiag K, const int *block_class, pmh_fexplicit *out)
{
  PMH_ARG(block_class);
  return fx_create(B, K, PMH_FX_CLASS_ORBIT, block_class, out);
}

// ... with the class sets extended (extra_ptr: nclasses + 1 offsets, extra_rel: block-relative dofs): decompositions whose blocks are symmetric boxes but not
// congruent (one material per subdomain) have one class per block, whose own touched set -- three interface faces, a Dirichlet or contact face -- is mapped
// onto itself by 2 ... 8 of the box's 48 operations only; on the closure of that set under the whole group (pmh_box_symmetry_closure: the whole boundary of a
// cube) every operation survives, the set-up needs one K^+ solve per orbit (715 instead of 17 000 ... 24 000 for a 44^3-node cube) and the apply is the GEMM of
// the orbit storage instead of the HBM-bound stream over a full W_b
extern "C" int pmh_fexplicit_create_shared_orbit_union(pmh_gluing B, pmh_blockdiag K, const int *block_class, const int *extra_ptr, const int *extra_rel, pmh_fexplicit *out)
{
  PMH_ARG(block_class && extra_ptr && (extra_rel || true));
  return fx_create(B, K, PMH_FX_CLASS_ORBIT, block_class, out, extra_ptr, extra_rel);
}

static int fx_create(pmh_gluing B, pmh_blockdiag K, int storage, const int *block_class, pmh_fexplicit *out, const int *extra_ptr, const int *extra_rel)
{
  PMH_ARG(B && K && out);
  PMH_ARG(B->n_x == K->n);
  pmh_ctx       ctx = B->ctx;
  pmh_fexplicit E   = new pmh_fexplicit_s();
  E->ctx = ctx, E->B = B, E->K = K, E->Bhat = nullptr, E->nb = K->nblocks;
  E->d_gamma_rel = nullptr, E->Wbase = nullptr, E->d_woff = nullptr, E->d_ld = E->d_gstart = E->d_ngam = E->d_wg_block = E->d_wg_row0 = nullptr;
  E->xh = E->yh = nullptr, E->assembled = 0, E->n_solves = 0, E->assemble_seconds = 0.0;
  E->stripe_rank = 0, E->stripe_size = 0, E->sh = nullptr;
  E->ev_used = E->ev_on = E->ev_seen = 0, E->ev_stride = 1;
  E->storage = storage, E->partial = E->ydir = nullptr, E->d_poff = E->d_doff = nullptr, E->d_sw_block = E->d_sw_band = E->d_sw_seg = E->d_fw_block = E->d_fw_col0 = E->d_own_ptr = E->d_own_list = nullptr, E->nsw = E->nfw = 0, E->sym_bytes = 0.0;
  const int nb = E->nb;
  // Gamma_b: the primal dofs with at least one leaf, ascending inside every block
  std::vector<char> touched((size_t)std::max(1, B->n_x), 0);
  for (int i = 0; i < B->n_leaves; i++) touched[B->h_row[i]] = 1;
  std::vector<int> newidx((size_t)std::max(1, B->n_x), -1);
  E->gstart.assign(nb + 1, 0), E->ngam.assign(nb, 0), E->goff.assign(nb + 1, 0), E->ld.assign(nb, 0);
  int off = 0;
  for (int b = 0; b < nb; b++) {
    E->gstart[b] = off;
    E->goff[b]   = E->gamma.size();
    int cnt = 0;
    for (int i = K->rowstart[b]; i < K->rowstart[b + 1]; i++)
      if (touched[i]) {
        newidx[i] = off + cnt++;
        E->gamma.push_back(i);
      }
    E->ngam[b] = cnt;
    E->ld[b]   = (cnt + FX_TC - 1) / FX_TC * FX_TC;
    off += E->ld[b]; // every block padded to a multiple of 128: aligned 16-byte loads, whole bands and tiles; the pad entries are empty rows of Bhat'
  }
  E->gstart[nb] = off, E->goff[nb] = E->gamma.size(), E->ntot = off;
  // the dense side lives in the class-shared object; the Gamma_b lists above serve sizes / get_block
  if (storage == PMH_FX_CLASS || storage == PMH_FX_CLASS_SYM || storage == PMH_FX_CLASS_ORBIT) {
    E->W.assign(nb, nullptr), E->woff.assign(nb, 0);
    PMH_CHK(fxs_create(B, K, block_class, storage == PMH_FX_CLASS_ORBIT ? 2 : (storage == PMH_FX_CLASS_SYM ? 1 : 0), &E->sh, extra_ptr, extra_rel));
    *out = E;
    return PMH_SUCCESS;
  }
  // Bhat: same leaves (same order => same summation order as B), primal index remapped
  std::vector<int> rows((size_t)std::max(1, B->n_leaves));
  for (int i = 0; i < B->n_leaves; i++) rows[i] = newidx[B->h_row[i]];
  PMH_CHK(pmh_gluing_create(ctx, E->ntot, B->n_lambda, B->n_leaves, rows.data(), B->h_root.data(), B->h_sign.data(), &E->Bhat));
  // dense blocks in one allocation (block offsets 256-byte aligned), zero-initialised (the pad column must stay 0)
  E->W.assign(nb, nullptr), E->woff.assign(nb, 0);
  long long wtot = 0;
  for (int b = 0; b < nb; b++) {
    E->woff[b] = wtot;
    wtot += (storage == PMH_FX_SYM) ? fx_sym_size(E->ld[b]) : (long long)E->ld[b] * E->ld[b];
  }
  {
    const size_t bytes = sizeof(double) * (size_t)std::max(32LL, wtot);
    hipError_t   e     = hipMalloc((void **)&E->Wbase, bytes);
    if (e != hipSuccess) return pmh_set_error(PMH_ERR_HIP, "pmh_fexplicit_create: %.2f GB for the explicit blocks: %s", bytes / 1e9, hipGetErrorString(e));
    PMH_HIP(hipMemsetAsync(E->Wbase, 0, bytes, ctx->stream));
  }
  for (int b = 0; b < nb; b++) E->W[b] = E->Wbase + E->woff[b];
  std::vector<int> grel(E->gamma.size() + 1);
  for (int b = 0; b < nb; b++)
    for (size_t k = E->goff[b]; k < E->goff[b + 1]; k++) grel[k] = E->gamma[k] - K->rowstart[b];
  PMH_CHK(pmh_malloc(ctx, sizeof(int) * grel.size(), (void **)&E->d_gamma_rel));
  PMH_CHK(pmh_memcpy_h2d(ctx, E->d_gamma_rel, grel.data(), sizeof(int) * grel.size()));
  PMH_CHK(pmh_malloc(ctx, sizeof(long long) * nb, (void **)&E->d_woff));
  PMH_CHK(pmh_memcpy_h2d(ctx, E->d_woff, E->woff.data(), sizeof(long long) * nb));
  PMH_CHK(pmh_malloc(ctx, sizeof(int) * nb, (void **)&E->d_ld));
  PMH_CHK(pmh_memcpy_h2d(ctx, E->d_ld, E->ld.data(), sizeof(int) * nb));
  PMH_CHK(pmh_malloc(ctx, sizeof(int) * (nb + 1), (void **)&E->d_gstart));
  PMH_CHK(pmh_memcpy_h2d(ctx, E->d_gstart, E->gstart.data(), sizeof(int) * (nb + 1)));
  PMH_CHK(pmh_malloc(ctx, sizeof(int) * nb, (void **)&E->d_ngam));
  PMH_CHK(pmh_memcpy_h2d(ctx, E->d_ngam, E->ngam.data(), sizeof(int) * nb));
  // GEMV launch table
  E->rw = 4; // rows per wave of the full-matrix GEMV (2 / 4 / 8 measured in round 2: profiles/r02_symv_tune.txt)
  const int        rows_per_wg = 4 * E->rw;
  std::vector<int> wb, wr;
  for (int b = 0; b < nb; b++)
    for (int r = 0; r < E->ngam[b]; r += rows_per_wg) wb.push_back(b), wr.push_back(r);
  E->nwg = (int)wb.size();
  PMH_CHK(pmh_malloc(ctx, sizeof(int) * std::max(1, E->nwg), (void **)&E->d_wg_block));
  PMH_CHK(pmh_malloc(ctx, sizeof(int) * std::max(1, E->nwg), (void **)&E->d_wg_row0));
  if (E->nwg) {
    PMH_CHK(pmh_memcpy_h2d(ctx, E->d_wg_block, wb.data(), sizeof(int) * E->nwg));
    PMH_CHK(pmh_memcpy_h2d(ctx, E->d_wg_row0, wr.data(), sizeof(int) * E->nwg));
  }
  if (storage == PMH_FX_SYM) { // partial / direct-product buffers (sized for the shortest segments) and the launch table
    std::vector<long long> poff(nb), doff(nb);
    long long              ptot = 0, dtot = 0, wgmax = 0;
    for (int b = 0; b < nb; b++) {
      const int nsb = E->ld[b] / FX_TC;
      poff[b] = ptot, doff[b] = dtot;
      ptot += (long long)nsb * E->ld[b];
      dtot += (long long)((nsb + FX_SEG_MIN - 1) / FX_SEG_MIN) * E->ld[b];
      for (int sb = 0; sb < nsb; sb++) wgmax += (sb + FX_SEG_MIN) / FX_SEG_MIN;
    }
    E->owned.assign(nb, std::vector<char>());
    for (int b = 0; b < nb; b++) E->owned[b].assign(E->ld[b] / FX_TC, 1);
    PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)std::max(1LL, ptot), (void **)&E->partial));
    PMH_CHK(pmh_memset(ctx, E->partial, 0, sizeof(double) * (size_t)std::max(1LL, ptot)));
    PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)std::max(1LL, dtot), (void **)&E->ydir));
    PMH_CHK(pmh_memset(ctx, E->ydir, 0, sizeof(double) * (size_t)std::max(1LL, dtot)));
    PMH_CHK(pmh_malloc(ctx, sizeof(long long) * nb, (void **)&E->d_poff));
    PMH_CHK(pmh_memcpy_h2d(ctx, E->d_poff, poff.data(), sizeof(long long) * nb));
    PMH_CHK(pmh_malloc(ctx, sizeof(long long) * nb, (void **)&E->d_doff));
    PMH_CHK(pmh_memcpy_h2d(ctx, E->d_doff, doff.data(), sizeof(long long) * nb));
    PMH_CHK(pmh_malloc(ctx, sizeof(int) * (size_t)(wgmax + 1), (void **)&E->d_sw_block));
    PMH_CHK(pmh_malloc(ctx, sizeof(int) * (size_t)(wgmax + 1), (void **)&E->d_sw_band));
    PMH_CHK(pmh_malloc(ctx, sizeof(int) * (size_t)(wgmax + 1), (void **)&E->d_sw_seg));
    PMH_CHK(fx_build_launch(E));
    PMH_CHK(fx_upload_owned(E));
    std::vector<int> fwb, fwc;
    for (int b = 0; b < nb; b++)
      for (int c0 = 0; c0 < E->ld[b]; c0 += 128) fwb.push_back(b), fwc.push_back(c0);
    E->nfw = (int)fwb.size();
    fwb.push_back(0), fwc.push_back(0);
    PMH_CHK(pmh_malloc(ctx, sizeof(int) * fwb.size(), (void **)&E->d_fw_block));
    PMH_CHK(pmh_malloc(ctx, sizeof(int) * fwc.size(), (void **)&E->d_fw_col0));
    PMH_CHK(pmh_memcpy_h2d(ctx, E->d_fw_block, fwb.data(), sizeof(int) * fwb.size()));
    PMH_CHK(pmh_memcpy_h2d(ctx, E->d_fw_col0, fwc.data(), sizeof(int) * fwc.size()));
  }
  PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)std::max(2, E->ntot), (void **)&E->xh));
  PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)std::max(2, E->ntot), (void **)&E->yh));
  PMH_CHK(pmh_memset(ctx, E->xh, 0, sizeof(double) * (size_t)std::max(2, E->ntot)));
  PMH_CHK(pmh_memset(ctx, E->yh, 0, sizeof(double) * (size_t)std::max(2, E->ntot)));
  *out = E;
  return PMH_SUCCESS;
}

extern "C" int pmh_fexplicit_destroy(pmh_fexplicit E)
{
  if (!E) return PMH_SUCCESS;
  pmh_ctx ctx = E->ctx;
  if (E->sh) {
    fxs_destroy(E->sh);
    delete E;
    return PMH_SUCCESS;
  }
  if (E->Wbase) (void)hipFree(E->Wbase);
  pmh_gluing_destroy(E->Bhat);
  pmh_free(ctx, E->d_gamma_rel), pmh_free(ctx, E->d_woff), pmh_free(ctx, E->d_ld), pmh_free(ctx, E->d_gstart), pmh_free(ctx, E->d_ngam);
  pmh_free(ctx, E->d_wg_block), pmh_free(ctx, E->d_wg_row0), pmh_free(ctx, E->xh), pmh_free(ctx, E->yh);
  if (E->partial) pmh_free(ctx, E->partial);
  if (E->ydir) pmh_free(ctx, E->ydir);
  if (E->d_poff) pmh_free(ctx, E->d_poff);
  if (E->d_sw_block) pmh_free(ctx, E->d_sw_block);
  if (E->d_sw_band) pmh_free(ctx, E->d_sw_band);
  if (E->d_sw_seg) pmh_free(ctx, E->d_sw_seg);
  if (E->d_doff) pmh_free(ctx, E->d_doff);
  if (E->d_fw_block) pmh_free(ctx, E->d_fw_block);
  if (E->d_fw_col0) pmh_free(ctx, E->d_fw_col0);
  if (E->d_own_ptr) pmh_free(ctx, E->d_own_ptr);
  if (E->d_own_list) pmh_free(ctx, E->d_own_list);
  for (hipEvent_t e : E->ev) (void)hipEventDestroy(e);
  for (hipEvent_t e : E->ev_mid) (void)hipEventDestroy(e);
  delete E;
  return PMH_SUCCESS;
}

extern "C" int pmh_fexplicit_sizes(pmh_fexplicit E, int *nblocks, int *n_gamma, long long *dense_bytes, double *gemv_bytes)
{
  PMH_ARG(E);
  if (nblocks) *nblocks = E->nb;
  long long tot = 0;
  double    alg = 0.0;
  for (int b = 0; b < E->nb; b++) {
    if (n_gamma) n_gamma[b] = E->ngam[b];
    if (!E->sh) tot += (long long)sizeof(double) * (E->storage == PMH_FX_SYM ? fx_sym_size(E->ld[b]) : (long long)E->ld[b] * E->ld[b]);
    alg += 8.0 * (double)E->ngam[b] * E->ngam[b] + 16.0 * E->ngam[b]; // FULL: the matrix once + x read + y written
  }
  if (E->sh) tot = fxs_dense_bytes(E->sh), alg = fxs_apply_bytes(E->sh);
  if (dense_bytes) *dense_bytes = tot;
  if (gemv_bytes) *gemv_bytes = (E->storage == PMH_FX_SYM) ? E->sym_bytes : alg;
  return PMH_SUCCESS;
}

// Several GPUs, congruent blocks: E is built over ALL blocks of the decomposition (B = the global gluing) and every rank takes
// the super bands (128-row stripes) number idx = rank (mod size) of the size-ordered list -- an even share of the dense bytes whatever
// the blocks' sizes (one block per rank would leave a 1.36 x imbalance on configs[2]: n_Gamma 17 031 ... 24 384).  lambda is replicated
// and B u is all-reduced anyway, so a rank may apply any rows of any W_b: y_rank = Bhat (rows it owns of blockdiag(W_b)) Bhat' lambda
// and the all-reduce of pmh_gluing_mult_transpose completes F lambda.  Call before the assembly; the storage of the other ranks'
// stripes stays allocated (zero) so that no index changes (14 GB of 288 at configs[2]).
// the dealing rule, host only: super band sb of block b (n_Gamma padded to 128) goes to rank idx % size, idx counting the super bands
// of all blocks from the largest down (ties by block number).  owner[b][sb] and the dense bytes every rank ends up with.
static void fx_stripe_plan(int nb, const int *npad, int size, std::vector<std::vector<int>> &owner, std::vector<double> &bytes)
{
  int maxsb = 0;
  for (int b = 0; b < nb; b++) maxsb = std::max(maxsb, npad[b] / FX_TC);
  owner.assign(nb, std::vector<int>());
  for (int b = 0; b < nb; b++) owner[b].assign(npad[b] / FX_TC, -1);
  bytes.assign(size, 0.0);
  int idx = 0;
  for (int sb = maxsb - 1; sb >= 0; sb--)
    for (int b = 0; b < nb; b++)
      if (sb < npad[b] / FX_TC) {
        const int r  = idx++ % size;
        owner[b][sb] = r;
        bytes[r] += 8.0 * 4.0 * FX_TILE * (sb + 1);
      }
}

// host helper (no device): dense bytes per rank under pmh_fexplicit_set_stripe for blocks with the given n_Gamma -- the load balance
// of a multi-GPU run can be inspected (and tested) without a GPU
extern "C" int pmh_fexplicit_stripe_bytes(int nblocks, const int *n_gamma, int size, double *bytes_per_rank)
{
  PMH_ARG(nblocks >= 1 && n_gamma && size >= 1 && bytes_per_rank);
  std::vector<int> npad(nblocks);
  for (int b = 0; b < nblocks; b++) npad[b] = (n_gamma[b] + FX_TC - 1) / FX_TC * FX_TC;
  std::vector<std::vector<int>> owner;
  std::vector<double>           bytes;
  fx_stripe_plan(nblocks, npad.data(), size, owner, bytes);
  for (int r = 0; r < size; r++) bytes_per_rank[r] = bytes[r];
  return PMH_SUCCESS;
}

// host helper (no device): the owner rank of every 128-row stripe, block after block (ceil(n_Gamma_b / 128) entries per block)
extern "C" int pmh_fexplicit_stripe_owner(int nblocks, const int *n_gamma, int size, int *owner_out)
{
  PMH_ARG(nblocks >= 1 && n_gamma && size >= 1 && owner_out);
  std::vector<int> npad(nblocks);
  for (int b = 0; b < nblocks; b++) npad[b] = (n_gamma[b] + FX_TC - 1) / FX_TC * FX_TC;
  std::vector<std::vector<int>> owner;
  std::vector<double>           bytes;
  fx_stripe_plan(nblocks, npad.data(), size, owner, bytes);
  int k = 0;
  for (int b = 0; b < nblocks; b++)
    for (int o : owner[b]) owner_out[k++] = o;
  return PMH_SUCCESS;
}

// class-shared storages: the touched dofs of a class (the numbering of W_c's rows) and the set-up by symmetry (fshared.hip)
extern "C" int pmh_fexplicit_class_union(pmh_fexplicit E, int cls, int *n_c, int *urel_out)
{
  PMH_ARG(E);
  if (!E->sh) return pmh_set_error(PMH_ERR_SUP, "pmh_fexplicit_class_union: needs a class-shared storage");
  return fxs_class_union(E->sh, cls, n_c, urel_out);
}

extern "C" int pmh_fexplicit_set_class_symmetry(pmh_fexplicit E, int cls, int nsym, const int *posmap, const signed char *sign)
{
  PMH_ARG(E);
  if (E->assembled) return pmh_set_error(PMH_ERR_STATE, "pmh_fexplicit_set_class_symmetry: call before the assembly");
  if (!E->sh) return pmh_set_error(PMH_ERR_SUP, "pmh_fexplicit_set_class_symmetry: needs the PMH_FX_CLASS_SYM storage");
  return fxs_set_symmetry(E->sh, cls, nsym, posmap, sign);
}

// box-shaped blocks: the symmetries of the box that leave the class matrix invariant (pmh_box_symmetries: generators checked against the CSR of one
// block of the class, column indices relative to the block) and map the class's touched dofs onto themselves -> pmh_fexplicit_set_class_symmetry
extern "C" int pmh_fexplicit_set_box_symmetry(pmh_fexplicit E, int cls, const int *dims, int ndof, const int *rowptr, const int *col, const double *val, int *nsym_used)
{
  PMH_ARG(E && dims && ndof >= 1);
  if (nsym_used) *nsym_used = 1;
  if (!E->sh || (E->storage != PMH_FX_CLASS_SYM && E->storage != PMH_FX_CLASS_ORBIT)) return pmh_set_error(PMH_ERR_SUP, "pmh_fexplicit_set_box_symmetry: needs the PMH_FX_CLASS_SYM or PMH_FX_CLASS_ORBIT storage");
  int nc = 0;
  PMH_CHK(fxs_class_union(E->sh, cls, &nc, nullptr));
  if (nc == 0) return PMH_SUCCESS;
  std::vector<int> urel((size_t)nc);
  PMH_CHK(fxs_class_union(E->sh, cls, &nc, urel.data()));
  const long long n = (long long)dims[0] * dims[1] * dims[2] * ndof;
  if (urel.back() >= n) return pmh_set_error(PMH_ERR_ARG, "pmh_fexplicit_set_box_symmetry: the blocks of class %d have more than %d x %d x %d x %d dofs", cls, dims[0], dims[1], dims[2], ndof);
  std::vector<int>         perm((size_t)48 * n), pos((size_t)n, -1), pm;
  std::vector<signed char> sign((size_t)48 * n), sg;
  int                      nsym = 0;
  PMH_CHK(pmh_box_symmetries(dims, ndof, rowptr, col, val, 4000, &nsym, perm.data(), sign.data()));
  for (int i = 0; i < nc; i++) pos[urel[i]] = i;
  int used = 0;
  for (int g = 0; g < nsym; g++) { // keep the operations under which the touched set is closed
    bool closed = true;
    for (int i = 0; i < nc && closed; i++) closed = pos[perm[(size_t)g * n + urel[i]]] >= 0;
    if (!closed) continue;
    for (int i = 0; i < nc; i++) pm.push_back(pos[perm[(size_t)g * n + urel[i]]]), sg.push_back(sign[(size_t)g * n + urel[i]]);
    used++;
  }
  if (nsym_used) *nsym_used = used;
  if (used <= 1) return PMH_SUCCESS;
  if (E->assembled) return pmh_set_error(PMH_ERR_STATE, "pmh_fexplicit_set_box_symmetry: call before the assembly");
  return fxs_set_symmetry(E->sh, cls, used, pm.data(), sg.data());
}

extern "C" int pmh_fexplicit_apply_flops(pmh_fexplicit E, double *flops)
{
  PMH_ARG(E && flops);
  *flops = E->sh ? fxs_apply_flops(E->sh) : 0.0;
  return PMH_SUCCESS;
}

extern "C" int pmh_fexplicit_apply_flops_detail(pmh_fexplicit E, double *issued, double *dense)
{
  PMH_ARG(E && issued && dense);
  *issued = *dense = 0.0;
  if (E->sh) fxs_apply_flops_detail(E->sh, issued, dense);
  return PMH_SUCCESS;
}

extern "C" int pmh_fexplicit_set_stripe(pmh_fexplicit E, int rank, int size)
{
  PMH_ARG(E && size >= 1 && rank >= 0 && rank < size);
  if (E->assembled) return pmh_set_error(PMH_ERR_STATE, "pmh_fexplicit_set_stripe: call before the assembly");
  if (E->sh) { // class-shared storage: contiguous row ranges of every W_c
    E->stripe_rank = rank, E->stripe_size = size;
    return fxs_set_stripe(E->sh, rank, size);
  }
  if (E->storage != PMH_FX_SYM) return pmh_set_error(PMH_ERR_SUP, "pmh_fexplicit_set_stripe: striping needs the symmetric or the class-shared storage");
  pmh_ctx   ctx = E->ctx;
  const int nb  = E->nb;
  std::vector<std::vector<int>> owner;
  std::vector<double>           bytes;
  fx_stripe_plan(nb, E->ld.data(), size, owner, bytes);
  for (int b = 0; b < nb; b++)
    for (size_t sb = 0; sb < owner[b].size(); sb++) E->owned[b][sb] = owner[b][sb] == rank ? 1 : 0;
  E->stripe_rank = rank, E->stripe_size = size;
  (void)ctx;
  return fx_build_launch(E);
}

// host helper: classes of identical diagonal blocks of a block-diagonal CSR (same size, pattern and values, bit for bit)
extern "C" int pmh_csr_block_classes(int nblocks, const int *rowstart, const int *rowptr, const int *col, const double *val, int *block_class, int *nclasses)
{
  PMH_ARG(nblocks >= 0 && rowstart && rowptr && col && val && block_class);
  std::vector<int> reps;
  for (int b = 0; b < nblocks; b++) {
    const int r0 = rowstart[b], n = rowstart[b + 1] - r0, k0 = rowptr[r0], nnz = rowptr[r0 + n] - k0;
    int       cls = -1;
    for (size_t c = 0; c < reps.size() && cls < 0; c++) {
      const int a = reps[c], q0 = rowstart[a], m = rowstart[a + 1] - q0, j0 = rowptr[q0];
      if (m != n || rowptr[q0 + m] - j0 != nnz) continue;
      bool same = true;
      for (int i = 0; i <= n && same; i++) same = (rowptr[r0 + i] - k0) == (rowptr[q0 + i] - j0);
      if (same) { // columns and values: 12 bytes per entry of two blocks -- 0.13 s on one thread for the 8 cubes of configs[2], and the set-up asks twice
        const int         nt = std::max(1, std::min({pmh_host_threads(), nnz / (1 << 20) + 1}));
        std::vector<char> eq((size_t)nt, 1);
        auto              cmp = [&](int t) {
          const int a0 = (int)((long long)nnz * t / nt), a1 = (int)((long long)nnz * (t + 1) / nt);
          bool      ok = true;
          for (int k = a0; k < a1 && ok; k++) ok = (col[k0 + k] - r0) == (col[j0 + k] - q0);
          if (ok) ok = memcmp(val + k0 + a0, val + j0 + a0, sizeof(double) * (size_t)(a1 - a0)) == 0;
          eq[t] = ok ? 1 : 0;
        };
        if (nt == 1) cmp(0);
        else {
          std::vector<std::thread> th;
          for (int t = 0; t < nt; t++) th.emplace_back(cmp, t);
          for (auto &x : th) x.join();
        }
        for (int t = 0; t < nt; t++) same = same && eq[t];
      }
      if (same) cls = (int)c;
    }
    if (cls < 0) {
      cls = (int)reps.size();
      reps.push_back(b);
    }
    block_class[b] = cls;
  }
  if (nclasses) *nclasses = (int)reps.size();
  return PMH_SUCCESS;
}

// row p of block b belongs to this rank's stripe (always, without striping)
static inline bool fx_owns_row(pmh_fexplicit E, int b, int p) { return E->stripe_size <= 1 || E->storage != PMH_FX_SYM || E->owned[b][p / FX_TC]; }

// ---- assembly -------------------------------------------------------------------------------------------------------------
// MatInvExplicitly_Private (matinv.c:640-665: KSPSolve on the columns of the identity, one row of the explicit matrix per solve),
// restricted to the columns / rows in Gamma and batched over the slots of `solver`.
// slot_class[s]: class of the matrix in slot s of the solver; block_class[b]: class of block b of this operator (blocks of one
// class have identical K_b, so K_b^+ e_j serves all of them).  NULL, NULL: slot s <-> block s (the solver is the operator's own K^+).
extern "C" int pmh_fexplicit_assemble(pmh_fexplicit E, pmh_matinv solver, int nslots, const int *slot_class, const int *block_class, double rtol, int max_it)
{
  PMH_ARG(E && solver && nslots >= 1 && (solver->nblocks == nslots || solver->nblocks * PMH_MV_R == nslots));
  PMH_ARG((slot_class && block_class) || (!slot_class && !block_class && (nslots == E->nb || nslots == E->nb * PMH_MV_R)));
  if (E->sh) { // class-shared storage: one full row of W_c per solve (the classes are those given at creation)
    PMH_ARG(slot_class);
    auto t0s = std::chrono::steady_clock::now();
    PMH_CHK(fxs_assemble(E->sh, solver, nslots, slot_class, rtol, max_it, &E->n_solves));
    E->assembled = 1;
    E->assemble_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0s).count();
    return PMH_SUCCESS;
  }
  pmh_ctx   ctx = E->ctx;
  const int nb  = E->nb;
  auto      t0  = std::chrono::steady_clock::now();
  std::vector<int> sc(nslots), bc(nb);
  for (int s = 0; s < nslots; s++) sc[s] = slot_class ? slot_class[s] : s / (nslots / nb); // (no classes given: slot s is block s, or column s % R of block s / R)
  for (int b = 0; b < nb; b++) bc[b] = block_class ? block_class[b] : b;
  int ncls = 0;
  for (int b = 0; b < nb; b++) ncls = std::max(ncls, bc[b] + 1);
  pmh_asm_solver A;
  struct closer {
    pmh_asm_solver &a;
    ~closer() { a.close(); }
  } closer_{A};
  PMH_CHK(A.open(solver, nslots));
  // per class: its slots, its blocks, the union of the blocks' relative Gamma indices
  std::vector<std::vector<int>> cslots(ncls), cblocks(ncls), cunion(ncls);
  for (int s = 0; s < nslots; s++)
    if (sc[s] >= 0 && sc[s] < ncls) cslots[sc[s]].push_back(s);
  for (int b = 0; b < nb; b++) {
    PMH_ARG(bc[b] >= 0);
    cblocks[bc[b]].push_back(b);
  }
  for (int c = 0; c < ncls; c++) {
    if (cblocks[c].empty()) continue;
    if (cslots[c].empty()) return pmh_set_error(PMH_ERR_ARG, "pmh_fexplicit_assemble: no solver slot for block class %d", c);
    const int nloc = E->K->rowstart[cblocks[c][0] + 1] - E->K->rowstart[cblocks[c][0]];
    for (int b : cblocks[c])
      if (E->K->rowstart[b + 1] - E->K->rowstart[b] != nloc) return pmh_set_error(PMH_ERR_ARG, "pmh_fexplicit_assemble: blocks of class %d differ in size", c);
    for (int s : cslots[c])
      if (A.rows(s) != nloc) return pmh_set_error(PMH_ERR_ARG, "pmh_fexplicit_assemble: slot %d has %d rows, class %d blocks have %d", s, A.rows(s), c, nloc);
    std::vector<char> in((size_t)std::max(1, nloc), 0);
    for (int b : cblocks[c])
      for (size_t k = E->goff[b]; k < E->goff[b + 1]; k++)
        if (fx_owns_row(E, b, (int)(k - E->goff[b]))) in[E->gamma[k] - E->K->rowstart[b]] = 1;
    for (int i = 0; i < nloc; i++)
      if (in[i]) cunion[c].push_back(i);
  }
  // position of a relative dof inside Gamma_b (-1: not in it)
  std::vector<std::vector<int>> pos(nb);
  for (int b = 0; b < nb; b++) {
    pos[b].assign((size_t)std::max(1, E->K->rowstart[b + 1] - E->K->rowstart[b]), -1);
    for (size_t k = E->goff[b]; k < E->goff[b + 1]; k++) pos[b][E->gamma[k] - E->K->rowstart[b]] = (int)(k - E->goff[b]);
  }
  int nbatch = 0;
  for (int c = 0; c < ncls; c++)
    if (!cblocks[c].empty()) nbatch = std::max(nbatch, (int)((cunion[c].size() + cslots[c].size() - 1) / cslots[c].size()));
  double *rhs, *sol;
  int    *d_idx, *h_idx;
  const size_t nsol = A.len();
  PMH_CHK(pmh_malloc(ctx, sizeof(double) * nsol, (void **)&rhs));
  PMH_CHK(pmh_malloc(ctx, sizeof(double) * nsol, (void **)&sol));
  PMH_CHK(pmh_malloc(ctx, sizeof(int) * nslots, (void **)&d_idx));
  PMH_HIP(hipHostMalloc((void **)&h_idx, sizeof(int) * nslots * 2, hipHostMallocDefault));
  PMH_CHK(pmh_memset(ctx, rhs, 0, sizeof(double) * nsol));
  double old_rtol, old_atol;
  int    old_maxit;
  PMH_CHK(pmh_matinv_get_tolerances(solver, &old_rtol, &old_atol, &old_maxit));
  PMH_CHK(pmh_matinv_set_tolerances(solver, rtol, 1e-300, max_it > 0 ? max_it : old_maxit));
  int        rc = PMH_SUCCESS;
  std::vector<int> col(nslots);
  const bool progress = getenv("PMH_PROGRESS") != nullptr || getenv("PMH_CONTACT_TIMING") != nullptr; // (set-up only: a long assembly says where it is every ~ 20 s)
  auto       t_prog   = std::chrono::steady_clock::now();
  for (int k = 0; k < nbatch && !rc; k++) {
    if (progress && std::chrono::duration<double>(std::chrono::steady_clock::now() - t_prog).count() > 20.0) {
      t_prog = std::chrono::steady_clock::now();
      fprintf(stderr, "  pmh_fexplicit_assemble: batch %d of %d (%d columns per batch), %.0f s\n", k, nbatch, nslots, std::chrono::duration<double>(t_prog - t0).count());
      fflush(stderr);
    }
    int *hh = h_idx + (k & 1) * nslots; // pinned staging, alternating halves (every K^+ application synchronises the stream at least once)
    for (int s = 0; s < nslots; s++) hh[s] = -1, col[s] = -1;
    for (int c = 0; c < ncls; c++) {
      if (cblocks[c].empty()) continue;
      for (size_t t = 0; t < cslots[c].size(); t++) {
        const size_t j = (size_t)k * cslots[c].size() + t;
        if (j < cunion[c].size()) {
          const int s = cslots[c][t];
          col[s]      = cunion[c][j];
          hh[s]       = A.rhs_index(s, col[s]);
        }
      }
    }
    if (hipMemcpyAsync(d_idx, hh, sizeof(int) * nslots, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
      rc = pmh_set_error(PMH_ERR_HIP, "pmh_fexplicit_assemble: index upload failed");
      break;
    }
    hipLaunchKernelGGL(k_fx_set_entries, dim3((nslots + 63) / 64), dim3(64), 0, ctx->stream, nslots, (const int *)d_idx, 1.0, rhs);
    if ((rc = A.solve(rhs, sol))) break;
    if (A.hit_the_limit()) { // a column that is not converged would silently make F inexact
      rc = pmh_set_error(PMH_ERR_STATE, "pmh_fexplicit_assemble: a set-up solve of batch %d did not reach rtol %.1e within %d iterations of the inner KSP", k, rtol, solver->max_it);
      break;
    }
    hipLaunchKernelGGL(k_fx_set_entries, dim3((nslots + 63) / 64), dim3(64), 0, ctx->stream, nslots, (const int *)d_idx, 0.0, rhs);
    for (int s = 0; s < nslots; s++) {
      if (col[s] < 0) continue;
      E->n_solves++;
      for (int b : cblocks[sc[s]]) {
        const int p = pos[b][col[s]];
        if (p < 0 || !fx_owns_row(E, b, p)) continue;
        // row p of W_b: FULL all n_Gamma columns; SYM the columns of its band (0 .. 32(k+1)-1, capped at n_Gamma: the rest is padding)
        const int       kb   = p / FX_RB;
        const int       n    = (E->storage == PMH_FX_SYM) ? std::min(E->ngam[b], FX_RB * (kb + 1)) : E->ngam[b];
        const dim3      grid(std::max(1, std::min(64, (n + PMH_BLOCK - 1) / PMH_BLOCK)));
        if (E->storage == PMH_FX_SYM)
          hipLaunchKernelGGL(k_fx_extract_tiled, grid, dim3(PMH_BLOCK), 0, ctx->stream, n, (const int *)(E->d_gamma_rel + E->goff[b]), A.sol_of(sol, s), E->W[b] + fx_band_off(kb), p - FX_RB * kb);
        else
          hipLaunchKernelGGL(k_fx_extract, grid, dim3(PMH_BLOCK), 0, ctx->stream, n, (const int *)(E->d_gamma_rel + E->goff[b]), A.sol_of(sol, s), E->W[b] + (long long)p * E->ld[b]);
      }
    }
    if (hipGetLastError() != hipSuccess) rc = pmh_set_error(PMH_ERR_HIP, "pmh_fexplicit_assemble: launch failed in batch %d", k);
  }
  if (!rc) rc = pmh_sync(ctx);
  pmh_matinv_set_tolerances(solver, old_rtol, old_atol, old_maxit);
  pmh_free(ctx, rhs), pmh_free(ctx, sol), pmh_free(ctx, d_idx);
  (void)hipHostFree(h_idx);
  if (rc) return rc;
  E->assembled = 1;
  E->assemble_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return PMH_SUCCESS;
}

// The set-up with 8 columns per block and application where the multi-right-hand-side K^+ applies to `solver` (matinv_mv.hip), one column per block otherwise
// (PMH_ERR_SUP of the first attempt is not an error here).  slot_class: class of every BLOCK of the solver (or NULL with block_class NULL: block s of the operator).
extern "C" int pmh_fexplicit_assemble_auto(pmh_fexplicit E, pmh_matinv solver, const int *slot_class, const int *block_class, double rtol, int max_it, int *used_multi_rhs)
{
  PMH_ARG(E && solver);
  if (used_multi_rhs) *used_multi_rhs = 0;
  if (pmh_knobs().multi_rhs) {
    const int        nb = solver->nblocks;
    std::vector<int> sc;
    if (slot_class) {
      sc.resize((size_t)nb * PMH_MV_R);
      for (int s = 0; s < nb * PMH_MV_R; s++) sc[s] = slot_class[s / PMH_MV_R];
    }
    const int rc = pmh_fexplicit_assemble(E, solver, nb * PMH_MV_R, slot_class ? sc.data() : nullptr, block_class, rtol, max_it);
    if (rc != PMH_ERR_SUP) {
      if (!rc && used_multi_rhs) *used_multi_rhs = 1;
      return rc;
    }
  }
  return pmh_fexplicit_assemble(E, solver, solver->nblocks, slot_class, block_class, rtol, max_it);
}

// kernel-tuning helper (scripts/symv_tune.py): fills the dense storage with a byte pattern instead of assembling it, so that the
// apply kernels can be timed at full size without the set-up solves.  The operator is NOT F afterwards.
extern "C" int pmh_fexplicit_fill_pattern(pmh_fexplicit E, int byte)
{
  PMH_ARG(E);
  if (E->sh) {
    E->assembled = 1;
    return fxs_fill_pattern(E->sh, byte);
  }
  long long tot = 0;
  for (int b = 0; b < E->nb; b++) tot += (E->storage == PMH_FX_SYM) ? fx_sym_size(E->ld[b]) : (long long)E->ld[b] * E->ld[b];
  PMH_HIP(hipMemsetAsync(E->Wbase, byte, sizeof(double) * (size_t)tot, E->ctx->stream));
  E->assembled = 1;
  return pmh_sync(E->ctx);
}

extern "C" int pmh_fexplicit_assemble_stats(pmh_fexplicit E, long long *n_solves, double *seconds)
{
  PMH_ARG(E);
  if (n_solves) *n_solves = E->n_solves;
  if (seconds) *seconds = E->assemble_seconds;
  return PMH_SUCCESS;
}

// copies W_b to the host (tests, post-processing): out is n_Gamma_b x n_Gamma_b row-major
extern "C" int pmh_fexplicit_get_block(pmh_fexplicit E, int b, double *out_host, int *gamma_host)
{
  PMH_ARG(E && b >= 0 && b < E->nb);
  const int n = E->ngam[b];
  if (out_host && n && E->sh) {
    PMH_CHK(fxs_get_block(E->sh, b, n, &E->gamma[E->goff[b]], out_host));
  } else if (out_host && n) {
    if (E->storage == PMH_FX_FULL) {
      PMH_HIP(hipMemcpy2D(out_host, sizeof(double) * n, E->W[b], sizeof(double) * E->ld[b], sizeof(double) * n, n, hipMemcpyDeviceToHost));
    } else { // unpack the lower block-triangle band by band, mirror it (the diagonal tile is stored in full)
      std::vector<double> band;
      for (int k = 0; k * FX_RB < n; k++) {
        const int ntile = k / 4 + 1, rows = std::min(FX_RB, n - k * FX_RB), ncol = std::min(FX_RB * (k + 1), n);
        band.resize((size_t)FX_TILE * ntile);
        PMH_HIP(hipMemcpy(band.data(), E->W[b] + fx_band_off(k), sizeof(double) * band.size(), hipMemcpyDeviceToHost));
        for (int r = 0; r < rows; r++) {
          const int j = k * FX_RB + r;
          for (int c = 0; c < ncol; c++) {
            const double v = band[(size_t)(c / FX_TC) * FX_TILE + (size_t)r * FX_TC + (c % FX_TC)];
            out_host[(size_t)j * n + c] = v;
            if (c < k * FX_RB) out_host[(size_t)c * n + j] = v;
          }
        }
      }
    }
  }
  if (gamma_host)
    for (int i = 0; i < n; i++) gamma_host[i] = E->gamma[E->goff[b] + i];
  return PMH_SUCCESS;
}

// ---- apply ----------------------------------------------------------------------------------------------------------------

static int fx_gemv(pmh_fexplicit E)
{
  if (!E->nwg) return PMH_SUCCESS;
  if (E->storage == PMH_FX_SYM) {
    hipStream_t st    = E->ctx->stream;
    bool        timed = false;
    if (E->ev_on && (E->ev_seen++ % E->ev_stride) == 0 && (size_t)(2 * E->ev_used + 2) <= E->ev.size() && (size_t)E->ev_used < E->ev_mid.size()) {
      timed = true;
      PMH_HIP(hipEventRecord(E->ev[2 * E->ev_used], st));
    }
#define SYMV_LAUNCH(V) hipLaunchKernelGGL(k_fx_symv<V>, dim3(E->nsw), dim3(PMH_BLOCK), 0, st, (const int *)E->d_sw_block, (const int *)E->d_sw_band, (const int *)E->d_sw_seg, (const int *)E->d_gstart, (const int *)E->d_ld, (const long long *)E->d_woff, (const long long *)E->d_poff, (const long long *)E->d_doff, (const double *)E->Wbase, (const double *)E->xh, E->ydir, E->partial, E->seg)
    SYMV_LAUNCH(0);
#undef SYMV_LAUNCH
    if (timed) PMH_HIP(hipEventRecord(E->ev_mid[E->ev_used], st));
    if (E->nfw)
      hipLaunchKernelGGL(k_fx_symv_fin, dim3(E->nfw), dim3(PMH_BLOCK), 0, st, (const int *)E->d_fw_block, (const int *)E->d_fw_col0, (const int *)E->d_gstart, (const int *)E->d_ld, (const long long *)E->d_poff,
                         (const long long *)E->d_doff, (const int *)E->d_own_ptr, (const int *)E->d_own_list, (const double *)E->ydir, (const double *)E->partial, E->yh, E->seg);
    if (timed) {
      PMH_HIP(hipEventRecord(E->ev[2 * E->ev_used + 1], st));
      E->ev_used++;
    }
    PMH_HIP(hipGetLastError());
    return PMH_SUCCESS;
  }
  hipStream_t st    = E->ctx->stream;
  bool        timed = false;
  if (E->ev_on && (E->ev_seen++ % E->ev_stride) == 0 && (size_t)(2 * E->ev_used + 2) <= E->ev.size()) {
    timed = true;
    PMH_HIP(hipEventRecord(E->ev[2 * E->ev_used], st));
  }
#define FX_LAUNCH(RW) hipLaunchKernelGGL(k_fx_gemv<RW>, dim3(E->nwg), dim3(PMH_BLOCK), 0, st, (const int *)E->d_wg_block, (const int *)E->d_wg_row0, (const int *)E->d_gstart, (const int *)E->d_ngam, (const int *)E->d_ld, (const long long *)E->d_woff, (const double *)E->Wbase, (const double *)E->xh, E->yh)
  if (E->rw == 8) FX_LAUNCH(8);
  else if (E->rw == 2) FX_LAUNCH(2);
  else FX_LAUNCH(4);
#undef FX_LAUNCH
  if (timed) {
    PMH_HIP(hipEventRecord(E->ev[2 * E->ev_used + 1], st));
    E->ev_used++;
  }
  PMH_HIP(hipGetLastError());
  return PMH_SUCCESS;
}

// E serves F = B K^+ B' when it was built from B itself, or -- striped over several GPUs -- from the global gluing with the same dual space
bool pmh_fexplicit_matches(pmh_fexplicit_s *E, pmh_gluing B) { return E && E->assembled && (E->B == B || (E->stripe_size >= 1 && E->B->n_lambda == B->n_lambda)); }

// y = F lambda = Bhat W Bhat' lambda (MatMult of the product F = B K^+ B', qptransform.c:1103-1128, with K^+ explicit)
int pmh_fexplicit_apply(pmh_fexplicit_s *E, const double *lambda, double *y)
{
  if (E->sh) return fxs_apply(E->sh, lambda, y);
  PMH_CHK(pmh_gluing_mult(E->Bhat, lambda, E->xh));
  PMH_CHK(fx_gemv(E));
  return pmh_gluing_mult_transpose(E->Bhat, E->yh, y); // ends with the all-reduce on several GPUs
}

int pmh_fexplicit_stages(pmh_fexplicit_s *E, pmh_csr *gather, double **mid_in, pmh_csr *scatter, const double **mid_out)
{
  if (E->sh) return fxs_stages(E->sh, gather, mid_in, scatter, mid_out);
  *gather = E->Bhat->Bt, *mid_in = E->xh, *scatter = E->Bhat->B, *mid_out = E->yh;
  return PMH_SUCCESS;
}
int pmh_fexplicit_mid(pmh_fexplicit_s *E) { return E->sh ? fxs_mid(E->sh) : fx_gemv(E); }

extern "C" int pmh_fexplicit_mult(pmh_fexplicit E, const double *lambda, double *y)
{
  PMH_ARG(E && lambda && y);
  if (!E->assembled) return pmh_set_error(PMH_ERR_STATE, "pmh_fexplicit_mult: the explicit blocks are not assembled yet");
  return pmh_fexplicit_apply(E, lambda, y);
}

// the dense kernel alone on the compressed vectors (tests, bench): yh = blockdiag(W_b) xh, both of length n_compressed
extern "C" int pmh_fexplicit_dense_mult(pmh_fexplicit E, const double *xh, double *yh)
{
  PMH_ARG(E && xh && yh);
  if (E->sh) { // vectors in the multivector numbering (pmh_fexplicit_compressed_size)
    const size_t nbytes = sizeof(double) * (size_t)fxs_multivector_length(E->sh);
    PMH_CHK(pmh_memcpy_d2d(E->ctx, fxs_X(E->sh), xh, nbytes));
    PMH_CHK(fxs_dense(E->sh));
    return pmh_memcpy_d2d(E->ctx, yh, fxs_Y(E->sh), nbytes);
  }
  PMH_CHK(pmh_memcpy_d2d(E->ctx, E->xh, xh, sizeof(double) * (size_t)E->ntot));
  PMH_CHK(fx_gemv(E));
  return pmh_memcpy_d2d(E->ctx, yh, E->yh, sizeof(double) * (size_t)E->ntot);
}

extern "C" int pmh_fexplicit_compressed_size(pmh_fexplicit E, int *ntot, int *gstart /* [nblocks+1] or NULL */)
{
  PMH_ARG(E);
  if (E->sh) {
    if (ntot) *ntot = (int)fxs_multivector_length(E->sh);
    if (gstart)
      for (int b = 0; b <= E->nb; b++) gstart[b] = -1; // no per-block layout in the multivector numbering
    return PMH_SUCCESS;
  }
  if (ntot) *ntot = E->ntot;
  if (gstart)
    for (int b = 0; b <= E->nb; b++) gstart[b] = E->gstart[b];
  return PMH_SUCCESS;
}

// MATINV with an explicit inverse attached: F = B K^+ B' built on this K^+ (pmh_op_create_feti_dual) applies through E
// whenever E was built from the same B; NULL detaches.  K^+ f for a general f (d = B K^+ f - c, the primal recovery) stays iterative.
extern "C" int pmh_matinv_attach_explicit(pmh_matinv Kplus, pmh_fexplicit E)
{
  PMH_ARG(Kplus);
  if (E && !E->assembled) return pmh_set_error(PMH_ERR_STATE, "pmh_matinv_attach_explicit: assemble the explicit blocks first");
  if (E && E->stripe_size < 1 && E->K->n != Kplus->n) return pmh_set_error(PMH_ERR_ARG, "pmh_matinv_attach_explicit: size mismatch (%d vs %d)", E->K->n, Kplus->n);
  Kplus->E = E;
  return PMH_SUCCESS;
}

extern "C" int pmh_fexplicit_timing_enable(pmh_fexplicit E, int max_launches, int stride)
{
  PMH_ARG(E && max_launches >= 0);
  if (E->sh) return fxs_timing_enable(E->sh, max_launches);
  while ((int)E->ev.size() < 2 * max_launches) {
    hipEvent_t e;
    PMH_HIP(hipEventCreate(&e));
    E->ev.push_back(e);
  }
  while ((int)E->ev_mid.size() < max_launches) {
    hipEvent_t e;
    PMH_HIP(hipEventCreate(&e));
    E->ev_mid.push_back(e);
  }
  E->ev_on = max_launches > 0, E->ev_used = 0, E->ev_seen = 0, E->ev_stride = std::max(1, stride);
  return PMH_SUCCESS;
}

extern "C" int pmh_fexplicit_timing_get(pmh_fexplicit E, int *launches, double *total_ms, double *first_kernel_ms)
{
  PMH_ARG(E && launches && total_ms);
  if (E->sh) {
    return fxs_timing_get(E->sh, launches, total_ms, first_kernel_ms); // orbit storage: first = the GEMM kernel alone
  }
  PMH_CHK(pmh_sync(E->ctx));
  double tot = 0.0, first = 0.0;
  for (int i = 0; i < E->ev_used; i++) {
    float ms = 0.f;
    PMH_HIP(hipEventElapsedTime(&ms, E->ev[2 * i], E->ev[2 * i + 1]));
    tot += ms;
    if (E->storage == PMH_FX_SYM) {
      PMH_HIP(hipEventElapsedTime(&ms, E->ev[2 * i], E->ev_mid[i]));
      first += ms;
    }
  }
  *launches = E->ev_used, *total_ms = tot;
  if (first_kernel_ms) *first_kernel_ms = (E->storage == PMH_FX_SYM) ? first : tot; // SYM: k_fx_symv alone (the rest is k_fx_symv_fin)
  return PMH_SUCCESS;
}
