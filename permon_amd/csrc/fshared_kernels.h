// Device kernels of the class-shared explicit local dual operators (included by fshared.hip only): the full-matrix product k_fxs_gemm8, the symmetric tile product
// k_fxs_symm8 / k_fxs_symfin, the orbit GEMM k_fxo_gemm16 with its finishing kernel k_fxo_fin, and the set-up helpers.
#pragma once
#include "fshared_types.h"
#include "reduce.h"


// Y = W_c X with 8 right-hand sides, W_c symmetric and stored in full: the product is taken as Y[c][s] = sum_r W[r][c] X[r][s], i.e.
// the lane OWNS its column pair (c, c+1) for the output and walks down the rows -- every load of a wave is one contiguous 1 KB piece of
// a row, the 8 values X[r][.] of the row are uniform across the wave (staged in LDS 128 rows at a time and read as broadcasts: scalar
// loads of them serialised on their latency, measured), the 16 sums stay in the lane's registers and no
// reduction across lanes is ever needed (this is the "transposed" half of the symmetric kernel; with both triangles stored it is all
// there is).  A workgroup owns 4 adjacent 128-column chunks (one per wave) x one segment of the rank's rows; the segment sums go to
// part[segment][c][s] and k_fxs_fin adds them in segment order.  16 rows (16 KB per wave) are in flight per trip.
#define FXS_U 16
// the dense product has no reference summation order to reproduce (W_c is exact K^+ data): fused multiply-adds, still a fixed order
#define FXS_MAD(a, b, c) __builtin_fma((a), (b), (c))
#define FXS_XB 128 // rows of X staged in LDS per step (8 KB, double-buffered: one barrier per 128 rows)
__global__ __launch_bounds__(PMH_BLOCK) void k_fxs_gemm8(const int *__restrict__ wg, const int *__restrict__ c_ld, const long long *__restrict__ c_woff, const long long *__restrict__ c_xoff,
                                                        const double *__restrict__ Wbase, const double *__restrict__ X, double *__restrict__ part, long long part_stride)
{
  // wg: (class, group, first column of the workgroup's 512, segment index, first row, one-past-last row) per workgroup
  __shared__ double xs[2][FXS_XB * FXS_S];
  const int *w6 = wg + 6 * blockIdx.x;
  const int  c = __builtin_amdgcn_readfirstlane(w6[0]), g = __builtin_amdgcn_readfirstlane(w6[1]), seg = __builtin_amdgcn_readfirstlane(w6[3]);
  const int  rlo = __builtin_amdgcn_readfirstlane(w6[4]), rhi = __builtin_amdgcn_readfirstlane(w6[5]);
  const int  lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int  ld = c_ld[c], col = w6[2] + wave * 128 + lane * 2;
  const bool active = (w6[2] + wave * 128) < ld; // the last workgroup of a row of chunks may have idle waves: they still stage X and join the barriers
  const double *__restrict__ A = Wbase + c_woff[c] + (active ? col : 0);
  const double *__restrict__ x = X + c_xoff[c] + (long long)g * ld * FXS_S;
  double acc0[FXS_S], acc1[FXS_S];
#pragma unroll
  for (int s = 0; s < FXS_S; s++) acc0[s] = acc1[s] = 0.0;
  int it = 0;
  for (int blk = rlo; blk < rhi; blk += FXS_XB, it++) {
    const int nrows = min(FXS_XB, rhi - blk);
    double   *xb    = xs[it & 1];
    // the X values of these rows (uniform across the lanes of the product below): 256 threads x 4 doubles, zero past the segment's end
    {
      const int  i0 = threadIdx.x * 4, row = i0 / FXS_S;
      const dbl2 z  = {0.0, 0.0};
      const dbl2 v0 = row < nrows ? *(const dbl2 *)(x + (long long)blk * FXS_S + i0) : z, v1 = row < nrows ? *(const dbl2 *)(x + (long long)blk * FXS_S + i0 + 2) : z;
      *(dbl2 *)(xb + i0)     = v0;
      *(dbl2 *)(xb + i0 + 2) = v1;
    }
    __syncthreads();
    if (active) {
      for (int r = 0; r < nrows; r += FXS_U) {
        dbl2 a[FXS_U];
#pragma unroll
        // rows past the end: X is zero there
        for (int u = 0; u < FXS_U; u++) a[u] = __builtin_nontemporal_load((const dbl2 *)(A + (long long)min(blk + r + u, rhi - 1) * ld));
#pragma unroll
        for (int u = 0; u < FXS_U; u++) {
          const dbl2 *xr = (const dbl2 *)(xb + (r + u) * FXS_S); // same address in every lane: LDS broadcast
          const dbl2  x01 = xr[0], x23 = xr[1], x45 = xr[2], x67 = xr[3];
          acc0[0] = FXS_MAD(a[u].x, x01.x, acc0[0]), acc1[0] = FXS_MAD(a[u].y, x01.x, acc1[0]);
          acc0[1] = FXS_MAD(a[u].x, x01.y, acc0[1]), acc1[1] = FXS_MAD(a[u].y, x01.y, acc1[1]);
          acc0[2] = FXS_MAD(a[u].x, x23.x, acc0[2]), acc1[2] = FXS_MAD(a[u].y, x23.x, acc1[2]);
          acc0[3] = FXS_MAD(a[u].x, x23.y, acc0[3]), acc1[3] = FXS_MAD(a[u].y, x23.y, acc1[3]);
          acc0[4] = FXS_MAD(a[u].x, x45.x, acc0[4]), acc1[4] = FXS_MAD(a[u].y, x45.x, acc1[4]);
          acc0[5] = FXS_MAD(a[u].x, x45.y, acc0[5]), acc1[5] = FXS_MAD(a[u].y, x45.y, acc1[5]);
          acc0[6] = FXS_MAD(a[u].x, x67.x, acc0[6]), acc1[6] = FXS_MAD(a[u].y, x67.x, acc1[6]);
          acc0[7] = FXS_MAD(a[u].x, x67.y, acc0[7]), acc1[7] = FXS_MAD(a[u].y, x67.y, acc1[7]);
        }
      }
    }
  }
  if (!active) return;
  // part[seg][(xoff + g ld 8) + col 8 + s]: 16 consecutive doubles per lane
  double *__restrict__ p = part + (long long)seg * part_stride + c_xoff[c] + (long long)g * ld * FXS_S + (long long)col * FXS_S;
#pragma unroll
  for (int q = 0; q < FXS_S / 2; q++) {
    *(dbl2 *)(p + 2 * q)         = dbl2{acc0[2 * q], acc0[2 * q + 1]};
    *(dbl2 *)(p + FXS_S + 2 * q) = dbl2{acc1[2 * q], acc1[2 * q + 1]};
  }
}

// Y[i] = sum over the segments of part[segment][i], in segment order (i runs over the whole multivector)
__global__ __launch_bounds__(PMH_BLOCK) void k_fxs_fin(long long n, int nseg, long long part_stride, const double *__restrict__ part, double *__restrict__ Y)
{
  const long long i = 2 * ((long long)blockIdx.x * PMH_BLOCK + threadIdx.x);
  if (i >= n) return;
  dbl2 s = *(const dbl2 *)(part + i);
  int  j = 1;
  for (; j + 8 <= nseg; j += 8) { // segment order kept, 8 loads in flight (a plain loop waits for every load before its add)
    dbl2 v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = *(const dbl2 *)(part + (long long)(j + k) * part_stride + i);
#pragma unroll
    for (int k = 0; k < 8; k++) s += v[k];
  }
  for (; j < nseg; j++) s += *(const dbl2 *)(part + (long long)j * part_stride + i);
  *(dbl2 *)(Y + i) = s;
}

__global__ __launch_bounds__(PMH_BLOCK) void k_fxs_extract(int n, const int *__restrict__ urel, const double *__restrict__ u, double *__restrict__ wrow)
{
  for (int i = blockIdx.x * PMH_BLOCK + threadIdx.x; i < n; i += gridDim.x * PMH_BLOCK) wrow[i] = u[urel[i]];
}

__global__ void k_fxs_set_entries(int m, const int *__restrict__ idx, double val, double *__restrict__ rhs)
{
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < m && idx[s] >= 0) rhs[idx[s]] = val;
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// Symmetric tile storage (PMH_FX_CLASS_SYM): W_c = W_c' kept as its lower block-triangle, HALF the bytes of the full storage above.
// With 8 right-hand sides every stored entry now feeds 16 multiply-adds (Y_I += W_IJ X_J and Y_J += W_IJ' X_I): 4 flop per byte, which a
// kernel with per-lane accumulators cannot organise without a reduction across lanes for one of the two products.  The fp64 matrix
// instruction v_mfma_f64_4x4x4_4b_f64 can: one instruction = four independent 4x4x4 products, i.e. 16 rows x 4 k x 4 right-hand sides,
// at the full fp64 rate (measured 72 TFLOP/s, scripts/micro/mfma_f64.hip; the 16x16x4 shape would waste half of its 16 columns on 8
// right-hand sides AND measured 44 TFLOP/s).  Operand maps (measured, same file): A lane l = A_b[i = l&3][k = l>>4] of block b = (l>>2)&3,
// B lane l = B_b[k = l>>4][j = l&3], D lane l = D_b[i = l>>4][j = l&3].
//
// Layout: rows in super bands of 256 (16 row tiles); super band sb holds, for every column tile J = 0 .. 16 (sb + 1) - 1 and row tile
// I = 0 .. 15, the 16 x 16 tile (sb, I, J) as 2 KB, column tile after column tile -- a wave streams 32 KB contiguous per column tile.
// Inside the square diagonal block the tiles above the diagonal are zero and the diagonal tiles keep their strict lower triangle plus HALF
// their diagonal, so that the kernel treats every tile alike (direct + transposed product) with no branch: L' X + L'' X = W X.
// Element (r, c) of a tile sits at double index (q >> 1) * 128 + 2 * l + (q & 1) with q = r >> 2, l = 16 (r & 3) + c: two 16-byte loads per
// lane give the four A operands of the transposed product (row group q, lane l <-> k = row & 3, column c) with no shuffling.  The direct
// product needs the transposed lane map: the tile goes through a wave-private 2 KB LDS image (rotation-swizzled, conflict-free both ways).
#define FXM_RT 16
#define FXM_RS 256
#define FXM_NB 8 // tiles in flight per wave (16 KB)
#define FXM_MB 4 // super bands per mega band = per workgroup: the transposed sums of 1024 rows are combined on chip before they are written
#define FXM_THREADS 512
static __device__ __forceinline__ double fxm_mfma(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

// Persistent grid, one workgroup of 8 waves per CU, each with an equal run of work: items = (class, group, mega band m = super bands
// 4m .. 4m+3, column tiles [jbeg, jend)).  Wave w works on super band 4m + (w >> 1) and the column tiles jbeg + (w & 1), + 2, ... -- the
// eight waves walk the column tiles in lock step (one barrier per pair), each down the 16 row tiles of its super band.  Per tile: 8 MFMA
// for Y_J += W_IJ' X_I (accumulated over the 16 row tiles in 2 registers; the four super bands' sums of a column tile are then added in
// LDS, in super band order, and stored as ONE 1 KB partial sum per (mega band, column tile): measured, the HBM writes of these partial
// sums are what limits the kernel -- with one per 256 rows 3 % of the bytes cost 10-18 % of the time) and 8 MFMA for Y_I += W_IJ X_J (32
// accumulators per lane for the 16 row tiles, kept for the whole item and stored once per item).  X of the mega band's rows is staged in
// LDS once per item; X of the column tile is fetched one tile ahead.  Every sum has a fixed order => bitwise reproducible.
__global__ __launch_bounds__(FXM_THREADS, 1) void k_fxs_symm8(const int *__restrict__ wg_first, const int *__restrict__ items, const long long *__restrict__ iteml, const int *__restrict__ c_ld,
                                                              const long long *__restrict__ c_xoff, const double *__restrict__ Wbase, const double *__restrict__ X, double *__restrict__ pd,
                                                              long long pd_stride, double *__restrict__ pt)
{
  __shared__ double xs[FXM_MB][FXM_RS * FXS_S];      // 64 KB: X of the mega band's rows; after the item: the direct sums of the odd waves
  __shared__ double scr[FXM_THREADS / 64][256];      // a tile's image per wave (transposition)
  __shared__ double xjst[FXM_THREADS / 64][16 * FXS_S];
  __shared__ double dtx[2][FXM_THREADS / 64][16 * FXS_S]; // transposed sums of a step, per wave
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), sbq = wave >> 1, par = wave & 1;
  // D lane l = (row / column cl of the tile, right-hand side 4 h + r4)
  const int kq = lane >> 4, r4 = lane & 3, a16 = lane & 15, cl = 4 * ((lane >> 2) & 3) + kq;
  const int xo = kq * FXS_S + r4; // operand of the products with X: lane l supplies X[row0 + 4 q + (l >> 4)][4 h + (l & 3)]
  double   *sc = scr[wave], *xjs = xjst[wave];
  int       wofs[4], rofs[4]; // LDS image of a tile: element (r, c) at r * 16 + ((c + r) & 15)
#pragma unroll
  for (int q = 0; q < 4; q++) wofs[q] = (4 * q + kq) * 16 + ((a16 + 4 * q + kq) & 15), rofs[q] = a16 * 16 + ((4 * q + kq + a16) & 15);
  const int it1 = __builtin_amdgcn_readfirstlane(wg_first[blockIdx.x + 1]);
  for (int it = __builtin_amdgcn_readfirstlane(wg_first[blockIdx.x]); it < it1; it++) {
    const int *w8 = items + 8 * it;
    const int  c = __builtin_amdgcn_readfirstlane(w8[0]), g = __builtin_amdgcn_readfirstlane(w8[1]), m = __builtin_amdgcn_readfirstlane(w8[2]);
    const int  jbeg = __builtin_amdgcn_readfirstlane(w8[3]), jend = __builtin_amdgcn_readfirstlane(w8[4]), seg = __builtin_amdgcn_readfirstlane(w8[5]);
    const int  ld = c_ld[c], nsb = ld / FXM_RS, sb = FXM_MB * m + sbq;
    const long long xbase = c_xoff[c] + (long long)g * ld * FXS_S;
    const double *__restrict__ x = X + xbase;
    // this wave's column tiles: J = jbeg + par, + 2, ... below jhi (a super band ends at its diagonal block)
    const int jhi = sb < nsb ? min(jend, (sb + 1) * FXM_RT) : jbeg, nst = (jend - jbeg + 1) >> 1, myst = jhi > jbeg + par ? (jhi - jbeg - par + 1) >> 1 : 0;
    const int ntile = myst * FXM_RT;
    const double *__restrict__ tp = Wbase + iteml[2 * it] + (long long)FXM_RS * FXM_RS * ((long long)sb * (sb + 1) / 2) + (long long)(jbeg + par) * (FXM_RT * 256) + lane * 2;
    double *__restrict__ ptp = pt + iteml[2 * it + 1];
    double dd[FXM_RT][2];
#pragma unroll
    for (int I = 0; I < FXM_RT; I++) dd[I][0] = dd[I][1] = 0.0;
    dbl2 ring[FXM_NB][2], xraw = {0.0, 0.0};
    if (ntile > 0) {
#pragma unroll
      for (int k = 0; k < FXM_NB; k++) { // tile t of the wave: step t >> 4 (column tile jbeg + par + 2 (t >> 4)), row tile t & 15
        const int     tt = min(k, ntile - 1);
        const double *q  = tp + (long long)(tt >> 4) * (2 * FXM_RT * 256) + (tt & 15) * 256;
        ring[k][0] = __builtin_nontemporal_load((const dbl2 *)q), ring[k][1] = __builtin_nontemporal_load((const dbl2 *)(q + 128));
      }
      xraw = *(const dbl2 *)(x + (long long)(jbeg + par) * 16 * FXS_S + lane * 2);
    }
    { // X of the mega band's rows -> LDS, with the first tiles of the stream already in flight
      const int     n  = min(FXM_MB * FXM_RS, ld - m * FXM_MB * FXM_RS) * FXS_S;
      const double *xm = x + (long long)m * FXM_MB * FXM_RS * FXS_S;
      double       *xf = &xs[0][0];
      for (int i = threadIdx.x * 2; i < n; i += 2 * FXM_THREADS) *(dbl2 *)(xf + i) = *(const dbl2 *)(xm + i);
    }
    __syncthreads();
    const double *xsb = xs[sbq];
    int           t   = 0;
    for (int s = 0; s < nst; s++) {
      double dt0 = 0.0, dt1 = 0.0;
      if (s < myst) {
        // X of the 16 columns of the column tile = 1 KB contiguous: every lane fetched 16 bytes of it one step ahead, the operands are read
        // back from a wave-private LDS image
        *(dbl2 *)(xjs + lane * 2) = xraw;
        double xj[4][2];
#pragma unroll
        for (int q = 0; q < 4; q++) xj[q][0] = xjs[4 * q * FXS_S + xo], xj[q][1] = xjs[4 * q * FXS_S + xo + 4];
        xraw = *(const dbl2 *)(x + (long long)(jbeg + par + 2 * min(s + 1, myst - 1)) * 16 * FXS_S + lane * 2);
        double u0 = 0.0, u1 = 0.0, u2 = 0.0, u3 = 0.0;
#pragma unroll
        for (int I = 0; I < FXM_RT; I++, t++) {
          const double t0 = ring[I % FXM_NB][0].x, t1 = ring[I % FXM_NB][0].y, t2 = ring[I % FXM_NB][1].x, t3 = ring[I % FXM_NB][1].y;
          sc[wofs[0]] = t0, sc[wofs[1]] = t1, sc[wofs[2]] = t2, sc[wofs[3]] = t3;
          const double *xi = xsb + (I * 16) * FXS_S + xo;
          const double  xi00 = xi[0], xi01 = xi[4], xi10 = xi[4 * FXS_S], xi11 = xi[4 * FXS_S + 4], xi20 = xi[8 * FXS_S], xi21 = xi[8 * FXS_S + 4], xi30 = xi[12 * FXS_S], xi31 = xi[12 * FXS_S + 4];
          if (I > 0) { // the direct product of the previous tile: its transposed image has arrived meanwhile
            dd[I ? I - 1 : 0][0] = fxm_mfma(u0, xj[0][0], dd[I ? I - 1 : 0][0]), dd[I ? I - 1 : 0][1] = fxm_mfma(u0, xj[0][1], dd[I ? I - 1 : 0][1]);
            dd[I ? I - 1 : 0][0] = fxm_mfma(u1, xj[1][0], dd[I ? I - 1 : 0][0]), dd[I ? I - 1 : 0][1] = fxm_mfma(u1, xj[1][1], dd[I ? I - 1 : 0][1]);
            dd[I ? I - 1 : 0][0] = fxm_mfma(u2, xj[2][0], dd[I ? I - 1 : 0][0]), dd[I ? I - 1 : 0][1] = fxm_mfma(u2, xj[2][1], dd[I ? I - 1 : 0][1]);
            dd[I ? I - 1 : 0][0] = fxm_mfma(u3, xj[3][0], dd[I ? I - 1 : 0][0]), dd[I ? I - 1 : 0][1] = fxm_mfma(u3, xj[3][1], dd[I ? I - 1 : 0][1]);
          }
          dt0 = fxm_mfma(t0, xi00, dt0), dt1 = fxm_mfma(t0, xi01, dt1);
          dt0 = fxm_mfma(t1, xi10, dt0), dt1 = fxm_mfma(t1, xi11, dt1);
          dt0 = fxm_mfma(t2, xi20, dt0), dt1 = fxm_mfma(t2, xi21, dt1);
          dt0 = fxm_mfma(t3, xi30, dt0), dt1 = fxm_mfma(t3, xi31, dt1);
          {
            const int     tt = min(t + FXM_NB, ntile - 1);
            const double *q  = tp + (long long)(tt >> 4) * (2 * FXM_RT * 256) + (tt & 15) * 256;
            ring[I % FXM_NB][0] = __builtin_nontemporal_load((const dbl2 *)q), ring[I % FXM_NB][1] = __builtin_nontemporal_load((const dbl2 *)(q + 128));
          }
          u0 = sc[rofs[0]], u1 = sc[rofs[1]], u2 = sc[rofs[2]], u3 = sc[rofs[3]];
          __builtin_amdgcn_sched_barrier(0);
        }
        dd[FXM_RT - 1][0] = fxm_mfma(u0, xj[0][0], dd[FXM_RT - 1][0]), dd[FXM_RT - 1][1] = fxm_mfma(u0, xj[0][1], dd[FXM_RT - 1][1]);
        dd[FXM_RT - 1][0] = fxm_mfma(u1, xj[1][0], dd[FXM_RT - 1][0]), dd[FXM_RT - 1][1] = fxm_mfma(u1, xj[1][1], dd[FXM_RT - 1][1]);
        dd[FXM_RT - 1][0] = fxm_mfma(u2, xj[2][0], dd[FXM_RT - 1][0]), dd[FXM_RT - 1][1] = fxm_mfma(u2, xj[2][1], dd[FXM_RT - 1][1]);
        dd[FXM_RT - 1][0] = fxm_mfma(u3, xj[3][0], dd[FXM_RT - 1][0]), dd[FXM_RT - 1][1] = fxm_mfma(u3, xj[3][1], dd[FXM_RT - 1][1]);
      }
      // the step's transposed sums: [column of the tile][right-hand side], added over the four super bands by waves 0 (even column tile) and 1
      double *dx = dtx[s & 1][wave] + cl * FXS_S + r4;
      dx[0] = dt0, dx[4] = dt1;
      __syncthreads(); // the buffer of this parity is rewritten two steps on, i.e. after the next barrier, which the adding waves reach after their reads
      if (wave < 2 && jbeg + 2 * s + wave < jend) {
        dbl2 v = *(const dbl2 *)(dtx[s & 1][wave] + lane * 2);
#pragma unroll
        for (int k = 1; k < FXM_MB; k++) v += *(const dbl2 *)(dtx[s & 1][2 * k + wave] + lane * 2);
        *(dbl2 *)(ptp + (long long)(jbeg + 2 * s + wave) * 16 * FXS_S + lane * 2) = v;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // direct sums: the odd wave's accumulators through LDS (the X stage is free now), the even wave adds its own and writes the item's sums
    __syncthreads();
    if (par) {
#pragma unroll
      for (int I = 0; I < FXM_RT; I++) xs[sbq][(I * 2) * 64 + lane] = dd[I][0], xs[sbq][(I * 2 + 1) * 64 + lane] = dd[I][1];
    }
    __syncthreads();
    if (!par && sb < nsb) {
      double *o = pd + (long long)seg * pd_stride + xbase + (long long)(sb * FXM_RS + cl) * FXS_S + r4;
#pragma unroll
      for (int I = 0; I < FXM_RT; I++) o[(I * 16) * FXS_S] = dd[I][0] + xs[sbq][(I * 2) * 64 + lane], o[(I * 16) * FXS_S + 4] = dd[I][1] + xs[sbq][(I * 2 + 1) * 64 + lane];
    }
    __syncthreads();
  }
}

// Y[position][slot] = the direct sums of the items of the position's mega band + the transposed partial sums of every owned mega band from that
// one on, in a fixed order.  grid (ld * 8 / 2 / 256, groups of the class); nseg_of[g * nmb + m] (0: not owned); the owned mega bands as a
// compact ascending list: own_first[m] = index of the first owned one >= m, own_ptoff[g * nown + k] = offset of its transposed sums
__global__ __launch_bounds__(PMH_BLOCK) void k_fxs_symfin(int ld, int nmb, int nown, const int *__restrict__ nseg_of, const int *__restrict__ own_first, const long long *__restrict__ own_ptoff,
                                                          long long xbase0, long long pd_stride, const double *__restrict__ pd, const double *__restrict__ pt, double *__restrict__ Y)
{
  // 8 lanes per pair of entries: lane `sub` adds the partial sums j = sub, sub + 8, ... (a small share of W_c cuts a mega band into > 100 items:
  // one thread per entry would walk them one load latency after the other), then a fixed shuffle tree -- still one summation order
  const long long t = (long long)blockIdx.x * PMH_BLOCK + threadIdx.x, i = 2 * (t >> 3);
  const int       sub = threadIdx.x & 7;
  dbl2            s = {0.0, 0.0};
  const bool      in = i < (long long)ld * FXS_S;
  const int       g = blockIdx.y;
  const long long xb = xbase0 + (long long)g * ld * FXS_S;
  if (in) {
    const int m0 = (int)(i / (FXM_MB * FXM_RS * FXS_S));
    const int ns = nseg_of[g * nmb + m0];
    for (int j = sub; j < ns; j += 8) s += *(const dbl2 *)(pd + (long long)j * pd_stride + xb + i);
    const long long *__restrict__ po = own_ptoff + (long long)g * nown;
    for (int k = own_first[m0] + sub; k < nown; k += 8) s += *(const dbl2 *)(pt + po[k] + i);
  }
#pragma unroll
  for (int o = 4; o > 0; o >>= 1) s.x += __shfl_down(s.x, o, 8), s.y += __shfl_down(s.y, o, 8);
  if (in && sub == 0) *(dbl2 *)(Y + xb + i) = s;
}

// row p of W_c from a K^+ solve: the entries c <= p go to the tiles of p's row tile (the diagonal entry halved, see above)
__global__ __launch_bounds__(PMH_BLOCK) void k_fxs_extract_sym(int p, const int *__restrict__ urel, const double *__restrict__ u, double *__restrict__ wsb)
{
  // wsb: first tile of p's super band
  const int Il = (p % FXM_RS) / 16, r = p & 15, q = r >> 2;
  for (int c = blockIdx.x * PMH_BLOCK + threadIdx.x; c <= p; c += gridDim.x * PMH_BLOCK) {
    const double v = u[urel[c]];
    const int    l = 16 * (r & 3) + (c & 15);
    wsb[((long long)(c >> 4) * FXM_RT + Il) * 256 + (q >> 1) * 128 + 2 * l + (q & 1)] = c == p ? 0.5 * v : v;
  }
}

// the same for a row obtained by symmetry: the solve gave row p (u), the operation g maps dof c to position posmap[c] with sign[c]:
// W[g p][g c] = sign[p] sign[c] W[p][c].  r = posmap[p] is the row written, sp = sign[p]
__global__ __launch_bounds__(PMH_BLOCK) void k_fxs_extract_symg(int r, int nc, double sp, const int *__restrict__ urel, const double *__restrict__ u, const int *__restrict__ posmap,
                                                               const signed char *__restrict__ sign, double *__restrict__ wsb)
{
  const int Il = (r % FXM_RS) / 16, rr = r & 15, q = rr >> 2;
  for (int c = blockIdx.x * PMH_BLOCK + threadIdx.x; c < nc; c += gridDim.x * PMH_BLOCK) {
    const int cc = posmap[c];
    if (cc > r) continue;
    const double v = sp * (double)sign[c] * u[urel[c]];
    const int    l = 16 * (rr & 3) + (cc & 15);
    wsb[((long long)(cc >> 4) * FXM_RT + Il) * 256 + (q >> 1) * 128 + 2 * l + (q & 1)] = cc == r ? 0.5 * v : v;
  }
}

// set-up self-check: max |stored row r - the directly solved row| and max |row| (entries c <= r), one value pair per workgroup
__global__ __launch_bounds__(PMH_BLOCK) void k_fxs_check_row(int r, const int *__restrict__ urel, const double *__restrict__ u, const double *__restrict__ wsb, double *__restrict__ out)
{
  __shared__ double red[PMH_BLOCK / 64];
  const int Il = (r % FXM_RS) / 16, rr = r & 15, q = rr >> 2;
  double    d = 0.0, m = 0.0;
  for (int c = blockIdx.x * PMH_BLOCK + threadIdx.x; c <= r; c += gridDim.x * PMH_BLOCK) {
    const int    l = 16 * (rr & 3) + (c & 15);
    const double w = wsb[((long long)(c >> 4) * FXM_RT + Il) * 256 + (q >> 1) * 128 + 2 * l + (q & 1)] * (c == r ? 2.0 : 1.0), v = u[urel[c]];
    d = fmax(d, fabs(w - v)), m = fmax(m, fabs(v));
  }
  d = -pmh_block_reduce<PMH_RED_MIN>(-d, red);
  m = -pmh_block_reduce<PMH_RED_MIN>(-m, red);
  if (threadIdx.x == 0) out[2 * blockIdx.x] = d, out[2 * blockIdx.x + 1] = m;
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// Orbit storage (PMH_FX_CLASS_ORBIT): W_c is invariant under the class's symmetries (fxs_set_symmetry), W[g p][g c] = s_g(p) s_g(c) W[p][c], so
// only the rows of the orbit REPRESENTATIVES are kept (configs[2]: 715 of 33 288 rows, 0.19 GB instead of 4.5 GB) and
//     Y[g p][s] = s_g(p) sum_c W[p][c] (s_g(c) X[g c][s])
// for every representative p, operation g and right-hand side s: a real GEMM, C = A B with A = the M representatives' rows (M x n_c), B[c][(g, s)] =
// s_g(c) X[g c][s] (n_c x 8 nsym, never formed: gathered from the L2-resident multivector through one index per (g, c) with the sign in its
// lowest bit).  2 M n_c 8 nsym flops on 8 M n_c bytes: 48 flop per byte for the cube's 48 operations -- the dense apply leaves the HBM roofline
// and runs on the fp64 matrix instruction (v_mfma_f64_4x4x4_4b_f64, as k_fxs_symm8).  Workgroup tile 128 x 128, k in chunks of 16, 4 waves of
// 64 x 64 (4 x 16 accumulators per lane), A pre-tiled in the order of its LDS image ([k][row] per (row tile, chunk): coalesced 16-byte loads),
// both operands double-buffered in LDS, split-K partial tiles summed in a fixed order by k_fxo_fin, which also applies s_g(p) and scatters row g p.
// (FXO_TM / FXO_TN / FXO_TK: fshared_types.h)
// items: (class, group, row tile, column tile (16 operations), first chunk, one-past-last chunk, split, 0); iteml: A offset of the class, X offset of
// the group, C offset of (class, group, split).
// (Rounds 2-3 ran this GEMM on v_mfma_f64_4x4x4_4b_f64 -- k_fxo_gemm, workgroup tile 128 x 128, and k_fxo_gemm4<NA> with row tiles 8 NA = 96 ... 120; removed at the end
// of round 6, the ladder is in docs/LAB_NOTEBOOK.md.)

// The GEMM on v_mfma_f64_16x16x4_f64: one instruction = a 16 x 16 tile over 4 k (2048 flop, 16 passes) where the 4x4x4_4b form of rounds 2-3 needed four (4 x 512
// flop, 4 passes each).  Same flop rate, but a quarter of the instructions and half of the operand registers read per flop:
// A[m = l & 15][k = l >> 4], B[k = l >> 4][n = l & 15], D column l & 15, rows (l >> 4) + 4 r in the 4 registers (scripts/micro/mfma_f64.hip).  Wave tile 16 NI
// x 64 (NI x 4 instruction tiles, 4 NI x 4 accumulator doubles per lane), workgroup tile 32 NI x 128 (2 x 2 waves); the LDS images are the ones of k_fxo_gemm
// (A) and k_fxo_gemm4 (B): per k step of 4 a lane reads NI + 4 operands for 4 NI instructions of 64 cycles (k_fxo_gemm4<15>: 19 operands for 60 instructions of
// 16 cycles).
typedef double dbl4 __attribute__((ext_vector_type(4)));
#ifndef FXO_IL_MFMA
#define FXO_IL_MFMA 2
#endif
#ifndef FXO_IL_VALU
#define FXO_IL_VALU 4
#endif
#ifdef FXO_TRACE // diagnostic build (make EXTRA=-DFXO_TRACE): cycle stamps of the phases of every chunk of a few workgroups' wave 0 (s_memrealtime, 100 MHz) and s_memtime (shader clock)
__device__ unsigned long long *fxo_trace_buf;
#define FXO_STAMP(slot)                                                                                              \
  do {                                                                                                               \
    if (trace_on) {                                                                                                  \
      const unsigned long long ts_ = __builtin_readcyclecounter();                                                   \
      if (lane == 0) fxo_trace_buf[((size_t)trace_wg * 64 + (size_t)trace_chunk) * 8 + (slot)] = ts_;                \
    }                                                                                                                \
  } while (0)
#else
#define FXO_STAMP(slot) \
  do {                  \
  } while (0)
#endif
// NWM waves down x (4 / NWM) across: NWM = 2: wave tile 16 NI x 64 (workgroup 32 NI x 128: 128 or 96 rows); NWM = 1: wave tile 16 NI x 32, the workgroup's rows
// are ANY multiple of 16 up to 144 (715 representatives pad to 720 = 5 x 144, as with the 4-row units of k_fxo_gemm4<15>; NI + 2 operand reads for 2 NI
// instructions per k step of 4) MULTI: one launch over the items of several classes (every class its own column lists, gather indices and symmetry count: the
// *_of tables, indexed by the item's class) TN: the workgroup's column tile (128; 64 for classes that list at most 64 columns per row tile: half the products
// of zeros)
template <int NI, int NWM, bool MULTI = false, int TN = FXO_TN>
__global__ __launch_bounds__(256, 2) void k_fxo_gemm16(const int *__restrict__ items, const long long *__restrict__ iteml, const int *__restrict__ c_nkc, const int *__restrict__ c_ldk,
                                                       const int *__restrict__ coltab /* of this launch's class */, int zrow, const double *__restrict__ A, const int *__restrict__ gidx /* of this launch's class */,
                                                       const double *__restrict__ X, double *__restrict__ cpart, const int *__restrict__ wgfirst, const int *const *__restrict__ coltab_of = nullptr,
                                                       const int *__restrict__ zrow_of = nullptr, const int *const *__restrict__ gidx_of = nullptr, const int *__restrict__ xshift_of = nullptr)
{
  // TN = 48 (classes of ONE block: at most the 48 operations of the cube as columns): the column LIST and the partial tiles keep their stride of 64 (TNL), the
  // gathers fill 64 columns of the LDS image (16 of them from the zero row) and only 48 are multiplied -- 4 waves down the rows (NWM = 4), 3 column blocks each
  constexpr int TNL = TN == 48 ? 64 : TN;
  constexpr int NWN = 4 / NWM, NJ = TN / (16 * NWN), WC = 16 * NJ, TM = 16 * NI * NWM, WR = 16 * NI, LDA = TM + 16;
  static_assert(TN == 128 || TN == 64 || (TN == 48 && NWM == 4), "column tile");
  // (Round 5, measured and not adopted: a THREE-stage operand pipeline -- the registers that hold chunk kc + 1 stored to LDS at the START of chunk kc, under
  // the products, then asked to fetch chunk kc + 2; the barrier directly behind the last product.  Same bits; 0.651 instead of 0.656-0.67 of the fp64 peak on
  // the 144 x 128 tile (256 VGPRs, an 8-byte spill), 0.506 instead of 0.51 on the 64-wide tile: the tail of a chunk -- wait, 13 LDS writes, barrier -- is not
  // what the pipe waits for.)
  __shared__ double As[2][FXO_TK][LDA];
  __shared__ double Bs[2][FXO_TK][TNL + 16];
  for (int it = __builtin_amdgcn_readfirstlane(wgfirst[blockIdx.x]), ite = __builtin_amdgcn_readfirstlane(wgfirst[blockIdx.x + 1]); it < ite; it++) {
  __builtin_amdgcn_sched_barrier(0);
  const int *w8 = items + 8 * it;
  const int  c = __builtin_amdgcn_readfirstlane(w8[0]), mt = __builtin_amdgcn_readfirstlane(w8[2]), nt = __builtin_amdgcn_readfirstlane(w8[3]);
  const int  kc0 = __builtin_amdgcn_readfirstlane(w8[4]), kc1 = __builtin_amdgcn_readfirstlane(w8[5]);
  const int  nkc = c_nkc[c], ldk = c_ldk[c], ncol = __builtin_amdgcn_readfirstlane(w8[7]);
  unsigned xsh = 6; // log2 of the bytes of a (position, sign) record of the signed multivector: 8 slots x 8 bytes
  if constexpr (MULTI) coltab = coltab_of[c], zrow = zrow_of[c], gidx = gidx_of[c], xsh = (unsigned)xshift_of[c];
  const double *__restrict__ Ab = A + iteml[4 * it];
  const double *__restrict__ x  = X + iteml[4 * it + 1];
  double *__restrict__ C        = cpart + iteml[4 * it + 2];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave / NWN, wn = wave % NWN;
  // a chunk of A: NEA passes of 16 bytes per lane + (RA = 128) one of 8
  constexpr int NQ = FXO_TK * TM / 2, NEA = NQ / 256, RA = NQ % 256, KPB = 256 / TNL, NEB = FXO_TK / KPB;
  static_assert(RA == 0 || RA == 128, "row tile");
  const int  col = t % TNL, kb = t / TNL;
  const int  ct  = coltab[iteml[4 * it + 3] + col];
  const int  sl  = ct < 0 ? 0 : (ct & 7);
  const int *gp  = gidx + (long long)(ct < 0 ? zrow : (ct >> 3)) * ldk;
  dbl4       acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; i++)
#pragma unroll
    for (int j = 0; j < NJ; j++) acc[i][j] = dbl4{0.0, 0.0, 0.0, 0.0};
  dbl2   ar[NEA];
  double ar1 = 0.0;
  double br[NEB];
  int    gn[NEB];
  // Addresses as UNIFORM 64-bit bases (scalar registers) + 32-bit per-lane offsets: one 32-bit vector operation per gather and none per load of A or of an
  // index (per-lane 64-bit pointers cost a sign extension, a 64-bit shift and a 64-bit add each -- vector-ALU cycles the fp64 products need)
  const char *__restrict__ xb   = (const char *)x;
  const char *__restrict__ gb   = (const char *)gidx;
  const unsigned           slo  = 8u * (unsigned)sl;
  // this lane's row of the index array (+ its k within a pass)
  const unsigned           goff = 4u * ((unsigned)(ct < 0 ? zrow : (ct >> 3)) * (unsigned)ldk + (unsigned)kb);
  const unsigned           aoff = 16u * (unsigned)t;
  auto loadA = [&](int kc) { // no lane is masked: a masked tail would move its load behind the products, next to the store that waits for it
    const char *blk = (const char *)(Ab + ((long long)mt * nkc + kc) * (FXO_TK * TM));
#pragma unroll
    for (int e = 0; e < NEA; e++) ar[e] = *(const dbl2 *)(blk + (aoff + 4096u * e));
    if (RA) ar1 = *(const double *)(blk + (4096u * NEA + 8u * (unsigned)t));
  };
  auto loadG = [&](int kc) {
    const char *gk = gb + 4 * (long long)kc * FXO_TK;
#pragma unroll
    for (int e = 0; e < NEB; e++) gn[e] = *(const int *)(gk + (goff + 4u * KPB * e));
  };
  auto gatherB = [&]() { // signed multivector: the index (position << 1 | negative) addresses the value with its sign; 64 bytes per (position, sign)
#pragma unroll
    for (int e = 0; e < NEB; e++) br[e] = *(const double *)(xb + (MULTI ? (((unsigned)gn[e] << xsh) + slo) : ((unsigned)gn[e] * 64u + slo)));
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int e = 0; e < NEA; e++) {
      const int q = t + 256 * e, k = q / (TM / 2), r2 = (q % (TM / 2)) * 2;
      *(dbl2 *)&As[buf][k][r2] = ar[e];
    }
    if (RA) {
      const int d = 512 * NEA + t;
      As[buf][d / TM][d % TM] = ar1;
    }
#pragma unroll
    for (int e = 0; e < NEB; e++) Bs[buf][kb + KPB * e][col] = br[e];
  };
  if (kc0 < kc1) {
    loadG(kc0);
    loadA(kc0);
    gatherB();
    if (kc0 + 1 < kc1) loadG(kc0 + 1);
    store(0);
  }
  __syncthreads();
  const int ka = lane >> 4, ra = lane & 15;
#ifdef FXO_TRACE
  const bool trace_on = wave == 0 && (blockIdx.x % 37) == 0 && blockIdx.x / 37 < 8 && it == __builtin_amdgcn_readfirstlane(wgfirst[blockIdx.x]);
  const int  trace_wg = blockIdx.x / 37;
#endif
  for (int kc = kc0; kc < kc1; kc++) {
    const int buf = (kc - kc0) & 1;
#ifdef FXO_TRACE
    const int trace_chunk = (kc - kc0) < 63 ? (kc - kc0) : 63;
#endif
    FXO_STAMP(0);
    // The next chunk's operands travel while this chunk is multiplied -- WITHOUT a branch: the last iterations ask for the last chunk again (kn, kg clamped)
    // and store it to the buffer nobody reads, so loads, products and stores are one basic block and the scheduler may place the address arithmetic and the
    // loads among the products.  The gathers need the indices asked for one chunk ago (the only loads outstanding here), so they go first.
    const int kn = kc + 1 < kc1 ? kc + 1 : kc1 - 1, kg = kc + 2 < kc1 ? kc + 2 : kc1 - 1;
    gatherB();
    loadG(kg);
    loadA(kn);
#ifdef FXO_TRACE_FULL
    FXO_STAMP(1);
#endif
#pragma unroll
    for (int k4 = 0; k4 < FXO_TK / 4; k4++) {
      double a[NI], b[NJ];
#pragma unroll
      for (int i = 0; i < NI; i++) a[i] = As[buf][4 * k4 + ka][wm * WR + i * 16 + ra];
#pragma unroll
      for (int j = 0; j < NJ; j++) b[j] = Bs[buf][4 * k4 + ka][wn * WC + j * 16 + ra];
#pragma unroll
      for (int i = 0; i < NI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
#ifndef FXO_NO_INTERLEAVE
    // the loads and their address arithmetic one by one BETWEEN the products (a product occupies the pipe for 64 cycles; what the wave issues meanwhile is
    // free, what it issues in a block of its own in front of the products is not): 3 products, 1 global load, 2 vector-ALU operations, ...
#pragma unroll
    // all loads within the first 2/3 of the products: the last one has a third of the chunk's products to arrive in
    for (int i = 0; i < NEB * 2 + NEA + (RA ? 1 : 0); i++) {
      __builtin_amdgcn_sched_group_barrier(0x008, FXO_IL_MFMA, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, FXO_IL_VALU, 0);
    }
#endif
    FXO_STAMP(2);
#ifdef FXO_TRACE_FULL
    __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0) only (gfx9 encoding: lgkmcnt / expcnt left at their maxima)
    FXO_STAMP(3);
#endif
    store(buf ^ 1);
#ifdef FXO_TRACE_FULL
    FXO_STAMP(4);
#endif
    __syncthreads();
    FXO_STAMP(5);
  }
#pragma unroll
  for (int i = 0; i < NI; i++)
#pragma unroll
    for (int j = 0; j < NJ; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) C[(long long)(wm * WR + i * 16 + ka + 4 * r) * ncol + nt * TNL + wn * WC + j * 16 + ra] = acc[i][j][r];
  }
}

// row tile of a class with M representatives: the padded row count decides; 128 (the faster orientation) unless a smaller tile saves more than 2.5 %
// the orbit GEMM runs on v_mfma_f64_16x16x4_f64 (k_fxo_gemm16)

// Y[g p][slot] = s_g(p) * (sum over the row tile's units (k segments) in unit order, over a unit's splits in split order) for the (row, operation) pairs that
// own their row (use = +-1: the operation the row was assigned to; rows fixed by several operations are written once), over the columns the (group, row tile)
// pairs list. One thread per (row, listed column); grid.y = group.  fintab per (group, row tile): offset of its column list, its padded column count, its first
// element in the group's numbering, its first unit; unittab per unit: offset of its look-up table (column of the tile's list -> column of the unit's list, -1:
// B is zero there on the whole segment, nothing was multiplied), its padded column count, its splits; unitbase: split 0 of the unit in cpart
#define FXO_FU 4
__device__ __forceinline__ void fxo_fin_body(int bx, int by, int ntile, int tm, int nsymp, int nc, const int *__restrict__ fintab, const int *__restrict__ unittab, const long long *__restrict__ unitbase,
                                                       const int *__restrict__ lut, const int *__restrict__ coltab, const double *__restrict__ cp, const signed char *__restrict__ use,
                                                       const int *__restrict__ reppos, const int *__restrict__ posmap, long long xbase0, int ld, double *__restrict__ Y, int nslot)
{
  const int  i  = bx * PMH_BLOCK + (int)threadIdx.x;
  const int *ft = fintab + 4 * (ntile + 1) * by;
  if (i >= ft[4 * ntile + 2]) return; // the group's element count
  int mt = 0;
  while (mt + 1 < ntile && i >= ft[4 * (mt + 1) + 2]) mt++;
  const int ncol = ft[4 * mt + 1], local = i - ft[4 * mt + 2], r = local / ncol, j = local % ncol;
  const int ct = coltab[ft[4 * mt] + j];
  if (ct < 0) return;
  const int g = ct >> 3, sl = ct & 7, row = mt * tm + r;
  const int u = use[(long long)row * nsymp + g];
  if (u == 0) return;
  const long long dst = xbase0 + (long long)by * ld * nslot + (long long)posmap[(long long)g * nc + reppos[row]] * nslot + sl;
  double          s   = 0.0;
  // FXO_FU units at a time: their look-ups, then the first 8 splits of each travel together (a plain loop compiles to load - wait - add per unit and split);
  // the sums are still taken unit after unit, split after split (+ 0.0 for a split that does not exist changes nothing)
  const int u1 = ft[4 * (mt + 1) + 3];
  for (int un = ft[4 * mt + 3]; un < u1; un += FXO_FU) {
    int           Su[FXO_FU];
    long long     st[FXO_FU];
    const double *q[FXO_FU];
#pragma unroll
    for (int e = 0; e < FXO_FU; e++) {
      const bool in  = un + e < u1;
      const int *ut  = unittab + 4 * (in ? un + e : un);
      const int  pos = lut[ut[0] + j], nct = ut[1];
      Su[e] = in && pos >= 0 ? ut[2] : 0;
      st[e] = (long long)tm * nct;
      q[e]  = cp + unitbase[in ? un + e : un] + (long long)r * nct + (pos >= 0 ? pos : 0);
    }
    double v[FXO_FU][8];
#pragma unroll
    for (int e = 0; e < FXO_FU; e++)
#pragma unroll
      for (int k = 0; k < 8; k++) v[e][k] = k < Su[e] ? q[e][(long long)k * st[e]] : 0.0;
#pragma unroll
    for (int e = 0; e < FXO_FU; e++) {
#pragma unroll
      for (int k = 0; k < 8; k++) s += v[e][k];
      for (int k = 8; k < Su[e]; k++) s += q[e][(long long)k * st[e]];
    }
  }
  Y[dst] = u > 0 ? s : -s;
}

__global__ __launch_bounds__(PMH_BLOCK) void k_fxo_fin(int ntile, int tm, int nsymp, int nc, const int *__restrict__ fintab, const int *__restrict__ unittab, const long long *__restrict__ unitbase,
                                                       const int *__restrict__ lut, const int *__restrict__ coltab, const double *__restrict__ cp, const signed char *__restrict__ use,
                                                       const int *__restrict__ reppos, const int *__restrict__ posmap, long long xbase0, int ld, double *__restrict__ Y, int nslot)
{
  fxo_fin_body(blockIdx.x, blockIdx.y, ntile, tm, nsymp, nc, fintab, unittab, unitbase, lut, coltab, cp, use, reppos, posmap, xbase0, ld, Y, nslot);
}

// (struct fxo_fin_args: fshared_types.h)
__global__ __launch_bounds__(PMH_BLOCK) void k_fxo_fin_all(const fxo_fin_args *__restrict__ args, const double *__restrict__ cp, double *__restrict__ Y)
{
  const fxo_fin_args a = args[blockIdx.z];
  if ((int)blockIdx.x >= a.nbx || (int)blockIdx.y >= a.ngroups) return;
  fxo_fin_body(blockIdx.x, blockIdx.y, a.ntile, a.tm, a.nsymp, a.nc, a.fintab, a.unittab, a.unitbase, a.lut, a.coltab, cp, a.use, a.reppos, a.posmap, a.xbase0, a.ld, Y, a.nslot);
}

// row of representative pl (local index) from its K^+ solve -> the pre-tiled A (column kinv[c] for position c)
__global__ __launch_bounds__(PMH_BLOCK) void k_fxo_store_row(int pl, int tm, int nc, int nkc, const int *__restrict__ urel, const int *__restrict__ kinv, const double *__restrict__ u, double *__restrict__ A)
{
  double *base = A + (long long)(pl / tm) * nkc * (FXO_TK * tm) + pl % tm;
  for (int c = blockIdx.x * PMH_BLOCK + threadIdx.x; c < nc; c += gridDim.x * PMH_BLOCK) {
    const int k = kinv[c];
    base[(long long)(k / FXO_TK) * (FXO_TK * tm) + (k % FXO_TK) * tm] = u[urel[c]];
  }
}

// set-up self-check: row r = g p from its own solve (u) against s_g(p) s_g(c) A[p][c] at column g c, for all c
__global__ __launch_bounds__(PMH_BLOCK) void k_fxo_check_row(int pl, int tm, int nc, int nkc, double sp, const int *__restrict__ urel, const int *__restrict__ kinv, const double *__restrict__ u,
                                                             const int *__restrict__ posmap, const signed char *__restrict__ sign, const double *__restrict__ A, double *__restrict__ out)
{
  __shared__ double red[PMH_BLOCK / 64];
  const double *base = A + (long long)(pl / tm) * nkc * (FXO_TK * tm) + pl % tm;
  double        d = 0.0, m = 0.0;
  for (int c = blockIdx.x * PMH_BLOCK + threadIdx.x; c < nc; c += gridDim.x * PMH_BLOCK) {
    const int    k = kinv[c];
    const double w = sp * (double)sign[c] * base[(long long)(k / FXO_TK) * (FXO_TK * tm) + (k % FXO_TK) * tm], v = u[urel[posmap[c]]];
    d = fmax(d, fabs(w - v)), m = fmax(m, fabs(v));
  }
  d = -pmh_block_reduce<PMH_RED_MIN>(-d, red);
  m = -pmh_block_reduce<PMH_RED_MIN>(-m, red);
  if (threadIdx.x == 0) out[2 * blockIdx.x] = d, out[2 * blockIdx.x + 1] = m;
}
