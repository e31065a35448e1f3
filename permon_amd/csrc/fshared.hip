// Explicit local dual operators SHARED by congruent blocks (storage PMH_FX_CLASS of pmh_fexplicit).
//
// Subdomains with bit-identical matrices (pmh_csr_block_classes: the 8 cubes of configs[2], the 64 of configs[3]) have the same K^+,
// so their dense operators W_b = (K^+)[Gamma_b, Gamma_b] are principal sub-matrices of ONE matrix W_c = (K^+)[U_c, U_c] on the union
// U_c of the dofs B touches in any block of the class (the whole boundary of the cube: 33 288 dofs for configs[2]).  Instead of one dense
// matrix per block (8 x 4 n_b^2 = 14.2 GB in symmetric storage) the class keeps W_c once in full (8 n_c^2 = 8.9 GB) and applies it to
// the blocks' vectors TOGETHER: X_c = [xhat_b scattered into U_c]_b is an n_c x 8 multivector (zero where a block does not touch a
// dof), Y_c = W_c X_c ONE pass over the matrix with eight right-hand sides (more blocks than 8: one pass per group of 8).  The matrix
// bytes per F apply drop by 1.6 x for configs[2] (by 2.9 x for the shape of configs[3]); the kernel walks down the rows with the lane
// owning its output columns (no reduction across lanes), and the rows of W_c deal over several GPUs as contiguous ranges.
// The gluing over the multivector numbering (index = (position in U_c) * 8 + slot of the block) is a pmh_gluing, so
//   F lambda = Bc' -> X,  Y = W_c X,  Bc Y (+ all-reduce)           stays three launches.
// Set-up: one K^+ solve per dof of U_c (what the per-block storage already did for congruent blocks), each giving one full row of W_c.
//
// Three storages of W_c live in this file (fx_shared::sym):
//   0  PMH_FX_CLASS        the full matrix, k_fxs_gemm8: 8 n_c^2 bytes per apply, HBM-bound
//   1  PMH_FX_CLASS_SYM    its lower block-triangle in 16 x 16 tiles, k_fxs_symm8 (both products of a tile on the fp64 matrix instruction): 4 n_c^2 bytes, HBM-bound
//   2  PMH_FX_CLASS_ORBIT  only the rows of the orbit representatives under the class's symmetries, k_fxo_gemm / k_fxo_gemm4: a GEMM on the fp64 matrix instruction,
//                          4 n_c^2 / 24 bytes for the cube's 48 operations, compute-bound (the default for congruent cubes)
// and the set-up by symmetry (fxs_set_symmetry: one K^+ solve per orbit of rows, self-checked against direct solves) serves 1 and 2.
#include <algorithm>
#include <chrono>
#include <map>
#include <cmath>

#include "feti_internal.h"
#include "fshared.h"
#include "pmh_internal.h"
#include "reduce.h"

typedef double dbl2 __attribute__((ext_vector_type(2)));

#define FXS_S 8    // right-hand sides per pass (blocks per group)
#define FXS_PAD 128

struct fxs_class {
  std::vector<int> blocks; // blocks of the class, ascending: slot = index % 8, group = index / 8
  std::vector<int> urel;   // sorted union of the touched dofs, relative to the block start
  std::vector<int> pos;    // relative dof -> position in urel (-1)
  int              nloc = 0, nc = 0, ld = 0, ngroups = 0, r0 = 0, r1 = 0;
  long long        woff = 0, xoff = 0;
  int             *d_urel = nullptr;
  // symmetric tile storage (fx_shared::sym): super bands of FXM_RS rows
  int              nsb = 0, nmb = 0; // mega bands of FXM_MB super bands
  std::vector<char> own;    // this rank applies / assembles super band sb (whole mega bands)
  long long        ptoff = 0, ptsize = 0; // transposed partial sums: ptoff + group * ptsize + ptm[mega band] + position * 8 + slot
  std::vector<long long> ptm;
  int             *d_nseg = nullptr;      // items (= segments of the direct sums) per (group, mega band) (0: not owned)
  long long       *d_ptoff = nullptr;
  int             *d_ownfirst = nullptr, nown = 0;
  // set-up by symmetry (fxs_set_symmetry): nsym signed permutations of U_c under which K_c^+ is invariant, op 0 = identity
  int                      nsym = 0;
  std::vector<int>         h_posmap; // [nsym][nc]: position of the image of the c-th touched dof
  std::vector<signed char> h_sign;   // [nsym][nc]: +-1
  int                     *d_posmap = nullptr;
  signed char             *d_sign = nullptr;
  // orbit storage (fx_shared::sym == 2): only the rows of W_c of the orbit representatives are kept, see the FXO section
  std::vector<int> reps, rep_of, op_of; // all representatives (positions, ascending); per row: its representative's position and the operation that reaches it
  int              M_all = 0, m0 = 0, m1 = 0, Mp = 0, ldk = 0, nkc = 0, nsymp = 0, tm = 128, tnw = 0; // tm: row tile of the GEMM (fxo_row_tile); tnw = 48: the 48-column kernel (one-block classes)
  long long        aoff = 0, coff = 0;  // offsets of the class in Afund / cpart
  int             *d_gidx = nullptr, *d_reppos = nullptr;
  // output pruning of the orbit GEMM: block (group, slot) touches only part of U_c, so row g p of Y is needed for the slots that touch it only.  Per (group, row tile)
  // the columns (operation << 3 | slot) some row of the tile needs, padded to 128 with -1; the representatives are ordered by their need pattern (rows of A)
  std::vector<char> tmask;              // [ngroups][nc][8]
  std::vector<int>  reprow;             // representative index (in reps) -> row of A / cpart
  int              *d_coltab = nullptr, *d_fintab = nullptr; // fintab per (group, row tile): coltab offset, padded columns, first element of the tile in the group's numbering
  long long        *d_finbase = nullptr;                     // per (group, row tile): offset of split 0 in cpart
  int               item_first = 0, item_count = 0, fin_elems = 0, wgf_first = 0, wg_count = 0; // the class's items; its workgroups (slice of fx_shared::d_wgfirst)
  int               tn = 128; // column tile of the class's GEMM: 64 when no (group, row tile) lists more than 64 columns (a class of ONE block lists at most its 48 operations)
  int               S = 8; // orbit storage: slots of a multivector record = the smallest power of two >= the class's blocks (<= 8): a class of ONE block gathers 8-byte records, not a 64-byte line with seven zeros
  signed char     *d_use = nullptr;
  // k segments of the orbit GEMM: the positions (= the k index of the product) are grouped by WHICH columns have a structural non-zero of B there (block (group, slot)
  // does not touch g c => B[c][(g, slot)] = 0), the rows of B are permuted segment after segment (each padded to whole chunks) and a (row tile, segment) multiplies
  // only the columns that are non-zero on the segment (fxo_prepare).  kinv: position -> row of B / column of the pre-tiled A
  std::vector<int> kinv;
  int             *d_kinv = nullptr, *d_unittab = nullptr, *d_lut = nullptr; // unittab per (group, row tile, segment) unit: offset of its look-up table, padded columns, splits, 0
  int              nseg = 1;
};

struct fx_shared {
  pmh_ctx                ctx;
  pmh_gluing             B;
  pmh_blockdiag          K;
  int                    nb, ncls;
  std::vector<int>       cls; // class of every block
  std::vector<fxs_class> C;
  pmh_gluing             Bc = nullptr;
  // orbit storage: the multivector the GEMM gathers from holds every entry TWICE, [position][+x | -x][slot] -- the gather index (position << 1 | negative) addresses the signed
  // value directly, no sign is applied to a loaded value inside the GEMM (a use of the loaded value in front of the products made every wave wait for all of a chunk's global
  // loads before its first MFMA).  Bc2: the gluing that fills it (every leaf of Bc twice, the second with the opposite sign); Bc stays for B Y on the way back
  pmh_gluing             Bc2 = nullptr;
  double                *X2  = nullptr;
  double                *Wbase = nullptr, *X = nullptr, *Y = nullptr;
  long long              nX = 0, wtot = 0;
  int                   *d_wg = nullptr; // launch table: (class, group, first column, segment, first row, one-past-last row) per workgroup
  int                    nwg = 0, nseg = 0;
  double                *part = nullptr; // [nseg][nX] segment sums of k_fxs_gemm8
  long long              part_cap = 0;
  int                   *d_ld = nullptr;
  long long             *d_woff = nullptr, *d_xoff = nullptr;
  double                 bytes = 0.0;
  std::vector<hipEvent_t> ev;
  int                    ev_used = 0, ev_on = 0;
  std::vector<hipEvent_t> ev_mid; // orbit storage: after the GEMM kernel, before k_fxo_fin (the first kernel's own duration)
  int                    ev_mid_pending = -1;
  // symmetric tile storage (PMH_FX_CLASS_SYM): the lower block-triangle of W_c in 16 x 16 tiles, k_fxs_symm8 (fp64 MFMA) + k_fxs_symfin
  int                    sym = 0, segj = 0;
  long long             *d_wgl = nullptr; // per item: offset of its class's tiles, offset of its transposed partial sums
  int                   *d_items = nullptr, *d_wgfirst = nullptr;
  // several classes on the same row tile: ONE launch over all their work items (fxo_gemm): workgroup -> items with global item numbers, per-class pointer tables
  int                   *d_wgfirst_all = nullptr, *d_zrow_of = nullptr, nwg_all = 0, merged_tm = 0, merged_tn = 128, merged_tnw = 0;
  const int            **d_coltab_of = nullptr, **d_gidx_of = nullptr;
  void                  *d_fin_args = nullptr; // fxo_fin_args per class: the classes' finishing launches as one (k_fxo_fin_all)
  int                    fin_nbx = 0, fin_ngroups = 0;
  double                *pt = nullptr;
  long long              pt_tot = 0;
  double                 owned_bytes = 0.0;
  // orbit storage
  double                *Afund = nullptr, *cpart = nullptr;
  long long              afund_tot = 0, cpart_cap = 0;
  int                    fxo_ready = 0, fxo_S = 1, stripe_rank = 0, stripe_size = 0;
  bool                   mfma16 = true; // the orbit GEMM on v_mfma_f64_16x16x4 (PMH_FXO_MFMA4 read ONCE, when the operator is created: nothing on the apply path asks the environment)
  double                 flops = 0.0, flops_issued = 0.0, flops_dense = 0.0; // listed columns x valid rows; padded tiles; every (representative, operation, block)
};

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
        for (int u = 0; u < FXS_U; u++) a[u] = __builtin_nontemporal_load((const dbl2 *)(A + (long long)min(blk + r + u, rhi - 1) * ld)); // rows past the end: X is zero there
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
  const int kq = lane >> 4, r4 = lane & 3, a16 = lane & 15, cl = 4 * ((lane >> 2) & 3) + kq; // D lane l = (row / column cl of the tile, right-hand side 4 h + r4)
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
#define FXO_TM 128
#define FXO_TN 128
#define FXO_TK 16
#define FXO_LDA (FXO_TM + 16)
#define FXO_LDB (FXO_TN + 4)
// items: (class, group, row tile, column tile (16 operations), first chunk, one-past-last chunk, split, 0); iteml: A offset of the class, X offset of
// the group, C offset of (class, group, split)
__global__ __launch_bounds__(256, 2) void k_fxo_gemm(const int *__restrict__ items, const long long *__restrict__ iteml, const int *__restrict__ c_nkc, const int *__restrict__ c_ldk,
                                                     const int *__restrict__ coltab /* of this launch's class */, int zrow, const double *__restrict__ A, const int *__restrict__ gidx /* of this launch's class */,
                                                     const double *__restrict__ X, double *__restrict__ cpart, const int *__restrict__ wgfirst)
{
  __shared__ double As[2][FXO_TK][FXO_LDA];
  __shared__ double Bs[2][FXO_TK][FXO_LDB];
  // the workgroup's items one after the other (a piece of the k range may end one unit and begin the next: fxo_prepare)
  for (int it = __builtin_amdgcn_readfirstlane(wgfirst[blockIdx.x]), ite = __builtin_amdgcn_readfirstlane(wgfirst[blockIdx.x + 1]); it < ite; it++) {
  __builtin_amdgcn_sched_barrier(0);
  const int *w8 = items + 8 * it;
  const int  c = __builtin_amdgcn_readfirstlane(w8[0]), mt = __builtin_amdgcn_readfirstlane(w8[2]), nt = __builtin_amdgcn_readfirstlane(w8[3]);
  const int  kc0 = __builtin_amdgcn_readfirstlane(w8[4]), kc1 = __builtin_amdgcn_readfirstlane(w8[5]);
  const int  nkc = c_nkc[c], ldk = c_ldk[c], ncol = __builtin_amdgcn_readfirstlane(w8[7]); // ncol: padded columns of this (group, row tile)
  const double *__restrict__ Ab = A + iteml[4 * it];
  const double *__restrict__ x  = X + iteml[4 * it + 1];
  double *__restrict__ C        = cpart + iteml[4 * it + 2]; // the (group, row tile, split) block: tile rows x ncol
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
  constexpr int NEA = FXO_TK * FXO_TM / 2 / 256, KPB = 256 / FXO_TN, NEB = FXO_TK / KPB;
  const int  col = t % FXO_TN, kb = t / FXO_TN;
  const int  ct  = coltab[iteml[4 * it + 3] + col]; // this column: operation << 3 | slot, -1 = padding (gathers the zero row)
  const int  sl  = ct < 0 ? 0 : (ct & 7);
  const int *gp  = gidx + (long long)(ct < 0 ? zrow : (ct >> 3)) * ldk;
  double     acc[4][16];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 16; j++) acc[i][j] = 0.0;
  dbl2   ar[NEA];
  double br[NEB];
  int    gn[NEB];
  auto loadA = [&](int kc) {
    const double *blk = Ab + ((long long)mt * nkc + kc) * (FXO_TK * FXO_TM);
#pragma unroll
    for (int e = 0; e < NEA; e++) ar[e] = *(const dbl2 *)(blk + 2 * (t + 256 * e)); // default cache policy: the workgroups of the other column tiles read the same chunk from the XCD's L2 (work-item order below)
  };
  auto loadG = [&](int kc) {
#pragma unroll
    for (int e = 0; e < NEB; e++) gn[e] = gp[kc * FXO_TK + kb + KPB * e];
  };
  auto gatherB = [&]() { // X holds +x and -x per (position, slot): the index (position << 1 | negative) addresses the signed value; nothing here may USE a loaded value (that would
                         // put the wait for all of the chunk's global loads in front of the products)
#pragma unroll
    for (int e = 0; e < NEB; e++) {
      br[e] = x[(long long)gn[e] * FXS_S + sl];
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int e = 0; e < NEA; e++) {
      const int q = t + 256 * e, k = q / (FXO_TM / 2), r2 = (q % (FXO_TM / 2)) * 2;
      *(dbl2 *)&As[buf][k][r2] = ar[e];
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
  const int ka = lane >> 4, ra = lane & 15, cb = lane & 3;
  for (int kc = kc0; kc < kc1; kc++) {
    const int buf = (kc - kc0) & 1;
    if (kc + 1 < kc1) { // the next chunk's operands travel while this chunk is multiplied.  Order matters: the gathers need the indices asked for one chunk ago -- the only loads
                        // outstanding here --, so they go first; with the loads of A in front of them the wait for those indices (s_waitcnt vmcnt is in order, and the compiler
                        // counts conservatively across A's exec-masked load blocks) became a wait for A itself, in front of the products
      gatherB();
      if (kc + 2 < kc1) loadG(kc + 2);
      loadA(kc + 1);
    }
#pragma unroll
    for (int k4 = 0; k4 < FXO_TK / 4; k4++) {
      double a[4], b[16];
#pragma unroll
      for (int i = 0; i < 4; i++) a[i] = As[buf][4 * k4 + ka][wm * 64 + i * 16 + ra];
#pragma unroll
      for (int j = 0; j < 16; j++) b[j] = Bs[buf][4 * k4 + ka][wn * 64 + j * 4 + cb];
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 16; j++) acc[i][j] = fxm_mfma(a[i], b[j], acc[i][j]);
    }
    if (kc + 1 < kc1) store(buf ^ 1);
    __syncthreads();
  }
  // D lane l: row 4 ((l >> 2) & 3) + (l >> 4) of the 16, column l & 3 of the 4
  const int rr = 4 * ((lane >> 2) & 3) + (lane >> 4);
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 16; j++) C[(long long)(wm * 64 + i * 16 + rr) * ncol + nt * FXO_TN + wn * 64 + j * 4 + cb] = acc[i][j];
  }
}

// The same GEMM with the instruction's operands the other way round: the SAME 4 rows of A in its 4 blocks, 16 columns of B (4 per block) -- rows come in
// units of 4 instead of 16, so the row tile can be 8 NA = 96 ... 120 and 715 representatives pad to 720 rows (6 x 120) instead of 768.  At equal tile
// this orientation is ~2 % slower than k_fxo_gemm (scripts/micro/orbit_gemm.hip), so it is used when it saves more than that in padding (fxo_row_tile).
// Wave tile 4 NA x 64: NA x 4 accumulators; D lane l = row l >> 4 of the 4, column l & 15 of the 16.
#define FXO_LDB4 (FXO_TN + 16) // 16 consecutive columns x 4 k per read: rows of B 32 banks apart
template <int NA>
__global__ __launch_bounds__(256, 2) void k_fxo_gemm4(const int *__restrict__ items, const long long *__restrict__ iteml, const int *__restrict__ c_nkc, const int *__restrict__ c_ldk,
                                                      const int *__restrict__ coltab /* of this launch's class */, int zrow, const double *__restrict__ A, const int *__restrict__ gidx /* of this launch's class */,
                                                      const double *__restrict__ X, double *__restrict__ cpart, const int *__restrict__ wgfirst)
{
  constexpr int TM = 8 * NA, WR = 4 * NA, LDA = TM + 16;
  __shared__ double As[2][FXO_TK][LDA];
  __shared__ double Bs[2][FXO_TK][FXO_LDB4];
  // the workgroup's items one after the other (a piece of the k range may end one unit and begin the next: fxo_prepare)
  for (int it = __builtin_amdgcn_readfirstlane(wgfirst[blockIdx.x]), ite = __builtin_amdgcn_readfirstlane(wgfirst[blockIdx.x + 1]); it < ite; it++) {
  __builtin_amdgcn_sched_barrier(0);
  const int *w8 = items + 8 * it;
  const int  c = __builtin_amdgcn_readfirstlane(w8[0]), mt = __builtin_amdgcn_readfirstlane(w8[2]), nt = __builtin_amdgcn_readfirstlane(w8[3]);
  const int  kc0 = __builtin_amdgcn_readfirstlane(w8[4]), kc1 = __builtin_amdgcn_readfirstlane(w8[5]);
  const int  nkc = c_nkc[c], ldk = c_ldk[c], ncol = __builtin_amdgcn_readfirstlane(w8[7]); // ncol: padded columns of this (group, row tile)
  const double *__restrict__ Ab = A + iteml[4 * it];
  const double *__restrict__ x  = X + iteml[4 * it + 1];
  double *__restrict__ C        = cpart + iteml[4 * it + 2]; // the (group, row tile, split) block: tile rows x ncol
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
  constexpr int NQ = FXO_TK * TM / 2, NEA = (NQ + 255) / 256, KPB = 256 / FXO_TN, NEB = FXO_TK / KPB; // NQ 16-byte pieces of A per chunk
  const int  col = t % FXO_TN, kb = t / FXO_TN;
  const int  ct  = coltab[iteml[4 * it + 3] + col]; // this column: operation << 3 | slot, -1 = padding (gathers the zero row)
  const int  sl  = ct < 0 ? 0 : (ct & 7);
  const int *gp  = gidx + (long long)(ct < 0 ? zrow : (ct >> 3)) * ldk;
  double     acc[NA][4];
#pragma unroll
  for (int i = 0; i < NA; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = 0.0;
  dbl2   ar[NEA];
  double br[NEB];
  int    gn[NEB];
  auto loadA = [&](int kc) {
    const double *blk = Ab + ((long long)mt * nkc + kc) * (FXO_TK * TM);
#pragma unroll
    for (int e = 0; e < NEA; e++)
      if (NQ % 256 == 0 || t + 256 * e < NQ) ar[e] = *(const dbl2 *)(blk + 2 * (t + 256 * e)); // default cache policy, see k_fxo_gemm
  };
  auto loadG = [&](int kc) {
#pragma unroll
    for (int e = 0; e < NEB; e++) gn[e] = gp[kc * FXO_TK + kb + KPB * e];
  };
  auto gatherB = [&]() { // X holds +x and -x per (position, slot): the index (position << 1 | negative) addresses the signed value; nothing here may USE a loaded value (that would
                         // put the wait for all of the chunk's global loads in front of the products)
#pragma unroll
    for (int e = 0; e < NEB; e++) {
      br[e] = x[(long long)gn[e] * FXS_S + sl];
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int e = 0; e < NEA; e++) {
      const int q = t + 256 * e, k = q / (TM / 2), r2 = (q % (TM / 2)) * 2;
      if (NQ % 256 == 0 || q < NQ) *(dbl2 *)&As[buf][k][r2] = ar[e];
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
  const int ka = lane >> 4, ra = lane & 15, cb = lane & 3;
  for (int kc = kc0; kc < kc1; kc++) {
    const int buf = (kc - kc0) & 1;
    if (kc + 1 < kc1) { // the next chunk's operands travel while this chunk is multiplied.  Order matters: the gathers need the indices asked for one chunk ago -- the only loads
                        // outstanding here --, so they go first; with the loads of A in front of them the wait for those indices (s_waitcnt vmcnt is in order, and the compiler
                        // counts conservatively across A's exec-masked load blocks) became a wait for A itself, in front of the products
      gatherB();
      if (kc + 2 < kc1) loadG(kc + 2);
      loadA(kc + 1);
    }
#pragma unroll
    for (int k4 = 0; k4 < FXO_TK / 4; k4++) {
      double a[NA], b[4];
#pragma unroll
      for (int i = 0; i < NA; i++) a[i] = As[buf][4 * k4 + ka][wm * WR + i * 4 + cb];
#pragma unroll
      for (int j = 0; j < 4; j++) b[j] = Bs[buf][4 * k4 + ka][wn * 64 + j * 16 + ra];
#pragma unroll
      for (int i = 0; i < NA; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = fxm_mfma(a[i], b[j], acc[i][j]);
    }
    if (kc + 1 < kc1) store(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < NA; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) C[(long long)(wm * WR + i * 4 + ka) * ncol + nt * FXO_TN + wn * 64 + j * 16 + ra] = acc[i][j];
  }
}

// The same GEMM on the other fp64 shape of the matrix pipe, v_mfma_f64_16x16x4_f64: one instruction = a 16 x 16 tile over 4 k (2048 flop, 16 passes) where the 4x4x4_4b form
// needs four (4 x 512 flop, 4 passes each).  Same flop rate, but a quarter of the instructions and half of the operand registers read per flop: A[m = l & 15][k = l >> 4],
// B[k = l >> 4][n = l & 15], D column l & 15, rows (l >> 4) + 4 r in the 4 registers (scripts/micro/mfma_f64.hip).  Wave tile 16 NI x 64 (NI x 4 instruction tiles, 4 NI x 4
// accumulator doubles per lane), workgroup tile 32 NI x 128 (2 x 2 waves); the LDS images are the ones of k_fxo_gemm (A) and k_fxo_gemm4 (B): per k step of 4 a lane reads
// NI + 4 operands for 4 NI instructions of 64 cycles (k_fxo_gemm4<15>: 19 operands for 60 instructions of 16 cycles).
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
// NWM waves down x (4 / NWM) across: NWM = 2: wave tile 16 NI x 64 (workgroup 32 NI x 128: 128 or 96 rows); NWM = 1: wave tile 16 NI x 32, the workgroup's rows are ANY multiple of 16
// up to 144 (715 representatives pad to 720 = 5 x 144, as with the 4-row units of k_fxo_gemm4<15>; NI + 2 operand reads for 2 NI instructions per k step of 4)
// MULTI: one launch over the items of several classes (every class its own column lists, gather indices and symmetry count: the *_of tables, indexed by the item's class)
// TN: the workgroup's column tile (128; 64 for classes that list at most 64 columns per row tile: half the products of zeros)
template <int NI, int NWM, bool MULTI = false, int TN = FXO_TN>
__global__ __launch_bounds__(256, 2) void k_fxo_gemm16(const int *__restrict__ items, const long long *__restrict__ iteml, const int *__restrict__ c_nkc, const int *__restrict__ c_ldk,
                                                       const int *__restrict__ coltab /* of this launch's class */, int zrow, const double *__restrict__ A, const int *__restrict__ gidx /* of this launch's class */,
                                                       const double *__restrict__ X, double *__restrict__ cpart, const int *__restrict__ wgfirst, const int *const *__restrict__ coltab_of = nullptr,
                                                       const int *__restrict__ zrow_of = nullptr, const int *const *__restrict__ gidx_of = nullptr, const int *__restrict__ xshift_of = nullptr)
{
  // TN = 48 (classes of ONE block: at most the 48 operations of the cube as columns): the column LIST and the partial tiles keep their stride of 64 (TNL), the gathers fill 64 columns
  // of the LDS image (16 of them from the zero row) and only 48 are multiplied -- 4 waves down the rows (NWM = 4), 3 column blocks each
  constexpr int TNL = TN == 48 ? 64 : TN;
  constexpr int NWN = 4 / NWM, NJ = TN / (16 * NWN), WC = 16 * NJ, TM = 16 * NI * NWM, WR = 16 * NI, LDA = TM + 16;
  static_assert(TN == 128 || TN == 64 || (TN == 48 && NWM == 4), "column tile");
  // (Round 5, measured and not adopted: a THREE-stage operand pipeline -- the registers that hold chunk kc + 1 stored to LDS at the START of chunk kc, under the products, then
  // asked to fetch chunk kc + 2; the barrier directly behind the last product.  Same bits; 0.651 instead of 0.656-0.67 of the fp64 peak on the 144 x 128 tile (256 VGPRs, an
  // 8-byte spill), 0.506 instead of 0.51 on the 64-wide tile: the tail of a chunk -- wait, 13 LDS writes, barrier -- is not what the pipe waits for.)
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
  constexpr int NQ = FXO_TK * TM / 2, NEA = NQ / 256, RA = NQ % 256, KPB = 256 / TNL, NEB = FXO_TK / KPB; // a chunk of A: NEA passes of 16 bytes per lane + (RA = 128) one of 8
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
  // Addresses as UNIFORM 64-bit bases (scalar registers) + 32-bit per-lane offsets: one 32-bit vector operation per gather and none per load of A or of an index
  // (per-lane 64-bit pointers cost a sign extension, a 64-bit shift and a 64-bit add each -- vector-ALU cycles the fp64 products need)
  const char *__restrict__ xb   = (const char *)x;
  const char *__restrict__ gb   = (const char *)gidx;
  const unsigned           slo  = 8u * (unsigned)sl;
  const unsigned           goff = 4u * ((unsigned)(ct < 0 ? zrow : (ct >> 3)) * (unsigned)ldk + (unsigned)kb); // this lane's row of the index array (+ its k within a pass)
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
    // The next chunk's operands travel while this chunk is multiplied -- WITHOUT a branch: the last iterations ask for the last chunk again (kn, kg clamped) and store it to the
    // buffer nobody reads, so loads, products and stores are one basic block and the scheduler may place the address arithmetic and the loads among the products.  The gathers need
    // the indices asked for one chunk ago (the only loads outstanding here), so they go first.
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
    // the loads and their address arithmetic one by one BETWEEN the products (a product occupies the pipe for 64 cycles; what the wave issues meanwhile is free, what it issues
    // in a block of its own in front of the products is not): 3 products, 1 global load, 2 vector-ALU operations, ...
#pragma unroll
    for (int i = 0; i < NEB * 2 + NEA + (RA ? 1 : 0); i++) { // all loads within the first 2/3 of the products: the last one has a third of the chunk's products to arrive in
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
// the orbit GEMM runs on v_mfma_f64_16x16x4_f64 (k_fxo_gemm16); PMH_FXO_MFMA4=1: the 4x4x4_4b kernels of rounds 2-3 (k_fxo_gemm / k_fxo_gemm4<NA>) for the A/B
static bool fxo_mfma16() { return getenv("PMH_FXO_MFMA4") == nullptr; }
static int fxo_row_tile(int M)
{
  if (fxo_mfma16()) { // 16-row instruction tiles: the workgroup tile that pads least among 144, 128, 112, 96, 80 (ties: the larger tile)
    if (const char *e = getenv("PMH_FXO_TM")) {
      const int v = atoi(e);
      if (v == 144 || v == 128 || v == 112 || v == 96 || v == 80) return v;
    }
    int best = 144, pad = (M + 143) / 144 * 144;
    for (int tm : {128, 112, 96, 80})
      if ((M + tm - 1) / tm * tm < pad) pad = (M + tm - 1) / tm * tm, best = tm;
    return best;
  }
  if (const char *e = getenv("PMH_FXO_TM")) {
    const int v = atoi(e);
    if (v == 128 || v == 120 || v == 112 || v == 104 || v == 96) return v;
  }
  int    best = 128;
  double cost = (double)((M + 127) / 128 * 128);
  for (int tm : {120, 112, 104, 96}) {
    const double cst = 1.025 * (double)((M + tm - 1) / tm * tm);
    if (cst < cost) cost = cst, best = tm;
  }
  return best;
}

// Y[g p][slot] = s_g(p) * (sum over the row tile's units (k segments) in unit order, over a unit's splits in split order) for the (row, operation) pairs that own
// their row (use = +-1: the operation the row was assigned to; rows fixed by several operations are written once), over the columns the (group, row tile) pairs list.
// One thread per (row, listed column); grid.y = group.  fintab per (group, row tile): offset of its column list, its padded column count, its first element in the
// group's numbering, its first unit; unittab per unit: offset of its look-up table (column of the tile's list -> column of the unit's list, -1: B is zero there on the
// whole segment, nothing was multiplied), its padded column count, its splits; unitbase: split 0 of the unit in cpart
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

// several classes in ONE launch (blockIdx.z = class; grid.x / grid.y = the largest class's): every class's parameters from a device table
struct fxo_fin_args {
  int              ntile, tm, nsymp, nc, ld, nslot, nbx, ngroups;
  const int       *fintab, *unittab, *lut, *coltab, *reppos, *posmap;
  const long long *unitbase;
  const signed char *use;
  long long        xbase0;
};
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

static int fxs_build_launch(fx_shared *S)
{
  if (S->sym) {
    // work = (class, group, owned mega band m) one after the other, the longest first, each a run of steps (pairs of column tiles, 128 KB
    // when all four super bands reach that far); the run is cut into one equal share per CU
    struct band { int c, g, m, nj; };
    std::vector<band> bands;
    long long         steps = 0;
    for (int c = 0; c < S->ncls; c++) {
      fxs_class &C = S->C[c];
      for (int m = C.nmb - 1; m >= 0; m--) {
        if (!C.own[FXM_MB * m]) continue;
        for (int g = 0; g < C.ngroups; g++) {
          const int nj = std::min(C.nsb, FXM_MB * (m + 1)) * FXM_RT;
          bands.push_back({c, g, m, nj});
          steps += (nj + 1) / 2;
        }
      }
    }
    int nwg = (int)std::max(1LL, std::min((long long)S->ctx->num_cus, steps));
    if (const char *e = getenv("PMH_FXM_NWG")) nwg = std::max(1, atoi(e));
    std::vector<int>       first(1, 0), items;
    std::vector<long long> iteml;
    std::vector<std::vector<int>> nseg_of(S->ncls);
    for (int c = 0; c < S->ncls; c++) nseg_of[c].assign((size_t)std::max(1, S->C[c].ngroups * S->C[c].nmb), 0);
    {
      long long done = 0; // steps handed out so far
      size_t    bi = 0;
      long long boff = 0; // steps of band bi already handed out
      for (int k = 0; k < nwg; k++) {
        const long long upto = steps * (k + 1) / nwg;
        while (done < upto && bi < bands.size()) {
          const band     &b   = bands[bi];
          const long long bst = (b.nj + 1) / 2, take = std::min(bst - boff, upto - done);
          fxs_class      &C   = S->C[b.c];
          int            &ns  = nseg_of[b.c][(size_t)b.g * C.nmb + b.m];
          items.insert(items.end(), {b.c, b.g, b.m, (int)(2 * boff), (int)std::min<long long>(b.nj, 2 * (boff + take)), ns, 0, 0});
          iteml.push_back(C.woff);
          iteml.push_back(C.ptoff + (long long)b.g * C.ptsize + C.ptm[b.m]);
          ns++, done += take, boff += take;
          if (boff == bst) bi++, boff = 0;
        }
        first.push_back((int)(items.size() / 8));
      }
    }
    int nsegmax = 1;
    S->bytes = 0.0, S->owned_bytes = 0.0;
    for (int c = 0; c < S->ncls; c++) {
      fxs_class &C     = S->C[c];
      double     tiles = 0.0, parts = 0.0;
      std::vector<int> ownm, own_first((size_t)C.nmb + 1, 0);
      for (int m = 0; m < C.nmb; m++) {
        own_first[m] = (int)ownm.size();
        if (C.own[FXM_MB * m]) ownm.push_back(m);
      }
      C.nown = (int)ownm.size();
      std::vector<long long> ptoff_of((size_t)std::max(1, C.ngroups * C.nown), 0);
      for (int g = 0; g < C.ngroups; g++)
        for (int k = 0; k < C.nown; k++) ptoff_of[(size_t)g * C.nown + k] = C.ptoff + (long long)g * C.ptsize + C.ptm[ownm[k]];
      if (!C.d_ownfirst) PMH_CHK(pmh_malloc(S->ctx, sizeof(int) * own_first.size(), (void **)&C.d_ownfirst));
      PMH_CHK(pmh_memcpy_h2d(S->ctx, C.d_ownfirst, own_first.data(), sizeof(int) * own_first.size()));
      for (int m = 0; m < C.nmb; m++) {
        if (!C.own[FXM_MB * m]) continue;
        const int sb1 = std::min(C.nsb, FXM_MB * (m + 1));
        for (int sb = FXM_MB * m; sb < sb1; sb++) tiles += (double)(sb + 1) * FXM_RT * FXM_RT * 2048.0;
        int ns = 0;
        for (int g = 0; g < C.ngroups; g++) ns = std::max(ns, nseg_of[c][(size_t)g * C.nmb + m]);
        nsegmax = std::max(nsegmax, ns);
        parts += (double)ns * (sb1 - FXM_MB * m) * FXM_RS * FXS_S * 8.0 + (double)sb1 * FXM_RS * FXS_S * 8.0; // direct sums per item + transposed sums per column
      }
      S->owned_bytes += tiles;
      // the owned tiles once per group + X read (rows + columns) + the partial sums written and read back + Y written
      S->bytes += (double)C.ngroups * (tiles + 2.0 * parts + 2.0 * 8.0 * FXS_S * C.ld);
      if (!C.d_nseg) PMH_CHK(pmh_malloc(S->ctx, sizeof(int) * nseg_of[c].size(), (void **)&C.d_nseg));
      PMH_CHK(pmh_memcpy_h2d(S->ctx, C.d_nseg, nseg_of[c].data(), sizeof(int) * nseg_of[c].size()));
      if (C.d_ptoff) pmh_free(S->ctx, C.d_ptoff), C.d_ptoff = nullptr;
      if (!C.d_ptoff) PMH_CHK(pmh_malloc(S->ctx, sizeof(long long) * ptoff_of.size(), (void **)&C.d_ptoff));
      PMH_CHK(pmh_memcpy_h2d(S->ctx, C.d_ptoff, ptoff_of.data(), sizeof(long long) * ptoff_of.size()));
    }
    S->nwg = items.empty() ? 0 : nwg, S->nseg = nsegmax;
    items.insert(items.end(), {0, 0, 0, 0, 0, 0, 0, 0});
    iteml.insert(iteml.end(), {0, 0});
    if (S->d_wg) pmh_free(S->ctx, S->d_wg);
    if (S->d_items) pmh_free(S->ctx, S->d_items);
    if (S->d_wgl) pmh_free(S->ctx, S->d_wgl);
    PMH_CHK(pmh_malloc(S->ctx, sizeof(int) * first.size(), (void **)&S->d_wg));
    PMH_CHK(pmh_memcpy_h2d(S->ctx, S->d_wg, first.data(), sizeof(int) * first.size()));
    PMH_CHK(pmh_malloc(S->ctx, sizeof(int) * items.size(), (void **)&S->d_items));
    PMH_CHK(pmh_memcpy_h2d(S->ctx, S->d_items, items.data(), sizeof(int) * items.size()));
    PMH_CHK(pmh_malloc(S->ctx, sizeof(long long) * iteml.size(), (void **)&S->d_wgl));
    PMH_CHK(pmh_memcpy_h2d(S->ctx, S->d_wgl, iteml.data(), sizeof(long long) * iteml.size()));
    const long long need = (long long)nsegmax * std::max(16LL, S->nX);
    if (need > S->part_cap) {
      if (S->part) pmh_free(S->ctx, S->part);
      PMH_CHK(pmh_malloc(S->ctx, sizeof(double) * (size_t)need, (void **)&S->part));
      S->part_cap = need;
    }
    return PMH_SUCCESS; // k_fxs_symfin reads only what the items of a mega band wrote
  }
  // segments of the rank's rows: enough of them to give the chip >= ~4000 waves (n_c / 128 column chunks each), at most 32
  int maxrows = 0, chunks = 0;
  for (auto &C : S->C) maxrows = std::max(maxrows, C.r1 - C.r0), chunks += C.ngroups * (C.ld / 128);
  int nseg = std::max(1, std::min(32, (4096 + std::max(1, chunks) - 1) / std::max(1, chunks)));
  nseg     = std::max(1, std::min(nseg, maxrows / FXS_U));
  if (const char *e = getenv("PMH_FXS_NSEG")) nseg = std::max(1, atoi(e));
  S->nseg = nseg;
  std::vector<int> wg;
  S->bytes = 0.0;
  for (int c = 0; c < S->ncls; c++) {
    fxs_class &C    = S->C[c];
    const int  rows = C.r1 - C.r0;
    for (int g = 0; g < C.ngroups; g++)
      for (int j = 0; j < nseg; j++) {
        const int lo = C.r0 + (int)((long long)rows * j / nseg), hi = C.r0 + (int)((long long)rows * (j + 1) / nseg);
        for (int c0 = 0; c0 < C.ld; c0 += 512) wg.insert(wg.end(), {c, g, c0, j, lo, hi});
      }
    // its rows once per group + X read + the segment sums written and read back + Y written
    S->bytes += (double)C.ngroups * (8.0 * (double)rows * C.ld + 8.0 * FXS_S * rows + (2.0 * nseg + 1.0) * 8.0 * FXS_S * C.ld);
  }
  S->nwg = (int)(wg.size() / 6);
  wg.insert(wg.end(), {0, 0, 0, 0, 0, 0});
  if (S->d_wg) pmh_free(S->ctx, S->d_wg);
  PMH_CHK(pmh_malloc(S->ctx, sizeof(int) * wg.size(), (void **)&S->d_wg));
  PMH_CHK(pmh_memcpy_h2d(S->ctx, S->d_wg, wg.data(), sizeof(int) * wg.size()));
  const long long need = (long long)nseg * std::max(16LL, S->nX);
  if (need > S->part_cap) {
    if (S->part) pmh_free(S->ctx, S->part);
    PMH_CHK(pmh_malloc(S->ctx, sizeof(double) * (size_t)need, (void **)&S->part));
    S->part_cap = need;
  }
  return pmh_memset(S->ctx, S->part, 0, sizeof(double) * (size_t)need); // column chunks beyond a class's ld / empty segments stay zero
}

// extra_ptr / extra_rel (optional): block-relative dofs ADDED to the touched set of class c (extra_rel[extra_ptr[c] .. extra_ptr[c + 1])): the closure of the touched set under the
// block's symmetry group (pmh_box_symmetry_closure), so that a class whose own touched set is not invariant keeps all its symmetries
int fxs_create(pmh_gluing B, pmh_blockdiag K, const int *block_class, int sym, fx_shared **out, const int *extra_ptr, const int *extra_rel)
{
  PMH_ARG(B && K && block_class && out && B->n_x == K->n);
  pmh_ctx    ctx = B->ctx;
  fx_shared *S   = new fx_shared();
  S->ctx = ctx, S->B = B, S->K = K, S->nb = K->nblocks, S->sym = sym;
  S->cls.assign(block_class, block_class + S->nb);
  S->ncls = 0;
  for (int b = 0; b < S->nb; b++) {
    PMH_ARG(block_class[b] >= 0);
    S->ncls = std::max(S->ncls, block_class[b] + 1);
  }
  S->C.resize(S->ncls);
  std::vector<int> slot(S->nb), group(S->nb);
  for (int b = 0; b < S->nb; b++) {
    fxs_class &C  = S->C[S->cls[b]];
    const int  nl = K->rowstart[b + 1] - K->rowstart[b];
    if (C.blocks.empty()) C.nloc = nl;
    else if (C.nloc != nl) return pmh_set_error(PMH_ERR_ARG, "pmh_fexplicit_create_shared: blocks of class %d differ in size", S->cls[b]);
    slot[b] = (int)C.blocks.size() % FXS_S, group[b] = (int)C.blocks.size() / FXS_S;
    C.blocks.push_back(b);
  }
  S->mfma16 = fxo_mfma16();
  if (sym == 2 && S->mfma16 && !getenv("PMH_FXO_SLOTS8"))
    for (auto &C : S->C) {
      C.S = 1;
      while (C.S < FXS_S && C.S < (int)C.blocks.size()) C.S *= 2;
    }
  // union of the touched dofs per class
  for (int c = 0; c < S->ncls; c++) S->C[c].pos.assign((size_t)std::max(1, S->C[c].nloc), -1);
  auto block_of = [&](int i) { return (int)(std::upper_bound(K->rowstart.begin(), K->rowstart.end(), i) - K->rowstart.begin()) - 1; };
  std::vector<int> lb((size_t)std::max(1, B->n_leaves));
  for (int i = 0; i < B->n_leaves; i++) {
    lb[i] = block_of(B->h_row[i]);
    S->C[S->cls[lb[i]]].pos[B->h_row[i] - K->rowstart[lb[i]]] = 0;
  }
  if (extra_ptr && extra_rel)
    for (int c = 0; c < S->ncls; c++)
      for (int e = extra_ptr[c]; e < extra_ptr[c + 1]; e++) {
        if (extra_rel[e] < 0 || extra_rel[e] >= S->C[c].nloc) return pmh_set_error(PMH_ERR_ARG, "pmh_fexplicit_create_shared: extra dof %d of class %d is outside its blocks (%d rows)", extra_rel[e], c, S->C[c].nloc);
        S->C[c].pos[extra_rel[e]] = 0;
      }
  long long wtot = 0, xtot = 0, pttot = 0;
  for (int c = 0; c < S->ncls; c++) {
    fxs_class &C = S->C[c];
    for (int i = 0; i < C.nloc; i++)
      if (C.pos[i] == 0) C.pos[i] = (int)C.urel.size(), C.urel.push_back(i);
    C.nc      = (int)C.urel.size();
    const int pad = sym == 2 ? 32 : (sym ? FXM_RS : FXS_PAD);
    C.ld      = (C.nc + pad - 1) / pad * pad;
    C.ngroups = ((int)C.blocks.size() + FXS_S - 1) / FXS_S;
    C.r0 = 0, C.r1 = C.ld;
    C.woff = wtot, C.xoff = xtot;
    if (sym == 2) {
      if (C.ld == C.nc) C.ld += 32; // a zero row of X behind the touched dofs for the padded k range of the GEMM
    } else if (sym) {
      C.nsb = C.ld / FXM_RS, C.nmb = (C.nsb + FXM_MB - 1) / FXM_MB;
      C.own.assign((size_t)std::max(1, C.nsb), 1);
      C.ptm.assign((size_t)C.nmb + 1, 0);
      for (int m = 0; m < C.nmb; m++) C.ptm[m + 1] = C.ptm[m] + (long long)std::min(C.nsb, FXM_MB * (m + 1)) * FXM_RS * FXS_S;
      C.ptoff = pttot, C.ptsize = C.ptm[C.nmb];
      pttot += C.ngroups * C.ptsize;
      wtot += (long long)FXM_RS * FXM_RS * ((long long)C.nsb * (C.nsb + 1) / 2); // super band sb: (sb + 1) * 16 column tiles x 16 row tiles x 256 doubles
    } else
      wtot += (long long)C.ld * C.ld;
    xtot += (long long)C.ngroups * C.ld * C.S;
    PMH_CHK(pmh_malloc(ctx, sizeof(int) * (size_t)std::max(1, C.nc), (void **)&C.d_urel));
    if (C.nc) PMH_CHK(pmh_memcpy_h2d(ctx, C.d_urel, C.urel.data(), sizeof(int) * (size_t)C.nc));
  }
  if (xtot >= (1LL << 31)) return pmh_set_error(PMH_ERR_SUP, "pmh_fexplicit_create_shared: the multivector numbering exceeds 32-bit indices");
  S->nX = xtot, S->wtot = wtot;
  // gluing over the multivector numbering: leaf of block b at relative dof i -> (position of i in U_c) * 8 + slot(b), group by group
  std::vector<int> rows((size_t)std::max(1, B->n_leaves));
  for (int i = 0; i < B->n_leaves; i++) {
    const int        b = lb[i];
    const fxs_class &C = S->C[S->cls[b]];
    rows[i]            = (int)(C.xoff + (long long)group[b] * C.ld * C.S + (long long)C.pos[B->h_row[i] - K->rowstart[b]] * C.S + slot[b]);
  }
  if (sym == 2) {
    for (auto &C : S->C) C.tmask.assign((size_t)std::max(1, C.ngroups) * std::max(1, C.nc) * FXS_S, 0);
    for (int i = 0; i < B->n_leaves; i++) {
      const int  b = lb[i];
      fxs_class &C = S->C[S->cls[b]];
      C.tmask[((size_t)group[b] * C.nc + C.pos[B->h_row[i] - K->rowstart[b]]) * FXS_S + slot[b]] = 1;
    }
  }
  PMH_CHK(pmh_gluing_create(ctx, (int)std::max(1LL, xtot), B->n_lambda, B->n_leaves, rows.data(), B->h_root.data(), B->h_sign.data(), &S->Bc));
  if (sym == 2) {
    if (2 * xtot >= (1LL << 31)) return pmh_set_error(PMH_ERR_SUP, "pmh_fexplicit_create_shared_orbit: the signed multivector numbering exceeds 32-bit indices");
    std::vector<int>    rows2((size_t)std::max(1, 2 * B->n_leaves)), root2((size_t)std::max(1, 2 * B->n_leaves));
    std::vector<double> sign2((size_t)std::max(1, 2 * B->n_leaves));
    for (int i = 0; i < B->n_leaves; i++) {
      const int slot_i = slot[lb[i]], base = rows[i] - slot_i, Sc = S->C[S->cls[lb[i]]].S; // entry (position, slot) -> (position, +, slot) and (position, -, slot)
      rows2[2 * i] = 2 * base + slot_i, rows2[2 * i + 1] = 2 * base + Sc + slot_i;
      root2[2 * i] = root2[2 * i + 1] = B->h_root[i];
      sign2[2 * i] = B->h_sign[i], sign2[2 * i + 1] = -B->h_sign[i];
    }
    PMH_CHK(pmh_gluing_create(ctx, (int)std::max(1LL, 2 * xtot), B->n_lambda, 2 * B->n_leaves, rows2.data(), root2.data(), sign2.data(), &S->Bc2));
    PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)std::max(32LL, 2 * xtot), (void **)&S->X2));
    PMH_CHK(pmh_memset(ctx, S->X2, 0, sizeof(double) * (size_t)std::max(32LL, 2 * xtot)));
  }
  {
    const size_t bytes = sizeof(double) * (size_t)std::max(32LL, wtot);
    hipError_t   e     = hipMalloc((void **)&S->Wbase, bytes);
    if (e != hipSuccess) return pmh_set_error(PMH_ERR_HIP, "pmh_fexplicit_create_shared: %.2f GB for the shared explicit operators: %s", bytes / 1e9, hipGetErrorString(e));
    PMH_HIP(hipMemsetAsync(S->Wbase, 0, bytes, ctx->stream));
  }
  if (sym == 1) {
    S->pt_tot = pttot;
    PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)std::max(16LL, pttot), (void **)&S->pt));
  }
  PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)std::max(16LL, xtot), (void **)&S->X));
  PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)std::max(16LL, xtot), (void **)&S->Y));
  PMH_CHK(pmh_memset(ctx, S->X, 0, sizeof(double) * (size_t)std::max(16LL, xtot)));
  PMH_CHK(pmh_memset(ctx, S->Y, 0, sizeof(double) * (size_t)std::max(16LL, xtot))); // rows of other ranks' stripes stay zero
  std::vector<int>       ldv(S->ncls);
  std::vector<long long> wo(S->ncls), xo(S->ncls);
  for (int c = 0; c < S->ncls; c++) ldv[c] = S->C[c].ld, wo[c] = S->C[c].woff, xo[c] = S->C[c].xoff;
  PMH_CHK(pmh_malloc(ctx, sizeof(int) * S->ncls, (void **)&S->d_ld));
  PMH_CHK(pmh_malloc(ctx, sizeof(long long) * S->ncls, (void **)&S->d_woff));
  PMH_CHK(pmh_malloc(ctx, sizeof(long long) * S->ncls, (void **)&S->d_xoff));
  PMH_CHK(pmh_memcpy_h2d(ctx, S->d_ld, ldv.data(), sizeof(int) * S->ncls));
  PMH_CHK(pmh_memcpy_h2d(ctx, S->d_woff, wo.data(), sizeof(long long) * S->ncls));
  PMH_CHK(pmh_memcpy_h2d(ctx, S->d_xoff, xo.data(), sizeof(long long) * S->ncls));
  if (sym != 2) PMH_CHK(fxs_build_launch(S)); // orbit storage: planned once the symmetries are known (fxo_prepare)
  *out = S;
  return PMH_SUCCESS;
}

void fxs_destroy(fx_shared *S)
{
  if (!S) return;
  pmh_ctx ctx = S->ctx;
  for (auto &C : S->C) {
    if (C.d_urel) pmh_free(ctx, C.d_urel);
    if (C.d_nseg) pmh_free(ctx, C.d_nseg);
    if (C.d_ptoff) pmh_free(ctx, C.d_ptoff);
    if (C.d_ownfirst) pmh_free(ctx, C.d_ownfirst);
    if (C.d_posmap) pmh_free(ctx, C.d_posmap), pmh_free(ctx, C.d_sign);
    if (C.d_gidx) pmh_free(ctx, C.d_gidx), pmh_free(ctx, C.d_reppos), pmh_free(ctx, C.d_use);
    if (C.d_coltab) pmh_free(ctx, C.d_coltab), pmh_free(ctx, C.d_fintab), pmh_free(ctx, C.d_finbase);
    if (C.d_kinv) pmh_free(ctx, C.d_kinv), pmh_free(ctx, C.d_unittab), pmh_free(ctx, C.d_lut);
  }
  if (S->pt) pmh_free(ctx, S->pt);
  if (S->d_wgl) pmh_free(ctx, S->d_wgl);
  if (S->d_items) pmh_free(ctx, S->d_items);
  if (S->d_wgfirst) pmh_free(ctx, S->d_wgfirst);
  if (S->d_wgfirst_all) pmh_free(ctx, S->d_wgfirst_all);
  if (S->d_fin_args) pmh_free(ctx, S->d_fin_args);
  if (S->d_zrow_of) pmh_free(ctx, S->d_zrow_of), pmh_free(ctx, (void *)S->d_coltab_of), pmh_free(ctx, (void *)S->d_gidx_of);
  pmh_gluing_destroy(S->Bc);
  for (auto e : S->ev_mid) (void)hipEventDestroy(e);
  pmh_gluing_destroy(S->Bc2);
  pmh_free(ctx, S->X2);
  if (S->Wbase) (void)hipFree(S->Wbase);
  if (S->Afund) (void)hipFree(S->Afund);
  if (S->cpart) pmh_free(ctx, S->cpart);
  if (S->part) pmh_free(ctx, S->part);
  pmh_free(ctx, S->X), pmh_free(ctx, S->Y), pmh_free(ctx, S->d_wg), pmh_free(ctx, S->d_ld), pmh_free(ctx, S->d_woff), pmh_free(ctx, S->d_xoff);
  for (hipEvent_t e : S->ev) (void)hipEventDestroy(e);
  delete S;
}

// host helper (no device): the row tile fxo_prepare picks for a class with M orbit representatives and the padded row count of its GEMM
extern "C" int pmh_fexplicit_orbit_row_tile(int M, int *tm, int *Mp)
{
  PMH_ARG(M >= 1);
  const int t = fxo_row_tile(M);
  if (tm) *tm = t;
  if (Mp) *Mp = (M + t - 1) / t * t;
  return PMH_SUCCESS;
}

// the dealing rule of the symmetric tile storage: mega band m of nmb (1024 rows; cost ~ m + 1) -> rank: from the longest down in snake order
static inline int fxm_owner(int m, int nmb, int size)
{
  const int i = nmb - 1 - m, round = i / size, k = i % size;
  return (round & 1) ? size - 1 - k : k;
}

// host helper (no device): the owner rank of every mega band of a class with n_c touched dofs and the tile bytes per rank under that rule
extern "C" int pmh_fexplicit_class_sym_plan(int n_c, int size, int *owner_out, double *bytes_per_rank)
{
  PMH_ARG(n_c >= 1 && size >= 1);
  const int nsb = (n_c + FXM_RS - 1) / FXM_RS, nmb = (nsb + FXM_MB - 1) / FXM_MB;
  if (bytes_per_rank)
    for (int r = 0; r < size; r++) bytes_per_rank[r] = 0.0;
  for (int m = 0; m < nmb; m++) {
    const int o = fxm_owner(m, nmb, size);
    if (owner_out) owner_out[m] = o;
    if (bytes_per_rank)
      for (int sb = FXM_MB * m; sb < std::min(nsb, FXM_MB * (m + 1)); sb++) bytes_per_rank[o] += (double)(sb + 1) * FXM_RT * FXM_RT * 2048.0;
  }
  return PMH_SUCCESS;
}

// several GPUs: rank r applies / assembles the rows [r0, r1) of every W_c, contiguous ranges of equal length (multiples of 32)
int fxs_set_stripe(fx_shared *S, int rank, int size)
{
  if (S->sym == 2) { // orbit storage: a contiguous range of the representatives per rank, planned by fxo_prepare
    S->stripe_rank = rank, S->stripe_size = size, S->fxo_ready = 0;
    return PMH_SUCCESS;
  }
  if (S->sym) {
    // whole mega bands of 1024 rows (a rank assembles exactly the rows it applies); mega band m costs ~ m + 1: dealt from the longest down in
    // snake order, so every rank gets the same number of long and short ones
    for (auto &C : S->C) {
      for (int m = 0; m < C.nmb; m++)
        for (int sb = FXM_MB * m; sb < std::min(C.nsb, FXM_MB * (m + 1)); sb++) C.own[sb] = fxm_owner(m, C.nmb, size) == rank;
    }
    return fxs_build_launch(S);
  }
  for (auto &C : S->C) {
    const int nrg = C.ld / 32; // row groups of 32
    C.r0 = (int)((long long)nrg * rank / size) * 32;
    C.r1 = (int)((long long)nrg * (rank + 1) / size) * 32;
  }
  return fxs_build_launch(S);
}

long long fxs_dense_bytes(fx_shared *S) { return S->sym ? (long long)S->owned_bytes : (long long)sizeof(double) * S->wtot; }
double    fxs_apply_flops(fx_shared *S) { return S->sym == 2 ? S->flops : 0.0; } // orbit storage: the GEMM's useful flops per apply
void      fxs_apply_flops_detail(fx_shared *S, double *issued, double *dense) { *issued = S->sym == 2 ? S->flops_issued : 0.0, *dense = S->sym == 2 ? S->flops_dense : 0.0; }
double    fxs_apply_bytes(fx_shared *S) { return S->bytes; }

// the touched dofs of class c, ascending, relative to the block start (the numbering of W_c's rows)
int fxs_class_union(fx_shared *S, int c, int *n_c, int *urel_out)
{
  PMH_ARG(S && c >= 0 && c < S->ncls);
  if (n_c) *n_c = S->C[c].nc;
  if (urel_out && S->C[c].nc) memcpy(urel_out, S->C[c].urel.data(), sizeof(int) * (size_t)S->C[c].nc);
  return PMH_SUCCESS;
}

// Set-up by symmetry: nsym signed permutations of U_c (posmap[g * n_c + c] = position of the image of the c-th touched dof, sign = +-1; operation 0 the
// identity) under which K_c, hence K_c^+, is invariant: W[g p][g c] = sign_g[p] sign_g[c] W[p][c], so ONE K^+ solve serves the whole orbit of a row
// (a cube of Q1 elasticity elements: the 48 signed coordinate permutations -> 48 x fewer solves).  The caller vouches for the invariance (permon_amd
// checks the generators against K); the assembly re-solves a handful of symmetry-filled rows directly and fails if they differ.
int fxs_set_symmetry(fx_shared *S, int c, int nsym, const int *posmap, const signed char *sign)
{
  PMH_ARG(S && c >= 0 && c < S->ncls && nsym >= 1 && posmap && sign);
  if (!S->sym) return pmh_set_error(PMH_ERR_SUP, "pmh_fexplicit_set_class_symmetry: needs the PMH_FX_CLASS_SYM or PMH_FX_CLASS_ORBIT storage");
  S->fxo_ready = 0;
  fxs_class &C  = S->C[c];
  const int  nc = C.nc;
  std::vector<char> seen((size_t)std::max(1, nc));
  for (int g = 0; g < nsym; g++) {
    std::fill(seen.begin(), seen.end(), 0);
    for (int i = 0; i < nc; i++) {
      const int t = posmap[(size_t)g * nc + i];
      if (t < 0 || t >= nc || seen[t] || (sign[(size_t)g * nc + i] != 1 && sign[(size_t)g * nc + i] != -1))
        return pmh_set_error(PMH_ERR_ARG, "pmh_fexplicit_set_class_symmetry: operation %d is not a signed permutation of the %d touched dofs", g, nc);
      if (g == 0 && (t != i || sign[i] != 1)) return pmh_set_error(PMH_ERR_ARG, "pmh_fexplicit_set_class_symmetry: operation 0 must be the identity");
      seen[t] = 1;
    }
  }
  C.nsym = nsym;
  C.h_posmap.assign(posmap, posmap + (size_t)nsym * nc);
  C.h_sign.assign(sign, sign + (size_t)nsym * nc);
  if (C.d_posmap) pmh_free(S->ctx, C.d_posmap), pmh_free(S->ctx, C.d_sign);
  PMH_CHK(pmh_malloc(S->ctx, sizeof(int) * (size_t)std::max(1, nsym * nc), (void **)&C.d_posmap));
  PMH_CHK(pmh_malloc(S->ctx, (size_t)std::max(1, nsym * nc), (void **)&C.d_sign));
  if (nc) {
    PMH_CHK(pmh_memcpy_h2d(S->ctx, C.d_posmap, C.h_posmap.data(), sizeof(int) * (size_t)nsym * nc));
    PMH_CHK(pmh_memcpy_h2d(S->ctx, C.d_sign, C.h_sign.data(), (size_t)nsym * nc));
  }
  return PMH_SUCCESS;
}

// ---- orbit storage: plan (after the symmetries and the stripe are known) ----------------------------------------------------------------------
struct fxo_unit { // a (group, row tile, k segment): its columns (a sub-list of the tile's), its chunks on this rank, its splits and partial tiles
  int       g = 0, mt = 0, seg = 0, coff = 0, nct = 0, listed = 0, lutoff = 0, kc0 = 0, kc1 = 0, S = 0;
  long long cbase = 0;
};
struct fxo_plan {
  std::vector<fxo_unit> units;
  std::vector<int>      coltab, fintab, lut, segc0;
};

static int fxo_prepare(fx_shared *S)
{
  if (S->fxo_ready) return PMH_SUCCESS;
  pmh_ctx   ctx = S->ctx;
  long long atot = 0;
  std::vector<fxo_plan> plan; // per class with touched dofs, in class order
  std::vector<int>      tab_of(S->ncls, -1);
  for (int c = 0; c < S->ncls; c++) {
    fxs_class &C = S->C[c];
    if (C.nc == 0) continue;
    tab_of[c] = (int)plan.size();
    if (C.nsym < 1) return pmh_set_error(PMH_ERR_STATE, "PMH_FX_CLASS_ORBIT: block class %d has no symmetries (pmh_fexplicit_set_class_symmetry / _set_box_symmetry before the assembly)", c);
    // orbits of the rows: representative and operation of every row; rows fixed by several operations keep the first
    C.rep_of.assign((size_t)C.nc, -1), C.op_of.assign((size_t)C.nc, 0), C.reps.clear();
    for (int p = 0; p < C.nc; p++) {
      if (C.rep_of[p] >= 0) continue;
      C.reps.push_back(p);
      for (int g = 0; g < C.nsym; g++) {
        const int r = C.h_posmap[(size_t)g * C.nc + p];
        if (C.rep_of[r] < 0) C.rep_of[r] = p, C.op_of[r] = g;
      }
    }
    C.M_all = (int)C.reps.size();
    // several GPUs: every rank keeps ALL representatives' rows (0.19 GB for configs[2]; it solves for them itself) and multiplies a contiguous share of
    // the k range (the columns of W): full tiles at every N, and the partial Y are summed by the all-reduce that ends B Y anyway
    C.m0 = 0, C.m1 = C.M_all;
    const int M = C.m1 - C.m0;
    C.tm   = fxo_row_tile(M);
    C.tnw  = 0;
    if (S->mfma16 && C.S == 1 && C.nsym * C.S <= 48 && !getenv("PMH_FXO_NO_TN48")) {
      // a class of ONE block lists at most 48 columns: the 64-wide tile multiplies a quarter of zeros.  48 columns x (4 waves x NI x 16 rows): 192 rows unless fewer pad less
      C.tnw = 48, C.tm = 192;
      for (int tm : {128, 64})
        if ((M + tm - 1) / tm * tm < (M + C.tm - 1) / C.tm * C.tm) C.tm = tm;
    }
    C.Mp   = std::max(1, (M + C.tm - 1) / C.tm) * C.tm;
    C.nsymp = (C.nsym + FXO_TN / 8 - 1) / (FXO_TN / 8) * (FXO_TN / 8);
    std::vector<int>         reppos((size_t)C.Mp, 0);
    std::vector<signed char> use((size_t)C.Mp * C.nsymp, 0), use_h((size_t)M * C.nsym, 0);
    for (int pl = 0; pl < M; pl++) {
      const int p = C.reps[C.m0 + pl];
      for (int g = 0; g < C.nsym; g++) {
        const int r = C.h_posmap[(size_t)g * C.nc + p];
        if (C.rep_of[r] == p && C.op_of[r] == g) use_h[(size_t)pl * C.nsym + g] = C.h_sign[(size_t)g * C.nc + p];
      }
    }
    // need pattern of a representative: bit ((group * nsym + g) * 8 + slot) = row g p is owned by (p, g) and block (group, slot) touches it.  The rows of A follow
    // the patterns (the widest first), so that a row tile holds few patterns and its column list stays short: a face-interior representative of a 2 x 2 x 2
    // decomposition needs 224 or 256 of the 384 columns
    const bool   prune = !getenv("PMH_FXO_NO_PRUNE") && !C.tmask.empty();
    const size_t nbits = (size_t)C.ngroups * C.nsym * FXS_S, nw = (nbits + 63) / 64;
    std::vector<unsigned long long> pat((size_t)M * nw, 0ULL);
    std::vector<int>                cnt((size_t)M, 0), rowrep((size_t)M);
    for (int pl = 0; pl < M; pl++) {
      const int p = C.reps[C.m0 + pl];
      for (int gr = 0; gr < C.ngroups; gr++)
        for (int g = 0; g < C.nsym; g++) {
          if (!use_h[(size_t)pl * C.nsym + g]) continue;
          const int r = C.h_posmap[(size_t)g * C.nc + p];
          for (int sl = 0; sl < FXS_S; sl++)
            if (!prune || C.tmask[((size_t)gr * C.nc + r) * FXS_S + sl]) {
              const size_t b = ((size_t)gr * C.nsym + g) * FXS_S + sl;
              pat[(size_t)pl * nw + b / 64] |= 1ULL << (b % 64), cnt[pl]++;
            }
        }
      rowrep[pl] = pl;
    }
    if (prune) {
      // rows with the same pattern together; the RARE patterns first (representatives on the cube's edges and corners need other columns than the face-interior
      // ones: they share the first row tile, whose list is the full one anyway), then the common ones, the wider first
      auto less_pat = [&](int a, int b) {
        return std::lexicographical_compare(pat.begin() + (size_t)a * nw, pat.begin() + (size_t)(a + 1) * nw, pat.begin() + (size_t)b * nw, pat.begin() + (size_t)(b + 1) * nw);
      };
      std::stable_sort(rowrep.begin(), rowrep.end(), less_pat);
      std::vector<int> gsize((size_t)M, 0); // size of the pattern group a representative belongs to
      for (int i = 0; i < M;) {
        int j = i + 1;
        while (j < M && !less_pat(rowrep[i], rowrep[j]) && !less_pat(rowrep[j], rowrep[i])) j++;
        for (int k = i; k < j; k++) gsize[rowrep[k]] = j - i;
        i = j;
      }
      std::stable_sort(rowrep.begin(), rowrep.end(), [&](int a, int b) {
        if (gsize[a] != gsize[b]) return gsize[a] < gsize[b];
        if (cnt[a] != cnt[b]) return cnt[a] > cnt[b];
        return less_pat(a, b);
      });
    }
    C.reprow.assign((size_t)M, 0);
    for (int row = 0; row < M; row++) {
      const int pl = rowrep[row];
      C.reprow[pl] = row, reppos[row] = C.reps[C.m0 + pl];
      for (int g = 0; g < C.nsym; g++) use[(size_t)row * C.nsymp + g] = use_h[(size_t)pl * C.nsym + g];
    }
    // column lists per (group, row tile): the columns some row of the tile needs (what k_fxo_fin walks)
    const int        ntile = C.Mp / C.tm, ncode = C.nsym * FXS_S, cw = (ncode + 63) / 64;
    std::vector<int> coltab, fintab((size_t)C.ngroups * (ntile + 1) * 4, 0);
    std::vector<unsigned long long> need((size_t)C.ngroups * ntile * cw, 0ULL); // the same lists as bit sets over the codes
    { // the class's column tile: 64 when no (group, row tile) lists more than 64 columns -- a class of ONE block lists at most its 48 operations, and a 128-wide tile would
      // multiply 80 columns of zeros (PMH_FXO_TN=128 keeps the wide tile for the A/B)
      int most = 0;
      for (int gr = 0; gr < C.ngroups; gr++)
        for (int mt = 0; mt < ntile; mt++) {
          int n = 0;
          for (int code = 0; code < ncode; code++) {
            const size_t b   = ((size_t)gr * C.nsym + (code >> 3)) * FXS_S + (code & 7);
            bool         any = false;
            for (int row = mt * C.tm; row < std::min(M, (mt + 1) * C.tm) && !any; row++) any = (pat[(size_t)rowrep[row] * nw + b / 64] >> (b % 64)) & 1ULL;
            n += any;
          }
          most = std::max(most, n);
        }
      const char *e = getenv("PMH_FXO_TN");
      C.tn = (S->mfma16 && C.S != FXS_S && most <= 64 && !(e && atoi(e) == 128)) ? 64 : 128; // (only classes on the table-driven kernel: the single-class kernel of 8-block classes is left as it is)
    }
    for (int gr = 0; gr < C.ngroups; gr++) {
      int elems = 0;
      for (int mt = 0; mt < ntile; mt++) {
        const int coff = (int)coltab.size();
        for (int code = 0; code < ncode; code++) {
          const size_t b   = ((size_t)gr * C.nsym + (code >> 3)) * FXS_S + (code & 7);
          bool         any = false;
          for (int row = mt * C.tm; row < std::min(M, (mt + 1) * C.tm) && !any; row++) any = (pat[(size_t)rowrep[row] * nw + b / 64] >> (b % 64)) & 1ULL;
          if (any) coltab.push_back(code), need[((size_t)gr * ntile + mt) * cw + code / 64] |= 1ULL << (code % 64);
        }
        while ((coltab.size() - coff) % C.tn) coltab.push_back(-1);
        int *ft = fintab.data() + ((size_t)gr * (ntile + 1) + mt) * 4;
        ft[0] = coff, ft[1] = (int)coltab.size() - coff, ft[2] = elems;
        elems += C.tm * ft[1];
      }
      fintab[((size_t)gr * (ntile + 1) + ntile) * 4 + 2] = elems;
      C.fin_elems = std::max(gr ? C.fin_elems : 0, elems);
    }
    // k segments: B[c][(g, slot)] = s_g(c) X[g c][slot] is structurally zero where block (group, slot) does not touch g c.  The signature of a position is the set of
    // columns that are NOT zero there; positions of one signature form a segment (the interior of a face of the cube with one dof component, ...), small ones are pooled,
    // and two segments are joined whenever that does not add column tiles (fewer, longer units split more evenly).  The k index of the product runs segment after segment,
    // each padded to whole chunks, and a (row tile, segment) unit multiplies only the columns of the tile's list that are non-zero on the segment: for a 2 x 2 x 2
    // decomposition a face segment keeps 128 ... 256 of the 384 columns.  PMH_FXO_NO_KSEG=1: one segment (every listed column over the whole k range).
    const size_t sw = (size_t)C.ngroups * cw;
    std::vector<std::vector<unsigned long long>> ssig;
    std::vector<std::vector<int>>                spos;
    if (prune && !getenv("PMH_FXO_NO_KSEG")) {
      std::map<std::vector<unsigned long long>, int> ids;
      std::vector<unsigned long long>                sg(sw);
      for (int cc = 0; cc < C.nc; cc++) {
        std::fill(sg.begin(), sg.end(), 0ULL);
        for (int gr = 0; gr < C.ngroups; gr++)
          for (int g = 0; g < C.nsym; g++) {
            const char *tm8 = &C.tmask[((size_t)gr * C.nc + C.h_posmap[(size_t)g * C.nc + cc]) * FXS_S];
            for (int sl = 0; sl < FXS_S; sl++)
              if (tm8[sl]) sg[(size_t)gr * cw + (g * FXS_S + sl) / 64] |= 1ULL << ((g * FXS_S + sl) % 64);
          }
        auto it = ids.find(sg);
        if (it == ids.end()) it = ids.emplace(sg, (int)ssig.size()).first, ssig.push_back(sg), spos.emplace_back();
        spos[it->second].push_back(cc);
      }
      const int minseg = getenv("PMH_FXO_SEGMIN") ? std::max(1, atoi(getenv("PMH_FXO_SEGMIN"))) : std::max(2 * FXO_TK, C.nc / 64);
      auto join = [&](size_t a, size_t b) { // b into a
        for (size_t w = 0; w < sw; w++) ssig[a][w] |= ssig[b][w];
        spos[a].insert(spos[a].end(), spos[b].begin(), spos[b].end());
        ssig.erase(ssig.begin() + b), spos.erase(spos.begin() + b);
      };
      long long pool = -1; // the small segments together
      for (size_t i = 0; i < spos.size();) {
        if ((int)spos[i].size() >= minseg) { i++; continue; }
        if (pool < 0) pool = (long long)i++;
        else join((size_t)pool, i);
      }
      auto cost = [&](const std::vector<unsigned long long> &sig, size_t npos) { // chunks x column tiles over the (group, row tile) pairs
        long long tiles = 0;
        for (int gr = 0; gr < C.ngroups; gr++)
          for (int mt = 0; mt < ntile; mt++) {
            int n = 0;
            for (int w = 0; w < cw; w++) n += __builtin_popcountll(need[((size_t)gr * ntile + mt) * cw + w] & sig[(size_t)gr * cw + w]);
            tiles += (n + C.tn - 1) / C.tn;
          }
        return (long long)((npos + FXO_TK - 1) / FXO_TK) * tiles;
      };
      for (;;) { // greedy: the pair whose union saves most (>= 0: equal cost still gives fewer, longer units)
        long long best = -1;
        size_t    ba = 0, bb = 0;
        std::vector<unsigned long long> un(sw);
        for (size_t a2 = 0; a2 < spos.size(); a2++)
          for (size_t b2 = a2 + 1; b2 < spos.size(); b2++) {
            for (size_t w = 0; w < sw; w++) un[w] = ssig[a2][w] | ssig[b2][w];
            const long long save = cost(ssig[a2], spos[a2].size()) + cost(ssig[b2], spos[b2].size()) - cost(un, spos[a2].size() + spos[b2].size());
            if (save > best) best = save, ba = a2, bb = b2;
          }
        if (best < 0) break;
        join(ba, bb);
      }
      for (auto &v : spos) std::sort(v.begin(), v.end());
      std::vector<size_t> order(spos.size());
      for (size_t i = 0; i < order.size(); i++) order[i] = i;
      std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return spos[x].size() != spos[y].size() ? spos[x].size() > spos[y].size() : spos[x][0] < spos[y][0]; }); // the long segments first
      std::vector<std::vector<unsigned long long>> s2;
      std::vector<std::vector<int>>                p2;
      for (size_t i : order) s2.push_back(ssig[i]), p2.push_back(spos[i]);
      ssig.swap(s2), spos.swap(p2);
    } else {
      ssig.emplace_back(sw, ~0ULL), spos.emplace_back((size_t)C.nc);
      for (int cc = 0; cc < C.nc; cc++) spos[0][cc] = cc;
    }
    C.nseg = (int)spos.size();
    std::vector<int> segc0((size_t)C.nseg + 1, 0); // first chunk of every segment
    C.kinv.assign((size_t)C.nc, 0);
    for (int sg = 0; sg < C.nseg; sg++) {
      for (size_t i = 0; i < spos[sg].size(); i++) C.kinv[spos[sg][i]] = segc0[sg] * FXO_TK + (int)i;
      segc0[sg + 1] = segc0[sg] + ((int)spos[sg].size() + FXO_TK - 1) / FXO_TK;
    }
    C.nkc  = std::max(1, segc0[C.nseg]);
    C.ldk  = C.nkc * FXO_TK;
    C.aoff = atot;
    atot += (long long)C.Mp * C.ldk;
    // gather indices of B: (position of g c) << 1 | (s_g(c) < 0) at row kinv[c]; padded k and padded operations read the zero row nc of X
    // (one more row of gather indices, all on the zero row of X: what the padding columns of the lists below read)
    std::vector<int> gidx((size_t)(C.nsymp + 1) * C.ldk, C.nc << 1);
    for (int g = 0; g < C.nsym; g++)
      for (int cc = 0; cc < C.nc; cc++) gidx[(size_t)g * C.ldk + C.kinv[cc]] = (C.h_posmap[(size_t)g * C.nc + cc] << 1) | (C.h_sign[(size_t)g * C.nc + cc] < 0 ? 1 : 0);
    // the units: (group, row tile, segment) with the columns of the tile's list that are non-zero on the segment; look-up table from the tile's list
    fxo_plan P;
    P.segc0 = segc0;
    for (int gr = 0; gr < C.ngroups; gr++)
      for (int mt = 0; mt <= ntile; mt++) {
        fintab[((size_t)gr * (ntile + 1) + mt) * 4 + 3] = (int)P.units.size();
        if (mt == ntile) break;
        const int *ft = fintab.data() + ((size_t)gr * (ntile + 1) + mt) * 4;
        for (int sg = 0; sg < C.nseg; sg++) {
          fxo_unit U;
          U.g = gr, U.mt = mt, U.seg = sg, U.coff = (int)coltab.size(), U.lutoff = (int)P.lut.size();
          int n = 0;
          for (int j = 0; j < ft[1]; j++) {
            const int  code = coltab[(size_t)ft[0] + j];
            const bool in   = code >= 0 && ((ssig[sg][(size_t)gr * cw + code / 64] >> (code % 64)) & 1ULL);
            P.lut.push_back(in ? n : -1);
            if (in) coltab.push_back(code), n++;
          }
          U.listed = n;
          while ((coltab.size() - U.coff) % C.tn) coltab.push_back(-1);
          U.nct = (int)coltab.size() - U.coff;
          P.units.push_back(U);
        }
      }
    P.coltab = coltab, P.fintab = fintab;
    plan.push_back(P);
    if (C.d_coltab) pmh_free(ctx, C.d_coltab), pmh_free(ctx, C.d_fintab), C.d_coltab = nullptr;
    if (C.d_kinv) pmh_free(ctx, C.d_kinv), pmh_free(ctx, C.d_lut), C.d_kinv = nullptr;
    PMH_CHK(pmh_malloc(ctx, sizeof(int) * std::max<size_t>(1, coltab.size()), (void **)&C.d_coltab));
    PMH_CHK(pmh_malloc(ctx, sizeof(int) * fintab.size(), (void **)&C.d_fintab));
    PMH_CHK(pmh_malloc(ctx, sizeof(int) * std::max<size_t>(1, C.kinv.size()), (void **)&C.d_kinv));
    PMH_CHK(pmh_malloc(ctx, sizeof(int) * std::max<size_t>(1, P.lut.size()), (void **)&C.d_lut));
    if (!coltab.empty()) PMH_CHK(pmh_memcpy_h2d(ctx, C.d_coltab, coltab.data(), sizeof(int) * coltab.size()));
    PMH_CHK(pmh_memcpy_h2d(ctx, C.d_fintab, fintab.data(), sizeof(int) * fintab.size()));
    if (!C.kinv.empty()) PMH_CHK(pmh_memcpy_h2d(ctx, C.d_kinv, C.kinv.data(), sizeof(int) * C.kinv.size()));
    if (!P.lut.empty()) PMH_CHK(pmh_memcpy_h2d(ctx, C.d_lut, P.lut.data(), sizeof(int) * P.lut.size()));
    if (C.d_gidx) pmh_free(ctx, C.d_gidx), pmh_free(ctx, C.d_reppos), pmh_free(ctx, C.d_use);
    PMH_CHK(pmh_malloc(ctx, sizeof(int) * gidx.size(), (void **)&C.d_gidx));
    PMH_CHK(pmh_malloc(ctx, sizeof(int) * reppos.size(), (void **)&C.d_reppos));
    PMH_CHK(pmh_malloc(ctx, use.size(), (void **)&C.d_use));
    PMH_CHK(pmh_memcpy_h2d(ctx, C.d_gidx, gidx.data(), sizeof(int) * gidx.size()));
    PMH_CHK(pmh_memcpy_h2d(ctx, C.d_reppos, reppos.data(), sizeof(int) * reppos.size()));
    PMH_CHK(pmh_memcpy_h2d(ctx, C.d_use, use.data(), use.size()));
  }
  if (S->Afund) (void)hipFree(S->Afund);
  {
    const size_t bytes = sizeof(double) * (size_t)std::max(32LL, atot) + 8192; // (k_fxo_gemm16 loads whole 4 KB pieces: up to one piece past the last chunk, never used)
    hipError_t   e     = hipMalloc((void **)&S->Afund, bytes);
    if (e != hipSuccess) return pmh_set_error(PMH_ERR_HIP, "PMH_FX_CLASS_ORBIT: %.2f GB for the representatives' rows: %s", bytes / 1e9, hipGetErrorString(e));
    PMH_HIP(hipMemsetAsync(S->Afund, 0, bytes, ctx->stream));
    S->afund_tot = atot;
  }
  // GEMM work items: (unit, column tile, split of the unit's chunks).  Every unit is split so that no workgroup has more than T chunks, T the smallest for which
  // the class's workgroups still fit ONE round of the 2 resident per CU (measured: 507 workgroups 0.275 ms, 513: 0.34); the k range of a rank (several GPUs) cuts the segments it crosses
  const int rank = S->stripe_size > 1 ? S->stripe_rank : 0, size = std::max(1, S->stripe_size);
  const int minch = getenv("PMH_FXO_MINCH") ? std::max(1, atoi(getenv("PMH_FXO_MINCH"))) : 8;
  // several classes on one row tile share ONE launch (fxo_gemm): the resident workgroups are divided among them
  int nplanned = 0, tm_first = 0, tn_first = 128, tnw_first = 0;
  bool one_tile = S->mfma16 && !getenv("PMH_FXO_NO_MERGE");
  for (int c = 0; c < S->ncls; c++)
    if (tab_of[c] >= 0) {
      if (!nplanned) tm_first = S->C[c].tm, tn_first = S->C[c].tn, tnw_first = S->C[c].tnw;
      else if (S->C[c].tm != tm_first || S->C[c].tn != tn_first || S->C[c].tnw != tnw_first) one_tile = false;
      nplanned++;
    }
  bool small_records = false; // a class with fewer than 8 slots per record: only the table-driven kernel knows the record size
  for (int c = 0; c < S->ncls; c++)
    if (tab_of[c] >= 0 && S->C[c].S != FXS_S) small_records = true;
  const bool merged = one_tile && (nplanned > 1 || small_records), tables = merged || small_records; // (classes on different row tiles: one table-driven launch per class)
  const int slots_all = getenv("PMH_FXO_SLOTS") ? std::max(1, atoi(getenv("PMH_FXO_SLOTS"))) : 2 * ctx->num_cus;
  const int slots = merged ? std::max(16, slots_all / nplanned) : slots_all;
  for (int c = 0; c < S->ncls; c++) {
    if (tab_of[c] < 0) continue;
    fxo_plan &P = plan[tab_of[c]];
    const fxs_class &C = S->C[c];
    const int klo = (int)((long long)C.nkc * rank / size), khi = (int)((long long)C.nkc * (rank + 1) / size); // this rank's chunks
    for (fxo_unit &U : P.units) {
      U.kc0 = std::max(klo, P.segc0[U.seg]), U.kc1 = std::min(khi, P.segc0[U.seg + 1]);
      if (U.kc1 <= U.kc0 || !U.listed) U.kc0 = U.kc1 = 0;
    }
  }
  const int fixedS = getenv("PMH_FXO_SPLIT") ? std::max(1, atoi(getenv("PMH_FXO_SPLIT"))) : 0;
  std::vector<int>       items, vnkc(S->ncls, 1), vldk(S->ncls, 16), vncol(S->ncls, 128);
  std::vector<long long> iteml;
  std::vector<int>       wgfirst; // per class: first item of every workgroup (relative to the class's first item) + the end
  long long              ctot = 0;
  S->flops = 0.0, S->flops_issued = 0.0, S->flops_dense = 0.0, S->bytes = 0.0, S->owned_bytes = 0.0;
  int Smax = 1;
  for (int c = 0; c < S->ncls; c++) {
    fxs_class &C = S->C[c];
    if (!C.nc) continue;
    fxo_plan &P = plan[tab_of[c]];
    const int klo = (int)((long long)C.nkc * rank / size), khi = (int)((long long)C.nkc * (rank + 1) / size), nk = khi - klo;
    vnkc[c] = C.nkc, vldk[c] = C.ldk, vncol[c] = C.nsymp * 8;
    C.coff = ctot;
    const int               ntile = C.Mp / C.tm, Mrows = C.m1 - C.m0;
    const std::vector<int> &ftab = P.fintab, &ctab = P.coltab;
    std::vector<long long>  unitbase(P.units.size(), 0);
    std::vector<int>        unittab(P.units.size() * 4, 0);
    double                  prod = 0.0, uprod = 0.0, ctiles = 0.0, ptiles = 0.0; // (valid rows) x (listed columns) of the tiles; x chunks of the units; padded tile x chunks; partial tiles
    for (int g = 0; g < C.ngroups; g++)
      for (int mt = 0; mt < ntile; mt++) {
        const int *ft = ftab.data() + ((size_t)g * (ntile + 1) + mt) * 4;
        int        listed = 0;
        for (int j = 0; j < ft[1]; j++) listed += ctab[(size_t)ft[0] + j] >= 0;
        prod += (double)std::max(0, std::min(Mrows, (mt + 1) * C.tm) - mt * C.tm) * listed;
      }
    C.item_first = (int)(items.size() / 8);
    // Pieces: the units with the same number of column tiles form one sequence of chunks (unit after unit), cut into equal pieces of at most T chunks -- T the smallest for
    // which the class's workgroups (one per piece and column tile) still fit ONE round of the 2 resident per CU (measured: 507 workgroups 0.275 ms, 513: 0.34).  A piece may end
    // one unit and begin the next (two items for its workgroups, two partial tiles): the kernel is bound by the latency of a workgroup's own chunk loop, so what counts is
    // the LONGEST workgroup, and unit-aligned splits (PMH_FXO_NO_STREAMK=1, or PMH_FXO_SPLIT) leave it at 48 chunks where the mean is 41.  Cuts closer than `snap`
    // chunks to a unit's end move there.
    struct part { int u, k0, k1, sp; };
    struct piece { int ntl; std::vector<part> parts; };
    std::vector<piece> pieces;
    int                Tbest = 1, wmax = 0;
    const bool         aligned = fixedS || getenv("PMH_FXO_NO_STREAMK");
    for (fxo_unit &U : P.units) U.S = 0;
    int ntlmax = 0;
    for (const fxo_unit &U : P.units) ntlmax = std::max(ntlmax, U.nct / C.tn);
    if (aligned) {
      auto wgs = [&](int T) {
        long long n = 0;
        for (const fxo_unit &U : P.units) {
          const int nku = U.kc1 - U.kc0;
          if (nku > 0) n += (long long)(U.nct / C.tn) * std::max(1, std::min((nku + T - 1) / T, std::max(1, nku / minch)));
        }
        return n;
      };
      int lo = 1, hi = 1;
      for (const fxo_unit &U : P.units) hi = std::max(hi, U.kc1 - U.kc0);
      while (lo < hi) { // wgs does not grow with T
        const int mid = (lo + hi) / 2;
        if (wgs(mid) <= slots) hi = mid;
        else lo = mid + 1;
      }
      Tbest = lo;
      for (size_t ui = 0; ui < P.units.size(); ui++) {
        fxo_unit &U  = P.units[ui];
        const int nku = U.kc1 - U.kc0;
        U.S          = nku > 0 ? std::max(1, std::min(fixedS ? fixedS : (nku + Tbest - 1) / Tbest, std::max(1, nku / minch))) : 0;
        for (int sp = 0; sp < U.S; sp++) pieces.push_back({U.nct / C.tn, {{(int)ui, U.kc0 + (int)((long long)nku * sp / U.S), U.kc0 + (int)((long long)nku * (sp + 1) / U.S), sp}}});
      }
    } else {
      std::vector<std::vector<int>> seq((size_t)ntlmax + 1); // units by column tile count, in unit order (group, row tile, segment)
      std::vector<long long>        N((size_t)ntlmax + 1, 0);
      for (size_t ui = 0; ui < P.units.size(); ui++)
        if (P.units[ui].kc1 > P.units[ui].kc0) seq[P.units[ui].nct / C.tn].push_back((int)ui), N[P.units[ui].nct / C.tn] += P.units[ui].kc1 - P.units[ui].kc0;
      auto wgs = [&](long long T) {
        long long n = 0;
        for (int k = 1; k <= ntlmax; k++) n += (long long)k * ((N[k] + T - 1) / T);
        return n;
      };
      auto search = [&](int nslots) {
        long long lo = 1, hi = 1;
        for (int k = 1; k <= ntlmax; k++) hi = std::max(hi, N[k]);
        while (lo < hi) {
          const long long mid = (lo + hi) / 2;
          if (wgs(mid) <= nslots) hi = mid;
          else lo = mid + 1;
        }
        return (int)lo;
      };
      Tbest = search(slots);
      // short pieces (a rank's 1/8 share of configs[2]: 5 chunks): the launch is prologue / epilogue / partial tiles rather than products, and one workgroup per CU with
      // pieces twice as long is faster (measured at the 1/8 share: 0.066 -> 0.062 ms per dense apply; 384 slots 0.070, 192: 0.075)
      if (!getenv("PMH_FXO_SLOTS") && Tbest < 12) Tbest = search(ctx->num_cus);
      const int snap = std::max(0, std::min(minch / 4, Tbest / 8));
      for (int k = ntlmax; k >= 1; k--) {
        if (!N[k]) continue;
        const long long W = (N[k] + Tbest - 1) / Tbest;
        std::vector<long long> ends; // prefix sums: the units' ends in the sequence
        long long              acc = 0;
        for (int ui : seq[k]) acc += P.units[ui].kc1 - P.units[ui].kc0, ends.push_back(acc);
        std::vector<long long> cut((size_t)W + 1, 0);
        for (long long i = 1; i < W; i++) {
          long long cpos = N[k] * i / W;
          auto      itb  = std::lower_bound(ends.begin(), ends.end(), cpos);
          if (itb != ends.end() && *itb - cpos <= snap) cpos = *itb;
          else if (itb != ends.begin() && cpos - *(itb - 1) <= snap) cpos = *(itb - 1);
          cut[i] = std::max(cut[i - 1], cpos);
        }
        cut[W] = N[k];
        size_t    iu = 0;
        long long ubeg = 0; // start of unit seq[k][iu] in the sequence
        for (long long i = 0; i < W; i++) {
          if (cut[i + 1] <= cut[i]) continue;
          piece pc{k, {}};
          long long pos = cut[i];
          while (pos < cut[i + 1]) {
            while (ends[iu] <= pos) ubeg = ends[iu], iu++;
            fxo_unit       &U   = P.units[seq[k][iu]];
            const long long upto = std::min(cut[i + 1], ends[iu]);
            pc.parts.push_back({seq[k][iu], U.kc0 + (int)(pos - ubeg), U.kc0 + (int)(upto - ubeg), U.S++});
            pos = upto;
          }
          pieces.push_back(pc);
        }
      }
    }
    // k_fxo_fin walks the units that have partial tiles on this rank only (a rank's share of the k range crosses one to three segments)
    std::vector<int> before(P.units.size() + 1, 0);
    unitbase.clear(), unittab.clear();
    for (size_t ui = 0; ui < P.units.size(); ui++) {
      fxo_unit &U  = P.units[ui];
      const int nku = U.kc1 - U.kc0;
      before[ui]   = (int)unitbase.size();
      U.cbase      = ctot;
      if (U.S > 0) unitbase.push_back(ctot), unittab.insert(unittab.end(), {U.lutoff, U.nct, U.S, 0});
      Smax = std::max(Smax, U.S);
      ctot += (long long)U.S * C.tm * U.nct;
      const double rows = (double)std::max(0, std::min(Mrows, (U.mt + 1) * C.tm) - U.mt * C.tm);
      uprod += rows * U.listed * nku * FXO_TK, ctiles += (double)C.tm * U.nct * nku * FXO_TK, ptiles += (double)U.S * C.tm * U.nct;
    }
    before[P.units.size()] = (int)unitbase.size();
    {
      std::vector<int> ft2 = P.fintab;
      for (size_t i = 3; i < ft2.size(); i += 4) ft2[i] = before[(size_t)ft2[i]];
      PMH_CHK(pmh_memcpy_h2d(ctx, C.d_fintab, ft2.data(), sizeof(int) * ft2.size()));
    }
    int nitem2 = 0; // workgroups with more than one item
    {
      // The column tiles of one piece read the SAME chunks of A at the same pace.  Workgroups b and b + 8 run on one XCD (one L2: MI355X_MICROARCH.md, workgroup
      // dispatch; scripts/micro/census.hip), so the workgroups go out 8 pieces at a time, column tile after column tile: the nt-th tile of a piece sits 8 nt workgroups after its
      // first one and finds the chunk in the XCD's L2 instead of fetching it again from beyond (with default-policy loads of A: scripts/micro/orbit_gemm.hip -DAPLAIN, OG_MAP=2:
      // -8 % per GEMM).  The pieces are grouped by their number of column tiles, so that the groups of 8 are uniform.  The partial sums stay indexed by (unit, split): the order of
      // the workgroups changes nothing in the result.  PMH_FXO_NO_XCDMAP=1: piece after piece, all column tiles each.
      static const bool xcdmap = !getenv("PMH_FXO_NO_XCDMAP");
      C.wgf_first = (int)wgfirst.size();
      auto emit = [&](const piece &pc, int nt) {
        wgfirst.push_back((int)(items.size() / 8) - C.item_first);
        int len = 0;
        for (const part &a : pc.parts) {
          const fxo_unit &U = P.units[a.u];
          items.insert(items.end(), {c, U.g, U.mt, nt, a.k0, a.k1, a.sp, U.nct});
          iteml.push_back(C.aoff);
          iteml.push_back(2 * (C.xoff + (long long)U.g * C.ld * C.S)); // in the signed multivector X2
          iteml.push_back(U.cbase + (long long)a.sp * C.tm * U.nct);
          iteml.push_back((long long)U.coff + (long long)nt * C.tn);
          len += a.k1 - a.k0;
        }
        wmax = std::max(wmax, len), nitem2 += pc.parts.size() > 1;
      };
      if (xcdmap) {
        std::stable_sort(pieces.begin(), pieces.end(), [](const piece &x, const piece &y) { return x.ntl > y.ntl; });
        for (size_t s0 = 0; s0 < pieces.size(); s0 += 8) {
          const size_t s1 = std::min(pieces.size(), s0 + 8);
          int          ntmax = 0;
          for (size_t i = s0; i < s1; i++) ntmax = std::max(ntmax, pieces[i].ntl);
          for (int nt = 0; nt < ntmax; nt++)
            for (size_t i = s0; i < s1; i++)
              if (nt < pieces[i].ntl) emit(pieces[i], nt);
        }
      } else {
        for (const piece &pc : pieces)
          for (int nt = 0; nt < pc.ntl; nt++) emit(pc, nt);
      }
      C.wg_count = (int)wgfirst.size() - C.wgf_first;
      wgfirst.push_back((int)(items.size() / 8) - C.item_first);
    }
    C.item_count = (int)(items.size() / 8) - C.item_first;
    if (getenv("PMH_FXO_VERBOSE")) {
      fprintf(stderr, "PMH_FX_CLASS_ORBIT class %d: %d representatives in %d row tiles of %d, %d k segments (chunks:", c, Mrows, ntile, C.tm, C.nseg);
      for (int sg = 0; sg < C.nseg; sg++) fprintf(stderr, " %d", P.segc0[sg + 1] - P.segc0[sg]);
      fprintf(stderr, "), %d workgroups (%d with two or more items) of at most %d chunks (limit %d of %d slots); padded columns per (group, row tile): unit by unit /", C.wg_count, nitem2, wmax, Tbest, slots);
      for (int g = 0; g < C.ngroups; g++)
        for (int mt = 0; mt < ntile; mt++) {
          for (const fxo_unit &U : P.units)
            if (U.g == g && U.mt == mt) fprintf(stderr, " %d", U.kc1 > U.kc0 ? U.nct : 0);
          fprintf(stderr, " of %d /", ftab[((size_t)g * (ntile + 1) + mt) * 4 + 1]);
        }
      fprintf(stderr, " (all: %d); listed x rows / all = %.3f, non-zero k of those = %.3f\n", C.nsym * 8, prod / std::max(1.0, (double)C.ngroups * Mrows * C.nsym * 8),
              uprod / std::max(1.0, prod * nk * FXO_TK));
    }
    if (C.d_finbase) pmh_free(ctx, C.d_finbase), pmh_free(ctx, C.d_unittab), C.d_finbase = nullptr;
    PMH_CHK(pmh_malloc(ctx, sizeof(long long) * std::max<size_t>(1, unitbase.size()), (void **)&C.d_finbase));
    PMH_CHK(pmh_malloc(ctx, sizeof(int) * std::max<size_t>(4, unittab.size()), (void **)&C.d_unittab));
    if (!unitbase.empty()) {
      PMH_CHK(pmh_memcpy_h2d(ctx, C.d_finbase, unitbase.data(), sizeof(long long) * unitbase.size()));
      PMH_CHK(pmh_memcpy_h2d(ctx, C.d_unittab, unittab.data(), sizeof(int) * unittab.size()));
    }
    const double M = Mrows, share = (double)nk / std::max(1, C.nkc);
    S->flops += 2.0 * C.nc * share * prod; // the products of the listed columns with the tiles' rows over the rank's k range (padding rows and columns not counted; structural zeros of B counted)
    S->flops_issued += 2.0 * ctiles, S->flops_dense += (double)C.ngroups * 2.0 * M * C.nc * share * 8.0 * C.nsym;
    S->owned_bytes += 8.0 * M * C.nc;
    // this rank's columns of A once + the gathered B + the split partial tiles written and read + Y
    S->bytes += (double)C.ngroups * (8.0 * M * C.nc * share + 8.0 * FXS_S * C.nc * share + 8.0 * FXS_S * C.nc) + 2.0 * 8.0 * ptiles;
  }
  S->fxo_S = Smax;
  S->nwg = 0;
  for (const fxs_class &C : S->C) S->nwg += C.wg_count;
  items.insert(items.end(), {0, 0, 0, 0, 0, 0, 0, 0});
  iteml.insert(iteml.end(), {0, 0, 0, 0});
  if (S->d_items) pmh_free(ctx, S->d_items);
  if (S->d_wgl) pmh_free(ctx, S->d_wgl);
  if (S->d_wg) pmh_free(ctx, S->d_wg);
  if (S->d_wgfirst) pmh_free(ctx, S->d_wgfirst);
  wgfirst.push_back(0);
  PMH_CHK(pmh_malloc(ctx, sizeof(int) * wgfirst.size(), (void **)&S->d_wgfirst));
  PMH_CHK(pmh_memcpy_h2d(ctx, S->d_wgfirst, wgfirst.data(), sizeof(int) * wgfirst.size()));
  PMH_CHK(pmh_malloc(ctx, sizeof(int) * items.size(), (void **)&S->d_items));
  PMH_CHK(pmh_memcpy_h2d(ctx, S->d_items, items.data(), sizeof(int) * items.size()));
  PMH_CHK(pmh_malloc(ctx, sizeof(long long) * iteml.size(), (void **)&S->d_wgl));
  PMH_CHK(pmh_memcpy_h2d(ctx, S->d_wgl, iteml.data(), sizeof(long long) * iteml.size()));
  // per class: nkc, ldk, ncol (ints) in d_wg
  std::vector<int> meta;
  meta.insert(meta.end(), vnkc.begin(), vnkc.end()), meta.insert(meta.end(), vldk.begin(), vldk.end()), meta.insert(meta.end(), vncol.begin(), vncol.end());
  PMH_CHK(pmh_malloc(ctx, sizeof(int) * meta.size(), (void **)&S->d_wg));
  PMH_CHK(pmh_memcpy_h2d(ctx, S->d_wg, meta.data(), sizeof(int) * meta.size()));
  if (ctot > S->cpart_cap) {
    if (S->cpart) pmh_free(ctx, S->cpart);
    PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)std::max(16LL, ctot), (void **)&S->cpart));
    S->cpart_cap = ctot;
  }
  // merged launch: workgroup -> items in the global item numbering, and the classes' own tables by class index
  if (S->d_wgfirst_all) pmh_free(ctx, S->d_wgfirst_all), S->d_wgfirst_all = nullptr;
  if (S->d_zrow_of) pmh_free(ctx, S->d_zrow_of), pmh_free(ctx, (void *)S->d_coltab_of), pmh_free(ctx, (void *)S->d_gidx_of), S->d_zrow_of = nullptr;
  S->nwg_all = 0, S->merged_tm = 0;
  if (tables) {
    std::vector<int>         wall, zr((size_t)S->ncls, 0), xs((size_t)S->ncls, 6);
    std::vector<const int *> ct((size_t)S->ncls, nullptr), gi((size_t)S->ncls, nullptr);
    for (int c = 0; c < S->ncls; c++) {
      const fxs_class &C = S->C[c];
      zr[c] = C.nsymp, ct[c] = C.d_coltab, gi[c] = C.d_gidx;
      for (xs[c] = 3; (1 << (xs[c] - 3)) < C.S; xs[c]++) {}
      for (int w = 0; w < C.wg_count; w++) wall.push_back(wgfirst[C.wgf_first + w] + C.item_first); // (a class's items are contiguous and the classes follow one another:
      S->nwg_all += C.wg_count;                                                                       //  a workgroup ends where the next one, of whichever class, begins)
    }
    int last_end = 0;
    for (int c = 0; c < S->ncls; c++)
      if (S->C[c].wg_count) last_end = wgfirst[S->C[c].wgf_first + S->C[c].wg_count] + S->C[c].item_first;
    wall.push_back(last_end);
    PMH_CHK(pmh_malloc(ctx, sizeof(int) * wall.size(), (void **)&S->d_wgfirst_all));
    PMH_CHK(pmh_memcpy_h2d(ctx, S->d_wgfirst_all, wall.data(), sizeof(int) * wall.size()));
    zr.insert(zr.end(), xs.begin(), xs.end()); // [zrow of the classes | record shifts of the classes]
    PMH_CHK(pmh_malloc(ctx, sizeof(int) * zr.size(), (void **)&S->d_zrow_of));
    PMH_CHK(pmh_memcpy_h2d(ctx, S->d_zrow_of, zr.data(), sizeof(int) * zr.size()));
    PMH_CHK(pmh_malloc(ctx, sizeof(const int *) * ct.size(), (void **)&S->d_coltab_of));
    PMH_CHK(pmh_memcpy_h2d(ctx, (void *)S->d_coltab_of, ct.data(), sizeof(const int *) * ct.size()));
    PMH_CHK(pmh_malloc(ctx, sizeof(const int *) * gi.size(), (void **)&S->d_gidx_of));
    PMH_CHK(pmh_memcpy_h2d(ctx, (void *)S->d_gidx_of, gi.data(), sizeof(const int *) * gi.size()));
    S->merged_tm = merged ? tm_first : 0, S->merged_tn = tn_first, S->merged_tnw = tnw_first;
  }
  if (S->d_fin_args) pmh_free(ctx, S->d_fin_args), S->d_fin_args = nullptr;
  S->fin_nbx = S->fin_ngroups = 0;
  if (S->ncls > 1 && !getenv("PMH_FXO_NO_MERGE")) { // the classes' finishing kernels in one launch
    std::vector<fxo_fin_args> fa((size_t)S->ncls);
    for (int c = 0; c < S->ncls; c++) {
      const fxs_class &C = S->C[c];
      fxo_fin_args     &a = fa[c];
      memset(&a, 0, sizeof(a));
      if (!C.nc || C.fin_elems <= 0) continue; // nbx = 0: the class's workgroups return at once
      a.ntile = C.Mp / C.tm, a.tm = C.tm, a.nsymp = C.nsymp, a.nc = C.nc, a.ld = C.ld, a.nslot = C.S, a.nbx = (C.fin_elems + PMH_BLOCK - 1) / PMH_BLOCK, a.ngroups = C.ngroups;
      a.fintab = C.d_fintab, a.unittab = C.d_unittab, a.lut = C.d_lut, a.coltab = C.d_coltab, a.reppos = C.d_reppos, a.posmap = C.d_posmap, a.unitbase = C.d_finbase, a.use = C.d_use, a.xbase0 = C.xoff;
      S->fin_nbx = std::max(S->fin_nbx, a.nbx), S->fin_ngroups = std::max(S->fin_ngroups, a.ngroups);
    }
    PMH_CHK(pmh_malloc(ctx, sizeof(fxo_fin_args) * fa.size(), &S->d_fin_args));
    PMH_CHK(pmh_memcpy_h2d(ctx, S->d_fin_args, fa.data(), sizeof(fxo_fin_args) * fa.size()));
  }
  S->fxo_ready = 1;
  return PMH_SUCCESS;
}

static int fxo_gemm(fx_shared *S)
{
  hipStream_t st = S->ctx->stream;
  const bool  merged = S->merged_tm > 0 && S->nwg_all > 0; // several classes on one row tile: one GEMM launch over all their items, then the classes' finishing launches
  if (merged) {
#define FXO_LAUNCH_ALL(NI, NWM, TNW)                                                                                                                                                                      \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fxo_gemm16<NI, NWM, true, TNW>), dim3(S->nwg_all), dim3(256), 0, st, (const int *)S->d_items, (const long long *)S->d_wgl, (const int *)S->d_wg,                 \
                     (const int *)(S->d_wg + S->ncls), (const int *)nullptr, 0, (const double *)S->Afund, (const int *)nullptr, (const double *)S->X2, S->cpart, (const int *)S->d_wgfirst_all, \
                     (const int *const *)S->d_coltab_of, (const int *)S->d_zrow_of, (const int *const *)S->d_gidx_of, (const int *)(S->d_zrow_of + S->ncls))
    switch (S->merged_tm + (S->merged_tn == 64 ? 1 : 0) + (S->merged_tnw == 48 ? 1 : 0)) {
    case 194: FXO_LAUNCH_ALL(3, 4, 48); break;
    case 130: FXO_LAUNCH_ALL(2, 4, 48); break;
    case 66: FXO_LAUNCH_ALL(1, 4, 48); break;
    case 144: FXO_LAUNCH_ALL(9, 1, 128); break;
    case 145: FXO_LAUNCH_ALL(9, 1, 64); break;
    case 128: FXO_LAUNCH_ALL(4, 2, 128); break;
    case 129: FXO_LAUNCH_ALL(4, 2, 64); break;
    case 112: FXO_LAUNCH_ALL(7, 1, 128); break;
    case 113: FXO_LAUNCH_ALL(7, 1, 64); break;
    case 96: FXO_LAUNCH_ALL(3, 2, 128); break;
    case 97: FXO_LAUNCH_ALL(3, 2, 64); break;
    case 80: FXO_LAUNCH_ALL(5, 1, 128); break;
    case 81: FXO_LAUNCH_ALL(5, 1, 64); break;
    default: return pmh_set_error(PMH_ERR_STATE, "PMH_FX_CLASS_ORBIT: row tile %d has no 16x16x4 kernel", S->merged_tm);
    }
#undef FXO_LAUNCH_ALL
    if (S->ev_mid_pending >= 0) PMH_HIP(hipEventRecord(S->ev_mid[S->ev_mid_pending], st));
  }
  for (int c = 0; c < S->ncls; c++) { // one launch per class (its own gather-index array and column lists); configs[2] / [3]: one class
    fxs_class &C = S->C[c];
    if (!C.nc) continue;
    const int first = C.item_first, count = C.wg_count; // the class's items are contiguous
    if (!count) continue;
#define FXO_LAUNCH(KERNEL)                                                                                                                                                                              \
  hipLaunchKernelGGL(KERNEL, dim3(count), dim3(256), 0, st, (const int *)(S->d_items + 8 * first), (const long long *)(S->d_wgl + 4 * first), (const int *)S->d_wg, (const int *)(S->d_wg + S->ncls), \
                     (const int *)C.d_coltab, C.nsymp, (const double *)S->Afund, (const int *)C.d_gidx, (const double *)S->X2, S->cpart, (const int *)(S->d_wgfirst + C.wgf_first))
#ifdef FXO_TRACE
    static unsigned long long *d_trace = nullptr;
    static int                 traced  = 0;
    if (!d_trace) {
      PMH_HIP(hipMalloc((void **)&d_trace, sizeof(unsigned long long) * 8 * 64 * 8));
      PMH_HIP(hipMemcpyToSymbol(HIP_SYMBOL(fxo_trace_buf), &d_trace, sizeof(d_trace)));
    }
    PMH_HIP(hipMemsetAsync(d_trace, 0, sizeof(unsigned long long) * 8 * 64 * 8, st));
#endif
    if (merged) {
      // (the class's products were part of the launch above)
    } else if (S->mfma16 && C.S != FXS_S) { // records of fewer than 8 slots: the table-driven kernel on this class's slice of the items
#define FXO_LAUNCH_T(NI, NWM, TNW)                                                                                                                                                                             \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fxo_gemm16<NI, NWM, true, TNW>), dim3(count), dim3(256), 0, st, (const int *)(S->d_items + 8 * first), (const long long *)(S->d_wgl + 4 * first), (const int *)S->d_wg, \
                     (const int *)(S->d_wg + S->ncls), (const int *)nullptr, 0, (const double *)S->Afund, (const int *)nullptr, (const double *)S->X2, S->cpart, (const int *)(S->d_wgfirst + C.wgf_first),     \
                     (const int *const *)S->d_coltab_of, (const int *)S->d_zrow_of, (const int *const *)S->d_gidx_of, (const int *)(S->d_zrow_of + S->ncls))
      switch (C.tm + (C.tn == 64 ? 1 : 0) + (C.tnw == 48 ? 1 : 0)) {
      case 194: FXO_LAUNCH_T(3, 4, 48); break;
      case 130: FXO_LAUNCH_T(2, 4, 48); break;
      case 66: FXO_LAUNCH_T(1, 4, 48); break;
      case 144: FXO_LAUNCH_T(9, 1, 128); break;
      case 145: FXO_LAUNCH_T(9, 1, 64); break;
      case 128: FXO_LAUNCH_T(4, 2, 128); break;
      case 129: FXO_LAUNCH_T(4, 2, 64); break;
      case 112: FXO_LAUNCH_T(7, 1, 128); break;
      case 113: FXO_LAUNCH_T(7, 1, 64); break;
      case 96: FXO_LAUNCH_T(3, 2, 128); break;
      case 97: FXO_LAUNCH_T(3, 2, 64); break;
      case 80: FXO_LAUNCH_T(5, 1, 128); break;
      case 81: FXO_LAUNCH_T(5, 1, 64); break;
      default: return pmh_set_error(PMH_ERR_STATE, "PMH_FX_CLASS_ORBIT: row tile %d has no 16x16x4 kernel", C.tm);
      }
#undef FXO_LAUNCH_T
    } else if (S->mfma16) {
      switch (C.tm) {
      case 144: FXO_LAUNCH((k_fxo_gemm16<9, 1>)); break;
      case 128: FXO_LAUNCH((k_fxo_gemm16<4, 2>)); break;
      case 112: FXO_LAUNCH((k_fxo_gemm16<7, 1>)); break;
      case 96: FXO_LAUNCH((k_fxo_gemm16<3, 2>)); break;
      case 80: FXO_LAUNCH((k_fxo_gemm16<5, 1>)); break;
      default: return pmh_set_error(PMH_ERR_STATE, "PMH_FX_CLASS_ORBIT: row tile %d has no 16x16x4 kernel", C.tm);
      }
    } else
    switch (C.tm) {
    case 128: FXO_LAUNCH(k_fxo_gemm); break;
    case 120: FXO_LAUNCH(k_fxo_gemm4<15>); break;
    case 112: FXO_LAUNCH(k_fxo_gemm4<14>); break;
    case 104: FXO_LAUNCH(k_fxo_gemm4<13>); break;
    case 96: FXO_LAUNCH(k_fxo_gemm4<12>); break;
    default: return pmh_set_error(PMH_ERR_STATE, "PMH_FX_CLASS_ORBIT: row tile %d has no kernel", C.tm);
    }
#undef FXO_LAUNCH
    if (!merged && S->ev_mid_pending >= 0 && c == S->ncls - 1) PMH_HIP(hipEventRecord(S->ev_mid[S->ev_mid_pending], st)); // (one class: configs[2] / [3]; several classes: after the last class's GEMM)
#ifdef FXO_TRACE
    if (++traced == 300) { // one launch in the steady state of the bench
      std::vector<unsigned long long> h(8 * 64 * 8);
      PMH_HIP(hipStreamSynchronize(st));
      PMH_HIP(hipMemcpy(h.data(), d_trace, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
      for (int w = 0; w < 8; w++) {
        fprintf(stderr, "FXO_TRACE workgroup %d (cycles of the shader clock; per chunk: loads issued | products | wait vmcnt | LDS store | barrier | total)\n", w * 37);
        for (int ch = 0; ch < 63; ch++) {
          const unsigned long long *q = &h[((size_t)w * 64 + ch) * 8];
          if (!q[5]) break;
#ifdef FXO_TRACE_FULL
          fprintf(stderr, "  chunk %2d: %6llu | %6llu | %6llu | %6llu | %6llu | %6llu\n", ch, q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], q[5] - q[4], q[5] - q[0]);
#else
          fprintf(stderr, "  chunk %2d: %6llu | %6llu | %6llu | %6llu | %6llu | %6llu\n", ch, 0ULL, q[2] - q[0], 0ULL, 0ULL, q[5] - q[2], q[5] - q[0]); // loads + products | store + barrier
#endif
        }
      }
    }
#endif
    if (C.fin_elems > 0 && !S->d_fin_args)
      hipLaunchKernelGGL(k_fxo_fin, dim3((unsigned)((C.fin_elems + PMH_BLOCK - 1) / PMH_BLOCK), C.ngroups), dim3(PMH_BLOCK), 0, st, C.Mp / C.tm, C.tm, C.nsymp, C.nc, (const int *)C.d_fintab,
                         (const int *)C.d_unittab, (const long long *)C.d_finbase, (const int *)C.d_lut, (const int *)C.d_coltab, (const double *)S->cpart, (const signed char *)C.d_use, (const int *)C.d_reppos, (const int *)C.d_posmap, C.xoff, C.ld,
                         S->Y, C.S);
  }
  if (S->d_fin_args && S->fin_nbx > 0) // after ALL classes' products
    hipLaunchKernelGGL(k_fxo_fin_all, dim3((unsigned)S->fin_nbx, (unsigned)S->fin_ngroups, (unsigned)S->ncls), dim3(PMH_BLOCK), 0, st, (const fxo_fin_args *)S->d_fin_args, (const double *)S->cpart, S->Y);
  PMH_HIP(hipGetLastError());
  return PMH_SUCCESS;
}

int fxs_assemble(fx_shared *S, pmh_matinv solver, int nslots, const int *slot_class, double rtol, int max_it, long long *n_solves)
{
  PMH_ARG(S && solver && nslots >= 1 && solver->nblocks == nslots && slot_class);
  pmh_ctx                       ctx = S->ctx;
  if (S->sym == 2) PMH_CHK(fxo_prepare(S));
  const std::vector<int>       &srs = solver->K->rowstart;
  std::vector<std::vector<int>> cslots(S->ncls), todo(S->ncls);
  std::vector<std::vector<int>> rep_of(S->ncls), op_of(S->ncls), check(S->ncls);
  std::vector<std::map<int, std::vector<int>>> members(S->ncls); // representative row -> the owned rows of its orbit
  for (int s = 0; s < nslots; s++)
    if (slot_class[s] >= 0 && slot_class[s] < S->ncls) cslots[slot_class[s]].push_back(s);
  int nbatch = 0;
  for (int c = 0; c < S->ncls; c++) {
    fxs_class &C = S->C[c];
    if (C.nc == 0) continue;
    if (cslots[c].empty()) return pmh_set_error(PMH_ERR_ARG, "pmh_fexplicit_assemble: no solver slot for block class %d", c);
    for (int s : cslots[c])
      if (srs[s + 1] - srs[s] != C.nloc) return pmh_set_error(PMH_ERR_ARG, "pmh_fexplicit_assemble: slot %d has %d rows, class %d blocks have %d", s, srs[s + 1] - srs[s], c, C.nloc);
    if (S->sym == 2) {
      // orbit storage: the owned representatives; self-check: rows of their orbits reached by a non-trivial operation
      for (int pl = 0; pl < C.m1 - C.m0; pl++) todo[c].push_back(C.reps[C.m0 + pl]);
      std::vector<int> filled;
      for (int r = 0; r < C.nc; r++)
        if (C.op_of[r] != 0) {
          const int k = (int)(std::lower_bound(C.reps.begin(), C.reps.end(), C.rep_of[r]) - C.reps.begin());
          if (k >= C.m0 && k < C.m1) filled.push_back(r);
        }
      const int nchk = (int)std::min(filled.size(), cslots[c].size());
      for (int i = 0; i < nchk; i++) check[c].push_back(filled[(size_t)((long long)filled.size() * (2 * i + 1) / (2 * nchk))]);
    } else if (S->sym && C.nsym > 1) {
      // orbits of the rows under the class's symmetries: rep_of / op_of, one solve per orbit that has a row in this rank's super bands
      rep_of[c].assign((size_t)C.nc, -1), op_of[c].assign((size_t)C.nc, 0);
      for (int p = 0; p < C.nc; p++) {
        if (rep_of[c][p] >= 0) continue;
        for (int g = 0; g < C.nsym; g++) {
          const int r = C.h_posmap[(size_t)g * C.nc + p];
          if (rep_of[c][r] < 0) rep_of[c][r] = p, op_of[c][r] = g;
        }
      }
      std::vector<char> need((size_t)C.nc, 0);
      for (int r = 0; r < C.nc; r++)
        if (C.own[r / FXM_RS]) need[rep_of[c][r]] = 1, members[c][rep_of[c][r]].push_back(r);
      for (int p = 0; p < C.nc; p++)
        if (need[p]) todo[c].push_back(p);
      // self-check: up to one batch of symmetry-filled owned rows, spread over the range, solved directly at the end
      std::vector<int> filled;
      for (int r = 0; r < C.nc; r++)
        if (C.own[r / FXM_RS] && op_of[c][r] != 0) filled.push_back(r);
      const int nchk = (int)std::min(filled.size(), cslots[c].size());
      for (int i = 0; i < nchk; i++) check[c].push_back(filled[(size_t)((long long)filled.size() * (2 * i + 1) / (2 * nchk))]);
    } else if (S->sym) {
      for (int p = 0; p < C.nc; p++)
        if (C.own[p / FXM_RS]) todo[c].push_back(p); // the rows of this rank's super bands
    } else
      for (int p = C.r0; p < std::min(C.r1, C.nc); p++) todo[c].push_back(p); // the rows of this rank's stripe
    nbatch = std::max(nbatch, (int)((todo[c].size() + cslots[c].size() - 1) / cslots[c].size()));
  }
  double      *rhs, *sol;
  int         *d_idx, *h_idx;
  const size_t nsol = (size_t)std::max(1, solver->n);
  PMH_CHK(pmh_malloc(ctx, sizeof(double) * nsol, (void **)&rhs));
  PMH_CHK(pmh_malloc(ctx, sizeof(double) * nsol, (void **)&sol));
  PMH_CHK(pmh_malloc(ctx, sizeof(int) * nslots, (void **)&d_idx));
  PMH_HIP(hipHostMalloc((void **)&h_idx, sizeof(int) * nslots * 2, hipHostMallocDefault));
  PMH_CHK(pmh_memset(ctx, rhs, 0, sizeof(double) * nsol));
  double old_rtol, old_atol;
  int    old_maxit;
  PMH_CHK(pmh_matinv_get_tolerances(solver, &old_rtol, &old_atol, &old_maxit));
  PMH_CHK(pmh_matinv_set_tolerances(solver, rtol, 1e-300, max_it > 0 ? max_it : old_maxit));
  int              rc = PMH_SUCCESS;
  std::vector<int> prow(nslots);
  for (int k = 0; k < nbatch && !rc; k++) {
    int *hh = h_idx + (k & 1) * nslots;
    for (int s = 0; s < nslots; s++) hh[s] = -1, prow[s] = -1;
    for (int c = 0; c < S->ncls; c++)
      for (size_t t = 0; t < cslots[c].size(); t++) {
        const size_t j = (size_t)k * cslots[c].size() + t;
        if (j < todo[c].size()) {
          const int s = cslots[c][t];
          prow[s]     = todo[c][j];
          hh[s]       = srs[s] + S->C[c].urel[prow[s]];
        }
      }
    if (hipMemcpyAsync(d_idx, hh, sizeof(int) * nslots, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
      rc = pmh_set_error(PMH_ERR_HIP, "pmh_fexplicit_assemble: index upload failed");
      break;
    }
    hipLaunchKernelGGL(k_fxs_set_entries, dim3((nslots + 63) / 64), dim3(64), 0, ctx->stream, nslots, (const int *)d_idx, 1.0, rhs);
    if ((rc = pmh_matinv_mult(solver, rhs, sol))) break;
    if (solver->last_max_its >= solver->max_it) {
      rc = pmh_set_error(PMH_ERR_STATE, "pmh_fexplicit_assemble: a set-up solve of batch %d did not reach rtol %.1e within %d iterations of the inner KSP", k, rtol, solver->max_it);
      break;
    }
    hipLaunchKernelGGL(k_fxs_set_entries, dim3((nslots + 63) / 64), dim3(64), 0, ctx->stream, nslots, (const int *)d_idx, 0.0, rhs);
    for (int s = 0; s < nslots; s++) {
      if (prow[s] < 0) continue;
      (*n_solves)++;
      const fxs_class &C = S->C[slot_class[s]];
      if (S->sym == 2) {
        const int pl = (int)(std::lower_bound(C.reps.begin(), C.reps.end(), prow[s]) - C.reps.begin()) - C.m0;
        hipLaunchKernelGGL(k_fxo_store_row, dim3(std::max(1, std::min(64, (C.nc + PMH_BLOCK - 1) / PMH_BLOCK))), dim3(PMH_BLOCK), 0, ctx->stream, C.reprow[pl], C.tm, C.nc, C.nkc, (const int *)C.d_urel,
                           (const int *)C.d_kinv, (const double *)(sol + srs[s]), S->Afund + C.aoff);
      } else if (S->sym && C.nsym > 1) {
        const int p = prow[s];
        for (int r : members[slot_class[s]][p]) {
          const int g = op_of[slot_class[s]][r], sb = r / FXM_RS;
          hipLaunchKernelGGL(k_fxs_extract_symg, dim3(std::max(1, std::min(64, (C.nc + PMH_BLOCK - 1) / PMH_BLOCK))), dim3(PMH_BLOCK), 0, ctx->stream, r, C.nc, (double)C.h_sign[(size_t)g * C.nc + p],
                             (const int *)C.d_urel, (const double *)(sol + srs[s]), (const int *)(C.d_posmap + (size_t)g * C.nc), (const signed char *)(C.d_sign + (size_t)g * C.nc),
                             S->Wbase + C.woff + (long long)FXM_RS * FXM_RS * ((long long)sb * (sb + 1) / 2));
        }
      } else if (S->sym) {
        const int sb = prow[s] / FXM_RS;
        hipLaunchKernelGGL(k_fxs_extract_sym, dim3(std::max(1, std::min(64, (prow[s] + PMH_BLOCK) / PMH_BLOCK))), dim3(PMH_BLOCK), 0, ctx->stream, prow[s], (const int *)C.d_urel, (const double *)(sol + srs[s]),
                           S->Wbase + C.woff + (long long)FXM_RS * FXM_RS * ((long long)sb * (sb + 1) / 2));
      } else
      hipLaunchKernelGGL(k_fxs_extract, dim3(std::max(1, std::min(64, (C.nc + PMH_BLOCK - 1) / PMH_BLOCK))), dim3(PMH_BLOCK), 0, ctx->stream, C.nc, (const int *)C.d_urel, (const double *)(sol + srs[s]),
                         S->Wbase + C.woff + (long long)prow[s] * C.ld); // row p of W_c = column p (K^+ symmetric)
    }
    if (hipGetLastError() != hipSuccess) rc = pmh_set_error(PMH_ERR_HIP, "pmh_fexplicit_assemble: launch failed in batch %d", k);
  }
  // self-check of the set-up by symmetry: a few symmetry-filled rows against their direct solves
  bool any_check = false;
  for (int c = 0; c < S->ncls; c++) any_check = any_check || !check[c].empty();
  if (!rc && any_check) rc = pmh_sync(ctx); // the pinned index buffer is reused below
  if (!rc && any_check) {
    int *hh = h_idx;
    for (int s = 0; s < nslots; s++) hh[s] = -1, prow[s] = -1;
    for (int c = 0; c < S->ncls; c++)
      for (size_t t = 0; t < check[c].size(); t++) {
        const int s = cslots[c][t];
        prow[s] = check[c][t], hh[s] = srs[s] + S->C[c].urel[prow[s]];
      }
    double *d_out = nullptr, h_out[2 * 64];
    rc = pmh_malloc(ctx, sizeof(double) * 2 * 64, (void **)&d_out);
    if (!rc && hipMemcpyAsync(d_idx, hh, sizeof(int) * nslots, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = pmh_set_error(PMH_ERR_HIP, "pmh_fexplicit_assemble: index upload failed");
    if (!rc) {
      hipLaunchKernelGGL(k_fxs_set_entries, dim3((nslots + 63) / 64), dim3(64), 0, ctx->stream, nslots, (const int *)d_idx, 1.0, rhs);
      rc = pmh_matinv_mult(solver, rhs, sol);
    }
    for (int s = 0; s < nslots && !rc; s++) {
      if (prow[s] < 0) continue;
      (*n_solves)++;
      const fxs_class &C  = S->C[slot_class[s]];
      const int        r = prow[s], sb = r / FXM_RS;
      int              nb = std::max(1, std::min(64, (r + PMH_BLOCK) / PMH_BLOCK));
      if (S->sym == 2) {
        const int g = C.op_of[r], p = C.rep_of[r], pl = (int)(std::lower_bound(C.reps.begin(), C.reps.end(), p) - C.reps.begin()) - C.m0;
        nb          = std::max(1, std::min(64, (C.nc + PMH_BLOCK - 1) / PMH_BLOCK));
        hipLaunchKernelGGL(k_fxo_check_row, dim3(nb), dim3(PMH_BLOCK), 0, ctx->stream, C.reprow[pl], C.tm, C.nc, C.nkc, (double)C.h_sign[(size_t)g * C.nc + p], (const int *)C.d_urel, (const int *)C.d_kinv, (const double *)(sol + srs[s]),
                           (const int *)(C.d_posmap + (size_t)g * C.nc), (const signed char *)(C.d_sign + (size_t)g * C.nc), (const double *)(S->Afund + C.aoff), d_out);
      } else
      hipLaunchKernelGGL(k_fxs_check_row, dim3(nb), dim3(PMH_BLOCK), 0, ctx->stream, r, (const int *)C.d_urel, (const double *)(sol + srs[s]),
                         (const double *)(S->Wbase + C.woff + (long long)FXM_RS * FXM_RS * ((long long)sb * (sb + 1) / 2)), d_out);
      if ((rc = pmh_memcpy_d2h(ctx, h_out, d_out, sizeof(double) * 2 * nb))) break;
      double d = 0.0, m = 0.0;
      for (int i = 0; i < nb; i++) d = std::max(d, h_out[2 * i]), m = std::max(m, h_out[2 * i + 1]);
      if (!(d <= 1e-8 * m))
        rc = pmh_set_error(PMH_ERR_STATE, "pmh_fexplicit_assemble: row %d of class %d filled by symmetry differs from its direct solve by %.2e (relative to the row's largest entry): the operations handed to pmh_fexplicit_set_class_symmetry are not symmetries of K^+", r, slot_class[s], d / m);
    }
    if (d_out) pmh_free(ctx, d_out);
  }
  if (!rc) rc = pmh_sync(ctx);
  pmh_matinv_set_tolerances(solver, old_rtol, old_atol, old_maxit);
  pmh_free(ctx, rhs), pmh_free(ctx, sol), pmh_free(ctx, d_idx);
  (void)hipHostFree(h_idx);
  return rc;
}

static int fxs_gemm(fx_shared *S)
{
  if (S->sym == 2 && !S->fxo_ready) return pmh_set_error(PMH_ERR_STATE, "PMH_FX_CLASS_ORBIT: apply before the assembly");
  if (!S->nwg) return PMH_SUCCESS;
  hipStream_t st    = S->ctx->stream;
  const bool  timed = S->ev_on && (size_t)(2 * S->ev_used + 2) <= S->ev.size();
  if (timed) PMH_HIP(hipEventRecord(S->ev[2 * S->ev_used], st));
  const long long stride = std::max(16LL, S->nX);
  if (S->sym == 2) {
    S->ev_mid_pending = timed ? S->ev_used : -1;
    PMH_CHK(fxo_gemm(S));
    S->ev_mid_pending = -1;
  } else if (S->sym) {
    hipLaunchKernelGGL(k_fxs_symm8, dim3(S->nwg), dim3(FXM_THREADS), 0, st, (const int *)S->d_wg, (const int *)S->d_items, (const long long *)S->d_wgl, (const int *)S->d_ld, (const long long *)S->d_xoff,
                       (const double *)S->Wbase, (const double *)S->X, S->part, stride, S->pt);
    for (auto &C : S->C)
      hipLaunchKernelGGL(k_fxs_symfin, dim3((unsigned)(((long long)C.ld * FXS_S / 2 * 8 + PMH_BLOCK - 1) / PMH_BLOCK), C.ngroups), dim3(PMH_BLOCK), 0, st, C.ld, C.nmb, C.nown, (const int *)C.d_nseg, (const int *)C.d_ownfirst,
                         (const long long *)C.d_ptoff, C.xoff, stride, (const double *)S->part, (const double *)S->pt, S->Y);
  } else {
  hipLaunchKernelGGL(k_fxs_gemm8, dim3(S->nwg), dim3(PMH_BLOCK), 0, st, (const int *)S->d_wg, (const int *)S->d_ld, (const long long *)S->d_woff, (const long long *)S->d_xoff, (const double *)S->Wbase,
                     (const double *)S->X, S->part, stride);
  hipLaunchKernelGGL(k_fxs_fin, dim3((unsigned)((S->nX / 2 + PMH_BLOCK - 1) / PMH_BLOCK)), dim3(PMH_BLOCK), 0, st, S->nX, S->nseg, stride, (const double *)S->part, S->Y);
  }
  if (timed) {
    PMH_HIP(hipEventRecord(S->ev[2 * S->ev_used + 1], st));
    S->ev_used++;
  }
  PMH_HIP(hipGetLastError());
  return PMH_SUCCESS;
}

// X (position, slot) -> X2 (position, +-, slot): only for the dense kernel alone on a multivector handed in (pmh_fexplicit_dense_mult); the operator fills X2 by its own gluing
__global__ __launch_bounds__(PMH_BLOCK) void k_fxo_signed_copy(long long n, int nslot, const double *__restrict__ X, double *__restrict__ X2)
{
  for (long long i = (long long)blockIdx.x * PMH_BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * PMH_BLOCK) {
    const long long p = i / nslot, sl = i % nslot;
    const double    v = X[i];
    X2[2 * p * nslot + sl] = v, X2[(2 * p + 1) * nslot + sl] = -v;
  }
}

int fxs_apply(fx_shared *S, const double *lambda, double *y)
{
  if (S->sym == 2) PMH_CHK(pmh_gluing_mult(S->Bc2, lambda, S->X2));
  else PMH_CHK(pmh_gluing_mult(S->Bc, lambda, S->X));
  PMH_CHK(fxs_gemm(S));
  return pmh_gluing_mult_transpose(S->Bc, S->Y, y); // ends with the all-reduce on several GPUs
}

int fxs_stages(fx_shared *S, pmh_csr *gather, double **mid_in, pmh_csr *scatter, const double **mid_out)
{
  if (S->sym == 2) *gather = S->Bc2->Bt, *mid_in = S->X2;
  else *gather = S->Bc->Bt, *mid_in = S->X;
  *scatter = S->Bc->B, *mid_out = S->Y;
  return PMH_SUCCESS;
}
int fxs_mid(fx_shared *S) { return fxs_gemm(S); }

// the dense kernel alone (tests, tuning): Y = blockdiag(W_c) X on the multivectors as they stand
int fxs_dense(fx_shared *S)
{
  if (S->sym == 2 && S->nX > 0) {
    for (const fxs_class &C : S->C) { // class by class: the records of a class have C.S slots
      const long long n = (long long)C.ngroups * C.ld * C.S;
      if (!n) continue;
      hipLaunchKernelGGL(k_fxo_signed_copy, dim3((unsigned)std::min<long long>(4096, (n + PMH_BLOCK - 1) / PMH_BLOCK)), dim3(PMH_BLOCK), 0, S->ctx->stream, n, C.S, (const double *)(S->X + C.xoff), S->X2 + 2 * C.xoff);
    }
    PMH_HIP(hipGetLastError());
  }
  return fxs_gemm(S);
}
long long fxs_multivector_length(fx_shared *S) { return S->nX; }
double   *fxs_X(fx_shared *S) { return S->X; }
double   *fxs_Y(fx_shared *S) { return S->Y; }

int fxs_fill_pattern(fx_shared *S, int byte)
{
  if (S->sym == 2) {
    PMH_CHK(fxo_prepare(S));
    PMH_HIP(hipMemsetAsync(S->Afund, byte, sizeof(double) * (size_t)S->afund_tot, S->ctx->stream));
    return pmh_sync(S->ctx);
  }
  PMH_HIP(hipMemsetAsync(S->Wbase, byte, sizeof(double) * (size_t)S->wtot, S->ctx->stream));
  return pmh_sync(S->ctx);
}

// W_b = W_c[pos_b, pos_b] on the host (tests); gamma: the block's touched dofs (rank-local primal indices, ascending)
int fxs_get_block(fx_shared *S, int b, int n, const int *gamma, double *out_host)
{
  const fxs_class    &C = S->C[S->cls[b]];
  if (S->sym == 2) {
    // orbit storage (tests: small classes, all representatives on this rank): row r = g p rebuilt from its representative's row
    if (!S->fxo_ready || C.m0 != 0 || C.m1 != C.M_all) return pmh_set_error(PMH_ERR_STATE, "pmh_fexplicit_get_block: the orbit storage holds only this rank's representatives");
    std::vector<double> T((size_t)C.Mp * C.ldk), row((size_t)C.nc);
    PMH_HIP(hipMemcpy(T.data(), S->Afund + C.aoff, sizeof(double) * T.size(), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) {
      const int r = C.pos[gamma[i] - S->K->rowstart[b]], p = C.rep_of[r], g = C.op_of[r], pl = (int)(std::lower_bound(C.reps.begin(), C.reps.end(), p) - C.reps.begin());
      const double  sp = (double)C.h_sign[(size_t)g * C.nc + p];
      const int     pr = C.reprow[pl]; // the representative's row of A
      const double *ab = T.data() + (size_t)(pr / C.tm) * C.nkc * (FXO_TK * C.tm) + pr % C.tm;
      for (int cc = 0; cc < C.nc; cc++) {
        const int k = C.kinv[cc];
        row[C.h_posmap[(size_t)g * C.nc + cc]] = sp * (double)C.h_sign[(size_t)g * C.nc + cc] * ab[(size_t)(k / FXO_TK) * (FXO_TK * C.tm) + (k % FXO_TK) * C.tm];
      }
      for (int k = 0; k < n; k++) out_host[(size_t)i * n + k] = row[C.pos[gamma[k] - S->K->rowstart[b]]];
    }
    return PMH_SUCCESS;
  }
  if (S->sym) {
    // the class's tiles on the host (tests: small classes), W[p][c] from the stored lower triangle
    const long long     len = (long long)FXM_RS * FXM_RS * ((long long)C.nsb * (C.nsb + 1) / 2);
    std::vector<double> T((size_t)len);
    PMH_HIP(hipMemcpy(T.data(), S->Wbase + C.woff, sizeof(double) * (size_t)len, hipMemcpyDeviceToHost));
    auto at = [&](int p, int c) {
      const int hi = std::max(p, c), lo = std::min(p, c), sb = hi / FXM_RS, Il = (hi % FXM_RS) / 16, r = hi & 15, q = r >> 2, l = 16 * (r & 3) + (lo & 15);
      const double v = T[(size_t)((long long)FXM_RS * FXM_RS * ((long long)sb * (sb + 1) / 2) + ((long long)(lo >> 4) * FXM_RT + Il) * 256 + (q >> 1) * 128 + 2 * l + (q & 1))];
      return hi == lo ? 2.0 * v : v;
    };
    for (int i = 0; i < n; i++)
      for (int k = 0; k < n; k++) out_host[(size_t)i * n + k] = at(C.pos[gamma[i] - S->K->rowstart[b]], C.pos[gamma[k] - S->K->rowstart[b]]);
    return PMH_SUCCESS;
  }
  std::vector<double> row((size_t)std::max(1, C.ld));
  for (int i = 0; i < n; i++) {
    const int p = C.pos[gamma[i] - S->K->rowstart[b]];
    PMH_HIP(hipMemcpy(row.data(), S->Wbase + C.woff + (long long)p * C.ld, sizeof(double) * (size_t)C.ld, hipMemcpyDeviceToHost));
    for (int k = 0; k < n; k++) out_host[(size_t)i * n + k] = row[C.pos[gamma[k] - S->K->rowstart[b]]];
  }
  return PMH_SUCCESS;
}

int fxs_timing_enable(fx_shared *S, int max_launches)
{
  while ((int)S->ev.size() < 2 * max_launches) {
    hipEvent_t e;
    PMH_HIP(hipEventCreate(&e));
    S->ev.push_back(e);
  }
  while ((int)S->ev_mid.size() < max_launches) {
    hipEvent_t e;
    PMH_HIP(hipEventCreate(&e));
    S->ev_mid.push_back(e);
  }
  S->ev_on = max_launches > 0, S->ev_used = 0;
  return PMH_SUCCESS;
}

int fxs_timing_get(fx_shared *S, int *launches, double *total_ms, double *first_kernel_ms)
{
  PMH_CHK(pmh_sync(S->ctx));
  double tot = 0.0, first = 0.0;
  for (int i = 0; i < S->ev_used; i++) {
    float ms = 0.f;
    PMH_HIP(hipEventElapsedTime(&ms, S->ev[2 * i], S->ev[2 * i + 1]));
    tot += ms;
    if (S->sym == 2) {
      PMH_HIP(hipEventElapsedTime(&ms, S->ev[2 * i], S->ev_mid[i]));
      first += ms;
    }
  }
  *launches = S->ev_used, *total_ms = tot;
  if (first_kernel_ms) *first_kernel_ms = S->sym == 2 ? first : tot; // orbit storage: the GEMM kernel alone (the rest is k_fxo_fin)
  return PMH_SUCCESS;
}
