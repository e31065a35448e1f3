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
#include <algorithm>
#include <chrono>
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
};

struct fx_shared {
  pmh_ctx                ctx;
  pmh_gluing             B;
  pmh_blockdiag          K;
  int                    nb, ncls;
  std::vector<int>       cls; // class of every block
  std::vector<fxs_class> C;
  pmh_gluing             Bc = nullptr;
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
  for (int j = 1; j < nseg; j++) s += *(const dbl2 *)(part + (long long)j * part_stride + i);
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

static int fxs_build_launch(fx_shared *S)
{
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

int fxs_create(pmh_gluing B, pmh_blockdiag K, const int *block_class, fx_shared **out)
{
  PMH_ARG(B && K && block_class && out && B->n_x == K->n);
  pmh_ctx    ctx = B->ctx;
  fx_shared *S   = new fx_shared();
  S->ctx = ctx, S->B = B, S->K = K, S->nb = K->nblocks;
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
  // union of the touched dofs per class
  for (int c = 0; c < S->ncls; c++) S->C[c].pos.assign((size_t)std::max(1, S->C[c].nloc), -1);
  auto block_of = [&](int i) { return (int)(std::upper_bound(K->rowstart.begin(), K->rowstart.end(), i) - K->rowstart.begin()) - 1; };
  std::vector<int> lb((size_t)std::max(1, B->n_leaves));
  for (int i = 0; i < B->n_leaves; i++) {
    lb[i] = block_of(B->h_row[i]);
    S->C[S->cls[lb[i]]].pos[B->h_row[i] - K->rowstart[lb[i]]] = 0;
  }
  long long wtot = 0, xtot = 0;
  for (int c = 0; c < S->ncls; c++) {
    fxs_class &C = S->C[c];
    for (int i = 0; i < C.nloc; i++)
      if (C.pos[i] == 0) C.pos[i] = (int)C.urel.size(), C.urel.push_back(i);
    C.nc      = (int)C.urel.size();
    C.ld      = (C.nc + FXS_PAD - 1) / FXS_PAD * FXS_PAD;
    C.ngroups = ((int)C.blocks.size() + FXS_S - 1) / FXS_S;
    C.r0 = 0, C.r1 = C.ld;
    C.woff = wtot, C.xoff = xtot;
    wtot += (long long)C.ld * C.ld;
    xtot += (long long)C.ngroups * C.ld * FXS_S;
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
    rows[i]            = (int)(C.xoff + (long long)group[b] * C.ld * FXS_S + (long long)C.pos[B->h_row[i] - K->rowstart[b]] * FXS_S + slot[b]);
  }
  PMH_CHK(pmh_gluing_create(ctx, (int)std::max(1LL, xtot), B->n_lambda, B->n_leaves, rows.data(), B->h_root.data(), B->h_sign.data(), &S->Bc));
  {
    const size_t bytes = sizeof(double) * (size_t)std::max(32LL, wtot);
    hipError_t   e     = hipMalloc((void **)&S->Wbase, bytes);
    if (e != hipSuccess) return pmh_set_error(PMH_ERR_HIP, "pmh_fexplicit_create_shared: %.2f GB for the shared explicit operators: %s", bytes / 1e9, hipGetErrorString(e));
    PMH_HIP(hipMemsetAsync(S->Wbase, 0, bytes, ctx->stream));
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
  PMH_CHK(fxs_build_launch(S));
  *out = S;
  return PMH_SUCCESS;
}

void fxs_destroy(fx_shared *S)
{
  if (!S) return;
  pmh_ctx ctx = S->ctx;
  for (auto &C : S->C)
    if (C.d_urel) pmh_free(ctx, C.d_urel);
  pmh_gluing_destroy(S->Bc);
  if (S->Wbase) (void)hipFree(S->Wbase);
  if (S->part) pmh_free(ctx, S->part);
  pmh_free(ctx, S->X), pmh_free(ctx, S->Y), pmh_free(ctx, S->d_wg), pmh_free(ctx, S->d_ld), pmh_free(ctx, S->d_woff), pmh_free(ctx, S->d_xoff);
  for (hipEvent_t e : S->ev) (void)hipEventDestroy(e);
  delete S;
}

// several GPUs: rank r applies / assembles the rows [r0, r1) of every W_c, contiguous ranges of equal length (multiples of 32)
int fxs_set_stripe(fx_shared *S, int rank, int size)
{
  for (auto &C : S->C) {
    const int nrg = C.ld / 32; // row groups of 32
    C.r0 = (int)((long long)nrg * rank / size) * 32;
    C.r1 = (int)((long long)nrg * (rank + 1) / size) * 32;
  }
  return fxs_build_launch(S);
}

long long fxs_dense_bytes(fx_shared *S) { return (long long)sizeof(double) * S->wtot; }
double    fxs_apply_bytes(fx_shared *S) { return S->bytes; }

int fxs_assemble(fx_shared *S, pmh_matinv solver, int nslots, const int *slot_class, double rtol, int max_it, long long *n_solves)
{
  PMH_ARG(S && solver && nslots >= 1 && solver->nblocks == nslots && slot_class);
  pmh_ctx                       ctx = S->ctx;
  const std::vector<int>       &srs = solver->K->rowstart;
  std::vector<std::vector<int>> cslots(S->ncls), todo(S->ncls);
  for (int s = 0; s < nslots; s++)
    if (slot_class[s] >= 0 && slot_class[s] < S->ncls) cslots[slot_class[s]].push_back(s);
  int nbatch = 0;
  for (int c = 0; c < S->ncls; c++) {
    fxs_class &C = S->C[c];
    if (C.nc == 0) continue;
    if (cslots[c].empty()) return pmh_set_error(PMH_ERR_ARG, "pmh_fexplicit_assemble: no solver slot for block class %d", c);
    for (int s : cslots[c])
      if (srs[s + 1] - srs[s] != C.nloc) return pmh_set_error(PMH_ERR_ARG, "pmh_fexplicit_assemble: slot %d has %d rows, class %d blocks have %d", s, srs[s + 1] - srs[s], c, C.nloc);
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
      hipLaunchKernelGGL(k_fxs_extract, dim3(std::max(1, std::min(64, (C.nc + PMH_BLOCK - 1) / PMH_BLOCK))), dim3(PMH_BLOCK), 0, ctx->stream, C.nc, (const int *)C.d_urel, (const double *)(sol + srs[s]),
                         S->Wbase + C.woff + (long long)prow[s] * C.ld); // row p of W_c = column p (K^+ symmetric)
    }
    if (hipGetLastError() != hipSuccess) rc = pmh_set_error(PMH_ERR_HIP, "pmh_fexplicit_assemble: launch failed in batch %d", k);
  }
  if (!rc) rc = pmh_sync(ctx);
  pmh_matinv_set_tolerances(solver, old_rtol, old_atol, old_maxit);
  pmh_free(ctx, rhs), pmh_free(ctx, sol), pmh_free(ctx, d_idx);
  (void)hipHostFree(h_idx);
  return rc;
}

static int fxs_gemm(fx_shared *S)
{
  if (!S->nwg) return PMH_SUCCESS;
  hipStream_t st    = S->ctx->stream;
  const bool  timed = S->ev_on && (size_t)(2 * S->ev_used + 2) <= S->ev.size();
  if (timed) PMH_HIP(hipEventRecord(S->ev[2 * S->ev_used], st));
  const long long stride = std::max(16LL, S->nX);
  hipLaunchKernelGGL(k_fxs_gemm8, dim3(S->nwg), dim3(PMH_BLOCK), 0, st, (const int *)S->d_wg, (const int *)S->d_ld, (const long long *)S->d_woff, (const long long *)S->d_xoff, (const double *)S->Wbase,
                     (const double *)S->X, S->part, stride);
  hipLaunchKernelGGL(k_fxs_fin, dim3((unsigned)((S->nX / 2 + PMH_BLOCK - 1) / PMH_BLOCK)), dim3(PMH_BLOCK), 0, st, S->nX, S->nseg, stride, (const double *)S->part, S->Y);
  if (timed) {
    PMH_HIP(hipEventRecord(S->ev[2 * S->ev_used + 1], st));
    S->ev_used++;
  }
  PMH_HIP(hipGetLastError());
  return PMH_SUCCESS;
}

int fxs_apply(fx_shared *S, const double *lambda, double *y)
{
  PMH_CHK(pmh_gluing_mult(S->Bc, lambda, S->X));
  PMH_CHK(fxs_gemm(S));
  return pmh_gluing_mult_transpose(S->Bc, S->Y, y); // ends with the all-reduce on several GPUs
}

// the dense kernel alone (tests, tuning): Y = blockdiag(W_c) X on the multivectors as they stand
int fxs_dense(fx_shared *S) { return fxs_gemm(S); }
long long fxs_multivector_length(fx_shared *S) { return S->nX; }
double   *fxs_X(fx_shared *S) { return S->X; }
double   *fxs_Y(fx_shared *S) { return S->Y; }

int fxs_fill_pattern(fx_shared *S, int byte)
{
  PMH_HIP(hipMemsetAsync(S->Wbase, byte, sizeof(double) * (size_t)S->wtot, S->ctx->stream));
  return pmh_sync(S->ctx);
}

// W_b = W_c[pos_b, pos_b] on the host (tests); gamma: the block's touched dofs (rank-local primal indices, ascending)
int fxs_get_block(fx_shared *S, int b, int n, const int *gamma, double *out_host)
{
  const fxs_class    &C = S->C[S->cls[b]];
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
  S->ev_on = max_launches > 0, S->ev_used = 0;
  return PMH_SUCCESS;
}

int fxs_timing_get(fx_shared *S, int *launches, double *total_ms)
{
  PMH_CHK(pmh_sync(S->ctx));
  double tot = 0.0;
  for (int i = 0; i < S->ev_used; i++) {
    float ms = 0.f;
    PMH_HIP(hipEventElapsedTime(&ms, S->ev[2 * i], S->ev[2 * i + 1]));
    tot += ms;
  }
  *launches = S->ev_used, *total_ms = tot;
  return PMH_SUCCESS;
}
