// 3x3-block sparse matrix-vector product for the elasticity blocks K_i (PETSc's SeqBAIJ role, bs = 3) in a
// device-private layout.  A CSR row of K_i costs 12 bytes per non-zero (fp64 value + int32 column); with one column index
// per 3x3 block it is 8.44 (fp64) or 4.44 (fp32, used only inside the multigrid preconditioner) -- the kernel is HBM bound,
// so the byte count is the run time.
//
// Layout: block rows are grouped into tiles of at most 512 (fp64) / 1024 (fp32) blocks (whole block rows per tile); inside a tile the nine
// entries of the blocks are stored as nine planes of length nbt (structure of arrays), so that thread j reads entry k of
// block j at plane k, offset j: every load of the value stream is lane-contiguous.  One workgroup per tile: each thread
// multiplies its blocks with the three x entries of the block column, writes the three partial products to LDS, then four
// lanes per scalar row add the row's partials in a fixed order (deterministic, no atomics).
#include "pmh_internal.h"

#include <algorithm>

#define BSR_TB_MAX 1024 // largest tile (blocks); a tile is BSR_TB = 512 or 1024 blocks (2 or 4 per thread)

template <typename T, int EPI, int BSR_TB>
__global__ __launch_bounds__(PMH_BLOCK) void k_bsr3(const int *__restrict__ tile_br, int ntiles, const int *__restrict__ browptr, const int *__restrict__ bcol, const T *__restrict__ val, const T *__restrict__ x, T *__restrict__ y,
                                                     const T *__restrict__ y1, const int *__restrict__ halt)
{
  if (halt && *halt) return;
  __shared__ T prod[3][BSR_TB];
  const int    tid   = threadIdx.x;
  const int    chunk = gridDim.x >> 3; // XCD-aware: XCD x works on a contiguous slab of tiles (x stays in its L2)
  const int    t     = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (t >= ntiles) return;
  const int br0 = tile_br[t], br1 = tile_br[t + 1];
  const int s0 = browptr[br0], nbt = browptr[br1] - s0;
  const T  *v = val + (size_t)s0 * 9;
#pragma unroll
  for (int jj = 0; jj < BSR_TB / PMH_BLOCK; jj++) {
    const int j = tid + jj * PMH_BLOCK;
    if (j < nbt) {
      const int c = __builtin_nontemporal_load(&bcol[s0 + j]);
      T         a[9];
#pragma unroll
      for (int k = 0; k < 9; k++) a[k] = __builtin_nontemporal_load(&v[(size_t)k * nbt + j]);
      const T x0 = x[3 * c], x1 = x[3 * c + 1], x2 = x[3 * c + 2];
      prod[0][j] = a[0] * x0 + a[1] * x1 + a[2] * x2;
      prod[1][j] = a[3] * x0 + a[4] * x1 + a[5] * x2;
      prod[2][j] = a[6] * x0 + a[7] * x1 + a[8] * x2;
    }
  }
  __syncthreads();
  const int nrows = 3 * (br1 - br0), sub = tid >> 2, lane = tid & 3;
  for (int rb = 0; rb < nrows; rb += PMH_BLOCK / 4) { // uniform trip count for the shuffles
    const int q   = rb + sub;
    T         sum = (T)0;
    if (q < nrows) {
      const int br = br0 + q / 3, r = q % 3;
      const int k0 = browptr[br] - s0, k1 = browptr[br + 1] - s0;
      for (int k = k0 + lane; k < k1; k += 4) sum += prod[r][k];
    }
    sum += __shfl_down(sum, 2, 4);
    sum += __shfl_down(sum, 1, 4);
    if (q < nrows && lane == 0) {
      const int row = 3 * br0 + q;
      if (EPI == PMH_EPI_ADD) sum = y1[row] + sum;
      if (EPI == PMH_EPI_SUB) sum = sum - y1[row];
      y[row] = sum;
    }
  }
}

// Build from a resident CSR (downloaded once); *out = NULL without error when the matrix has no 3x3 block structure that
// fits the tile (n not a multiple of 3, or a block row with more blocks than a tile holds).
int pmh_bsr3_from_csr(pmh_csr A, int is_float, pmh_bsr3 *out)
{
  PMH_ARG(A && out);
  *out        = nullptr;
  pmh_ctx ctx = A->ctx;
  if (A->nrows != A->ncols || A->nrows % 3 || A->nrows == 0) return PMH_SUCCESS;
  const int        n = A->nrows, nbr = n / 3;
  int              tb = 512; // measured on MI355X (profiles/): 512 beats 1024 for both precisions
  if (const char *e = getenv("PMH_BSR_TB")) tb = atoi(e); // tuning knob
  if (tb != 512 && tb != 1024) return pmh_set_error(PMH_ERR_ARG, "PMH_BSR_TB must be 512 or 1024");
  std::vector<int> rp((size_t)n + 1), ci((size_t)A->nnz);
  std::vector<double> va((size_t)A->nnz);
  PMH_CHK(pmh_memcpy_d2h(ctx, rp.data(), A->d_rowptr, sizeof(int) * rp.size()));
  if (A->nnz) {
    PMH_CHK(pmh_memcpy_d2h(ctx, ci.data(), A->d_col, sizeof(int) * ci.size()));
    PMH_CHK(pmh_memcpy_d2h(ctx, va.data(), A->d_val, sizeof(double) * va.size()));
  }
  // block structure: union of the block columns of the three rows of each block row (sorted)
  std::vector<int> browptr((size_t)nbr + 1, 0), bcol;
  bcol.reserve((size_t)A->nnz / 9 + 16);
  std::vector<int> tmp;
  for (int br = 0; br < nbr; br++) {
    tmp.clear();
    for (int r = 0; r < 3; r++)
      for (int k = rp[3 * br + r]; k < rp[3 * br + r + 1]; k++) tmp.push_back(ci[k] / 3);
    std::sort(tmp.begin(), tmp.end());
    tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
    if ((int)tmp.size() > tb) return PMH_SUCCESS;
    bcol.insert(bcol.end(), tmp.begin(), tmp.end());
    if (bcol.size() > (size_t)0x7fffff00) return PMH_SUCCESS;
    browptr[br + 1] = (int)bcol.size();
  }
  const long long nblocks = (long long)bcol.size();
  if (nblocks * 9 > 2 * A->nnz + 64) return PMH_SUCCESS; // blocks mostly empty: the CSR kernel moves fewer bytes
  // tiles of whole block rows
  std::vector<int> tile_br(1, 0);
  for (int br = 0, start = 0; br < nbr; br++) {
    if (browptr[br + 1] - browptr[start] > tb) {
      tile_br.push_back(br);
      start = br;
    }
    if (br == nbr - 1) tile_br.push_back(nbr);
  }
  const int ntiles = (int)tile_br.size() - 1;
  // values, tile-wise structure of arrays
  std::vector<double> bv((size_t)nblocks * 9, 0.0);
  for (int t = 0; t < ntiles; t++) {
    const int    s0 = browptr[tile_br[t]], nbt = browptr[tile_br[t + 1]] - s0;
    double      *v  = bv.data() + (size_t)s0 * 9;
    for (int br = tile_br[t]; br < tile_br[t + 1]; br++) {
      const int *bc = bcol.data() + browptr[br];
      const int  nb = browptr[br + 1] - browptr[br];
      for (int r = 0; r < 3; r++)
        for (int k = rp[3 * br + r]; k < rp[3 * br + r + 1]; k++) {
          const int  c = ci[k] / 3, cc = ci[k] % 3;
          const int *p = std::lower_bound(bc, bc + nb, c);
          const int  j = (int)(p - bc) + browptr[br] - s0;
          v[(size_t)(3 * r + cc) * nbt + j] += va[k];
        }
    }
  }
  pmh_bsr3 B = new pmh_bsr3_s();
  B->ctx = ctx, B->n = n, B->nbr = nbr, B->ntiles = ntiles, B->nblocks = nblocks, B->is_float = is_float, B->tb = tb;
  B->ev_used = 0, B->ev_on = 0;
  PMH_CHK(pmh_malloc(ctx, sizeof(int) * tile_br.size(), (void **)&B->d_tile_br));
  PMH_CHK(pmh_malloc(ctx, sizeof(int) * browptr.size(), (void **)&B->d_browptr));
  PMH_CHK(pmh_malloc(ctx, sizeof(int) * (size_t)(nblocks + 1), (void **)&B->d_bcol));
  PMH_CHK(pmh_memcpy_h2d(ctx, B->d_tile_br, tile_br.data(), sizeof(int) * tile_br.size()));
  PMH_CHK(pmh_memcpy_h2d(ctx, B->d_browptr, browptr.data(), sizeof(int) * browptr.size()));
  PMH_CHK(pmh_memcpy_h2d(ctx, B->d_bcol, bcol.data(), sizeof(int) * (size_t)nblocks));
  if (is_float) {
    std::vector<float> bf(bv.size());
    for (size_t i = 0; i < bv.size(); i++) bf[i] = (float)bv[i];
    PMH_CHK(pmh_malloc(ctx, sizeof(float) * (bf.size() + 1), &B->d_val));
    PMH_CHK(pmh_memcpy_h2d(ctx, B->d_val, bf.data(), sizeof(float) * bf.size()));
  } else {
    PMH_CHK(pmh_malloc(ctx, sizeof(double) * (bv.size() + 1), &B->d_val));
    PMH_CHK(pmh_memcpy_h2d(ctx, B->d_val, bv.data(), sizeof(double) * bv.size()));
  }
  *out = B;
  return PMH_SUCCESS;
}

int pmh_bsr3_destroy(pmh_bsr3 B)
{
  if (!B) return PMH_SUCCESS;
  pmh_free(B->ctx, B->d_tile_br);
  pmh_free(B->ctx, B->d_browptr);
  pmh_free(B->ctx, B->d_bcol);
  pmh_free(B->ctx, B->d_val);
  for (auto e : B->ev) (void)hipEventDestroy(e);
  delete B;
  return PMH_SUCCESS;
}

// algorithmic bytes of one launch: values + one index per block, block-row pointers, x read once, y written once
double pmh_bsr3_bytes(pmh_bsr3 B)
{
  const double w = B->is_float ? 4.0 : 8.0;
  return (double)B->nblocks * (9.0 * w + 4.0) + 4.0 * (B->nbr + 1) + 2.0 * w * B->n;
}

template <typename T>
static int bsr3_launch(pmh_bsr3 B, const T *x, T *y, int epi, const T *y1, const int *halt)
{
  const dim3 grid((unsigned)(((B->ntiles + 7) / 8) * 8)), blk(PMH_BLOCK);
  hipStream_t st = B->ctx->stream;
  const bool  timed = B->ev_on && (size_t)(B->ev_used + 2) <= B->ev.size();
  if (timed) PMH_HIP(hipEventRecord(B->ev[B->ev_used], st));
  const int *tb = B->d_tile_br, *bp = B->d_browptr, *bc = B->d_bcol;
  const T   *v  = (const T *)B->d_val;
#define BSR_LAUNCH(EPI, TB) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bsr3<T, EPI, TB>), grid, blk, 0, st, tb, B->ntiles, bp, bc, v, x, y, y1, halt)
#define BSR_LAUNCH_TB(EPI) \
  do { \
    if (B->tb == 512) BSR_LAUNCH(EPI, 512); \
    else BSR_LAUNCH(EPI, 1024); \
  } while (0)
  if (epi == PMH_EPI_NONE) BSR_LAUNCH_TB(PMH_EPI_NONE);
  else if (epi == PMH_EPI_ADD) BSR_LAUNCH_TB(PMH_EPI_ADD);
  else if (epi == PMH_EPI_SUB) BSR_LAUNCH_TB(PMH_EPI_SUB);
  else return pmh_set_error(PMH_ERR_ARG, "bsr3: unsupported epilogue %d", epi);
  PMH_HIP(hipGetLastError());
  if (timed) {
    PMH_HIP(hipEventRecord(B->ev[B->ev_used + 1], st));
    B->ev_used += 2;
  }
  return PMH_SUCCESS;
}

int pmh_bsr3_spmv_f64(pmh_bsr3 B, const double *x, double *y, int epi, const double *y1, const int *halt)
{
  PMH_ARG(B && !B->is_float);
  return bsr3_launch<double>(B, x, y, epi, y1, halt);
}

int pmh_bsr3_spmv_f32(pmh_bsr3 B, const float *x, float *y, int epi, const float *y1, const int *halt)
{
  PMH_ARG(B && B->is_float);
  return bsr3_launch<float>(B, x, y, epi, y1, halt);
}

int pmh_bsr3_timing_enable(pmh_bsr3 B, int max_launches)
{
  PMH_ARG(B && max_launches >= 0);
  PMH_HIP(hipStreamSynchronize(B->ctx->stream));
  while ((int)B->ev.size() < 2 * max_launches) {
    hipEvent_t e;
    PMH_HIP(hipEventCreate(&e));
    B->ev.push_back(e);
  }
  B->ev_used = 0;
  B->ev_on   = max_launches > 0;
  return PMH_SUCCESS;
}

// launches whose duration is below a quarter of the longest are halted no-ops and are not counted
int pmh_bsr3_timing_get(pmh_bsr3 B, int *launches, double *total_ms)
{
  PMH_ARG(B && launches && total_ms);
  PMH_HIP(hipStreamSynchronize(B->ctx->stream));
  std::vector<float> ms(B->ev_used / 2);
  float              mx = 0.f;
  for (int i = 0; i < B->ev_used / 2; i++) {
    PMH_HIP(hipEventElapsedTime(&ms[i], B->ev[2 * i], B->ev[2 * i + 1]));
    mx = std::max(mx, ms[i]);
  }
  *launches = 0, *total_ms = 0.0;
  for (float m : ms)
    if (m >= 0.25f * mx) (*launches)++, *total_ms += m;
  return PMH_SUCCESS;
}
