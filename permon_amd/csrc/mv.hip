// Multi-right-hand-side K^+, part 1: the operator.  ELL copy of a matrix of 3 x 3 blocks (built on the device from the resident CSR) and y = A x on interleaved
// multivectors of R = PMH_MV_R columns with the smoothing epilogues of k_bsr3 (bsr.hip).  See mv_internal.h for what this path is for.
//
// FOUR lanes per block row (a first form with one thread per block row left the chip at 1.2 waves per SIMD: 92 us per fp64 product of a 43^3 block where its
// bytes need 35): lane l of the quad takes the slots s = 4 g + l, per slot ONE block column index, the 9 entries of the block (the slot planes are interleaved
// by 4, so a wave reads 512 contiguous bytes per entry plane) and the 3 R operand values of that block column -- ONE contiguous piece (192 bytes in fp64, 96 in
// fp32) loaded as 16-byte vectors.  3 R accumulators per lane, summed across the quad by two butterfly steps (a fixed order); lanes 0 .. 2 of the quad then
// finish one row each.
#include "mv_internal.h"

static thread_local const char *g_mv_why = "";
const char *pmh_mv_why() { return g_mv_why; }
void        pmh_mv_set_why(const char *why) { g_mv_why = why; }

typedef _Float16 mv_half4 __attribute__((ext_vector_type(4)));
typedef double   mv_dbl2 __attribute__((ext_vector_type(2)));
typedef float    mv_flt4 __attribute__((ext_vector_type(4)));

// ---- ELL builder --------------------------------------------------------------------------------------------------------------------------------------------
// The blocks of block row br = the sorted union of the block columns (column / 3) its three rows list; an entry a row does not store is a zero of the block.  A
// three-way merge over the (sorted) rows: walk(br, ...) calls f(slot, block column, the 9 entries) slot after slot and returns the slot count, or -1 for a row
// whose columns are not ascending.
template <typename F>
static __device__ __forceinline__ int mv_walk(int br, const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val, F f)
{
  int k[3], e[3], s = 0;
#pragma unroll
  for (int q = 0; q < 3; q++) k[q] = rowptr[3 * br + q], e[q] = rowptr[3 * br + q + 1];
  while (k[0] < e[0] || k[1] < e[1] || k[2] < e[2]) {
    int cb = 0x7fffffff;
#pragma unroll
    for (int q = 0; q < 3; q++)
      if (k[q] < e[q]) cb = min(cb, col[k[q]] / 3);
    double a[9];
#pragma unroll
    for (int i = 0; i < 9; i++) a[i] = 0.0;
#pragma unroll
    for (int q = 0; q < 3; q++) {
      int last = -1;
      while (k[q] < e[q] && col[k[q]] / 3 == cb) {
        const int c = col[k[q]] - 3 * cb;
        if (c <= last) return -1;
        last = c;
        const double v = val ? val[k[q]] : 0.0;
        if (c == 0) a[3 * q] = v;
        else if (c == 1) a[3 * q + 1] = v;
        else a[3 * q + 2] = v;
        k[q]++;
      }
      if (k[q] < e[q] && col[k[q]] / 3 < cb) return -1;
    }
    f(s, cb, a);
    s++;
  }
  return s;
}

// info[0] = the largest slot count of a block row, info[1] = rows that are not sorted
__global__ __launch_bounds__(PMH_BLOCK) void k_mv_ell_count(int nbr, const int *__restrict__ rowptr, const int *__restrict__ col, int *__restrict__ info)
{
  const int br = blockIdx.x * PMH_BLOCK + threadIdx.x;
  if (br >= nbr) return;
  const int n = mv_walk(br, rowptr, col, (const double *)nullptr, [](int, int, const double *) {});
  if (n < 0) atomicAdd(&info[1], 1);
  else atomicMax(&info[0], n);
}

__global__ __launch_bounds__(PMH_BLOCK) void k_mv_absmax(long long nnz, const double *__restrict__ val, unsigned long long *__restrict__ out)
{
  double m = 0.0;
  for (long long k = (long long)blockIdx.x * PMH_BLOCK + threadIdx.x; k < nnz; k += (long long)gridDim.x * PMH_BLOCK) m = fmax(m, fabs(val[k]));
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_down(m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(m)); // non-negative doubles order as their bit patterns
}

// Layout of the entries (round 6: until then one plane per entry -- 9 loads of 8 / 4 bytes per slot, fp16 three of 8 with every fourth half unused; the product is bound by
// the number of vector-memory instructions its wavefronts can issue, ~ 41 cycles each (SQ counters, profiles/r06_mv_spmv_pmc.txt)).  Per plane g of 4 slots x nbr block rows:
// the entries 0 ... 7 of a slot as NW = 8 / VW vectors of 16 bytes (VW = 2 / 4 / 8 entries), vector k of (br, l) at 16-byte index (k nbr + br) 4 + l, then entry 8 of (br, l) as
// one element at index 4 br + l behind them; a plane starts at a multiple of 16 bytes.  5 / 3 / 2 loads per slot instead of 9 / 9 / 3, fp16 18 instead of 24 bytes.
template <typename TM> struct mv_lay {
  static constexpr int VW = 16 / (int)sizeof(TM), NW = 8 / VW;
  static __host__ __device__ __forceinline__ size_t plane_bytes(int nbr) { return ((size_t)nbr * 4 * 9 * sizeof(TM) + 15) & ~(size_t)15; }
};
template <typename TM> static __device__ __forceinline__ TM mv_entry(double v);
template <> __device__ __forceinline__ double mv_entry<double>(double v) { return v; }
template <> __device__ __forceinline__ float mv_entry<float>(double v) { return (float)v; }
template <> __device__ __forceinline__ _Float16 mv_entry<_Float16>(double v) { return (_Float16)(float)v; }

template <typename TM>
__global__ __launch_bounds__(PMH_BLOCK) void k_mv_ell_fill(int nbr, int W, const int *__restrict__ rowptr, const int *__restrict__ col,
                        const double *__restrict__ val, double inv_scale, int *__restrict__ ecol, char *__restrict__ eval, int rect)
{
  const int br = blockIdx.x * PMH_BLOCK + threadIdx.x;
  if (br >= nbr) return;
  constexpr int VW = mv_lay<TM>::VW, NW = mv_lay<TM>::NW;
  const size_t  pb = mv_lay<TM>::plane_bytes(nbr);
  auto put = [&](int s, int cb, const double *a) {
    ecol[((size_t)(s >> 2) * nbr + br) * 4 + (s & 3)] = cb;
    char *p = eval + (size_t)(s >> 2) * pb;
#pragma unroll
    for (int i = 0; i < 8; i++) ((TM *)(p + (((size_t)(i / VW) * nbr + br) * 4 + (s & 3)) * 16))[i % VW] = mv_entry<TM>(a[i] * inv_scale);
    ((TM *)(p + (size_t)NW * nbr * 64))[br * 4 + (s & 3)] = mv_entry<TM>(a[8] * inv_scale);
  };
  const int    n = mv_walk(br, rowptr, col, val, put);
  const double z[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int s = n; s < W; s++) put(s, rect ? 0 : br, z); // padding: the row's own block column (a rectangular matrix: block column 0), zero entries
}

static int mv_ell_build(pmh_csr A, int nrep, int storage, int rect, int negate, pmh_mv_ell *out);
int pmh_mv_ell_create(pmh_csr A, int storage, pmh_mv_ell *out) { return pmh_mv_ell_create_prefix(A, 1, storage, out); }
int pmh_mv_ell_create_prefix(pmh_csr A, int nrep, int storage, pmh_mv_ell *out) { return mv_ell_build(A, nrep, storage, 0, 0, out); }
int pmh_mv_ell_create_rect(pmh_csr A, int storage, int negate, pmh_mv_ell *out) { return mv_ell_build(A, 1, storage, 1, negate, out); }

// nrep > 1: A is block diagonal with nrep congruent blocks (the caller has verified it): the ELL copy of its FIRST block (rows / columns [0, n / nrep), nnz /
// nrep entries).  rect: A may be rectangular (3 | rows, 3 | columns).  negate: the copy holds -A
static int mv_ell_build(pmh_csr A, int nrep, int storage, int rect, int negate, pmh_mv_ell *out)
{
  PMH_ARG(A && out && nrep >= 1 && (storage == PMH_BSR_F64 || storage == PMH_BSR_F32 || storage == PMH_BSR_F16) && !(rect && nrep > 1));
  *out        = nullptr;
  pmh_ctx ctx = A->ctx;
  if ((!rect && A->nrows != A->ncols) || A->ncols % 3 || A->nrows % (3 * nrep) || A->nrows == 0 || A->nnz % nrep) return PMH_SUCCESS;
  const int       nbr  = A->nrows / 3 / nrep;
  const long long nnzb = A->nnz / nrep;
  hipStream_t st  = ctx->stream;
  int        *d_info;
  PMH_CHK(pmh_malloc(ctx, sizeof(unsigned long long) * 2, (void **)&d_info));
  PMH_HIP(hipMemsetAsync(d_info, 0, sizeof(unsigned long long) * 2, st));
  hipLaunchKernelGGL(k_mv_ell_count, dim3((nbr + PMH_BLOCK - 1) / PMH_BLOCK), dim3(PMH_BLOCK), 0, st, nbr, (const int *)A->d_rowptr, (const int *)A->d_col,
                     d_info);
  int info[2];
  PMH_CHK(pmh_memcpy_d2h(ctx, info, d_info, sizeof(info)));
  // (32 slots until round 6: the coarse operators of an aggregation hierarchy couple an aggregate's two 3 x 3 block rows to 50 ... 150 block columns, and the small levels
  // near the bottom of such a hierarchy are almost dense: up to 2048 slots, as long as the padded copy stays under 2 GB)
  if (info[1] || info[0] < 1 || info[0] > 2048 || (double)((info[0] + 15) / 16 * 16) * nbr * 76.0 > 2.0e9) {
    pmh_free(ctx, d_info);
    return PMH_SUCCESS;
  }
  pmh_mv_ell E = new pmh_mv_ell_s();
  E->ctx = ctx, E->nbr = nbr, E->storage = storage, E->scale = 1.0, E->col = nullptr, E->val = nullptr;
  // long rows (W > 48: the coarse operators of an aggregation hierarchy) and small levels (fewer block rows than fill the chip with 4 lanes each: the trips of a row are a
  // serial chain -- 7 of them for a 27-point operator) take 16 lanes per block row
  E->lpr = (info[0] > 48 || nbr < 16384) ? 16 : 4;
  E->W   = (info[0] + E->lpr - 1) / E->lpr * E->lpr;
  double inv_scale = 1.0;
  if (storage == PMH_BSR_F16) { // power-of-two scale that brings the largest entry to [1, 2) (as pmh_bsr3_from_csr)
    PMH_HIP(hipMemsetAsync(d_info, 0, sizeof(unsigned long long) * 2, st));
    hipLaunchKernelGGL(k_mv_absmax, dim3(1024), dim3(PMH_BLOCK), 0, st, nnzb, (const double *)A->d_val, (unsigned long long *)d_info);
    double amax = 0.0;
    PMH_CHK(pmh_memcpy_d2h(ctx, &amax, d_info, sizeof(double)));
    int ex = 0;
    if (amax > 0.0) frexp(amax, &ex);
    E->scale  = ldexp(1.0, ex - 1);
    inv_scale = 1.0 / E->scale;
    if (negate) E->scale = -E->scale;
  } else if (negate) inv_scale = -1.0;
  pmh_free(ctx, d_info);
  const size_t nslot = (size_t)E->W * nbr;
  const dim3   g((nbr + PMH_BLOCK - 1) / PMH_BLOCK), blk(PMH_BLOCK);
  int          rc = pmh_malloc(ctx, sizeof(int) * nslot, (void **)&E->col);
  const size_t planeb = storage == PMH_BSR_F64 ? mv_lay<double>::plane_bytes(nbr) : (storage == PMH_BSR_F32 ? mv_lay<float>::plane_bytes(nbr) : mv_lay<_Float16>::plane_bytes(nbr));
  if (!rc) rc = pmh_malloc(ctx, planeb * (size_t)(E->W / 4), &E->val);
  if (rc) {
    pmh_mv_ell_destroy(E);
    return rc;
  }
  if (storage == PMH_BSR_F64) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mv_ell_fill<double>), g, blk, 0, st, nbr, E->W, (const int *)A->d_rowptr,
                          (const int *)A->d_col, (const double *)A->d_val, inv_scale, E->col, (char *)E->val, rect);
  else if (storage == PMH_BSR_F32) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mv_ell_fill<float>), g, blk, 0, st, nbr, E->W, (const int *)A->d_rowptr,
                          (const int *)A->d_col, (const double *)A->d_val, inv_scale, E->col, (char *)E->val, rect);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mv_ell_fill<_Float16>), g, blk, 0, st, nbr, E->W, (const int *)A->d_rowptr, (const int *)A->d_col, (const double *)A->d_val, inv_scale,
                          E->col, (char *)E->val, rect);
  PMH_HIP(hipGetLastError());
  *out = E;
  return PMH_SUCCESS;
}

int pmh_mv_ell_destroy(pmh_mv_ell E)
{
  if (!E) return PMH_SUCCESS;
  pmh_free(E->ctx, E->col);
  pmh_free(E->ctx, E->val);
  delete E;
  return PMH_SUCCESS;
}

// ---- the product --------------------------------------------------------------------------------------------------------------------------------------------
// (Round 6, built, measured, dropped: a lane map with four COLUMN-PAIR lanes per slot -- 16 lanes per block row, every lane 2 of the 8 columns, the operand of a slot read
// as 4 adjacent 8 / 16-byte pieces instead of 6 / 12 loads of 16 bytes per lane.  Same bits; 78 / 39 / 38 us per product of one 43^3 block with fp64 / fp32 / fp16 entries
// against 58 / 28 / 25.5 us of the map below: the four lanes of a slot each issue the loads of the 3 x 3 block, and the texture addresser's cost is per quad of lanes,
// not per distinct address.  What bounds the product is that rate -- one cache line per clock and CU; a 16-byte piece of a gathered operand costs a line access of its own.
// Also built, measured, dropped: the operand STAGED in LDS per tile of 64 (fp64: 32) block rows -- the tile's distinct block columns copied with coalesced 16-byte loads,
// the slots addressing them by 16-bit local indices (tables built on the host).  Same bits; 81 / 31 / 30 us: 45 ... 90 KB of LDS per workgroup leave 1 - 3 workgroups per
// CU, and their copy phases do not overlap anybody's products.
// Built, measured, KEPT (later in round 6): the quad of a block row loads its trip's four operand pieces together and exchanges them through its own LDS lines -- the
// MV_STAGED branch of the kernel below; 57.8 / 27.3 / 25.4 -> 51.0 / 26.1 / 23.8 us.  Control experiment behind it: the operand loads replaced by same-sized loads from
// contiguous addresses (wrong results, timing only) 35.9 / 18.5 / 15.6 us.)
template <typename T, int N> struct mv_vec;
template <int N> struct mv_vec<double, N> { // N doubles = N / 2 loads of 16 bytes
  static __device__ __forceinline__ void load(const double *p, double (&v)[N])
  {
#pragma unroll
    for (int k = 0; k < N / 2; k++) {
      const mv_dbl2 t = ((const mv_dbl2 *)p)[k];
      v[2 * k] = t.x, v[2 * k + 1] = t.y;
    }
  }
  static __device__ __forceinline__ void store(double *p, const double (&v)[N])
  {
#pragma unroll
    for (int k = 0; k < N / 2; k++) ((mv_dbl2 *)p)[k] = mv_dbl2{v[2 * k], v[2 * k + 1]};
  }
};
template <int N> struct mv_vec<float, N> {
  static __device__ __forceinline__ void load(const float *p, float (&v)[N])
  {
#pragma unroll
    for (int k = 0; k < N / 4; k++) {
      const mv_flt4 t = ((const mv_flt4 *)p)[k];
      v[4 * k] = t.x, v[4 * k + 1] = t.y, v[4 * k + 2] = t.z, v[4 * k + 3] = t.w;
    }
  }
  static __device__ __forceinline__ void store(float *p, const float (&v)[N])
  {
#pragma unroll
    for (int k = 0; k < N / 4; k++) ((mv_flt4 *)p)[k] = mv_flt4{v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
  }
};

template <typename TM, typename T> struct mv_blk { // the 9 entries of slot 4 g + l, block row br (layout: mv_lay)
  static __device__ __forceinline__ void load(const void *val, size_t g, int nbr, int br, int l, T (&a)[9]) // g: plane (slot >> 2), l: slot & 3
  {
    constexpr int VW = mv_lay<TM>::VW, NW = mv_lay<TM>::NW;
    const char   *p  = (const char *)val + g * mv_lay<TM>::plane_bytes(nbr);
#pragma unroll
    for (int k = 0; k < NW; k++) {
      const mv_flt4 w = __builtin_nontemporal_load((const mv_flt4 *)p + ((size_t)k * nbr + br) * 4 + l);
      TM            t[VW];
      __builtin_memcpy(t, &w, 16);
#pragma unroll
      for (int j = 0; j < VW; j++) a[k * VW + j] = (T)t[j];
    }
    a[8] = (T)__builtin_nontemporal_load((const TM *)(p + (size_t)NW * nbr * 64) + br * 4 + l);
  }
};

// LPR lanes per block row (4; 16 for the long rows of an aggregation hierarchy's coarse operators: 7 500 block rows of ~ 80 ... 150 blocks left the chip idle with 4):
// lane l takes the slots LPR g + l, i.e. plane (LPR / 4) g + (l >> 2), entry l & 3 of the plane
#ifndef MV_STAGED
#define MV_STAGED 1
#endif
#ifndef MV_EARLY_EPI
#define MV_EARLY_EPI 1
#endif
template <typename TM, typename T, int R, int EPI, int LPR = 4>
__global__ __launch_bounds__(PMH_BLOCK) void k_mv_spmv(int nbr, int W4, const int *__restrict__ col, const void *__restrict__ val, T scale,
                        const T *__restrict__ x, T *__restrict__ y, pmh_mv_epi<T> e, const int *__restrict__ halt, int xcd_map)
{
  const int hlt = halt ? *halt : 0;
  // XCD-aware order of the workgroups: consecutive workgroup ids land on different XCDs (8 private L2s), so with the natural order every XCD walks the WHOLE operand
  // multivector through its 4 MB L2 -- the gathers of a block row reach +- one plane of nodes, ~ 30 workgroups away.  Workgroup b takes the b / 8-th tile of the (b % 8)-th
  // contiguous eighth of the block rows: an XCD's gathers stay inside its own slab (+ a halo).  Same bits; 5 - 7 % of the product (round 6)
  int wg = blockIdx.x;
  if (xcd_map) {
    const int nwg = gridDim.x, q = nwg >> 3, rr = nwg & 7, xc = wg & 7, j = wg >> 3;
    wg = xc * q + min(xc, rr) + j;
  }
  const int tg  = wg * PMH_BLOCK + threadIdx.x, br = tg / LPR, lw = tg % LPR, l = lw & 3, pl = lw >> 2;
  constexpr int PS = LPR / 4; // planes per trip
  if (br >= nbr) return; // whole groups of LPR lanes
  int cn = col[((size_t)pl * nbr + br) * 4 + l];
  if (hlt) return;
  T acc[3][R];
#pragma unroll
  for (int q = 0; q < 3; q++)
#pragma unroll
    for (int r = 0; r < R; r++) acc[q][r] = (T)0;
  // the epilogue's operands asked for BEFORE the slots (fp32 vectors, 4 lanes per block row): the wavefronts of a launch are all resident and reach their epilogues together --
  // a streaming phase behind the gather phase instead of under it
  constexpr bool EARLY = MV_EARLY_EPI && sizeof(T) == 4 && LPR == 4;
  T              t1[R], t2[R], t3[R], di0 = (T)0;
  if (EARLY && lw < 3) {
    const size_t o0 = ((size_t)3 * br + l) * R;
    if (EPI == PMH_EPI_ADD || EPI == PMH_EPI_SUB) mv_vec<T, R>::load(e.y1 + o0, t1);
    if (EPI == PMH_BSR_EPI_PRE || EPI == PMH_BSR_EPI_POST1) {
      mv_vec<T, R>::load(e.y1 + o0, t1);
      mv_vec<T, R>::load(x + o0, t2);
      di0 = e.dinv[3 * br + l];
    }
    if (EPI == PMH_BSR_EPI_POST2) {
      mv_vec<T, R>::load(y + o0, t1);
      mv_vec<T, R>::load(x + o0, t2);
      mv_vec<T, R>::load(e.r + o0, t3);
      di0 = e.dinv[3 * br + l];
    }
  }
  if constexpr (LPR == 4 && MV_STAGED) {
    // Round 6, the operand through LDS.  A lane's slot needs the 3 R operand values of ITS block column -- NP = 6 (fp64: 12) pieces of 16 bytes, each lane of a quad from
    // another block column: four cache-line accesses per quad and load instruction, the rate that bounds the product (header comment).  Here the quad loads its trip's four
    // pieces of 3 R values TOGETHER: lane l of the quad takes the pieces p = 4 j + l, j < NP, of their concatenation, four adjacent pieces = 64 contiguous bytes of one
    // (or two) block columns per instruction -- one or two line accesses, the same number of instructions; the pieces go to the quad's own 64 bytes x NP of LDS and every lane
    // reads its block column's NP pieces back.  A wavefront's LDS operations execute in order and a quad exchanges with nobody else: no barrier.  Same values into the same
    // FMAs: the same bits.
    constexpr int NP = 3 * R * (int)sizeof(T) / 16, QS = 4 * NP * 16 + (NP == 6 ? 64 : 16); // bytes per quad (the padding that leaves the 16-byte stores of a lane group of 8 / the loads of one of 16 the fewest bank conflicts)
    __shared__ __attribute__((aligned(16))) char stage_[PMH_BLOCK / 4 * QS];
    char *const  sq = stage_ + (threadIdx.x >> 2) * QS;
    // piece p = 4 j + l lies in the chunk of quad lane p / NP at 16 (p % NP) bytes
    auto gather = [&](int c, mv_flt4(&pc)[NP]) {
      int cq[4];
      cq[0] = __builtin_amdgcn_update_dpp(0, c, 0x00, 0xf, 0xf, true), cq[1] = __builtin_amdgcn_update_dpp(0, c, 0x55, 0xf, 0xf, true);
      cq[2] = __builtin_amdgcn_update_dpp(0, c, 0xAA, 0xf, 0xf, true), cq[3] = __builtin_amdgcn_update_dpp(0, c, 0xFF, 0xf, 0xf, true);
#pragma unroll
      for (int j = 0; j < NP; j++) {
        const int lo = (4 * j) / NP, hi = (4 * j + 3) / NP; // the chunks this instruction touches (compile-time: at most two)
        const int p  = 4 * j + l;
        const int cc = (lo == hi || p / NP == lo) ? cq[lo] : cq[hi];
        pc[j]        = *(const mv_flt4 *)((const char *)(x + (size_t)3 * cc * R) + 16 * (p % NP));
      }
    };
    auto exchange = [&](const mv_flt4(&pc)[NP], T(&xv)[3 * R]) {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < NP; j++) *(mv_flt4 *)(sq + 16 * (4 * j + l)) = pc[j];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int o = 0; o < NP; o++) {
        const mv_flt4 t = *(const mv_flt4 *)(sq + 16 * (NP * l + o));
        __builtin_memcpy(&xv[o * (16 / (int)sizeof(T))], &t, 16);
      }
      __builtin_amdgcn_wave_barrier();
    };
    // (No operand prefetch across trips: the pieces of trip g + 1 asked for during the products of trip g cost 24 / 48 registers -- fp16 entries 124 -> 100 VGPRs = 5 instead of
    // 4 wavefronts per SIMD, 23.8 -> 23.1 us; fp64 192 -> 146, 52.0 -> 51.0 us.  Forcing 6 per SIMD spills 21 registers: 46 us.  The entries one trip ahead: 51.0 -> 55.8 us.)
    for (int g = 0; g < W4; g++) {
      T       a[9], xv[3 * R];
      mv_flt4 pc[NP];
      gather(cn, pc);
      if (g + 1 < W4) cn = col[((size_t)(g + 1) * nbr + br) * 4 + l];
      mv_blk<TM, T>::load(val, (size_t)g, nbr, br, l, a);
      exchange(pc, xv);
#pragma unroll
      for (int q = 0; q < 3; q++)
#pragma unroll
        for (int r = 0; r < R; r++) acc[q][r] += a[3 * q] * xv[r] + a[3 * q + 1] * xv[R + r] + a[3 * q + 2] * xv[2 * R + r];
    }
  } else if constexpr (sizeof(T) == 4) {
    // software pipeline (fp32 vectors: the V-cycle): the entries and the operand values of trip g + 1 are loaded before the products of trip g (the index of trip g + 2 with
    // them).  Measured on one 43^3 block, 8 columns: fp32 entries 34.6 -> 29.5 us, fp16 entries unchanged (27 us); with fp64 vectors the second operand set costs
    // the occupancy more than the chain costs (57 -> 62 us): the plain loop below
    T a[9], xv[3 * R];
    mv_blk<TM, T>::load(val, (size_t)pl, nbr, br, l, a);
    mv_vec<T, 3 * R>::load(x + (size_t)3 * cn * R, xv);
    if (pl + PS < W4) cn = col[((size_t)(pl + PS) * nbr + br) * 4 + l];
    for (int g = pl; g < W4; g += PS) {
      T          an[9], xn[3 * R];
      const bool more = g + PS < W4;
      if (more) {
        mv_blk<TM, T>::load(val, (size_t)(g + PS), nbr, br, l, an);
        mv_vec<T, 3 * R>::load(x + (size_t)3 * cn * R, xn);
        if (g + 2 * PS < W4) cn = col[((size_t)(g + 2 * PS) * nbr + br) * 4 + l];
      }
#pragma unroll
      for (int q = 0; q < 3; q++)
#pragma unroll
        for (int r = 0; r < R; r++) acc[q][r] += a[3 * q] * xv[r] + a[3 * q + 1] * xv[R + r] + a[3 * q + 2] * xv[2 * R + r];
      if (more) {
#pragma unroll
        for (int i = 0; i < 9; i++) a[i] = an[i];
#pragma unroll
        for (int i = 0; i < 3 * R; i++) xv[i] = xn[i];
      }
    }
  } else {
    for (int g = pl; g < W4; g += PS) {
      const int c = cn;
      if (g + PS < W4) cn = col[((size_t)(g + PS) * nbr + br) * 4 + l]; // the next slot's index travels during this slot's products
      T a[9], xv[3 * R];
      mv_blk<TM, T>::load(val, (size_t)g, nbr, br, l, a);
      mv_vec<T, 3 * R>::load(x + (size_t)3 * c * R, xv);
#pragma unroll
      for (int q = 0; q < 3; q++)
#pragma unroll
        for (int r = 0; r < R; r++) acc[q][r] += a[3 * q] * xv[r] + a[3 * q + 1] * xv[R + r] + a[3 * q + 2] * xv[2 * R + r];
    }
  }
  // the group's partial sums: (l0 + l1) + (l2 + l3) [+ the other planes' quads] on every lane
#pragma unroll
  for (int q = 0; q < 3; q++)
#pragma unroll
    for (int r = 0; r < R; r++) {
      T v = acc[q][r];
      v += __shfl_xor(v, 1, 64);
      v += __shfl_xor(v, 2, 64);
      if (LPR > 4) v += __shfl_xor(v, 4, 64);
      if (LPR > 8) v += __shfl_xor(v, 8, 64);
      acc[q][r] = v;
    }
  if (lw >= 3) return;
  // lane l finishes row 3 br + l: R contiguous values
  const size_t o = ((size_t)3 * br + l) * R;
  T            out[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    const T v = (l == 0) ? acc[0][r] : (l == 1 ? acc[1][r] : acc[2][r]);
    out[r]    = (sizeof(TM) == 2) ? v * scale : v; // fp16 storage keeps A / scale
  }
  if (EPI == PMH_EPI_ADD || EPI == PMH_EPI_SUB) {
    if (!EARLY) mv_vec<T, R>::load(e.y1 + o, t1);
#pragma unroll
    for (int k = 0; k < R; k++) out[k] = (EPI == PMH_EPI_ADD) ? t1[k] + out[k] : out[k] - t1[k];
  }
  if (EPI == PMH_MV_EPI_RESTRICT) { // the coarse level's first smoothing direction d = dinv y c0
    if (e.d) {
      const T di = e.dinv[3 * br + l] * e.c0;
#pragma unroll
      for (int k = 0; k < R; k++) t1[k] = di * out[k];
      mv_vec<T, R>::store(e.d + o, t1);
    }
  }
  if (EPI == PMH_BSR_EPI_PRE) { // y = c0 d0 + c2 dinv (b - A d0)
    if (!EARLY) mv_vec<T, R>::load(e.y1 + o, t1), mv_vec<T, R>::load(x + o, t2);
    const T di = EARLY ? di0 : e.dinv[3 * br + l];
#pragma unroll
    for (int k = 0; k < R; k++) out[k] = e.c0 * t2[k] + e.c2 * di * (t1[k] - out[k]);
  }
  if (EPI == PMH_BSR_EPI_POST1) { // r = dinv (b - A x); d = c0 r; y = x + d
    if (!EARLY) mv_vec<T, R>::load(e.y1 + o, t1), mv_vec<T, R>::load(x + o, t2);
    const T di = EARLY ? di0 : e.dinv[3 * br + l];
    T       rr[R], dd[R];
#pragma unroll
    for (int k = 0; k < R; k++) {
      rr[k]  = di * (t1[k] - out[k]);
      dd[k]  = e.c0 * rr[k];
      out[k] = t2[k] + dd[k];
    }
    mv_vec<T, R>::store(e.r + o, rr);
    mv_vec<T, R>::store(e.d + o, dd);
  }
  if (EPI == PMH_BSR_EPI_POST2) { // y += c1 d + c2 (r - dinv A d)
    if (!EARLY) mv_vec<T, R>::load(y + o, t1), mv_vec<T, R>::load(x + o, t2), mv_vec<T, R>::load(e.r + o, t3);
    const T di = EARLY ? di0 : e.dinv[3 * br + l];
#pragma unroll
    for (int k = 0; k < R; k++) out[k] = t1[k] + e.c1 * t2[k] + e.c2 * (t3[k] - di * out[k]);
    if (e.z64) {
#pragma unroll
      for (int k = 0; k < R; k++) e.z64[o + k] = (double)out[k];
    }
  }
  mv_vec<T, R>::store(y + o, out);
}

template <typename TM, typename T>
static int mv_launch(pmh_mv_ell E, const T *x, T *y, int epi, const pmh_mv_epi<T> *ep, const int *halt)
{
  const dim3    g((unsigned)(((long long)E->nbr * E->lpr + PMH_BLOCK - 1) / PMH_BLOCK)), blk(PMH_BLOCK);
  hipStream_t   st = E->ctx->stream;
  pmh_mv_epi<T> e;
  if (ep) e = *ep;
  else memset(&e, 0, sizeof(e));
  const T sc = (T)E->scale;
  const int xmap = g.x >= 64 ? 1 : 0;
#define MV_LAUNCH(EPI) \
  do { \
    if (E->lpr == 16) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mv_spmv<TM, T, PMH_MV_R, EPI, 16>), g, blk, 0, st, E->nbr, E->W / 4, (const int *)E->col, (const void *)E->val, sc, x, y, e, halt, xmap); \
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mv_spmv<TM, T, PMH_MV_R, EPI, 4>), g, blk, 0, st, E->nbr, E->W / 4, (const int *)E->col, (const void *)E->val, sc, x, y, e, halt, xmap); \
  } while (0)
  switch (epi) {
  case PMH_EPI_NONE: MV_LAUNCH(PMH_EPI_NONE); break;
  case PMH_EPI_ADD: MV_LAUNCH(PMH_EPI_ADD); break;
  case PMH_EPI_SUB: MV_LAUNCH(PMH_EPI_SUB); break;
  case PMH_BSR_EPI_PRE: MV_LAUNCH(PMH_BSR_EPI_PRE); break;
  case PMH_BSR_EPI_POST1: MV_LAUNCH(PMH_BSR_EPI_POST1); break;
  case PMH_BSR_EPI_POST2: MV_LAUNCH(PMH_BSR_EPI_POST2); break;
  case PMH_MV_EPI_RESTRICT: MV_LAUNCH(PMH_MV_EPI_RESTRICT); break;
  default: return pmh_set_error(PMH_ERR_ARG, "mv: unsupported epilogue %d", epi);
  }
#undef MV_LAUNCH
  PMH_HIP(hipGetLastError());
  return PMH_SUCCESS;
}

int pmh_mv_spmv_f64(pmh_mv_ell E, const double *x, double *y, int epi, const pmh_mv_epi<double> *e, const int *halt)
{
  PMH_ARG(E && x && y);
  if (E->storage != PMH_BSR_F64) return pmh_set_error(PMH_ERR_ARG, "pmh_mv_spmv_f64: the fp64 product needs fp64 entries");
  return mv_launch<double, double>(E, x, y, epi, e, halt);
}

int pmh_mv_spmv_f32(pmh_mv_ell E, const float *x, float *y, int epi, const pmh_mv_epi<float> *e, const int *halt)
{
  PMH_ARG(E && x && y);
  if (E->storage == PMH_BSR_F16) return mv_launch<_Float16, float>(E, x, y, epi, e, halt);
  if (E->storage == PMH_BSR_F32) return mv_launch<float, float>(E, x, y, epi, e, halt);
  return pmh_set_error(PMH_ERR_ARG, "pmh_mv_spmv_f32: the fp32 product needs fp32 or fp16 entries");
}

// ---- test / measurement entry (tests/test_gpu_mv.py, scripts): the product alone on a pmh_csr ---------------------------------------------------------------
extern "C" int pmh_mv_test_spmv(pmh_csr A, int storage, const double *x /* 3 nbr R, device */, double *y, int repeats, float *ms_per_launch)
{
  PMH_ARG(A && x && y && repeats >= 1);
  pmh_mv_ell E = nullptr;
  PMH_CHK(pmh_mv_ell_create(A, storage, &E));
  if (!E) return pmh_set_error(PMH_ERR_SUP, "pmh_mv_test_spmv: no regular 3 x 3 block structure");
  pmh_ctx     ctx = A->ctx;
  const size_t n  = (size_t)A->nrows * PMH_MV_R;
  int          rc = PMH_SUCCESS;
  hipEvent_t   e0, e1;
  PMH_HIP(hipEventCreate(&e0));
  PMH_HIP(hipEventCreate(&e1));
  if (storage == PMH_BSR_F64) {
    rc = pmh_mv_spmv_f64(E, x, y, PMH_EPI_NONE, nullptr, nullptr);
    PMH_HIP(hipEventRecord(e0, ctx->stream));
    for (int k = 0; k < repeats && !rc; k++) rc = pmh_mv_spmv_f64(E, x, y, PMH_EPI_NONE, nullptr, nullptr);
    PMH_HIP(hipEventRecord(e1, ctx->stream));
  } else { // fp32 vectors: converted on the host side of this test entry
    std::vector<double> hx(n);
    std::vector<float>  fx(n), fy(n);
    float              *dx, *dy;
    PMH_CHK(pmh_memcpy_d2h(ctx, hx.data(), x, sizeof(double) * n));
    for (size_t i = 0; i < n; i++) fx[i] = (float)hx[i];
    PMH_CHK(pmh_malloc(ctx, sizeof(float) * n, (void **)&dx));
    PMH_CHK(pmh_malloc(ctx, sizeof(float) * n, (void **)&dy));
    PMH_CHK(pmh_memcpy_h2d(ctx, dx, fx.data(), sizeof(float) * n));
    rc = pmh_mv_spmv_f32(E, dx, dy, PMH_EPI_NONE, nullptr, nullptr);
    PMH_HIP(hipEventRecord(e0, ctx->stream));
    for (int k = 0; k < repeats && !rc; k++) rc = pmh_mv_spmv_f32(E, dx, dy, PMH_EPI_NONE, nullptr, nullptr);
    PMH_HIP(hipEventRecord(e1, ctx->stream));
    if (!rc) rc = pmh_memcpy_d2h(ctx, fy.data(), dy, sizeof(float) * n);
    for (size_t i = 0; i < n; i++) hx[i] = (double)fy[i];
    if (!rc) rc = pmh_memcpy_h2d(ctx, y, hx.data(), sizeof(double) * n);
    pmh_free(ctx, dx), pmh_free(ctx, dy);
  }
  PMH_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  PMH_HIP(hipEventElapsedTime(&ms, e0, e1));
  if (ms_per_launch) *ms_per_launch = ms / repeats;
  (void)hipEventDestroy(e0), (void)hipEventDestroy(e1);
  pmh_mv_ell_destroy(E);
  return rc;
}
