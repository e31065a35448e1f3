// Multi-right-hand-side K^+, part 2: the V-cycle of mg.hip on interleaved multivectors of R = PMH_MV_R columns (see mv_internal.h).
//
// The hierarchy is the one pmh_mg holds (level operators as CSR, node-wise transfer operators, Jacobi scaling, Chebyshev constants, dense coarse
// pseudo-inverses): nothing is set up twice but the ELL copies of the level operators (built on the device, mv.hip) and the work multivectors.  The cycle is
// the fused form of mg.hip (mg_level_fused): degree-2 Chebyshev/Jacobi smoothing finished inside the operator kernel, fp32 vectors, level operators in the
// precision pmh_mg keeps them in (fp16 on the finest levels by default).  Per smoothed level: d0 | PRE | SUB | restrict (+ the coarse d0) | ... |
// prolong-subtract | POST1 | POST2.  The transfers are node-wise where P = P_node (x) I_3 (the box hierarchies) and scalar CSR otherwise (smoothed aggregation).  A
// hierarchy of another shape (fp64 cycle, other degree, a level without 3 x 3 blocks) is refused (PMH_EPI_UNSUPPORTED, no error recorded): the caller keeps the
// one-column solver.
#include "mg_internal.h"
#include "mv_internal.h"

#define MV_R PMH_MV_R
typedef float mvg_flt4 __attribute__((ext_vector_type(4)));
static_assert(MV_R == 8, "k_mvg_restrict_s: 8 entry groups x 8 columns per wavefront");

struct mg_mv_level {
  pmh_mv_ell E = nullptr;
  pmh_mv_ell EP = nullptr, ER = nullptr; // a prolongation that is not node-wise as 3 x 3 blocks: -P and P' (fp32 entries), else NULL (node-wise / scalar kernels)
  float     *x = nullptr, *b = nullptr, *r = nullptr, *d = nullptr, *t = nullptr, *xa = nullptr;
};
struct pmh_mg_mv_s {
  pmh_mg                   mg;
  pmh_ctx                  ctx; // (kept: the caller may destroy the pmh_mg before this object)
  // > 1: every level is block diagonal with nrep congruent blocks and this object works on the FIRST one (a prefix of every array of pmh_mg)
  int                      nrep = 1;
  std::vector<mg_mv_level> L;
};

// max |pinv_b - pinv_0| over the coarse blocks b > 0 (fp16 or fp32 entries, equal sizes): the coarse inverses of congruent blocks must be equal too
template <typename TP> __global__ __launch_bounds__(PMH_BLOCK) void k_mvg_pinv_diff(int nb, long long m2, const TP *__restrict__ pinv,
                        int *__restrict__ differs)
{
  for (long long i = (long long)blockIdx.x * PMH_BLOCK + threadIdx.x; i < m2; i += (long long)gridDim.x * PMH_BLOCK)
    for (int b = 1; b < nb; b++)
      if ((float)pinv[(long long)b * m2 + i] != (float)pinv[i]) *differs = 1;
}

// d0 = D^-1 b / theta (and the fp32 copy of an fp64 b): one thread per 4 consecutive entries of a row's R columns
template <typename TB>
__global__ __launch_bounds__(PMH_BLOCK) void k_mvg_d0(long long nR, const int *__restrict__ halt, const float *__restrict__ dinv, const TB *__restrict__ b,
                        float itheta, float *__restrict__ d, float *__restrict__ bcopy)
{
  if (halt && *halt) return;
  for (long long i = 4 * ((long long)blockIdx.x * PMH_BLOCK + threadIdx.x); i < nR; i += 4LL * gridDim.x * PMH_BLOCK) {
    const float di = dinv[i / MV_R] * itheta;
    float       v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = (float)b[i + k];
    if (bcopy) *(mvg_flt4 *)(bcopy + i) = mvg_flt4{v[0], v[1], v[2], v[3]};
    *(mvg_flt4 *)(d + i) = mvg_flt4{di * v[0], di * v[1], di * v[2], di * v[3]};
  }
}

// b_c = P' t, node-wise P' (<= 27 entries per coarse node), every entry serving the 3 R values of its fine node: lane (coarse node, column r); optionally the
// coarse level's first smoothing direction d_c = D_c^-1 b_c / theta_c
__global__ __launch_bounds__(PMH_BLOCK) void k_mvg_restrict(int ncn, const int *__restrict__ halt, const int *__restrict__ rowptr, const int *__restrict__ col,
                        const float *__restrict__ val, const float *__restrict__ t,
                                                           float *__restrict__ bc, const float *__restrict__ dinv_c, float itheta_c, float *__restrict__ d_c)
{
  // 32 lanes per coarse node: 4 entry groups x 8 columns, the groups summed by two butterfly steps (a fixed order).  (Until round 6 one lane per (node, column) walked the up
  // to 27 entries alone: 14 us per launch on every level of the 43^3 hierarchy -- three of them per cycle, 10 % of an inner-Krylov step.)
  if (halt && *halt) return;
  const int lane = threadIdx.x & 31, r = lane % MV_R, g = lane / MV_R;
  for (int i = blockIdx.x * (PMH_BLOCK / 32) + (threadIdx.x >> 5); i < ncn; i += gridDim.x * (PMH_BLOCK / 32)) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int k = rowptr[i] + g; k < rowptr[i + 1]; k += 4) {
      const float  w = val[k];
      const float *p = t + (size_t)3 * col[k] * MV_R + r;
      s0 += w * p[0], s1 += w * p[MV_R], s2 += w * p[2 * MV_R];
    }
#pragma unroll
    for (int o = 8; o < 32; o <<= 1) s0 += __shfl_xor(s0, o, 32), s1 += __shfl_xor(s1, o, 32), s2 += __shfl_xor(s2, o, 32);
    if (g == 0) {
      float *o = bc + (size_t)3 * i * MV_R + r;
      o[0] = s0, o[MV_R] = s1, o[2 * MV_R] = s2;
      if (dinv_c) {
        float *dd = d_c + (size_t)3 * i * MV_R + r;
        dd[0] = dinv_c[3 * i] * s0 * itheta_c, dd[MV_R] = dinv_c[3 * i + 1] * s1 * itheta_c, dd[2 * MV_R] = dinv_c[3 * i + 2] * s2 * itheta_c;
      }
    }
  }
}

// x -= P x_c, node-wise P (<= 8 entries per fine node): lane (fine node, column r)
__global__ __launch_bounds__(PMH_BLOCK) void k_mvg_prolong_sub(int nn, const int *__restrict__ halt, const int *__restrict__ rowptr,
                        const int *__restrict__ col, const float *__restrict__ val, const float *__restrict__ xc, float *__restrict__ x)
{
  if (halt && *halt) return;
  const int r = threadIdx.x % MV_R;
  for (int i = blockIdx.x * (PMH_BLOCK / MV_R) + threadIdx.x / MV_R; i < nn; i += gridDim.x * (PMH_BLOCK / MV_R)) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int k = rowptr[i]; k < rowptr[i + 1]; k++) {
      const float  w = val[k];
      const float *p = xc + (size_t)3 * col[k] * MV_R + r;
      s0 += w * p[0], s1 += w * p[MV_R], s2 += w * p[2 * MV_R];
    }
    float *xi = x + (size_t)3 * i * MV_R + r;
    xi[0] -= s0, xi[MV_R] -= s1, xi[2 * MV_R] -= s2;
  }
}

// The same two transfers for a prolongation that is NOT node-wise (smoothed aggregation, mgsa.hip: a fine dof interpolates from the m dofs of several aggregates): the scalar
// CSR of P' / P (values in the cycle's precision), lane (row, column r); the operand of an entry is the R-vector of ONE dof (32 contiguous bytes per 8 lanes).
__global__ __launch_bounds__(PMH_BLOCK) void k_mvg_restrict_s(int nc, const int *__restrict__ halt, const int *__restrict__ rowptr, const int *__restrict__ col,
                        const float *__restrict__ val, const float *__restrict__ t, float *__restrict__ bc, const float *__restrict__ dinv_c, float itheta_c,
                        float *__restrict__ d_c)
{
  // a row of P' has hundreds of entries (the dofs of an aggregate and of its neighbours' rims): ONE wavefront per coarse row, lane (entry group g, column r), the 8 groups
  // summed by three butterfly steps (a fixed order) -- one lane per (row, column) walked the row alone: 330 us per launch where the operator products take 55
  if (halt && *halt) return;
  const int lane = threadIdx.x & 63, r = lane % MV_R, g = lane / MV_R;
  for (int i = blockIdx.x * (PMH_BLOCK / 64) + (threadIdx.x >> 6); i < nc; i += gridDim.x * (PMH_BLOCK / 64)) {
    float s = 0.f;
    for (int k = rowptr[i] + g; k < rowptr[i + 1]; k += 64 / MV_R) s += val[k] * t[(size_t)col[k] * MV_R + r];
    s += __shfl_xor(s, 8, 64);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (g == 0) {
      bc[(size_t)i * MV_R + r] = s;
      if (dinv_c) d_c[(size_t)i * MV_R + r] = dinv_c[i] * s * itheta_c;
    }
  }
}
__global__ __launch_bounds__(PMH_BLOCK) void k_mvg_prolong_sub_s(int n, const int *__restrict__ halt, const int *__restrict__ rowptr, const int *__restrict__ col,
                        const float *__restrict__ val, const float *__restrict__ xc, float *__restrict__ x)
{
  if (halt && *halt) return;
  const int r = threadIdx.x % MV_R;
  for (int i = blockIdx.x * (PMH_BLOCK / MV_R) + threadIdx.x / MV_R; i < n; i += gridDim.x * (PMH_BLOCK / MV_R)) {
    float s = 0.f;
    for (int k = rowptr[i]; k < rowptr[i + 1]; k++) s += val[k] * xc[(size_t)col[k] * MV_R + r];
    x[(size_t)i * MV_R + r] -= s;
  }
}

// coarsest level: X_b = pinv_b B_b for the R columns: one wavefront per row of the dense block, lanes stride the row, R sums per lane
template <typename TP>
__global__ __launch_bounds__(PMH_BLOCK) void k_mvg_coarse(int nb, int n, const int *__restrict__ halt, const int *__restrict__ rs,
                        const long long *__restrict__ ofs, const TP *__restrict__ pinv, float scale, const float *__restrict__ b,
                                                         float *__restrict__ x)
{
  if (halt && *halt) return;
  const int lane = threadIdx.x & 63;
  const int row  = blockIdx.x * (PMH_BLOCK / 64) + (threadIdx.x >> 6);
  if (row >= n) return;
  int lo = 0, hi = nb;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (rs[mid] <= row) lo = mid;
    else hi = mid;
  }
  const int    r0 = rs[lo], m = rs[lo + 1] - r0;
  const TP    *a  = pinv + ofs[lo] + (size_t)(row - r0) * m;
  const float *bb = b + (size_t)r0 * MV_R;
  float        s[MV_R];
#pragma unroll
  for (int r = 0; r < MV_R; r++) s[r] = 0.f;
  for (int j = lane; j < m; j += 64) {
    const float    w  = (float)a[j];
    const mvg_flt4 v0 = *(const mvg_flt4 *)(bb + (size_t)j * MV_R), v1 = *(const mvg_flt4 *)(bb + (size_t)j * MV_R + 4);
    s[0] += w * v0.x, s[1] += w * v0.y, s[2] += w * v0.z, s[3] += w * v0.w;
    s[4] += w * v1.x, s[5] += w * v1.y, s[6] += w * v1.z, s[7] += w * v1.w;
  }
#pragma unroll
  for (int r = 0; r < MV_R; r++)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s[r] += __shfl_down(s[r], o, 64);
  if (lane == 0) {
    float *xo = x + (size_t)row * MV_R;
#pragma unroll
    for (int r = 0; r < MV_R; r++) xo[r] = (sizeof(TP) == 2) ? s[r] * scale : s[r];
  }
}

// The same on the matrix cores (round 6), for the LARGE dense coarse blocks of the one-block regime (5 184 rows on the 43^3 cube: 54 MB of fp16 entries per solve).  The form
// above reads the 32 bytes of B's row j once per ROW of pinv -- 860 MB through the L1s per solve, 35.6 us (1.5 TB/s on the matrix) -- and sums over k with 48 butterfly steps
// per row.  Here a workgroup of MVG_CW wavefronts owns 16 rows, wavefront w the steps [w per, (w+1) per) of 32 k's: lane (i, q) = (l & 15, l >> 4) loads the 8 entries
// pinv[row0 + i][k0 + 8 q ... + 8) in one 16-byte load and feeds them to 8 v_mfma_f32_16x16x4_f32 (exact fp32 FMAs; A[i][k = q] / B[k = q][j = l & 15], k taken as 8 q + e for
// the e-th one: any assignment of k's to the 4 k-slots is a permutation of the sum) against B[k0 + 8 q + e][j & 7] -- the 8 columns twice: half of the instruction's N = 16 is
// idle.  The sum over k happens inside the instruction; the wavefronts' 16 x 8 partials are added through LDS in wave order (fixed).  21.4 us = 2.5 TB/s on the matrix: 324
// workgroups on 256 CUs leave 68 CUs with two.  (The 4x4x1 16-block form -- no idle half, 8- or 4-row workgroups -- was built too: 18.7 - 20.1 us, removed again;
// docs/LAB_NOTEBOOK.md "Round 6" 5, lane maps in scripts/micro/mfma_f32_4x4.hip.)
// Needs every block's size to be a multiple of 16 (rows of a workgroup in ONE block, 16-byte aligned rows); else the form above.
typedef float    mvg_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 mvg_h8 __attribute__((ext_vector_type(8)));
template <int MVG_CU> struct mvg_cstage {
  mvg_h8   a[MVG_CU];
  mvg_flt4 b[MVG_CU];
};
template <int MVG_CW, int MVG_CU> __global__ __launch_bounds__(64 * MVG_CW) void k_mvg_coarse_mfma(int nb, const int *__restrict__ halt, const int *__restrict__ rs, const long long *__restrict__ ofs,
                        const _Float16 *__restrict__ pinv, float scale, const float *__restrict__ b, float *__restrict__ x)
{
  if (halt && *halt) return;
  __shared__ float part[MVG_CW][16][MV_R];
  __shared__ float tr[MVG_CW][MVG_CU][4 * 8 * 8]; // per wavefront and step: the 32 x 8 entries of B, [q][column][k & 7]
  const int        row0 = blockIdx.x * 16;
  int              lo = 0, hi = nb;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (rs[mid] <= row0) lo = mid;
    else hi = mid;
  }
  const int       r0 = rs[lo], m = rs[lo + 1] - r0;
  const int       lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, q = lane >> 4, jr = i & 7, h = i >> 3;
  const _Float16 *a  = pinv + ofs[lo] + (size_t)(row0 - r0 + i) * m + 8 * q;
  // B[k0 + 8 q ... + 8)[0 ... 8) is 256 contiguous bytes: lane (jr, h) of the 16 with this q brings 16 of them (row jr, columns 4 h ... 4 h + 3) -- ONE coalesced load per
  // step where every lane fetching its own 8 operands (column jr of the 8 rows) took 8 and left the kernel bound by the address rate of those (26 us)
  const float *bb = b + ((size_t)r0 + 8 * q + jr) * MV_R + 4 * h;
  const int    nsteps = (m + 31) / 32, per = (nsteps + MVG_CW - 1) / MVG_CW;
  const int    s0 = wave * per, s1 = min(nsteps, s0 + per);
  auto         load = [&](mvg_cstage<MVG_CU> &S, int s) {
#pragma unroll
    for (int u = 0; u < MVG_CU; u++) {
      const int  k0 = 32 * (s + u);
      const bool ok = (s + u < s1) && (k0 + 8 * q < m); // (m % 8 == 0: the 8 entries are in the row or not)
      if (ok) {
        S.a[u] = *(const mvg_h8 *)(a + k0);
        S.b[u] = *(const mvg_flt4 *)(bb + (size_t)k0 * MV_R);
      } else {
        S.a[u] = mvg_h8{0, 0, 0, 0, 0, 0, 0, 0};
        S.b[u] = mvg_flt4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  mvg_f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f}; // two chains: the instruction's dependent latency (40 cycles) is above its issue time (32)
  auto      mults = [&](const mvg_cstage<MVG_CU> &S) {
    // the 8 x 8 blocks transposed through the wavefront's own LDS lines (its LDS operations execute in order: no barrier, the compiler is only kept from reordering them)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < MVG_CU; u++) {
      float *t = &tr[wave][u][(q * 8 + 4 * h) * 8 + jr];
      t[0] = S.b[u].x, t[8] = S.b[u].y, t[16] = S.b[u].z, t[24] = S.b[u].w;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int u = 0; u < MVG_CU; u++) {
      const mvg_flt4 v0 = *(const mvg_flt4 *)&tr[wave][u][(q * 8 + jr) * 8], v1 = *(const mvg_flt4 *)&tr[wave][u][(q * 8 + jr) * 8 + 4];
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32((float)S.a[u][0], v0.x, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32((float)S.a[u][1], v0.y, c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32((float)S.a[u][2], v0.z, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32((float)S.a[u][3], v0.w, c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32((float)S.a[u][4], v1.x, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32((float)S.a[u][5], v1.y, c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32((float)S.a[u][6], v1.z, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32((float)S.a[u][7], v1.w, c1, 0, 0, 0);
    }
  };
  mvg_cstage<MVG_CU> S0, S1;
  load(S0, s0);
  for (int s = s0; s < s1; s += 2 * MVG_CU) { // (the loads of the next stage are in flight while this one's products issue)
    load(S1, s + MVG_CU);
    mults(S0);
    load(S0, s + 2 * MVG_CU);
    mults(S1);
  }
  // C[row = 4 q + v][col = i]: columns 8 ... 15 repeat 0 ... 7
  if (i < MV_R) {
#pragma unroll
    for (int v = 0; v < 4; v++) part[wave][4 * q + v][i] = c0[v] + c1[v];
  }
  __syncthreads();
  if (threadIdx.x < 16 * MV_R) {
    const int rr = threadIdx.x / MV_R, cc = threadIdx.x % MV_R;
    float     sum = part[0][rr][cc];
#pragma unroll
    for (int w = 1; w < MVG_CW; w++) sum += part[w][rr][cc];
    x[(size_t)(row0 + rr) * MV_R + cc] = sum * scale;
  }
}

static inline dim3 mvg_grid(long long work_items)
{
  long long g = (work_items + PMH_BLOCK - 1) / PMH_BLOCK;
  return dim3((unsigned)(g < 1 ? 1 : (g > PMH_MAX_VEC_BLOCKS ? PMH_MAX_VEC_BLOCKS : g)));
}

int pmh_mg_mv_destroy(pmh_mg_mv M)
{
  if (!M) return PMH_SUCCESS;
  pmh_ctx ctx = M->ctx;
  for (auto &l : M->L) {
    pmh_mv_ell_destroy(l.E), pmh_mv_ell_destroy(l.EP), pmh_mv_ell_destroy(l.ER);
    pmh_free(ctx, l.x), pmh_free(ctx, l.b), pmh_free(ctx, l.r), pmh_free(ctx, l.d), pmh_free(ctx, l.t), pmh_free(ctx, l.xa);
  }
  delete M;
  return PMH_SUCCESS;
}

int pmh_mg_mv_create(pmh_mg mg, pmh_mg_mv *out, int nrep)
{
  PMH_ARG(mg && out && nrep >= 1);
  *out = nullptr;
  if (nrep > 1) { // congruence of every level as pmh_bsr3_from_csr verified it, equal coarse inverses
    for (int l = 0; l + 1 < mg->nlevels; l++)
      if (!mg->L[l].Ab || mg->L[l].Ab->nrep != nrep) {
        pmh_mv_set_why("the blocks are not congruent on every level of the hierarchy");
        return PMH_EPI_UNSUPPORTED;
      }
    const int nc = mg->L[mg->nlevels - 1].n;
    if (mg->nb_coarse != nrep || nc % nrep) {
      pmh_mv_set_why("the coarse level does not have one block per congruent block");
      return PMH_EPI_UNSUPPORTED;
    }
    int *d_diff, h_diff = 0;
    PMH_CHK(pmh_malloc(mg->ctx, sizeof(int), (void **)&d_diff));
    PMH_HIP(hipMemsetAsync(d_diff, 0, sizeof(int), mg->ctx->stream));
    const long long m2 = (long long)(nc / nrep) * (nc / nrep);
    if (mg->cp_half) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mvg_pinv_diff<_Float16>), dim3(256), dim3(PMH_BLOCK), 0, mg->ctx->stream, nrep, m2,
                            (const _Float16 *)mg->d_cpinv, d_diff);
    else if (mg->is_float) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mvg_pinv_diff<float>), dim3(256), dim3(PMH_BLOCK), 0, mg->ctx->stream, nrep, m2,
                            (const float *)mg->d_cpinv, d_diff);
    PMH_CHK(pmh_memcpy_d2h(mg->ctx, &h_diff, d_diff, sizeof(int)));
    pmh_free(mg->ctx, d_diff);
    if (h_diff) {
      pmh_mv_set_why("the coarse pseudo-inverses of the congruent blocks differ");
      return PMH_EPI_UNSUPPORTED;
    }
  }
  if (!(mg->is_float && mg->fused && mg->nlevels > 1 && mg->degree == 2)) {
    pmh_mv_set_why(!mg->is_float ? "the V-cycle runs in fp64" : (mg->degree != 2 ? "the smoother is not of degree 2" : (mg->nlevels < 2 ? "the hierarchy has one level" : "the V-cycle is not the fused form (PMH_MG_FUSED=0 or a level without 3 x 3 blocks)")));
    return PMH_EPI_UNSUPPORTED;
  }
  for (int l = 0; l + 1 < mg->nlevels; l++) {
    const bool nodal = mg->L[l].pn_rowptr && mg->L[l].rn_rowptr;
    if (!mg->L[l].Ab || mg->L[l].n % 3 || (!nodal && (nrep > 1 || !mg->L[l].pv_owned))) { // (a scalar P of congruent blocks: its prefix is not addressed separately here)
      pmh_mv_set_why(!mg->L[l].Ab ? "a level operator has no 3 x 3 block copy" : "a prolongation that is not node-wise (P = P_node (x) I_3) on congruent blocks");
      return PMH_EPI_UNSUPPORTED;
    }
  }
  pmh_ctx   ctx = mg->ctx;
  pmh_mg_mv M   = new pmh_mg_mv_s();
  M->mg = mg, M->ctx = ctx, M->nrep = nrep;
  M->L.resize(mg->nlevels);
  int rc = PMH_SUCCESS;
  for (int l = 0; l < mg->nlevels && !rc; l++) {
    mg_level    &Lv = mg->L[l];
    mg_mv_level &Ml = M->L[l];
    const size_t nR = (size_t)(Lv.n / nrep) * MV_R;
    if (l + 1 < mg->nlevels) {
      rc = pmh_mv_ell_create_prefix(Lv.A, nrep, Lv.Ab->storage == PMH_BSR_F64 ? PMH_BSR_F32 : Lv.Ab->storage, &Ml.E);
      if (!rc && !Ml.E) {
        pmh_mg_mv_destroy(M);
        pmh_mv_set_why("a level operator has rows with unsorted columns or more than 2048 blocks of 3 x 3 in a block row");
        return PMH_EPI_UNSUPPORTED;
      }
      if (!rc && !(Lv.pn_rowptr && Lv.rn_rowptr) && Lv.P->ncols % 3 == 0) { // aggregation hierarchy: P couples a fine node to the 6 dofs of a few aggregates -- two 3 x 3 blocks each
        rc = pmh_mv_ell_create_rect(Lv.P, PMH_BSR_F32, 1, &Ml.EP);
        if (!rc && Ml.EP) rc = pmh_mv_ell_create_rect(Lv.P->transpose, PMH_BSR_F32, 0, &Ml.ER);
        if (!rc && !Ml.ER) pmh_mv_ell_destroy(Ml.EP), Ml.EP = nullptr; // (both or none: the scalar kernels otherwise)
      }
      for (float **v : {&Ml.r, &Ml.d, &Ml.t, &Ml.xa})
        if (!rc) rc = pmh_malloc(ctx, sizeof(float) * nR, (void **)v);
    }
    for (float **v : {&Ml.x, &Ml.b})
      if (!rc) rc = pmh_malloc(ctx, sizeof(float) * nR, (void **)v);
  }
  if (rc) {
    pmh_mg_mv_destroy(M);
    return rc;
  }
  *out = M;
  return PMH_SUCCESS;
}

static int mvg_cycle(pmh_mg_mv M, int l, const double *b64, double *z64, bool d0_ready, const int *halt)
{
  pmh_mg       mg = M->mg;
  mg_level    &Lv = mg->L[l];
  mg_mv_level &Ml = M->L[l];
  hipStream_t  st = mg->ctx->stream;
  const dim3   blk(PMH_BLOCK);
  const int    nrep = M->nrep, n_l = Lv.n / nrep; // (congruent blocks: the first block's share of every level)
  if (l == mg->nlevels - 1) {
    const int nbc = mg->nb_coarse / nrep;
    if (mg->cp_half && mg->coarse_m16)
      // (16 wavefronts x 2 steps per stage: 21.4 us on the 5 184-row block; 8 x 3: 23.2, 8 x 6: 26.9, 16 x 3: 25.3, 4 x 6: 26.6)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mvg_coarse_mfma<16, 2>), dim3(n_l / 16), dim3(64 * 16), 0, st, nbc, halt, (const int *)mg->d_crs, (const long long *)mg->d_cofs,
                         (const _Float16 *)mg->d_cpinv, (float)mg->cp_scale, (const float *)Ml.b, Ml.x);
    else if (mg->cp_half)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mvg_coarse<_Float16>), dim3((n_l + 3) / 4), blk, 0, st, nbc, n_l, halt, (const int *)mg->d_crs,
                         (const long long *)mg->d_cofs, (const _Float16 *)mg->d_cpinv, (float)mg->cp_scale,
                         (const float *)Ml.b, Ml.x);
    else
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mvg_coarse<float>), dim3((n_l + 3) / 4), blk, 0, st, nbc, n_l, halt, (const int *)mg->d_crs,
                         (const long long *)mg->d_cofs, (const float *)mg->d_cpinv, 1.f, (const float *)Ml.b, Ml.x);
    PMH_HIP(hipGetLastError());
    return PMH_SUCCESS;
  }
  mg_level    &Lc = mg->L[l + 1];
  mg_mv_level &Mc = M->L[l + 1];
  const long long nR     = (long long)n_l * MV_R;
  const float    *dinv   = (const float *)Lv.dinv;
  const float     itheta = (float)(1.0 / Lv.theta), c1 = (float)Lv.c1[1], c2 = (float)Lv.c2[1];
  if (!d0_ready) {
    if (b64) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mvg_d0<double>), mvg_grid(nR / 4), blk, 0, st, nR, halt, dinv, b64, itheta, Ml.d, Ml.b);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mvg_d0<float>), mvg_grid(nR / 4), blk, 0, st, nR, halt, dinv, (const float *)Ml.b, itheta, Ml.d,
                            (float *)nullptr);
  }
  pmh_mv_epi<float> e;
  memset(&e, 0, sizeof(e));
  e.y1 = Ml.b, e.dinv = dinv, e.r = Ml.r, e.d = Ml.d;
  e.c0 = 1.f + c1, e.c1 = c1, e.c2 = c2;
  PMH_CHK(pmh_mv_spmv_f32(Ml.E, Ml.d, Ml.xa, PMH_BSR_EPI_PRE, &e, halt));
  PMH_CHK(pmh_mv_spmv_f32(Ml.E, Ml.xa, Ml.t, PMH_EPI_SUB, &e, halt));
  const bool cf  = l + 2 < mg->nlevels; // the coarse level is a smoothed one: its d0 rides on the restriction
  const int  ncn = Lc.n / 3 / nrep;
  const bool nodal = Lv.rn_rowptr != nullptr;
  if (Ml.ER) {
    pmh_mv_epi<float> er;
    memset(&er, 0, sizeof(er));
    if (cf) er.dinv = (const float *)Lc.dinv, er.d = Mc.d, er.c0 = (float)(1.0 / Lc.theta);
    PMH_CHK(pmh_mv_spmv_f32(Ml.ER, Ml.t, Mc.b, PMH_MV_EPI_RESTRICT, &er, halt));
  } else if (nodal)
    hipLaunchKernelGGL(k_mvg_restrict, mvg_grid((long long)ncn * 32), blk, 0, st, ncn, halt, (const int *)Lv.rn_rowptr, (const int *)Lv.rn_col,
                       (const float *)Lv.rn_val, (const float *)Ml.t, Mc.b,
                       cf ? (const float *)Lc.dinv : (const float *)nullptr, cf ? (float)(1.0 / Lc.theta) : 0.f, cf ? Mc.d : (float *)nullptr);
  else
    hipLaunchKernelGGL(k_mvg_restrict_s, mvg_grid((long long)Lc.n * 64), blk, 0, st, Lc.n, halt, (const int *)Lv.P->transpose->d_rowptr, (const int *)Lv.P->transpose->d_col,
                       (const float *)Lv.rv, (const float *)Ml.t, Mc.b,
                       cf ? (const float *)Lc.dinv : (const float *)nullptr, cf ? (float)(1.0 / Lc.theta) : 0.f, cf ? Mc.d : (float *)nullptr);
  PMH_CHK(mvg_cycle(M, l + 1, nullptr, nullptr, cf, halt));
  if (Ml.EP) {
    pmh_mv_epi<float> ep;
    memset(&ep, 0, sizeof(ep));
    ep.y1 = Ml.xa;
    PMH_CHK(pmh_mv_spmv_f32(Ml.EP, Mc.x, Ml.xa, PMH_EPI_ADD, &ep, halt)); // xa += (-P) x_c
  } else if (nodal)
    hipLaunchKernelGGL(k_mvg_prolong_sub, mvg_grid((long long)(n_l / 3) * MV_R), blk, 0, st, n_l / 3, halt, (const int *)Lv.pn_rowptr, (const int *)Lv.pn_col,
                       (const float *)Lv.pn_val, (const float *)Mc.x, Ml.xa);
  else
    hipLaunchKernelGGL(k_mvg_prolong_sub_s, mvg_grid((long long)n_l * MV_R), blk, 0, st, n_l, halt, (const int *)Lv.P->d_rowptr, (const int *)Lv.P->d_col,
                       (const float *)Lv.pv, (const float *)Mc.x, Ml.xa);
  PMH_HIP(hipGetLastError());
  e.c0 = itheta;
  PMH_CHK(pmh_mv_spmv_f32(Ml.E, Ml.xa, Ml.x, PMH_BSR_EPI_POST1, &e, halt));
  e.z64 = z64;
  return pmh_mv_spmv_f32(Ml.E, Ml.d, Ml.x, PMH_BSR_EPI_POST2, &e, halt);
}

// Z = V(B) for R columns: b, z are fp64 multivectors of n_0 R entries
int pmh_mg_mv_apply(pmh_mg_mv M, const double *b, double *z, const int *halt)
{
  PMH_ARG(M && b);
  return mvg_cycle(M, 0, b, z, false, halt);
}
const float *pmh_mg_mv_result(pmh_mg_mv M) { return M->L[0].x; }
