// QPPF projector factory on gfx950: Q = G'(GG')^{-1}G, P = I - Q (src/qppf/interface/qppf.c) and the two
// shell operators of the QP transform chain built on it: A + rho*G'G (src/qp/utils/matpenalized.c) and
// P*A*P / P*A (QPTEnforceEqByProjector, src/qp/interface/qptransform.c:215-316).
//
// G (m x n, m = 6 * #subdomains rigid-body rows, each dense over one subdomain's interface) is an explicit
// CSR in HBM; G v uses the long-row path of the SpMV (one workgroup per row), G' w the short-row stream
// path on the transposed CSR.  The coarse problem (GG')^{-1} is small and dense: GG' is assembled and
// inverted on the host once (the reference's QPPFSetUpGGt_Private / -qppf_explicit_inv path, qppf.c:213-333)
// and applied as a dense GEMV, redundantly on every GPU (mirrors -qppf_redundancy, qppf.c:182,305).
#include <chrono>

#include "pmh_internal.h"
#include "reduce.h"
#include "box_inline.h"
#include "dualchain.h"


// y = M x, M dense m x m row-major; one 64-lane wavefront per row
__global__ __launch_bounds__(PMH_BLOCK) void k_dense_gemv(int m, const double *__restrict__ M, const double *__restrict__ x, double *__restrict__ y)
{
  const int row = blockIdx.x * (PMH_BLOCK / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= m) return;
  double s = 0.0;
  for (int j = lane; j < m; j += 64) s += M[(size_t)row * m + j] * x[j];
  s = pmh_wave_sum(s);
  if (lane == 0) y[row] = s;
}

// ---- GG' assembly on the matrix cores (the one GEMM-shaped item of the path, SURVEY 2.3 K20: the reference builds the
// 6x6 blocks of GG' with BLASgemm("N","T"), extension.c:961) -----------------------------------------------------------
// G is densified TRANSPOSED (Gt: n x Mp row-major, Mp = m rounded up to 16) so that the MFMA operand loads are
// contiguous: for v_mfma_f64_16x16x4_f64 lane l holds A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; with C = G G'
// both come from row (k0 + (l>>4)) of Gt, columns i0 + (l&15) and j0 + (l&15).  C/D: col = l&15, row = (l>>4) + 4*reg.
// Every workgroup owns one 16x16 output tile and one K chunk; its 4 wavefronts split the chunk, partial tiles are summed
// in wave order through LDS, and the per-chunk partials are added in chunk order by k_ggt_reduce (deterministic).
typedef double v4f64 __attribute__((ext_vector_type(4)));
#define GGT_KCHUNK 8192

__global__ __launch_bounds__(PMH_BLOCK) void k_densify_gt(int m, int Mp, const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val, double *__restrict__ Gt)
{
  const int row = blockIdx.x;
  if (row >= m) return;
  for (int k = rowptr[row] + (int)threadIdx.x; k < rowptr[row + 1]; k += PMH_BLOCK) Gt[(size_t)col[k] * Mp + row] = val[k];
}

__global__ __launch_bounds__(PMH_BLOCK) void k_ggt_mfma(int n, int Mp, const double *__restrict__ Gt, double *__restrict__ part)
{
  __shared__ double lds[PMH_BLOCK / 64][256];
  const int         nt = Mp / 16, ti = blockIdx.x / nt, tj = blockIdx.x % nt, chunk = blockIdx.y;
  const int         lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int         k_lo = chunk * GGT_KCHUNK + wave * (GGT_KCHUNK / 4);
  const int         k_hi = min(n, k_lo + GGT_KCHUNK / 4);
  v4f64             acc  = {0.0, 0.0, 0.0, 0.0};
  const int         kk = lane >> 4, ij = lane & 15;
  for (int k = k_lo; k < k_hi; k += 4) {
    const int  kr = k + kk;
    const bool ok = kr < k_hi;
    const double a = ok ? Gt[(size_t)kr * Mp + ti * 16 + ij] : 0.0;
    const double b = ok ? Gt[(size_t)kr * Mp + tj * 16 + ij] : 0.0;
    acc            = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; r++) lds[wave][((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc[r];
  __syncthreads();
  const int t = threadIdx.x; // 256 threads = 256 tile entries
  double    v = lds[0][t];
#pragma unroll
  for (int w = 1; w < PMH_BLOCK / 64; w++) v += lds[w][t];
  part[((size_t)chunk * nt * nt + blockIdx.x) * 256 + t] = v;
}

__global__ __launch_bounds__(PMH_BLOCK) void k_ggt_reduce(int nchunks, int Mp, const double *__restrict__ part, double *__restrict__ GGt)
{
  const int nt = Mp / 16, tile = blockIdx.x, ti = tile / nt, tj = tile % nt, t = threadIdx.x;
  double    v = 0.0;
  for (int c = 0; c < nchunks; c++) v += part[((size_t)c * nt * nt + tile) * 256 + t];
  GGt[(size_t)(ti * 16 + t / 16) * Mp + tj * 16 + (t % 16)] = v;
}

// device GG' (m x m, row-major, returned on the host)
static int device_ggt(pmh_ctx ctx, pmh_csr G, std::vector<double> &ggt, double *mfma_ms)
{
  const int m = G->nrows, n = G->ncols, Mp = ((m + 15) / 16) * 16, nt = Mp / 16;
  const int nchunks = (n + GGT_KCHUNK - 1) / GGT_KCHUNK;
  double   *Gt = nullptr, *part = nullptr, *dggt = nullptr;
  PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)n * Mp, (void **)&Gt));
  PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)nchunks * nt * nt * 256, (void **)&part));
  PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)Mp * Mp, (void **)&dggt));
  PMH_CHK(pmh_memset(ctx, Gt, 0, sizeof(double) * (size_t)n * Mp));
  hipLaunchKernelGGL(k_densify_gt, dim3(m), dim3(PMH_BLOCK), 0, ctx->stream, m, Mp, (const int *)G->d_rowptr, (const int *)G->d_col, (const double *)G->d_val, Gt);
  PMH_HIP(hipEventRecord(ctx->ev0, ctx->stream));
  hipLaunchKernelGGL(k_ggt_mfma, dim3(nt * nt, nchunks), dim3(PMH_BLOCK), 0, ctx->stream, n, Mp, (const double *)Gt, part);
  hipLaunchKernelGGL(k_ggt_reduce, dim3(nt * nt), dim3(PMH_BLOCK), 0, ctx->stream, nchunks, Mp, (const double *)part, dggt);
  PMH_HIP(hipEventRecord(ctx->ev1, ctx->stream));
  PMH_HIP(hipGetLastError());
  PMH_HIP(hipEventSynchronize(ctx->ev1));
  float ms = 0.f;
  PMH_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  if (mfma_ms) *mfma_ms = ms;
  std::vector<double> full((size_t)Mp * Mp);
  PMH_CHK(pmh_memcpy_d2h(ctx, full.data(), dggt, sizeof(double) * full.size()));
  ggt.assign((size_t)m * m, 0.0);
  for (int i = 0; i < m; i++)
    for (int j = 0; j < m; j++) ggt[(size_t)i * m + j] = full[(size_t)i * Mp + j];
  pmh_free(ctx, Gt);
  pmh_free(ctx, part);
  pmh_free(ctx, dggt);
  return PMH_SUCCESS;
}

static int host_cholesky_inverse(int m, std::vector<double> &a)
{
  // in-place lower Cholesky, then inverse via forward/back substitution against the identity
  for (int j = 0; j < m; j++) {
    double d = a[(size_t)j * m + j];
    for (int k = 0; k < j; k++) d -= a[(size_t)j * m + k] * a[(size_t)j * m + k];
    if (!(d > 0.0)) return 1;
    d                    = sqrt(d);
    a[(size_t)j * m + j] = d;
    for (int i = j + 1; i < m; i++) {
      double s = a[(size_t)i * m + j];
      for (int k = 0; k < j; k++) s -= a[(size_t)i * m + k] * a[(size_t)j * m + k];
      a[(size_t)i * m + j] = s / d;
    }
  }
  std::vector<double> inv((size_t)m * m, 0.0), col(m);
  for (int c = 0; c < m; c++) {
    for (int i = 0; i < m; i++) {
      double s = (i == c) ? 1.0 : 0.0;
      for (int k = 0; k < i; k++) s -= a[(size_t)i * m + k] * col[k];
      col[i] = s / a[(size_t)i * m + i];
    }
    for (int i = m - 1; i >= 0; i--) {
      double s = col[i];
      for (int k = i + 1; k < m; k++) s -= a[(size_t)k * m + i] * col[k];
      col[i] = s / a[(size_t)i * m + i];
    }
    for (int i = 0; i < m; i++) inv[(size_t)i * m + c] = col[i];
  }
  a.swap(inv);
  return 0;
}

// a = L L' (lower Cholesky, in place); T = L^{-1} (lower triangular, row-major), S = T'T = a^{-1} (symmetrised by construction)
static int host_cholesky_T(int m, std::vector<double> &a, std::vector<double> &T, std::vector<double> &S)
{
  for (int j = 0; j < m; j++) {
    double d = a[(size_t)j * m + j];
    for (int k = 0; k < j; k++) d -= a[(size_t)j * m + k] * a[(size_t)j * m + k];
    if (!(d > 0.0)) return 1;
    d                    = sqrt(d);
    a[(size_t)j * m + j] = d;
    for (int i = j + 1; i < m; i++) {
      double s = a[(size_t)i * m + j];
      for (int k = 0; k < j; k++) s -= a[(size_t)i * m + k] * a[(size_t)j * m + k];
      a[(size_t)i * m + j] = s / d;
    }
  }
  T.assign((size_t)m * m, 0.0);
  for (int c = 0; c < m; c++) // column c of L^{-1} by forward substitution against e_c
    for (int i = c; i < m; i++) {
      double s = (i == c) ? 1.0 : 0.0;
      for (int k = c; k < i; k++) s -= a[(size_t)i * m + k] * T[(size_t)k * m + c];
      T[(size_t)i * m + c] = s / a[(size_t)i * m + i];
    }
  S.assign((size_t)m * m, 0.0);
  for (int i = 0; i < m; i++)
    for (int j = 0; j <= i; j++) {
      double s = 0.0;
      for (int k = i; k < m; k++) s += T[(size_t)k * m + i] * T[(size_t)k * m + j]; // (T'T)_ij = sum_k T_ki T_kj, k >= max(i, j) = i
      S[(size_t)i * m + j] = S[(size_t)j * m + i] = s;
    }
  return 0;
}

extern "C" int pmh_qppf_create(pmh_ctx ctx, pmh_csr G, int orthonormal, pmh_qppf *out)
{
  PMH_ARG(ctx && G && out);
  pmh_qppf pf    = new pmh_qppf_s();
  pf->ctx        = ctx;
  pf->G          = G;
  pf->m          = G->nrows;
  pf->n          = G->ncols;
  pf->orthonormal = orthonormal ? 1 : 0;
  pf->implicit_orth = 0;
  pf->d_Tt = pf->d_S = pf->tmp_m = nullptr;
  pf->d_inv      = nullptr;
  pf->ggt_mfma_ms = pf->host_inverse_ms = 0.0;
  const int m    = pf->m;
  PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)(m ? m : 1), (void **)&pf->G_left));
  PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)(m ? m : 1), (void **)&pf->Gt_right));
  if (!pf->orthonormal && m > 0) {
    // GG' on the device with fp64 MFMA (QPPFSetUpGGt_Private qppf.c:213-278); the small dense factorisation stays on the host
    std::vector<double> ggt;
    PMH_CHK(device_ggt(ctx, G, ggt, &pf->ggt_mfma_ms));
    const auto t0 = std::chrono::steady_clock::now();
    const int  bad = host_cholesky_inverse(m, ggt);
    pf->host_inverse_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (bad) {
      pmh_free(ctx, pf->G_left);
      pmh_free(ctx, pf->Gt_right);
      delete pf;
      return pmh_set_error(PMH_ERR_ARG, "pmh_qppf_create: G G' is not positive definite (G must have full row rank)");
    }
    PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)m * m, (void **)&pf->d_inv));
    PMH_CHK(pmh_memcpy_h2d(ctx, pf->d_inv, ggt.data(), sizeof(double) * (size_t)m * m));
  }
  if (orthonormal == 2 && m > 0) {
    // implicit orthonormalisation (the reference's -qp_E_orth_form implicit, qptransform.c:647, permonmatorth.c:176-205: the orthonormalised
    // matrix is never formed): G stays as sparse as it came (an explicit T G0 fills every row of a subdomain's modes with the columns of all
    // subdomains before it: 3.4 x the non-zeros for configs[2]); GG' = L L' on the device / host as above, T = L^{-1}, S = T'T = (GG')^{-1}
    std::vector<double> ggt;
    PMH_CHK(device_ggt(ctx, G, ggt, &pf->ggt_mfma_ms));
    const auto          t0 = std::chrono::steady_clock::now();
    std::vector<double> T, S;
    const int           bad = host_cholesky_T(m, ggt, T, S);
    pf->host_inverse_ms     = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (bad) {
      pmh_free(ctx, pf->G_left);
      pmh_free(ctx, pf->Gt_right);
      delete pf;
      return pmh_set_error(PMH_ERR_ARG, "pmh_qppf_create: G G' is not positive definite (G must have full row rank)");
    }
    std::vector<double> Tt((size_t)m * m);
    for (int i = 0; i < m; i++)
      for (int j = 0; j < m; j++) Tt[(size_t)j * m + i] = T[(size_t)i * m + j];
    const size_t bytes = sizeof(double) * (size_t)m * m;
    PMH_CHK(pmh_malloc(ctx, bytes, (void **)&pf->d_Tt));
    PMH_CHK(pmh_malloc(ctx, bytes, (void **)&pf->d_S));
    PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)m, (void **)&pf->tmp_m));
    PMH_CHK(pmh_memcpy_h2d(ctx, pf->d_Tt, Tt.data(), bytes));
    PMH_CHK(pmh_memcpy_h2d(ctx, pf->d_S, S.data(), bytes)); // symmetric: its own transpose
    pf->h_T           = T;
    pf->implicit_orth = 1;
  }
  *out = pf;
  return PMH_SUCCESS;
}

// e = T e0: the right-hand side of the implicitly orthonormalised constraint (T G0) lambda = T e0 (host vectors of length m)
extern "C" int pmh_qppf_orth_rhs(pmh_qppf pf, const double *e0, double *e)
{
  PMH_ARG(pf && e0 && e);
  if (!pf->implicit_orth) return pmh_set_error(PMH_ERR_STATE, "pmh_qppf_orth_rhs: the projector was not created with implicit orthonormalisation");
  const int m = pf->m;
  for (int i = 0; i < m; i++) {
    double s = 0.0;
    for (int j = 0; j <= i; j++) s += pf->h_T[(size_t)i * m + j] * e0[j];
    e[i] = s;
  }
  return PMH_SUCCESS;
}

// set-up cost of the coarse problem (SURVEY 8d asks for them separately): the GG' assembly on the matrix cores
// (k_ggt_mfma + its fixed-order reduction; 2 Mp^2 n flops with Mp = m rounded up to 16) and the host Cholesky + inverse
extern "C" int pmh_qppf_setup_stats(pmh_qppf pf, double *ggt_mfma_ms, double *ggt_flops, double *host_inverse_ms)
{
  PMH_ARG(pf);
  const double Mp = (double)(((pf->m + 15) / 16) * 16);
  if (ggt_mfma_ms) *ggt_mfma_ms = pf->ggt_mfma_ms;
  if (ggt_flops) *ggt_flops = pf->orthonormal ? 0.0 : 2.0 * Mp * Mp * (double)pf->n;
  if (host_inverse_ms) *host_inverse_ms = pf->host_inverse_ms;
  return PMH_SUCCESS;
}

extern "C" int pmh_qppf_destroy(pmh_qppf pf)
{
  if (!pf) return PMH_SUCCESS;
  pmh_free(pf->ctx, pf->G_left);
  pmh_free(pf->ctx, pf->Gt_right);
  if (pf->d_inv) pmh_free(pf->ctx, pf->d_inv);
  if (pf->d_Tt) pmh_free(pf->ctx, pf->d_Tt), pmh_free(pf->ctx, pf->d_S), pmh_free(pf->ctx, pf->tmp_m);
  delete pf;
  return PMH_SUCCESS;
}

extern "C" int pmh_qppf_apply_G(pmh_qppf pf, const double *v, double *Gv)
{
  PMH_ARG(pf);
  if (pf->m == 0) return PMH_SUCCESS;
  if (pf->implicit_orth) return pmh_csr_mult_then_dense(pf->G, v, pf->d_Tt, pf->tmp_m, Gv); // (T G0) v
  return pmh_csr_mult(pf->G, v, Gv);
}

// G v and ||G v||^2 (-> scalar slot, device + pinned host) without waiting: one launch pair where the implicit form with m <= 64 allows it
int pmh_qppf_apply_G_norm2(pmh_qppf pf, const double *v, double *Gv, int slot)
{
  if (pf->m == 0) return PMH_SUCCESS;
  if (pf->implicit_orth && pf->m <= 64) return pmh_csr_mult_then_dense(pf->G, v, pf->d_Tt, pf->tmp_m, Gv, slot);
  PMH_CHK(pmh_qppf_apply_G(pf, v, Gv));
  return pmh_k_dot_partials(pf->ctx, pf->m, Gv, Gv, slot);
}

// G_left = what G' is applied to in Q v = G'(..): G v for orthonormal rows, S G0 v under implicit orthonormalisation
static int qppf_left(pmh_qppf pf, const double *v)
{
  if (pf->implicit_orth) return pmh_csr_mult_then_dense(pf->G, v, pf->d_S, pf->tmp_m, pf->G_left);
  return pmh_csr_mult(pf->G, v, pf->G_left);
}

// QPPFApplyCP qppf.c:610-645
extern "C" int pmh_qppf_apply_CP(pmh_qppf pf, const double *x, double *y)
{
  PMH_ARG(pf);
  if (pf->m == 0) return PMH_SUCCESS;
  if (!pf->d_inv) return pmh_vec_copy(pf->ctx, pf->m, x, y);
  hipLaunchKernelGGL(k_dense_gemv, dim3((pf->m + 3) / 4), dim3(PMH_BLOCK), 0, pf->ctx->stream, pf->m, pf->d_inv, x, y);
  PMH_HIP(hipGetLastError());
  return PMH_SUCCESS;
}

// QPPFApplyQ qppf.c:454-503
extern "C" int pmh_qppf_apply_Q(pmh_qppf pf, const double *v, double *Qv)
{
  PMH_ARG(pf && (const void *)v != (const void *)Qv);
  if (pf->m == 0) return pmh_vec_set(pf->ctx, pf->n, Qv, 0.0);
  PMH_CHK(qppf_left(pf, v));
  if (pf->d_inv) {
    PMH_CHK(pmh_qppf_apply_CP(pf, pf->G_left, pf->Gt_right));
    return pmh_csr_mult_transpose(pf->G, pf->Gt_right, Qv);
  }
  return pmh_csr_mult_transpose(pf->G, pf->G_left, Qv);
}

// QPPFApplyP qppf.c:563-575: Pv = v - Qv  (VecAYPX(Pv,-1,v))
extern "C" int pmh_qppf_apply_P(pmh_qppf pf, const double *v, double *Pv)
{
  PMH_CHK(pmh_qppf_apply_Q(pf, v, Pv));
  return pmh_vec_aypx(pf->ctx, pf->n, Pv, -1.0, v);
}

// QPPFApplyGtG qppf.c:580-605
extern "C" int pmh_qppf_apply_GtG(pmh_qppf pf, const double *v, double *y)
{
  PMH_ARG(pf);
  if (pf->orthonormal) return pmh_qppf_apply_Q(pf, v, y);
  if (pf->m == 0) return pmh_vec_set(pf->ctx, pf->n, y, 0.0);
  PMH_CHK(pmh_csr_mult(pf->G, v, pf->G_left));
  return pmh_csr_mult_transpose(pf->G, pf->G_left, y);
}

// QPPFApplyHalfQ qppf.c:507-527: y = (GG')^{-1} G x  (length m)
extern "C" int pmh_qppf_apply_halfQ(pmh_qppf pf, const double *x, double *y)
{
  PMH_ARG(pf);
  if (pf->m == 0) return PMH_SUCCESS;
  if (pf->implicit_orth) return pmh_csr_mult_then_dense(pf->G, x, pf->d_Tt, pf->tmp_m, y); // (GG')^{-1} = I for G = T G0: y = T G0 x
  PMH_CHK(pmh_csr_mult(pf->G, x, pf->G_left));
  return pmh_qppf_apply_CP(pf, pf->G_left, y);
}

// QPPFApplyHalfQTranspose qppf.c:531-559: y = G'(GG')^{-1} x
extern "C" int pmh_qppf_apply_halfQ_transpose(pmh_qppf pf, const double *x, double *y)
{
  PMH_ARG(pf);
  if (pf->m == 0) return pmh_vec_set(pf->ctx, pf->n, y, 0.0);
  if (pf->implicit_orth) { // (T G0)' x = G0' (T' x); k_dense_gemv takes a row-major matrix: T' row-major = d_Tt
    hipLaunchKernelGGL(k_dense_gemv, dim3((pf->m + 3) / 4), dim3(PMH_BLOCK), 0, pf->ctx->stream, pf->m, (const double *)pf->d_Tt, x, pf->Gt_right);
    PMH_HIP(hipGetLastError());
    return pmh_csr_mult_transpose(pf->G, pf->Gt_right, y);
  }
  if (pf->d_inv) {
    PMH_CHK(pmh_qppf_apply_CP(pf, x, pf->Gt_right));
    return pmh_csr_mult_transpose(pf->G, pf->Gt_right, y);
  }
  return pmh_csr_mult_transpose(pf->G, x, y);
}

// ---- fused epilogues of the projector's G' product (orthonormal G) ----------------------------------------------------
// One launch instead of G' w + VecWAXPY, resp. G' w + VecAYPX + VecScale + VecAXPY: the replicated dual-space work of one
// A_rho = P F P + rho Q application drops from 15 to 10 launches (each ~5 us: what limits the 8-GPU share of configs[2]).
// The row sum is taken exactly as the stream SpMV takes it for these rows (8 lanes per row striding the row, shuffle tree 4-2-1),
// and the epilogue arithmetic is the unfused sequence term by term, so the result is bit-identical to the separate calls:
//   mode 0:  s = (G'w)_r;  y_r = s;  z_r = (-1) s + x_r                      (Q x, and P x = x - Q x: VecWAXPY)
//   mode 1:  s = (G'w)_r;  t = x_r + (-1) s;  y_r = y_r rho + t              (VecAYPX, VecScale, VecAXPY of matpenalized.c:12-22)
__global__ __launch_bounds__(PMH_BLOCK) void k_gt_fused(int n, const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val, const double *__restrict__ w, int mode,
                                                       const double *__restrict__ x, double *__restrict__ y, double *__restrict__ z, double rho)
{
  const int sub = threadIdx.x >> 3, lane = threadIdx.x & 7;
  const int r   = blockIdx.x * (PMH_BLOCK / 8) + sub;
  double    sum = 0.0;
  if (r < n) {
    const int k0 = rowptr[r], k1 = rowptr[r + 1];
    for (int k = k0 + lane; k < k1; k += 8) sum += val[k] * w[col[k]];
  }
#pragma unroll
  for (int o = 4; o > 0; o >>= 1) sum += __shfl_down(sum, o, 8);
  if (r < n && lane == 0) {
    if (mode == 0) {
      y[r] = sum;
      z[r] = -1.0 * sum + x[r];
    } else {
      const double t = x[r] + -1.0 * sum;
      y[r]           = y[r] * rho + t;
    }
  }
}

// the same for short rows (the one-lane-per-row case of the stream kernel: a row is summed left to right by one thread), e.g. the <= 12
// entries per row of an un-filled G0' under implicit orthonormalisation
__global__ __launch_bounds__(PMH_BLOCK) void k_gt_fused1(int n, const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val, const double *__restrict__ w, int mode,
                                                        const double *__restrict__ x, double *__restrict__ y, double *__restrict__ z, double rho)
{
  const int r = blockIdx.x * PMH_BLOCK + threadIdx.x;
  if (r >= n) return;
  const int k0 = rowptr[r], k1 = rowptr[r + 1];
  double    sum = 0.0;
  for (int k = k0; k < k1; k += 16) { // left to right as the plain loop, the loads of 16 entries in flight together
    double v[16];
    int    c[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const bool in = k + j < k1;
      v[j] = in ? val[k + j] : 0.0, c[j] = in ? col[k + j] : 0;
    }
#pragma unroll
    for (int j = 0; j < 16; j++)
      if (k + j < k1) sum += v[j] * w[c[j]];
  }
  if (mode == 0) {
    y[r] = sum;
    z[r] = -1.0 * sum + x[r];
  } else {
    const double t = x[r] + -1.0 * sum;
    y[r]           = y[r] * rho + t;
  }
}

// G not orthonormalised (the dense (G G')^{-1} of QPPFSetUp, qppf.c:213-278): the penalty term rho G'G x and the projector's G'(GG')^{-1}G x are two
// G' products with different coarse vectors -- one pass over G' for both: y = G' t0 (t0 = G x), z = x - G' w (w = (GG')^{-1} t0), each sum left to right
__global__ __launch_bounds__(PMH_BLOCK) void k_gt_dual1(int n, const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val, const double *__restrict__ t0,
                                                       const double *__restrict__ w, const double *__restrict__ x, double *__restrict__ y, double *__restrict__ z)
{
  const int r = blockIdx.x * PMH_BLOCK + threadIdx.x;
  if (r >= n) return;
  const int k0 = rowptr[r], k1 = rowptr[r + 1];
  double    s0 = 0.0, s1 = 0.0;
  for (int k = k0; k < k1; k += 16) {
    double v[16];
    int    c[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const bool in = k + j < k1;
      v[j] = in ? val[k + j] : 0.0, c[j] = in ? col[k + j] : 0;
    }
#pragma unroll
    for (int j = 0; j < 16; j++)
      if (k + j < k1) s0 += v[j] * t0[c[j]], s1 += v[j] * w[c[j]];
  }
  y[r] = s0;
  z[r] = -1.0 * s1 + x[r];
}

// ... and with the finishing step of G0 v and the dense T'T product folded in (implicit orthonormalisation, m <= 64): every workgroup adds the chunk sums of
// k_spmv_long_part per row in chunk order and applies the m x m matrix exactly as k_rows_then_dense does (same order => same bits), then takes its rows -- one
// launch less per projector application aux (workgroup 0 only; SMALXE's ||B u|| riding along): the chunk sums of G0 u (part2, from k_spmv_long_part2) are added
// per row, T is applied and ||T G0 u||^2 is summed exactly as k_rows_then_dense does it -- the same bits as the two launches of pmh_qppf_apply_G_norm2. epi:
// the MPGP vector phase that follows the product, folded in (mode 1 only; one row per thread = the grid of the streaming Vec kernels, so the block partials
// equal those of k_p1_dots / k_axpy + k_split_setp).
struct gt_aux {
  const double *part2, *Mt2; // chunk sums of G0 u, T' (row-major)
  double       *y2, *norm_d, *norm_h;
};
template <int EPI>
__global__ __launch_bounds__(PMH_BLOCK) void k_gt_fused1d(int n, const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val, int m, const int *__restrict__ lrow,
                                                         const double *__restrict__ part, const double *__restrict__ Mt, int mode, const double *__restrict__ x, double *__restrict__ y,
                                                         double *__restrict__ z, double rho, gt_aux aux, pmh_vec_epi epi, const double *__restrict__ pin)
{
  // A 5-8 us kernel is made of memory latencies, not of bytes: every loop below keeps the order of its sum (the same bits as the plain loops) but sends its
  // loads out together -- the plain forms compile to load - wait - add per entry, i.e. ~6 + 12 + 12 latencies in a row (measured 7.7 us against ~4 for a
  // launch).
  __shared__ double t0[64], w[64], Ms[64 * 64];
  const int t = threadIdx.x, mm = m * m;
  // the small matrix goes to LDS while the chunk sums are added (m <= 64: at most 16 entries per thread)
  double mreg[16];
#pragma unroll
  for (int e = 0; e < 16; e++) mreg[e] = (t + PMH_BLOCK * e < mm) ? Mt[t + PMH_BLOCK * e] : 0.0;
  const int r  = blockIdx.x * PMH_BLOCK + t;
  int       k0 = 0, k1 = 0;
  if (r < n) k0 = rowptr[r], k1 = rowptr[r + 1];
  if (t < m) {
    const int c0 = lrow[t], c1 = lrow[t + 1];
    double    sum = 0.0;
    for (int c = c0; c < c1; c += 8) {
      double v[8];
#pragma unroll
      for (int j = 0; j < 8; j++) v[j] = (c + j < c1) ? part[c + j] : 0.0;
#pragma unroll
      for (int j = 0; j < 8; j++)
        if (c + j < c1) sum += v[j];
    }
    t0[t] = sum;
  }
#pragma unroll
  for (int e = 0; e < 16; e++)
    if (t + PMH_BLOCK * e < mm) Ms[t + PMH_BLOCK * e] = mreg[e];
  __syncthreads();
  if (t < m) {
    double s = 0.0;
    for (int c = 0; c < m; c++) s += Ms[c * m + t] * t0[c];
    w[t] = s;
  }
  __syncthreads();
  if (aux.part2 && blockIdx.x == 0) { // uniform per workgroup.  k_rows_then_dense(m, lrow, part2, Mt2, y2, norm): same sums, same order
    __shared__ double t2[64], red2[PMH_BLOCK / 64];
    if (t < m) {
      const int c0 = lrow[t], c1 = lrow[t + 1];
      double    sum = 0.0;
      for (int c = c0; c < c1; c += 8) {
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = (c + j < c1) ? aux.part2[c + j] : 0.0;
#pragma unroll
        for (int j = 0; j < 8; j++)
          if (c + j < c1) sum += v[j];
      }
      t2[t] = sum;
    }
    __syncthreads();
    double sq = 0.0;
    if (t < m) {
      double s = 0.0;
      for (int c = 0; c < m; c += 16) {
        double v[16];
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = (c + j < m) ? aux.Mt2[(size_t)(c + j) * m + t] : 0.0;
#pragma unroll
        for (int j = 0; j < 16; j++)
          if (c + j < m) s += v[j] * t2[c + j];
      }
      aux.y2[t] = s;
      sq += s * s;
    }
    sq = pmh_block_reduce<PMH_RED_SUM>(sq, red2);
    if (t == 0) *aux.norm_d = sq, *aux.norm_h = sq;
  }
  double sum = 0.0;
  if (r < n) {
    for (int k = k0; k < k1; k += 16) { // G0' has 6 or 12 entries per row: one trip
      double v[16];
      int    c[16];
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const bool in = k + j < k1;
        v[j] = in ? val[k + j] : 0.0, c[j] = in ? col[k + j] : 0;
      }
#pragma unroll
      for (int j = 0; j < 16; j++)
        if (k + j < k1) sum += v[j] * w[c[j]];
    }
  }
  double out = 0.0;
  if (r < n) {
    if (mode == 0) {
      y[r] = sum;
      z[r] = -1.0 * sum + x[r];
    } else {
      const double tt = x[r] + -1.0 * sum;
      out             = y[r] * rho + tt;
      if (EPI != PMH_VEPI_GRAD_SPLIT) y[r] = out;
    }
  }
  if (EPI == PMH_VEPI_P1) { // k_p1_dots (mpgp.hip) on this workgroup's rows: p'Ap, g'p, QPCFeas -- `out` is Ap[r], pin the operator's input p
    __shared__ double redp[PMH_BLOCK / 64];
    double            s0 = 0.0, s1 = 0.0, mn = INFINITY;
    if (r < n) {
      const double pi = pin[r];
      s0 += pi * out;
      s1 += epi.g[r] * pi;
      if (pi > 0. && epi.lb) {
        const double l = epi.lb[r];
        if (l > -INFINITY) mn = fmin(mn, (epi.xx[r] - l) / pi);
      }
      if (pi < 0. && epi.ub) {
        const double u = epi.ub[r];
        if (u < INFINITY) mn = fmin(mn, (epi.xx[r] - u) / pi);
      }
    }
    s0 = pmh_block_reduce<PMH_RED_SUM>(s0, redp);
    s1 = pmh_block_reduce<PMH_RED_SUM>(s1, redp);
    mn = pmh_block_reduce<PMH_RED_MIN>(mn, redp);
    if (t == 0) {
      epi.partials[(size_t)epi.prow * epi.ld + blockIdx.x]       = s0;
      epi.partials[(size_t)(epi.prow + 1) * epi.ld + blockIdx.x] = s1;
      epi.partials[(size_t)(epi.prow + 2) * epi.ld + blockIdx.x] = mn;
    }
  }
  // k_axpy(g, -1, b) + k_split_setp (mpgp.hip): g = A x - b, gf, p = gf, the partials of (0, |gP|^2, |gc|^2, |gf|^2) -- pin is the iterate
  if (EPI == PMH_VEPI_GRAD_SPLIT) {
    __shared__ double redg[PMH_BLOCK / 64];
    double            acc[4] = {0.0, 0.0, 0.0, 0.0};
    if (r < n) {
      double gi = out;
      gi += -1.0 * epi.b[r];
      y[r] = gi;
      double f, c;
      pmh_box_split(pin[r], gi, epi.lb, epi.ub, r, epi.astol, f, c);
      epi.gf[r]        = f;
      epi.p[r]         = f;
      const double gPi = f + c;
      acc[1] += gPi * gPi;
      acc[2] += c * c;
      acc[3] += f * f;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const double rr = pmh_block_reduce<PMH_RED_SUM>(acc[k], redg);
      if (t == 0) epi.partials[(size_t)(epi.prow + k) * epi.ld + blockIdx.x] = rr;
    }
  }
}

// the fused path applies when G' is the 8-lanes-per-row stream case (a few dozen entries per row: the rigid-body modes of the
// subdomains a dual row touches), otherwise the callers keep the unfused sequence
static bool gt_fusable(pmh_qppf pf)
{
  if (!pf->orthonormal || pf->d_inv || pf->m == 0 || !pmh_knobs().gt_fusion) return false;
  if (!pf->G->transpose && pmh_csr_ensure_transpose(pf->G)) return false;
  const pmh_csr Gt = pf->G->transpose;
  return Gt->kind == PMH_SPMV_STREAM && (Gt->st_rl == 8 || Gt->st_rl == 1) && Gt->l_nchunks == 0;
}

// the same for G with its dense (G G')^{-1}: one-lane-per-row G' only (k_gt_dual1 / k_gt_fused1)
static bool gt_fusable_dense_inverse(pmh_qppf pf)
{
  if (pf->orthonormal || !pf->d_inv || pf->m == 0 || !pmh_knobs().gt_fusion) return false;
  if (!pf->G->transpose && pmh_csr_ensure_transpose(pf->G)) return false;
  const pmh_csr Gt = pf->G->transpose;
  return Gt->kind == PMH_SPMV_STREAM && Gt->st_rl == 1 && Gt->l_nchunks == 0;
}

// Q v's G' product with its vector epilogue, starting from v: G0 v (chunk sums), then everything else in ONE launch where the folded kernel
// applies (implicit orthonormalisation, long-row G0, short-row G0', m <= 64); otherwise qppf_left + gt_fused
static int gt_fused(pmh_qppf pf, const double *w, int mode, const double *x, double *y, double *z, double rho);
static int qppf_left(pmh_qppf pf, const double *v);
// the one-launch form (k_gt_fused1d) applies: implicit orthonormalisation, long-row G0, one-lane-per-row G0', m <= 64
static bool q_fused_dense(pmh_qppf pf)
{
  return pf->implicit_orth && pf->G->l_nchunks > 0 && pf->m <= 64 && pf->G->transpose && pf->G->transpose->st_rl == 1;
}
// aux_u != nullptr (one-launch form only): G0 aux_u shares the pass over G0, T G0 aux_u -> aux_Gu and its squared norm -> scalar slot aux_slot by workgroup 0.
// epi (mode 1, one-launch form only): the vector phase folded into the kernel, pin = the operator's input vector.
static int q_fused(pmh_qppf pf, const double *v, int mode, const double *x, double *y, double *z, double rho, const double *aux_u = nullptr, double *aux_Gu = nullptr, int aux_slot = -1,
                   const pmh_vec_epi *epi = nullptr, const double *pin = nullptr)
{
  const pmh_csr Gt = pf->G->transpose;
  if (q_fused_dense(pf)) {
    const int    *lrow;
    const double *part, *part2 = nullptr;
    if (aux_u) PMH_CHK(pmh_csr_mult_partials2(pf->G, v, aux_u, &lrow, &part, &part2));
    else PMH_CHK(pmh_csr_mult_partials(pf->G, v, &lrow, &part));
    gt_aux aux;
    memset(&aux, 0, sizeof(aux));
    if (aux_u) aux.part2 = part2, aux.Mt2 = pf->d_Tt, aux.y2 = aux_Gu, aux.norm_d = pf->ctx->d_scal + aux_slot, aux.norm_h = pf->ctx->h_scal + aux_slot;
    pmh_vec_epi e;
    memset(&e, 0, sizeof(e));
    if (epi) e = *epi;
    const dim3 grid((Gt->nrows + PMH_BLOCK - 1) / PMH_BLOCK);
#define GT1D(EPI)                                                                                                                                                                                            \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gt_fused1d<EPI>), grid, dim3(PMH_BLOCK), 0, pf->ctx->stream, Gt->nrows, (const int *)Gt->d_rowptr, (const int *)Gt->d_col, (const double *)Gt->d_val, pf->m, lrow, part, \
                     (const double *)pf->d_S, mode, x, y, z, rho, aux, e, pin)
    if (e.kind == PMH_VEPI_P1) GT1D(PMH_VEPI_P1);
    else if (e.kind == PMH_VEPI_GRAD_SPLIT) GT1D(PMH_VEPI_GRAD_SPLIT);
    else GT1D(0);
#undef GT1D
    PMH_HIP(hipGetLastError());
    return PMH_SUCCESS;
  }
  if (aux_u || epi) return pmh_set_error(PMH_ERR_STATE, "q_fused: the folded forms need the one-launch projector kernel");
  PMH_CHK(qppf_left(pf, v));
  return gt_fused(pf, pf->G_left, mode, x, y, z, rho);
}

static int gt_fused(pmh_qppf pf, const double *w, int mode, const double *x, double *y, double *z, double rho)
{
  const pmh_csr Gt = pf->G->transpose;
  if (Gt->st_rl == 1)
    hipLaunchKernelGGL(k_gt_fused1, dim3((Gt->nrows + PMH_BLOCK - 1) / PMH_BLOCK), dim3(PMH_BLOCK), 0, pf->ctx->stream, Gt->nrows, (const int *)Gt->d_rowptr, (const int *)Gt->d_col, (const double *)Gt->d_val, w,
                       mode, x, y, z, rho);
  else
  hipLaunchKernelGGL(k_gt_fused, dim3((Gt->nrows + PMH_BLOCK / 8 - 1) / (PMH_BLOCK / 8)), dim3(PMH_BLOCK), 0, pf->ctx->stream, Gt->nrows, (const int *)Gt->d_rowptr, (const int *)Gt->d_col,
                     (const double *)Gt->d_val, w, mode, x, y, z, rho);
  PMH_HIP(hipGetLastError());
  return PMH_SUCCESS;
}

// ---- shell operators of the transform chain ------------------------------------------------------------------------
struct ProjectedOp : pmh_op_s {
  pmh_op   A;
  pmh_qppf pf;
  int      symmetric;
  double  *w1, *w2;
  ~ProjectedOp() override
  {
    pmh_free(ctx, w1);
    pmh_free(ctx, w2);
  }
  // MatCreateProd(P,A,P) / (P,A): qptransform.c:273-284, matprod.c:42-48
  int mult(const double *x, double *y) override
  {
    if (symmetric) {
      PMH_CHK(pmh_qppf_apply_P(pf, x, w1));
      PMH_CHK(A->mult(w1, w2));
    } else {
      PMH_CHK(A->mult(x, w2));
    }
    return pmh_qppf_apply_P(pf, w2, y);
  }
  // (P A P)' = P A' P; (P A)' = A' P
  int mult_transpose(const double *x, double *y) override
  {
    PMH_CHK(pmh_qppf_apply_P(pf, x, w1));
    if (symmetric) {
      PMH_CHK(A->mult_transpose(w1, w2));
      return pmh_qppf_apply_P(pf, w2, y);
    }
    return A->mult_transpose(w1, y);
  }
  // same, with Q x supplied by the caller (the QPPFApplyQ cache hit of qppf.c:464-467)
  int mult_with_Qx(const double *x, const double *Qx, double *y)
  {
    if (symmetric) {
      PMH_CHK(pmh_vec_waxpy(ctx, n, w1, -1.0, Qx, x)); // P x = x - Q x
      PMH_CHK(A->mult(w1, w2));
    } else {
      PMH_CHK(A->mult(x, w2));
    }
    return pmh_qppf_apply_P(pf, w2, y);
  }
};

struct PenalizedOp : pmh_op_s {
  pmh_op   A;
  pmh_qppf pf;
  double   rho;
  double  *t, *xwork = nullptr;
  // ||G u|| request riding on the next product (pmh_op_penalized_arm_aux_normG)
  const double *aux_u = nullptr;
  double       *aux_Gu = nullptr;
  int           aux_slot = -1, aux_done = 0;
  // the five-launch chain (dualchain.hip) where F offers its stages: created at the first product
  pmh_dualchain dc = nullptr;
  int           dc_state = -1; // -1 not asked yet, 0 does not apply, 1 live
  int           chain_allowed = 1; // pmh_set_knob("chain") as it stood when the operator was created
  pmh_dualchain chain()
  {
    if (dc_state < 0) {
      dc_state        = 0;
      ProjectedOp *pa = dynamic_cast<ProjectedOp *>(A);
      if (chain_allowed && pa && pa->pf == pf && pa->symmetric && pmh_dc_create(pf, pa->A, &dc) == PMH_SUCCESS && dc) {
        dc_state = 1;
        if (norm_Gu) (void)pmh_dc_set_norm_target(dc, norm_Gu, norm_slot);
      }
    }
    return dc_state == 1 ? dc : nullptr;
  }
  double *norm_Gu = nullptr;
  int     norm_slot = -1;
  int  emit_begin(const double *x, const double *p, pmh_emit_args *ea) override { return chain() ? pmh_dc_emit_begin(dc, x, p, ea) : PMH_EPI_UNSUPPORTED; }
  void emit_invalidate() override
  {
    if (dc) pmh_dc_invalidate(dc);
  }
  // the fully fused product y = rho Q x + P F (P x) [+ vector epilogue] of the one-launch projector form
  bool fused_dense()
  {
    ProjectedOp *pa = dynamic_cast<ProjectedOp *>(A);
    return pa && pa->pf == pf && pa->symmetric && gt_fusable(pf) && q_fused_dense(pf);
  }
  int mult_fused_dense(const double *x, double *y, const pmh_vec_epi *epi)
  {
    ProjectedOp *pa = static_cast<ProjectedOp *>(A);
    const double *au = aux_u;
    aux_u            = nullptr;
    PMH_CHK(q_fused(pf, x, 0, x, y, pa->w1, 0.0, au, aux_Gu, aux_slot)); // y = Q x, w1 = P x (+ T G0 u and its squared norm)
    if (au) aux_done = 1;
    PMH_CHK(pa->A->mult(pa->w1, pa->w2));
    return q_fused(pf, pa->w2, 1, pa->w2, y, nullptr, rho, nullptr, nullptr, -1, epi, x); // y = rho y + (w2 - Q w2) (+ the vector phase)
  }
  int mult_epi(const double *x, double *y, const pmh_vec_epi &e) override
  {
    // (the chain leaves its block partials per 1024-entry tile for the HOST to sum: a caller that finalises on the device -- row-distributed scalars, e.hosted == NULL --
    // counts on one partial per workgroup of the streaming Vec kernels' grid, so it gets the separate vector kernels behind the chain's plain product instead)
    if (chain()) return e.hosted ? pmh_dc_apply(dc, x, y, rho, &e) : PMH_EPI_UNSUPPORTED;
    if (!pmh_knobs().vec_epi) return PMH_EPI_UNSUPPORTED; // A/B: the separate vector kernels
    if (!fused_dense() || n > PMH_MAX_VEC_BLOCKS * PMH_BLOCK || pmh_vec_grid(n) != (n + PMH_BLOCK - 1) / PMH_BLOCK) return PMH_EPI_UNSUPPORTED;
    return mult_fused_dense(x, y, &e);
  }
  ~PenalizedOp() override
  {
    pmh_dc_destroy(dc);
    pmh_free(ctx, t);
    if (xwork) pmh_free(ctx, xwork);
  }
  // MatMult_Penalized matpenalized.c:12-22: y = BtB x; y *= rho; y = y + A x
  int mult(const double *x, double *y) override
  {
    ProjectedOp *pa = dynamic_cast<ProjectedOp *>(A);
    if (chain()) return pmh_dc_apply(dc, x, y, rho, nullptr);
    if (fused_dense()) return mult_fused_dense(x, y, nullptr);
    aux_u = nullptr; // a ||G u|| request can only ride on the one-launch form
    if (pa && pa->pf == pf && pa->symmetric && gt_fusable(pf)) {
      // A = P F P with the same orthonormal projector: y = rho Q x + P F (P x) in 10 launches (see k_gt_fused)
      PMH_CHK(q_fused(pf, x, 0, x, y, pa->w1, 0.0)); // y = Q x, w1 = P x
      PMH_CHK(pa->A->mult(pa->w1, pa->w2));
      return q_fused(pf, pa->w2, 1, pa->w2, y, nullptr, rho); // y = rho y + (w2 - Q w2)
    }
    if (pa && pa->pf == pf && pa->symmetric && gt_fusable_dense_inverse(pf)) {
      // y = rho G'G x + P F P x, P = I - G'(GG')^{-1}G: G x once for both terms, one pass over G' for G'(G x) and x - G'(GG')^{-1}(G x), the second
      // projector with the penalty update in its epilogue -- 12 launches instead of 19, the same bits as the sequence below
      const pmh_csr Gt = pf->G->transpose;
      const dim3    grid((Gt->nrows + PMH_BLOCK - 1) / PMH_BLOCK);
      PMH_CHK(pmh_csr_mult(pf->G, x, pf->G_left));
      PMH_CHK(pmh_qppf_apply_CP(pf, pf->G_left, pf->Gt_right));
      hipLaunchKernelGGL(k_gt_dual1, grid, dim3(PMH_BLOCK), 0, ctx->stream, Gt->nrows, (const int *)Gt->d_rowptr, (const int *)Gt->d_col, (const double *)Gt->d_val, (const double *)pf->G_left,
                         (const double *)pf->Gt_right, x, y, pa->w1);
      PMH_HIP(hipGetLastError());
      PMH_CHK(pa->A->mult(pa->w1, pa->w2));
      PMH_CHK(pmh_csr_mult(pf->G, pa->w2, pf->G_left));
      PMH_CHK(pmh_qppf_apply_CP(pf, pf->G_left, pf->Gt_right));
      hipLaunchKernelGGL(k_gt_fused1, grid, dim3(PMH_BLOCK), 0, ctx->stream, Gt->nrows, (const int *)Gt->d_rowptr, (const int *)Gt->d_col, (const double *)Gt->d_val, (const double *)pf->Gt_right, 1,
                         (const double *)pa->w2, y, (double *)nullptr, rho);
      PMH_HIP(hipGetLastError());
      return PMH_SUCCESS;
    }
    PMH_CHK(pmh_qppf_apply_GtG(pf, x, y));
    if (pa && pa->pf == pf && pa->symmetric && pf->orthonormal) {
      PMH_CHK(pa->mult_with_Qx(x, y, t)); // y currently holds Q x: reuse it before scaling
    } else {
      PMH_CHK(A->mult(x, t));
    }
    PMH_CHK(pmh_vec_scale(ctx, n, y, rho));
    return pmh_vec_axpy(ctx, n, y, 1.0, t);
  }
  // MatMultTranspose_Penalized matpenalized.c:26-36: y = BtB x; y *= rho; y = y + A' x
  int mult_transpose(const double *x, double *y) override
  {
    PMH_CHK(pmh_qppf_apply_GtG(pf, x, y));
    PMH_CHK(A->mult_transpose(x, t));
    PMH_CHK(pmh_vec_scale(ctx, n, y, rho));
    return pmh_vec_axpy(ctx, n, y, 1.0, t);
  }
  // MatMultAdd_Penalized / MatMultTransposeAdd_Penalized matpenalized.c:40-78: x2 != y: y = BtB x; y = rho y + x2;
  // x2 == y: xwork = BtB x; xwork *= rho; y += xwork.  Then y = y + A x (resp. A' x).
  int mult_add(const double *x, const double *x2, double *y, bool transpose)
  {
    if (x2 != y) {
      PMH_CHK(pmh_qppf_apply_GtG(pf, x, y));
      PMH_CHK(pmh_vec_aypx(ctx, n, y, rho, x2));
    } else {
      if (!xwork) PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)n, (void **)&xwork));
      PMH_CHK(pmh_qppf_apply_GtG(pf, x, xwork));
      PMH_CHK(pmh_vec_scale(ctx, n, xwork, rho));
      PMH_CHK(pmh_vec_axpy(ctx, n, y, 1.0, xwork));
    }
    PMH_CHK(transpose ? A->mult_transpose(x, t) : A->mult(x, t));
    return pmh_vec_axpy(ctx, n, y, 1.0, t);
  }
};

extern "C" int pmh_op_create_penalized(pmh_op A, pmh_qppf pf, double rho, pmh_op *op)
{
  PMH_ARG(A && pf && op && rho >= 0);
  PMH_ARG(A->n == pf->n);
  PenalizedOp *o = new PenalizedOp();
  o->ctx         = A->ctx;
  o->n           = A->n;
  o->A           = A;
  o->pf          = pf;
  o->rho         = rho;
  o->chain_allowed = pmh_knobs().chain;
  PMH_CHK(pmh_malloc(o->ctx, sizeof(double) * (size_t)o->n, (void **)&o->t));
  *op = o;
  return PMH_SUCCESS;
}

extern "C" int pmh_op_penalized_mult_add(pmh_op op, const double *x, const double *x2, double *y)
{
  PenalizedOp *o = dynamic_cast<PenalizedOp *>(op);
  PMH_ARG(o && x && x2 && y);
  return o->mult_add(x, x2, y, false);
}

extern "C" int pmh_op_penalized_mult_transpose_add(pmh_op op, const double *x, const double *x2, double *y)
{
  PenalizedOp *o = dynamic_cast<PenalizedOp *>(op);
  PMH_ARG(o && x && x2 && y);
  return o->mult_add(x, x2, y, true);
}

int pmh_op_penalized_arm_aux_normG(pmh_op op, const double *u, double *Gu, int slot)
{
  PenalizedOp *o = dynamic_cast<PenalizedOp *>(op);
  PMH_ARG(o && u && Gu && slot >= 0 && slot < PMH_NSCAL);
  o->aux_done = 0;
  if (o->dc || !o->fused_dense()) return PMH_SUCCESS; // not armed: the caller's own launches follow
  o->aux_u = u, o->aux_Gu = Gu, o->aux_slot = slot;
  return PMH_SUCCESS;
}

// SMALXE's ||B u|| from the chain: every emission of the iterate leaves T G0 u in Gu and its squared norm in the scalar slot (pmh_dc_set_norm_target);
// ready = the last such emission was for this very u and nothing has invalidated it since
int pmh_op_penalized_set_normG_target(pmh_op op, double *Gu, int slot)
{
  PenalizedOp *o = dynamic_cast<PenalizedOp *>(op);
  PMH_ARG(o && Gu && slot >= 0 && slot < PMH_NSCAL);
  o->norm_Gu = Gu, o->norm_slot = slot;
  if (o->dc) PMH_CHK(pmh_dc_set_norm_target(o->dc, Gu, slot));
  return PMH_SUCCESS;
}
int pmh_op_penalized_normG_ready(pmh_op op, const double *u)
{
  PenalizedOp *o = dynamic_cast<PenalizedOp *>(op);
  return (o && o->dc && pmh_dc_norm_ready(o->dc, u)) ? 1 : 0;
}
int pmh_op_penalized_chain_launches(pmh_op op)
{
  PenalizedOp *o = dynamic_cast<PenalizedOp *>(op);
  return (o && o->dc) ? pmh_dc_last_launches(o->dc) : -1;
}

int pmh_op_penalized_take_aux_done(pmh_op op)
{
  PenalizedOp *o = dynamic_cast<PenalizedOp *>(op);
  if (!o) return 0;
  const int d = o->aux_done;
  o->aux_done = 0, o->aux_u = nullptr;
  return d;
}

extern "C" int pmh_op_penalized_set_penalty(pmh_op op, double rho)
{
  PenalizedOp *o = dynamic_cast<PenalizedOp *>(op);
  PMH_ARG(o && rho >= 0);
  o->rho = rho;
  return PMH_SUCCESS;
}

extern "C" int pmh_op_penalized_get_penalty(pmh_op op, double *rho)
{
  PenalizedOp *o = dynamic_cast<PenalizedOp *>(op);
  PMH_ARG(o && rho);
  *rho = o->rho;
  return PMH_SUCCESS;
}

extern "C" int pmh_op_create_projected(pmh_op A, pmh_qppf pf, int symmetric, pmh_op *op)
{
  PMH_ARG(A && pf && op);
  PMH_ARG(A->n == pf->n);
  ProjectedOp *o = new ProjectedOp();
  o->ctx         = A->ctx;
  o->n           = A->n;
  o->A           = A;
  o->pf          = pf;
  o->symmetric   = symmetric ? 1 : 0;
  PMH_CHK(pmh_malloc(o->ctx, sizeof(double) * (size_t)o->n, (void **)&o->w1));
  PMH_CHK(pmh_malloc(o->ctx, sizeof(double) * (size_t)o->n, (void **)&o->w2));
  *op = o;
  return PMH_SUCCESS;
}
