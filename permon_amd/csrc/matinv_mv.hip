// Multi-right-hand-side K^+, part 3: the block CG of MATINV (feti.hip: pmh_matinv_mult) for R = PMH_MV_R columns per block on interleaved multivectors
// V[(dof) * R + column] -- every (block, column) pair is its own CG with its own scalars and its own convergence test (KSPConvergedDefault on the true residual
// recurrence, as the one-column solver), the operator product, the V-cycle and every vector kernel serve all of them at once.  See mv_internal.h.
//   U = K^+ F:   F <- P_R F (Moore-Penrose form) | CG on K U = F, preconditioned by the V-cycle (mg_mv.hip) or by Jacobi | U <- P_R U
// Workgroup (block b, part w) of the vector kernels covers rows lo_b + 32 w + (t / 8), ... of block b and, per row, the 8 columns: thread t works on column t %
// 8.
#include "feti_internal.h"
#include "mv_internal.h"

#define MV_R PMH_MV_R
#define MVC_ROWS (PMH_BLOCK / MV_R)
static_assert(MV_R == 8 && PMH_BLOCK == 256, "the reductions below are written for 8 columns and 4 wavefronts");

struct pmh_matinv_mv_s {
  pmh_matinv M;
  pmh_ctx    ctx;
  pmh_mg_mv  mgmv = nullptr;
  pmh_mv_ell K64  = nullptr;
  int        nb = 0, n = 0, ncol = 0, wgs = 0;
  // nrep = 8: the solver's 8 congruent blocks are the 8 columns of its FIRST block (n = rows of one block); ldR: row stride of M->d_R
  int        nrep = 1, ldR = 0;
  double    *fin = nullptr, *uout = nullptr;  // congruent mode: the interleaved right-hand side / result around pmh_matinv_mv_mult
  std::vector<hipEvent_t> ev;                 // optional timing of the fp64 products (pairs)
  int        ev_on = 0, ev_used = 0;
  double    *r = nullptr, *z = nullptr, *p = nullptr, *Ap = nullptr, *fproj = nullptr;
  double    *partA = nullptr, *partB = nullptr, *partC = nullptr; // [ncol][wgs]: p'Ap | r'z | r'r
  double    *cs = nullptr;                                        // [2 parities][ncol][rz, tol]
  int       *ci = nullptr;                                        // [2 parities][ncol][active, its]
  int       *d_nactive = nullptr, *d_done = nullptr, *h_state = nullptr;
  double    *d_coef = nullptr, *d_kpart = nullptr, *d_fnorm2 = nullptr;
  int        last_max_its = 0;
  long long  products = 0;
};

#define MV_ROW_LOOP(i, b, rs, wgs)                                                                                                                                \
  const int b = blockIdx.x / (wgs), w_ = blockIdx.x % (wgs), cr_ = (int)threadIdx.x % MV_R;                                                                      \
  const int lo_ = (rs)[b], hi_ = (rs)[b + 1];                                                                                                                    \
  _Pragma("unroll 4") for (long long i = ((long long)lo_ + w_ * MVC_ROWS + (int)threadIdx.x / MV_R) * MV_R + cr_; i < (long long)hi_ * MV_R; i += (long long)(wgs)*MVC_ROWS * MV_R)

// sum over the threads that work on the same column (t % 8): every thread gets its column's total.  lds: 32 doubles
static __device__ __forceinline__ double mvc_red8(double v, double *lds)
{
  v += __shfl_xor(v, 8, 64);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane < 8) lds[wave * 8 + lane] = v;
  __syncthreads();
  const int c = threadIdx.x % MV_R;
  return (lds[c] + lds[8 + c]) + (lds[16 + c] + lds[24 + c]);
}
// total of the wgs partials of column (b, t % 8): the same fixed order in every workgroup
static __device__ __forceinline__ double mvc_total(const double *__restrict__ part, int b, int wgs, double *lds)
{
  const int c = threadIdx.x % MV_R;
  double    s = 0.0;
  for (int w = threadIdx.x / MV_R; w < wgs; w += MVC_ROWS) s += part[((size_t)b * MV_R + c) * wgs + w];
  return mvc_red8(s, lds);
}
#define CSQ(cs, q, c, k) (cs)[(((size_t)(q)*ncol + (c)) * 2) + (k)]
#define CIQ(ci, q, c, k) (ci)[(((size_t)(q)*ncol + (c)) * 2) + (k)]

__global__ __launch_bounds__(PMH_BLOCK) void k_mvc_start(const int *__restrict__ rs, int wgs, int extpc, const double *__restrict__ f,
                        const double *__restrict__ dinv, double *__restrict__ u, double *__restrict__ r, double *__restrict__ z,
                                                        double *__restrict__ p, double *__restrict__ partB, double *__restrict__ partC)
{
  __shared__ double lds[32];
  double            s0 = 0.0, s1 = 0.0;
  MV_ROW_LOOP(i, b, rs, wgs)
  {
    const double ri = f[i];
    u[i] = 0.0;
    r[i] = ri;
    if (!extpc) {
      const double zi = dinv[i / MV_R] * ri;
      z[i] = zi;
      p[i] = zi;
      s0 += ri * zi;
    }
    s1 += ri * ri;
  }
  s0 = mvc_red8(s0, lds);
  s1 = mvc_red8(s1, lds);
  if (threadIdx.x < MV_R) {
    const size_t k = ((size_t)b * MV_R + threadIdx.x) * wgs + w_;
    partB[k] = s0, partC[k] = s1;
  }
}

// (TZ = float: z is the V-cycle's own fp32 result, read where it lies -- the fp64 copy the cycle used to leave for the three readers below was 16 MB written and 2 x 16 MB
// read per iteration for the same values, round 6)
template <typename TZ>
__global__ __launch_bounds__(PMH_BLOCK) void k_mvc_start_pz(const int *__restrict__ rs, int wgs, const double *__restrict__ r, const TZ *__restrict__ z,
                        double *__restrict__ p, double *__restrict__ partB)
{
  __shared__ double lds[32];
  double            s0 = 0.0;
  MV_ROW_LOOP(i, b, rs, wgs)
  {
    const double zi = (double)z[i];
    p[i] = zi;
    s0 += r[i] * zi;
  }
  s0 = mvc_red8(s0, lds);
  if (threadIdx.x < MV_R) partB[((size_t)b * MV_R + threadIdx.x) * wgs + w_] = s0;
}

// one workgroup per block: rz, the threshold max(rtol ||f||, atol) and the active flag of its 8 columns (k_cg_init of feti.hip, column by column)
__global__ __launch_bounds__(PMH_BLOCK) void k_mvc_init(int ncol, int wgs, const double *__restrict__ partB, const double *__restrict__ partC,
                        double *__restrict__ cs, int *__restrict__ ci, int *__restrict__ nactive, double rtol, double atol,
                                                       const double *__restrict__ fnorm2, double kernel_tol)
{
  __shared__ double lds[32];
  const int         b  = blockIdx.x;
  const double      rz = mvc_total(partB, b, wgs, lds), rr = mvc_total(partC, b, wgs, lds);
  if (threadIdx.x < MV_R) {
    const int    c   = b * MV_R + threadIdx.x;
    const double tol = fmax(rtol * sqrt(rr), atol);
    int          act = (sqrt(rr) > tol) ? 1 : 0;
    if (fnorm2 && sqrt(rr) <= kernel_tol * 2.220446049250313e-16 * sqrt(fnorm2[c])) act = 0; // the load lies in the kernel: u = 0
    for (int q = 0; q < 2; q++) CSQ(cs, q, c, 0) = rz, CSQ(cs, q, c, 1) = tol, CIQ(ci, q, c, 0) = act, CIQ(ci, q, c, 1) = 0;
    if (act) atomicAdd(nactive, 1);
  }
}
__global__ void k_mvc_init_done(const int *nactive, int *done) { *done = (*nactive == 0) ? 1 : 0; }

template <typename TY>
__global__ __launch_bounds__(PMH_BLOCK) void k_mvc_dot(const int *__restrict__ rs, int wgs, const int *__restrict__ done, const double *__restrict__ x,
                        const TY *__restrict__ y, double *__restrict__ part)
{
  __shared__ double lds[32];
  if (*done) return;
  double s = 0.0;
  MV_ROW_LOOP(i, b, rs, wgs) s += x[i] * (double)y[i];
  s = mvc_red8(s, lds);
  if (threadIdx.x < MV_R) part[((size_t)b * MV_R + threadIdx.x) * wgs + w_] = s;
}

// alpha_c = rz_c / (p'Ap)_c for the active columns (0 for the frozen ones: nothing moves there); u += alpha p; r -= alpha Ap; z = D^-1 r (Jacobi); partials
// r'z, r'r
__global__ __launch_bounds__(PMH_BLOCK) void k_mvc_update_ur(const int *__restrict__ rs, int ncol, int wgs, int q, int extpc, const int *__restrict__ done,
                        const double *__restrict__ cs, const int *__restrict__ ci,
                                                            const double *__restrict__ partA, const double *__restrict__ dinv, const double *__restrict__ p, const double *__restrict__ Ap, double *__restrict__ u,
                                                            double *__restrict__ r, double *__restrict__ z, double *__restrict__ partB, double *__restrict__ partC)
{
  // (the V-cycle's first smoothing direction D^-1 r / theta and its fp32 copy of r written from here, round 6: this kernel 27 -> 36 us for the 6.5 us of k_mvg_d0 -- these
  // passes run at the latency of their few wavefronts, two more store streams cost more than the launch they save)
  __shared__ double lds[32];
  if (*done) return;
  const int    bb  = blockIdx.x / wgs, c = bb * MV_R + (int)threadIdx.x % MV_R;
  const double pAp = mvc_total(partA, bb, wgs, lds);
  const bool   act = CIQ(ci, q, c, 0) != 0;
  const double alpha = act ? CSQ(cs, q, c, 0) / pAp : 0.0;
  double       s0 = 0.0, s1 = 0.0;
  MV_ROW_LOOP(i, b, rs, wgs)
  {
    double ri = r[i];
    if (act) {
      ri -= alpha * Ap[i];
      u[i] += alpha * p[i];
      r[i] = ri;
    }
    if (!extpc) {
      const double zi = dinv[i / MV_R] * ri;
      if (act) z[i] = zi;
      s0 += ri * zi;
    }
    s1 += ri * ri;
  }
  s0 = mvc_red8(s0, lds);
  s1 = mvc_red8(s1, lds);
  if (threadIdx.x < MV_R) {
    const size_t k = ((size_t)bb * MV_R + threadIdx.x) * wgs + w_;
    if (!extpc) partB[k] = s0; // external preconditioner: k_mvc_dot(r, z) fills it afterwards
    partC[k] = s1;
  }
}

// beta_c = rz_new / rz; convergence of column c; p = z + beta p; the block's first workgroup publishes the next state
template <typename TZ>
__global__ __launch_bounds__(PMH_BLOCK) void k_mvc_update_p(const int *__restrict__ rs, int ncol, int wgs, int q, int it, int max_it, double *__restrict__ cs,
                        int *__restrict__ ci, int *__restrict__ nactive, int *__restrict__ done,
                                                           const double *__restrict__ partB, const double *__restrict__ partC, const TZ *__restrict__ z, double *__restrict__ p)
{
  __shared__ double lds[32];
  if (*done) return;
  const int    bb  = blockIdx.x / wgs, c = bb * MV_R + (int)threadIdx.x % MV_R;
  const double rzn = mvc_total(partB, bb, wgs, lds), rr = mvc_total(partC, bb, wgs, lds);
  const bool   act = CIQ(ci, q, c, 0) != 0;
  const double beta = rzn / CSQ(cs, q, c, 0);
  const bool   conv = (sqrt(rr) <= CSQ(cs, q, c, 1)) || (it + 1 >= max_it) || !(rr == rr);
  if (act && !conv) {
    MV_ROW_LOOP(i, b, rs, wgs) p[i] = (double)z[i] + beta * p[i];
  }
  if (blockIdx.x % wgs == 0 && threadIdx.x < MV_R) {
    if (!act) { // carry the frozen state to the other parity
      CSQ(cs, q ^ 1, c, 0) = CSQ(cs, q, c, 0);
      CIQ(ci, q ^ 1, c, 0) = 0;
      CIQ(ci, q ^ 1, c, 1) = CIQ(ci, q, c, 1);
    } else {
      CSQ(cs, q ^ 1, c, 0) = rzn;
      CIQ(ci, q ^ 1, c, 0) = conv ? 0 : 1;
      CIQ(ci, q ^ 1, c, 1) = it + 1;
      if (conv && atomicSub(nactive, 1) == 1) *done = 1; // last active column: later launches of this solve are no-ops
    }
  }
}

__global__ void k_mvc_publish(int ncol, const int *__restrict__ ci, const int *nactive, int *h)
{
  int mx = 0;
  for (int c = 0; c < 2 * ncol; c++) mx = max(mx, ci[(size_t)c * 2 + 1]);
  h[1] = mx;
  h[0] = *nactive;
}

// ---- P_R = I - R R' column by column (k_seg_rt_dot / k_seg_coef / k_seg_project of feti.hip) ----------------------------------------------------------------
#define MVC_MAX_KDIM 8
__global__ __launch_bounds__(PMH_BLOCK) void k_mvc_rt_dot(const int *__restrict__ rs, int wgs, int n, int kdim, size_t ld, const double *__restrict__ R,
                        const double *__restrict__ v, double *__restrict__ part)
{
  __shared__ double lds[32];
  double            acc[MVC_MAX_KDIM], vv = 0.0;
#pragma unroll
  for (int k = 0; k < MVC_MAX_KDIM; k++) acc[k] = 0.0;
  MV_ROW_LOOP(i, b, rs, wgs)
  {
    const double vi = v[i];
    vv += vi * vi;
    // the row's kdim <= 8 kernel entries: ONE load per lane (lane c of the row's 8 brings R_c) handed round by lane index -- every lane loading all of them itself was
    // 6 loads of 64 useful bytes per wavefront, 57 us for the 28 MB of a 43^3 cube (round 6)
    const double rk = (cr_ < kdim) ? R[(size_t)cr_ * n + i / MV_R] : 0.0;
#pragma unroll
    for (int k = 0; k < MVC_MAX_KDIM; k++)
      if (k < kdim) acc[k] += __shfl(rk, ((int)threadIdx.x & 56) + k, 64) * vi;
  }
  const size_t o = ((size_t)b * MV_R + threadIdx.x % MV_R) * wgs + w_;
  vv             = mvc_red8(vv, lds);
  if (threadIdx.x < MV_R) part[(size_t)MVC_MAX_KDIM * ld + o] = vv;
#pragma unroll
  for (int k = 0; k < MVC_MAX_KDIM; k++)
    if (k < kdim) {
      const double s = mvc_red8(acc[k], lds);
      if (threadIdx.x < MV_R) part[(size_t)k * ld + o] = s;
    }
}
__global__ __launch_bounds__(PMH_BLOCK) void k_mvc_coef(int wgs, int kdim, size_t ld, const double *__restrict__ part, double *__restrict__ coef,
                        double *__restrict__ vnorm2)
{
  // grid (blocks, kdim + 1): workgroup (b, k) sums the partials of R_k' v, (b, kdim) those of |v|^2 (one workgroup per block doing the 7 sums in turn took 18 us)
  __shared__ double lds[32];
  const int         b = blockIdx.x, k = blockIdx.y, c = b * MV_R + (int)threadIdx.x % MV_R;
  if (k == kdim) {
    if (!vnorm2) return;
    const double v = mvc_total(part + (size_t)MVC_MAX_KDIM * ld, b, wgs, lds);
    if (threadIdx.x < MV_R) vnorm2[c] = v;
    return;
  }
  const double v = mvc_total(part + (size_t)k * ld, b, wgs, lds);
  if (threadIdx.x < MV_R) coef[(size_t)c * MVC_MAX_KDIM + k] = v;
}
__global__ __launch_bounds__(PMH_BLOCK) void k_mvc_project(const int *__restrict__ rs, int wgs, int n, int kdim, const double *__restrict__ R,
                        const double *__restrict__ coef, const double *v, double *out) // (v == out is allowed)
{
  MV_ROW_LOOP(i, b, rs, wgs)
  {
    const int    c  = b * MV_R + cr_;
    double       s  = v[i];
    const double rk = (cr_ < kdim) ? R[(size_t)cr_ * n + i / MV_R] : 0.0; // (as in k_mvc_rt_dot)
#pragma unroll
    for (int k = 0; k < MVC_MAX_KDIM; k++)
      if (k < kdim) s -= coef[(size_t)c * MVC_MAX_KDIM + k] * __shfl(rk, ((int)threadIdx.x & 56) + k, 64);
    out[i] = s;
  }
}

// V[i][r] -> C[r][i]: the columns one after the other (what the row-extraction kernels of the assembly read)
__global__ __launch_bounds__(PMH_BLOCK) void k_mvc_columns(int n, const double *__restrict__ v, double *__restrict__ c)
{
  __shared__ double tile[MVC_ROWS][MV_R + 1];
  const int         i0 = blockIdx.x * MVC_ROWS;
  const int         ti = threadIdx.x / MV_R, tr = threadIdx.x % MV_R;
  if (i0 + ti < n) tile[ti][tr] = v[(size_t)(i0 + ti) * MV_R + tr];
  __syncthreads();
  const int oi = threadIdx.x % MVC_ROWS, orr = threadIdx.x / MVC_ROWS;
  if (i0 + oi < n) c[(size_t)orr * n + i0 + oi] = tile[oi][orr];
}

// C[r][i] -> V[i][r]
__global__ __launch_bounds__(PMH_BLOCK) void k_mvc_from_columns(int n, const double *__restrict__ c, double *__restrict__ v)
{
  __shared__ double tile[MVC_ROWS][MV_R + 1];
  const int         i0 = blockIdx.x * MVC_ROWS;
  const int         ii = threadIdx.x % MVC_ROWS, ir = threadIdx.x / MVC_ROWS;
  if (i0 + ii < n) tile[ii][ir] = c[(size_t)ir * n + i0 + ii];
  __syncthreads();
  const int ti = threadIdx.x / MV_R, tr = threadIdx.x % MV_R;
  if (i0 + ti < n) v[(size_t)(i0 + ti) * MV_R + tr] = tile[ti][tr];
}

int pmh_matinv_mv_destroy(pmh_matinv_mv V)
{
  if (!V) return PMH_SUCCESS;
  pmh_ctx ctx = V->ctx;
  pmh_mg_mv_destroy(V->mgmv);
  pmh_mv_ell_destroy(V->K64);
  for (double *p : {V->r, V->z, V->p, V->Ap, V->fproj, V->partA, V->partB, V->partC, V->cs, V->d_coef, V->d_kpart, V->d_fnorm2, V->fin, V->uout}) pmh_free(ctx,
                          p);
  for (hipEvent_t e : V->ev) (void)hipEventDestroy(e);
  pmh_free(ctx, V->ci), pmh_free(ctx, V->d_nactive), pmh_free(ctx, V->d_done);
  if (V->h_state) (void)hipHostFree(V->h_state);
  delete V;
  return PMH_SUCCESS;
}

// *out = NULL with PMH_EPI_UNSUPPORTED (no error recorded) where this solver does not apply: K without regular 3 x 3 blocks, a V-cycle of another shape than
// mg_mv.hip runs, the left generalised inverse
static int mvc_create(pmh_matinv M, int nrep, pmh_matinv_mv *out);
int        pmh_matinv_mv_create(pmh_matinv M, pmh_matinv_mv *out) { return mvc_create(M, 1, out); }

// *differs = 1 if the kernel vectors of block b > 0 are not those of block 0 entry by entry, to rounding: |difference| <= tol (R: kdim x ldR, the blocks' rows one after the
// other, nb rows each; bases computed from shifted coordinates agree to ~ 1e-16 but not to the bit)
__global__ __launch_bounds__(PMH_BLOCK) void k_mvc_kernel_differs(int kdim, int nrep, int nb, size_t ldR, const double *__restrict__ R, double tol, int *__restrict__ differs)
{
  for (long long t = (long long)blockIdx.x * PMH_BLOCK + threadIdx.x; t < (long long)kdim * nb; t += (long long)gridDim.x * PMH_BLOCK) {
    const int    k = (int)(t / nb), i = (int)(t % nb);
    const double r0 = R[(size_t)k * ldR + i];
    for (int b = 1; b < nrep; b++)
      if (!(fabs(R[(size_t)k * ldR + (size_t)b * nb + i] - r0) <= tol)) *differs = 1;
  }
}

int pmh_matinv_mv_create_congruent(pmh_matinv M, pmh_matinv_mv *out)
{
  PMH_ARG(M && out);
  *out = nullptr;
  if (M->nblocks != MV_R || !M->Kb || M->Kb->nrep != MV_R || !M->mg) {
    pmh_mv_set_why("not 8 congruent blocks with a V-cycle (pmh_matinv_enable_bsr3 verifies the congruence)");
    return PMH_EPI_UNSUPPORTED;
  }
  if (M->kdim && M->d_R) {
    // the 8 columns are projected with block 0's kernel vectors: the caller's bases must be the same vectors block after block (the kernel SPACE follows from the
    // matrix, the basis does not: another basis, scaling or zero-padded column per block would silently give another K^+ than the one-column path)
    int *d_diff, h_diff = 0;
    PMH_CHK(pmh_malloc(M->ctx, sizeof(int), (void **)&d_diff));
    PMH_HIP(hipMemsetAsync(d_diff, 0, sizeof(int), M->ctx->stream));
    const int nb = M->n / MV_R;
    hipLaunchKernelGGL(k_mvc_kernel_differs, dim3(256), dim3(PMH_BLOCK), 0, M->ctx->stream, M->kdim, MV_R, nb, (size_t)M->n, (const double *)M->d_R, 1e-9 / sqrt((double)nb), d_diff); // (entries of an orthonormal basis are ~ 1 / sqrt(nb))
    PMH_CHK(pmh_memcpy_d2h(M->ctx, &h_diff, d_diff, sizeof(int)));
    pmh_free(M->ctx, d_diff);
    if (h_diff) {
      pmh_mv_set_why("the congruent blocks come with different kernel bases");
      return PMH_EPI_UNSUPPORTED;
    }
  }
  return mvc_create(M, MV_R, out);
}

static int mvc_create(pmh_matinv M, int nrep, pmh_matinv_mv *out)
{
  PMH_ARG(M && out);
  *out = nullptr;
  if (M->left || M->n == 0 || M->n % 3) {
    pmh_mv_set_why(M->left ? "the left generalised inverse is not served" : "the matrix has no 3 x 3 blocks");
    return PMH_EPI_UNSUPPORTED;
  }
  pmh_ctx       ctx = M->ctx;
  pmh_matinv_mv V   = new pmh_matinv_mv_s();
  V->M = M, V->ctx = ctx, V->nrep = nrep, V->nb = M->nblocks / nrep, V->n = M->n / nrep, V->ncol = V->nb * MV_R, V->ldR = M->n;
  int maxrows = 1;
  for (int b = 0; b < V->nb; b++) maxrows = std::max(maxrows, M->K->rowstart[b + 1] - M->K->rowstart[b]);
  // workgroups per block of the vector kernels = partial sums per column: 8 per CU over all blocks, at most 512 a block.  (Measured on the 43^3 cube, one block: 256 -> 512:
  // k_mvc_rt_dot 28.8 -> 17.4 us, k_mvc_project 29.8 -> 20.7, k_mvc_update_ur 28.0 -> 24.1, k_mvc_update_p 17.9 -> 19.4; at 1 024 the consumers' prologue -- every workgroup
  // sums all the partials of its block -- takes the gain back: update_p 35 us, 85 at 2 048.)
  V->wgs = std::max(1, std::min({2 * PMH_BLOCK, (maxrows + 4 * MVC_ROWS - 1) / (4 * MVC_ROWS), std::max(1, 8 * ctx->num_cus / std::max(1, V->nb))}));
  int rc = pmh_mv_ell_create_prefix(M->K->K, nrep, PMH_BSR_F64, &V->K64);
  if (!rc && !V->K64) pmh_mv_set_why("K has rows with unsorted columns or more than 32 blocks of 3 x 3 in a block row"), rc = PMH_EPI_UNSUPPORTED;
  if (!rc && M->mg) rc = pmh_mg_mv_create(M->mg, &V->mgmv, nrep);
  const size_t nR = (size_t)V->n * MV_R, np = (size_t)V->ncol * V->wgs;
  for (double **v : {&V->r, &V->z, &V->p, &V->Ap, &V->fproj})
    if (!rc && !(v == &V->z && V->mgmv)) rc = pmh_malloc(ctx, sizeof(double) * nR, (void **)v); // (z: with a V-cycle the block CG reads the cycle's own fp32 result)
  if (nrep > 1)
    for (double **v : {&V->fin, &V->uout})
      if (!rc) rc = pmh_malloc(ctx, sizeof(double) * nR, (void **)v);
  for (double **v : {&V->partA, &V->partB, &V->partC})
    if (!rc) rc = pmh_malloc(ctx, sizeof(double) * np, (void **)v);
  if (!rc) rc = pmh_malloc(ctx, sizeof(double) * 4 * V->ncol, (void **)&V->cs);
  if (!rc) rc = pmh_malloc(ctx, sizeof(int) * 4 * V->ncol, (void **)&V->ci);
  if (!rc) rc = pmh_malloc(ctx, sizeof(int), (void **)&V->d_nactive);
  if (!rc) rc = pmh_malloc(ctx, sizeof(int), (void **)&V->d_done);
  if (!rc) rc = pmh_malloc(ctx, sizeof(double) * (size_t)V->ncol * MVC_MAX_KDIM, (void **)&V->d_coef);
  if (!rc) rc = pmh_malloc(ctx, sizeof(double) * (MVC_MAX_KDIM + 1) * np, (void **)&V->d_kpart);
  if (!rc) rc = pmh_malloc(ctx, sizeof(double) * V->ncol, (void **)&V->d_fnorm2);
  if (!rc && hipHostMalloc((void **)&V->h_state, sizeof(int) * 2, hipHostMallocDefault) != hipSuccess) rc = pmh_set_error(PMH_ERR_HIP,
                          "pmh_matinv_mv_create: pinned allocation failed");
  if (rc) {
    pmh_matinv_mv_destroy(V);
    return rc;
  }
  *out = V;
  return PMH_SUCCESS;
}

int pmh_matinv_mv_columns(pmh_matinv_mv V) { return V ? V->ncol : 0; }
int pmh_matinv_mv_last_iterations(pmh_matinv_mv V) { return V ? V->last_max_its : 0; }

static int mvc_project(pmh_matinv_mv V, const double *v, double *out, double *vnorm2)
{
  pmh_matinv   M    = V->M;
  const int    grid = V->nb * V->wgs;
  const size_t ld   = (size_t)V->ncol * V->wgs;
  hipStream_t  st   = V->ctx->stream;
  hipLaunchKernelGGL(k_mvc_rt_dot, dim3(grid), dim3(PMH_BLOCK), 0, st, (const int *)M->K->d_rowstart, V->wgs, V->ldR, M->kdim, ld, (const double *)M->d_R, v,
                     V->d_kpart);
  hipLaunchKernelGGL(k_mvc_coef, dim3(V->nb, M->kdim + 1), dim3(PMH_BLOCK), 0, st, V->wgs, M->kdim, ld, (const double *)V->d_kpart, V->d_coef, vnorm2);
  hipLaunchKernelGGL(k_mvc_project, dim3(grid), dim3(PMH_BLOCK), 0, st, (const int *)M->K->d_rowstart, V->wgs, V->ldR, M->kdim, (const double *)M->d_R,
                     (const double *)V->d_coef, v, out);
  PMH_HIP(hipGetLastError());
  return PMH_SUCCESS;
}

// U = K^+ F for the R columns of every block (F, U: n R doubles on the device, interleaved); tolerances and iteration limit are the base solver's
int pmh_matinv_mv_mult(pmh_matinv_mv V, const double *f, double *u)
{
  PMH_ARG(V && f && u && (const void *)f != (const void *)u);
  pmh_matinv  M    = V->M;
  pmh_ctx     ctx  = V->ctx;
  const int   nb = V->nb, wgs = V->wgs, grid = nb * wgs, ncol = V->ncol;
  const int  *rs  = M->K->d_rowstart;
  hipStream_t st  = ctx->stream;
  if (M->kdim) { // F <- P_R F
    PMH_CHK(mvc_project(V, f, V->fproj, V->d_fnorm2));
    f = V->fproj;
  }
  PMH_HIP(hipMemsetAsync(V->d_nactive, 0, sizeof(int), st));
  PMH_HIP(hipMemsetAsync(V->d_done, 0, sizeof(int), st));
  const int    extpc = V->mgmv ? 1 : 0;
  const float *zf    = extpc ? pmh_mg_mv_result(V->mgmv) : nullptr;
  hipLaunchKernelGGL(k_mvc_start, dim3(grid), dim3(PMH_BLOCK), 0, st, rs, wgs, extpc, f, (const double *)M->dinv, u, V->r, V->z, V->p, V->partB, V->partC);
  if (extpc) {
    PMH_CHK(pmh_mg_mv_apply(V->mgmv, V->r, nullptr, V->d_done));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mvc_start_pz<float>), dim3(grid), dim3(PMH_BLOCK), 0, st, rs, wgs, (const double *)V->r, zf, V->p, V->partB);
  }
  hipLaunchKernelGGL(k_mvc_init, dim3(nb), dim3(PMH_BLOCK), 0, st, ncol, wgs, (const double *)V->partB, (const double *)V->partC, V->cs, V->ci, V->d_nactive,
                     M->rtol, M->atol, (const double *)(M->kdim ? V->d_fnorm2 : nullptr), M->kernel_tol);
  hipLaunchKernelGGL(k_mvc_init_done, dim3(1), dim3(1), 0, st, (const int *)V->d_nactive, V->d_done);
  PMH_HIP(hipGetLastError());
  int it = 0, next_check = (V->last_max_its > 0) ? V->last_max_its : (extpc ? 1 : 4);
  while (it < M->max_it) {
    const int q = it & 1;
    const bool timed = V->ev_on && V->ev_used + 2 <= (int)V->ev.size();
    if (timed) {
      PMH_HIP(hipEventRecord(V->ev[V->ev_used], st));
    }
    PMH_CHK(pmh_mv_spmv_f64(V->K64, V->p, V->Ap, PMH_EPI_NONE, nullptr, V->d_done));
    if (timed) {
      PMH_HIP(hipEventRecord(V->ev[V->ev_used + 1], st));
      V->ev_used += 2;
    }
    V->products++;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mvc_dot<double>), dim3(grid), dim3(PMH_BLOCK), 0, st, rs, wgs, (const int *)V->d_done, (const double *)V->p, (const double *)V->Ap, V->partA);
    hipLaunchKernelGGL(k_mvc_update_ur, dim3(grid), dim3(PMH_BLOCK), 0, st, rs, ncol, wgs, q, extpc, (const int *)V->d_done, (const double *)V->cs,
                       (const int *)V->ci, (const double *)V->partA, (const double *)M->dinv, (const double *)V->p,
                       (const double *)V->Ap, u, V->r, V->z, V->partB, V->partC);
    if (extpc) {
      PMH_CHK(pmh_mg_mv_apply(V->mgmv, V->r, nullptr, V->d_done));
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mvc_dot<float>), dim3(grid), dim3(PMH_BLOCK), 0, st, rs, wgs, (const int *)V->d_done, (const double *)V->r, zf, V->partB);
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mvc_update_p<float>), dim3(grid), dim3(PMH_BLOCK), 0, st, rs, ncol, wgs, q, it, M->max_it, V->cs, V->ci, V->d_nactive, V->d_done,
                         (const double *)V->partB, (const double *)V->partC, zf, V->p);
    } else
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mvc_update_p<double>), dim3(grid), dim3(PMH_BLOCK), 0, st, rs, ncol, wgs, q, it, M->max_it, V->cs, V->ci, V->d_nactive, V->d_done,
                         (const double *)V->partB, (const double *)V->partC, (const double *)V->z, V->p);
    PMH_HIP(hipGetLastError());
    it++;
    if (it >= next_check || it >= M->max_it) {
      hipLaunchKernelGGL(k_mvc_publish, dim3(1), dim3(1), 0, st, ncol, (const int *)V->ci, (const int *)V->d_nactive, V->h_state);
      PMH_HIP(hipStreamSynchronize(st));
      if (V->h_state[0] == 0) break;
      next_check = it + (extpc ? 1 : 2);
    }
  }
  V->last_max_its = V->h_state[1];
  if (M->kdim) PMH_CHK(mvc_project(V, u, u, nullptr)); // U <- P_R U, in place: the coefficients are complete before k_mvc_project starts, which reads and writes entry by entry
  return PMH_SUCCESS;
}

// congruent mode: u = K^+ f in the solver's own layout (block after block = column after column of the first block)
int pmh_matinv_mv_mult_blocks(pmh_matinv_mv V, const double *f, double *u)
{
  PMH_ARG(V && V->nrep == MV_R && f && u);
  const dim3 g((V->n + MVC_ROWS - 1) / MVC_ROWS), blk(PMH_BLOCK);
  hipLaunchKernelGGL(k_mvc_from_columns, g, blk, 0, V->ctx->stream, V->n, f, V->fin);
  PMH_CHK(pmh_matinv_mv_mult(V, V->fin, V->uout));
  hipLaunchKernelGGL(k_mvc_columns, g, blk, 0, V->ctx->stream, V->n, (const double *)V->uout, u);
  PMH_HIP(hipGetLastError());
  return PMH_SUCCESS;
}

long long pmh_matinv_mv_products(pmh_matinv_mv V) { return V ? V->products : 0; }

// event pairs around the fp64 products: enable > 0: (re)start with room for that many launches; enable == 0: read (launches, total ms, bytes one launch moves:
// the ELL copy once + x + y)
int pmh_matinv_mv_timing(pmh_matinv_mv V, int enable, int *launches, double *total_ms, double *bytes_per_launch)
{
  PMH_ARG(V);
  if (enable > 0) {
    while ((int)V->ev.size() < 2 * enable) {
      hipEvent_t e;
      PMH_HIP(hipEventCreate(&e));
      V->ev.push_back(e);
    }
    V->ev_on = 1, V->ev_used = 0;
    return PMH_SUCCESS;
  }
  PMH_CHK(pmh_sync(V->ctx));
  double tot = 0.0;
  for (int i = 0; i + 1 < V->ev_used; i += 2) {
    float ms = 0.f;
    PMH_HIP(hipEventElapsedTime(&ms, V->ev[i], V->ev[i + 1]));
    tot += ms;
  }
  if (launches) *launches = V->ev_used / 2;
  if (total_ms) *total_ms = tot;
  if (bytes_per_launch) *bytes_per_launch = (double)V->K64->W * V->K64->nbr * (9.0 * 8.0 + 4.0) + 2.0 * 8.0 * (double)V->n * MV_R;
  return PMH_SUCCESS;
}

// the interleaved result as R separate columns: cols[r * n + i] = u[i * R + r]
int pmh_matinv_mv_to_columns(pmh_matinv_mv V, const double *u, double *cols)
{
  PMH_ARG(V && u && cols);
  hipLaunchKernelGGL(k_mvc_columns, dim3((V->n + MVC_ROWS - 1) / MVC_ROWS), dim3(PMH_BLOCK), 0, V->ctx->stream, V->n, u, cols);
  PMH_HIP(hipGetLastError());
  return PMH_SUCCESS;
}

// ---- test / direct use: U = K^+ F for 8 columns per block, host-side convenience around the calls above (F, U: device, n x 8 interleaved) -------------------
extern "C" int pmh_matinv_mult_multi(pmh_matinv M, const double *F, double *U, int *max_iterations)
{
  PMH_ARG(M && F && U);
  pmh_matinv_mv V  = nullptr;
  int           rc = pmh_matinv_mv_create(M, &V);
  if (rc == PMH_EPI_UNSUPPORTED) return pmh_set_error(PMH_ERR_SUP, "pmh_matinv_mult_multi: the multi-right-hand-side solver does not apply to this K^+: %s",
                          pmh_mv_why());
  if (rc) return rc;
  rc = pmh_matinv_mv_mult(V, F, U);
  if (!rc) rc = pmh_sync(M->ctx);
  if (max_iterations) *max_iterations = V->last_max_its;
  pmh_matinv_mv_destroy(V);
  return rc;
}
