// QPS MPGP on gfx950: QPSSetup_MPGP / QPSSolve_MPGP (src/qps/impls/mpgp/mpgp.c:359-650).
//
// Two drivers behind one entry point:
//  * solve_unfused(): the reference's operation sequence, one device kernel per PETSc Vec/Mat/QPC call
//    (every expansion type x step-length rule, fallback, fallback2).  It exists so that every variant of
//    the reference is available and so that each op-table slot is exercised exactly as PERMON calls it.
//  * solve_fused(): the default variant (std expansion, fixed length, no fallback): the ~25 passes of an
//    iteration collapse into the phases of SURVEY 8d --
//      P1  Ap = A p  with  p'Ap, g'p, QPCFeas fused into the SpMV epilogue          (B_spmv + 24 n bytes)
//      P2  x -= a p, g -= a Ap, gradient split, Ap'gf, |gP|^2, |gc|^2, |gf|^2       (64 n bytes)
//      P3  p = gf - beta p                                                            (24 n bytes)
//    gP, gc, gr are never stored (gc / gr are recomputed inside the rare proportioning / expansion
//    kernels).  Step lengths are read by the kernels from device scalars; the host only decides the step
//    TYPE, and the next iteration's SpMV is enqueued speculatively before the host reads the scalars.
// Both produce the same iterates up to reduction-order rounding; tests compare them with the CPU oracle.
#include <cmath>
#include <string>
#include <vector>

#include "pmh_internal.h"
#include "reduce.h"
#include "box_inline.h"
#include "emit_inline.h"

#define GRID_STRIDE(i, n) for (long long i = (long long)blockIdx.x * PMH_BLOCK + threadIdx.x; i < (n); i += (long long)gridDim.x * PMH_BLOCK)
// the entries of a vector kernel: grid-strided over workgroups of 256 threads, or -- EMIT variants, which end with pmh_emit_tail -- thread t of the 1024-thread
// workgroup b owns entry 1024 b + t
#define VEC_ENTRIES(i, n) \
  for (long long i = EMIT ? ((long long)blockIdx.x * PMH_EMIT_TILE + threadIdx.x) : ((long long)blockIdx.x * PMH_BLOCK + threadIdx.x); i < (n); i += EMIT ? (long long)(n) : (long long)gridDim.x * PMH_BLOCK)
#define VEC_BOUNDS __launch_bounds__(EMIT ? PMH_EMIT_TILE : PMH_BLOCK)

// scalar slots
#define S_PAP 8
#define S_GP 9
#define S_FEAS 10
#define S_APGF 16
#define S_GP2 17
#define S_GC2 18
#define S_GF2 19
#define S_TMP 24

struct pmh_mpgp_s {
  pmh_ctx       ctx;
  pmh_op        A;
  pmh_csr       csr;
  int           n;
  const double *b;
  double       *x;
  const double *lb, *ub;
  pmh_mpgp_opts o;
  // QPS_MPGP state
  double alpha, alpha_user, maxeig;
  int    expproject;
  double *work[10];
  int     nwork;
  // convergence
  pmh_converged_fn cvg;
  // optional: enqueues what the injected convergence test will read, BEFORE the host waits for the step's scalars (one round trip instead of two)
  int (*pre_test)(void *);
  void *pre_test_user;
  // optional: called right before the speculative Ap = A p of the NEXT iteration is enqueued (the iterate is final then): SMALXE lets its ||B u|| ride on that
  // product
  int (*pre_p1)(void *);
  void *pre_p1_user;
  // set by the convergence test: rnorm / (the threshold it has to fall below), 0 = unknown -- how close the NEXT test is to ending the solve
  double cvg_margin;
  // work[3] already holds A x - b for the x and b the next solve starts from (pmh_mpgp_set_gradient_valid): the fused driver skips its first product
  int   g_valid;
  int   epi_ok;            // 1: the operator folds the vector phases into its last kernel (pmh_op_s::mult_epi), 0: it does not, -1: not asked yet
  // > 0: rows 0..3 (gradient split) / 4..6 (P1) of the pinned block partials wait for the host's sum over that many blocks (host_sums)
  int   hsum4 = 0, hsum3 = 0;
  // fused dual-space chain (pmh_op_s::emit_begin): the operator holds G0 x (0) / G0 p (1) of the CURRENT x / p, emitted by the kernel that wrote them
  bool  cx[2] = {false, false};
  int   fin4_pending;      // the partials of the gradient split (rows 0..3) wait for the finalising launch of the next P1 (rows 4..6): one launch for both
  void            *cvg_user;
  double           norm_rhs, ttol, norm_rhs_div;
  int              cvg_setup;
  // results
  double rnorm, gfnorm, gcnorm;
  int    iteration, reason;
  int    nmv, ncg, nexp, nprop, nfinc, nfall;
  char   step;
  int    fallback_state;
  // monitor
  std::vector<char>   t_step;
  std::vector<double> t_gp, t_gf, t_gc, t_alpha;
  // throughput mode
  // speculative device-side CG chain
  int    *d_ctl, *h_ctl;
  double *d_ring, *h_ring;
};

// --------------------------------------------------------------------------------------------------------------------
// device helpers: the box predicates of qpcbox.c restated per element
// --------------------------------------------------------------------------------------------------------------------
// (the box predicates pmh_box_split / pmh_box_reduced: box_inline.h)
template <int K, bool EMIT = false>
__device__ __forceinline__ void write_partials(double (&v)[K], double *lds, double *__restrict__ partials, int ld, double *__restrict__ h_partials = nullptr)
{
  // 1024-thread workgroups; the rows also go to the pinned host copy: the host adds the block sums after its next wait (host_sums), no finalising launch
  if (EMIT) {
    int op[K];
#pragma unroll
    for (int k = 0; k < K; k++) op[k] = PMH_RED_SUM;
    pmh_block_partials<K>(v, op, partials, h_partials, ld);
    return;
  }
#pragma unroll
  for (int k = 0; k < K; k++) {
    double r = pmh_block_reduce<PMH_RED_SUM>(v[k], lds);
    if (threadIdx.x == 0) partials[(size_t)k * ld + blockIdx.x] = r;
  }
}

// gradient split + norms (+ p = gf): after the initial gradient and after an expansion step
// (MPGPGrads mpgp.c:198-223 + VecCopy(gf,p) :507/:615 + the three reductions of :514-521)
template <bool EMIT>
__global__ VEC_BOUNDS void k_split_setp(long long n, const double *__restrict__ x, const double *__restrict__ g, const double *__restrict__ lb, const double *__restrict__ ub, double astol, double *__restrict__ gf, double *__restrict__ p, double *__restrict__ partials, int ld, pmh_emit_args ea, double *__restrict__ h_partials)
{
  __shared__ double lds[PMH_BLOCK / 64];
  double            acc[4] = {0.0, 0.0, 0.0, 0.0}, pv = 0.0, zv = 0.0;
  pmh_emit_regs     R;
  if (EMIT) pmh_emit_prefetch(ea, R);
  VEC_ENTRIES(i, n)
  {
    double f, c;
    pmh_box_split(x[i], g[i], lb, ub, i, astol, f, c);
    gf[i]      = f;
    p[i]       = f;
    pv     = f;
    double gPi = f + c; // VecWAXPY(gP,1,gf,gc)
    acc[1] += gPi * gPi;
    acc[2] += c * c;
    acc[3] += f * f;
  }
  write_partials<4, EMIT>(acc, lds, partials, ld, h_partials);
  if (EMIT) pmh_emit_tail(ea, R, zv, pv); // the segment sums of G0 p for the p = gf just written
}

// device control words of the speculative CG chain
enum { CTL_HALT = 0, CTL_ITER, CTL_NCG, CTL_BASE, CTL_NWORDS };

struct pmh_spec_args { // constants of one solve, passed by value
  int    *ctl;   // device: [halt, iteration, ncg, base]
  double *ring;  // device: per device-side iteration (|gP|^2, |gf|^2, |gc|^2) for the monitor trace
  int     ring_cap;
  int     max_it;
  double  ttol, divtol_rhs, gamma2;
};

// P2: CG / proportioning update.  acg = (g'p)/(p'Ap) from device scalars (mpgp.c:541-543, :628-630);
// x -= acg p; g -= acg Ap (:553-554 / :633-634); split (:555 / :635); Ap'gf for beta (:558); norms.
// SPEC: the kernel first evaluates, from the device scalars alone, everything the host loop would check
// at the top of this iteration -- QPSConvergedDefault (qps.c:688-712), proportionality (mpgp.c:535) and
// acg <= afeas (mpgp.c:547).  If this is a plain CG step it is taken without any host round trip; otherwise
// the chain halts with the state untouched and the host driver takes this iteration.
template <bool SETP, bool SPEC, bool EMIT = false>
__global__ VEC_BOUNDS void k_step_update(long long n, const double *__restrict__ scal, pmh_spec_args sa, double *__restrict__ x, double *__restrict__ g, double *__restrict__ p, const double *__restrict__ Ap, const double *__restrict__ lb, const double *__restrict__ ub, double astol, double *__restrict__ gf, double *__restrict__ partials, int ld, pmh_emit_args ea, double *__restrict__ h_partials, double acg_host, int nb_p1)
{
  __shared__ double lds[PMH_BLOCK / 64];
  double            acg;
  pmh_emit_regs     R;
  if (EMIT) pmh_emit_prefetch(ea, R);
  // no finalising launch ran: acg from the host (it has read p'Ap, g'p) or -- proportioning, taken without a host wait -- from the P1 block partials (rows 4,
  // 5)
  if (EMIT) {
    acg = acg_host;
    if (nb_p1 > 0) acg = pmh_sum_block_partials(partials + (size_t)5 * ld, nb_p1) / pmh_sum_block_partials(partials + (size_t)4 * ld, nb_p1);
  } else {
    acg = scal[S_GP] / scal[S_PAP];
  }
  if (SPEC) {
    if (sa.ctl[CTL_HALT]) return;
    const int    it  = sa.ctl[CTL_ITER];
    const double gP2 = scal[S_GP2], gc2 = scal[S_GC2], gf2 = scal[S_GF2], rnorm = sqrt(gP2);
    bool         go;
    go = (it <= sa.max_it) && (rnorm > sa.ttol) && (rnorm < sa.divtol_rhs); // NaN fails every comparison -> halt
    go = go && (gc2 <= sa.gamma2 * gf2) && (acg <= scal[S_FEAS]) && (it - sa.ctl[CTL_BASE] < sa.ring_cap);
    if (!go) {
      if (blockIdx.x == 0 && threadIdx.x == 0) sa.ctl[CTL_HALT] = 1; // every workgroup reaches the same verdict
      return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      const int k        = it - sa.ctl[CTL_BASE];
      sa.ring[3 * k]     = gP2;
      sa.ring[3 * k + 1] = gf2;
      sa.ring[3 * k + 2] = gc2;
    }
  }
  const double ma     = -acg;
  double       acc[4] = {0.0, 0.0, 0.0, 0.0}, xv = 0.0, pv = 0.0;
  VEC_ENTRIES(i, n)
  {
    double api = Ap[i];
    double xi  = x[i] + ma * p[i];
    double gi  = g[i] + ma * api;
    double f, c;
    pmh_box_split(xi, gi, lb, ub, i, astol, f, c);
    x[i]  = xi;
    g[i]  = gi;
    gf[i] = f;
    if (SETP) p[i] = f;
    xv = xi, pv = f;
    double gPi = f + c;
    acc[0] += api * f;
    acc[1] += gPi * gPi;
    acc[2] += c * c;
    acc[3] += f * f;
  }
  write_partials<4, EMIT>(acc, lds, partials, ld, h_partials);
  if (EMIT) pmh_emit_tail(ea, R, xv, pv); // the segment sums of G0 x (and of G0 p where p = gf was set)
}

// P3: p = gf - bcg p, bcg = (Ap'gf)/(p'Ap) (mpgp.c:558-560: VecAYPX(p,-bcg,gf))
template <bool EMIT>
__global__ VEC_BOUNDS void k_dir_update(long long n, const double *__restrict__ scal, const int *__restrict__ halt, const double *__restrict__ gf, double *__restrict__ p, pmh_emit_args ea, const double *__restrict__ partials, int nb, double pAp_host)
{
  if (halt && *halt) return;
  pmh_emit_regs R;
  if (EMIT) pmh_emit_prefetch(ea, R);
  // EMIT: Ap'gf from the block partials k_step_update just left (row 0: no finalising launch ran), p'Ap from the host
  double       bcg = EMIT ? pmh_sum_block_partials(partials, nb) / pAp_host : scal[S_APGF] / scal[S_PAP];
  const double mb  = -bcg;
  double       pv = 0.0, zv = 0.0;
  VEC_ENTRIES(i, n) p[i] = pv = gf[i] + mb * p[i];
  if (EMIT) pmh_emit_tail(ea, R, zv, pv);
}

// proportioning direction p = gc (mpgp.c:623), gc recomputed from x, g
template <bool EMIT>
__global__ VEC_BOUNDS void k_prop_dir(long long n, const double *__restrict__ x, const double *__restrict__ g, const double *__restrict__ lb, const double *__restrict__ ub, double astol, double *__restrict__ p, pmh_emit_args ea)
{
  double        pv = 0.0, zv = 0.0;
  pmh_emit_regs R;
  if (EMIT) pmh_emit_prefetch(ea, R);
  VEC_ENTRIES(i, n)
  {
    double f, c;
    pmh_box_split(x[i], g[i], lb, ub, i, astol, f, c);
    p[i] = pv = c;
  }
  if (EMIT) pmh_emit_tail(ea, R, zv, pv);
}

// expansion (std direction, fixed length; MPGPExpansion_Std mpgp.c:299-323):
// x -= afeas p; g -= afeas Ap; split; gr; x -= alpha gr.  g is not stored: it is recomputed as A x - b next.
template <bool EMIT>
__global__ VEC_BOUNDS void k_expansion_std(long long n, double afeas, double alpha, double *__restrict__ x, const double *__restrict__ g, const double *__restrict__ p, const double *__restrict__ Ap, const double *__restrict__ lb, const double *__restrict__ ub, double astol, pmh_emit_args ea)
{
  const double maf = -afeas, mal = -alpha;
  double       xv = 0.0, zv = 0.0;
  pmh_emit_regs R;
  if (EMIT) pmh_emit_prefetch(ea, R);
  VEC_ENTRIES(i, n)
  {
    double xi = x[i] + maf * p[i];
    double gi = g[i] + maf * Ap[i];
    double f, c;
    pmh_box_split(xi, gi, lb, ub, i, astol, f, c);
    double r = pmh_box_reduced(xi, f, lb, ub, i, alpha);
    x[i] = xv = xi + mal * r;
  }
  if (EMIT) pmh_emit_tail(ea, R, xv, zv);
}

// p'Ap, g'p, QPCFeas for operators without a fused SpMV epilogue (shell operators: F, P F P, A + rho Q ...)
__global__ __launch_bounds__(PMH_BLOCK) void k_p1_dots(long long n, const double *__restrict__ p, const double *__restrict__ Ap, const double *__restrict__ g, const double *__restrict__ x, const double *__restrict__ lb, const double *__restrict__ ub, double *__restrict__ partials, int ld)
{
  __shared__ double lds[PMH_BLOCK / 64];
  double            s0 = 0.0, s1 = 0.0, m = INFINITY;
  GRID_STRIDE(i, n)
  {
    double pi = p[i];
    s0 += pi * Ap[i];
    s1 += g[i] * pi;
    if (pi > 0. && lb) {
      double l = lb[i];
      if (l > -INFINITY) m = fmin(m, (x[i] - l) / pi);
    }
    if (pi < 0. && ub) {
      double u = ub[i];
      if (u < INFINITY) m = fmin(m, (x[i] - u) / pi);
    }
  }
  s0 = pmh_block_reduce<PMH_RED_SUM>(s0, lds);
  s1 = pmh_block_reduce<PMH_RED_SUM>(s1, lds);
  m  = pmh_block_reduce<PMH_RED_MIN>(m, lds);
  if (threadIdx.x == 0) {
    partials[blockIdx.x]          = s0;
    partials[ld + blockIdx.x]     = s1;
    partials[2 * ld + blockIdx.x] = m;
  }
}

// three squared norms in one pass (unfused driver: VecNorm(gP), VecDot(gc,gc), VecDot(gf,gf) mpgp.c:514-521)
__global__ __launch_bounds__(PMH_BLOCK) void k_three_norms(long long n, const double *__restrict__ gP, const double *__restrict__ gc, const double *__restrict__ gf, double *__restrict__ partials, int ld)
{
  __shared__ double lds[PMH_BLOCK / 64];
  double            acc[4] = {0.0, 0.0, 0.0, 0.0};
  GRID_STRIDE(i, n)
  {
    double a = gP[i], c = gc[i], f = gf[i];
    acc[1] += a * a;
    acc[2] += c * c;
    acc[3] += f * f;
  }
  write_partials<4>(acc, lds, partials, ld);
}

__global__ __launch_bounds__(PMH_BLOCK) void k_filter(long long n, double *v, double tol)
{
  GRID_STRIDE(i, n) if (fabs(v[i]) < tol) v[i] = 0.0;
}

// objective from gradient f = 1/2 x'(g - b) (QPComputeObjectiveFromGradient qp.c:981-996)
__global__ __launch_bounds__(PMH_BLOCK) void k_obj_from_grad(long long n, const double *__restrict__ x, const double *__restrict__ g, const double *__restrict__ b, double *__restrict__ partials)
{
  __shared__ double lds[PMH_BLOCK / 64];
  double            s = 0.0;
  GRID_STRIDE(i, n) s += x[i] * (-1.0 * b[i] + g[i]);
  s = pmh_block_reduce<PMH_RED_SUM>(s, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

#define LAUNCH(kern, ...) \
  do { \
    if (s->n > 0) hipLaunchKernelGGL(kern, dim3(pmh_vec_grid(s->n)), dim3(PMH_BLOCK), 0, s->ctx->stream, (long long)s->n, __VA_ARGS__); \
    PMH_HIP(hipGetLastError()); \
  } while (0)

// the EMIT variants: workgroups of PMH_EMIT_TILE threads, one entry each
#define LAUNCH_EMIT(kern, ...) \
  do { \
    if (s->n > 0) hipLaunchKernelGGL(kern, dim3((s->n + PMH_EMIT_TILE - 1) / PMH_EMIT_TILE), dim3(PMH_EMIT_TILE), 0, s->ctx->stream, (long long)s->n, __VA_ARGS__); \
    PMH_HIP(hipGetLastError()); \
  } while (0)

// --------------------------------------------------------------------------------------------------------------------
// set-up
// --------------------------------------------------------------------------------------------------------------------
extern "C" int pmh_mpgp_default_opts(pmh_mpgp_opts *o)
{
  PMH_ARG(o);
  memset(o, 0, sizeof(*o));
  o->rtol          = 1e-5; // QPSCreate qps.c:73-76
  o->atol          = 1e-50;
  o->divtol        = 1e4;
  o->max_it        = 10000;
  o->alpha_user    = PMH_DECIDE; // QPSCreate_MPGP mpgp.c:827-843
  o->alpha_direct  = 0;
  o->gamma         = 1.0;
  o->maxeig        = PMH_DECIDE;
  o->maxeig_tol    = PMH_DECIDE;
  o->maxeig_iter   = -1;
  o->bchop_tol     = 0.0;
  o->astol         = 10 * 2.220446049250313e-16; // qpc.c:28
  o->exptype       = PMH_EXP_STD;
  o->explengthtype = PMH_EXPLEN_FIXED;
  return PMH_SUCCESS;
}

static int ensure_work(pmh_mpgp s, int i)
{
  if (!s->work[i]) {
    PMH_CHK(pmh_malloc(s->ctx, sizeof(double) * (size_t)s->n, (void **)&s->work[i]));
    PMH_CHK(pmh_memset(s->ctx, s->work[i], 0, sizeof(double) * (size_t)s->n));
  }
  return PMH_SUCCESS;
}

static bool use_fused(pmh_mpgp s)
{
  return !s->o.unfused && s->o.exptype == PMH_EXP_STD && s->o.explengthtype == PMH_EXPLEN_FIXED && !s->o.fallback && !s->o.fallback2;
}

// QPSSetup_MPGP mpgp.c:359-428
extern "C" int pmh_mpgp_create(pmh_ctx ctx, pmh_op A, const double *b, double *x, const double *lb, const double *ub, const pmh_mpgp_opts *o, pmh_mpgp *out)
{
  PMH_ARG(ctx && A && b && x && o && out);
  PMH_ARG(o->exptype >= PMH_EXP_STD && o->exptype <= PMH_EXP_GGR);
  PMH_ARG(o->explengthtype >= PMH_EXPLEN_FIXED && o->explengthtype <= PMH_EXPLEN_BB);
  pmh_mpgp s = new pmh_mpgp_s();
  s->ctx     = ctx;
  s->A       = A;
  s->csr     = A->as_csr();
  s->n       = A->n;
  s->b       = b;
  s->x       = x;
  s->lb      = lb;
  s->ub      = ub;
  s->o       = *o;
  if (s->o.fallback2) s->o.fallback = 0; // mpgp.c:744
  for (int i = 0; i < 10; i++) s->work[i] = nullptr;
  s->nwork = 0;
  s->cvg = nullptr, s->cvg_user = nullptr, s->cvg_setup = 0;
  s->pre_test = nullptr, s->pre_test_user = nullptr;
  s->pre_p1 = nullptr, s->pre_p1_user = nullptr;
  s->epi_ok = -1, s->fin4_pending = 0, s->g_valid = 0, s->cvg_margin = 0.0;
  s->norm_rhs = s->ttol = s->norm_rhs_div = 0.0;
  s->rnorm = s->gfnorm = s->gcnorm = 0.0;
  s->iteration = 0, s->reason = 0;
  s->nmv = s->ncg = s->nexp = s->nprop = s->nfinc = s->nfall = 0;
  s->step           = ' ';
  s->fallback_state = s->o.fallback;
  s->d_ctl = s->h_ctl = nullptr;
  s->d_ring = s->h_ring = nullptr;
  if (s->o.bchop_tol) { // mpgp.c:379-382: VecFilter on the user's bound vectors, in place as in the reference
    if (lb) LAUNCH(k_filter, (double *)lb, s->o.bchop_tol);
    if (ub) LAUNCH(k_filter, (double *)ub, s->o.bchop_tol);
  }
  s->expproject = 1; // mpgp.c:839
  if (s->o.exptype == PMH_EXP_STD && s->o.explengthtype == PMH_EXPLEN_FIXED) s->expproject = 0; // mpgp.c:388
  s->maxeig     = s->o.maxeig;
  s->alpha_user = s->o.alpha_user;
  if (!s->o.alpha_direct) { // QPS_ARG_MULTIPLE mpgp.c:417-422
    if (s->maxeig == PMH_DECIDE) {
      ctx->dist_scalars = s->o.distributed;
      int rc            = pmh_op_max_eigenvalue(A, s->o.maxeig_tol, s->o.maxeig_iter, &s->maxeig, nullptr);
      ctx->dist_scalars = 0;
      if (rc) {
        delete s;
        return rc;
      }
    }
    if (s->alpha_user == PMH_DECIDE) s->alpha_user = 2.0;
    s->alpha = s->alpha_user / s->maxeig;
  } else {
    s->alpha = s->alpha_user;
  }
  *out = s;
  return PMH_SUCCESS;
}

extern "C" int pmh_mpgp_destroy(pmh_mpgp s)
{
  if (!s) return PMH_SUCCESS;
  for (int i = 0; i < 10; i++)
    if (s->work[i]) pmh_free(s->ctx, s->work[i]);
  if (s->d_ctl) pmh_free(s->ctx, s->d_ctl);
  if (s->d_ring) pmh_free(s->ctx, s->d_ring);
  if (s->h_ctl) hipHostFree(s->h_ctl);
  if (s->h_ring) hipHostFree(s->h_ring);
  delete s;
  return PMH_SUCCESS;
}

// internal (SMALXE): f is called before the host waits for the scalars of a step, with the iterate already updated; what it enqueues is complete when the
// injected convergence test runs
int pmh_mpgp_set_pre_test_hook(pmh_mpgp s, int (*f)(void *), void *user)
{
  PMH_ARG(s);
  s->pre_test = f, s->pre_test_user = user;
  return PMH_SUCCESS;
}

int pmh_mpgp_set_pre_p1_hook(pmh_mpgp s, int (*f)(void *), void *user)
{
  PMH_ARG(s);
  s->pre_p1 = f, s->pre_p1_user = user;
  return PMH_SUCCESS;
}

extern "C" int pmh_mpgp_set_convergence_test(pmh_mpgp s, pmh_converged_fn f, void *user)
{
  PMH_ARG(s);
  s->cvg      = f;
  s->cvg_user = user;
  return PMH_SUCCESS;
}

extern "C" int pmh_mpgp_set_tolerances(pmh_mpgp s, double rtol, double atol, double divtol, int max_it)
{
  PMH_ARG(s);
  s->o.rtol    = rtol;
  s->o.atol    = atol;
  s->o.divtol  = divtol;
  s->o.max_it  = max_it;
  s->cvg_setup = 0;
  return PMH_SUCCESS;
}

// "QPSMPGPSetOperatorMaxEigenvalue_MPGP_C" mpgp.c:107-115 (then QPSSetup_MPGP recomputes alpha :417-422)
extern "C" int pmh_mpgp_set_operator_max_eigenvalue(pmh_mpgp s, double maxeig)
{
  PMH_ARG(s && (maxeig >= 0 || maxeig == PMH_DECIDE));
  s->maxeig = maxeig;
  if (!s->o.alpha_direct) {
    if (s->maxeig == PMH_DECIDE) PMH_CHK(pmh_op_max_eigenvalue(s->A, s->o.maxeig_tol, s->o.maxeig_iter, &s->maxeig, nullptr));
    s->alpha = s->alpha_user / s->maxeig;
  }
  return PMH_SUCCESS;
}

// "QPSMPGPUpdateMaxEigenvalue_MPGP_C" mpgp.c:119-143
extern "C" int pmh_mpgp_update_max_eigenvalue(pmh_mpgp s, double maxeig_update)
{
  PMH_ARG(s);
  if (maxeig_update == 1.0) return PMH_SUCCESS; // mpgp.c:1007
  s->maxeig = s->maxeig * maxeig_update;
  if (!s->o.alpha_direct) s->alpha = s->alpha / maxeig_update;
  return PMH_SUCCESS;
}

extern "C" int pmh_mpgp_get_current_step_type(pmh_mpgp s, char *step)
{
  PMH_ARG(s && step);
  *step = s->step;
  return PMH_SUCCESS;
}

extern "C" int pmh_mpgp_reset_statistics(pmh_mpgp s)
{
  PMH_ARG(s);
  s->ncg = s->nexp = s->nmv = s->nprop = 0; // mpgp.c:659-662 (nfinc/nfall are not reset there)
  return PMH_SUCCESS;
}

// QPSConvergedDefault qps.c:675-714 + QPSConvergedDefaultSetUp :718-731
static int converged_default(pmh_mpgp s)
{
  if (!s->cvg_setup) {
    PMH_CHK(pmh_vec_norm2(s->ctx, s->n, s->b, &s->norm_rhs));
    s->ttol         = fmax(s->o.rtol * s->norm_rhs, s->o.atol);
    s->norm_rhs_div = s->norm_rhs;
    s->cvg_setup    = 1;
  }
  s->reason = PMH_CONVERGED_ITERATING;
  if (s->iteration > s->o.max_it) { // strict, qps.c:688
    s->reason = PMH_DIVERGED_ITS;
    return PMH_SUCCESS;
  }
  if (std::isnan(s->rnorm) || std::isinf(s->rnorm)) s->reason = PMH_DIVERGED_NANORINF;
  else if (s->rnorm <= s->ttol) s->reason = (s->rnorm < s->o.atol) ? PMH_CONVERGED_ATOL : PMH_CONVERGED_RTOL;
  else if (s->rnorm >= s->o.divtol * s->norm_rhs_div) s->reason = PMH_DIVERGED_DTOL;
  s->cvg_margin = (s->iteration >= s->o.max_it) ? 1e-300 : s->rnorm / fmax(s->ttol, 1e-300);
  return PMH_SUCCESS;
}

// monitor (mpgp.c:524-528) + convergence test (mpgp.c:531)
static int test_convergence(pmh_mpgp s)
{
  if (s->o.monitor) {
    s->t_step.push_back(s->step);
    s->t_gp.push_back(s->rnorm);
    s->t_gf.push_back(s->gfnorm);
    s->t_gc.push_back(s->gcnorm);
    s->t_alpha.push_back(s->alpha);
  }
  if (s->cvg) {
    int reason = 0;
    int rc     = s->cvg(s->cvg_user, s->iteration, s->rnorm, &reason);
    if (rc) return pmh_set_error(PMH_ERR_ARG, "convergence test callback returned %d", rc);
    s->reason = reason;
  } else {
    PMH_CHK(converged_default(s));
  }
  return PMH_SUCCESS;
}

// ---- fused dual-space chain: emission by the vector kernels (pmh_op_s::emit_begin, emit_inline.h) ----------------------------
static pmh_emit_args g_noea;
// the kernel about to be launched writes x (wx) and / or p (wp), one entry per thread: true = *ea is filled and the EMIT variant must be launched
static bool emit_try(pmh_mpgp s, bool wx, bool wp, pmh_emit_args *ea)
{
  if (wx) s->cx[0] = false;
  if (wp) s->cx[1] = false;
  if (s->csr || s->epi_ok != 1 || s->o.distributed) {
    if (wx) s->A->emit_invalidate();
    return false;
  }
  if (s->A->emit_begin(wx ? s->x : nullptr, wp ? s->work[4] : nullptr, ea) != PMH_SUCCESS) {
    if (wx) s->A->emit_invalidate();
    return false;
  }
  if (wx) s->cx[0] = true;
  if (wp) s->cx[1] = true;
  return true;
}
static int emit_blocks(pmh_mpgp s) { return (s->n + PMH_EMIT_TILE - 1) / PMH_EMIT_TILE; } // grid of the EMIT variants

// The last step of a reduction whose block partials a kernel left in the pinned host copy, taken by the host after its wait: lane l of a wave of 64 adds its
// entries l, l + 64, ... in that order, then the butterflies of pmh_wave_all (emit_inline.h) -- the order in which a device consumer adds the same row
// (pmh_sum_block_partials), so both see the same number.
static double host_sum_row(const double *row, int nb, int op)
{
  double v[64];
  for (int l = 0; l < 64; l++) {
    double acc = (op == PMH_RED_SUM) ? 0.0 : INFINITY;
    for (int u = 0; u < 8; u++) {
      const double p = (l + 64 * u < nb) ? row[l + 64 * u] : ((op == PMH_RED_SUM) ? 0.0 : INFINITY);
      acc            = (op == PMH_RED_SUM) ? acc + p : fmin(acc, p);
    }
    v[l] = acc;
  }
  auto comb = [&](double a, double b) { return (op == PMH_RED_SUM) ? a + b : fmin(a, b); };
  double t[64];
  for (int l = 0; l < 64; l++) t[l] = comb(v[l], v[l ^ 1]);
  for (int l = 0; l < 64; l++) v[l] = comb(t[l], t[l ^ 2]);
  for (int l = 0; l < 64; l++) t[l] = comb(v[l], v[(l & ~7) | (7 - (l & 7))]);
  for (int l = 0; l < 64; l++) v[l] = comb(t[l], t[(l & ~15) | (15 - (l & 15))]);
  return comb(comb(v[0], v[16]), comb(v[32], v[48]));
}
// after a wait on the stream: the sums the EMIT kernels / the operator's last kernel left to the host (the role of k_finalize) -> h_scal
static void host_sums(pmh_mpgp s)
{
  pmh_ctx   ctx = s->ctx;
  const int ld  = ctx->partials_cap;
  if (s->hsum4) {
    for (int k = 0; k < 4; k++) ctx->h_scal[S_APGF + k] = host_sum_row(ctx->h_partials + (size_t)k * ld, s->hsum4, PMH_RED_SUM);
    s->hsum4 = 0;
  }
  if (s->hsum3) {
    ctx->h_scal[S_PAP]  = host_sum_row(ctx->h_partials + (size_t)4 * ld, s->hsum3, PMH_RED_SUM);
    ctx->h_scal[S_GP]   = host_sum_row(ctx->h_partials + (size_t)5 * ld, s->hsum3, PMH_RED_SUM);
    ctx->h_scal[S_FEAS] = host_sum_row(ctx->h_partials + (size_t)6 * ld, s->hsum3, PMH_RED_MIN);
    s->hsum3            = 0;
  }
}

static int finalize_vec4_from(pmh_mpgp s, int nblocks) // rows 0..3 written by an EMIT variant (its own grid) after all: finalise on the device
{
  const int ops[4] = {PMH_RED_SUM, PMH_RED_SUM, PMH_RED_SUM, PMH_RED_SUM};
  s->hsum4         = 0;
  return pmh_finalize_partials(s->ctx, s->ctx->d_partials, s->ctx->partials_cap, nblocks, 4, ops, S_APGF);
}
static int finalize_vec4(pmh_mpgp s, const int *halt = nullptr, int *post = nullptr)
{
  const int ops[4] = {PMH_RED_SUM, PMH_RED_SUM, PMH_RED_SUM, PMH_RED_SUM};
  return pmh_finalize_partials(s->ctx, s->ctx->d_partials, s->ctx->partials_cap, s->n > 0 ? pmh_vec_grid(s->n) : 0, 4, ops, S_APGF, halt, post);
}

// --------------------------------------------------------------------------------------------------------------------
// unfused driver: QPSSolve_MPGP mpgp.c:438-650 call by call
// --------------------------------------------------------------------------------------------------------------------
static double *exp_direction(pmh_mpgp s)
{
  switch (s->o.exptype) { // mpgp.c:384-414
  case PMH_EXP_STD: return s->work[6];
  case PMH_EXP_GF: return s->work[1];
  case PMH_EXP_G: return s->work[3];
  case PMH_EXP_GFGR: return s->work[1];
  case PMH_EXP_GGR: return s->work[3];
  default: return s->work[1]; // projcg fallback vectors
  }
}
static double *exp_lengthvec(pmh_mpgp s)
{
  switch (s->o.exptype) {
  case PMH_EXP_STD: return s->work[6];
  case PMH_EXP_GF: return s->work[1];
  case PMH_EXP_G: return s->work[3];
  case PMH_EXP_GFGR: return s->work[6];
  case PMH_EXP_GGR: return s->work[6];
  default: return s->work[1];
  }
}

// MPGPGrads mpgp.c:198-223
static int u_grads(pmh_mpgp s, const double *x, const double *g)
{
  PMH_CHK(pmh_qpc_box_grads(s->ctx, s->n, x, g, s->lb, s->ub, s->o.astol, s->work[1], s->work[2]));
  PMH_CHK(pmh_qpc_box_gradreduced(s->ctx, s->n, x, s->work[1], s->lb, s->ub, s->alpha, s->work[6]));
  return pmh_vec_waxpy(s->ctx, s->n, s->work[0], 1.0, s->work[1], s->work[2]);
}

// MPGPExpansionLength mpgp.c:233-287
static int u_expansion_length(pmh_mpgp s, double *xold, double *explengthvecold)
{
  pmh_ctx ctx = s->ctx;
  int     n   = s->n;
  double *lv  = exp_lengthvec(s), dots[2];
  switch (s->o.explengthtype) {
  case PMH_EXPLEN_FIXED: break;
  case PMH_EXPLEN_OPT:
    PMH_CHK(s->A->mult(lv, s->work[5]));
    s->nmv++;
    PMH_CHK(pmh_vec_dot(ctx, n, lv, s->work[3], &dots[0]));
    PMH_CHK(pmh_vec_dot(ctx, n, lv, s->work[5], &dots[1]));
    if (dots[1] == .0 && s->o.resetalpha) s->alpha = s->alpha / s->maxeig;
    else s->alpha = s->alpha_user * (dots[0] / dots[1]);
    break;
  case PMH_EXPLEN_OPTAPPROX:
    if (s->work[3] != lv) {
      PMH_CHK(pmh_vec_dot(ctx, n, lv, s->work[3], &dots[0]));
      PMH_CHK(pmh_vec_dot(ctx, n, lv, lv, &dots[1]));
      s->alpha = s->alpha_user * (dots[0] / dots[1]);
    } else {
      s->alpha = s->alpha_user;
    }
    s->alpha = s->alpha / s->maxeig;
    break;
  case PMH_EXPLEN_BB:
    PMH_CHK(pmh_vec_aypx(ctx, n, explengthvecold, -1.0, lv));
    PMH_CHK(pmh_vec_aypx(ctx, n, xold, -1.0, s->x));
    PMH_CHK(pmh_vec_dot(ctx, n, explengthvecold, explengthvecold, &dots[0]));
    PMH_CHK(pmh_vec_dot(ctx, n, explengthvecold, xold, &dots[1]));
    if (dots[1] == .0 && s->o.resetalpha) s->alpha = s->alpha / s->maxeig;
    else s->alpha = s->alpha_user * (dots[0] / dots[1]);
    break;
  }
  return PMH_SUCCESS;
}

// MPGPExpansion_Std mpgp.c:299-323
static int u_expansion_std(pmh_mpgp s, double afeas, double *xold, double *explengthvecold)
{
  pmh_ctx ctx = s->ctx;
  int     n   = s->n;
  PMH_CHK(pmh_vec_axpy(ctx, n, s->x, -afeas, s->work[4]));
  PMH_CHK(pmh_vec_axpy(ctx, n, s->work[3], -afeas, s->work[5]));
  PMH_CHK(u_grads(s, s->x, s->work[3]));
  PMH_CHK(u_expansion_length(s, xold, explengthvecold));
  return pmh_vec_axpy(ctx, n, s->x, -s->alpha, exp_direction(s));
}

static int u_objective_from_gradient(pmh_mpgp s, const double *x, const double *g, double *f)
{
  const int ops[1] = {PMH_RED_SUM};
  LAUNCH(k_obj_from_grad, x, g, s->b, s->ctx->d_partials);
  PMH_CHK(pmh_finalize_partials(s->ctx, s->ctx->d_partials, s->ctx->partials_cap, s->n > 0 ? pmh_vec_grid(s->n) : 0, 1, ops, S_TMP));
  double v;
  PMH_CHK(pmh_host_scalar(s->ctx, S_TMP, &v));
  *f = .5 * v;
  return PMH_SUCCESS;
}

static int solve_unfused(pmh_mpgp s)
{
  s->g_valid  = 0; // (this driver always forms its own gradient)
  s->cx[0] = s->cx[1] = false;
  s->A->emit_invalidate();
  pmh_ctx ctx = s->ctx;
  int     n   = s->n;
  int     nw  = 7;
  if (s->o.fallback || s->o.fallback2) nw = (s->o.explengthtype != PMH_EXPLEN_BB) ? 9 : 10; // mpgp.c:366-376
  else if (s->o.explengthtype == PMH_EXPLEN_BB) nw = 9;
  for (int i = 0; i < nw; i++) PMH_CHK(ensure_work(s, i));
  double *gP = s->work[0], *gf = s->work[1], *gc = s->work[2], *g = s->work[3], *p = s->work[4], *Ap = s->work[5];
  double *gold = nullptr, *xold = nullptr, *explengthvecold = nullptr;
  if (s->o.explengthtype == PMH_EXPLEN_BB) { // mpgp.c:479-486
    explengthvecold = s->work[7];
    xold            = s->work[8];
    if (s->o.fallback || s->o.fallback2) gold = s->work[9];
  } else if (s->o.fallback || s->o.fallback2) {
    xold = s->work[7];
    gold = s->work[8];
  }
  const double gamma2 = s->o.gamma * s->o.gamma;
  double       acg, bcg, afeas, pAp, gcTgc, gfTgf, f, fold;
  int          nmv = 0, ncg = 0, nprop = 0, nexp = 0, nfinc = 0, nfall = 0;
  int          fallback = s->fallback_state;
  double      *x = s->x;

  PMH_CHK(pmh_qpc_box_project(ctx, n, x, s->lb, s->ub, x)); // mpgp.c:497
  PMH_CHK(s->A->mult(x, g));
  nmv++;
  PMH_CHK(pmh_vec_axpy(ctx, n, g, -1.0, s->b));
  PMH_CHK(u_grads(s, x, g));
  PMH_CHK(pmh_vec_copy(ctx, n, gf, p));
  s->step      = ' ';
  s->iteration = 0;
  while (1) {
    LAUNCH(k_three_norms, (const double *)gP, (const double *)gc, (const double *)gf, ctx->d_partials, ctx->partials_cap);
    PMH_CHK(finalize_vec4(s));
    PMH_CHK(pmh_sync(ctx));
    s->rnorm  = sqrt(ctx->h_scal[S_GP2]);
    gcTgc     = ctx->h_scal[S_GC2];
    gfTgf     = ctx->h_scal[S_GF2];
    s->gfnorm = sqrt(gfTgf);
    s->gcnorm = sqrt(gcTgc);
    PMH_CHK(test_convergence(s));
    if (s->reason != PMH_CONVERGED_ITERATING) break;

    if (gcTgc <= gamma2 * gfTgf) {
      PMH_CHK(s->A->mult(p, Ap));
      nmv++;
      PMH_CHK(pmh_vec_dot(ctx, n, p, Ap, &pAp));
      PMH_CHK(pmh_vec_dot(ctx, n, g, p, &acg));
      acg = acg / pAp;
      PMH_CHK(pmh_qpc_box_feas(ctx, n, x, p, s->lb, s->ub, &afeas));
      if (acg <= afeas) {
        ncg++;
        s->step = 'c';
        PMH_CHK(pmh_vec_axpy(ctx, n, x, -acg, p));
        PMH_CHK(pmh_vec_axpy(ctx, n, g, -acg, Ap));
        PMH_CHK(u_grads(s, x, g));
        PMH_CHK(pmh_vec_dot(ctx, n, Ap, gf, &bcg));
        bcg = bcg / pAp;
        PMH_CHK(pmh_vec_aypx(ctx, n, p, -bcg, gf));
      } else {
        nexp++;
        s->step = 'e';
        if (s->o.explengthtype == PMH_EXPLEN_BB || fallback || s->o.fallback2) {
          PMH_CHK(pmh_vec_copy(ctx, n, x, xold));
          if (s->o.explengthtype == PMH_EXPLEN_BB) PMH_CHK(pmh_vec_copy(ctx, n, exp_lengthvec(s), explengthvecold));
        }
        if (s->o.exptype == PMH_EXP_PROJCG) PMH_CHK(pmh_vec_axpy(ctx, n, x, -acg, p)); // MPGPExpansion_ProjCG mpgp.c:335-349
        else PMH_CHK(u_expansion_std(s, afeas, xold, explengthvecold));
        if (s->expproject) PMH_CHK(pmh_qpc_box_project(ctx, n, x, s->lb, s->ub, x));
        if (fallback || s->o.fallback2) PMH_CHK(pmh_vec_copy(ctx, n, g, gold));
        PMH_CHK(s->A->mult(x, g));
        nmv++;
        PMH_CHK(pmh_vec_axpy(ctx, n, g, -1.0, s->b));
        if (fallback || s->o.fallback2) {
          PMH_CHK(u_objective_from_gradient(s, xold, gold, &fold));
          PMH_CHK(u_objective_from_gradient(s, x, g, &f));
          if (f > fold) {
            nfinc++;
            if (s->o.fallback2) {
              PMH_CHK(u_grads(s, x, g));
              PMH_CHK(pmh_vec_dot(ctx, n, gc, gc, &gcTgc));
              PMH_CHK(pmh_vec_dot(ctx, n, gf, gf, &gfTgf));
              fallback = (gcTgc <= gamma2 * gfTgf) ? 0 : 1;
            }
            if (fallback) {
              nfall++;
              s->step = 'f';
              PMH_CHK(pmh_vec_copy(ctx, n, xold, x));
              PMH_CHK(pmh_vec_copy(ctx, n, gold, g));
              if (s->o.fallback2) PMH_CHK(u_grads(s, xold, gold));
              PMH_CHK(u_expansion_std(s, afeas, xold, explengthvecold));
              PMH_CHK(pmh_qpc_box_project(ctx, n, x, s->lb, s->ub, x));
              PMH_CHK(s->A->mult(x, g));
              nmv++;
              PMH_CHK(pmh_vec_axpy(ctx, n, g, -1.0, s->b));
            }
          }
        }
        PMH_CHK(u_grads(s, x, g));
        PMH_CHK(pmh_vec_copy(ctx, n, gf, p));
      }
    } else {
      nprop++;
      s->step = 'p';
      PMH_CHK(pmh_vec_copy(ctx, n, gc, p));
      PMH_CHK(s->A->mult(p, Ap));
      nmv++;
      PMH_CHK(pmh_vec_dot(ctx, n, p, Ap, &pAp));
      PMH_CHK(pmh_vec_dot(ctx, n, g, p, &acg));
      acg = acg / pAp;
      PMH_CHK(pmh_vec_axpy(ctx, n, x, -acg, p));
      PMH_CHK(pmh_vec_axpy(ctx, n, g, -acg, Ap));
      PMH_CHK(u_grads(s, x, g));
      PMH_CHK(pmh_vec_copy(ctx, n, gf, p));
    }
    s->iteration++;
  }
  s->fallback_state = fallback;
  s->ncg += ncg, s->nexp += nexp, s->nmv += nmv, s->nprop += nprop, s->nfinc += nfinc, s->nfall += nfall;
  return PMH_SUCCESS;
}

// --------------------------------------------------------------------------------------------------------------------
// fused driver (std expansion, fixed step length, no fallback)
// --------------------------------------------------------------------------------------------------------------------
// P1: Ap = A p and the three reductions into d_scal/h_scal[S_PAP..S_FEAS]
static int f_apply_p1(pmh_mpgp s, const int *halt = nullptr, bool p_fresh = false)
{
  double *g = s->work[3], *p = s->work[4], *Ap = s->work[5];
  if (s->csr) {
    pmh_spmv_epi e;
    memset(&e, 0, sizeof(e));
    e.kind      = PMH_EPI_MPGP;
    e.g         = g;
    e.xx        = s->x;
    e.lb        = s->lb;
    e.ub        = s->ub;
    e.scal_base = S_PAP;
    e.halt      = halt;
    return pmh_csr_spmv_launch(s->csr, p, Ap, e);
  }
  if (halt) return pmh_set_error(PMH_ERR_STATE, "speculative chain needs a CSR operator");
  const int ops[3] = {PMH_RED_SUM, PMH_RED_SUM, PMH_RED_MIN};
  const int nb     = s->n > 0 ? pmh_vec_grid(s->n) : 0;
  // the three reductions inside the operator's last kernel (rows 4..6 of the partials: rows 0..3 may still hold a gradient split that waits for its finalising
  // launch)
  if (s->epi_ok) {
    pmh_vec_epi e;
    memset(&e, 0, sizeof(e));
    e.kind = PMH_VEPI_P1, e.g = g, e.xx = s->x, e.lb = s->lb, e.ub = s->ub, e.partials = s->ctx->d_partials, e.ld = s->ctx->partials_cap, e.prow = 4;
    e.p_fresh = p_fresh, e.spec_alpha = s->alpha, e.astol = s->o.astol; // (operators that pair their passes: svm.hip)
    int hosted = 0;
    // (the fused dual-space chain: G0 p left behind by the kernel that wrote p; the three sums' block partials go to the pinned host copy)
    e.in_slot = s->cx[1] ? 2 : 0, e.hosted = s->ctx->dist_scalars ? nullptr : &hosted;
    const int rc = s->A->mult_epi(p, Ap, e);
    if (rc != PMH_EPI_UNSUPPORTED) {
      PMH_CHK(rc);
      s->epi_ok = 1;
      if (hosted) {
        if (s->fin4_pending) PMH_CHK(finalize_vec4(s)); // (an operator that leaves its sums to the host does so for the gradient split too: not reached)
        s->fin4_pending = 0;
        s->hsum3        = hosted;
        return PMH_SUCCESS;
      }
      // (Ap'gf, |gP|^2, |gc|^2, |gf|^2) of the gradient split and (p'Ap, g'p, afeas) in ONE finalising launch: every quantity reduced as on its own
      if (s->fin4_pending) {
        const int ops7[7] = {PMH_RED_SUM, PMH_RED_SUM, PMH_RED_SUM, PMH_RED_SUM, PMH_RED_SUM, PMH_RED_SUM, PMH_RED_MIN};
        const int slots7[7] = {S_APGF, S_GP2, S_GC2, S_GF2, S_PAP, S_GP, S_FEAS};
        s->fin4_pending = 0;
        return pmh_finalize_partials_slots(s->ctx, s->ctx->d_partials, s->ctx->partials_cap, nb, 7, ops7, slots7);
      }
      return pmh_finalize_partials(s->ctx, s->ctx->d_partials + (size_t)4 * s->ctx->partials_cap, s->ctx->partials_cap, nb, 3, ops, S_PAP);
    }
    s->epi_ok = 0;
  }
  // (cannot happen with one operator: it answers mult_epi the same way every time) the split's partials sit in the rows k_p1_dots is about to overwrite
  if (s->fin4_pending) {
    PMH_CHK(finalize_vec4(s));
    s->fin4_pending = 0;
  }
  PMH_CHK(s->A->mult(p, Ap));
  LAUNCH(k_p1_dots, (const double *)p, (const double *)Ap, (const double *)g, (const double *)s->x, s->lb, s->ub, s->ctx->d_partials, s->ctx->partials_cap);
  return pmh_finalize_partials(s->ctx, s->ctx->d_partials, s->ctx->partials_cap, nb, 3, ops, S_PAP);
}

// g = A x - b (mpgp.c:500-502, :578-580)
static int f_gradient(pmh_mpgp s)
{
  double *g = s->work[3];
  if (s->csr) {
    pmh_spmv_epi e;
    memset(&e, 0, sizeof(e));
    e.kind = PMH_EPI_SUB;
    e.y1   = s->b;
    return pmh_csr_spmv_launch(s->csr, s->x, g, e);
  }
  PMH_CHK(s->A->mult(s->x, g));
  return pmh_vec_axpy(s->ctx, s->n, g, -1.0, s->b);
}

// g = A x - b, the gradient split with p = gf and the partials of its norms (mpgp.c:500-507, :578-580 + :612-615): inside the operator's last kernel where it
// offers that (the finalising launch then waits for the next P1: fin4_pending), else as three launches + the finalising one
static int f_gradient_split(pmh_mpgp s, bool defer_finalize, bool x_from_spec = false)
{
  pmh_ctx ctx = s->ctx;
  double *gf = s->work[1], *g = s->work[3], *p = s->work[4];
  if (s->epi_ok && !s->csr) {
    pmh_vec_epi e;
    memset(&e, 0, sizeof(e));
    e.kind = PMH_VEPI_GRAD_SPLIT, e.b = s->b, e.lb = s->lb, e.ub = s->ub, e.astol = s->o.astol, e.gf = gf, e.p = p, e.partials = ctx->d_partials, e.ld = ctx->partials_cap, e.prow = 0;
    e.x_from_spec = x_from_spec, e.x_out = s->x;
    int hosted = 0, p_emitted = 0;
    e.in_slot = s->cx[0] ? 1 : 0, e.hosted = ctx->dist_scalars ? nullptr : &hosted, e.emitted_p = &p_emitted;
    const int rc = s->A->mult_epi(s->x, g, e);
    if (rc != PMH_EPI_UNSUPPORTED) {
      PMH_CHK(rc);
      s->epi_ok = 1;
      s->cx[1]  = p_emitted != 0; // p = gf was written by the operator's last kernel
      if (hosted) {
        s->hsum4 = hosted;
        return PMH_SUCCESS;
      }
      if (defer_finalize && !ctx->dist_scalars) {
        s->fin4_pending = 1;
        return PMH_SUCCESS;
      }
      return finalize_vec4(s);
    }
    s->epi_ok = 0;
  }
  if (x_from_spec) return pmh_set_error(PMH_ERR_STATE, "pmh_mpgp: the operator prepared an expansion step and then refused the gradient");
  PMH_CHK(f_gradient(s));
  s->cx[1] = false;
  LAUNCH(k_split_setp<false>, (const double *)s->x, (const double *)g, s->lb, s->ub, s->o.astol, gf, p, ctx->d_partials, ctx->partials_cap, g_noea, (double *)nullptr);
  return finalize_vec4(s);
}

static int solve_fused(pmh_mpgp s)
{
  pmh_ctx ctx = s->ctx;
  int     n   = s->n;
  PMH_CHK(ensure_work(s, 1));
  PMH_CHK(ensure_work(s, 3));
  PMH_CHK(ensure_work(s, 4));
  PMH_CHK(ensure_work(s, 5));
  double      *gf = s->work[1], *g = s->work[3], *p = s->work[4], *Ap = s->work[5], *x = s->x;
  const double gamma2 = s->o.gamma * s->o.gamma, astol = s->o.astol;
  int          nmv = 0, ncg = 0, nprop = 0, nexp = 0;
  bool         spec = false; // P1 for the current p already enqueued
  double       prev_margin = 0.0; // cvg_margin of the previous test (see the speculation at the end of the loop)
  bool         p_fresh = false; // p is the gf of the last gradient split, untouched (told to operators that pair their passes)

  s->cx[0] = s->cx[1] = false; // x and p come from outside
  s->hsum4 = s->hsum3 = 0;
  s->A->emit_invalidate();
  PMH_CHK(pmh_qpc_box_project(ctx, n, x, s->lb, s->ub, x)); // mpgp.c:497
  s->fin4_pending = 0;
  // asked once: a refusal (PMH_EPI_UNSUPPORTED) clears it.  (Row-distributed vectors: the operator's
  if (s->epi_ok < 0) s->epi_ok = (s->csr || !pmh_knobs().vec_epi) ? 0 : 1;
                                                                                  // partials are finalised at once and completed across the ranks,
                                                                                  // pmh_finalize_partials)
  if (s->g_valid) { // the caller carried g = A x - b over from the previous solve (pmh_smalxe_set_reuse_products): the split, p = gf and the norms only
    s->g_valid = 0;
    pmh_emit_args ea;
    if (emit_try(s, false, true, &ea)) {
      LAUNCH_EMIT(k_split_setp<true>, (const double *)x, (const double *)g, s->lb, s->ub, astol, gf, p, ctx->d_partials, ctx->partials_cap, ea, ctx->h_partials);
      s->hsum4 = emit_blocks(s);
    } else {
      LAUNCH(k_split_setp<false>, (const double *)x, (const double *)g, s->lb, s->ub, astol, gf, p, ctx->d_partials, ctx->partials_cap, g_noea, (double *)nullptr);
      PMH_CHK(finalize_vec4(s));
    }
  } else {
    PMH_CHK(f_gradient_split(s, false)); // :500-507 (the host reads the norms before any P1: finalised at once)
    nmv++;
    p_fresh = true;
  }
  // (the carried-gradient branch did not run the operator's own gradient split: whatever pairing state an operator keeps from the previous solve must not be
  // matched with this p -- p_fresh stays false there, the first product of the solve is then a lone application)
  s->step      = ' ';
  s->iteration = 0;
  pmh_spec_args nosa;
  memset(&nosa, 0, sizeof(nosa));
  auto k_cg_spec = k_step_update<false, true>;
  auto k_cg_host = k_step_update<false, false>;
  auto k_prop_host = k_step_update<true, false>;
  auto k_cg_emit = k_step_update<false, false, true>;
  auto k_prop_emit = k_step_update<true, false, true>;
  // the device-side CG chain needs the default convergence test (its constants go to the kernels) and a CSR operator
  const int SPEC_BATCH = 16;
  bool      can_spec   = s->csr && !s->cvg && !s->o.distributed && pmh_knobs().mpgp_spec;
  pmh_spec_args sa     = nosa;
  if (can_spec) {
    if (!s->d_ctl) {
      PMH_CHK(pmh_malloc(ctx, sizeof(int) * CTL_NWORDS, (void **)&s->d_ctl));
      PMH_CHK(pmh_malloc(ctx, sizeof(double) * 3 * SPEC_BATCH, (void **)&s->d_ring));
      PMH_HIP(hipHostMalloc((void **)&s->h_ctl, sizeof(int) * CTL_NWORDS, hipHostMallocMapped));
      PMH_HIP(hipHostMalloc((void **)&s->h_ring, sizeof(double) * 3 * SPEC_BATCH, hipHostMallocMapped));
    }
    if (!s->cvg_setup) { // QPSConvergedDefaultSetUp qps.c:718-731
      PMH_CHK(pmh_vec_norm2(ctx, n, s->b, &s->norm_rhs));
      s->ttol         = fmax(s->o.rtol * s->norm_rhs, s->o.atol);
      s->norm_rhs_div = s->norm_rhs;
      s->cvg_setup    = 1;
    }
    sa.ctl = s->d_ctl, sa.ring = s->d_ring, sa.ring_cap = SPEC_BATCH;
    sa.max_it = s->o.max_it;
    sa.ttol = s->ttol, sa.divtol_rhs = s->o.divtol * s->norm_rhs_div, sa.gamma2 = gamma2;
  }
  // length of the next speculative batch: a batch that ran to its end doubles it, one that halted early cuts it to what it ran (an expansion-heavy stretch
  // would otherwise enqueue 16 x 4 no-op launches behind every expansion step: ~0.1 ms each time); 0 = the host takes the next step.  The iterates do not
  // depend on it: the device chain takes exactly the steps the host would.
  int spec_len = SPEC_BATCH;
  while (1) {
    if (can_spec && spec && spec_len > 0) {
      // a batch of CG steps decided on the device: no host round trip between them
      const int nbatch = spec_len;
      s->h_ctl[CTL_HALT] = 0, s->h_ctl[CTL_ITER] = s->iteration, s->h_ctl[CTL_NCG] = 0, s->h_ctl[CTL_BASE] = s->iteration;
      PMH_HIP(hipMemcpyAsync(s->d_ctl, s->h_ctl, sizeof(int) * CTL_NWORDS, hipMemcpyHostToDevice, ctx->stream));
      for (int j = 0; j < nbatch; j++) {
        LAUNCH(k_cg_spec, (const double *)ctx->d_scal, sa, x, g, p, (const double *)Ap, s->lb, s->ub, astol, gf, ctx->d_partials, ctx->partials_cap, g_noea, (double *)nullptr, 0.0, 0);
        PMH_CHK(finalize_vec4(s, s->d_ctl + CTL_HALT, s->d_ctl + CTL_ITER));
        LAUNCH(k_dir_update<false>, (const double *)ctx->d_scal, (const int *)(s->d_ctl + CTL_HALT), (const double *)gf, p, g_noea, (const double *)nullptr, 0, 0.0);
        PMH_CHK(f_apply_p1(s, s->d_ctl + CTL_HALT));
      }
      PMH_HIP(hipMemcpyAsync(s->h_ctl, s->d_ctl, sizeof(int) * CTL_NWORDS, hipMemcpyDeviceToHost, ctx->stream));
      PMH_HIP(hipMemcpyAsync(s->h_ring, s->d_ring, sizeof(double) * 3 * SPEC_BATCH, hipMemcpyDeviceToHost, ctx->stream));
      PMH_CHK(pmh_sync(ctx));
      const int ndev = s->h_ctl[CTL_ITER] - s->iteration;
      for (int j = 0; j < ndev; j++) {
        if (s->o.monitor) { // QPSMonitorDefault_MPGP line of iteration s->iteration + j
          s->t_step.push_back(s->step);
          s->t_gp.push_back(sqrt(s->h_ring[3 * j]));
          s->t_gf.push_back(sqrt(s->h_ring[3 * j + 1]));
          s->t_gc.push_back(sqrt(s->h_ring[3 * j + 2]));
          s->t_alpha.push_back(s->alpha);
        }
        s->step = 'c';
      }
      s->iteration += ndev;
      ncg += ndev;
      nmv += ndev;
      // 0 after a batch that took no step: the host path decides when speculation resumes
      spec_len = (ndev >= nbatch) ? std::min(SPEC_BATCH, 2 * nbatch) : ndev;
      if (!s->h_ctl[CTL_HALT]) continue; // whole batch were CG steps
    }
    if (s->pre_test) PMH_CHK(s->pre_test(s->pre_test_user));
    PMH_CHK(pmh_sync(ctx));
    host_sums(s);
    s->rnorm           = sqrt(ctx->h_scal[S_GP2]);
    const double gcTgc = ctx->h_scal[S_GC2], gfTgf = ctx->h_scal[S_GF2];
    s->gfnorm = sqrt(gfTgf);
    s->gcnorm = sqrt(gcTgc);
    PMH_CHK(test_convergence(s));
    if (s->reason != PMH_CONVERGED_ITERATING) break;

    if (gcTgc <= gamma2 * gfTgf) { // proportional (mpgp.c:535)
      if (!spec) {
        PMH_CHK(f_apply_p1(s, nullptr, p_fresh));
        PMH_CHK(pmh_sync(ctx));
        host_sums(s);
      }
      spec = false;
      nmv++;
      const double pAp = ctx->h_scal[S_PAP], acg = ctx->h_scal[S_GP] / pAp, afeas = ctx->h_scal[S_FEAS];
      if (acg <= afeas) { // CG step (mpgp.c:547-560)
        ncg++;
        s->step = 'c';
        spec_len = std::max(spec_len, 1); // a CG step taken by the host: the next ones may well be CG steps too
        pmh_emit_args ea;
        bool emitted_step = false;
        // x -= acg p with the segment sums of G0 x left behind; acg as the host has it; the four sums' block partials go to the pinned host copy
        if (emit_try(s, true, false, &ea)) {
          LAUNCH_EMIT(k_cg_emit, (const double *)ctx->d_scal, nosa, x, g, p, (const double *)Ap, s->lb, s->ub, astol, gf, ctx->d_partials, ctx->partials_cap, ea, ctx->h_partials, acg, 0);
          s->hsum4 = emit_blocks(s), emitted_step = true;
        } else {
          LAUNCH(k_cg_host, (const double *)ctx->d_scal, nosa, x, g, p, (const double *)Ap, s->lb, s->ub, astol, gf, ctx->d_partials, ctx->partials_cap, g_noea, (double *)nullptr, 0.0, 0);
          PMH_CHK(finalize_vec4(s));
        }
        if (emitted_step && emit_try(s, false, true, &ea)) {
          LAUNCH_EMIT(k_dir_update<true>, (const double *)ctx->d_scal, (const int *)nullptr, (const double *)gf, p, ea, (const double *)ctx->d_partials, emit_blocks(s), pAp);
        } else {
          if (emitted_step) PMH_CHK(finalize_vec4_from(s, emit_blocks(s))); // (not reached: an operator that took the step's emission takes the direction's)
          s->cx[1] = false;
          LAUNCH(k_dir_update<false>, (const double *)ctx->d_scal, (const int *)nullptr, (const double *)gf, p, g_noea, (const double *)nullptr, 0, 0.0);
        }
        p_fresh = false;
      } else { // expansion (mpgp.c:561-616), std direction + fixed length => no re-projection (:388)
        nexp++;
        s->step = 'e';
        // the operator's P1 pass already formed k_expansion_std's iterate (svm.hip): it hands it over with the gradient
        const bool prepared = s->epi_ok == 1 && s->A->spec_expansion_ready();
        if (!prepared) {
          pmh_emit_args ea;
          if (emit_try(s, true, false, &ea)) LAUNCH_EMIT(k_expansion_std<true>, afeas, s->alpha, x, (const double *)g, (const double *)p, (const double *)Ap, s->lb, s->ub, astol, ea);
          else LAUNCH(k_expansion_std<false>, afeas, s->alpha, x, (const double *)g, (const double *)p, (const double *)Ap, s->lb, s->ub, astol, g_noea);
        } else {
          s->cx[0] = false;
          s->A->emit_invalidate();
        }
        PMH_CHK(f_gradient_split(s, true, prepared)); // the speculative P1 below finalises both groups of partial sums
        p_fresh = true;
        nmv++;
      }
    } else { // proportioning (mpgp.c:617-639)
      nprop++;
      s->step = 'p';
      spec    = false; // a speculative P1 (if any) used the wrong direction; it is simply not counted
      p_fresh = false;
      pmh_emit_args ea;
      if (emit_try(s, false, true, &ea)) LAUNCH_EMIT(k_prop_dir<true>, (const double *)x, (const double *)g, s->lb, s->ub, astol, p, ea);
      else LAUNCH(k_prop_dir<false>, (const double *)x, (const double *)g, s->lb, s->ub, astol, p, g_noea);
      PMH_CHK(f_apply_p1(s));
      nmv++;
      // the product left its sums to the host: the step forms acg from the P1 block partials itself (no host wait in between)
      if (s->hsum3 && emit_try(s, true, true, &ea)) {
        LAUNCH_EMIT(k_prop_emit, (const double *)ctx->d_scal, nosa, x, g, p, (const double *)Ap, s->lb, s->ub, astol, gf, ctx->d_partials, ctx->partials_cap, ea, ctx->h_partials, 0.0, s->hsum3);
        s->hsum4 = emit_blocks(s);
      } else {
        // (not reached: the operator that left the sums to the host takes the emission) the step reads acg from d_scal: finalise the P1 rows on the device
        if (s->hsum3) {
          const int ops[3] = {PMH_RED_SUM, PMH_RED_SUM, PMH_RED_MIN};
          PMH_CHK(pmh_finalize_partials(ctx, ctx->d_partials + (size_t)4 * ctx->partials_cap, ctx->partials_cap, s->hsum3, 3, ops, S_PAP));
          s->hsum3 = 0;
        }
        s->cx[0] = s->cx[1] = false;
        LAUNCH(k_prop_host, (const double *)ctx->d_scal, nosa, x, g, p, (const double *)Ap, s->lb, s->ub, astol, gf, ctx->d_partials, ctx->partials_cap, g_noea, (double *)nullptr, 0.0, 0);
        PMH_CHK(finalize_vec4(s));
      }
    }
    // speculation: whatever the next step type, unless it is a proportioning step it starts with Ap = A p -- enqueued before the host has seen this step's
    // norms, so that the round trip costs nothing.  If the NEXT test ends the solve that product is wasted (0.35 ms for configs[2] against the ~30 us of an
    // exposed round trip): the test reports how far the norm is above its threshold (cvg_margin), and with the last step's reduction the driver skips the
    // speculation when the next norm is likely to pass (SMALXE's early outer iterations end their inner solves after 2-3 steps: 8 of 63 products in the
    // driver's 20-step window were such orphans). Nothing numerical depends on it: the product is then enqueued after the test, by the `!spec` branch above.
    bool orphan_risk = false;
    {
      const double m = s->cvg_margin;
      if (m > 0.0) {
        const double red = (prev_margin > 0.0 && m < prev_margin) ? m / prev_margin : 1.0;
        orphan_risk      = (m < 3.0) || (m * red < 2.0);
      }
      prev_margin = m;
    }
    if (orphan_risk) {
      spec = false;
      if (s->fin4_pending) { // the gradient split left its partial sums for the product's finalising launch: the host reads the norms next
        PMH_CHK(finalize_vec4(s));
        s->fin4_pending = 0;
      }
    } else {
      if (s->pre_p1) PMH_CHK(s->pre_p1(s->pre_p1_user));
      PMH_CHK(f_apply_p1(s, nullptr, p_fresh));
      spec = true;
    }
    s->iteration++;
  }
  s->ncg += ncg, s->nexp += nexp, s->nmv += nmv, s->nprop += nprop;
  return PMH_SUCCESS;
}

extern "C" int pmh_mpgp_solve(pmh_mpgp s)
{
  PMH_ARG(s);
  s->t_step.clear(), s->t_gp.clear(), s->t_gf.clear(), s->t_gc.clear(), s->t_alpha.clear();
  s->ctx->dist_scalars = s->o.distributed;
  int rc               = use_fused(s) ? solve_fused(s) : solve_unfused(s);
  s->ctx->dist_scalars = 0;
  return rc;
}

extern "C" int pmh_mpgp_get_tolerances(pmh_mpgp s, double *rtol, double *atol, double *divtol, int *max_it) // QPSGetTolerances
{
  PMH_ARG(s);
  if (rtol) *rtol = s->o.rtol;
  if (atol) *atol = s->o.atol;
  if (divtol) *divtol = s->o.divtol;
  if (max_it) *max_it = s->o.max_it;
  return PMH_SUCCESS;
}

extern "C" int pmh_mpgp_get_stats(pmh_mpgp s, pmh_mpgp_stats *st)
{
  PMH_ARG(s && st);
  memset(st, 0, sizeof(*st));
  st->iteration = s->iteration, st->reason = s->reason;
  st->rnorm = s->rnorm, st->gfnorm = s->gfnorm, st->gcnorm = s->gcnorm, st->alpha = s->alpha, st->maxeig = s->maxeig;
  st->nmv = s->nmv, st->ncg = s->ncg, st->nexp = s->nexp, st->nprop = s->nprop, st->nfinc = s->nfinc, st->nfall = s->nfall;
  st->norm_rhs = s->norm_rhs, st->ttol = s->ttol;
  st->current_step_type = s->step;
  return PMH_SUCCESS;
}

extern "C" int pmh_mpgp_get_trace(pmh_mpgp s, int cap, char *step, double *gp, double *gf, double *gc, double *alpha, int *len)
{
  PMH_ARG(s && len);
  int m = (int)s->t_step.size();
  *len  = m;
  if (m > cap) m = cap;
  for (int i = 0; i < m; i++) {
    if (step) step[i] = s->t_step[i];
    if (gp) gp[i] = s->t_gp[i];
    if (gf) gf[i] = s->t_gf[i];
    if (gc) gc[i] = s->t_gc[i];
    if (alpha) alpha[i] = s->t_alpha[i];
  }
  return PMH_SUCCESS;
}

// work[0..6] = gP, gf, gc, g, p, Ap, gr (mpgp.c:6-17).  The fused driver does not keep gP, gc, gr:
// they are recomputed here from the final x, g (MPGPGrads) so that callers see the reference's work vectors.
// internal (an injected convergence test says how close it is to ending the solve, see solve_fused's speculation)
int pmh_mpgp_set_convergence_margin(pmh_mpgp s, double margin)
{
  PMH_ARG(s);
  s->cvg_margin = margin;
  return PMH_SUCCESS;
}

// internal (smalxe.hip): the next pmh_mpgp_solve may take work[3] as the gradient at its starting point (fused driver only; others ignore it)
int pmh_mpgp_set_gradient_valid(pmh_mpgp s, int valid, double **g)
{
  PMH_ARG(s);
  s->g_valid = (valid && use_fused(s) && s->work[3]) ? 1 : 0;
  if (g) *g = s->work[3];
  return PMH_SUCCESS;
}

extern "C" int pmh_mpgp_get_work(pmh_mpgp s, int idx, const double **dptr)
{
  PMH_ARG(s && dptr && idx >= 0 && idx < 10);
  if (use_fused(s) && (idx == 0 || idx == 2 || idx == 6)) {
    PMH_ARG(s->work[3]);
    PMH_CHK(ensure_work(s, 0));
    PMH_CHK(ensure_work(s, 2));
    PMH_CHK(ensure_work(s, 6));
    PMH_CHK(ensure_work(s, 1));
    PMH_CHK(u_grads(s, s->x, s->work[3]));
  }
  PMH_ARG(s->work[idx]);
  *dptr = s->work[idx];
  return PMH_SUCCESS;
}
