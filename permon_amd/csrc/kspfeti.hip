// KSPFETI (src/ksp/impls/feti/feti.c): the reference's KSP facade that solves a decomposed linear problem K u = f by (T)FETI --
// KSPFETISetUp (:71-94: QPTMatISToBlockDiag -> QPFetiSetDirichlet -> QPFetiSetUp -> QPTFromOptions "-feti" = dualize + project)
// and KSPSolve_FETI (:144-156: QPSSolve on the last QP of the chain, then the post-solve chain).  This is the same driver over
// the pieces of this library, for the input the reference has AFTER QPTMatISToBlockDiag: the subdomain blocks K_i, the right-hand
// side already split among the copies of shared dofs, the local-to-global dof map, the Dirichlet dofs and the kernel bases R_i.
//   B  = [B_d ; B_g]: one row per Dirichlet dof copy with a single 1 (QPFetiAssembleDirichlet qpfeti.c:153-312, enforce_by_B), then
//        the gluing rows of QPFetiGetBgtSF (pmh_feti_gluing_from_l2g);
//   K^+: MATINV on K_reg = MatRegularize(K, R) (-regularize 1, the default) or P_R K^- P_R (-regularize 0 -qpt_dualize_Kplus_mp);
//   G  = R'B' (explicit, qptransform.c:838), e = R'f;  chain: pmh_qpt_feti_chain_create (dualize, homogenize, project);
//        or, with -qpt_dualize_Kplus_left (what the reference switches to when it computed the kernel itself, qptransform.c:997-1008), K^- P_R;
//   QPS: the dual QP has no box, so QPSSetDefaultType picks QPSKSP = CG on P F (qps.c:448) with PC none or P (B K B') (PCDUAL lumped);
//        with -project 0 the equality constraint stays, the QP is homogenised and QPSSetDefaultType picks SMALXE (qps.c:437-441), optionally on orthonormalised
//          G
//        (-dual_qp_E_orth_type gs | implicit);
//   post-solve: lambda = lambda_child + lambda~, u = K^+(f - B' lambda) - R alpha with G' alpha = d - F lambda (qptransform.c:783-833).
// Host orchestration in C++; every operator application runs on the device.
#include <algorithm>
#include <cmath>
#include <map>
#include <string>

#include "pmh_internal.h"

namespace {
struct LumpedOp : pmh_op_s { // PCApply_Dual lumped (pcdual.c:63-78) as an operator
  pmh_gluing    B;
  pmh_blockdiag K;
  int           mult(const double *x, double *y) override { return pmh_pc_dual_lumped_apply(B, K, x, y); }
};
} // namespace

// QPChainPostSolve's report (src/qp/interface/qpchain.c:198-275) for the chain KSPFETI builds (feti.c:71-94, QPTAllInOne qptransform.c:2152-2207):
//   #0 the assembled MATIS QP -> #1 QPTMatISToBlockDiag -> #2 QPTScale -> #3 QPTDualize -> #4 QPTScale [-> #5 QPTHomogenizeEq -> #6 QPTEnforceEqByProjector
//     with floating subdomains].
// QPTScale adds a child even when nothing is scaled (:1459, QP_DUPLICATE_COPY_POINTERS: operator, right-hand side, solution and multipliers are the parent's
// own objects), so #4 / #3 and #2 / #1 print the same lines.  Every QP is viewed after the post-solve of the QP below it.  -qpt_matis_to_diag_norm adds the
// line of QPTPostSolve_QPTMatISToBlockDiag (:1954-1979) between #2 and #1; with the Dirichlet dofs enforced by B that routine zeroes rows / columns of the
// local matrices IN PLACE (:1933 -- the MATIS of #0 shares them) and restores them with MatCopy (:1967), which for MATBLOCKDIAG is PETSc's MatCopy_Basic:
// MatZeroEntries + a row loop the type has no MatGetRow for -- what #1 and #0 print afterwards is consistent with an operator left at ZERO
// (feti/output/ex1_1.out: ||B' lambda - f|| = 2.31e-02 and ||b|| / ||b|| = 1.00e+00; with -dir_in_hess, ex1_2.out, nothing is touched).  Reproduced as
// observed, said here so that nobody mistakes those two lines for residuals of the solve.
static int kspfeti_view_smalxe(pmh_ctx ctx, const pmh_kspfeti_opts *o, pmh_smalxe S, pmh_feti_chain ch, const pmh_feti_chain_kkt &k, pmh_qppf pf0, const double *e0, int nl, int m,
                               const double *d_x, const double *d_lam, std::string &text);

// S != NULL: the dual QP was solved unprojected by SMALXE (-project 0); pf0 / e0: the projector of the plain G = R'B' and e = R'f
static int kspfeti_view(pmh_ctx ctx, const pmh_kspfeti_opts *o, const pmh_pcpg_stats &ks, pmh_feti_chain ch, pmh_blockdiag Kb, int N, int nsub, const int *l2g, int n_dir, const int *dir_local,
                        const double *f, const double *u_host, const double *d_x, const double *d_lam, double *d_u, pmh_smalxe S, pmh_qppf pf0, const double *e0, int nl, int m,
                        std::string &text)
{
  (void)nsub;
  char line[256];
  if (o->view_convergence) { // QPSViewConvergence qps.c:1188-1230 (KSPConvergedReasons: 2 RTOL, 3 ATOL, -3 ITS, -4 DTOL)
    const char *why = ks.reason == 2 ? "CONVERGED_RTOL" : ks.reason == 3 ? "CONVERGED_ATOL" : ks.reason == 1 ? "CONVERGED_RTOL_NORMAL" : ks.reason == 5 ? "CONVERGED_HAPPY_BREAKDOWN"
                      : ks.reason == -3 ? "DIVERGED_ITS" : ks.reason == -4 ? "DIVERGED_DTOL" : ks.reason == -9 ? "DIVERGED_NANORINF" : "DIVERGED_BREAKDOWN";
    snprintf(line, sizeof(line), "  last QPSSolve %s due to %s, KSPReason=%d, required %d iterations\n", ks.reason > 0 ? "CONVERGED" : "DIVERGED", why, ks.reason, ks.iteration);
    text += line;
  }
  if (!o->view_kkt && !o->matis_to_diag_norm) return PMH_SUCCESS;
  PMH_CHK(pmh_memcpy_h2d(ctx, d_u, u_host, sizeof(double) * (size_t)N)); // u with its rigid-body part
  pmh_feti_chain_kkt k;
  PMH_CHK(pmh_qpt_feti_chain_kkt(ch, Kb, d_x, d_lam, d_u, &k));
  auto kkt = [&](const char *name, double r, double nb) { // qp.c:296
    snprintf(line, sizeof(line), "r = ||%s|| = %.2e    rO/||b|| = %.2e\n", name, r, r / nb);
    text += line;
  };
  auto be = [&](bool with_c, double r, double nb) { // qp.c:310-312
    snprintf(line, sizeof(line), with_c ? "r = ||BE*x-cE||          = %.2e    r/||b|| = %.2e\n" : "r = ||BE*x||             = %.2e    r/||b|| = %.2e\n", r, r / nb);
    text += line;
  };
  if (o->view_kkt && S) {
    PMH_CHK(kspfeti_view_smalxe(ctx, o, S, ch, k, pf0, e0, nl, m, d_x, d_lam, text));
    kkt("A*x - b + B'*lambda", k.prim_r, k.prim_normb);
    be(false, k.prim_be, k.prim_normb);
  } else if (o->view_kkt) {
    if (k.has_coarse) {
      kkt("A*x - b", k.proj_r, k.proj_normb);                 // #6
      kkt("A*x - b + (B'*lambda)", k.hom_r, k.hom_normb);     // #5
      be(false, k.hom_be, k.hom_normb);
    }
    for (int rep = 0; rep < 2; rep++) {                       // #4, #3
      kkt(k.has_coarse ? "A*x - b + (B'*lambda)" : "A*x - b", k.dual_r, k.dual_normb);
      if (k.has_coarse) be(true, k.dual_be, k.dual_normb);
    }
    kkt("A*x - b + B'*lambda", k.prim_r, k.prim_normb);       // #2
    be(false, k.prim_be, k.prim_normb);
  }
  // ---- the assembled problem: x0 = INSERT_VALUES assembly of u (:1948-1952), A0 = sum of the local matrices, b0 = the assembled right-hand side
  int ng = 0;
  for (int i = 0; i < N; i++) ng = std::max(ng, l2g[i] + 1);
  std::vector<double> x0((size_t)ng, 0.0), b0((size_t)ng, 0.0), xl((size_t)N), yl((size_t)N), r0((size_t)ng, 0.0);
  for (int i = 0; i < N; i++) x0[l2g[i]] = u_host[i], b0[l2g[i]] += f[i];
  auto assembled_residual = [&](bool zero_dirichlet, double *rn, double *bn) -> int {
    // r = A x0 - b: A applied through the local matrices (MatMult_IS: scatter, local products, add); zero_dirichlet: the Dirichlet rows / columns zeroed with a
    // unit diagonal and b_dir = 0 (MatZeroRowsColumnsIS(child->A, isDir, 1.0, dir = 0, b) :1926-1933)
    std::vector<char> isd((size_t)N, 0);
    if (zero_dirichlet)
      for (int q = 0; q < n_dir; q++) isd[dir_local[q]] = 1;
    for (int i = 0; i < N; i++) xl[i] = isd[i] ? 0.0 : x0[l2g[i]];
    PMH_CHK(pmh_memcpy_h2d(ctx, d_u, xl.data(), sizeof(double) * (size_t)N));
    double *d_y = nullptr;
    PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)std::max(N, 1), (void **)&d_y));
    int rc = pmh_blockdiag_mult(Kb, d_u, d_y);
    if (!rc) rc = pmh_memcpy_d2h(ctx, yl.data(), d_y, sizeof(double) * (size_t)N);
    pmh_free(ctx, d_y);
    PMH_CHK(rc);
    std::fill(r0.begin(), r0.end(), 0.0);
    std::vector<double> bb((size_t)ng, 0.0);
    for (int i = 0; i < N; i++) {
      if (isd[i]) r0[l2g[i]] += x0[l2g[i]]; // unit diagonal
      else r0[l2g[i]] += yl[i], bb[l2g[i]] += f[i];
    }
    double sr = 0.0, sb = 0.0;
    for (int g = 0; g < ng; g++) sr += (r0[g] - bb[g]) * (r0[g] - bb[g]), sb += bb[g] * bb[g];
    *rn = std::sqrt(sr), *bn = std::sqrt(sb);
    return PMH_SUCCESS;
  };
  const bool zeroed = o->matis_to_diag_norm && n_dir > 0; // see the comment above this function
  if (o->matis_to_diag_norm) {
    double rn = 0.0, bn = 0.0;
    PMH_CHK(assembled_residual(n_dir > 0, &rn, &bn));
    snprintf(line, sizeof(line), "Dirichlet in Hess: %d, r = ||Ax-b|| = %e, r/||b|| = %e\n", n_dir > 0 ? 0 : 1, rn, rn / bn);
    text += line;
  }
  if (o->view_kkt) {
    kkt("A*x - b + B'*lambda", zeroed ? k.prim_r_zeroed_operator : k.prim_r, k.prim_normb); // #1
    be(false, k.prim_be, k.prim_normb);
    double rn = 0.0, bn = 0.0;
    PMH_CHK(assembled_residual(false, &rn, &bn));
    kkt("A*x - b", zeroed ? bn : rn, bn);                                                  // #0
  }
  return PMH_SUCCESS;
}


// MatOrthRows with MAT_ORTH_GS, explicit form (permonmatorth.c:208-236 MatOrthColumns_GS_Default on the columns of G'): classical Gram-Schmidt -- all the dots
// against the finished rows at once -- repeated while the norm has dropped to half or less; T collects the same row operations (TG0 = G, Te0 = e).  Host, dense
// m x n rows.
static int orth_rows_gs(int m, int n, std::vector<double> &G, std::vector<double> &T)
{
  T.assign((size_t)m * m, 0.0);
  for (int i = 0; i < m; i++) T[(size_t)i * m + i] = 1.0;
  std::vector<double> dots((size_t)std::max(m, 1));
  auto nrm2 = [&](const double *v) {
    double t = 0.0;
    for (int k = 0; k < n; k++) t += v[k] * v[k];
    return std::sqrt(t);
  };
  for (int i = 0; i < m; i++) {
    double *q = &G[(size_t)i * n], norm = nrm2(q), norm_last;
    do {
      norm_last = norm;
      for (int j = 0; j < i; j++) {
        const double *qj = &G[(size_t)j * n];
        double        t  = 0.0;
        for (int k = 0; k < n; k++) t += q[k] * qj[k];
        dots[j] = -t;
      }
      for (int j = 0; j < i; j++) {
        const double *qj = &G[(size_t)j * n];
        for (int k = 0; k < n; k++) q[k] += dots[j] * qj[k];
        for (int k = 0; k < m; k++) T[(size_t)i * m + k] += dots[j] * T[(size_t)j * m + k];
      }
      norm = nrm2(q);
      if (!(norm >= 1e2 * 2.220446049250313e-16)) return pmh_set_error(PMH_ERR_ARG, "pmh_kspfeti_solve: the rows 0 - %d of G are linearly dependent", i);
    } while (norm <= 0.5 * norm_last);
    for (int k = 0; k < n; k++) q[k] /= norm;
    for (int k = 0; k < m; k++) T[(size_t)i * m + k] /= norm;
  }
  return PMH_SUCCESS;
}

// MAT_ORTH_GS_LINGEN (permonmatorth.c:248-288 MatOrthColumns_GS_Lingen): the same projections, but the norm of the projected row is NOT recomputed -- it
// follows from Pythagoras, delta = delta_last sqrt|1 - ||p||^2 / delta_last^2| with p the dots just subtracted --, the row is re-projected while delta <=
// delta_last / 2, and it is scaled by 1 / delta
static int orth_rows_gs_lingen(int m, int n, std::vector<double> &G, std::vector<double> &T)
{
  T.assign((size_t)m * m, 0.0);
  for (int i = 0; i < m; i++) T[(size_t)i * m + i] = 1.0;
  std::vector<double> p((size_t)m + 1);
  for (int k = 0; k < m; k++) {
    double *q = &G[(size_t)k * n];
    auto    dots = [&](int upto) {
      for (int j = 0; j < upto; j++) {
        const double *qj = &G[(size_t)j * n];
        double        t  = 0.0;
        for (int c = 0; c < n; c++) t += q[c] * qj[c];
        p[j] = t;
      }
    };
    dots(k + 1); // p[k] = q_k . q_k
    double delta_last = std::sqrt(p[k]), delta;
    for (;;) {
      for (int j = 0; j < k; j++) {
        const double *qj = &G[(size_t)j * n];
        for (int c = 0; c < n; c++) q[c] -= p[j] * qj[c];
        for (int c = 0; c < m; c++) T[(size_t)k * m + c] -= p[j] * T[(size_t)j * m + c];
      }
      double pp = 0.0;
      for (int j = 0; j < k; j++) pp += p[j] * p[j];
      const double beta = 1.0 - pp / (delta_last * delta_last);
      delta             = delta_last * std::sqrt(std::fabs(beta));
      if (!(delta >= 1e2 * 2.220446049250313e-16)) return pmh_set_error(PMH_ERR_ARG, "pmh_kspfeti_solve: the rows 0 - %d of G are linearly dependent", k);
      if (delta > 0.5 * delta_last) break;
      dots(k);
      delta_last = delta;
    }
    for (int c = 0; c < n; c++) q[c] /= delta;
    for (int c = 0; c < m; c++) T[(size_t)k * m + c] /= delta;
  }
  return PMH_SUCCESS;
}

// The -qp_chain_view_kkt lines of the QPs that differ when the dual QP is NOT projected (-project 0), last QP first (QPTAllInOne qptransform.c:2178-2207):
//   QPTEnforceEqByPenalty (SMALXE's inner QP: A + rho G'G, b - B'mu), QPTHomogenizeEq (multiplier term B'mu, which QPSSolve_SMALXE leaves in Bt_lambda),
//     [QPTOrthonormalizeEq: the
//   same multiplier term -- QPTHomogenizeEqPostSolve copies it up -- against d; ||BE x - cE|| only when BE can be multiplied with, i.e. not for the implicit
//     type, qp.c:303-318],
//   then QPTScale / QPTDualize's child twice with G0, e0 and the multiplier term QPViewKKT computes itself for a QP that was handed none
//     (QPTPostSolve_QPTOrthonormalizeEq skips
//   both multipliers): r = 0 by construction.  Without orthonormalisation the last two print the homogenised QP's multiplier term as in the projected chain.
static int kspfeti_view_smalxe(pmh_ctx ctx, const pmh_kspfeti_opts *o, pmh_smalxe S, pmh_feti_chain ch, const pmh_feti_chain_kkt &k, pmh_qppf pf0, const double *e0, int nl, int m,
                               const double *d_x, const double *d_lam, std::string &text)
{
  char    line[256];
  pmh_op  F = nullptr, Arho = nullptr;
  double *d = nullptr, *bbar = nullptr, *b_in = nullptr, *btmu = nullptr, *t = nullptr, *gm = nullptr;
  PMH_CHK(pmh_qpt_feti_chain_get(ch, &F, nullptr, &d, &bbar, nullptr, nullptr, nullptr));
  PMH_CHK(pmh_smalxe_get_penalized(S, &Arho, &b_in, &btmu));
  PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)std::max(nl, 1), (void **)&t));
  if (const int rc_gm = pmh_malloc(ctx, sizeof(double) * (size_t)std::max(m, 1), (void **)&gm)) {
    pmh_free(ctx, t);
    return rc_gm;
  }
  int    rc = PMH_SUCCESS;
  double r7 = 0, nb7 = 0, r6 = 0, r5 = 0, be4 = 0;
#define GO(call) \
  do { \
    if ((rc = (call))) goto done; \
  } while (0)
  GO(pmh_op_mult(Arho, d_x, t));
  GO(pmh_vec_axpy(ctx, nl, t, -1.0, b_in));
  GO(pmh_vec_norm2(ctx, nl, t, &r7));
  GO(pmh_vec_norm2(ctx, nl, b_in, &nb7));
  GO(pmh_op_mult(F, d_x, t));
  GO(pmh_vec_axpy(ctx, nl, t, -1.0, bbar));
  GO(pmh_vec_axpy(ctx, nl, t, 1.0, btmu));
  GO(pmh_vec_norm2(ctx, nl, t, &r6));
  GO(pmh_op_mult(F, d_lam, t));
  GO(pmh_vec_axpy(ctx, nl, t, -1.0, d));
  GO(pmh_vec_axpy(ctx, nl, t, 1.0, btmu));
  GO(pmh_vec_norm2(ctx, nl, t, &r5));
  {
    std::vector<double> g((size_t)std::max(m, 1));
    GO(pmh_qppf_apply_G(pf0, d_lam, gm));
    GO(pmh_memcpy_d2h(ctx, g.data(), gm, sizeof(double) * (size_t)m));
    for (int i = 0; i < m; i++) be4 += (g[i] - e0[i]) * (g[i] - e0[i]);
    be4 = std::sqrt(be4);
  }
  {
    auto kkt = [&](const char *name, double r, double nb) {
      snprintf(line, sizeof(line), "r = ||%s|| = %.2e    rO/||b|| = %.2e\n", name, r, r / nb);
      text += line;
    };
    auto be = [&](bool with_c, double r, double nb) {
      snprintf(line, sizeof(line), with_c ? "r = ||BE*x-cE||          = %.2e    r/||b|| = %.2e\n" : "r = ||BE*x||             = %.2e    r/||b|| = %.2e\n", r, r / nb);
      text += line;
    };
    kkt("A*x - b", r7, nb7);                                  // penalised
    kkt("A*x - b + (B'*lambda)", r6, k.hom_normb);            // homogenised
    be(false, k.hom_be, k.hom_normb);
    if (o->E_orth_type) {                                     // orthonormalised
      kkt("A*x - b + (B'*lambda)", r5, k.dual_normb);
      if (o->E_orth_type == 4) text += "r = ||BE*x-cE||         not available\n";
      else be(true, k.dual_be, k.dual_normb);
    }
    for (int rep = 0; rep < 2; rep++) {                       // QPTScale's child, QPTDualize's child
      kkt("A*x - b + (B'*lambda)", o->E_orth_type ? 0.0 : r5, k.dual_normb);
      be(true, be4, k.dual_normb);
    }
  }
done:
#undef GO
  pmh_free(ctx, t), pmh_free(ctx, gm);
  return rc;
}

extern "C" int pmh_kspfeti_default_opts(pmh_kspfeti_opts *o)
{
  PMH_ARG(o);
  memset(o, 0, sizeof(*o));
  o->gluing_type = 1;   // FETI_GLUING_FULL, qpfeti.c:322
  o->scale       = 1;   // -SCALE_ON, qpfeti.c:757
  o->regularize  = 1;   // QPTFromOptions qptransform.c:2215
  // KSPFETI never hands QPTDualize a kernel (feti.c:71-94), so the reference computes one and switches to K^- P_R, -regularize 0 (qptransform.c:997-1008)
  o->kplus_left  = 1;
  o->project     = 1;   // -feti (qptransform.c:2224)
  PMH_CHK(pmh_smalxe_default_opts(&o->smalxe));
  o->kplus_rtol = 1e-12, o->kplus_max_it = 20000; // a stand-in for "direct": the reference factorises K_reg
  o->rtol = 1e-5, o->atol = 1e-50, o->divtol = 1e4, o->max_it = 10000; // QPSCreate qps.c:73-76
  o->explicit_dual = 0, o->explicit_rtol = 1e-13;
  return PMH_SUCCESS;
}

extern "C" int pmh_kspfeti_solve(pmh_ctx ctx, int nsub, const int *block_rowstart, const int *rowptr, const int *col, const double *val, const double *f, const int *l2g, int n_dir,
                                 const int *dir_local, int kdim, const double *R, const pmh_kspfeti_opts *o, double *u_host, double *lambda_host, int lambda_cap, pmh_kspfeti_stats *st)
{
  PMH_ARG(ctx && nsub >= 1 && block_rowstart && rowptr && f && l2g && o && u_host && st && kdim >= 0 && kdim <= 8 && (kdim == 0 || R) && (n_dir == 0 || dir_local));
  const int N = block_rowstart[nsub];
  PMH_ARG(block_rowstart[0] == 0 && N >= 0 && (rowptr[N] == 0 || (col && val)));
  memset(st, 0, sizeof(*st));
  auto block_of = [&](int i) { return (int)(std::upper_bound(block_rowstart, block_rowstart + nsub + 1, i) - block_rowstart) - 1; };

  // ---- kernel bases: block-wise Gram-Schmidt (QPTDualize orthonormalises a computed R, qptransform.c:1001); which blocks float
  std::vector<double> Rn((size_t)kdim * N, 0.0); // row k = k-th kernel vector over all blocks (layout of pmh_matinv_set_nullspace)
  std::vector<int>    bdim(nsub, 0);             // kernel dimension per block (leading non-zero vectors)
  for (int s = 0; s < nsub; s++) {
    const int lo = block_rowstart[s], hi = block_rowstart[s + 1];
    int       d  = 0;
    for (int k = 0; k < kdim; k++) {
      std::vector<double> v(R + (size_t)k * N + lo, R + (size_t)k * N + hi);
      double              nrm0 = 0.0;
      for (double x : v) nrm0 += x * x;
      if (nrm0 == 0.0) continue;
      for (int pass = 0; pass < 2; pass++) // twice is enough
        for (int j = 0; j < d; j++) {
          const double *q = &Rn[(size_t)j * N + lo];
          double        t = 0.0;
          for (int i = 0; i < hi - lo; i++) t += q[i] * v[i];
          for (int i = 0; i < hi - lo; i++) v[i] -= t * q[i];
        }
      double nrm = 0.0;
      for (double x : v) nrm += x * x;
      if (nrm <= 1e-24 * nrm0) return pmh_set_error(PMH_ERR_ARG, "pmh_kspfeti_solve: the kernel vectors of block %d are linearly dependent", s);
      nrm = std::sqrt(nrm);
      for (int i = 0; i < hi - lo; i++) Rn[(size_t)d * N + lo + i] = v[i] / nrm;
      d++;
    }
    bdim[s] = d;
  }

  // ---- B = [B_d ; B_g] as leaves
  std::vector<int>    lrow, lroot;
  std::vector<double> lval;
  for (int i = 0; i < n_dir; i++) {
    PMH_ARG(dir_local[i] >= 0 && dir_local[i] < N);
    lrow.push_back(dir_local[i]), lroot.push_back(i), lval.push_back(1.0);
  }
  std::vector<int> excl;
  if (o->exclude_dirichlet) { // -feti_gluing_exclude_dirichlet (qpfeti.c:423-431): the global dofs of the Dirichlet set
    for (int i = 0; i < n_dir; i++) excl.push_back(l2g[dir_local[i]]);
    std::sort(excl.begin(), excl.end());
    excl.erase(std::unique(excl.begin(), excl.end()), excl.end());
  }
  int ng = 0, nleaf = 0;
  PMH_CHK(pmh_feti_gluing_from_l2g(nsub, block_rowstart, l2g, o->gluing_type, o->scale, (int)excl.size(), excl.data(), &ng, &nleaf, nullptr, nullptr, nullptr));
  {
    std::vector<int>    r((size_t)nleaf), t((size_t)nleaf);
    std::vector<double> v((size_t)nleaf);
    PMH_CHK(pmh_feti_gluing_from_l2g(nsub, block_rowstart, l2g, o->gluing_type, o->scale, (int)excl.size(), excl.data(), &ng, &nleaf, r.data(), t.data(), v.data()));
    for (int i = 0; i < nleaf; i++) lrow.push_back(r[i]), lroot.push_back(n_dir + t[i]), lval.push_back(v[i]);
  }
  const int nl = n_dir + ng;
  st->n_lambda = nl, st->n_dirichlet_rows = n_dir;
  if (lambda_host) PMH_ARG(lambda_cap >= nl);

  // ---- device objects
  pmh_csr        Kc = nullptr, Kregc = nullptr, Gc = nullptr;
  pmh_blockdiag  Kb = nullptr, Kregb = nullptr;
  pmh_matinv     Kp = nullptr;
  pmh_mg         mg = nullptr; // -dual_mat_inv_pc_type gamg
  pmh_gluing     B  = nullptr;
  pmh_qppf       pf = nullptr, pfo = nullptr; // of G = R'B'; of the orthonormalised G (-dual_qp_E_orth_type)
  pmh_csr        Goc = nullptr;
  pmh_smalxe     S  = nullptr;
  pmh_feti_chain ch = nullptr;
  pmh_fexplicit  E  = nullptr;
  LumpedOp      *lump = nullptr;
  pmh_op         pc = nullptr;
  double        *d_f = nullptr, *d_c = nullptr, *d_e = nullptr, *d_x = nullptr, *d_lam = nullptr, *d_u0 = nullptr, *d_r = nullptr, *d_alpha = nullptr;
  int            rc = PMH_SUCCESS;
#define GO(call) \
  do { \
    if ((rc = (call))) goto done; \
  } while (0)
  {
    GO(pmh_csr_create(ctx, N, N, rowptr, col, val, &Kc));
    GO(pmh_blockdiag_create(ctx, nsub, block_rowstart, Kc, &Kb));
    const bool any_kernel = std::any_of(bdim.begin(), bdim.end(), [](int d) { return d > 0; });
    // (the explicit local dual operators store symmetric blocks: they stay on K_reg^{-1} / the Moore-Penrose form)
    if (o->kplus_left && !o->explicit_dual && any_kernel) {
      // K^+ = K^- P_R (qptransform.c:1040-1062).  The reference's K^- is the MUMPS solve with null-pivot detection (MatInvComputeNullSpace's factorisation):
      // the null pivots carry 0. Which dofs those are is MUMPS's choice; here they are the fixing dofs MatRegularize would take (permonmatregularize.c:57-124),
      // one set per floating block, eliminated from K by identity rows / columns -- for feti/ex1.c that is an end dof of every subdomain, and the reference's
      // outputs are reproduced with it (tests/test_gpu_feti_kkt_text.py).
      std::vector<int>  fix;
      std::vector<char> isfix((size_t)N, 0);
      for (int s = 0; s < nsub; s++) {
        const int lo = block_rowstart[s], hi = block_rowstart[s + 1], p = hi - lo, d = bdim[s];
        if (!d) continue;
        std::vector<double> Rb((size_t)d * p);
        std::vector<int>    piv((size_t)d);
        for (int k = 0; k < d; k++) std::copy(&Rn[(size_t)k * N + lo], &Rn[(size_t)k * N + hi], &Rb[(size_t)k * p]);
        GO(pmh_mat_regularize_pivots(p, d, Rb.data(), piv.data()));
        for (int k = 0; k < d; k++) fix.push_back(lo + piv[k]), isfix[lo + piv[k]] = 1;
      }
      std::vector<int>    rp((size_t)N + 1, 0), ci;
      std::vector<double> va;
      ci.reserve((size_t)rowptr[N]), va.reserve((size_t)rowptr[N]);
      for (int i = 0; i < N; i++) {
        if (isfix[i]) ci.push_back(i), va.push_back(1.0);
        else
          for (int k = rowptr[i]; k < rowptr[i + 1]; k++)
            if (!isfix[col[k]]) ci.push_back(col[k]), va.push_back(val[k]);
        rp[i + 1] = (int)ci.size();
      }
      GO(pmh_csr_create(ctx, N, N, rp.data(), ci.data(), va.data(), &Kregc)); // (the handles of the regularised K serve the fixed one)
      GO(pmh_blockdiag_create(ctx, nsub, block_rowstart, Kregc, &Kregb));
      GO(pmh_matinv_create(Kregb, o->kplus_rtol, 1e-50, o->kplus_max_it, 1, &Kp));
      GO(pmh_matinv_set_nullspace(Kp, kdim, Rn.data()));
      GO(pmh_matinv_set_left_inverse(Kp, (int)fix.size(), fix.data()));
    } else if (o->regularize && any_kernel) {
      // MatRegularize block by block (permonmatregularize.c:241-266 works on the rank's diagonal block)
      std::vector<int>    rp((size_t)N + 1, 0), ci;
      std::vector<double> va;
      ci.reserve((size_t)rowptr[N] + 64 * nsub), va.reserve((size_t)rowptr[N] + 64 * nsub);
      for (int s = 0; s < nsub; s++) {
        const int lo = block_rowstart[s], hi = block_rowstart[s + 1], p = hi - lo, d = bdim[s], z0 = rowptr[lo], nz = rowptr[hi] - z0;
        std::vector<int>    brp((size_t)p + 1), bci((size_t)nz), orp((size_t)p + 1), oci((size_t)nz + d * d), piv((size_t)std::max(d, 1));
        std::vector<double> ova((size_t)nz + d * d), Rb((size_t)d * p);
        for (int i = 0; i <= p; i++) brp[i] = rowptr[lo + i] - z0;
        for (int k = 0; k < nz; k++) {
          bci[k] = col[z0 + k] - lo;
          if (bci[k] < 0 || bci[k] >= p) {
            rc = pmh_set_error(PMH_ERR_ARG, "pmh_kspfeti_solve: K is not block diagonal (row block %d)", s);
            goto done;
          }
        }
        for (int k = 0; k < d; k++) std::copy(&Rn[(size_t)k * N + lo], &Rn[(size_t)k * N + hi], &Rb[(size_t)k * p]);
        double rho = o->regularize_rho;
        if (d && !(rho > 0.0)) { // rho = MatGetMaxEigenvalue(K_loc, NULL, &rho, 1, 20) (:254)
          pmh_csr Kblk = nullptr;
          pmh_op  op   = nullptr;
          GO(pmh_csr_create(ctx, p, p, brp.data(), bci.data(), val + z0, &Kblk));
          rc = pmh_op_create_csr(Kblk, &op);
          if (!rc) rc = pmh_op_max_eigenvalue(op, 1.0, 20, &rho, nullptr);
          pmh_op_destroy(op), pmh_csr_destroy(Kblk);
          if (rc) goto done;
        }
        long long onz = 0;
        GO(pmh_mat_regularize_csr(p, brp.data(), bci.data(), val + z0, d, Rb.data(), rho, piv.data(), orp.data(), oci.data(), ova.data(), &onz));
        for (int i = 0; i < p; i++) rp[lo + i + 1] = rp[lo + i] + (orp[i + 1] - orp[i]);
        for (long long k = 0; k < onz; k++) ci.push_back(oci[k] + lo), va.push_back(ova[k]);
      }
      GO(pmh_csr_create(ctx, N, N, rp.data(), ci.data(), va.data(), &Kregc));
      GO(pmh_blockdiag_create(ctx, nsub, block_rowstart, Kregc, &Kregb));
      GO(pmh_matinv_create(Kregb, o->kplus_rtol, 1e-50, o->kplus_max_it, 1, &Kp));
    } else {
      GO(pmh_matinv_create(Kb, o->kplus_rtol, 1e-50, o->kplus_max_it, 1, &Kp));
      if (any_kernel) GO(pmh_matinv_set_nullspace(Kp, kdim, Rn.data()));
    }
    if (o->kplus_pc == 1) {
      // -dual_mat_inv_pc_type gamg: the algebraic V-cycle on the matrix this MATINV inverts.  The regularised / fixed matrices are non-singular: the rigid-body modes go in
      // as the near-kernel of non-singular blocks; the Moore-Penrose form keeps them as the kernel.  Single precision cycles where the levels have 3 x 3 blocks
      int ndof = o->kplus_pc_ndof;
      if (ndof <= 0) {
        ndof = (kdim == 6) ? 3 : 1;
        for (int s = 0; s < nsub && ndof == 3; s++)
          if ((block_rowstart[s + 1] - block_rowstart[s]) % 3 || (bdim[s] != 6 && bdim[s] != 0)) ndof = 1;
      }
      const bool    singular = !Kregc;
      const pmh_csr Kpc      = Kregc ? Kregc : Kc;
      std::vector<int>    hrp, hci;
      std::vector<double> hva;
      const int          *prp = rowptr, *pci = col;
      const double       *pva = val;
      if (Kregc) { // (host copy of the matrix the PC is built on: the regularised / fixed one was assembled above and lives on the device)
        hrp.resize((size_t)N + 1), hci.resize((size_t)Kregc->nnz), hva.resize((size_t)Kregc->nnz);
        GO(pmh_memcpy_d2h(ctx, hrp.data(), Kregc->d_rowptr, sizeof(int) * hrp.size()));
        GO(pmh_memcpy_d2h(ctx, hci.data(), Kregc->d_col, sizeof(int) * hci.size()));
        GO(pmh_memcpy_d2h(ctx, hva.data(), Kregc->d_val, sizeof(double) * hva.size()));
        prp = hrp.data(), pci = hci.data(), pva = hva.data();
      }
      bool blocks3 = ndof == 3;
      if (blocks3 && pmh_matinv_enable_bsr3(Kp)) blocks3 = false; // (no 3 x 3 block structure: the fp64 cycle on the CSR kernels)
      const int prec = blocks3 ? PMH_MG_FP16 : PMH_MG_FP64;
      rc = pmh_mg_create_sa(ctx, Kpc, nsub, block_rowstart, ndof, prp, pci, pva, singular ? kdim : 0, singular ? Rn.data() : nullptr, singular ? 0 : kdim, singular ? nullptr : Rn.data(), 1500, 0.08, 2, prec, &mg);
      if (rc && prec != PMH_MG_FP64) rc = pmh_mg_create_sa(ctx, Kpc, nsub, block_rowstart, ndof, prp, pci, pva, singular ? kdim : 0, singular ? Rn.data() : nullptr, singular ? 0 : kdim, singular ? nullptr : Rn.data(), 1500, 0.08, 2, PMH_MG_FP64, &mg);
      if (rc) goto done;
      GO(pmh_matinv_set_pc_mg(Kp, mg));
    }
    GO(pmh_gluing_create(ctx, N, nl, (int)lrow.size(), lrow.data(), lroot.data(), lval.data(), &B));
    if (o->explicit_dual) { // the exact K^+ path: W_b = (K_b^+)[Gamma_b, Gamma_b] by one K^+ solve per column, then F = Bhat W Bhat'
      GO(pmh_fexplicit_create(B, Kregb ? Kregb : Kb, PMH_FX_SYM, &E));
      GO(pmh_fexplicit_assemble_auto(E, Kp, nullptr, nullptr, o->explicit_rtol, 0, nullptr));
      GO(pmh_matinv_attach_explicit(Kp, E));
    }

    // ---- G = R'B' (rows: the kernel vectors of the floating blocks), e = R'f
    std::vector<int> grow0(nsub + 1, 0);
    for (int s = 0; s < nsub; s++) grow0[s + 1] = grow0[s] + bdim[s];
    const int m = grow0[nsub];
    st->coarse_dim = m;
    std::vector<double> e((size_t)std::max(m, 1), 0.0), eo((size_t)std::max(m, 1), 0.0); // e = R'f; T e
    if (m) {
      std::vector<std::map<int, double>> rows((size_t)m);
      for (size_t q = 0; q < lrow.size(); q++) {
        const int s = block_of(lrow[q]);
        for (int k = 0; k < bdim[s]; k++) {
          const double w = Rn[(size_t)k * N + lrow[q]] * lval[q];
          if (w != 0.0) rows[grow0[s] + k][lroot[q]] += w;
        }
      }
      std::vector<int>    grp((size_t)m + 1, 0), gci;
      std::vector<double> gva;
      for (int r = 0; r < m; r++) {
        for (auto &kv : rows[r]) gci.push_back(kv.first), gva.push_back(kv.second);
        grp[r + 1] = (int)gci.size();
      }
      for (int s = 0; s < nsub; s++)
        for (int k = 0; k < bdim[s]; k++) {
          double t = 0.0;
          for (int i = block_rowstart[s]; i < block_rowstart[s + 1]; i++) t += Rn[(size_t)k * N + i] * f[i];
          e[grow0[s] + k] = t;
        }
      GO(pmh_csr_create(ctx, m, nl, grp.data(), gci.data(), gva.data(), &Gc));
      GO(pmh_qppf_create(ctx, Gc, 0, &pf));
      // QPTOrthonormalizeEq, MAT_ORTH_GS / MAT_ORTH_GS_LINGEN: the child QP gets the explicit T G and T e (qptransform.c:593-606)
      if (o->E_orth_type == 1 || o->E_orth_type == 2) {
        if ((double)m * nl > 5e7) {
          rc = pmh_set_error(PMH_ERR_SUP, "pmh_kspfeti_solve: -dual_qp_E_orth_type gs forms the dense %d x %d G on the host; use implicit", m, nl);
          goto done;
        }
        std::vector<double> Gd((size_t)m * nl, 0.0), T;
        for (int r = 0; r < m; r++)
          for (auto &kv : rows[r]) Gd[(size_t)r * nl + kv.first] = kv.second;
        GO(o->E_orth_type == 1 ? orth_rows_gs(m, nl, Gd, T) : orth_rows_gs_lingen(m, nl, Gd, T));
        std::vector<int>    orp((size_t)m + 1, 0), oci;
        std::vector<double> ova;
        for (int r = 0; r < m; r++) {
          for (int c = 0; c < nl; c++)
            if (Gd[(size_t)r * nl + c] != 0.0) oci.push_back(c), ova.push_back(Gd[(size_t)r * nl + c]);
          orp[r + 1] = (int)oci.size();
        }
        for (int r = 0; r < m; r++) {
          double t = 0.0;
          for (int c = 0; c < m; c++) t += T[(size_t)r * m + c] * e[c];
          eo[r] = t;
        }
        GO(pmh_csr_create(ctx, m, nl, orp.data(), oci.data(), ova.data(), &Goc));
        GO(pmh_qppf_create(ctx, Goc, 0, &pfo)); // QPSetEq re-creates the QPPF from T G (qptransform.c:609): GG' is formed and inverted like any other
      } else if (o->E_orth_type == 4 || o->E_orth_type == 3) {
        // MAT_ORTH_IMPLICIT: G stays, the projector carries T (:612-619).  MAT_ORTH_CHOLESKY in its (default) implicit form: BE = L^{-1} G as a product of the
        // forward solve and G (permonmatorth.c:121-128) -- the same object here; unlike the dummy BE of the implicit TYPE it can be multiplied with, so its
        // ||BE x - cE|| line is printed
        GO(pmh_qppf_create(ctx, Gc, 2, &pfo));
        GO(pmh_qppf_orth_rhs(pfo, e.data(), eo.data()));
      }
    }
    pmh_qppf pfs = pfo ? pfo : pf; // the equality constraint of the QP that gets solved

    // ---- chain, solve, post-solve
    const size_t bl = sizeof(double) * (size_t)std::max(nl, 1), bx = sizeof(double) * (size_t)std::max(N, 1);
    GO(pmh_malloc(ctx, bx, (void **)&d_f));
    GO(pmh_malloc(ctx, bl, (void **)&d_c));
    GO(pmh_malloc(ctx, sizeof(double) * (size_t)std::max(m, 1), (void **)&d_e));
    GO(pmh_malloc(ctx, bl, (void **)&d_x));
    GO(pmh_malloc(ctx, bl, (void **)&d_lam));
    GO(pmh_malloc(ctx, bx, (void **)&d_u0));
    GO(pmh_malloc(ctx, bl, (void **)&d_r));
    GO(pmh_malloc(ctx, sizeof(double) * (size_t)std::max(m, 1), (void **)&d_alpha));
    GO(pmh_memcpy_h2d(ctx, d_f, f, sizeof(double) * (size_t)N));
    GO(pmh_memset(ctx, d_c, 0, bl)); // c = B x0 with x0 = 0 (qpfeti.c:268-270)
    GO(pmh_memset(ctx, d_x, 0, bl));
    if (m) GO(pmh_memcpy_h2d(ctx, d_e, (pfo ? eo : e).data(), sizeof(double) * (size_t)m));
    GO(pmh_qpt_feti_chain_create(B, Kp, d_f, d_c, pfs, m ? d_e : nullptr, nullptr, &ch));
    pmh_op  A;
    double *b;
    GO(pmh_qpt_feti_chain_get(ch, nullptr, &A, nullptr, nullptr, &b, nullptr, nullptr));
    if (o->lumped_pc) { // PCDUAL lumped, projected as qptransform.c:301-309 does for an equality-only QP
      lump      = new LumpedOp();
      lump->ctx = ctx, lump->n = nl, lump->B = B, lump->K = Kb;
      if (pfs) GO(pmh_op_create_projected(lump, pfs, 0, &pc));
    }
    pmh_pcpg_stats ks;
    memset(&ks, 0, sizeof(ks));
    if (m && !o->project) {
      // -project 0: QPTAllInOne skips QPTEnforceEqByProjector (qptransform.c:2185), QPSSetDefaultType sees the equality constraint and picks SMALXE
      // (qps.c:437-441), whose set-up homogenises (smalxe.c:800-806: the chain's lambda~ / b_bar) and penalises.  The inner QP has no box: the reference's
      // default inner solver is then QPSKSP = CG (qps.c:448) under SMALXE's stopping rule; this library's inner solver is MPGP with no bounds, whose steps are
      // all CG steps -- the same iteration.
      if (o->lumped_pc) {
        rc = pmh_set_error(PMH_ERR_SUP, "pmh_kspfeti_solve: -dual_pc_dual_type lumped with -project 0 is not built");
        goto done;
      }
      pmh_op  F    = nullptr;
      double *bbar = nullptr;
      GO(pmh_qpt_feti_chain_get(ch, &F, nullptr, nullptr, &bbar, nullptr, nullptr, nullptr));
      pmh_smalxe_opts sx = o->smalxe;
      sx.rtol = o->rtol, sx.atol = o->atol, sx.divtol = o->divtol;
      // -qps_max_it given (max_it_set: also an explicit 10000): QPSCreate_SMALXE's own default (100, smalxe.c:1203) otherwise
      if (o->max_it_set || o->max_it != 10000) sx.max_it = o->max_it;
      if (o->E_orth_type == 4) sx.be_implicit = 1;   // the implicit BE has no product of its own: ||BE u|| through B'B (smalxe.c:878-886)
      GO(pmh_smalxe_create(ctx, F, bbar, d_x, nullptr, nullptr, pfs, &sx, &S));
      GO(pmh_smalxe_solve(S));
      GO(pmh_smalxe_get_stats(S, &st->smalxe));
      ks.iteration = st->smalxe.iteration, ks.reason = st->smalxe.reason, ks.rnorm = st->smalxe.rnorm;
    } else {
      GO(pmh_ksp_cg_solve(ctx, A, b, d_x, o->lumped_pc ? (pc ? pc : (pmh_op)lump) : nullptr, o->rtol, o->atol, o->divtol, o->max_it, &ks));
    }
    st->iteration = ks.iteration, st->reason = ks.reason, st->rnorm = ks.rnorm;
    GO(pmh_qpt_feti_chain_post_solve(ch, d_x, d_lam, d_u0, d_r));
    GO(pmh_memcpy_d2h(ctx, u_host, d_u0, sizeof(double) * (size_t)N));
    if (lambda_host) GO(pmh_memcpy_d2h(ctx, lambda_host, d_lam, sizeof(double) * (size_t)nl));
    if (m) { // u = u0 - R alpha, G' alpha = d - F lambda  =>  alpha = -(G G')^{-1} G (F lambda - d)
      std::vector<double> alpha((size_t)m);
      GO(pmh_qppf_apply_halfQ(pf, d_r, d_alpha));
      GO(pmh_memcpy_d2h(ctx, alpha.data(), d_alpha, sizeof(double) * (size_t)m));
      for (int s = 0; s < nsub; s++)
        for (int k = 0; k < bdim[s]; k++) {
          const double a = -alpha[grow0[s] + k];
          for (int i = block_rowstart[s]; i < block_rowstart[s + 1]; i++) u_host[i] -= Rn[(size_t)k * N + i] * a;
        }
    }
    if (o->view_convergence || o->view_kkt || o->matis_to_diag_norm) {
      std::string text;
      if (pfo && o->view_kkt && !S) {
        rc = pmh_set_error(PMH_ERR_SUP, "pmh_kspfeti_solve: -qp_chain_view_kkt for the projected chain with an orthonormalised G is not built");
        goto done;
      }
      GO(kspfeti_view(ctx, o, ks, ch, Kb, N, nsub, l2g, n_dir, dir_local, f, u_host, d_x, d_lam, d_u0, S, pf, e.data(), nl, m, text));
      if (o->view_buf && o->view_cap > 0) snprintf(o->view_buf, (size_t)o->view_cap, "%s", text.c_str());
      else fputs(text.c_str(), stdout), fflush(stdout);
    }
  }
done:
#undef GO
  pmh_free(ctx, d_f), pmh_free(ctx, d_c), pmh_free(ctx, d_e), pmh_free(ctx, d_x), pmh_free(ctx, d_lam), pmh_free(ctx, d_u0), pmh_free(ctx, d_r), pmh_free(ctx, d_alpha);
  if (pc) pmh_op_destroy(pc);
  if (lump) delete lump;
  pmh_smalxe_destroy(S);
  pmh_qpt_feti_chain_destroy(ch);
  if (Kp && E) pmh_matinv_attach_explicit(Kp, nullptr);
  pmh_fexplicit_destroy(E);
  pmh_qppf_destroy(pfo), pmh_qppf_destroy(pf);
  pmh_gluing_destroy(B);
  pmh_matinv_destroy(Kp);
  pmh_mg_destroy(mg);
  pmh_blockdiag_destroy(Kregb), pmh_blockdiag_destroy(Kb);
  pmh_csr_destroy(Goc), pmh_csr_destroy(Gc), pmh_csr_destroy(Kregc), pmh_csr_destroy(Kc);
  return rc;
}

// ---- QPTMatISToBlockDiag, vector part (src/qp/interface/qptransform.c:2007-2150 and its post-solve :1905-1982) -----------------
// Decomposing the assembled right-hand side: every interface dof's value is divided by the number of subdomains it belongs to
// (the scaling "matrix" D = 1/counter, :2095-2104) and then copied to all its copies (:2105-2113).  Host routine.
extern "C" int pmh_qpt_matis_split_rhs(int N, const int *l2g, int n_global, const double *b_global, double *f_local)
{
  PMH_ARG(N >= 0 && n_global >= 0 && (N == 0 || (l2g && b_global && f_local)));
  std::vector<int> mult((size_t)n_global, 0);
  for (int i = 0; i < N; i++) {
    if (l2g[i] < 0 || l2g[i] >= n_global) return pmh_set_error(PMH_ERR_ARG, "pmh_qpt_matis_split_rhs: l2g[%d] = %d out of [0,%d)", i, l2g[i], n_global);
    mult[l2g[i]]++;
  }
  for (int i = 0; i < N; i++) f_local[i] = b_global[l2g[i]] / (double)mult[l2g[i]];
  return PMH_SUCCESS;
}

// QPTPostSolve_QPTMatISToBlockDiag (:1945-1949): the global solution is assembled from the local ones by a reverse scatter with
// INSERT_VALUES -- one copy of every shared dof wins (here: the one in the highest-numbered subdomain, what a rank-ordered
// scatter delivers last), nothing is averaged.  Pinned by the last KKT line of the ex71 goldens (tests/test_feti_goldens.py).
extern "C" int pmh_qpt_matis_assemble_solution(int N, const int *l2g, const double *u_local, int n_global, double *x_global)
{
  PMH_ARG(N >= 0 && n_global >= 0 && (N == 0 || (l2g && u_local)) && (n_global == 0 || x_global));
  for (int g = 0; g < n_global; g++) x_global[g] = 0.0;
  for (int i = 0; i < N; i++) {
    if (l2g[i] < 0 || l2g[i] >= n_global) return pmh_set_error(PMH_ERR_ARG, "pmh_qpt_matis_assemble_solution: l2g[%d] = %d out of [0,%d)", i, l2g[i], n_global);
    x_global[l2g[i]] = u_local[i];
  }
  return PMH_SUCCESS;
}

// QPTMatISToBlockDiag, matrix side (qptransform.c:2007-2150): the MATIS operator -- one unassembled local matrix per subdomain plus its
// local-to-global map -- becomes the MATBLOCKDIAG of the child QP (MatCreateBlockDiag(comm, matis->A), :2044-2046), and the mapping
// yields what the rest of the transform needs: matis->counter (how many subdomains share a dof: the scaling D = 1/counter of the
// right-hand side, :2095-2104), the interface / interior split of the local dofs (PetscBT over the shared nodes, :2057-2071) and
// i2g, the sorted global numbers of the interface dofs (:2123-2125, QPFetiSetInterfaceToGlobalMapping).  Host routine (set-up, integers).
// loc_rowptr: the subdomains' CSR row pointers one after the other (n_s + 1 entries each, every block starting at 0), loc_col local
// column indices, loc_val values; out: the block-diagonal CSR in the concatenated local numbering (rowptr N + 1, col / val nnz).
extern "C" int pmh_qpt_matis_to_blockdiag(int nsub, const int *l2g_start, const int *l2g, int n_global, const int *loc_rowptr, const int *loc_col, const double *loc_val, int *block_rowstart,
                                          int *rowptr, int *col, double *val, int *counter, int *is_interface, int *n_i2g, int *i2g)
{
  PMH_ARG(nsub >= 1 && l2g_start && l2g && n_global >= 0 && loc_rowptr && block_rowstart && rowptr && n_i2g);
  const int N = l2g_start[nsub];
  PMH_ARG(l2g_start[0] == 0);
  std::vector<int> mult((size_t)std::max(1, n_global), 0);
  for (int i = 0; i < N; i++) {
    if (l2g[i] < 0 || l2g[i] >= n_global) return pmh_set_error(PMH_ERR_ARG, "pmh_qpt_matis_to_blockdiag: l2g[%d] = %d out of [0,%d)", i, l2g[i], n_global);
    mult[l2g[i]]++;
  }
  long long knz = 0; // running non-zero offset of the block diagonal
  int       rp0 = 0; // running offset into loc_rowptr
  rowptr[0] = 0;
  for (int s = 0; s < nsub; s++) {
    const int lo = l2g_start[s], n = l2g_start[s + 1] - lo;
    block_rowstart[s] = lo;
    const int *rp = loc_rowptr + rp0;
    if (rp[0] != 0) return pmh_set_error(PMH_ERR_ARG, "pmh_qpt_matis_to_blockdiag: the row pointers of subdomain %d do not start at 0", s);
    for (int i = 0; i < n; i++) {
      for (int k = rp[i]; k < rp[i + 1]; k++) {
        const int c = loc_col[knz + k - 0];
        if (c < 0 || c >= n) return pmh_set_error(PMH_ERR_ARG, "pmh_qpt_matis_to_blockdiag: subdomain %d, row %d: column %d out of [0,%d)", s, i, c, n);
        if (col) col[knz + k] = c + lo;
        if (val) val[knz + k] = loc_val[knz + k];
      }
      rowptr[lo + i + 1] = (int)(knz + rp[i + 1]);
    }
    knz += rp[n];
    rp0 += n + 1;
  }
  block_rowstart[nsub] = N;
  std::vector<int> inter;
  for (int i = 0; i < N; i++) {
    const int m = mult[l2g[i]];
    if (counter) counter[i] = m;
    if (is_interface) is_interface[i] = m > 1;
  }
  for (int g = 0; g < n_global; g++)
    if (mult[g] > 1) inter.push_back(g);
  *n_i2g = (int)inter.size();
  if (i2g) std::copy(inter.begin(), inter.end(), i2g);
  return PMH_SUCCESS;
}
