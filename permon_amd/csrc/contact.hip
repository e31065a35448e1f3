// One C entry point for a TFETI CONTACT problem -- what QPTFromOptions / QPTAllInOne (src/qp/interface/qptransform.c:2152-2237) +
// QPSSolve + the post-solve chain do for a decomposed QP with equality (gluing, Dirichlet) and inequality (non-penetration) rows:
//   QPTDualize (:1102-1174)  F = B K^+ B', d = B K^+ f - c, G = R'B', e = R'f, dual box lambda_I >= 0 (:1136-1162)
//   QPTOrthonormalizeEq      G <- L^{-1} G, e <- L^{-1} e with G G' = L L'   (optional, the bench default; implicitly by default: pmh_qppf_create)
//   QPTHomogenizeEq (:437-527), QPTEnforceEqByProjector (:215-316)          A = P F P, b = P (d - F lambda~)
//   QPSSetDefaultType (qps.c:443-444): equality constraint present -> SMALXE with inner MPGP
//   post-solve: lambda = lambda_child + lambda~ (:423-431); u = K^+(f - B' lambda) + R alpha (:783-833)
// over the device operators of this library.  K^+ is the Moore-Penrose wrapped MATINV (-regularize 0 -qpt_dualize_Kplus_mp) with the
// box-multigrid PC built by pmh_mg_create_box when the blocks' node boxes are given, and -- the fast exact path -- the explicit local
// dual operators (pmh_fexplicit_*).  Host orchestration in C++; every operator application runs on the device.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <map>
#include <thread>

#include "pmh_internal.h"

extern "C" int pmh_feti_contact_default_opts(pmh_feti_contact_opts *o)
{
  PMH_ARG(o);
  memset(o, 0, sizeof(*o));
  PMH_CHK(pmh_smalxe_default_opts(&o->smalxe)); // outer rtol 1e-5, max_it 100 (smalxe.c:1203), inner MPGP defaults
  o->kplus_rtol = 1e-9, o->kplus_max_it = 20000;
  o->mg = 1, o->mg_min_nodes = 400, o->mg_degree = 2, o->mg_precision = PMH_MG_FP16;
  o->bsr3 = 1;
  o->explicit_dual = 1, o->explicit_rtol = 1e-12, o->explicit_storage = PMH_FX_SYM, o->explicit_symmetry = 1;
  o->orthonormalize = 2; // implicit form (the reference's default form): G = R'B' keeps its sparsity
  return PMH_SUCCESS;
}

// dense SPD solve helper: in place Cholesky of the m x m row-major matrix (lower triangle); returns non-zero if not SPD
static int chol_lower(int m, std::vector<double> &A)
{
  for (int j = 0; j < m; j++) {
    double d = A[(size_t)j * m + j];
    for (int p = 0; p < j; p++) d -= A[(size_t)j * m + p] * A[(size_t)j * m + p];
    if (!(d > 0.0)) return 1;
    d                     = std::sqrt(d);
    A[(size_t)j * m + j] = d;
    for (int i = j + 1; i < m; i++) {
      double s = A[(size_t)i * m + j];
      for (int p = 0; p < j; p++) s -= A[(size_t)i * m + p] * A[(size_t)j * m + p];
      A[(size_t)i * m + j] = s / d;
    }
  }
  return 0;
}

extern "C" int pmh_feti_contact_solve(pmh_ctx ctx, int nsub, const int *block_rowstart, const int *rowptr, const int *col, const double *val, const double *f, int n_lambda, int n_eq, int n_leaves,
                                      const int *leaves_row, const int *leaves_root, const double *leaves_val, const double *c, int kdim, const double *R, const int *dims, int ndof,
                                      const pmh_feti_contact_opts *o, double *u_host, double *lambda_host, pmh_feti_contact_stats *st)
{
  PMH_ARG(ctx && nsub >= 1 && block_rowstart && rowptr && col && val && f && o && u_host && st && n_lambda >= 1 && n_eq >= 0 && n_eq <= n_lambda);
  PMH_ARG(n_leaves >= 0 && (n_leaves == 0 || (leaves_row && leaves_root && leaves_val)) && c && kdim >= 1 && kdim <= 8 && R);
  const int N = block_rowstart[nsub], nl = n_lambda;
  PMH_ARG(block_rowstart[0] == 0 && N >= 1);
  memset(st, 0, sizeof(*st));
  st->n_lambda = nl, st->n_eq = n_eq;
  auto t_start  = std::chrono::steady_clock::now();
  auto       t_last = t_start;
  const bool verbose = getenv("PMH_CONTACT_TIMING") != nullptr;
  auto       stage  = [&](const char *what) { // PMH_CONTACT_TIMING=1: where the set-up time goes (stderr)
    if (!verbose) return;
    hipDeviceSynchronize();
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "pmh_feti_contact_solve: %-44s %7.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
    t_last = now;
  };
  auto block_of = [&](int i) { return (int)(std::upper_bound(block_rowstart, block_rowstart + nsub + 1, i) - block_rowstart) - 1; };
  // K must be block diagonal over block_rowstart (MATBLOCKDIAG: one sequential block per subdomain, matblockdiag.c:787-801): the per-class CSR views below index the
  // block's own columns only.  One pass over the column indices (threads over blocks), an off-block entry is an argument error.
  {
    std::vector<int> bad(nsub, -1);
    std::vector<std::thread> th;
    for (int s = 0; s < nsub; s++)
      th.emplace_back([&, s]() {
        const int lo = block_rowstart[s], hi = block_rowstart[s + 1];
        if (hi < lo) { bad[s] = lo; return; }
        for (int i = lo; i < hi && bad[s] < 0; i++)
          for (int k = rowptr[i]; k < rowptr[i + 1]; k++)
            if (col[k] < lo || col[k] >= hi) { bad[s] = i; break; }
      });
    for (auto &t : th) t.join();
    for (int s = 0; s < nsub; s++)
      if (bad[s] >= 0) return pmh_set_error(PMH_ERR_ARG, "pmh_feti_contact_solve: K is not block diagonal over block_rowstart (row %d of block %d has a column outside [%d, %d))", bad[s], s, block_rowstart[s], block_rowstart[s + 1]);
  }

  // ---- kernel bases: block-wise Gram-Schmidt (QPTDualize orthonormalises R, qptransform.c:1001)
  std::vector<double> Rn((size_t)kdim * N, 0.0);
  std::vector<int>    bdim(nsub, 0);
  std::vector<int> gs_bad(nsub, 0);
  auto gram_schmidt = [&](int s) { // one host thread per block (8 x 6 vectors of 255 552 entries: 0.06 s on one thread)
    const int lo = block_rowstart[s], hi = block_rowstart[s + 1];
    int       d  = 0;
    for (int k = 0; k < kdim; k++) {
      std::vector<double> v(R + (size_t)k * N + lo, R + (size_t)k * N + hi);
      double              nrm0 = 0.0;
      for (double x : v) nrm0 += x * x;
      if (nrm0 == 0.0) continue;
      for (int pass = 0; pass < 2; pass++)
        for (int j = 0; j < d; j++) {
          const double *q = &Rn[(size_t)j * N + lo];
          double        t = 0.0;
          for (int i = 0; i < hi - lo; i++) t += q[i] * v[i];
          for (int i = 0; i < hi - lo; i++) v[i] -= t * q[i];
        }
      double nrm = 0.0;
      for (double x : v) nrm += x * x;
      if (nrm <= 1e-24 * nrm0) {
        gs_bad[s] = 1;
        return;
      }
      nrm = std::sqrt(nrm);
      for (int i = 0; i < hi - lo; i++) Rn[(size_t)d * N + lo + i] = v[i] / nrm;
      d++;
    }
    bdim[s] = d;
  };
  {
    std::vector<std::thread> th;
    for (int s = 0; s < nsub; s++) th.emplace_back(gram_schmidt, s);
    for (auto &t : th) t.join();
    for (int s = 0; s < nsub; s++)
      if (gs_bad[s]) return pmh_set_error(PMH_ERR_ARG, "pmh_feti_contact_solve: the kernel vectors of block %d are linearly dependent", s);
  }
  std::vector<int> grow0(nsub + 1, 0);
  for (int s = 0; s < nsub; s++) grow0[s + 1] = grow0[s] + bdim[s];
  const int m = grow0[nsub];
  if (m == 0) return pmh_set_error(PMH_ERR_SUP, "pmh_feti_contact_solve: no floating subdomain -- the dual QP has no equality constraint; use pmh_mpgp_* on F directly");
  st->coarse_dim = m;

  // ---- G = R'B' (dense on the host: m x n_lambda, m = a few dozen), e = R'f; optional QPTOrthonormalizeEq
  std::vector<double> Gd((size_t)m * nl, 0.0), e((size_t)m, 0.0), L;
  for (int q = 0; q < n_leaves; q++) {
    PMH_ARG(leaves_row[q] >= 0 && leaves_row[q] < N && leaves_root[q] >= 0 && leaves_root[q] < nl);
    const int s = block_of(leaves_row[q]);
    for (int k = 0; k < bdim[s]; k++) Gd[(size_t)(grow0[s] + k) * nl + leaves_root[q]] += Rn[(size_t)k * N + leaves_row[q]] * leaves_val[q];
  }
  for (int s = 0; s < nsub; s++)
    for (int k = 0; k < bdim[s]; k++) {
      double t = 0.0;
      for (int i = block_rowstart[s]; i < block_rowstart[s + 1]; i++) t += Rn[(size_t)k * N + i] * f[i];
      e[grow0[s] + k] = t;
    }
  const std::vector<double> G0 = Gd; // the un-orthonormalised G serves the rigid-body recovery
  if (o->orthonormalize == 1) { // explicit form: the filled T G0 (2: implicit, the library keeps G0 sparse, see pmh_qppf_create)
    L.assign((size_t)m * m, 0.0);
    for (int i = 0; i < m; i++)
      for (int j = 0; j <= i; j++) {
        double t = 0.0;
        for (int q = 0; q < nl; q++) t += Gd[(size_t)i * nl + q] * Gd[(size_t)j * nl + q];
        L[(size_t)i * m + j] = t;
      }
    if (chol_lower(m, L)) return pmh_set_error(PMH_ERR_ARG, "pmh_feti_contact_solve: G G' is not positive definite (dependent rows of G = R'B')");
    for (int i = 0; i < m; i++) { // row i of L^{-1} G by forward substitution over the rows
      for (int p = 0; p < i; p++) {
        const double lip = L[(size_t)i * m + p];
        if (lip != 0.0)
          for (int q = 0; q < nl; q++) Gd[(size_t)i * nl + q] -= lip * Gd[(size_t)p * nl + q];
        e[i] -= lip * e[p];
      }
      const double d = 1.0 / L[(size_t)i * m + i];
      for (int q = 0; q < nl; q++) Gd[(size_t)i * nl + q] *= d;
      e[i] *= d;
    }
  }
  std::vector<int>    grp((size_t)m + 1, 0), gci;
  std::vector<double> gva;
  for (int r = 0; r < m; r++) {
    for (int q = 0; q < nl; q++)
      if (std::fabs(Gd[(size_t)r * nl + q]) >= 1e-300) gci.push_back(q), gva.push_back(Gd[(size_t)r * nl + q]);
    grp[r + 1] = (int)gci.size();
  }

  pmh_csr        Kc = nullptr, Gc = nullptr;
  pmh_blockdiag  Kb = nullptr;
  pmh_matinv     Kp = nullptr;
  pmh_mg         mg = nullptr;
  pmh_gluing     B  = nullptr;
  pmh_qppf       pf = nullptr;
  pmh_fexplicit  E  = nullptr;
  pmh_feti_chain ch = nullptr;
  pmh_smalxe     sx = nullptr;
  double        *d_f = nullptr, *d_c = nullptr, *d_e = nullptr, *d_x = nullptr, *d_lb = nullptr, *d_lam = nullptr, *d_u0 = nullptr, *d_r = nullptr;
  int            rc = PMH_SUCCESS;
#define GO(call) \
  do { \
    if ((rc = (call))) goto done; \
  } while (0)
  {
    stage("kernel bases, dense G, e (host)");
    GO(pmh_csr_create(ctx, N, N, rowptr, col, val, &Kc));
    pmh_csr_set_host_hint(Kc, rowptr, col, val); // the set-up builders (3x3-block copies) read the caller's arrays instead of downloading them again
    GO(pmh_blockdiag_create(ctx, nsub, block_rowstart, Kc, &Kb));
    GO(pmh_matinv_create(Kb, o->kplus_rtol, 1e-50, o->kplus_max_it, 1, &Kp));
    GO(pmh_matinv_set_nullspace(Kp, kdim, Rn.data())); // P_R K^- P_R
    stage("K upload, block structure, inner KSP");
    if (o->bsr3 && ndof == 3 && pmh_matinv_enable_bsr3(Kp)) (void)0; // no 3x3 block structure: the CSR kernel stays
    if (o->mg && dims) {
      GO(pmh_mg_create_box(ctx, Kc, nsub, block_rowstart, dims, ndof, rowptr, col, val, kdim, Rn.data(), std::max(1, o->mg_min_nodes), std::max(1, o->mg_degree), o->mg_precision, &mg));
      GO(pmh_matinv_set_pc_mg(Kp, mg));
      stage("multigrid hierarchy (pmh_mg_create_box)");
    } else if (o->mg && ndof >= 1 && N % ndof == 0) { // blocks of any shape: the algebraic hierarchy (smoothed aggregation on the kernel vectors, mgsa.hip)
      const int prec = (ndof == 3 && kdim % 3 == 0) ? o->mg_precision : PMH_MG_FP64; // (the single-precision cycles need 3 x 3 blocks on every level)
      GO(pmh_mg_create_sa(ctx, Kc, nsub, block_rowstart, ndof, rowptr, col, val, kdim, Rn.data(), 0, nullptr, 3 * std::max(1, o->mg_min_nodes), 0.08, std::max(1, o->mg_degree), prec, &mg));
      GO(pmh_matinv_set_pc_mg(Kp, mg));
      stage("multigrid hierarchy (pmh_mg_create_sa)");
    }
    GO(pmh_gluing_create(ctx, N, nl, n_leaves, leaves_row, leaves_root, leaves_val, &B));
    GO(pmh_csr_create(ctx, m, nl, grp.data(), gci.data(), gva.data(), &Gc));
    GO(pmh_qppf_create(ctx, Gc, o->orthonormalize, &pf));
    stage("gluing, G, projector");
    std::vector<double> e_raw = e; // pairs with Gd = G0 in the diagnostic below
    if (o->orthonormalize == 2) GO(pmh_qppf_orth_rhs(pf, e_raw.data(), e.data())); // the constraint becomes (T G0) lambda = T e0
    if (o->explicit_dual) { // MatInvExplicitly restricted to the dofs B touches; congruent blocks share their columns
      std::vector<int> cls(nsub);
      GO(pmh_csr_block_classes(nsub, block_rowstart, rowptr, col, val, cls.data(), nullptr));
      stage("  explicit operators: block classes");
      // blocks of one class share K, hence K^+ (the Moore-Penrose inverse does not depend on the basis chosen for the kernel)
      int ncls = 0;
      for (int b = 0; b < nsub; b++) ncls = std::max(ncls, cls[b] + 1);
      // box-shaped blocks: the symmetries of the box that leave the class matrix invariant -> one K^+ solve per orbit of rows (PMH_FX_CLASS_SYM), or only the
      // representatives' rows kept and F's dense part applied as a GEMM (PMH_FX_CLASS_ORBIT; falls back to the symmetric tiles when a class has < 16 operations)
      auto set_symmetries = [&](int *least) -> int {
        *least = 1 << 30;
        for (int c = 0; c < ncls; c++) {
          int b0 = 0;
          while (b0 < nsub && cls[b0] != c) b0++;
          const int r0 = block_rowstart[b0], r1 = block_rowstart[b0 + 1];
          int       used = 1;
          if (r0 == 0) { // K is block diagonal: the first block's rows ARE its CSR (no copy of 20 M entries)
            PMH_CHK(pmh_fexplicit_set_box_symmetry(E, c, dims + 3 * b0, ndof, rowptr, col, val, &used));
          } else {
            const int           k0 = rowptr[r0], nz = rowptr[r1] - k0;
            std::vector<int>    rp((size_t)(r1 - r0) + 1), cj((size_t)nz);
            for (int i = r0; i <= r1; i++) rp[i - r0] = rowptr[i] - k0;
            for (int k = 0; k < nz; k++) cj[k] = col[k0 + k] - r0;
            PMH_CHK(pmh_fexplicit_set_box_symmetry(E, c, dims + 3 * b0, ndof, rp.data(), cj.data(), val + k0, &used));
          }
          st->explicit_symmetries = std::max(st->explicit_symmetries, used);
          *least = std::min(*least, used);
        }
        return PMH_SUCCESS;
      };
      int storage = o->explicit_storage;
      if (storage == PMH_FX_CLASS_ORBIT && !(dims && o->explicit_symmetry)) storage = PMH_FX_CLASS_SYM;
      if (storage == PMH_FX_CLASS_ORBIT) {
        int least = 0;
        GO(pmh_fexplicit_create_shared_orbit(B, Kb, cls.data(), &E));
        GO(set_symmetries(&least));
        if (least < 16) {
          // a class whose own touched set (a block's interface faces + a Dirichlet or contact face) is not invariant under the box's group keeps few operations -- the case of
          // non-congruent decompositions, one class per block.  On the CLOSURE of the touched set under the group (the whole boundary of a cube) every operation survives
          pmh_fexplicit_destroy(E);
          E = nullptr, st->explicit_symmetries = 0;
          std::vector<int> eptr((size_t)ncls + 1, 0), erel;
          std::vector<std::vector<int>> touched((size_t)ncls);
          for (int i = 0; i < n_leaves; i++) {
            const int b = block_of(leaves_row[i]);
            touched[cls[b]].push_back(leaves_row[i] - block_rowstart[b]);
          }
          for (int c = 0; c < ncls; c++) {
            std::sort(touched[c].begin(), touched[c].end());
            touched[c].erase(std::unique(touched[c].begin(), touched[c].end()), touched[c].end());
            int b0 = 0;
            while (b0 < nsub && cls[b0] != c) b0++;
            const int r0 = block_rowstart[b0], r1 = block_rowstart[b0 + 1], k0 = rowptr[r0], nz = rowptr[r1] - k0;
            std::vector<int> rp((size_t)(r1 - r0) + 1), cj((size_t)nz), out((size_t)(r1 - r0));
            for (int i = r0; i <= r1; i++) rp[i - r0] = rowptr[i] - k0;
            for (int k = 0; k < nz; k++) cj[k] = col[k0 + k] - r0;
            int n_out = 0;
            GO(pmh_box_symmetry_closure(dims + 3 * b0, ndof, rp.data(), cj.data(), val + k0, (int)touched[c].size(), touched[c].data(), &n_out, out.data(), nullptr));
            erel.insert(erel.end(), out.begin(), out.begin() + n_out);
            eptr[c + 1] = (int)erel.size();
          }
          if (erel.empty()) erel.push_back(0);
          GO(pmh_fexplicit_create_shared_orbit_union(B, Kb, cls.data(), eptr.data(), erel.data(), &E));
          GO(set_symmetries(&least));
          if (least < 16) { // still too few operations for the GEMM form (boxes with three different sides): the symmetric tiles
            pmh_fexplicit_destroy(E);
            E = nullptr, storage = PMH_FX_CLASS_SYM, st->explicit_symmetries = 0;
          }
        }
      }
      if (!E) {
        if (storage == PMH_FX_CLASS) GO(pmh_fexplicit_create_shared(B, Kb, cls.data(), &E)); // congruent blocks share one matrix per class
        else if (storage == PMH_FX_CLASS_SYM) GO(pmh_fexplicit_create_shared_sym(B, Kb, cls.data(), &E));
        else GO(pmh_fexplicit_create(B, Kb, storage, &E));
        if (storage == PMH_FX_CLASS_SYM && dims && o->explicit_symmetry) {
          int least = 0;
          GO(set_symmetries(&least));
        }
      }
      stage("  explicit operators: class union, gluing of the classes, symmetries");
      GO(pmh_fexplicit_assemble_auto(E, Kp, cls.data(), cls.data(), o->explicit_rtol, 0, nullptr)); // 8 columns per block and application where the multi-right-hand-side K^+ applies
      GO(pmh_matinv_attach_explicit(Kp, E));
      stage("  explicit operators: assembly (K^+ solves, self-check)");
      long long ns;
      GO(pmh_fexplicit_assemble_stats(E, &ns, &st->explicit_seconds));
      st->explicit_solves = (int)ns;
    }
    const size_t bl = sizeof(double) * (size_t)nl, bx = sizeof(double) * (size_t)N;
    GO(pmh_malloc(ctx, bx, (void **)&d_f));
    GO(pmh_malloc(ctx, bl, (void **)&d_c));
    GO(pmh_malloc(ctx, sizeof(double) * (size_t)m, (void **)&d_e));
    GO(pmh_malloc(ctx, bl, (void **)&d_x));
    GO(pmh_malloc(ctx, bl, (void **)&d_lb));
    GO(pmh_malloc(ctx, bl, (void **)&d_lam));
    GO(pmh_malloc(ctx, bx, (void **)&d_u0));
    GO(pmh_malloc(ctx, bl, (void **)&d_r));
    GO(pmh_memcpy_h2d(ctx, d_f, f, bx));
    GO(pmh_memcpy_h2d(ctx, d_c, c, bl));
    GO(pmh_memcpy_h2d(ctx, d_e, e.data(), sizeof(double) * (size_t)m));
    GO(pmh_memset(ctx, d_x, 0, bl)); // zero initial guess of the child (qptransform.c:1164-1165)
    {
      std::vector<double> lb((size_t)nl, -INFINITY); // lb(E) = -inf, lb(I) = 0 (qptransform.c:1136-1162)
      for (int i = n_eq; i < nl; i++) lb[i] = 0.0;
      GO(pmh_memcpy_h2d(ctx, d_lb, lb.data(), bl));
    }
    GO(pmh_qpt_feti_chain_create(B, Kp, d_f, d_c, pf, d_e, d_lb, &ch));
    pmh_op  A;
    double *b, *lbn;
    GO(pmh_qpt_feti_chain_get(ch, nullptr, &A, nullptr, nullptr, &b, &lbn, nullptr));
    GO(pmh_smalxe_create(ctx, A, b, d_x, lbn, nullptr, pf, &o->smalxe, &sx));
    GO(pmh_sync(ctx));
    stage("dual chain (d, lambda~, bounds), SMALXE set-up");
    pmh_csr_set_host_hint(Kc, nullptr, nullptr, nullptr);
    auto t_solve     = std::chrono::steady_clock::now();
    st->setup_seconds = std::chrono::duration<double>(t_solve - t_start).count();
    GO(pmh_smalxe_solve(sx));
    GO(pmh_smalxe_get_stats(sx, &st->smalxe));
    GO(pmh_sync(ctx));
    st->solve_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_solve).count();
    // ---- post-solve: lambda, u0 = K^+(f - B' lambda), r = F lambda - d; then the rigid-body amplitudes on the host
    GO(pmh_qpt_feti_chain_post_solve(ch, d_x, d_lam, d_u0, d_r));
    std::vector<double> lam((size_t)nl), r((size_t)nl);
    GO(pmh_memcpy_d2h(ctx, lam.data(), d_lam, bl));
    GO(pmh_memcpy_d2h(ctx, r.data(), d_r, bl));
    GO(pmh_memcpy_d2h(ctx, u_host, d_u0, bx));
    if (lambda_host) memcpy(lambda_host, lam.data(), bl);
    // G lambda = e (the equality constraint of the dual QP), measured with the G the solver was handed (implicit form: G0 lambda = e0)
    {
      double t2 = 0.0;
      for (int i = 0; i < m; i++) {
        double t = -(o->orthonormalize == 2 ? e_raw[i] : e[i]);
        for (int q = 0; q < nl; q++) t += Gd[(size_t)i * nl + q] * lam[q];
        t2 += t * t;
      }
      st->norm_Glambda_minus_e = std::sqrt(t2);
    }
    // B u0 - c = d - F lambda = -r and B R = G0', so u = u0 + R a gives B u - c = -r + G0' a, which must vanish on the equality rows
    // and on the ACTIVE contact rows (lambda_i > 0): least squares on those rows, (G0_T G0_T') a = G0_T r_T
    double lmax = 0.0;
    for (int q = 0; q < nl; q++) lmax = std::max(lmax, std::fabs(lam[q]));
    std::vector<char> tight((size_t)nl, 0);
    int               nact = 0;
    for (int q = 0; q < nl; q++) {
      tight[q] = (q < n_eq) || (lam[q] > 1e-8 * lmax);
      if (q >= n_eq && tight[q]) nact++;
    }
    st->n_active = nact;
    std::vector<double> M((size_t)m * m, 0.0), rhs((size_t)m, 0.0);
    for (int i = 0; i < m; i++) {
      for (int j = 0; j <= i; j++) {
        double t = 0.0;
        for (int q = 0; q < nl; q++)
          if (tight[q]) t += G0[(size_t)i * nl + q] * G0[(size_t)j * nl + q];
        M[(size_t)i * m + j] = t;
      }
      double t = 0.0;
      for (int q = 0; q < nl; q++)
        if (tight[q]) t += G0[(size_t)i * nl + q] * r[q];
      rhs[i] = t;
    }
    if (chol_lower(m, M)) {
      rc = pmh_set_error(PMH_ERR_STATE, "pmh_feti_contact_solve: the tight rows do not fix the rigid-body modes (G_T G_T' singular)");
      goto done;
    }
    for (int i = 0; i < m; i++) { // L y = rhs
      for (int p = 0; p < i; p++) rhs[i] -= M[(size_t)i * m + p] * rhs[p];
      rhs[i] /= M[(size_t)i * m + i];
    }
    for (int i = m - 1; i >= 0; i--) { // L' a = y
      for (int p = i + 1; p < m; p++) rhs[i] -= M[(size_t)p * m + i] * rhs[p];
      rhs[i] /= M[(size_t)i * m + i];
    }
    for (int s = 0; s < nsub; s++)
      for (int k = 0; k < bdim[s]; k++) {
        const double a = rhs[grow0[s] + k];
        for (int i = block_rowstart[s]; i < block_rowstart[s + 1]; i++) u_host[i] += Rn[(size_t)k * N + i] * a;
      }
  }
done:
#undef GO
  pmh_free(ctx, d_f), pmh_free(ctx, d_c), pmh_free(ctx, d_e), pmh_free(ctx, d_x), pmh_free(ctx, d_lb), pmh_free(ctx, d_lam), pmh_free(ctx, d_u0), pmh_free(ctx, d_r);
  pmh_smalxe_destroy(sx);
  pmh_qpt_feti_chain_destroy(ch);
  if (Kp) pmh_matinv_attach_explicit(Kp, nullptr);
  pmh_fexplicit_destroy(E);
  pmh_qppf_destroy(pf);
  pmh_gluing_destroy(B);
  pmh_matinv_destroy(Kp);
  pmh_mg_destroy(mg);
  pmh_blockdiag_destroy(Kb);
  pmh_csr_destroy(Gc), pmh_csr_destroy(Kc);
  return rc;
}
