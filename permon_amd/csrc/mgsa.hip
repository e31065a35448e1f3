// Algebraic multigrid hierarchy for block-diagonal matrices whose blocks have NO structure to lean on (subdomains that are not boxes: what a mesh partitioner
// hands the reference) -- the set-up of the PC of MATINV's inner KSP where pmh_mg_create_box does not apply.  The reference factorises ANY block
// (src/mat/impls/inv/matinv.c:481-580) and, on its iterative path, preconditions the inner KSP with whatever -mat_inv_pc_type names (PCGAMG is PETSc's
// smoothed aggregation); this is that set-up restated for the V-cycle of mg.hip.  Host C++ (it runs once per solve):
//   * smoothed aggregation (Vanek, Mandel, Brezina 1996): the node graph of the level (ndof dofs per node on the fine level, m dofs per aggregate below), strength
//     of a coupling = Frobenius norm of its block against theta * sqrt(|A_ii| |A_jj|), greedy aggregates in index order (roots with free neighbourhoods, leftovers
//     to the strongest neighbouring aggregate, the rest among themselves);
//   * the tentative prolongation reproduces the block's near-kernel B (m columns: for a floating block its kernel R_b, i.e. the rigid-body modes; else the caller's
//     vectors or the ndof translations) exactly: per aggregate B_a = Q_a R_a by Gram-Schmidt (twice), Q_a the aggregate's block of P_t, R_a the coarse near-kernel;
//     a column that is dependent over a small aggregate is dropped (a dead coarse dof: zero column, unit diagonal) so that every aggregate keeps m dofs and the
//     coarse operators keep their m x m (hence 3 x 3) block structure;
//   * one damped-Jacobi step P = (I - 4/3 / lambda_max(D^-1 A) D^-1 A) P_t, Galerkin operators A_{l+1} = P' A P (symmetrised) -- a floating block stays
//     consistently singular down the hierarchy (A B = 0 => P B_c = B and A_c B_c = 0), and its coarsest operator gets the pseudo-inverse of mgbox.hip;
//   * every block is coarsened the SAME number of times (that of the block that needs most to get under max_coarse dofs; a block that is down to <= 8 nodes meanwhile is
//     carried to the next level by P = I); congruent blocks are processed once.
// The scipy restatement the tests compare with: permon_amd/feti.py sa_mg_hierarchy.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <functional>
#include <thread>

#include "mg_host.h"
#include "pmh_internal.h"

namespace {
using namespace mgh;
#define SA_MIN_NODES 8 // a level of a class with at most this many nodes is not aggregated any further

// strength graph of the level over its nodes (bs dofs each): CSR without the diagonal, weights = Frobenius norms of the bs x bs blocks, kept where
// w_ij > theta sqrt(w_ii w_jj); columns ascending
struct NodeGraph {
  int                 nn = 0;
  std::vector<int>    rp, ci;
  std::vector<double> w;
};
NodeGraph strength_graph(const HCsr &A, int bs, double theta)
{
  NodeGraph G;
  const int nn = A.nr / bs;
  G.nn = nn;
  // pass 1: squared block norms per (node row, node column), rows on the host threads
  std::vector<std::vector<int>>    tci;
  std::vector<std::vector<double>> tw;
  std::vector<int>                 cnt((size_t)nn + 1, 0);
  std::vector<double>              dg((size_t)nn, 0.0);
  const int                        nt = std::max(1, std::min(pmh_host_threads(), nn / 2048 + 1));
  tci.resize(nt), tw.resize(nt);
  std::vector<int> lo(nt + 1);
  for (int t = 0; t <= nt; t++) lo[t] = (int)((long long)nn * t / nt);
  auto work = [&](int t) {
    std::vector<double> acc((size_t)nn, 0.0);
    std::vector<int>    mark((size_t)nn, -1), cols;
    for (int I = lo[t]; I < lo[t + 1]; I++) {
      cols.clear();
      for (int r = I * bs; r < (I + 1) * bs; r++)
        for (int k = A.rp[r]; k < A.rp[r + 1]; k++) {
          const int J = A.ci[k] / bs;
          if (mark[J] != I) mark[J] = I, acc[J] = 0.0, cols.push_back(J);
          acc[J] += A.va[k] * A.va[k];
        }
      std::sort(cols.begin(), cols.end());
      for (int J : cols) {
        const double v = std::sqrt(acc[J]);
        if (J == I) dg[I] = v;
        else if (v > 0.0) tci[t].push_back(J), tw[t].push_back(v), cnt[I + 1]++;
      }
    }
  };
  {
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++) th.emplace_back(work, t);
    for (auto &x : th) x.join();
  }
  // pass 2: the threshold (it needs both diagonal blocks)
  G.rp.assign((size_t)nn + 1, 0);
  size_t tot = 0;
  for (int t = 0; t < nt; t++) tot += tci[t].size();
  G.ci.reserve(tot), G.w.reserve(tot);
  for (int t = 0; t < nt; t++) {
    size_t p = 0;
    for (int I = lo[t]; I < lo[t + 1]; I++) {
      for (int q = 0; q < cnt[I + 1]; q++, p++) {
        const int    J = tci[t][p];
        const double v = tw[t][p];
        if (v > theta * std::sqrt(dg[I] * dg[J])) G.ci.push_back(J), G.w.push_back(v);
      }
      G.rp[I + 1] = (int)G.ci.size();
    }
  }
  return G;
}

// greedy aggregation (see the header comment); agg[i] = aggregate of node i
int aggregate(const NodeGraph &G, std::vector<int> &agg)
{
  const int nn = G.nn;
  agg.assign((size_t)nn, -1);
  int na = 0;
  for (int i = 0; i < nn; i++) { // phase 1
    if (agg[i] >= 0 || G.rp[i] == G.rp[i + 1]) continue;
    bool free_nb = true;
    for (int k = G.rp[i]; k < G.rp[i + 1] && free_nb; k++) free_nb = agg[G.ci[k]] < 0;
    if (!free_nb) continue;
    agg[i] = na;
    for (int k = G.rp[i]; k < G.rp[i + 1]; k++) agg[G.ci[k]] = na;
    na++;
  }
  const std::vector<int> agg1 = agg;
  for (int i = 0; i < nn; i++) { // phase 2
    if (agg[i] >= 0) continue;
    int    best = -1;
    double bw   = -1.0;
    for (int k = G.rp[i]; k < G.rp[i + 1]; k++)
      if (agg1[G.ci[k]] >= 0 && G.w[k] > bw * (1.0 + 1e-10)) best = G.ci[k], bw = G.w[k]; // (ties within rounding go to the lowest index whatever the summation order of the norms)
    if (best >= 0) agg[i] = agg1[best];
  }
  for (int i = 0; i < nn; i++) { // phase 3
    if (agg[i] >= 0) continue;
    agg[i] = na;
    for (int k = G.rp[i]; k < G.rp[i + 1]; k++)
      if (agg[G.ci[k]] < 0) agg[G.ci[k]] = na;
    na++;
  }
  return na;
}

// tentative prolongation P_t (n x na m, rows sorted) and the coarse near-kernel Bc (m x na m, row-major per vector) from the near-kernel B (m x n, row-major per vector)
void tentative(int n, int bs, int m, const std::vector<int> &agg, int na, const std::vector<double> &B, HCsr &Pt, std::vector<double> &Bc, std::vector<char> &dead)
{
  const int        nn = n / bs, nc = na * m;
  std::vector<int> start((size_t)na + 1, 0), order((size_t)nn);
  for (int i = 0; i < nn; i++) start[agg[i] + 1]++;
  for (int a = 0; a < na; a++) start[a + 1] += start[a];
  {
    std::vector<int> pos(start.begin(), start.end() - 1);
    for (int i = 0; i < nn; i++) order[pos[agg[i]]++] = i; // nodes of an aggregate in ascending order
  }
  Bc.assign((size_t)m * nc, 0.0);
  dead.assign((size_t)nc, 0);
  std::vector<double> Q((size_t)n * m, 0.0); // row i of P_t: the m entries of its aggregate's columns
  parallel_for(na, [&](int a0, int a1) {
    std::vector<double> V;
    for (int a = a0; a < a1; a++) {
      const int cntn = start[a + 1] - start[a], nd = cntn * bs;
      V.assign((size_t)m * nd, 0.0); // column-major: V[k * nd + r]
      for (int k = 0; k < m; k++)
        for (int q = 0; q < cntn; q++)
          for (int c = 0; c < bs; c++) V[(size_t)k * nd + q * bs + c] = B[(size_t)k * n + (size_t)order[start[a] + q] * bs + c];
      for (int k = 0; k < m; k++) {
        double *vk = &V[(size_t)k * nd];
        double  n0 = 0.0;
        for (int r = 0; r < nd; r++) n0 += vk[r] * vk[r];
        for (int pass = 0; pass < 2; pass++)
          for (int j = 0; j < k; j++) {
            const double *vj = &V[(size_t)j * nd];
            double        t  = 0.0;
            for (int r = 0; r < nd; r++) t += vj[r] * vk[r];
            for (int r = 0; r < nd; r++) vk[r] -= t * vj[r];
          }
        double nk = 0.0;
        for (int r = 0; r < nd; r++) nk += vk[r] * vk[r];
        n0 = std::sqrt(n0), nk = std::sqrt(nk);
        if (n0 == 0.0 || nk <= 1e-8 * n0) {
          for (int r = 0; r < nd; r++) vk[r] = 0.0;
          dead[(size_t)a * m + k] = 1;
        } else {
          for (int r = 0; r < nd; r++) vk[r] /= nk;
        }
      }
      // coarse near-kernel: Bc[:, a m + k] = Q_k' B_a  (B_a = Q (Q' B_a))
      for (int k = 0; k < m; k++)
        for (int j = 0; j < m; j++) {
          double t = 0.0;
          for (int q = 0; q < cntn; q++)
            for (int c = 0; c < bs; c++) t += V[(size_t)k * nd + q * bs + c] * B[(size_t)j * n + (size_t)order[start[a] + q] * bs + c];
          Bc[(size_t)j * nc + (size_t)a * m + k] = t;
        }
      for (int q = 0; q < cntn; q++)
        for (int c = 0; c < bs; c++)
          for (int k = 0; k < m; k++) Q[((size_t)order[start[a] + q] * bs + c) * m + k] = V[(size_t)k * nd + q * bs + c];
    }
  });
  Pt.nr = n, Pt.nc = nc;
  Pt.rp.assign((size_t)n + 1, 0);
  for (int i = 0; i < n; i++) {
    int c = 0;
    for (int k = 0; k < m; k++) c += Q[(size_t)i * m + k] != 0.0;
    Pt.rp[i + 1] = Pt.rp[i] + c;
  }
  Pt.ci.resize((size_t)Pt.rp[n]), Pt.va.resize((size_t)Pt.rp[n]);
  parallel_for(n, [&](int i0, int i1) {
    for (int i = i0; i < i1; i++) {
      int p = Pt.rp[i];
      for (int k = 0; k < m; k++)
        if (Q[(size_t)i * m + k] != 0.0) Pt.ci[p] = agg[i / bs] * m + k, Pt.va[p] = Q[(size_t)i * m + k], p++;
    }
  });
}

// P = P_t - s D^-1 (A P_t): the pattern of A P_t contains that of P_t wherever A has its diagonal; merged row by row (columns ascending)
HCsr smooth_prolongation(const HCsr &A, const HCsr &Pt, double s)
{
  const HCsr          AP = spgemm(A, Pt);
  const int           n  = A.nr;
  std::vector<double> dinv((size_t)n, 1.0);
  parallel_for(n, [&](int i0, int i1) {
    for (int i = i0; i < i1; i++)
      for (int k = A.rp[i]; k < A.rp[i + 1]; k++)
        if (A.ci[k] == i && A.va[k] != 0.0) dinv[i] = 1.0 / A.va[k];
  });
  HCsr P;
  P.nr = n, P.nc = Pt.nc;
  P.rp.assign((size_t)n + 1, 0);
  for (int i = 0; i < n; i++) { // size of the union per row
    int a = Pt.rp[i], b = AP.rp[i], c = 0;
    while (a < Pt.rp[i + 1] || b < AP.rp[i + 1]) {
      const int ca = a < Pt.rp[i + 1] ? Pt.ci[a] : INT32_MAX, cb = b < AP.rp[i + 1] ? AP.ci[b] : INT32_MAX;
      const int cc = std::min(ca, cb);
      a += ca == cc, b += cb == cc, c++;
    }
    P.rp[i + 1] = P.rp[i] + c;
  }
  P.ci.resize((size_t)P.rp[n]), P.va.resize((size_t)P.rp[n]);
  parallel_for(n, [&](int i0, int i1) {
    for (int i = i0; i < i1; i++) {
      int          a = Pt.rp[i], b = AP.rp[i], p = P.rp[i];
      const double f = s * dinv[i];
      while (a < Pt.rp[i + 1] || b < AP.rp[i + 1]) {
        const int ca = a < Pt.rp[i + 1] ? Pt.ci[a] : INT32_MAX, cb = b < AP.rp[i + 1] ? AP.ci[b] : INT32_MAX;
        const int cc = std::min(ca, cb);
        double    v  = 0.0;
        if (ca == cc) v += Pt.va[a++];
        if (cb == cc) v -= f * AP.va[b++];
        P.ci[p] = cc, P.va[p] = v, p++;
      }
    }
  });
  return P;
}
struct ClassX { // what the coarsening of a class carries besides its levels
  std::vector<double> B; // near-kernel of the current coarsest level, m x n_l
  int                 m = 0, bs = 0;
  bool                singular = false;
  bool                stalled = false; // its level cannot be aggregated any further (one aggregate): it does not ask for more levels
};

// the coarsening proper: every class the same number of times, until every class that still aggregates has at most max_coarse dofs
int sa_coarsen(std::vector<ClassH> &H, std::vector<ClassX> &X, int max_coarse, double theta, int *nlev_out, const std::function<void(const char *)> &stage)
{
  const int ncls = (int)H.size();
  int nlev = 1;
  for (;;) {
    int big = 0;
    for (int c = 0; c < ncls; c++)
      if (!X[c].stalled) big = std::max(big, H[c].L.back().A.nr);
    if (big <= max_coarse || nlev >= 10) break;
    for (int c = 0; c < ncls; c++) {
      ClassH   &C = H[c];
      ClassX   &Xc = X[c];
      Level    &F = C.L.back();
      const int n = F.A.nr, m = Xc.m;
      if (n % Xc.bs) return pmh_set_error(PMH_ERR_ARG, "pmh_mg_create_sa: level %d of block class %d has %d rows, not a multiple of its node size %d", nlev - 1, c, n, Xc.bs);
      auto carry_as_is = [&]() { // the class goes to the next level AS IT IS (P = I)
        Level Cn;
        HCsr  I;
        I.nr = I.nc = n;
        I.rp.resize((size_t)n + 1), I.ci.resize((size_t)n), I.va.assign((size_t)n, 1.0);
        for (int i = 0; i <= n; i++) I.rp[i] = i;
        for (int i = 0; i < n; i++) I.ci[i] = i;
        F.lam = lambda_max_dinv_a(F.A, 20);
        F.P = I, F.Pt = I;
        Cn.A = F.A;
        if (Xc.singular) Cn.R = Xc.B;
        C.L.push_back(std::move(Cn));
      };
      // a class that is already down to a handful of nodes while a larger one still coarsens, or whose (small, almost dense) level would collapse into ONE aggregate -- a
      // coarse operator that is zero on a floating block and a coarse correction that does nothing -- is carried over unchanged
      if (n / Xc.bs <= SA_MIN_NODES) {
        carry_as_is();
        continue;
      }
      const NodeGraph  G = strength_graph(F.A, Xc.bs, theta * std::pow(0.5, nlev - 1));
      std::vector<int> agg;
      const int        na = aggregate(G, agg);
      stage("  (class) strength graph, aggregates");
      if (na < 2) {
        Xc.stalled = true;
        carry_as_is();
        continue;
      }
      HCsr                Ptent;
      std::vector<double> Bc;
      std::vector<char>   dead;
      tentative(n, Xc.bs, m, agg, na, Xc.B, Ptent, Bc, dead);
      stage("  (class) tentative prolongation");
      F.lam = lambda_max_dinv_a(F.A, 20);
      stage("  (class) lambda_max, 20 products");
      F.P  = smooth_prolongation(F.A, Ptent, (4.0 / 3.0) / F.lam);
      F.Pt = transpose(F.P);
      stage("  (class) smoothed prolongation, transpose");
      Level Cn;
      {
        const HCsr AP = spgemm(F.A, F.P);
        const HCsr G2 = spgemm(F.Pt, AP);
        Cn.A          = symmetrize(G2);
      }
      for (int i = 0; i < Cn.A.nr; i++) // a dead coarse dof: its row and column are zero, a unit diagonal keeps the level operator definite there
        if (dead[i])
          for (int k = Cn.A.rp[i]; k < Cn.A.rp[i + 1]; k++)
            if (Cn.A.ci[k] == i) Cn.A.va[k] = 1.0;
      {
        // (a dead dof whose diagonal entry is not even stored cannot happen: A P_t has the column of every live dof only, and symmetrize keeps stored zeros; guard anyway)
        bool missing = false;
        for (int i = 0; i < Cn.A.nr && !missing; i++)
          if (dead[i]) {
            bool has = false;
            for (int k = Cn.A.rp[i]; k < Cn.A.rp[i + 1]; k++) has = has || Cn.A.ci[k] == i;
            missing = !has;
          }
        if (missing) { // rebuild with explicit diagonal entries for the dead dofs
          HCsr T;
          T.nr = T.nc = Cn.A.nr;
          T.rp.assign((size_t)T.nr + 1, 0);
          for (int i = 0; i < T.nr; i++) {
            bool has = false;
            for (int k = Cn.A.rp[i]; k < Cn.A.rp[i + 1]; k++) {
              if (dead[i] && !has && Cn.A.ci[k] > i) T.ci.push_back(i), T.va.push_back(1.0), has = true;
              has = has || Cn.A.ci[k] == i;
              T.ci.push_back(Cn.A.ci[k]), T.va.push_back(Cn.A.va[k]);
            }
            if (dead[i] && !has) T.ci.push_back(i), T.va.push_back(1.0);
            T.rp[i + 1] = (int)T.ci.size();
          }
          Cn.A = std::move(T);
        }
      }
      stage("  (class) Galerkin operator");
      if (Xc.singular) Cn.R = Bc;
      Xc.B  = std::move(Bc);
      Xc.bs = m;
      C.L.push_back(std::move(Cn));
    }
    nlev++;
  }
  *nlev_out = nlev;
  return PMH_SUCCESS;
}
} // namespace

// host routine (tests, diagnostics): the aggregates of ONE level of one block -- A (n x n host CSR, bs dofs per node), threshold theta; agg_out[n / bs], *n_agg
extern "C" int pmh_sa_aggregate(int n, int bs, const int *rowptr, const int *col, const double *val, double theta, int *agg_out, int *n_agg)
{
  PMH_ARG(n >= 1 && bs >= 1 && n % bs == 0 && rowptr && col && val && theta >= 0.0 && agg_out && n_agg);
  HCsr A;
  A.nr = A.nc = n;
  A.rp.assign(rowptr, rowptr + n + 1);
  A.ci.assign(col, col + rowptr[n]);
  A.va.assign(val, val + rowptr[n]);
  const NodeGraph  G = strength_graph(A, bs, theta);
  std::vector<int> agg;
  *n_agg = aggregate(G, agg);
  std::copy(agg.begin(), agg.end(), agg_out);
  return PMH_SUCCESS;
}

// host routine (tests, sanitizer runs, diagnostics; no device): the whole hierarchy of ONE block -- A: n x n host CSR with ndof dofs per node, R: kdim x n kernel vectors
// (kdim = 0: the ndof translations as near-kernel of a non-singular block).  level_rows[0 .. *nlevels): rows per level (cap 16); defect[0]: max over the levels of
// max |A_l B_l| / max |A_l| (a floating block stays consistently singular), defect[1]: max over the levels of max |P_l B_{l+1} - B_l| (the prolongations reproduce the
// near-kernel), defect[2]: max |A_c A_c^+ A_c - A_c| / max |A_c| of the coarsest operator and its dense pseudo-inverse
extern "C" int pmh_sa_hierarchy_host(int n, int ndof, const int *rowptr, const int *col, const double *val, int kdim, const double *R, int max_coarse, double theta, int *nlevels, int *level_rows,
                                     double *defect)
{
  PMH_ARG(n >= 1 && ndof >= 1 && n % ndof == 0 && rowptr && col && val && kdim >= 0 && kdim <= 8 && (kdim == 0 || R) && max_coarse >= 1 && theta >= 0.0 && nlevels && level_rows && defect);
  std::vector<ClassH> H(1);
  std::vector<ClassX> X(1);
  Level               L0;
  L0.A.nr = L0.A.nc = n;
  L0.A.rp.assign(rowptr, rowptr + n + 1);
  L0.A.ci.assign(col, col + rowptr[n]);
  L0.A.va.assign(val, val + rowptr[n]);
  X[0].bs = ndof;
  if (kdim) {
    X[0].B.assign(R, R + (size_t)kdim * n), X[0].m = kdim, X[0].singular = true, H[0].kd = kdim;
    L0.R = X[0].B;
  } else {
    X[0].m = ndof;
    X[0].B.assign((size_t)ndof * n, 0.0);
    for (int i = 0; i < n; i++) X[0].B[(size_t)(i % ndof) * n + i] = 1.0;
  }
  H[0].L.push_back(std::move(L0));
  int nlev = 1;
  PMH_CHK(sa_coarsen(H, X, max_coarse, theta, &nlev, [](const char *) {}));
  if (nlev > 16) return pmh_set_error(PMH_ERR_STATE, "pmh_sa_hierarchy_host: more than 16 levels");
  *nlevels = nlev;
  const int m = X[0].m;
  double    d_ab = 0.0, d_pb = 0.0;
  // (for a singular block the level's kernel vectors L[l].R ARE its near-kernel B_l; for a non-singular block only the sizes are reported)
  for (int l = 0; l < nlev; l++) {
    const Level &Lv = H[0].L[l];
    level_rows[l]   = Lv.A.nr;
    if (!X[0].singular) continue;
    double amax = 0.0;
    for (double v : Lv.A.va) amax = std::max(amax, std::fabs(v));
    for (int k = 0; k < m; k++)
      for (int i = 0; i < Lv.A.nr; i++) {
        double t = 0.0;
        for (int q = Lv.A.rp[i]; q < Lv.A.rp[i + 1]; q++) t += Lv.A.va[q] * Lv.R[(size_t)k * Lv.A.nr + Lv.A.ci[q]];
        d_ab = std::max(d_ab, std::fabs(t) / std::max(amax, 1e-300));
      }
    if (l + 1 < nlev) {
      const Level &Lc = H[0].L[l + 1];
      for (int k = 0; k < m; k++)
        for (int i = 0; i < Lv.P.nr; i++) {
          double t = 0.0;
          for (int q = Lv.P.rp[i]; q < Lv.P.rp[i + 1]; q++) t += Lv.P.va[q] * Lc.R[(size_t)k * Lc.A.nr + Lv.P.ci[q]];
          d_pb = std::max(d_pb, std::fabs(t - Lv.R[(size_t)k * Lv.A.nr + i]));
        }
    }
  }
  std::vector<double> pinv;
  if (coarse_pinv(H[0].L[nlev - 1].A, H[0].kd, H[0].L[nlev - 1].R, pinv)) return pmh_set_error(PMH_ERR_ARG, "pmh_sa_hierarchy_host: the coarsest operator is not positive definite on the complement of the kernel");
  const HCsr &Ac = H[0].L[nlev - 1].A;
  const int   nc = Ac.nr;
  std::vector<double> Ad((size_t)nc * nc, 0.0), T((size_t)nc * nc, 0.0);
  double              amax = 0.0;
  for (int i = 0; i < nc; i++)
    for (int q = Ac.rp[i]; q < Ac.rp[i + 1]; q++) Ad[(size_t)i * nc + Ac.ci[q]] = Ac.va[q], amax = std::max(amax, std::fabs(Ac.va[q]));
  for (int i = 0; i < nc; i++) // T = A pinv
    for (int k = 0; k < nc; k++) {
      const double a = Ad[(size_t)i * nc + k];
      if (a != 0.0)
        for (int j = 0; j < nc; j++) T[(size_t)i * nc + j] += a * pinv[(size_t)k * nc + j];
    }
  double d_pi = 0.0;
  for (int i = 0; i < nc; i++)
    for (int j = 0; j < nc; j++) {
      double t = 0.0;
      for (int k = 0; k < nc; k++) t += T[(size_t)i * nc + k] * Ad[(size_t)k * nc + j];
      d_pi = std::max(d_pi, std::fabs(t - Ad[(size_t)i * nc + j]) / std::max(amax, 1e-300));
    }
  defect[0] = d_ab, defect[1] = d_pb, defect[2] = d_pi;
  return PMH_SUCCESS;
}

// nns_host: nns x N near-kernel vectors for the NON-singular blocks (NULL / zero over a block: the ndof translations); a block over which R_host (kdim x N) is
// non-zero is singular with exactly that kernel, which is then its near-kernel as well.  max_coarse: coarsening stops when every block has at most that many dofs.
extern "C" int pmh_mg_create_sa(pmh_ctx ctx, pmh_csr A_fine, int nblocks, const int *block_rowstart, int ndof, const int *rowptr, const int *col, const double *val, int kdim, const double *R_host,
                                int nns, const double *nns_host, int max_coarse, double theta, int degree, int precision, pmh_mg *out)
{
  PMH_ARG(ctx && A_fine && out && nblocks >= 1 && block_rowstart && ndof >= 1 && rowptr && col && val && kdim >= 0 && kdim <= 8 && (kdim == 0 || R_host) && nns >= 0 && nns <= 8 && (nns == 0 || nns_host));
  PMH_ARG(max_coarse >= 1 && theta >= 0.0 && theta < 1.0 && degree >= 1);
  const int N = block_rowstart[nblocks];
  PMH_ARG(A_fine->nrows == N && block_rowstart[0] == 0);
  for (int b = 0; b < nblocks; b++)
    if ((block_rowstart[b + 1] - block_rowstart[b]) % ndof || block_rowstart[b + 1] <= block_rowstart[b])
      return pmh_set_error(PMH_ERR_ARG, "pmh_mg_create_sa: block %d has %d rows, not a positive multiple of ndof = %d", b, block_rowstart[b + 1] - block_rowstart[b], ndof);
  const bool verbose = getenv("PMH_CONTACT_TIMING") != nullptr;
  auto       t_last  = std::chrono::steady_clock::now();
  auto       stage   = [&](const char *what) {
    if (!verbose) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "  pmh_mg_create_sa: %-40s %7.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
    t_last = now;
  };
  std::vector<int> cls(nblocks);
  int              ncls = 0;
  PMH_CHK(pmh_csr_block_classes(nblocks, block_rowstart, rowptr, col, val, cls.data(), &ncls));
  stage("block classes");
  std::vector<ClassH> H(ncls);
  std::vector<ClassX> X(ncls);
  std::vector<char>   seen(ncls, 0);
  for (int b = 0; b < nblocks; b++) {
    const int c = cls[b];
    if (seen[c]) continue;
    seen[c]     = 1;
    ClassH &C   = H[c];
    C.rep       = b;
    const int r0 = block_rowstart[b], n = block_rowstart[b + 1] - r0, k0 = rowptr[r0];
    Level     L0;
    L0.A.nr = L0.A.nc = n;
    L0.A.rp.resize((size_t)n + 1);
    for (int i = 0; i <= n; i++) L0.A.rp[i] = rowptr[r0 + i] - k0;
    const size_t nz = (size_t)L0.A.rp[n];
    L0.A.ci.resize(nz), L0.A.va.resize(nz);
    parallel_for(n, [&](int i0, int i1) {
      for (int i = i0; i < i1; i++)
        for (int k = L0.A.rp[i]; k < L0.A.rp[i + 1]; k++) L0.A.ci[k] = col[k0 + k] - r0, L0.A.va[k] = val[k0 + k];
    });
    for (size_t k = 0; k < nz; k++)
      if (L0.A.ci[k] < 0 || L0.A.ci[k] >= n) return pmh_set_error(PMH_ERR_ARG, "pmh_mg_create_sa: block %d couples to a column outside itself", b);
    // the block's near-kernel: its kernel vectors if it has any, else the caller's vectors, else the translations
    ClassX &Xc = X[c];
    auto    take = [&](int nv, const double *V) {
      for (int k = 0; k < nv; k++) {
        const double *r  = V + (size_t)k * N + r0;
        bool          nzv = false;
        for (int i = 0; i < n && !nzv; i++) nzv = r[i] != 0.0;
        if (nzv) Xc.B.insert(Xc.B.end(), r, r + n), Xc.m++;
      }
    };
    take(kdim, R_host);
    Xc.singular = Xc.m > 0;
    C.kd        = Xc.m;
    if (!Xc.m && nns) take(nns, nns_host);
    if (!Xc.m) {
      Xc.m = ndof;
      Xc.B.assign((size_t)ndof * n, 0.0);
      for (int i = 0; i < n; i++) Xc.B[(size_t)(i % ndof) * n + i] = 1.0;
    }
    Xc.bs = ndof;
    if (Xc.singular) L0.R = Xc.B;
    C.L.push_back(std::move(L0));
  }
  stage("class representatives, near-kernels");
  int nlev = 1;
  PMH_CHK(sa_coarsen(H, X, max_coarse, theta, &nlev, stage));
  stage("hierarchies (host)");
  for (int c = 0; c < ncls; c++)
    if (coarse_pinv(H[c].L[nlev - 1].A, H[c].kd, H[c].L[nlev - 1].R, H[c].pinv))
      return pmh_set_error(PMH_ERR_ARG, "pmh_mg_create_sa: the coarsest operator of block class %d (n = %d) is not positive definite on the complement of the given kernel", c, H[c].L[nlev - 1].A.nr);
  stage("dense coarse pseudo-inverses");
  PMH_CHK(finish(ctx, A_fine, nblocks, cls, H, nlev, degree, precision, verbose, out));
  return PMH_SUCCESS;
}
