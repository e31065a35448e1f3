// Geometric multigrid hierarchy for block-diagonal matrices whose blocks are Q1 discretisations on boxes of nodes -- the set-up
// of the PC of MATINV's inner KSP (the reference configures it through -mat_inv_pc_type / PCMG, src/mat/impls/inv/matinv.c
// MatInvGetKSP / MatInvSetUp; PETSc builds Galerkin hierarchies with PCMGSetGalerkin).  Host C++ (it runs once per solve):
//   * trilinear prolongation P_l = P_z (x) P_y (x) P_x (x) I_ndof between the node boxes (coarse nodes {0,2,4,..} U {n-1}; exact for
//     linear functions also when the last interval is short), so rigid-body modes are reproduced exactly and floating blocks stay
//     consistently singular down to the coarsest level;
//   * Galerkin operators A_{l+1} = P_l' A_l P_l by two row-wise sparse products (Gustavson), symmetrised;
//   * lambda_max(D^-1 A_l) by 20 power iterations (what KSPChebyshev's eigen-estimate provides);
//   * the dense (pseudo-)inverse of every coarsest block: A^+ = (A + Q Q')^{-1} - Q Q' with Q the orthonormalised kernel basis
//     injected from the fine level (P R_{l+1} = R_l), by a threaded blocked Cholesky; plain inverse for a non-singular block.
// Congruent blocks (bit-identical matrices, pmh_csr_block_classes) are processed once.  The result goes to pmh_mg_create.
#include <algorithm>
#include <deque>
#include <chrono>
#include <climits>
#include <cmath>
#include <functional>
#include <future>
#include <thread>

#include "pmh_internal.h"
#include "mg_host.h"

int pmh_mg_adopt_csr(pmh_mg mg, pmh_csr A); // mg.hip: the hierarchy owns the CSR handles this builder created

namespace mgh {
HCsr transpose(const HCsr &A)
{
  HCsr T;
  T.nr = A.nc, T.nc = A.nr;
  T.rp.assign((size_t)A.nc + 1, 0);
  for (int c : A.ci) T.rp[c + 1]++;
  for (int i = 0; i < A.nc; i++) T.rp[i + 1] += T.rp[i];
  T.ci.resize(A.ci.size()), T.va.resize(A.va.size());
  std::vector<int> pos(T.rp.begin(), T.rp.end() - 1);
  for (int i = 0; i < A.nr; i++)
    for (int k = A.rp[i]; k < A.rp[i + 1]; k++) {
      const int p = pos[A.ci[k]]++;
      T.ci[p] = i, T.va[p] = A.va[k];
    }
  return T;
}

// C = A B, rows computed independently (Gustavson, dense accumulator per thread), columns sorted
HCsr spgemm(const HCsr &A, const HCsr &B)
{
  HCsr C;
  C.nr = A.nr, C.nc = B.nc;
  C.rp.assign((size_t)A.nr + 1, 0);
  const int nt = std::max(1, pmh_host_threads());
  std::vector<std::vector<int>>    tci(nt);
  std::vector<std::vector<double>> tva(nt);
  std::vector<int>                 lo(nt + 1);
  for (int t = 0; t <= nt; t++) lo[t] = (int)((long long)A.nr * t / nt);
  auto work = [&](int t) {
    std::vector<double> acc((size_t)B.nc, 0.0);
    std::vector<int>    mark((size_t)B.nc, -1), cols;
    for (int i = lo[t]; i < lo[t + 1]; i++) {
      cols.clear();
      for (int k = A.rp[i]; k < A.rp[i + 1]; k++) {
        const int    j = A.ci[k];
        const double a = A.va[k];
        for (int q = B.rp[j]; q < B.rp[j + 1]; q++) {
          const int c = B.ci[q];
          if (mark[c] != i) mark[c] = i, acc[c] = 0.0, cols.push_back(c);
          acc[c] += a * B.va[q];
        }
      }
      std::sort(cols.begin(), cols.end());
      C.rp[i + 1] = (int)cols.size();
      for (int c : cols) tci[t].push_back(c), tva[t].push_back(acc[c]);
    }
  };
  std::vector<std::thread> th;
  for (int t = 0; t < nt; t++) th.emplace_back(work, t);
  for (auto &x : th) x.join();
  for (int i = 0; i < A.nr; i++) C.rp[i + 1] += C.rp[i];
  C.ci.reserve((size_t)C.rp[A.nr]), C.va.reserve((size_t)C.rp[A.nr]);
  for (int t = 0; t < nt; t++) C.ci.insert(C.ci.end(), tci[t].begin(), tci[t].end()), C.va.insert(C.va.end(), tva[t].begin(), tva[t].end());
  return C;
}

// 0.5 (A + A'), columns sorted (A square with sorted columns)
HCsr symmetrize(const HCsr &A)
{
  const HCsr T = transpose(A); // sorted columns by construction
  HCsr       S;
  S.nr = S.nc = A.nr;
  S.rp.assign((size_t)A.nr + 1, 0);
  for (int i = 0; i < A.nr; i++) {
    int a = A.rp[i], b = T.rp[i];
    while (a < A.rp[i + 1] || b < T.rp[i + 1]) {
      const int ca = a < A.rp[i + 1] ? A.ci[a] : INT32_MAX, cb = b < T.rp[i + 1] ? T.ci[b] : INT32_MAX;
      const int c = std::min(ca, cb);
      double    v = 0.0;
      if (ca == c) v += A.va[a++];
      if (cb == c) v += T.va[b++];
      S.ci.push_back(c), S.va.push_back(0.5 * v);
    }
    S.rp[i + 1] = (int)S.ci.size();
  }
  return S;
}

double lambda_max_dinv_a(const HCsr &A, int its)
{
  const int           n = A.nr;
  std::vector<double> dinv(n, 1.0), v(n), w(n);
  for (int i = 0; i < n; i++)
    for (int k = A.rp[i]; k < A.rp[i + 1]; k++)
      if (A.ci[k] == i && A.va[k] != 0.0) dinv[i] = 1.0 / A.va[k];
  unsigned long long s = 0x9E3779B97F4A7C15ULL; // fixed start vector (splitmix64 -> [-1, 1))
  for (int i = 0; i < n; i++) {
    s += 0x9E3779B97F4A7C15ULL;
    unsigned long long z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL, z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL, z ^= z >> 31;
    v[i] = (double)(z >> 11) / 4503599627370496.0 - 1.0;
  }
  // 20 products with the 20 M-entry fine matrix were 0.25 s on one thread: rows in chunks of 4096 on the host threads, the two norms summed per chunk and
  // then over the chunks in order -- the estimate does not depend on the number of threads
  const int           CH = 4096, nch = (n + CH - 1) / CH;
  std::vector<double> pw((size_t)nch), pv((size_t)nch);
  double              lam = 1.0;
  for (int it = 0; it < its; it++) {
    parallel_for(nch, [&](int c0, int c1) {
      for (int c = c0; c < c1; c++) {
        double nv = 0.0, nw = 0.0;
        for (int i = c * CH; i < std::min(n, (c + 1) * CH); i++) {
          double t = 0.0;
          for (int k = A.rp[i]; k < A.rp[i + 1]; k++) t += A.va[k] * v[A.ci[k]];
          w[i] = dinv[i] * t;
          nw += w[i] * w[i], nv += v[i] * v[i];
        }
        pw[c] = nw, pv[c] = nv;
      }
    });
    double nv = 0.0, nw = 0.0;
    for (int c = 0; c < nch; c++) nw += pw[c], nv += pv[c];
    nw = std::sqrt(nw), nv = std::sqrt(nv);
    lam = nw / std::max(nv, 1e-300);
    const double inv = std::max(nw, 1e-300);
    parallel_for(n, [&](int i0, int i1) {
      for (int i = i0; i < i1; i++) v[i] = w[i] / inv;
    });
  }
  return lam;
}

void parallel_for(int n, const std::function<void(int, int)> &f)
{
  const int nt = std::max(1, std::min({pmh_host_threads(), n}));
  if (nt == 1) return f(0, n);
  std::vector<std::thread> th;
  for (int t = 0; t < nt; t++) th.emplace_back(f, (int)((long long)n * t / nt), (int)((long long)n * (t + 1) / nt));
  for (auto &x : th) x.join();
}

// in place: M (n x n, row-major, SPD) <- M^{-1}, by a blocked right-looking Cholesky M = L L' and M^{-1} = L^{-T} L^{-1}
int spd_inverse(int n, std::vector<double> &M)
{
  const int NB = 64;
  auto      at = [&](int i, int j) -> double & { return M[(size_t)i * n + j]; };
  for (int k0 = 0; k0 < n; k0 += NB) {
    const int k1 = std::min(n, k0 + NB);
    for (int k = k0; k < k1; k++) { // panel factorisation (lower triangle)
      double d = at(k, k);
      for (int p = k0; p < k; p++) d -= at(k, p) * at(k, p);
      if (!(d > 0.0)) return 1;
      d        = std::sqrt(d);
      at(k, k) = d;
      for (int i = k + 1; i < k1; i++) {
        double s = at(i, k);
        for (int p = k0; p < k; p++) s -= at(i, p) * at(k, p);
        at(i, k) = s / d;
      }
    }
    if (k1 == n) break;
    parallel_for(n - k1, [&](int a, int b) { // rows below the panel: L21 = A21 L11^{-T}
      for (int i = k1 + a; i < k1 + b; i++)
        for (int k = k0; k < k1; k++) {
          double s = at(i, k);
          for (int p = k0; p < k; p++) s -= at(i, p) * at(k, p);
          at(i, k) = s / at(k, k);
        }
    });
    parallel_for(n - k1, [&](int a, int b) { // trailing update of the lower triangle: A22 -= L21 L21'
      for (int i = k1 + a; i < k1 + b; i++)
        for (int j = k1; j <= i; j++) {
          double        s  = 0.0;
          const double *ri = &M[(size_t)i * n + k0], *rj = &M[(size_t)j * n + k0];
          for (int p = 0; p < k1 - k0; p++) s += ri[p] * rj[p];
          at(i, j) -= s;
        }
    });
  }
  // X = L^{-1} (lower), column blocks in parallel; then M^{-1} = X' X
  std::vector<double> X((size_t)n * n, 0.0);
  parallel_for(n, [&](int a, int b) {
    for (int j = a; j < b; j++) { // column j of L^{-1}: forward substitution of e_j
      double *x = &X[(size_t)j * n]; // stored as row j of X' (= column j of X)
      x[j]      = 1.0 / at(j, j);
      for (int i = j + 1; i < n; i++) {
        double        s  = 0.0;
        const double *li = &M[(size_t)i * n];
        for (int p = j; p < i; p++) s += li[p] * x[p];
        x[i] = -s / li[i];
      }
    }
  });
  // (M^{-1})_{ij} = sum_{p >= max(i,j)} X_{pi} X_{pj}, X_{pi} = Xt[i][p]
  parallel_for(n, [&](int a, int b) {
    for (int i = a; i < b; i++)
      for (int j = 0; j <= i; j++) {
        double        s  = 0.0;
        const double *xi = &X[(size_t)i * n], *xj = &X[(size_t)j * n];
        for (int p = i; p < n; p++) s += xi[p] * xj[p];
        M[(size_t)i * n + j] = s;
      }
  });
  for (int i = 0; i < n; i++)
    for (int j = i + 1; j < n; j++) M[(size_t)i * n + j] = M[(size_t)j * n + i];
  return 0;
}

// dense (pseudo-)inverse of a coarsest operator: A^+ = (A + s Q Q')^{-1} - Q Q' / s with Q the orthonormalised kernel vectors R (kd x n; kd = 0: the plain inverse);
// 1 if A is not positive definite on the complement of span(Q)
int coarse_pinv(const HCsr &A, int kd, const std::vector<double> &R, std::vector<double> &pinv)
{
  const int   n = A.nr;
  std::vector<double> M((size_t)n * n, 0.0), Q((size_t)kd * n);
  for (int i = 0; i < n; i++)
    for (int k = A.rp[i]; k < A.rp[i + 1]; k++) M[(size_t)i * n + A.ci[k]] = A.va[k];
  int kq = 0;
  for (int k = 0; k < kd; k++) { // Gram-Schmidt (twice) of the injected kernel vectors
    std::vector<double> v(R.begin() + (size_t)k * n, R.begin() + (size_t)(k + 1) * n);
    double              n0 = 0.0;
    for (double x : v) n0 += x * x;
    for (int pass = 0; pass < 2; pass++)
      for (int j = 0; j < kq; j++) {
        double t = 0.0;
        for (int i = 0; i < n; i++) t += Q[(size_t)j * n + i] * v[i];
        for (int i = 0; i < n; i++) v[i] -= t * Q[(size_t)j * n + i];
      }
    double nn = 0.0;
    for (double x : v) nn += x * x;
    if (nn <= 1e-20 * n0) continue;
    nn = std::sqrt(nn);
    for (int i = 0; i < n; i++) Q[(size_t)kq * n + i] = v[i] / nn;
    kq++;
  }
  // scale Q Q' to the size of A so that A + s Q Q' is well conditioned; (A + s Q Q')^{-1} = A^+ + Q Q' / s
  double sc = 0.0;
  for (int i = 0; i < n; i++) sc = std::max(sc, std::fabs(M[(size_t)i * n + i]));
  if (sc == 0.0) sc = 1.0;
  for (int k = 0; k < kq; k++)
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) M[(size_t)i * n + j] += sc * Q[(size_t)k * n + i] * Q[(size_t)k * n + j];
  if (spd_inverse(n, M)) return 1;
  for (int k = 0; k < kq; k++)
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) M[(size_t)i * n + j] -= Q[(size_t)k * n + i] * Q[(size_t)k * n + j] / sc;
  pinv = std::move(M);
  return 0;
}

// the classes' levels -> block-diagonal concatenation per level -> device CSRs -> pmh_mg_create (the hierarchy owns the handles created here)
int finish(pmh_ctx ctx, pmh_csr A_fine, int nblocks, const std::vector<int> &cls, std::vector<ClassH> &H, int nlev, int degree, int precision, bool verbose, pmh_mg *out)
{
  const int ncls   = (int)H.size();
  auto      t_last = std::chrono::steady_clock::now();
  auto      stage  = [&](const char *what) {
    if (!verbose) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "  pmh_mg hierarchy: %-40s %7.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
    t_last = now;
  };
  // block-diagonal concatenation per level -> device CSRs -> pmh_mg_create
  std::vector<pmh_csr> Ah(nlev), Ph(std::max(1, nlev - 1)), created;
  std::vector<double>  lam(std::max(1, nlev - 1), 1.0);
  Ah[0] = A_fine;
  struct HostCat { // the concatenated host arrays stay alive until pmh_mg_create has built its block copies from them (host hint: no download of what was just uploaded)
    std::vector<int>    rp, ci;
    std::vector<double> va;
  };
  std::deque<HostCat> kept;
  auto cat = [&](int l, int which, pmh_csr *dst) -> int { // which: 0 the level's operator, 1 its prolongation, 2 the transpose of the prolongation
    // sized once, every block filled by its own thread (the entry-by-entry push_back of one thread was 0.2 s of the set-up for configs[2])
    size_t nr_tot = 0, nz_tot = 0;
    std::vector<size_t> r0(nblocks), k0(nblocks);
    std::vector<int>    c0(nblocks);
    int                 roff = 0, coff = 0;
    for (int b = 0; b < nblocks; b++) {
      const HCsr &M = which == 1 ? H[cls[b]].L[l].P : (which == 2 ? H[cls[b]].L[l].Pt : H[cls[b]].L[l].A);
      r0[b] = nr_tot, k0[b] = nz_tot, c0[b] = coff;
      nr_tot += (size_t)M.nr, nz_tot += (size_t)M.rp[M.nr], roff += M.nr, coff += M.nc;
    }
    if (nz_tot > (size_t)0x7fffff00) return pmh_set_error(PMH_ERR_SUP, "pmh_mg hierarchy: level %d has %zu non-zeros (int32 row pointers)", l, nz_tot);
    kept.emplace_back();
    std::vector<int>    &rp = kept.back().rp, &ci = kept.back().ci;
    std::vector<double> &va = kept.back().va;
    rp.resize(nr_tot + 1), ci.resize(nz_tot), va.resize(nz_tot);
    rp[0] = 0;
    {
      auto fillb = [&](int b) {
        const HCsr &M = which == 1 ? H[cls[b]].L[l].P : (which == 2 ? H[cls[b]].L[l].Pt : H[cls[b]].L[l].A);
        for (int i = 0; i < M.nr; i++) rp[r0[b] + i + 1] = (int)(k0[b] + (size_t)M.rp[i + 1]);
        const size_t nz = (size_t)M.rp[M.nr];
        for (size_t k = 0; k < nz; k++) ci[k0[b] + k] = M.ci[k] + c0[b];
        std::copy(M.va.begin(), M.va.begin() + nz, va.begin() + k0[b]);
      };
      std::vector<std::thread> th;
      for (int b = 0; b < nblocks; b++) th.emplace_back(fillb, b);
      for (auto &x : th) x.join();
    }
    PMH_CHK(pmh_csr_create(ctx, roff, coff, rp.data(), ci.data(), va.data(), dst));
    pmh_csr_set_host_hint(*dst, rp.data(), ci.data(), va.data());
    if (which != 2) created.push_back(*dst);
    return PMH_SUCCESS;
  };
  for (int l = 0; l < nlev; l++) {
    if (l > 0) PMH_CHK(cat(l, 0, &Ah[l]));
    if (l + 1 < nlev) {
      PMH_CHK(cat(l, 1, &Ph[l]));
      pmh_csr Pt = nullptr; // the classes' own transposes, concatenated: pmh_mg_create would otherwise download P and transpose the 16 M entries on one host thread
      PMH_CHK(cat(l, 2, &Pt));
      PMH_CHK(pmh_csr_adopt_transpose(Ph[l], Pt));
      for (int c = 0; c < ncls; c++) lam[l] = (c == 0) ? H[c].L[l].lam : std::max(lam[l], H[c].L[l].lam);
    }
  }
  stage("level matrices -> device CSR");
  std::vector<int>    crs(nblocks + 1, 0);
  std::vector<double> cp;
  for (int b = 0; b < nblocks; b++) {
    const ClassH &C = H[cls[b]];
    crs[b + 1]      = crs[b] + C.L[nlev - 1].A.nr;
    cp.insert(cp.end(), C.pinv.begin(), C.pinv.end());
  }
  PMH_CHK(pmh_mg_create(ctx, nlev, Ah.data(), Ph.data(), degree, lam.data(), 0.1, 1.1, nblocks, crs.data(), cp.data(), precision, out));
  stage("pmh_mg_create (block copies, transfer operators)");
  for (pmh_csr a : created) {
    pmh_csr_set_host_hint(a, nullptr, nullptr, nullptr); // (the host arrays go away with this function)
    if (a->transpose) pmh_csr_set_host_hint(a->transpose, nullptr, nullptr, nullptr);
    PMH_CHK(pmh_mg_adopt_csr(*out, a));
  }
  return PMH_SUCCESS;
}
} // namespace mgh

namespace {
using namespace mgh;
// linear interpolation onto n grid nodes from the coarse nodes {0,2,4,...} U {n-1}: per fine node <= 2 (coarse index, weight)
struct Interp1 {
  int                 nc;
  std::vector<int>    c0, c1;
  std::vector<double> w0, w1;
  std::vector<int>    cnode; // fine index of every coarse node
};
Interp1 interp1d(int n)
{
  Interp1 I;
  for (int i = 0; i < n; i += 2) I.cnode.push_back(i);
  if (I.cnode.back() != n - 1) I.cnode.push_back(n - 1);
  I.nc = (int)I.cnode.size();
  I.c0.assign(n, 0), I.c1.assign(n, -1), I.w0.assign(n, 1.0), I.w1.assign(n, 0.0);
  if (I.nc == n) {
    for (int i = 0; i < n; i++) I.c0[i] = i;
    return I;
  }
  for (int j = 0; j + 1 < I.nc; j++) {
    const int a = I.cnode[j], b = I.cnode[j + 1];
    for (int i = a; i < b; i++) {
      const double t = (double)(i - a) / (double)(b - a);
      I.c0[i] = j, I.w0[i] = 1.0 - t;
      if (t > 0.0) I.c1[i] = j + 1, I.w1[i] = t;
    }
  }
  I.c0[n - 1] = I.nc - 1, I.w0[n - 1] = 1.0, I.c1[n - 1] = -1;
  return I;
}

// P = (Pz (x) Py (x) Px) (x) I_ndof for an nx x ny x nz node box (x fastest, node-major dofs)
HCsr prolongation(const int d[3], int ndof, int dc[3], std::vector<int> &coarse_to_fine_node)
{
  Interp1 I[3] = {interp1d(d[0]), interp1d(d[1]), interp1d(d[2])};
  for (int a = 0; a < 3; a++) dc[a] = I[a].nc;
  HCsr P;
  P.nr = d[0] * d[1] * d[2] * ndof, P.nc = dc[0] * dc[1] * dc[2] * ndof;
  P.rp.assign((size_t)P.nr + 1, 0);
  for (int k = 0; k < d[2]; k++)
    for (int j = 0; j < d[1]; j++)
      for (int i = 0; i < d[0]; i++) {
        // node entries sorted by coarse index: z outer, y, x inner (the candidates come in ascending order along every axis)
        int    cz[2] = {I[2].c0[k], I[2].c1[k]}, cy[2] = {I[1].c0[j], I[1].c1[j]}, cx[2] = {I[0].c0[i], I[0].c1[i]};
        double wz[2] = {I[2].w0[k], I[2].w1[k]}, wy[2] = {I[1].w0[j], I[1].w1[j]}, wx[2] = {I[0].w0[i], I[0].w1[i]};
        std::vector<std::pair<int, double>> ent;
        for (int a = 0; a < 2; a++)
          for (int b = 0; b < 2; b++)
            for (int c = 0; c < 2; c++)
              if (cz[a] >= 0 && cy[b] >= 0 && cx[c] >= 0) ent.push_back({(cz[a] * dc[1] + cy[b]) * dc[0] + cx[c], wz[a] * wy[b] * wx[c]});
        std::sort(ent.begin(), ent.end());
        const int node = (k * d[1] + j) * d[0] + i;
        for (int q = 0; q < ndof; q++) {
          for (auto &e : ent) P.ci.push_back(e.first * ndof + q), P.va.push_back(e.second);
          P.rp[(size_t)node * ndof + q + 1] = (int)P.ci.size();
        }
      }
  coarse_to_fine_node.clear();
  for (int k = 0; k < dc[2]; k++)
    for (int j = 0; j < dc[1]; j++)
      for (int i = 0; i < dc[0]; i++) coarse_to_fine_node.push_back((I[2].cnode[k] * d[1] + I[1].cnode[j]) * d[0] + I[0].cnode[i]);
  return P;
}

} // namespace

// dims: nblocks x 3 node counts (x fastest); rowptr / col / val: the host copy of the block-diagonal fine matrix A_fine holds on the
// device; R_host: kdim x n kernel vectors (block-wise, zero over non-singular blocks) or NULL; coarsening stops at <= min_nodes nodes.
extern "C" int pmh_mg_create_box(pmh_ctx ctx, pmh_csr A_fine, int nblocks, const int *block_rowstart, const int *dims, int ndof, const int *rowptr, const int *col, const double *val, int kdim,
                                 const double *R_host, int min_nodes, int degree, int precision, pmh_mg *out)
{
  PMH_ARG(ctx && A_fine && out && nblocks >= 1 && block_rowstart && dims && ndof >= 1 && rowptr && col && val && kdim >= 0 && kdim <= 8 && (kdim == 0 || R_host) && min_nodes >= 1 && degree >= 1);
  const int N = block_rowstart[nblocks];
  PMH_ARG(A_fine->nrows == N && block_rowstart[0] == 0);
  for (int b = 0; b < nblocks; b++)
    if (block_rowstart[b + 1] - block_rowstart[b] != dims[3 * b] * dims[3 * b + 1] * dims[3 * b + 2] * ndof)
      return pmh_set_error(PMH_ERR_ARG, "pmh_mg_create_box: block %d has %d rows, its box %d x %d x %d x %d dof", b, block_rowstart[b + 1] - block_rowstart[b], dims[3 * b], dims[3 * b + 1], dims[3 * b + 2], ndof);
  const bool verbose = getenv("PMH_CONTACT_TIMING") != nullptr;
  auto       t_last  = std::chrono::steady_clock::now();
  auto       stage   = [&](const char *what) {
    if (!verbose) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "  pmh_mg_create_box: %-40s %7.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
    t_last = now;
  };
  std::vector<int> cls(nblocks);
  int              ncls = 0;
  PMH_CHK(pmh_csr_block_classes(nblocks, block_rowstart, rowptr, col, val, cls.data(), &ncls));
  stage("block classes");
  // blocks of one class must also share the box and the kernel dimension (the kernel SPACE follows from the matrix)
  std::vector<ClassH> H(ncls);
  std::vector<char>   seen(ncls, 0);
  int                 nlev = 1 << 30;
  for (int b = 0; b < nblocks; b++) {
    ClassH &C = H[cls[b]];
    if (seen[cls[b]]) {
      if (dims[3 * b] != dims[3 * C.rep] || dims[3 * b + 1] != dims[3 * C.rep + 1] || dims[3 * b + 2] != dims[3 * C.rep + 2]) return pmh_set_error(PMH_ERR_ARG, "pmh_mg_create_box: congruent blocks %d and %d on different boxes", C.rep, b);
      continue;
    }
    seen[cls[b]] = 1, C.rep = b;
    const int r0 = block_rowstart[b], n = block_rowstart[b + 1] - r0, k0 = rowptr[r0];
    Level     L0;
    L0.A.nr = L0.A.nc = n;
    L0.A.rp.resize((size_t)n + 1);
    for (int i = 0; i <= n; i++) L0.A.rp[i] = rowptr[r0 + i] - k0;
    {
      // the class representative's block as a matrix of its own (column indices relative to the block): host threads over its entries (19 M for configs[2]: 45 ms on one thread)
      const size_t nz = (size_t)L0.A.rp[n];
      L0.A.ci.resize(nz), L0.A.va.resize(nz);
      const int                nt = std::max(1, std::min(pmh_host_threads(), (int)(nz / 1000000) + 1));
      std::vector<std::thread> th;
      for (int t = 0; t < nt; t++)
        th.emplace_back([&, t]() {
          const size_t a0 = nz * t / nt, a1 = nz * (t + 1) / nt;
          std::copy(val + k0 + a0, val + k0 + a1, L0.A.va.begin() + a0);
          for (size_t k = a0; k < a1; k++) L0.A.ci[k] = col[k0 + k] - r0;
        });
      for (auto &x : th) x.join();
    }
    C.kd = 0;
    for (int k = 0; k < kdim; k++) { // the block's non-zero kernel vectors
      const double *r = R_host + (size_t)k * N + r0;
      bool          nz = false;
      for (int i = 0; i < n && !nz; i++) nz = r[i] != 0.0;
      if (nz) L0.R.insert(L0.R.end(), r, r + n), C.kd++;
    }
    C.L.push_back(std::move(L0));
    int d[3] = {dims[3 * b], dims[3 * b + 1], dims[3 * b + 2]};
    while ((int)C.L.size() < 12 && (long long)d[0] * d[1] * d[2] > min_nodes && std::max({d[0], d[1], d[2]}) > 2) {
      int              dc[3];
      std::vector<int> c2f;
      Level           &F = C.L.back();
      stage("  (class) level copy / injected kernel");
      F.P                = prolongation(d, ndof, dc, c2f);
      stage("  (class) prolongation");
      // lambda_max(D^-1 A) needs only F.A: it runs beside the Galerkin product below (its result does not depend on how many threads it gets)
      auto lam_job = std::async(std::launch::async, [&F]() { return lambda_max_dinv_a(F.A, 20); });
      Level Cn;
      {
        const HCsr AP = spgemm(F.A, F.P);
        stage("  (class) A P");
        F.Pt = transpose(F.P);
        stage("  (class) P'");
        const HCsr G2 = spgemm(F.Pt, AP);
        stage("  (class) P' (A P)");
        Cn.A = symmetrize(G2);
        stage("  (class) symmetrise");
      }
      F.lam = lam_job.get();
      stage("  (class) lambda_max, 20 products (beside the Galerkin product)");
      const int nc = Cn.A.nr, nf = F.A.nr;
      Cn.R.resize((size_t)C.kd * nc);
      for (int k = 0; k < C.kd; k++)
        for (int i = 0; i < nc; i++) Cn.R[(size_t)k * nc + i] = F.R[(size_t)k * nf + (size_t)c2f[i / ndof] * ndof + i % ndof];
      C.L.push_back(std::move(Cn));
      d[0] = dc[0], d[1] = dc[1], d[2] = dc[2];
    }
    nlev = std::min(nlev, (int)C.L.size());
  }
  stage("Galerkin operators, lambda_max (host)");
  // dense (pseudo-)inverse of the level every class is cut at
  for (int c = 0; c < ncls; c++)
    if (coarse_pinv(H[c].L[nlev - 1].A, H[c].kd, H[c].L[nlev - 1].R, H[c].pinv)) return pmh_set_error(PMH_ERR_ARG, "pmh_mg_create_box: the coarsest operator of block class %d (n = %d) is not positive definite on the complement of the given kernel", c, H[c].L[nlev - 1].A.nr);
  stage("dense coarse pseudo-inverses");
  PMH_CHK(finish(ctx, A_fine, nblocks, cls, H, nlev, degree, precision, verbose, out));
  return PMH_SUCCESS;
}
