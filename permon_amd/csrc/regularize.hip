// MatRegularize (src/mat/interface/permonmatregularize.c): the set-up step of the reference's default FETI path
// (-regularize 1, qptransform.c:2215,2231 -> MAT_REG_EXPLICIT) that turns the singular stiffness block K_i of a floating
// subdomain into the SPD K_reg,i = K_i + rho^2 Q_i whose inverse is a generalised inverse of K_i; MATINV then works on
// K_reg (matinv.c:449-459) and its KSP needs no null-space handling.
// Host code (it runs once per block on a p x d kernel basis, d <= 6 for 3-D elasticity; O(p d^2) work): the index
// bookkeeping -- which d "fixing" DOFs are picked -- is reproduced exactly, the d x d dense algebra to rounding.
#include <algorithm>
#include <cmath>

#include "pmh_internal.h"

namespace {
constexpr double kEps = 2.220446049250313e-16; // PETSC_MACHINE_EPSILON (real double)

// Column-major p x d work copy with the access pattern of MatRegularize_GetPivots_Private (:6-116).
struct PivotSearch {
  int                 p, d;
  std::vector<double> w;
  std::vector<int>    perm;
  PivotSearch(int p_, int d_, const double *R) : p(p_), d(d_), w(R, R + (size_t)p_ * d_), perm((size_t)p_)
  {
    for (int i = 0; i < p; i++) perm[i] = i;
  }
  double &at(int i, int j) { return w[(size_t)j * p + i]; }

  void run()
  {
    int last_row = p - 1;
    for (int last_col = d - 1; last_col >= 0; last_col--, last_row--) {
      // [vpivot, ipivot, jpivot] = first largest |R(0:last_row, 0:last_col)|, columns outer, rows inner (:37-50)
      int    ip = 0, jp = 0;
      double vp = 0.0;
      for (int j = 0; j <= last_col; j++) {
        const double *c = &w[(size_t)j * p];
        for (int i = 0; i <= last_row; i++)
          if (std::fabs(c[i]) > std::fabs(vp)) ip = i, jp = j, vp = c[i];
      }
      for (int j = 0; j <= last_col; j++) std::swap(at(ip, j), at(last_row, j)); // rows, columns 0..J only (:56-64)
      std::swap(perm[ip], perm[last_row]);
      if (jp != last_col) std::swap_ranges(&at(0, jp), &at(0, jp) + last_row + 1, &at(0, last_col)); // columns, rows 0..II (:69-75)
      const double *piv = &w[(size_t)last_col * p];
      for (int j = 0; j < last_col; j++) { // make row II vanish in the remaining columns (:78-99)
        double *c = &w[(size_t)j * p];
        if (std::fabs(c[last_row]) < kEps) continue;
        const double alpha = -vp / c[last_row];
        for (int i = 0; i <= last_row; i++) {
          double v = c[i];
          v *= alpha;
          v += piv[i];
          c[i] = v;
        }
      }
    }
  }
};

// lower Cholesky factor of the SPD d x d matrix a (row-major, in place); false if a pivot is not positive
bool small_cholesky(int d, std::vector<double> &a)
{
  for (int j = 0; j < d; j++) {
    double s = a[j * d + j];
    for (int k = 0; k < j; k++) s -= a[j * d + k] * a[j * d + k];
    if (!(s > 0.0)) return false;
    const double l = std::sqrt(s);
    a[j * d + j]   = l;
    for (int i = j + 1; i < d; i++) {
      double t = a[i * d + j];
      for (int k = 0; k < j; k++) t -= a[i * d + k] * a[j * d + k];
      a[i * d + j] = t / l;
    }
  }
  return true;
}
} // namespace

// MatRegularize_GetPivots_Private (permonmatregularize.c:6-116): R_host = R_loc, p x d column-major; pivots_out: the d
// selected rows in ascending order (ISSort :106).
extern "C" int pmh_mat_regularize_pivots(int p, int d, const double *R_host, int *pivots_out)
{
  PMH_ARG(p >= 0 && d >= 0 && d <= p && (d == 0 || (R_host && pivots_out)));
  if (!d) return PMH_SUCCESS;
  PivotSearch ps(p, d, R_host);
  ps.run();
  std::copy(ps.perm.end() - d, ps.perm.end(), pivots_out);
  std::sort(pivots_out, pivots_out + d);
  return PMH_SUCCESS;
}

// MatRegularize_GetRegularization_Private (:118-196): Q_condensed = RI (RI'RI)^{-1} RI', RI = R(pivots,:), filtered with
// MatFilterZeros(10 eps) (kept iff |q| > 10 eps, permonmatutils.c:547).  Q_out: d x d row-major, dropped entries = 0,
// keep_out flags the stored ones.
extern "C" int pmh_mat_regularization_Q(int p, int d, const double *R_host, const int *pivots, double *Q_out, int *keep_out)
{
  PMH_ARG(p >= 0 && d >= 0 && (d == 0 || (R_host && pivots && Q_out && keep_out)));
  if (!d) return PMH_SUCCESS;
  std::vector<double> RI((size_t)d * d), M((size_t)d * d), Minv((size_t)d * d), T((size_t)d * d), y((size_t)d);
  for (int i = 0; i < d; i++) {
    PMH_ARG(pivots[i] >= 0 && pivots[i] < p);
    for (int j = 0; j < d; j++) RI[i * d + j] = R_host[(size_t)j * p + pivots[i]];
  }
  for (int i = 0; i < d; i++)
    for (int j = 0; j < d; j++) {
      double s = 0.0;
      for (int k = 0; k < d; k++) s += RI[k * d + i] * RI[k * d + j];
      M[i * d + j] = s;
    }
  if (!small_cholesky(d, M)) return pmh_set_error(PMH_ERR_STATE, "pmh_mat_regularization_Q: R(pivots,:) is rank deficient (RI'RI not positive definite)");
  for (int c = 0; c < d; c++) { // column c of (RI'RI)^{-1}: L y = e_c, L' z = y
    for (int i = 0; i < d; i++) {
      double s = (i == c) ? 1.0 : 0.0;
      for (int k = 0; k < i; k++) s -= M[i * d + k] * y[k];
      y[i] = s / M[i * d + i];
    }
    for (int i = d - 1; i >= 0; i--) {
      double s = y[i];
      for (int k = i + 1; k < d; k++) s -= M[k * d + i] * y[k];
      y[i] = s / M[i * d + i];
    }
    for (int i = 0; i < d; i++) Minv[i * d + c] = y[i];
  }
  for (int i = 0; i < d; i++)
    for (int j = 0; j < d; j++) {
      double s = 0.0;
      for (int k = 0; k < d; k++) s += RI[i * d + k] * Minv[k * d + j];
      T[i * d + j] = s;
    }
  for (int i = 0; i < d; i++)
    for (int j = 0; j < d; j++) {
      double s = 0.0;
      for (int k = 0; k < d; k++) s += T[i * d + k] * RI[j * d + k];
      keep_out[i * d + j] = std::fabs(s) > 10.0 * kEps;
      Q_out[i * d + j]    = keep_out[i * d + j] ? s : 0.0;
    }
  return PMH_SUCCESS;
}

// MatRegularize (:198-287) for one sequential block, MAT_REG_EXPLICIT: K_reg = K + rho (rho Q) on the union pattern
// (MatScale(Q_loc,rho) :256 then MatAXPY(Kreg_loc,rho,Q_loc,DIFFERENT_NONZERO_PATTERN) :265 -- rho enters twice, kept).
// rho: the caller's MatGetMaxEigenvalue(K_loc,NULL,&rho,1,20) (:254; pmh_op_max_eigenvalue(K, 1.0, 20, ...)).
// Host CSR in, host CSR out (columns sorted in every row, as PETSc keeps them); the output arrays must hold
// rowptr[n] + d*d entries; *nnz_out receives the stored count.  d = 0 (no kernel): K_reg = K.
extern "C" int pmh_mat_regularize_csr(int n, const int *rowptr, const int *col, const double *val, int d, const double *R_host, double rho, int *pivots_out, int *rowptr_out, int *col_out,
                                      double *val_out, long long *nnz_out)
{
  PMH_ARG(n >= 0 && rowptr && rowptr_out && nnz_out && d >= 0 && d <= n);
  PMH_ARG(rowptr[n] == 0 || (col && val && col_out && val_out));
  std::vector<int>    slot((size_t)n, -1), keep((size_t)d * d);
  std::vector<double> Q((size_t)d * d);
  if (d) {
    PMH_ARG(R_host && pivots_out);
    PMH_CHK(pmh_mat_regularize_pivots(n, d, R_host, pivots_out));
    PMH_CHK(pmh_mat_regularization_Q(n, d, R_host, pivots_out, Q.data(), keep.data()));
    for (int i = 0; i < d; i++) slot[pivots_out[i]] = i;
  }
  long long nz = 0;
  rowptr_out[0] = 0;
  for (int i = 0; i < n; i++) {
    const int k1 = rowptr[i + 1], s = slot[i];
    int       k = rowptr[i], j = 0;
    for (int c = k; c + 1 < k1; c++)
      if (col[c] >= col[c + 1]) return pmh_set_error(PMH_ERR_ARG, "pmh_mat_regularize_csr: row %d has unsorted or repeated columns", i);
    auto next_q = [&]() { // next stored entry of row s of Q, or d
      while (s >= 0 && j < d && !keep[s * d + j]) j++;
      return (s >= 0) ? j : d;
    };
    for (;;) {
      const int jq = next_q();
      const int cq = (jq < d) ? pivots_out[jq] : 0x7fffffff, ck = (k < k1) ? col[k] : 0x7fffffff;
      if (cq == 0x7fffffff && ck == 0x7fffffff) break;
      const double add = (jq < d) ? rho * (Q[s * d + jq] * rho) : 0.0;
      if (ck < cq) {
        col_out[nz] = ck, val_out[nz] = val[k], k++;
      } else if (ck == cq) {
        col_out[nz] = ck, val_out[nz] = val[k] + add, k++, j++;
      } else {
        col_out[nz] = cq, val_out[nz] = add, j++;
      }
      nz++;
    }
    if (nz > 0x7fffffffLL) return pmh_set_error(PMH_ERR_SUP, "pmh_mat_regularize_csr: more than 2^31-1 entries");
    rowptr_out[i + 1] = (int)nz;
  }
  *nnz_out = nz;
  return PMH_SUCCESS;
}

// ---- QPFetiGetBgtSF (src/qp/impls/feti/qpfeti.c:465-925): the signed gluing matrix B_g of a decomposition ----------------------
// Input: the local-to-global dof maps of the nsub subdomains ("ranks" of the reference: one sequential block per rank,
// matblockdiag.c:787-788), concatenated; a global dof that appears in m >= 2 subdomains has m copies, ordered by subdomain index
// (PetscSFSetRankOrder :507,598).  Links (rows of B_g, the dual unknowns) are numbered root by root in ascending global dof
// (:636-688 with the sorted i2g of qptransform.c:2118), within a root in the order the reference assigns them:
//   nonred (:643-648)  m-1 links, copy 0 against copy k;          full (:650-657)  the m(m-1)/2 pairs (i, k > i), i outer;
//   both: +1 on the lower, -1 on the highest rank of the link (:786-806), times 1/sqrt(m) with -SCALE_ON (the default);
//   orth (:659-667, :700-716, :807-817)  m-1 orthonormal links: link k couples copies 0..d-1 (value 1/d) with copy d = m-1-k
//   (value -1), all divided by sqrt(1/d + 1).
// Output: the leaves (local dof in the concatenated numbering, link, value) grouped by link, copies in rank order -- the order
// MatMultTranspose_Gluing's PetscSFReduce sums them in.  exclude: sorted global dofs left out (-feti_gluing_exclude_dirichlet,
// qpfeti.c:423-431).  leaves_* may be NULL: only the counts are returned (first of two calls).
extern "C" int pmh_feti_gluing_from_l2g(int nsub, const int *l2g_start, const int *l2g, int type, int scale, int n_exclude, const int *exclude, int *n_lambda, int *n_leaves, int *leaves_row,
                                        int *leaves_root, double *leaves_val)
{
  PMH_ARG(nsub >= 1 && l2g_start && n_lambda && n_leaves && (type >= 0 && type <= 2) && (n_exclude == 0 || exclude));
  PMH_ARG(l2g_start[0] == 0 && (l2g_start[nsub] == 0 || l2g));
  if (type < 0 || type > 2) return pmh_set_error(PMH_ERR_ARG, "Unknown FETI gluing type"); // qpfeti.c:561
  const int ntot = l2g_start[nsub];
  // (global dof, subdomain, local position) sorted by global dof, then subdomain: the copies of every root in rank order
  struct Copy {
    int g, s, loc;
  };
  std::vector<Copy> cp;
  cp.reserve((size_t)ntot);
  for (int s = 0; s < nsub; s++) {
    PMH_ARG(l2g_start[s + 1] >= l2g_start[s]);
    for (int i = l2g_start[s]; i < l2g_start[s + 1]; i++) {
      if (l2g[i] < 0) return pmh_set_error(PMH_ERR_ARG, "pmh_feti_gluing_from_l2g: negative global index at position %d", i);
      cp.push_back({l2g[i], s, i});
    }
  }
  std::stable_sort(cp.begin(), cp.end(), [](const Copy &a, const Copy &b) { return a.g != b.g ? a.g < b.g : a.s < b.s; });
  for (size_t i = 1; i < cp.size(); i++)
    if (cp[i].g == cp[i - 1].g && cp[i].s == cp[i - 1].s) return pmh_set_error(PMH_ERR_ARG, "pmh_feti_gluing_from_l2g: global dof %d appears twice in subdomain %d", cp[i].g, cp[i].s);
  for (int i = 1; i < n_exclude; i++) PMH_ARG(exclude[i] > exclude[i - 1]);
  long long nl = 0, nleaf = 0;
  const bool fill = leaves_row && leaves_root && leaves_val;
  auto emit = [&](int row, long long link, double v) {
    if (fill) leaves_row[nleaf] = row, leaves_root[nleaf] = (int)link, leaves_val[nleaf] = v;
    nleaf++;
  };
  for (size_t a = 0; a < cp.size();) {
    size_t b = a;
    while (b < cp.size() && cp[b].g == cp[a].g) b++;
    const int m = (int)(b - a);
    if (m >= 2 && !(n_exclude && std::binary_search(exclude, exclude + n_exclude, cp[a].g))) {
      const double sc = scale ? 1.0 / sqrt((double)m) : 1.0;
      if (type == 0) {
        for (int k = 1; k < m; k++, nl++) emit(cp[a].loc, nl, sc), emit(cp[a + k].loc, nl, -sc);
      } else if (type == 1) {
        for (int i = 0; i < m - 1; i++)
          for (int k = i + 1; k < m; k++, nl++) emit(cp[a + i].loc, nl, sc), emit(cp[a + k].loc, nl, -sc);
      } else {
        for (int k = 0; k < m - 1; k++, nl++) {
          const int    d = m - 1 - k;
          const double x = sqrt(1.0 / d + 1.0);
          for (int t = 0; t < d; t++) emit(cp[a + t].loc, nl, 1.0 / d / x);
          emit(cp[a + d].loc, nl, -1.0 / x);
        }
      }
    }
    a = b;
  }
  if (nl > 0x7fffffffLL || nleaf > 0x7fffffffLL) return pmh_set_error(PMH_ERR_SUP, "pmh_feti_gluing_from_l2g: more than 2^31-1 links or leaves");
  *n_lambda = (int)nl, *n_leaves = (int)nleaf;
  return PMH_SUCCESS;
}
