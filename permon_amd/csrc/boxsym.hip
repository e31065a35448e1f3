// Symmetries of a box-shaped block (host code): the signed coordinate permutations that map a box of nx x ny x nz nodes (x fastest, ndof
// dofs per node) onto itself, as signed permutations of its dofs -- a cube: the 48 elements of the hyperoctahedral group.  Where the block's
// matrix is invariant under them (identical elements of an isotropic material), so is its (pseudo-)inverse, and the set-up of the explicit
// local dual operators needs ONE K^+ solve per orbit of rows (pmh_fexplicit_set_class_symmetry, fshared.hip).
#include <algorithm>
#include <array>
#include <cmath>
#include <vector>

#include "pmh_internal.h"

namespace {
struct box_op {
  std::array<int, 3> ax, fl; // new coordinate a = fl[a] * old coordinate ax[a]
  bool operator==(const box_op &o) const { return ax == o.ax && fl == o.fl; }
};

bool compatible(const box_op &g, const int dims[3])
{
  for (int a = 0; a < 3; a++)
    if (dims[g.ax[a]] != dims[a]) return false;
  return true;
}

// dof i -> perm[i] with sign[i]; ndof == 3: the components transform as a vector, otherwise as scalars
void materialise(const box_op &g, const int dims[3], int ndof, int *perm, signed char *sign)
{
  const int nx = dims[0], ny = dims[1], nz = dims[2];
  for (int k = 0; k < nz; k++)
    for (int j = 0; j < ny; j++)
      for (int i = 0; i < nx; i++) {
        const int old[3] = {i, j, k};
        int       nw[3];
        for (int a = 0; a < 3; a++) nw[a] = g.fl[a] < 0 ? dims[a] - 1 - old[g.ax[a]] : old[g.ax[a]];
        const long long node = i + (long long)nx * (j + (long long)ny * k), node2 = nw[0] + (long long)nx * (nw[1] + (long long)ny * nw[2]);
        if (ndof == 3) {
          for (int a = 0; a < 3; a++) perm[node * 3 + g.ax[a]] = (int)(node2 * 3 + a), sign[node * 3 + g.ax[a]] = (signed char)g.fl[a];
        } else {
          for (int d = 0; d < ndof; d++) perm[node * ndof + d] = (int)(node2 * ndof + d), sign[node * ndof + d] = 1;
        }
      }
}

double csr_entry(const int *rowptr, const int *col, const double *val, int i, int j)
{
  const int *b = col + rowptr[i], *e = col + rowptr[i + 1];
  const int *p = std::lower_bound(b, e, j);
  if (p != e && *p == j) return val[p - col];
  for (const int *q = b; q != e; q++) // unsorted row
    if (*q == j) return val[q - col];
  return 0.0;
}
} // namespace

// dims: nodes per direction; rowptr / col / val: the block's matrix (n = nx ny nz ndof rows) or NULL: every GENERATOR (3 reflections, the swaps of
// equal axes) is checked on `nsample` rows spread over the block (K[g i][g j] s_i s_j = K[i][j] to 1e-11 of the largest sampled entry) and dropped if
// it fails; the operations returned are the closure of the surviving generators (<= 48), operation 0 the identity.  perm / sign: [48 * n] each.
extern "C" int pmh_box_symmetries(const int *dims, int ndof, const int *rowptr, const int *col, const double *val, int nsample, int *nsym, int *perm, signed char *sign)
{
  PMH_ARG(dims && dims[0] >= 1 && dims[1] >= 1 && dims[2] >= 1 && ndof >= 1 && nsym && perm && sign);
  const long long nn = (long long)dims[0] * dims[1] * dims[2] * ndof;
  PMH_ARG(nn < (1LL << 31));
  const int           n = (int)nn;
  std::vector<box_op> gens;
  for (int a = 0; a < 3; a++) {
    box_op g{{0, 1, 2}, {1, 1, 1}};
    g.fl[a] = -1;
    gens.push_back(g);
  }
  for (auto ax : {std::array<int, 3>{1, 0, 2}, std::array<int, 3>{0, 2, 1}, std::array<int, 3>{2, 1, 0}}) {
    box_op g{ax, {1, 1, 1}};
    if (compatible(g, dims)) gens.push_back(g);
  }
  if (rowptr && col && val && n > 0) {
    std::vector<int>         p((size_t)n);
    std::vector<signed char> s((size_t)n);
    const int                ns = std::max(1, std::min(nsample > 0 ? nsample : 4000, n));
    std::vector<box_op>      ok;
    double                   scale = 0.0;
    for (int t = 0; t < ns; t++) {
      const int i = (int)((long long)n * t / ns);
      for (int k = rowptr[i]; k < rowptr[i + 1]; k++) scale = std::max(scale, std::fabs(val[k]));
    }
    for (const box_op &g : gens) {
      materialise(g, dims, ndof, p.data(), s.data());
      bool good = true;
      for (int t = 0; t < ns && good; t++) {
        const int i = (int)((long long)n * t / ns);
        for (int k = rowptr[i]; k < rowptr[i + 1] && good; k++) {
          const int j = col[k];
          if (j < 0 || j >= n) continue; // entries outside the block
          const double w = csr_entry(rowptr, col, val, p[i], p[j]) * (double)s[i] * (double)s[j];
          if (!(std::fabs(w - val[k]) <= 1e-11 * scale)) good = false;
        }
      }
      if (good) ok.push_back(g);
    }
    gens.swap(ok);
  }
  std::vector<box_op> group{box_op{{0, 1, 2}, {1, 1, 1}}}, frontier = group;
  while (!frontier.empty()) {
    std::vector<box_op> next;
    for (const box_op &g1 : frontier)
      for (const box_op &g2 : gens) { // g1 first, then g2
        box_op c;
        for (int a = 0; a < 3; a++) c.ax[a] = g1.ax[g2.ax[a]], c.fl[a] = g2.fl[a] * g1.fl[g2.ax[a]];
        if (std::find(group.begin(), group.end(), c) == group.end()) group.push_back(c), next.push_back(c);
      }
    frontier.swap(next);
  }
  *nsym = (int)group.size();
  for (size_t g = 0; g < group.size(); g++) materialise(group[g], dims, ndof, perm + g * (size_t)n, sign + g * (size_t)n);
  return PMH_SUCCESS;
}

// The closure of a set of a block's dofs (in_rel: n_in block-relative indices) under the box's symmetries that leave the block's matrix invariant (pmh_box_symmetries):
// out_rel receives the sorted union of all images (capacity n = nx ny nz ndof), *n_out its size, *nsym the number of operations.  For a cube and a set that contains a whole face
// it is the whole boundary.  Host routine (set-up).
extern "C" int pmh_box_symmetry_closure(const int *dims, int ndof, const int *rowptr, const int *col, const double *val, int n_in, const int *in_rel, int *n_out, int *out_rel, int *nsym_out)
{
  PMH_ARG(dims && ndof >= 1 && n_in >= 0 && (n_in == 0 || in_rel) && n_out);
  const long long nn = (long long)dims[0] * dims[1] * dims[2] * ndof;
  PMH_ARG(nn >= 1 && nn < (1LL << 31));
  const int                n = (int)nn;
  std::vector<int>         perm((size_t)48 * n);
  std::vector<signed char> sign((size_t)48 * n);
  int                      nsym = 0;
  PMH_CHK(pmh_box_symmetries(dims, ndof, rowptr, col, val, 4000, &nsym, perm.data(), sign.data()));
  std::vector<char> in((size_t)n, 0);
  for (int i = 0; i < n_in; i++) {
    PMH_ARG(in_rel[i] >= 0 && in_rel[i] < n);
    for (int g = 0; g < nsym; g++) in[perm[(size_t)g * n + in_rel[i]]] = 1; // a group: the images under every element ARE the closure
  }
  int cnt = 0;
  for (int i = 0; i < n; i++)
    if (in[i]) {
      if (out_rel) out_rel[cnt] = i;
      cnt++;
    }
  *n_out = cnt;
  if (nsym_out) *nsym_out = nsym;
  return PMH_SUCCESS;
}
