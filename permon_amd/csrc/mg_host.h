// Host-side pieces shared by the builders of multigrid hierarchies (mgbox.hip: geometric, boxes of Q1 nodes; mgsa.hip: algebraic, smoothed aggregation): CSR
// products on the host threads, the power-method estimate, dense coarse pseudo-inverses, and the hand-over of a finished hierarchy to pmh_mg_create.  Internal.
#pragma once
#include <functional>
#include <vector>

#include "pmh_internal.h"

namespace mgh {
struct HCsr {
  int                 nr = 0, nc = 0;
  std::vector<int>    rp, ci;
  std::vector<double> va;
};
struct Level {
  HCsr                A, P, Pt;
  double              lam = 0.0;
  std::vector<double> R; // kd x n_l kernel vectors of this level (injected from / reproduced on the fine level)
};
struct ClassH { // the private hierarchy of one class of congruent blocks
  std::vector<Level>  L;
  std::vector<double> pinv; // dense (pseudo-)inverse of the level the class is cut at
  int                 rep = 0, kd = 0;
};

HCsr   transpose(const HCsr &A);
HCsr   spgemm(const HCsr &A, const HCsr &B);  // rows on the host threads, columns sorted
HCsr   symmetrize(const HCsr &A);             // 0.5 (A + A'), sorted columns
double lambda_max_dinv_a(const HCsr &A, int its); // power method on D^-1 A from a fixed start vector (independent of the thread count)
void   parallel_for(int n, const std::function<void(int, int)> &f);
int    spd_inverse(int n, std::vector<double> &M);
int    coarse_pinv(const HCsr &A, int kd, const std::vector<double> &R, std::vector<double> &pinv);
int    finish(pmh_ctx ctx, pmh_csr A_fine, int nblocks, const std::vector<int> &cls, std::vector<ClassH> &H, int nlev, int degree, int precision, bool verbose, pmh_mg *out);
} // namespace mgh
