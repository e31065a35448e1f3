// Structures of the multigrid V-cycle (mg.hip) shared with its multi-right-hand-side form (mg_mv.hip); internal
#pragma once
#include <vector>

#include "pmh_internal.h"

struct mg_level {
  pmh_csr  A, P;          // P: n_l x n_{l+1} (NULL on the coarsest level); A is used directly only if Ab == NULL
  pmh_bsr3 Ab;            // 3x3-block copy of A in the cycle's precision, or NULL
  int      n;
  void    *dinv, *x, *b, *r, *d, *t, *xa; // vectors in the cycle's precision; in fp64 x and b of level 0 are the caller's
  void    *pv, *rv;       // values of P and P' in the cycle's precision (fp32 copies: 8 instead of 12 bytes per entry; the
                          // trilinear weights 1, 1/2, 1/4, 1/8 are exact in any precision), or the CSR's own fp64 arrays
  bool     pv_owned;
  bool     long_rows = false; // P has > 12 entries per row on average (an aggregation hierarchy): the wavefront-per-row transfer kernels
  // node-level copies of P and P' when P = P_node (x) I_3 (cycle precision values), else NULL
  int     *pn_rowptr, *pn_col, *rn_rowptr, *rn_col;
  void    *pn_val, *rn_val;
  double   theta, delta;
  std::vector<double> c1, c2; // Chebyshev recurrence coefficients of steps 1..degree-1
};

struct pmh_mg_s {
  pmh_ctx               ctx;
  int                   nlevels, degree, is_float;
  std::vector<mg_level> L;
  int                   nb_coarse;
  int                  *d_crs;   // coarse block row starts [nb_coarse+1]
  long long            *d_cofs;  // offsets of the dense blocks [nb_coarse]
  void                 *d_cpinv; // concatenated dense pseudo-inverses, row-major, cycle precision (fp16 entries / cp_scale with PMH_MG_FP16)
  int                   cp_half;
  int                   coarse_m16; // every dense block has a multiple of 16 rows, at least 512 (the 8-column cycle's matrix-core coarse solve, mg_mv.hip)
  double                cp_scale;
  const int            *halt;
  long long             fine_spmv; // fine-level SpMVs issued (statistics)
  int                       timing_on;
  int                       fused; // degree 2 + block operators: smoothing steps finished inside the operator kernel
  std::vector<pmh_csr>      owned; // CSR handles created for this hierarchy by pmh_mg_create_box (destroyed with it)
};

