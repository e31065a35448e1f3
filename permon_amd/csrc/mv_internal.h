// Multi-right-hand-side K^+ (mv.hip, mg_mv.hip, matinv_mv.hip; internal): R = PMH_MV_R columns per block solved TOGETHER on interleaved multivectors.
//
// The set-up of the explicit local dual operators needs (K^+)[Gamma_b, Gamma_b] column by column -- what the reference gets from its factorisation with a block
// of right-hand sides (MatInvExplicitly_Inv, src/mat/impls/inv/matinv.c:640-730: MatMatSolve on column blocks).  The one-column-per-block solver streams K_b
// once per column; here a multivector V[(dof i) * R + r] keeps the R columns of a dof side by side, so that
//   * a 3 x 3 block of K_b is loaded ONCE and applied to R columns (8 x fewer matrix bytes per column; fp64: 154 MB per product of a 43^3 block),
//   * the gathered operand of a block column is ONE contiguous piece of 3 R values (192 bytes in fp64) instead of R separate 24-byte gathers,
//   * every launch of the V-cycle and of the CG serves R columns (the coarse levels are launch-latency bound).
// Blocks are concatenated as in MATBLOCKDIAG (global dof index), every block with its own matrix: no congruence, no symmetry, no box structure is assumed by
// the kernels (the hierarchy is whatever pmh_mg holds: its P must be node-wise, P = P_node (x) I_3).
#pragma once
#include "pmh_internal.h"

#define PMH_MV_R 8

// ELL copy of a matrix of 3 x 3 blocks, W slots per block row (a multiple of 4; padding: the row's own block column with zero values), slot planes interleaved
// by 4 for the four lanes of a block row: col[((s / 4) * nbr + br) * 4 + s % 4]; the entries (fp64 / fp32 / fp16 = entries / scale) per plane s / 4 as 16-byte vectors of the
// entries 0 ... 7 and one element for entry 8 (e = 3 q + c): struct mv_lay in mv.hip
struct pmh_mv_ell_s {
  pmh_ctx ctx;
  int     nbr, W, storage; // PMH_BSR_F64 / F32 / F16
  int     lpr = 4;         // lanes per block row of the product: 4, or 16 where the rows are long (W > 48: the coarse operators of an aggregation hierarchy); W is a multiple of it
  int    *col;
  void   *val;
  double  scale;
};
typedef pmh_mv_ell_s *pmh_mv_ell;

template <typename T> struct pmh_mv_epi {
  const T *y1, *dinv; // dinv: per ROW (not per column)
  T       *r, *d;
  double  *z64;
  T        c0, c1, c2;
};

// *out = NULL without error when A has no regular 3 x 3 block structure (unsorted rows, or more than 2048 blocks in a block row)
int pmh_mv_ell_create(pmh_csr A, int storage, pmh_mv_ell *out);
int pmh_mv_ell_create_prefix(pmh_csr A, int nrep, int storage, pmh_mv_ell *out); // the first of nrep congruent diagonal blocks of A
// the same copy of a RECTANGULAR matrix of 3 x 3 blocks (a prolongation of an aggregation hierarchy or its transpose: 3 | rows, 3 | columns); negate: the copy holds -A
int pmh_mv_ell_create_rect(pmh_csr A, int storage, int negate, pmh_mv_ell *out);
int pmh_mv_ell_destroy(pmh_mv_ell E);
#define PMH_MV_EPI_RESTRICT 20 // y = A x and, where e.d != NULL, e.d = e.dinv[row] * y * e.c0 (the coarse level's first smoothing direction rides on the restriction)
// y = A x on multivectors of R = PMH_MV_R columns (x, y: 3 nbr R entries) with the epilogues of k_bsr3 (PMH_EPI_NONE / ADD / SUB, PMH_BSR_EPI_PRE / POST1 /
// POST2)
int pmh_mv_spmv_f64(pmh_mv_ell E, const double *x, double *y, int epi, const pmh_mv_epi<double> *e, const int *halt);
int pmh_mv_spmv_f32(pmh_mv_ell E, const float *x, float *y, int epi, const pmh_mv_epi<float> *e, const int *halt);

// the V-cycle of pmh_mg on multivectors (mg_mv.hip).  create: PMH_EPI_UNSUPPORTED (no error recorded) for a hierarchy of another shape than the fused fp32
// cycle
struct pmh_mg_mv_s;
typedef pmh_mg_mv_s *pmh_mg_mv;
// nrep > 1: the hierarchy of the FIRST of nrep congruent blocks (every level is block diagonal with nrep equal blocks)
int pmh_mg_mv_create(pmh_mg mg, pmh_mg_mv *out, int nrep = 1);
int pmh_mg_mv_destroy(pmh_mg_mv M);
int pmh_mg_mv_apply(pmh_mg_mv M, const double *b, double *z, const int *halt); // z == NULL: no fp64 copy of the result, the caller reads pmh_mg_mv_result()
const float *pmh_mg_mv_result(pmh_mg_mv M); // the cycle's own (fp32) result multivector: valid until the next apply

// the block CG of MATINV for R columns per block (matinv_mv.hip) on top of a one-column solver's K, V-cycle, kernel basis and tolerances
struct pmh_matinv_mv_s;
typedef pmh_matinv_mv_s *pmh_matinv_mv;
int pmh_matinv_mv_create(pmh_matinv M, pmh_matinv_mv *out); // PMH_EPI_UNSUPPORTED (no error recorded) where it does not apply
// The solver's 8 CONGRUENT blocks as the 8 columns of ONE block: u = K^+ f of pmh_matinv_mult itself through the multi-right-hand-side kernels (f, u in the
// solver's own block-after-block layout).  PMH_EPI_UNSUPPORTED unless the solver has exactly PMH_MV_R blocks whose congruence pmh_bsr3_from_csr has verified on
// every level.
int pmh_matinv_mv_create_congruent(pmh_matinv M, pmh_matinv_mv *out);
int pmh_matinv_mv_mult_blocks(pmh_matinv_mv V, const double *f, double *u);
long long pmh_matinv_mv_products(pmh_matinv_mv V);
int pmh_matinv_mv_timing(pmh_matinv_mv V, int enable, int *launches, double *total_ms, double *bytes_per_launch);
int pmh_matinv_mv_destroy(pmh_matinv_mv V);
int pmh_matinv_mv_mult(pmh_matinv_mv V, const double *f, double *u);               // f, u: n R doubles, interleaved
int pmh_matinv_mv_to_columns(pmh_matinv_mv V, const double *u, double *cols);       // cols[r * n + i] = u[i * R + r]
int pmh_matinv_mv_columns(pmh_matinv_mv V);                                         // nblocks * R
int pmh_matinv_mv_last_iterations(pmh_matinv_mv V);

// why the last pmh_mg_mv_create / pmh_matinv_mv_create answered PMH_EPI_UNSUPPORTED (a static string, for the caller's error message)
const char *pmh_mv_why();
void        pmh_mv_set_why(const char *why);
