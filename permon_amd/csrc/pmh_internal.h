// Internal declarations of libpermonhip (gfx950 only).  Public ABI: include/permon_hip.h
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "permon_hip.h"

#define PMH_BLOCK 256          // threads per workgroup (4 wavefronts of 64)
#define PMH_MAX_VEC_BLOCKS 2048 // 256 CUs x 8 resident workgroups: grid cap of the streaming kernels
#define PMH_MAX_RED 8          // reductions finalised per pass
#define PMH_NSCAL 64           // scalar slots (device + pinned host mirror)

int pmh_set_error(int code, const char *fmt, ...);

// process-wide run-time switches (pmh_set_knob; initial values from the environment, read ONCE)
struct pmh_knobs_s {
  int chain; // the five-launch dual-space chain (dualchain.hip)
  // counters (pmh_get_knob; pmh_set_knob resets them): applications of the chain and its OWN launches (the middle stage's are not counted)
  int chain_applies = 0, chain_launches = 0;
  // A/B switches that sit on per-product / per-iteration paths: the environment is read ONCE (ctx.hip pmh_knobs), never inside a solver loop
  int gt_fusion = 1;       // PMH_NO_GT_FUSION: the projector's v - G'(...) epilogue folded into the G' kernel (qppf.hip)
  int smalxe_prefetch = 1; // PMH_SMALXE_NO_PREFETCH: ||B u|| enqueued before the inner solver's host wait (smalxe.hip)
  int vec_epi = 1;         // PMH_NO_VEC_EPI: MPGP's vector phase in the operator's last kernel (mpgp.hip, qppf.hip)
  int mpgp_spec = 1;       // PMH_MPGP_NO_SPEC: batches of device-side CG steps for CSR operators (mpgp.hip)
  int mg_d0_fusion = 1;    // PMH_MG_NO_D0_FUSION: the first smoothing step written by the producer of the right-hand side (feti.hip)
  int kplus_mv = 1;        // PMH_NO_KPLUS_MV: pmh_matinv_mult on 8 congruent blocks runs them as the 8 columns of one block on the multi-right-hand-side kernels (feti.hip)
  int multi_rhs = 1;       // PMH_NO_MULTI_RHS: the set-up of the explicit operators solves 8 columns per block at a time where matinv_mv.hip applies (pmh_fexplicit_assemble_auto)
  int svm_pairing = 1;  // the SVM dual's paired passes over X inside MPGP (svm.hip); 0 (PMH_SVM_NO_PAIRING): every Hessian application as its own two passes
  // threads of the host-side set-up builders (bsr.hip, mgbox.hip, fexplicit.hip, contact.hip): PMH_HOST_THREADS, else OMP_NUM_THREADS, else min(16, the
  int host_threads = 1;
                        // CPUs this process may run on) -- several ranks per node must share the node's cores (bench.py hands every rank its share)
};
pmh_knobs_s &pmh_knobs();
inline int   pmh_host_threads() { return pmh_knobs().host_threads > 1 ? pmh_knobs().host_threads : 1; }

#define PMH_HIP(call) \
  do { \
    hipError_t e_ = (call); \
    if (e_ != hipSuccess) return pmh_set_error(PMH_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
  } while (0)
#define PMH_CHK(call) \
  do { \
    int rc_ = (call); \
    if (rc_) return rc_; \
  } while (0)
#define PMH_NCCL(call) \
  do { \
    ncclResult_t r_ = (call); \
    if (r_ != ncclSuccess) return pmh_set_error(PMH_ERR_COMM, "%s:%d %s -> %s", __FILE__, __LINE__, #call, ncclGetErrorString(r_)); \
  } while (0)
#define PMH_ARG(cond) \
  do { \
    if (!(cond)) return pmh_set_error(PMH_ERR_ARG, "%s:%d argument check failed: %s", __FILE__, __LINE__, #cond); \
  } while (0)

struct pmh_ctx_s {
  int         device;
  int         num_cus;
  hipStream_t stream;
  hipEvent_t  ev0, ev1;
  double     *d_partials; // [PMH_MAX_RED][partials_cap] block partials of the streaming kernels
  int         partials_cap;
  double     *d_scal; // finalised reductions (device copy), consumed by follow-up kernels
  double     *h_scal; // pinned host mirror written by the finalise kernel
  double     *h_partials; // pinned host copy of d_partials for kernels that leave the last reduction step to the host (emit_inline.h pmh_block_partials)
  // multi-GPU
  ncclComm_t comm;
  int        rank, size, force_comm;
  double    *d_commbuf; // small staging buffer for scalar allreduces
  int        dist_scalars; // set while a solver with row-distributed vectors runs: finalised scalars are all-reduced
  // host-staged transport of the collectives (pmh_comm_set_host_transport: e.g. MPI_Allreduce on the glue's communicator) instead of RCCL
  pmh_comm_host_fn hook      = nullptr;
  void            *hook_user = nullptr;
  double          *h_stage   = nullptr; // pinned
  size_t           stage_cap = 0;
  // optional timing of the vector all-reduces (pmh_comm_timing_enable)
  hipEvent_t *comm_ev      = nullptr;
  int         comm_ev_cap = 0, comm_ev_used = 0;
  double      comm_ev_bytes = 0.0;
};
// the data-path collectives are live: a transport exists (RCCL communicator or host transport) and there is more than one rank (or PMH_COMM_FORCE=1)
static inline bool pmh_comm_on(pmh_ctx c) { return (c->comm || c->hook) && (c->size > 1 || c->force_comm); }
int pmh_comm_allreduce_scalars(pmh_ctx c, double *dscal, int K, const int *ops /* PMH_RED_SUM / PMH_RED_MIN per scalar */); // K device scalars, one grouped exchange

// ---- CSR -----------------------------------------------------------------------------------------------
enum { PMH_SPMV_STREAM = 0, PMH_SPMV_VECTOR = 1 };
struct pmh_csr_s {
  pmh_ctx   ctx;
  int       nrows, ncols;
  long long nnz;
  int      *d_rowptr, *d_col;
  double   *d_val;
  int       kind;          // PMH_SPMV_STREAM (row-blocked, LDS staged) or PMH_SPMV_VECTOR (sub-wave per row)
  int       lanes_per_row; // VECTOR kind
  int      *d_rowblocks;   // STREAM kind: row block boundaries [n_rowblocks+1]
  unsigned short *d_col16; // STREAM kind, short rows: 16-bit column offsets from d_cbase[row block] (nullptr: a row block spans >= 65 536 columns)
  int      *d_cbase;
  // STREAM kind, uniformly short rows (<= 8 non-zeros, little padding): slot-major copy per block of 256 rows (k_spmv_ell, no LDS staging)
  double         *d_ell_val;
  unsigned short *d_ell_c16; // offsets from d_ell_cbase[row block] (blocks span < 65 536 columns) ...
  int            *d_ell_col; // ... or absolute columns
  int            *d_ell_cbase;
  int             ell_w, ell_nrb;
  int       st_nnzb, st_mode, st_nt, st_rl; // STREAM kind: tile size, persistence mode, non-temporal streams
  int       n_rowblocks;
  double   *d_blockpart;   // [4][n_launch_blocks] partials of the fused MPGP epilogue
  int       n_launch_blocks;
  pmh_csr   transpose;     // built lazily for mult_transpose
  // borrowed host copy of the arrays (set-up builders only: pmh_csr_set_host_hint; the caller keeps them alive and unchanged until it clears the hint)
  const int    *h_rowptr, *h_col;
  const double *h_val;
  unsigned long long uid = 0;   // unique per created matrix (spmv.hip pmh_csr_create): caches keyed on a matrix compare this, not the address a later matrix may reuse
  int           congruent_nrep = 0; // > 1: pmh_bsr3_from_csr has compared the nrep diagonal blocks entry by entry and found them equal (the values never change after creation: a second conversion skips the comparison)
  // very long rows (G of the coarse problem: a few dozen rows of ~10^4 non-zeros): rows split into chunks, see spmv.hip
  int      *d_lchunks, *d_lrow; // [3*l_nchunks] (row, k0, k1) and [nrows+1] first chunk of each row
  double   *d_lpart;            // [l_nchunks] chunk sums
  double   *d_lpart2;           // second vector of pmh_csr_mult_partials2 (lazily allocated)
  int       l_nchunks;
  // optional per-launch timing (HIP event pairs recorded on the launch stream)
  std::vector<hipEvent_t> *ev;
  std::vector<int>        *ev_kind;
  int                      ev_used, ev_pending;
};

// epilogues of the SpMV kernels
enum { PMH_EPI_NONE = 0, PMH_EPI_ADD = 1, PMH_EPI_SUB = 2, PMH_EPI_MPGP = 3 };
struct pmh_spmv_epi {
  int           kind;
  const double *y1;             // ADD: y = y1 + A x ; SUB: y = A x - y1
  const double *g, *xx, *lb, *ub; // MPGP: partials p'Ap, g'p, min feasible step (p is the SpMV input)
  int           scal_base;      // MPGP: d_scal slots [base..base+2] receive pAp, gp, afeas
  const int    *halt;           // optional device flag: when set the launch (and its finalise) is a no-op
};
int pmh_csr_spmv_launch(pmh_csr A, const double *x, double *y, const pmh_spmv_epi &epi);
int pmh_csr_adopt_transpose(pmh_csr A, pmh_csr At); // A' built by the caller; A owns it afterwards
inline void pmh_csr_set_host_hint(pmh_csr A, const int *rowptr, const int *col, const double *val) { A->h_rowptr = rowptr, A->h_col = col, A->h_val = val; }
int pmh_csr_mult_partials(pmh_csr A, const double *x, const int **lrow, const double **part); // chunk sums of A x (long rows), summed by the consumer
// the same for two vectors in ONE pass over A (each sum as the single form takes it)
int pmh_csr_mult_partials2(pmh_csr A, const double *x, const double *x2, const int **lrow, const double **part, const double **part2);
// y = M (A x), Mt = M' (m x m, device); norm_slot >= 0 (m <= 64): ||y||^2 -> d_scal / h_scal[slot]
int pmh_csr_mult_then_dense(pmh_csr A, const double *x, const double *Mt, double *tmp, double *y, int norm_slot = -1);

// ---- operators -------------------------------------------------------------------------------------------
// Vector epilogues an operator may fold into the LAST kernel of its product y = A x (one entry per thread, the grid of the streaming Vec kernels:
// the same per-workgroup partial sums as the separate kernels, hence the same bits).  mult_epi returns PMH_EPI_UNSUPPORTED where an operator (or its
// current configuration) has no such kernel: the caller then runs the separate launches.
#define PMH_EPI_UNSUPPORTED (-77)
enum { PMH_VEPI_P1 = 1, PMH_VEPI_GRAD_SPLIT = 2 };
struct pmh_vec_epi {
  int           kind;
  // PMH_VEPI_P1 (x = p, y = Ap): p'Ap, g'p, QPCFeas(xx, p) -> partials rows prow .. prow + 2 (k_p1_dots) PMH_VEPI_GRAD_SPLIT (x = the iterate, y = g): g = A x
  // - b, then gf, p = gf and the partials of Ap'gf (0), |gP|^2, |gc|^2, |gf|^2 -> rows prow .. prow + 3 (k_axpy + k_split_setp)
  const double *g, *xx, *lb, *ub, *b;
  double        astol;
  double       *gf, *p;
  double       *partials;
  int           ld, prow;
  // pairing of passes for operators that stream a big matrix twice per application (the dense-row SVM Hessian, svm.hip); ignored by the others. P1: p_fresh = p
  // is still the gf the last PMH_VEPI_GRAD_SPLIT of this operator wrote (nothing touched it since); spec_alpha > 0: the driver will, if the step turns out to
  // be an expansion (std direction, fixed length alpha), ask for the gradient at exactly k_expansion_std(x, g, p, Ap, afeas, alpha): the operator may prepare
  // it. GRAD_SPLIT: x_from_spec = the iterate is that prepared expansion (the driver did NOT run k_expansion_std; the operator also writes it to x)
  int           p_fresh, x_from_spec;
  double        spec_alpha;
  double       *x_out;
  // fused dual-space chain (dualchain.hip): in_slot = 1 + the emission target that holds G0 x for this input (pmh_op_s::emit_begin: 1 = iterate, 2 = direction;
  // 0 = none: the operator forms it itself).  hosted != nullptr: the operator may ALSO store its block partials in the pinned host copy h_partials (same rows /
  // ld) and then sets *hosted = the number of blocks it wrote: the caller adds them up on the host after its next wait, no finalising launch.  emitted_p
  // (GRAD_SPLIT): the last kernel also emits G0 p for the p = gf it writes and sets *emitted_p = 1
  int           in_slot;
  double       *h_partials;
  int          *hosted, *emitted_p;
};
// ---- coarse-space emission of the fused dual-space chain (dualchain.hip, emit_inline.h) ----------------------------------
// G0 of the projector cut into (row, block of 256 columns) segments; the kernel that writes a dual vector sums its own segments (see emit_inline.h)
struct pmh_emit_tab {
  // [nwg][64][3] per tile of 1024 dual entries its segments (k0, k1, position of the segment's sum in the row-major array of partial sums), k1 = k0: none
  const int    *seg;
  const int    *gcol;    // G0 as CSR (device)
  const double *gval;
  const int    *lrow;    // [m + 1] a row's partial sums are part[lrow[r] .. lrow[r + 1])
  int           m, nwg;
};
struct pmh_emit_out { // one emitted vector v
  double *part; // [nseg] segment sums of G0 v (nullptr: target off)
};
struct pmh_emit_args {
  pmh_emit_tab tab;
  pmh_emit_out o[2]; // target 0: the iterate, target 1: the direction (mpgp.hip); dualchain.hip's own kernels use target 0
};

struct pmh_op_s {
  pmh_ctx ctx;
  int     n;
  virtual ~pmh_op_s() {}
  virtual int     mult(const double *x, double *y) = 0;
  virtual int     mult_epi(const double *, double *, const pmh_vec_epi &) { return PMH_EPI_UNSUPPORTED; }
  // Coarse emission (fused dual-space chain): the caller is about to launch a kernel that writes the iterate x (x != nullptr) and / or the direction p (p !=
  // nullptr), one entry per thread on the grid of the streaming Vec kernels; on PMH_SUCCESS *ea is filled and the kernel MUST end with pmh_emit_tail(*ea, ...)
  // -- the operator then takes G0 x / G0 p as given when it is applied with pmh_vec_epi::in_slot 0 / 1.  PMH_EPI_UNSUPPORTED: no such chain, launch the plain
  // kernel.
  virtual int     emit_begin(const double * /*x*/, const double * /*p*/, pmh_emit_args *) { return PMH_EPI_UNSUPPORTED; }
  virtual void    emit_invalidate() {} // x or p are about to change without emission
  virtual int     spec_expansion_ready() { return 0; } // the last PMH_VEPI_P1 prepared the expansion step (pmh_vec_epi::spec_alpha)
  // MatMultTranspose slot; operators that are symmetric by construction forward to mult
  virtual int     mult_transpose(const double *, double *) { return pmh_set_error(PMH_ERR_SUP, "this operator has no MatMultTranspose slot"); }
  virtual pmh_csr as_csr() { return nullptr; }
  // Operators of the form (scatter) o (middle) o (gather) -- F = B K^+ B' with either K^+ (feti.hip): gather = B' as CSR (rows = entries of mid_in), scatter =
  // B as CSR (rows = dual entries, columns = entries of mid_out); mid_apply runs the middle stage mid_in -> mid_out.  The fused dual-space chain folds the
  // projector into the two sparse stages.
  virtual int     stages(pmh_csr * /*gather*/, double ** /*mid_in*/, pmh_csr * /*scatter*/, const double ** /*mid_out*/) { return PMH_EPI_UNSUPPORTED; }
  virtual int     mid_apply() { return pmh_set_error(PMH_ERR_SUP, "this operator has no middle stage"); }
};

// ---- projector factory -------------------------------------------------------------------------------------
struct pmh_qppf_s {
  pmh_ctx ctx;
  pmh_csr G;
  int     m, n;
  int     orthonormal;
  double *d_inv; // (GG')^{-1}, m x m row-major
  // implicit orthonormalisation (pmh_qppf_create orthonormal = 2): G stays G0 as handed over, the orthonormal-row matrix is T G0 with GG' = LL',
  // T = L^{-1}; d_Tt = T' (row-major) and d_S = T'T = (G0 G0')^{-1} (symmetric), both m x m
  int     implicit_orth;
  double *d_Tt, *d_S, *tmp_m;
  std::vector<double> h_T;
  double *G_left, *Gt_right;
  double  ggt_mfma_ms, host_inverse_ms; // set-up timings (pmh_qppf_setup_stats)
  // QPPFApplyQ's (v,state) -> Qv cache (qppf.c:464-467,495-498) is realised structurally: the penalised
  // operator over a projected operator computes Q x once and reuses it (see PenalizedOp::mult).
};

// ---- reductions ---------------------------------------------------------------------------------------------
enum { PMH_RED_SUM = 0, PMH_RED_MIN = 1 };
// finalise K block-partial arrays (stride = ld) into d_scal[base+k] and h_scal[base+k]
int pmh_finalize_partials(pmh_ctx ctx, const double *partials, int ld, int nblocks, int K, const int *ops, int scal_base, const int *halt = nullptr, int *post_inc = nullptr);
// the same with one scalar slot per quantity: two groups of reductions in ONE launch
int pmh_finalize_partials_slots(pmh_ctx ctx, const double *partials, int ld, int nblocks, int K, const int *ops, const int *slots);
int pmh_vec_grid(int n); // deterministic grid size of the streaming kernels (function of n only)

// vec kernels needed across translation units (device pointers, enqueue only)
int pmh_k_dot_partials(pmh_ctx ctx, int n, const double *x, const double *y, int slot); // -> d_scal/h_scal[slot]
int pmh_qppf_apply_G_norm2(pmh_qppf pf, const double *v, double *Gv, int slot);          // qppf.hip: G v and ||G v||^2 -> scalar slot, enqueue only
int pmh_mpgp_set_pre_test_hook(pmh_mpgp s, int (*f)(void *), void *user);             // mpgp.hip: see there
// from an injected convergence test: rnorm / its threshold (the fused driver's speculation reads it)
int pmh_mpgp_set_convergence_margin(pmh_mpgp s, double margin);
// the next solve starts from a gradient the caller has put into the solver's g (returned): smalxe.hip
int pmh_mpgp_set_gradient_valid(pmh_mpgp s, int valid, double **g);
// mpgp.hip: called right before the speculative Ap = A p of the next iteration is enqueued
int pmh_mpgp_set_pre_p1_hook(pmh_mpgp s, int (*f)(void *), void *user);
// SMALXE's ||B u|| riding on the next product of the penalised operator (qppf.hip): G0 u shares the pass over G0 with G0 x, T (G0 u) and its squared norm are
// finished by workgroup 0 of the projector's kernel -- two launches less per inner iteration, the same bits as pmh_qppf_apply_G_norm2
int pmh_op_penalized_arm_aux_normG(pmh_op op, const double *u, double *Gu, int slot);
int pmh_op_penalized_take_aux_done(pmh_op op);
// the five-launch chain (dualchain.hip): ||T G0 u||^2 of every emitted iterate -> Gu, scalar slot
int pmh_op_penalized_set_normG_target(pmh_op op, double *Gu, int slot);
int pmh_op_penalized_normG_ready(pmh_op op, const double *u);           // 1: slot and Gu hold the values of this u
// launches of the chain's last application (-1: no chain) // 1 if the armed request was served by the last product (and clears it), 0 otherwise (and disarms)
int pmh_op_penalized_chain_launches(pmh_op op);
#define PMH_SLOT_NORMBU2 48 // ||B u||^2 prefetched for SMALXE's inner convergence test
int pmh_host_scalar(pmh_ctx ctx, int slot, double *v);                                  // sync + read h_scal[slot]

// ---- 3x3-block SpMV (bsr.hip) -------------------------------------------------------------------------------------
struct pmh_bsr3_s {
  pmh_ctx   ctx;
  int       n, nbr, ntiles, storage, W, tb; // storage: PMH_BSR_F64 / F32 / F16 (matrix entries); W: blocks per load
  int       nrep, rep_rows;                 // nrep > 1: the tiles hold ONE of nrep congruent diagonal blocks of rep_rows rows each (n = nrep * rep_rows)
  long long nblocks, npad;
  double    scale; // F16: the stored entries are A / scale
  int      *d_tile_br, *d_browptr, *d_bcol;
  long long *d_tile_off;
  void     *d_val;
  std::vector<hipEvent_t> ev; // optional per-launch timing (event pairs on the launch stream)
  std::vector<double>     ev_extra; // bytes of the fused epilogue's own operands of every timed launch
  int                     ev_used, ev_on, ev_seen, ev_stride;
};
typedef pmh_bsr3_s *pmh_bsr3;
enum { PMH_BSR_F64 = 0, PMH_BSR_F32 = 1, PMH_BSR_F16 = 2 };
// fused epilogues of the block kernel (continuing the PMH_EPI_* numbering): see bsr.hip
enum { PMH_BSR_EPI_PRE = 10, PMH_BSR_EPI_POST1 = 11, PMH_BSR_EPI_POST2 = 12 };
template <typename T> struct pmh_bsr3_epi {
  const T *y1;   // ADD / SUB operand; b of the smoothing steps
  const T *dinv; // Jacobi scaling
  T       *r, *d; // POST1 outputs; r is a POST2 input
  double  *z64;  // POST2: optional fp64 copy of the result
  T        c0, c1, c2;
};
// *out = NULL (no error) if A has no usable 3x3 block structure; tile 0 = default; nrep_hint > 1: A is said to be block diagonal with that many congruent
// blocks (checked entry by entry: one device copy then serves all)
int    pmh_bsr3_from_csr(pmh_csr A, int storage, pmh_bsr3 *out, int tile = 0, int nrep_hint = 1);
int    pmh_bsr3_destroy(pmh_bsr3 B);
double pmh_bsr3_bytes(pmh_bsr3 B);           // HBM bytes of one launch (a shared device copy is streamed once)
double pmh_bsr3_bytes_blockdiag(pmh_bsr3 B); // SURVEY 8d's figure of the block-diagonal product (every replica's matrix counted)
int    pmh_bsr3_replicas(pmh_bsr3 B);
int    pmh_bsr3_spmv_f64(pmh_bsr3 B, const double *x, double *y, int epi, const double *y1, const int *halt);
int    pmh_bsr3_spmv_f32(pmh_bsr3 B, const float *x, float *y, int epi, const float *y1, const int *halt);
int    pmh_bsr3_spmv_epi_f64(pmh_bsr3 B, const double *x, double *y, int epi, const pmh_bsr3_epi<double> &e, const int *halt);
int    pmh_bsr3_spmv_epi_f32(pmh_bsr3 B, const float *x, float *y, int epi, const pmh_bsr3_epi<float> &e, const int *halt);
int    pmh_bsr3_timing_enable(pmh_bsr3 B, int max_launches);
int    pmh_bsr3_timing_get(pmh_bsr3 B, int *launches, double *total_ms, double *epilogue_bytes = nullptr);
int    pmh_csr_ensure_transpose(pmh_csr A); // builds A->transpose if missing

// ---- multigrid preconditioner (mg.hip) -------------------------------------------------------------------------
int pmh_mg_apply_halt(pmh_mg mg, const double *b, double *x, const int *halt, bool d0_ready = false); // halt: device flag turning the launches into no-ops
int pmh_mg_fine_d0_slots(pmh_mg mg, const float **dinv, float *itheta, float **d0, float **b32); // see mg.hip
