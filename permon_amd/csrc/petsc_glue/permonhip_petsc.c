/*
 * permonhip_petsc.c -- PETSc-side glue that registers libpermonhip behind PERMON's plugin surface.
 *
 * NOT compiled in this repository's build: PETSc is absent from the build image and from the GPU box
 * (SURVEY.md section 0.2).  A PERMON maintainer adds this file to libpermon (it needs PETSC_DIR and a
 * PETSc configured --with-hip so that Vec data lives on the device: VECHIP / MATAIJHIPSPARSE only provide
 * the device pointers, all arithmetic is libpermonhip's).  It binds exactly the entry points of
 * include/permon_hip.h to the reference's op tables:
 *   _QPSOps  include/permon/private/qpsimpl.h:12-24   -> QPSCreate_MPGPHIP below
 *   _QPCOps  include/permon/private/qpcimpl.h:8-25    -> QPCCreate_BoxHIP   below
 *   Mat mult slots (matblockdiag.c:742-746, gluing.c:280-284, extension.c:1114-1118, matinv.c:957-964)  -> MatMult*_*HIP below
 *   PC apply slot  (pcdual.c:105)                      -> PCApply_DualHIP    below
 * Device handles travel with the PETSc objects as PetscContainers composed under "pmh_*" keys (destroyed with the object).
 * Every VecHIPGetArray* is paired with its Restore: the Restore is what bumps the Vec's state stamp and offload mask, which the
 * reference's caches key on (QPPFApplyQ qppf.c:464,498; SMALXE's BtBu reuse smalxe.c:421-430).
 */
#include <permon/private/qpsimpl.h>
#include <permon/private/qpcimpl.h>
#include <permonmat.h>
#include <permon/private/permonmatimpl.h>
#include <permon/private/permonpcimpl.h>
#include <petscdevice_hip.h>
#include "permon_hip.h"

#define PMHCall(call) \
  do { \
    int pmh_rc_ = (call); \
    PetscCheck(!pmh_rc_, PETSC_COMM_SELF, PETSC_ERR_LIB, "libpermonhip error %d: %s", pmh_rc_, pmh_last_error()); \
  } while (0)

static pmh_ctx PermonHipCtx = NULL; /* one context per rank = one GPU (matblockdiag.c:787-788) */

static PetscErrorCode PermonHipGetCtx(pmh_ctx *ctx)
{
  PetscFunctionBegin;
  /* libpermonhip's index type is int32 (pmh_csr_create, the leaves of pmh_gluing_create): a --with-64-bit-indices PETSc would need
     converted copies of every index array, refused loudly instead of reinterpreting memory */
  PetscCheck(sizeof(PetscInt) == sizeof(int), PETSC_COMM_SELF, PETSC_ERR_SUP, "libpermonhip needs a PETSc with 32-bit PetscInt (sizeof(PetscInt) = %d)", (int)sizeof(PetscInt));
  PetscCheck(sizeof(PetscScalar) == sizeof(double), PETSC_COMM_SELF, PETSC_ERR_SUP, "libpermonhip needs a real double-precision PETSc");
  if (!PermonHipCtx) {
    PetscDeviceContext dctx;
    PetscDevice        dev;
    PetscInt           id;
    PetscCall(PetscDeviceContextGetCurrentContext(&dctx));
    PetscCall(PetscDeviceContextGetDevice(dctx, &dev));
    PetscCall(PetscDeviceGetDeviceId(dev, &id));
    PMHCall(pmh_init((int)id, &PermonHipCtx));
  }
  *ctx = PermonHipCtx;
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ---- device handles as composed containers ------------------------------------------------------------------------- */
typedef struct {
  void *handle;
  int (*destroy)(void *);
} PermonHipHandle;

static PetscErrorCode PermonHipHandleDestroy(void **ctx)
{
  PermonHipHandle *h = (PermonHipHandle *)*ctx;
  PetscFunctionBegin;
  if (h->destroy) (void)h->destroy(h->handle);
  PetscCall(PetscFree(h));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PermonHipCompose(PetscObject obj, const char key[], void *handle, int (*destroy)(void *))
{
  PetscContainer   c;
  PermonHipHandle *h;
  PetscFunctionBegin;
  PetscCall(PetscNew(&h));
  h->handle = handle, h->destroy = destroy;
  PetscCall(PetscContainerCreate(PETSC_COMM_SELF, &c));
  PetscCall(PetscContainerSetPointer(c, h));
  PetscCall(PetscContainerSetCtxDestroy(c, PermonHipHandleDestroy));
  PetscCall(PetscObjectCompose(obj, key, (PetscObject)c));
  PetscCall(PetscContainerDestroy(&c)); /* the object holds the reference */
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PermonHipQuery(PetscObject obj, const char key[], void **handle)
{
  PetscContainer   c;
  PermonHipHandle *h;
  PetscFunctionBegin;
  PetscCall(PetscObjectQuery(obj, key, (PetscObject *)&c));
  PetscCheck(c, PETSC_COMM_SELF, PETSC_ERR_ARG_WRONGSTATE, "object has no device handle \"%s\": call the *AttachHIP routine after its set-up", key);
  PetscCall(PetscContainerGetPointer(c, (void **)&h));
  *handle = h->handle;
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* device CSR of a sequential AIJ matrix (host arrays copied once) */
static PetscErrorCode PermonHipCsrFromSeqAIJ(pmh_ctx ctx, Mat A, pmh_csr *out)
{
  const PetscInt    *ia, *ja;
  const PetscScalar *va;
  PetscInt           m, n;
  PetscBool          done;
  PetscFunctionBegin;
  PetscCall(MatGetSize(A, NULL, &n));
  PetscCall(MatGetRowIJ(A, 0, PETSC_FALSE, PETSC_FALSE, &m, &ia, &ja, &done));
  PetscCheck(done, PETSC_COMM_SELF, PETSC_ERR_SUP, "the local block must be MATSEQAIJ");
  PetscCall(MatSeqAIJGetArrayRead(A, &va));
  PMHCall(pmh_csr_create(ctx, (int)m, (int)n, (const int *)ia, (const int *)ja, va, out));
  PetscCall(MatSeqAIJRestoreArrayRead(A, &va));
  PetscCall(MatRestoreRowIJ(A, 0, PETSC_FALSE, PETSC_FALSE, &m, &ia, &ja, &done));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* the four argument patterns of the Mat / PC slots: device pointers in, kernel, pointers restored (state stamps bumped) */
#define PMH_MAT_XY(mat, key, T, x, y, CALL) \
  do { \
    T                  h_; \
    const PetscScalar *x_; \
    PetscScalar       *y_; \
    PetscCall(PermonHipQuery((PetscObject)(mat), key, (void **)&h_)); \
    PetscCall(VecHIPGetArrayRead(x, &x_)); \
    PetscCall(VecHIPGetArrayWrite(y, &y_)); \
    PMHCall(CALL); \
    PetscCall(VecHIPRestoreArrayWrite(y, &y_)); \
    PetscCall(VecHIPRestoreArrayRead(x, &x_)); \
  } while (0)
/* y = y1 + op(x); y1 may be y (MatMultAdd semantics): read-write access of y in that case */
#define PMH_MAT_XY1Y(mat, key, T, x, y1, y, CALL) \
  do { \
    T                  h_; \
    const PetscScalar *x_, *y1_; \
    PetscScalar       *y_; \
    PetscCall(PermonHipQuery((PetscObject)(mat), key, (void **)&h_)); \
    PetscCall(VecHIPGetArrayRead(x, &x_)); \
    if ((y1) == (y)) { \
      PetscCall(VecHIPGetArray(y, &y_)); \
      y1_ = y_; \
    } else { \
      PetscCall(VecHIPGetArrayRead(y1, &y1_)); \
      PetscCall(VecHIPGetArrayWrite(y, &y_)); \
    } \
    PMHCall(CALL); \
    if ((y1) == (y)) { \
      PetscCall(VecHIPRestoreArray(y, &y_)); \
    } else { \
      PetscCall(VecHIPRestoreArrayWrite(y, &y_)); \
      PetscCall(VecHIPRestoreArrayRead(y1, &y1_)); \
    } \
    PetscCall(VecHIPRestoreArrayRead(x, &x_)); \
  } while (0)

/* ---------------------------------------------------------------------------------------------------
 * QPS type "mpgphip": replaces QPSSolve_MPGP / QPSSetup_MPGP (src/qps/impls/mpgp/mpgp.c:359-650)
 * --------------------------------------------------------------------------------------------------- */
typedef struct {
  pmh_csr       A;
  pmh_op        op;
  pmh_mpgp      solver;
  pmh_mpgp_opts opts;
  const double *b_d, *lb_d, *ub_d; /* device addresses the solver was created on (stable for the life of the VECHIPs) */
  double       *x_d;
} QPS_MPGPHIP;

/* qps->convergencetest must see rnorm / iteration every iteration (mpgp.c:531): SMALXE replaces it
   (smalxe.c:874-875) and reads qps_inner->solQP->x, which is a VECHIP living in the same device memory */
static int QPSMPGPHIPConverged(void *user, int iteration, double rnorm, int *reason)
{
  QPS qps        = (QPS)user;
  qps->iteration = iteration;
  qps->rnorm     = rnorm;
  if ((*qps->convergencetest)(qps, &qps->reason)) return 1;
  *reason = (int)qps->reason;
  return 0;
}

/* device addresses of the QP's vectors; `restore` releases them again (every Get is paired) */
static PetscErrorCode QPSMPGPHIPVecs(QPS qps, PetscBool restore, const PetscScalar **b_d, PetscScalar **x_d, const PetscScalar **lb_d, const PetscScalar **ub_d)
{
  Vec b, x, lb, ub;
  PetscFunctionBegin;
  PetscCall(QPGetRhs(qps->solQP, &b));
  PetscCall(QPGetSolutionVector(qps->solQP, &x));
  PetscCall(QPGetBox(qps->solQP, NULL, &lb, &ub));
  if (!restore) {
    *lb_d = *ub_d = NULL;
    PetscCall(VecHIPGetArrayRead(b, b_d));
    PetscCall(VecHIPGetArray(x, x_d));
    if (lb) PetscCall(VecHIPGetArrayRead(lb, lb_d));
    if (ub) PetscCall(VecHIPGetArrayRead(ub, ub_d));
  } else {
    if (ub) PetscCall(VecHIPRestoreArrayRead(ub, ub_d));
    if (lb) PetscCall(VecHIPRestoreArrayRead(lb, lb_d));
    PetscCall(VecHIPRestoreArray(x, x_d)); /* bumps x's state: SMALXE's BtBu cache and QPPFApplyQ's (v,state) key see the change */
    PetscCall(VecHIPRestoreArrayRead(b, b_d));
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode QPSSetup_MPGPHIP(QPS qps)
{
  QPS_MPGPHIP       *hip = (QPS_MPGPHIP *)qps->data;
  pmh_ctx            ctx;
  Mat                A;
  const PetscScalar *b_d, *lb_d, *ub_d;
  PetscScalar       *x_d;

  PetscFunctionBegin;
  PetscCall(PermonHipGetCtx(&ctx));
  PetscCall(QPGetOperator(qps->solQP, &A));
  PetscCall(PermonHipCsrFromSeqAIJ(ctx, A, &hip->A)); /* host CSR of the local SeqAIJ block -> device (once) */
  PMHCall(pmh_op_create_csr(hip->A, &hip->op));
  PetscCall(QPSMPGPHIPVecs(qps, PETSC_FALSE, &b_d, &x_d, &lb_d, &ub_d));
  hip->opts.rtol = qps->rtol, hip->opts.atol = qps->atol, hip->opts.divtol = qps->divtol, hip->opts.max_it = (int)qps->max_it;
  PMHCall(pmh_mpgp_create(ctx, hip->op, b_d, x_d, lb_d, ub_d, &hip->opts, &hip->solver));
  PMHCall(pmh_mpgp_set_convergence_test(hip->solver, QPSMPGPHIPConverged, qps));
  hip->b_d = b_d, hip->x_d = x_d, hip->lb_d = lb_d, hip->ub_d = ub_d;
  PetscCall(QPSMPGPHIPVecs(qps, PETSC_TRUE, &b_d, &x_d, &lb_d, &ub_d));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode QPSSolve_MPGPHIP(QPS qps)
{
  QPS_MPGPHIP       *hip = (QPS_MPGPHIP *)qps->data;
  pmh_mpgp_stats     st;
  const PetscScalar *b_d, *lb_d, *ub_d;
  PetscScalar       *x_d;

  PetscFunctionBegin;
  /* take the arrays for the duration of the solve: up-to-date on the device, locked against host access, state bumped on return */
  PetscCall(QPSMPGPHIPVecs(qps, PETSC_FALSE, &b_d, &x_d, &lb_d, &ub_d));
  PetscCheck(b_d == hip->b_d && x_d == hip->x_d && lb_d == hip->lb_d && ub_d == hip->ub_d, PetscObjectComm((PetscObject)qps), PETSC_ERR_ARG_WRONGSTATE,
             "a vector of the QP was re-allocated after QPSSetUp: call QPSReset");
  PMHCall(pmh_mpgp_set_tolerances(hip->solver, qps->rtol, qps->atol, qps->divtol, (int)qps->max_it));
  PMHCall(pmh_mpgp_solve(hip->solver));
  PMHCall(pmh_mpgp_get_stats(hip->solver, &st));
  qps->iteration = st.iteration;
  qps->rnorm     = st.rnorm;
  qps->reason    = (KSPConvergedReason)st.reason;
  PetscCall(QPSMPGPHIPVecs(qps, PETSC_TRUE, &b_d, &x_d, &lb_d, &ub_d));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* composed methods SMALXE needs from its inner solver (mpgp.c:858-869) */
static PetscErrorCode QPSMPGPSetOperatorMaxEigenvalue_MPGPHIP(QPS qps, PetscReal maxeig)
{
  QPS_MPGPHIP *hip = (QPS_MPGPHIP *)qps->data;
  PetscFunctionBegin;
  hip->opts.maxeig = maxeig;
  if (hip->solver) PMHCall(pmh_mpgp_set_operator_max_eigenvalue(hip->solver, maxeig));
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode QPSMPGPUpdateMaxEigenvalue_MPGPHIP(QPS qps, PetscReal upd)
{
  QPS_MPGPHIP *hip = (QPS_MPGPHIP *)qps->data;
  PetscFunctionBegin;
  PMHCall(pmh_mpgp_update_max_eigenvalue(hip->solver, upd));
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode QPSMPGPGetCurrentStepType_MPGPHIP(QPS qps, char *stepType)
{
  QPS_MPGPHIP *hip = (QPS_MPGPHIP *)qps->data;
  PetscFunctionBegin;
  PMHCall(pmh_mpgp_get_current_step_type(hip->solver, stepType));
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode QPSResetStatistics_MPGPHIP(QPS qps)
{
  QPS_MPGPHIP *hip = (QPS_MPGPHIP *)qps->data;
  PetscFunctionBegin;
  if (hip->solver) PMHCall(pmh_mpgp_reset_statistics(hip->solver));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode QPSDestroy_MPGPHIP(QPS qps)
{
  QPS_MPGPHIP *hip = (QPS_MPGPHIP *)qps->data;
  PetscFunctionBegin;
  pmh_mpgp_destroy(hip->solver);
  pmh_op_destroy(hip->op);
  pmh_csr_destroy(hip->A);
  PetscCall(PetscObjectComposeFunction((PetscObject)qps, "QPSMPGPSetOperatorMaxEigenvalue_MPGP_C", NULL));
  PetscCall(PetscObjectComposeFunction((PetscObject)qps, "QPSMPGPUpdateMaxEigenvalue_MPGP_C", NULL));
  PetscCall(PetscObjectComposeFunction((PetscObject)qps, "QPSMPGPGetCurrentStepType_MPGP_C", NULL));
  PetscCall(QPSDestroyDefault(qps));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PERMON_EXTERN PetscErrorCode QPSCreate_MPGPHIP(QPS qps)
{
  QPS_MPGPHIP *hip;
  PetscFunctionBegin;
  PetscCall(PetscNew(&hip));
  qps->data = (void *)hip;
  PMHCall(pmh_mpgp_default_opts(&hip->opts));
  qps->ops->setup           = QPSSetup_MPGPHIP;
  qps->ops->solve           = QPSSolve_MPGPHIP;
  qps->ops->destroy         = QPSDestroy_MPGPHIP;
  qps->ops->resetstatistics = QPSResetStatistics_MPGPHIP;
  PetscCall(PetscObjectComposeFunction((PetscObject)qps, "QPSMPGPSetOperatorMaxEigenvalue_MPGP_C", QPSMPGPSetOperatorMaxEigenvalue_MPGPHIP));
  PetscCall(PetscObjectComposeFunction((PetscObject)qps, "QPSMPGPUpdateMaxEigenvalue_MPGP_C", QPSMPGPUpdateMaxEigenvalue_MPGPHIP));
  PetscCall(PetscObjectComposeFunction((PetscObject)qps, "QPSMPGPGetCurrentStepType_MPGP_C", QPSMPGPGetCurrentStepType_MPGPHIP));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ---------------------------------------------------------------------------------------------------
 * QPC type "boxhip": the four _QPCOps slots (include/permon/private/qpcimpl.h:20-23) called by the wrappers of
 * src/qpc/interface/qpc.c:466-615 -- the wrappers' pre-ops (gf = g, gc = 0 at :551-552; gr = gf at :600; the MPI_Allreduce(MIN)
 * of QPCFeas at :521) stay where they are; the kernels overwrite / complete their outputs the same way the CPU ops do.
 * Data = the reference's QPC_Box (src/qpc/impls/box/qpcboximpl.h:5-10), so QPCBoxSet / QPCBoxGet keep working.
 * --------------------------------------------------------------------------------------------------- */
#include <../src/qpc/impls/box/qpcboximpl.h>

static PetscErrorCode QPCBoxHIPBounds(QPC qpc, PetscBool restore, const PetscScalar **lb_d, const PetscScalar **ub_d)
{
  QPC_Box *ctx = (QPC_Box *)qpc->data;
  PetscFunctionBegin;
  if (!restore) {
    *lb_d = *ub_d = NULL;
    if (ctx->lb) PetscCall(VecHIPGetArrayRead(ctx->lb, lb_d));
    if (ctx->ub) PetscCall(VecHIPGetArrayRead(ctx->ub, ub_d));
  } else {
    if (ctx->ub) PetscCall(VecHIPRestoreArrayRead(ctx->ub, ub_d));
    if (ctx->lb) PetscCall(VecHIPRestoreArrayRead(ctx->lb, lb_d));
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode QPCGrads_BoxHIP(QPC qpc, Vec x, Vec g, Vec gf, Vec gc) /* QPCGrads_Box qpcbox.c:21-64 */
{
  const PetscScalar *x_d, *g_d, *lb_d, *ub_d;
  PetscScalar       *gf_d, *gc_d;
  pmh_ctx            h;
  PetscInt           n;
  PetscFunctionBegin;
  PetscCall(PermonHipGetCtx(&h));
  PetscCall(VecGetLocalSize(x, &n));
  PetscCall(QPCBoxHIPBounds(qpc, PETSC_FALSE, &lb_d, &ub_d));
  PetscCall(VecHIPGetArrayRead(x, &x_d));
  PetscCall(VecHIPGetArrayRead(g, &g_d));
  PetscCall(VecHIPGetArrayWrite(gf, &gf_d));
  PetscCall(VecHIPGetArrayWrite(gc, &gc_d));
  PMHCall(pmh_qpc_box_grads(h, (int)n, x_d, g_d, lb_d, ub_d, qpc->astol, gf_d, gc_d)); /* includes gf = g, gc = 0 of qpc.c:551-552 */
  PetscCall(VecHIPRestoreArrayWrite(gc, &gc_d));
  PetscCall(VecHIPRestoreArrayWrite(gf, &gf_d));
  PetscCall(VecHIPRestoreArrayRead(g, &g_d));
  PetscCall(VecHIPRestoreArrayRead(x, &x_d));
  PetscCall(QPCBoxHIPBounds(qpc, PETSC_TRUE, &lb_d, &ub_d));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode QPCGradReduced_BoxHIP(QPC qpc, Vec x, Vec gf, PetscReal alpha, Vec gr) /* QPCGradReduced_Box qpcbox.c:68-100 */
{
  const PetscScalar *x_d, *gf_d, *lb_d, *ub_d;
  PetscScalar       *gr_d;
  pmh_ctx            h;
  PetscInt           n;
  PetscFunctionBegin;
  PetscCall(PermonHipGetCtx(&h));
  PetscCall(VecGetLocalSize(x, &n));
  PetscCall(QPCBoxHIPBounds(qpc, PETSC_FALSE, &lb_d, &ub_d));
  PetscCall(VecHIPGetArrayRead(x, &x_d));
  PetscCall(VecHIPGetArrayRead(gf, &gf_d));
  PetscCall(VecHIPGetArrayWrite(gr, &gr_d));
  PMHCall(pmh_qpc_box_gradreduced(h, (int)n, x_d, gf_d, lb_d, ub_d, alpha, gr_d)); /* includes gr = gf of qpc.c:600 */
  PetscCall(VecHIPRestoreArrayWrite(gr, &gr_d));
  PetscCall(VecHIPRestoreArrayRead(gf, &gf_d));
  PetscCall(VecHIPRestoreArrayRead(x, &x_d));
  PetscCall(QPCBoxHIPBounds(qpc, PETSC_TRUE, &lb_d, &ub_d));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode QPCFeas_BoxHIP(QPC qpc, Vec x, Vec d, PetscReal *alpha) /* QPCFeas_Box qpcbox.c:104-146: the LOCAL minimum */
{
  const PetscScalar *x_d, *d_d, *lb_d, *ub_d;
  pmh_ctx            h;
  PetscInt           n;
  double             a;
  PetscFunctionBegin;
  PetscCall(PermonHipGetCtx(&h));
  PetscCall(VecGetLocalSize(x, &n));
  PetscCall(QPCBoxHIPBounds(qpc, PETSC_FALSE, &lb_d, &ub_d));
  PetscCall(VecHIPGetArrayRead(x, &x_d));
  PetscCall(VecHIPGetArrayRead(d, &d_d));
  PMHCall(pmh_qpc_box_feas(h, (int)n, x_d, d_d, lb_d, ub_d, &a));
  PetscCall(VecHIPRestoreArrayRead(d, &d_d));
  PetscCall(VecHIPRestoreArrayRead(x, &x_d));
  PetscCall(QPCBoxHIPBounds(qpc, PETSC_TRUE, &lb_d, &ub_d));
  *alpha = a; /* QPCFeas (qpc.c:521) completes it with MPI_Allreduce(MIN) over the communicator */
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode QPCProject_BoxHIP(QPC qpc, Vec x, Vec Px) /* QPCProject_Box qpcbox.c:290-305 */
{
  const PetscScalar *x_d, *lb_d, *ub_d;
  PetscScalar       *p_d;
  pmh_ctx            h;
  PetscInt           n;
  PetscFunctionBegin;
  PetscCall(PermonHipGetCtx(&h));
  PetscCall(VecGetLocalSize(x, &n));
  PetscCall(QPCBoxHIPBounds(qpc, PETSC_FALSE, &lb_d, &ub_d));
  if (x == Px) {
    PetscCall(VecHIPGetArray(Px, &p_d));
    PMHCall(pmh_qpc_box_project(h, (int)n, p_d, lb_d, ub_d, p_d));
    PetscCall(VecHIPRestoreArray(Px, &p_d));
  } else {
    PetscCall(VecHIPGetArrayRead(x, &x_d));
    PetscCall(VecHIPGetArrayWrite(Px, &p_d));
    PMHCall(pmh_qpc_box_project(h, (int)n, x_d, lb_d, ub_d, p_d));
    PetscCall(VecHIPRestoreArrayWrite(Px, &p_d));
    PetscCall(VecHIPRestoreArrayRead(x, &x_d));
  }
  PetscCall(QPCBoxHIPBounds(qpc, PETSC_TRUE, &lb_d, &ub_d));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* the reference's constructor fills every other slot (view, viewkkt, getblocksize, ...) and allocates QPC_Box: reuse it, then
   point the four numerical slots at the device (QPCCreate_Box qpcbox.c:450-477 is registered as QPCBOX, qpcreg.c:23-28) */
PERMON_EXTERN PetscErrorCode QPCCreate_Box(QPC qpc);
PERMON_EXTERN PetscErrorCode QPCCreate_BoxHIP(QPC qpc)
{
  PetscFunctionBegin;
  PetscCall(QPCCreate_Box(qpc));
  qpc->ops->grads       = QPCGrads_BoxHIP;
  qpc->ops->gradreduced = QPCGradReduced_BoxHIP;
  qpc->ops->feas        = QPCFeas_BoxHIP;
  qpc->ops->project     = QPCProject_BoxHIP;
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ---------------------------------------------------------------------------------------------------
 * MATBLOCKDIAG (src/mat/impls/blockdiag/matblockdiag.c:190-250, slots :742-746): the rank's sequential block(s) on the device
 * --------------------------------------------------------------------------------------------------- */
static int PermonHipBlockDiagDestroy(void *h) { return pmh_blockdiag_destroy((pmh_blockdiag)h); }
static int PermonHipCsrDestroy(void *h) { return pmh_csr_destroy((pmh_csr)h); }

static PetscErrorCode MatMult_BlockDiagHIP(Mat mat, Vec x, Vec y)
{
  PetscFunctionBegin;
  PMH_MAT_XY(mat, "pmh_blockdiag", pmh_blockdiag, x, y, pmh_blockdiag_mult(h_, x_, y_));
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode MatMultTranspose_BlockDiagHIP(Mat mat, Vec x, Vec y)
{
  PetscFunctionBegin;
  PMH_MAT_XY(mat, "pmh_blockdiag", pmh_blockdiag, x, y, pmh_blockdiag_mult_transpose(h_, x_, y_));
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode MatMultAdd_BlockDiagHIP(Mat mat, Vec x, Vec y1, Vec y)
{
  PetscFunctionBegin;
  PMH_MAT_XY1Y(mat, "pmh_blockdiag", pmh_blockdiag, x, y1, y, pmh_blockdiag_mult_add(h_, x_, y1_, y_));
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode MatMultTransposeAdd_BlockDiagHIP(Mat mat, Vec x, Vec y1, Vec y)
{
  PetscFunctionBegin;
  PMH_MAT_XY1Y(mat, "pmh_blockdiag", pmh_blockdiag, x, y1, y, pmh_blockdiag_mult_transpose_add(h_, x_, y1_, y_));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* after MatCreateBlockDiag (matblockdiag.c:777-854): one sequential block per rank (:787-788) = one block of the pmh_blockdiag */
PERMON_EXTERN PetscErrorCode MatBlockDiagAttachHIP(Mat mat)
{
  Mat_BlockDiag *data = (Mat_BlockDiag *)mat->data;
  pmh_ctx        ctx;
  pmh_csr        K;
  pmh_blockdiag  Kb;
  PetscInt       n;
  int            rowstart[2];
  PetscFunctionBegin;
  PetscCall(PermonHipGetCtx(&ctx));
  PetscCall(PermonHipCsrFromSeqAIJ(ctx, data->localBlock, &K));
  PetscCall(MatGetLocalSize(mat, &n, NULL));
  rowstart[0] = 0, rowstart[1] = (int)n;
  PMHCall(pmh_blockdiag_create(ctx, 1, rowstart, K, &Kb));
  PetscCall(PermonHipCompose((PetscObject)mat, "pmh_csr", K, PermonHipCsrDestroy));
  PetscCall(PermonHipCompose((PetscObject)mat, "pmh_blockdiag", Kb, PermonHipBlockDiagDestroy));
  mat->ops->mult             = MatMult_BlockDiagHIP;
  mat->ops->multtranspose    = MatMultTranspose_BlockDiagHIP;
  mat->ops->multadd          = MatMultAdd_BlockDiagHIP;
  mat->ops->multtransposeadd = MatMultTransposeAdd_BlockDiagHIP;
  PetscCall(MatSetVecType(mat, VECHIP));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ---------------------------------------------------------------------------------------------------
 * MATGLUING (src/mat/impls/gluing/gluing.c:47-199, slots :280-284).  The dual vector is kept REPLICATED on every rank
 * (DESIGN.md section 5): mult is local, mult_transpose ends with the RCCL all-reduce that replaces PetscSFReduce (:144-147).
 * --------------------------------------------------------------------------------------------------- */
static int PermonHipGluingDestroy(void *h) { return pmh_gluing_destroy((pmh_gluing)h); }

static PetscErrorCode MatMult_GluingHIP(Mat mat, Vec right, Vec left) /* x = B' lambda, gluing.c:47-81 */
{
  PetscFunctionBegin;
  PMH_MAT_XY(mat, "pmh_gluing", pmh_gluing, right, left, pmh_gluing_mult(h_, x_, y_));
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode MatMultTranspose_GluingHIP(Mat mat, Vec right, Vec left) /* lambda = B x, gluing.c:125-159 */
{
  PetscFunctionBegin;
  PMH_MAT_XY(mat, "pmh_gluing", pmh_gluing, right, left, pmh_gluing_mult_transpose(h_, x_, y_));
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode MatMultAdd_GluingHIP(Mat mat, Vec right, Vec add, Vec left) /* gluing.c:85-123 */
{
  PetscFunctionBegin;
  PMH_MAT_XY1Y(mat, "pmh_gluing", pmh_gluing, right, add, left, pmh_gluing_mult_add(h_, x_, y1_, y_));
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode MatMultTransposeAdd_GluingHIP(Mat mat, Vec right, Vec add, Vec left) /* gluing.c:163-199 */
{
  PetscFunctionBegin;
  PMH_MAT_XY1Y(mat, "pmh_gluing", pmh_gluing, right, add, left, pmh_gluing_mult_transpose_add(h_, x_, y1_, y_));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* after MatCreateGluing (gluing.c:216-258): leaves (local primal row, sign) from Mat_Gluing, the dual index of every leaf from the
   SF graph (leaf i <-> root iremote[i]: rank * local dual size + index = global dual row, the replicated numbering) */
PERMON_EXTERN PetscErrorCode MatGluingAttachHIP(Mat mat)
{
  Mat_Gluing        *data = (Mat_Gluing *)mat->data;
  pmh_ctx            ctx;
  pmh_gluing         B;
  PetscInt           nroots, nleaves, i, n_x, N_lambda;
  const PetscInt    *ilocal;
  const PetscSFNode *iremote;
  const PetscInt    *ranges;
  int               *root;
  PetscFunctionBegin;
  PetscCall(PermonHipGetCtx(&ctx));
  PetscCall(PetscSFGetGraph(data->SF, &nroots, &nleaves, &ilocal, &iremote));
  PetscCall(MatGetLocalSize(mat, &n_x, NULL));
  PetscCall(MatGetSize(mat, NULL, &N_lambda));
  PetscCall(MatGetOwnershipRangesColumn(mat, &ranges));
  PetscCall(PetscMalloc1(nleaves, &root));
  for (i = 0; i < nleaves; i++) root[ilocal ? ilocal[i] : i] = (int)(ranges[iremote[i].rank] + iremote[i].index);
  PMHCall(pmh_gluing_create(ctx, (int)n_x, (int)N_lambda, (int)nleaves, (const int *)data->leaves_row, root, data->leaves_sign, &B));
  PetscCall(PetscFree(root));
  PetscCall(PermonHipCompose((PetscObject)mat, "pmh_gluing", B, PermonHipGluingDestroy));
  mat->ops->mult             = MatMult_GluingHIP;
  mat->ops->multtranspose    = MatMultTranspose_GluingHIP;
  mat->ops->multadd          = MatMultAdd_GluingHIP;
  mat->ops->multtransposeadd = MatMultTransposeAdd_GluingHIP;
  PetscCall(MatSetVecType(mat, VECHIP));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ---------------------------------------------------------------------------------------------------
 * MATEXTENSION (src/mat/impls/extension/extension.c:476-540, slots :1114-1118): TA = scatter(ris) A gather(cis)
 * --------------------------------------------------------------------------------------------------- */
typedef struct { /* extension.c:4-10 (private to that file in the reference: a maintainer moves it to permonmatimpl.h) */
  Mat        A;
  IS         cis, ris, ris_local;
  Vec        cwork, rwork;
  VecScatter cscatter, rscatter;
  PetscBool  setupcalled, rows_use_global_numbering;
} Mat_Extension;
static int PermonHipExtensionDestroy(void *h) { return pmh_extension_destroy((pmh_extension)h); }

static PetscErrorCode MatMult_ExtensionHIP(Mat TA, Vec c, Vec r) /* extension.c:476-489 */
{
  PetscFunctionBegin;
  PMH_MAT_XY(TA, "pmh_extension", pmh_extension, c, r, pmh_extension_mult(h_, x_, y_));
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode MatMultTranspose_ExtensionHIP(Mat TA, Vec r, Vec c) /* extension.c:510-523 */
{
  PetscFunctionBegin;
  PMH_MAT_XY(TA, "pmh_extension", pmh_extension, r, c, pmh_extension_mult_transpose(h_, x_, y_));
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode MatMultAdd_ExtensionHIP(Mat TA, Vec c, Vec r1, Vec r) /* extension.c:493-506 */
{
  PetscFunctionBegin;
  PMH_MAT_XY1Y(TA, "pmh_extension", pmh_extension, c, r1, r, pmh_extension_mult_add(h_, x_, y1_, y_));
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode MatMultTransposeAdd_ExtensionHIP(Mat TA, Vec r, Vec c1, Vec c) /* extension.c:527-540 */
{
  PetscFunctionBegin;
  PMH_MAT_XY1Y(TA, "pmh_extension", pmh_extension, r, c1, c, pmh_extension_mult_transpose_add(h_, x_, y1_, y_));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* after MatExtensionSetUp (extension.c:233-264): the condensed block A, the column IS (local) and the row IS in the replicated
   dual numbering */
PERMON_EXTERN PetscErrorCode MatExtensionAttachHIP(Mat TA)
{
  Mat_Extension  *data = (Mat_Extension *)TA->data;
  pmh_ctx         ctx;
  pmh_csr         A;
  pmh_extension   E;
  const PetscInt *ris, *cis;
  PetscInt        n_r, n_c;
  PetscFunctionBegin;
  PetscCall(PermonHipGetCtx(&ctx));
  PetscCall(PermonHipCsrFromSeqAIJ(ctx, data->A, &A));
  PetscCall(MatGetSize(TA, &n_r, NULL)); /* rows: the replicated dual vector */
  PetscCall(MatGetLocalSize(TA, NULL, &n_c));
  PetscCall(ISGetIndices(data->ris, &ris));
  PetscCall(ISGetIndices(data->cis, &cis));
  PMHCall(pmh_extension_create(ctx, (int)n_r, (int)n_c, A, (const int *)ris, (const int *)cis, &E));
  PetscCall(ISRestoreIndices(data->cis, &cis));
  PetscCall(ISRestoreIndices(data->ris, &ris));
  PetscCall(PermonHipCompose((PetscObject)TA, "pmh_csr", A, PermonHipCsrDestroy));
  PetscCall(PermonHipCompose((PetscObject)TA, "pmh_extension", E, PermonHipExtensionDestroy));
  TA->ops->mult             = MatMult_ExtensionHIP;
  TA->ops->multtranspose    = MatMultTranspose_ExtensionHIP;
  TA->ops->multadd          = MatMultAdd_ExtensionHIP;
  TA->ops->multtransposeadd = MatMultTransposeAdd_ExtensionHIP;
  PetscCall(MatSetVecType(TA, VECHIP));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ---------------------------------------------------------------------------------------------------
 * MATINV (src/mat/impls/inv/matinv.c:734-743, slot :957): K^+ f on the device.  The inner matrix must be a MATBLOCKDIAG with its
 * device handle attached; tolerances come from the inner KSP (MatInvGetKSP), the kernel from MatInvSetNullSpace (Mat_Inv.R).
 * --------------------------------------------------------------------------------------------------- */
static int PermonHipMatInvDestroy(void *h) { return pmh_matinv_destroy((pmh_matinv)h); }

static PetscErrorCode MatMult_InvHIP(Mat imat, Vec right, Vec left)
{
  PetscFunctionBegin;
  PMH_MAT_XY(imat, "pmh_matinv", pmh_matinv, right, left, pmh_matinv_mult(h_, x_, y_));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PERMON_EXTERN PetscErrorCode MatInvAttachHIP(Mat imat, PetscBool moore_penrose)
{
  Mat_Inv      *inv = (Mat_Inv *)imat->data;
  pmh_blockdiag Kb;
  pmh_matinv    Kp;
  PetscReal     rtol, abstol;
  PetscInt      maxits, n, kdim, k;
  PetscFunctionBegin;
  PetscCall(PermonHipQuery((PetscObject)inv->A, "pmh_blockdiag", (void **)&Kb));
  PetscCall(KSPGetTolerances(inv->ksp, &rtol, &abstol, NULL, &maxits));
  PMHCall(pmh_matinv_create(Kb, rtol, abstol, (int)maxits, 1, &Kp));
  if (moore_penrose && inv->R) { /* P_R K^- P_R (QPTDualize -qpt_dualize_Kplus_mp, qptransform.c:1020-1062): R's local rows, column-major */
    const PetscScalar *r;
    Mat                Rloc;
    PetscCall(MatGetLocalSize(inv->R, &n, NULL));
    PetscCall(MatGetSize(inv->R, NULL, &kdim));
    PetscCall(MatDenseGetLocalMatrix(inv->R, &Rloc));
    PetscCall(MatDenseGetArrayRead(Rloc, &r));
    PetscCall(MatDenseGetLDA(Rloc, &k));
    PetscCheck(k == n, PETSC_COMM_SELF, PETSC_ERR_SUP, "R must be stored with lda = local rows");
    PMHCall(pmh_matinv_set_nullspace(Kp, (int)kdim, r)); /* kdim columns of length n = the layout pmh_matinv_set_nullspace takes */
    PetscCall(MatDenseRestoreArrayRead(Rloc, &r));
  }
  PetscCall(PermonHipCompose((PetscObject)imat, "pmh_matinv", Kp, PermonHipMatInvDestroy));
  imat->ops->mult = MatMult_InvHIP;
  PetscCall(MatSetVecType(imat, VECHIP));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ---------------------------------------------------------------------------------------------------
 * PCDUAL (src/pc/impls/dual/pcdual.c): PCApply_Dual :63-78, set-up :100-118 (queries "Bt" and "K" on F, composed by QPTDualize at
 * qptransform.c:1116-1120).  The lumped preconditioner y = B K B' x in one call on the device.
 * --------------------------------------------------------------------------------------------------- */
typedef struct { /* pcdual.c:9-14 */
  PetscBool  setfromoptionscalled;
  PCDualType pcdualtype;
  Mat        C_bb, At;
  Vec        xwork, ywork;
} PC_Dual;

static PetscErrorCode PCApply_DualHIP(PC pc, Vec x, Vec y)
{
  PC_Dual           *ctx = (PC_Dual *)pc->data;
  pmh_gluing         B;
  pmh_blockdiag      K;
  const PetscScalar *x_d;
  PetscScalar       *y_d;
  PetscFunctionBegin;
  PetscCall(PermonHipQuery((PetscObject)ctx->At, "pmh_gluing", (void **)&B));   /* At = "Bt" of F (pcdual.c:107,111) */
  PetscCall(PermonHipQuery((PetscObject)ctx->C_bb, "pmh_blockdiag", (void **)&K)); /* C_bb = "K" of F (pcdual.c:108,113) */
  PetscCall(VecHIPGetArrayRead(x, &x_d));
  PetscCall(VecHIPGetArrayWrite(y, &y_d));
  PMHCall(pmh_pc_dual_lumped_apply(B, K, x_d, y_d)); /* xwork = B' x; ywork = K xwork; y = B ywork (pcdual.c:69-75) */
  PetscCall(VecHIPRestoreArrayWrite(y, &y_d));
  PetscCall(VecHIPRestoreArrayRead(x, &x_d));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* after PCSetUp_Dual picked PC_DUAL_LUMPED (pcdual.c:105): same queries, device apply */
PERMON_EXTERN PetscErrorCode PCDualAttachHIP(PC pc)
{
  PC_Dual *ctx = (PC_Dual *)pc->data;
  Mat      F, Bt, K;
  PetscFunctionBegin;
  if (ctx->pcdualtype != PC_DUAL_LUMPED) PetscFunctionReturn(PETSC_SUCCESS); /* PC_DUAL_NONE copies (pcdual.c:48-58): nothing to move */
  PetscCall(PCGetOperators(pc, &F, NULL));
  PetscCall(PetscObjectQuery((PetscObject)F, "Bt", (PetscObject *)&Bt));
  PetscCall(PetscObjectQuery((PetscObject)F, "K", (PetscObject *)&K));
  PetscCheck(Bt && K, PetscObjectComm((PetscObject)pc), PETSC_ERR_ARG_WRONGSTATE, "the operator of PCDUAL must carry \"Bt\" and \"K\" (QPTDualize composes them)");
  pc->ops->apply = PCApply_DualHIP;
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ---------------------------------------------------------------------------------------------------
 * MatRegularize for one sequential block (replaces the body of MatRegularize, permonmatregularize.c:241-266, for
 * MAT_REG_EXPLICIT): K_loc (SeqAIJ) and R_loc (SeqDense, p x d, column-major with lda = p) stay host objects at set-up
 * time; rho = MatGetMaxEigenvalue(K_loc,NULL,&rho,1,20) is the caller's (pmh_op_max_eigenvalue on the device copy).
 * --------------------------------------------------------------------------------------------------- */
PERMON_EXTERN PetscErrorCode MatRegularizeLocal_HIP(Mat K_loc, Mat R_loc, PetscReal rho, Mat *Kreg_loc)
{
  PetscInt           p, d, nz;
  const PetscInt    *ia, *ja;
  const PetscScalar *a, *r;
  PetscBool          done;
  PetscInt          *pivots, *ia_new, *ja_new;
  PetscScalar       *a_new;
  long long          nnz;
  PetscFunctionBegin;
  PetscCall(MatGetSize(R_loc, &p, &d));
  PetscCall(MatGetRowIJ(K_loc, 0, PETSC_FALSE, PETSC_FALSE, &p, &ia, &ja, &done));
  PetscCheck(done, PETSC_COMM_SELF, PETSC_ERR_SUP, "K_loc must be MATSEQAIJ");
  PetscCall(MatSeqAIJGetArrayRead(K_loc, &a));
  PetscCall(MatDenseGetArrayRead(R_loc, &r));
  nz = ia[p];
  PetscCall(PetscMalloc4(d, &pivots, p + 1, &ia_new, nz + d * d, &ja_new, nz + d * d, &a_new));
  /* PetscInt is 32 bit here: checked by PermonHipGetCtx */
  {
    pmh_ctx ctx;
    PetscCall(PermonHipGetCtx(&ctx));
  }
  PMHCall(pmh_mat_regularize_csr((int)p, (const int *)ia, (const int *)ja, a, (int)d, r, rho, (int *)pivots, (int *)ia_new, (int *)ja_new, a_new, &nnz));
  PetscCall(MatCreateSeqAIJWithArrays(PETSC_COMM_SELF, p, p, ia_new, ja_new, a_new, Kreg_loc)); /* ownership of the arrays: see MatSeqAIJ docs */
  PetscCall(MatDenseRestoreArrayRead(R_loc, &r));
  PetscCall(MatSeqAIJRestoreArrayRead(K_loc, &a));
  PetscCall(MatRestoreRowIJ(K_loc, 0, PETSC_FALSE, PETSC_FALSE, &p, &ia, &ja, &done));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* registration: called from PermonInitialize next to QPSRegisterAll (src/sys/permoninit.c:86-88) */
PERMON_EXTERN PetscErrorCode PermonHipRegisterAll(void)
{
  PetscFunctionBegin;
  PetscCall(QPSRegister("mpgphip", QPSCreate_MPGPHIP)); /* -qps_type mpgphip, or register as QPSMPGP to override */
  PetscCall(QPCRegister("boxhip", QPCCreate_BoxHIP));   /* QPCSetType(qpc, "boxhip"), or register as QPCBOX to override (qpcreg.c:23-28) */
  /* the Mat / PC slots are attached per object after its set-up: MatBlockDiagAttachHIP, MatGluingAttachHIP, MatExtensionAttachHIP,
     MatInvAttachHIP, PCDualAttachHIP (called from QPTDualize right after it creates B, Bt, K, Kplus and from PCSetUp_Dual) */
  PetscFunctionReturn(PETSC_SUCCESS);
}
