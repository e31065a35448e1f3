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
 *   QPS types "smalxehip", "pcpghip", "ksphip" (QPSSolve_SMALXE smalxe.c:893-997, QPSSolve_PCPG pcpg.c:51-134, QPSSolve_KSP qpsksp.c:127-143)
 *   the operator towers the chain composes: F = B K^+ B' (qptransform.c:1103-1128), P F P (:273-284), A + rho B'B (matpenalized.c:212-243)
 *     -> PermonHipOpFromMat below turns them into ONE pmh_op, so that -qps_type smalxehip / mpgphip run the fused device loops on them
 *   MatInvAttachExplicitHIP: the explicit local dual operators (pmh_fexplicit_*) behind F -- the path bench.py measures
 *   QPPF (qppf.c:454-645) -> QPPFAttachHIP + QPPFApply*_HIP
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
#include <permon/private/qppfimpl.h>
#include <permon/private/qpimpl.h>
#include "permon_hip.h"

/* defined in libpermon but in none of its headers (src/mat/impls/timer/mattimer.c:5): PermonHipOpFromMat recognises a MATTIMER shell by its mult slot */
PERMON_EXTERN PetscErrorCode MatMult_Timer(Mat W, Vec x, Vec y);

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
    {
      /* one rank <-> one GPU: the RCCL communicator that carries the all-reduce of B u (pmh_gluing_mult_transpose; replaces PetscSFReduce, gluing.c:144-147).
         The ncclUniqueId travels over MPI once */
      PetscMPIInt   rank, size;
      unsigned char id128[PMH_UNIQUE_ID_BYTES];
      PetscCallMPI(MPI_Comm_rank(PETSC_COMM_WORLD, &rank));
      PetscCallMPI(MPI_Comm_size(PETSC_COMM_WORLD, &size));
      if (size > 1) {
        if (!rank) PMHCall(pmh_comm_unique_id(id128));
        PetscCallMPI(MPI_Bcast(id128, PMH_UNIQUE_ID_BYTES, MPI_BYTE, 0, PETSC_COMM_WORLD));
        PMHCall(pmh_comm_init(PermonHipCtx, (int)rank, (int)size, id128));
      }
    }
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

/* forward declarations: the operator-tower resolver (defined after the Mat slots) and the QPPF handle */
static PetscErrorCode PermonHipOpFromMat(Mat A, pmh_op *op, PetscInt *nowned, pmh_op owned[]);
static PetscErrorCode PermonHipQPPFHandle(QPPF cp, pmh_qppf *pf);
#define PMH_MAX_TOWER 8

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
 * QPS type "mpgphip": replaces QPSSolve_MPGP / QPSSetup_MPGP (src/qps/impls/mpgp/mpgp.c:359-650) and fills all eight slots of
 * QPSCreate_MPGP's op table (mpgp.c:849-856) and its twelve composed methods (mpgp.c:858-869).
 * The Hessian may be any Mat: a SeqAIJ block (CSR kernels with the fused MPGP epilogues), one of the chain's operator towers
 * (MatPenalized over P F P over F = B K^+ B': PermonHipOpFromMat), or an arbitrary Mat through a shell that calls MatMult.
 * --------------------------------------------------------------------------------------------------- */
typedef struct {
  pmh_op        op;                    /* the Hessian as the library sees it */
  pmh_op        owned[PMH_MAX_TOWER];  /* operators PermonHipOpFromMat created for it (destroyed with the QPS, outermost first) */
  PetscInt      nowned;
  pmh_mpgp      solver;
  pmh_mpgp_opts opts;                  /* QPS_MPGP's parameters (mpgpimpl.h:5-38) in the library's struct */
  PetscBool     borrowed;              /* solver belongs to a pmh_smalxe (QPS type "smalxehip"): view / statistics only */
  const double *b_d, *lb_d, *ub_d;     /* device addresses the solver was created on (stable for the life of the VECHIPs) */
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

static PetscErrorCode QPSMPGPHIPReleaseSolver(QPS_MPGPHIP *hip)
{
  PetscInt i;
  PetscFunctionBegin;
  if (hip->solver && !hip->borrowed) (void)pmh_mpgp_destroy(hip->solver);
  hip->solver = NULL;
  for (i = 0; i < hip->nowned; i++) (void)pmh_op_destroy(hip->owned[i]);
  hip->nowned = 0;
  hip->op     = NULL;
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
  if (hip->borrowed) PetscFunctionReturn(PETSC_SUCCESS); /* inner solver of a "smalxehip": pmh_smalxe_create owns the set-up */
  PetscCall(PermonHipGetCtx(&ctx));
  PetscCall(QPSMPGPHIPReleaseSolver(hip)); /* QPSMPGPSetAlpha & co. clear qps->setupcalled (mpgp.c:68,113): set up again from the current parameters */
  PetscCall(QPGetOperator(qps->solQP, &A));
  /* SeqAIJ -> CSR; MatPenalized / P A P / F = B K^+ B' -> the library's fused towers; anything else -> a shell calling MatMult */
  PetscCall(PermonHipOpFromMat(A, &hip->op, &hip->nowned, hip->owned));
  PetscCall(QPSMPGPHIPVecs(qps, PETSC_FALSE, &b_d, &x_d, &lb_d, &ub_d));
  hip->opts.rtol = qps->rtol, hip->opts.atol = qps->atol, hip->opts.divtol = qps->divtol, hip->opts.max_it = (int)qps->max_it;
  PMHCall(pmh_mpgp_create(ctx, hip->op, b_d, x_d, lb_d, ub_d, &hip->opts, &hip->solver)); /* bound chop, power method (unless maxeig was injected), alpha: mpgp.c:359-428 */
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
  PetscCheck(!hip->borrowed, PetscObjectComm((PetscObject)qps), PETSC_ERR_ARG_WRONGSTATE, "the inner solver of a smalxehip QPS is driven by pmh_smalxe_solve");
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

/* ---- the twelve composed methods of mpgp.c:858-869 (definitions :38-190): parameters live in hip->opts ---- */
static PetscErrorCode QPSMPGPGetCurrentStepType_MPGPHIP(QPS qps, char *stepType) /* mpgp.c:38-45 */
{
  QPS_MPGPHIP *hip = (QPS_MPGPHIP *)qps->data;
  PetscFunctionBegin;
  *stepType = ' ';
  if (hip->solver) PMHCall(pmh_mpgp_get_current_step_type(hip->solver, stepType));
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode QPSMPGPGetAlpha_MPGPHIP(QPS qps, PetscReal *alpha, QPSScalarArgType *argtype) /* mpgp.c:49-58 */
{
  QPS_MPGPHIP *hip = (QPS_MPGPHIP *)qps->data;
  PetscFunctionBegin;
  if (alpha) *alpha = hip->opts.alpha_user;
  if (argtype) *argtype = hip->opts.alpha_direct ? QPS_ARG_DIRECT : QPS_ARG_MULTIPLE;
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode QPSMPGPSetAlpha_MPGPHIP(QPS qps, PetscReal alpha, QPSScalarArgType argtype) /* mpgp.c:62-71 */
{
  QPS_MPGPHIP *hip = (QPS_MPGPHIP *)qps->data;
  PetscFunctionBegin;
  hip->opts.alpha_user   = alpha;
  hip->opts.alpha_direct = (argtype == QPS_ARG_DIRECT);
  qps->setupcalled       = PETSC_FALSE;
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode QPSMPGPGetGamma_MPGPHIP(QPS qps, PetscReal *gamma) /* mpgp.c:75-82 */
{
  PetscFunctionBegin;
  *gamma = ((QPS_MPGPHIP *)qps->data)->opts.gamma;
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode QPSMPGPSetGamma_MPGPHIP(QPS qps, PetscReal gamma) /* mpgp.c:86-93; gamma is read at create time by the fused kernels */
{
  PetscFunctionBegin;
  ((QPS_MPGPHIP *)qps->data)->opts.gamma = gamma;
  qps->setupcalled                        = PETSC_FALSE;
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode QPSMPGPGetOperatorMaxEigenvalue_MPGPHIP(QPS qps, PetscReal *maxeig) /* mpgp.c:97-104 */
{
  QPS_MPGPHIP   *hip = (QPS_MPGPHIP *)qps->data;
  pmh_mpgp_stats st;
  PetscFunctionBegin;
  *maxeig = hip->opts.maxeig;
  if (hip->solver) { /* after the set-up: the value the power method found / SMALXE's updates left */
    PMHCall(pmh_mpgp_get_stats(hip->solver, &st));
    *maxeig = st.maxeig;
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode QPSMPGPSetOperatorMaxEigenvalue_MPGPHIP(QPS qps, PetscReal maxeig) /* mpgp.c:108-116 */
{
  QPS_MPGPHIP *hip = (QPS_MPGPHIP *)qps->data;
  PetscFunctionBegin;
  hip->opts.maxeig = maxeig;
  if (hip->solver) PMHCall(pmh_mpgp_set_operator_max_eigenvalue(hip->solver, maxeig));
  qps->setupcalled = PETSC_FALSE;
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode QPSMPGPUpdateMaxEigenvalue_MPGPHIP(QPS qps, PetscReal upd) /* mpgp.c:119-143 */
{
  QPS_MPGPHIP *hip = (QPS_MPGPHIP *)qps->data;
  PetscFunctionBegin;
  PetscCheck(qps->setupcalled && hip->solver, PetscObjectComm((PetscObject)qps), PETSC_ERR_ARG_WRONGSTATE, "this routine is intended to be called after QPSSetUp");
  PMHCall(pmh_mpgp_update_max_eigenvalue(hip->solver, upd));
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode QPSMPGPGetOperatorMaxEigenvalueTolerance_MPGPHIP(QPS qps, PetscReal *tol) /* mpgp.c:147-154 */
{
  PetscFunctionBegin;
  *tol = ((QPS_MPGPHIP *)qps->data)->opts.maxeig_tol;
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode QPSMPGPSetOperatorMaxEigenvalueTolerance_MPGPHIP(QPS qps, PetscReal tol) /* mpgp.c:158-165 */
{
  PetscFunctionBegin;
  ((QPS_MPGPHIP *)qps->data)->opts.maxeig_tol = tol;
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode QPSMPGPGetOperatorMaxEigenvalueIterations_MPGPHIP(QPS qps, PetscInt *numit) /* mpgp.c:169-176 */
{
  PetscFunctionBegin;
  *numit = ((QPS_MPGPHIP *)qps->data)->opts.maxeig_iter;
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode QPSMPGPSetOperatorMaxEigenvalueIterations_MPGPHIP(QPS qps, PetscInt numit) /* mpgp.c:180-187 */
{
  PetscFunctionBegin;
  ((QPS_MPGPHIP *)qps->data)->opts.maxeig_iter = (int)numit;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static const char *const QPSMPGPHIPComposed[] = {"QPSMPGPGetCurrentStepType_MPGP_C", "QPSMPGPGetAlpha_MPGP_C", "QPSMPGPSetAlpha_MPGP_C", "QPSMPGPGetGamma_MPGP_C", "QPSMPGPSetGamma_MPGP_C",
                                                "QPSMPGPGetOperatorMaxEigenvalue_MPGP_C", "QPSMPGPSetOperatorMaxEigenvalue_MPGP_C", "QPSMPGPSetOperatorMaxEigenvalueTolerance_MPGP_C",
                                                "QPSMPGPGetOperatorMaxEigenvalueTolerance_MPGP_C", "QPSMPGPGetOperatorMaxEigenvalueIterations_MPGP_C",
                                                "QPSMPGPSetOperatorMaxEigenvalueIterations_MPGP_C", "QPSMPGPUpdateMaxEigenvalue_MPGP_C"};

/* ---- the remaining _QPSOps slots (qpsimpl.h:12-24) ---- */
static PetscErrorCode QPSResetStatistics_MPGPHIP(QPS qps) /* mpgp.c:654-664 */
{
  QPS_MPGPHIP *hip = (QPS_MPGPHIP *)qps->data;
  PetscFunctionBegin;
  if (hip->solver) PMHCall(pmh_mpgp_reset_statistics(hip->solver));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode QPSIsQPCompatible_MPGPHIP(QPS qps, QP qp, PetscBool *flg) /* mpgp.c:695-711: no linear constraints left, separable constraints of type box */
{
  Mat Beq, Bineq;
  Vec ceq, cineq;
  QPC qpc;
  PetscFunctionBegin;
  PetscCall(QPGetEq(qp, &Beq, &ceq));
  PetscCall(QPGetIneq(qp, &Bineq, &cineq));
  PetscCall(QPGetQPC(qp, &qpc));
  if (Beq || ceq || Bineq || cineq) {
    *flg = PETSC_FALSE;
  } else {
    PetscCall(PetscObjectTypeCompareAny((PetscObject)qpc, flg, QPCBOX, "boxhip", ""));
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* the option keys and their order are those of QPSSetFromOptions_MPGP (mpgp.c:715-745); the values land in pmh_mpgp_opts */
static PetscErrorCode QPSSetFromOptions_MPGPHIP(QPS qps, PetscOptionItems PetscOptionsObject)
{
  QPS_MPGPHIP *hip = (QPS_MPGPHIP *)qps->data;
  PetscBool    flg1, flg2, alpha_direct, b;
  PetscReal    maxeig, maxeig_tol, alpha, gamma;
  PetscInt     maxeig_iter;
  PetscEnum    e;

  PetscFunctionBegin;
  PetscOptionsHeadBegin(PetscOptionsObject, "QPS MPGP (HIP) options");
  alpha_direct = PETSC_FALSE;
  PetscCall(PetscOptionsBool("-qps_mpgp_alpha_direct", "", "QPSMPGPSetAlpha", (PetscBool)hip->opts.alpha_direct, &alpha_direct, &flg1));
  PetscCall(PetscOptionsReal("-qps_mpgp_alpha", "", "QPSMPGPSetAlpha", hip->opts.alpha_user, &alpha, &flg2));
  if (flg1 || flg2) PetscCall(QPSMPGPSetAlpha(qps, alpha, (QPSScalarArgType)alpha_direct));
  PetscCall(PetscOptionsReal("-qps_mpgp_gamma", "", "QPSMPGPSetGamma", hip->opts.gamma, &gamma, &flg1));
  if (flg1) PetscCall(QPSMPGPSetGamma(qps, gamma));
  PetscCall(PetscOptionsReal("-qps_mpgp_maxeig", "Approximate maximum eigenvalue of the Hessian, PETSC_DECIDE means this is automatically computed.", "QPSMPGPSetOperatorMaxEigenvalue", hip->opts.maxeig, &maxeig, &flg1));
  if (flg1) PetscCall(QPSMPGPSetOperatorMaxEigenvalue(qps, maxeig));
  PetscCall(PetscOptionsReal("-qps_mpgp_maxeig_tol", "Relative tolerance of the power method", "QPSMPGPSetOperatorMaxEigenvalueTolerance", hip->opts.maxeig_tol, &maxeig_tol, &flg1));
  if (flg1) PetscCall(QPSMPGPSetOperatorMaxEigenvalueTolerance(qps, maxeig_tol));
  PetscCall(PetscOptionsInt("-qps_mpgp_maxeig_iter", "Number of iterations of the power method", "QPSMPGPSetOperatorMaxEigenvalueIterations", hip->opts.maxeig_iter, &maxeig_iter, &flg1));
  if (flg1) PetscCall(QPSMPGPSetOperatorMaxEigenvalueIterations(qps, maxeig_iter));
  PetscCall(PetscOptionsReal("-qps_mpgp_btol", "Boundary overshoot tolerance; default: 10*PETSC_MACHINE_EPSILON", "", hip->opts.astol, &hip->opts.astol, &flg1));
  PetscCall(PetscOptionsReal("-qps_mpgp_bound_chop_tol", "Sets boundary to 0 for |boundary|<tol ; default: 0", "", hip->opts.bchop_tol, &hip->opts.bchop_tol, NULL));
  e = (PetscEnum)hip->opts.exptype;
  PetscCall(PetscOptionsEnum("-qps_mpgp_expansion_type", "Set expansion step type", "", QPSMPGPExpansionTypes, e, &e, NULL));
  hip->opts.exptype = (int)e; /* PMH_EXP_* follow QPSMPGPExpansionType (mpgp.c:3) */
  e                 = (PetscEnum)hip->opts.explengthtype;
  PetscCall(PetscOptionsEnum("-qps_mpgp_expansion_length_type", "Set expansion step length type", "", QPSMPGPExpansionLengthTypes, e, &e, NULL));
  hip->opts.explengthtype = (int)e; /* PMH_EXPLEN_* follow QPSMPGPExpansionLengthType (mpgp.c:4) */
  b = (PetscBool)hip->opts.resetalpha;
  PetscCall(PetscOptionsBool("-qps_mpgp_alpha_reset", "If alpha=Nan reset to initial value, otherwise keep last alpha", "QPSMPGPSetAlpha", b, &b, NULL));
  hip->opts.resetalpha = (int)b;
  b = (PetscBool)hip->opts.fallback;
  PetscCall(PetscOptionsBool("-qps_mpgp_fallback", "Throw away expansion step if cost function increased and do a std expansion step.", "", b, &b, NULL));
  hip->opts.fallback = (int)b;
  b = (PetscBool)hip->opts.fallback2;
  PetscCall(PetscOptionsBool("-qps_mpgp_fallback2", "Same as fallback which is done only if the next step is proportioning", "", b, &b, NULL));
  hip->opts.fallback2 = (int)b;
  if (hip->opts.fallback2) hip->opts.fallback = 0;
  PetscOptionsHeadEnd();
  qps->setupcalled = PETSC_FALSE; /* expansion type, fallback, tolerances are read by pmh_mpgp_create */
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* QPSMonitorDefault_MPGP (mpgp.c:21-34): the same line, from the solver's statistics (the norms of the iteration just tested) */
static PetscErrorCode QPSMonitorDefault_MPGPHIP(QPS qps, PetscInt n, PetscViewer viewer)
{
  QPS_MPGPHIP   *hip = (QPS_MPGPHIP *)qps->data;
  pmh_mpgp_stats st;
  PetscFunctionBegin;
  PMHCall(pmh_mpgp_get_stats(hip->solver, &st));
  if (n == 0 && ((PetscObject)qps)->prefix) PetscCall(PetscViewerASCIIPrintf(viewer, "  Projected gradient norms for %s solve.\n", ((PetscObject)qps)->prefix));
  PetscCall(PetscViewerASCIIPrintf(viewer, "%3" PetscInt_FMT " MPGP [%c] ||gp||=%.10e", n, st.current_step_type, (double)qps->rnorm));
  PetscCall(PetscViewerASCIIPrintf(viewer, ",\t||gf||=%.10e", (double)st.gfnorm));
  PetscCall(PetscViewerASCIIPrintf(viewer, ",\t||gc||=%.10e", (double)st.gcnorm));
  PetscCall(PetscViewerASCIIPrintf(viewer, ",\talpha=%.10e", (double)st.alpha));
  PetscCall(PetscViewerASCIIPrintf(viewer, "\n"));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode QPSViewConvergence_MPGPHIP(QPS qps, PetscViewer v) /* mpgp.c:751-770 */
{
  QPS_MPGPHIP   *hip = (QPS_MPGPHIP *)qps->data;
  pmh_mpgp_stats st;
  PetscBool      iascii;
  PetscFunctionBegin;
  PetscCall(PetscObjectTypeCompare((PetscObject)v, PETSCVIEWERASCII, &iascii));
  if (iascii && hip->solver) {
    PMHCall(pmh_mpgp_get_stats(hip->solver, &st));
    PetscCall(PetscViewerASCIIPrintf(v, "from the last QPSReset:\n"));
    PetscCall(PetscViewerASCIIPrintf(v, "number of Hessian multiplications %" PetscInt_FMT "\n", (PetscInt)st.nmv));
    PetscCall(PetscViewerASCIIPrintf(v, "number of CG steps %" PetscInt_FMT "\n", (PetscInt)st.ncg));
    PetscCall(PetscViewerASCIIPrintf(v, "number of expansion steps %" PetscInt_FMT "\n", (PetscInt)st.nexp));
    PetscCall(PetscViewerASCIIPrintf(v, "number of proportioning steps %" PetscInt_FMT "\n", (PetscInt)st.nprop));
    if (hip->opts.fallback || hip->opts.fallback2) {
      PetscCall(PetscViewerASCIIPrintf(v, "number of cost function value increases: %" PetscInt_FMT "\n", (PetscInt)st.nfinc));
      PetscCall(PetscViewerASCIIPrintf(v, "number of fallbacks: %" PetscInt_FMT "\n", (PetscInt)st.nfall));
    }
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode QPSDestroy_MPGPHIP(QPS qps) /* mpgp.c:672-691 */
{
  QPS_MPGPHIP *hip = (QPS_MPGPHIP *)qps->data;
  size_t       i;
  PetscFunctionBegin;
  PetscCall(QPSMPGPHIPReleaseSolver(hip));
  for (i = 0; i < sizeof(QPSMPGPHIPComposed) / sizeof(QPSMPGPHIPComposed[0]); i++) PetscCall(PetscObjectComposeFunction((PetscObject)qps, QPSMPGPHIPComposed[i], NULL));
  PetscCall(QPSDestroyDefault(qps));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PERMON_EXTERN PetscErrorCode QPSCreate_MPGPHIP(QPS qps) /* QPSCreate_MPGP mpgp.c:819-871 */
{
  QPS_MPGPHIP *hip;
  PetscFunctionBegin;
  PetscCall(PetscNew(&hip));
  qps->data = (void *)hip;
  PMHCall(pmh_mpgp_default_opts(&hip->opts)); /* the defaults of mpgp.c:827-843 */
  qps->ops->setup           = QPSSetup_MPGPHIP;
  qps->ops->solve           = QPSSolve_MPGPHIP;
  qps->ops->resetstatistics = QPSResetStatistics_MPGPHIP;
  qps->ops->destroy         = QPSDestroy_MPGPHIP;
  qps->ops->isqpcompatible  = QPSIsQPCompatible_MPGPHIP;
  qps->ops->setfromoptions  = QPSSetFromOptions_MPGPHIP;
  qps->ops->monitor         = QPSMonitorDefault_MPGPHIP;
  qps->ops->viewconvergence = QPSViewConvergence_MPGPHIP;
  /* the reference's public QPSMPGPGet/Set* wrappers dispatch on these names (PetscTryMethod / PetscUseMethod, mpgp.c:873-1100): same keys */
  PetscCall(PetscObjectComposeFunction((PetscObject)qps, "QPSMPGPGetCurrentStepType_MPGP_C", QPSMPGPGetCurrentStepType_MPGPHIP));
  PetscCall(PetscObjectComposeFunction((PetscObject)qps, "QPSMPGPGetAlpha_MPGP_C", QPSMPGPGetAlpha_MPGPHIP));
  PetscCall(PetscObjectComposeFunction((PetscObject)qps, "QPSMPGPSetAlpha_MPGP_C", QPSMPGPSetAlpha_MPGPHIP));
  PetscCall(PetscObjectComposeFunction((PetscObject)qps, "QPSMPGPGetGamma_MPGP_C", QPSMPGPGetGamma_MPGPHIP));
  PetscCall(PetscObjectComposeFunction((PetscObject)qps, "QPSMPGPSetGamma_MPGP_C", QPSMPGPSetGamma_MPGPHIP));
  PetscCall(PetscObjectComposeFunction((PetscObject)qps, "QPSMPGPGetOperatorMaxEigenvalue_MPGP_C", QPSMPGPGetOperatorMaxEigenvalue_MPGPHIP));
  PetscCall(PetscObjectComposeFunction((PetscObject)qps, "QPSMPGPSetOperatorMaxEigenvalue_MPGP_C", QPSMPGPSetOperatorMaxEigenvalue_MPGPHIP));
  PetscCall(PetscObjectComposeFunction((PetscObject)qps, "QPSMPGPSetOperatorMaxEigenvalueTolerance_MPGP_C", QPSMPGPSetOperatorMaxEigenvalueTolerance_MPGPHIP));
  PetscCall(PetscObjectComposeFunction((PetscObject)qps, "QPSMPGPGetOperatorMaxEigenvalueTolerance_MPGP_C", QPSMPGPGetOperatorMaxEigenvalueTolerance_MPGPHIP));
  PetscCall(PetscObjectComposeFunction((PetscObject)qps, "QPSMPGPGetOperatorMaxEigenvalueIterations_MPGP_C", QPSMPGPGetOperatorMaxEigenvalueIterations_MPGPHIP));
  PetscCall(PetscObjectComposeFunction((PetscObject)qps, "QPSMPGPSetOperatorMaxEigenvalueIterations_MPGP_C", QPSMPGPSetOperatorMaxEigenvalueIterations_MPGPHIP));
  PetscCall(PetscObjectComposeFunction((PetscObject)qps, "QPSMPGPUpdateMaxEigenvalue_MPGP_C", QPSMPGPUpdateMaxEigenvalue_MPGPHIP));
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

/* form: 0 = the MATINV as it is (K_reg^{-1}, or a non-singular K); 1 = P_R K^- P_R (QPTDualize -qpt_dualize_Kplus_mp, qptransform.c:1020-1062: MatCreateProd(P_R, Kplus, P_R));
 * 2 = K^- P_R, the LEFT generalised inverse QPTDualize takes when PERMON had to compute the kernel itself (qptransform.c:997-1008, :1040-1062: MatCreateProd(P_R, Kplus) with
 * -regularize 0).  For 1 and 2 the library carries the wrapping (pmh_matinv_set_nullspace / pmh_matinv_set_left_inverse), so that F = B K^+ B' is ONE library operator.
 * The reference's K^- of form 2 is its factorisation with null pivots; here the fixing dofs MatRegularize would take (permonmatregularize.c:57-124) stand in for them: the
 * local block gets identity rows / columns there (a second device copy of K, as pmh_kspfeti_solve builds it) */
typedef enum { PERMONHIP_KPLUS_PLAIN = 0, PERMONHIP_KPLUS_MP = 1, PERMONHIP_KPLUS_LEFT = 2 } PermonHipKplusForm;

PERMON_EXTERN PetscErrorCode MatInvAttachHIPForm(Mat imat, PermonHipKplusForm form)
{
  Mat_Inv      *inv = (Mat_Inv *)imat->data;
  pmh_ctx       ctx;
  pmh_blockdiag Kb;
  pmh_matinv    Kp;
  PetscReal     rtol, abstol;
  PetscInt      maxits, n = 0, kdim = 0, k;
  PetscFunctionBegin;
  PetscCall(PermonHipGetCtx(&ctx));
  PetscCall(PermonHipQuery((PetscObject)inv->A, "pmh_blockdiag", (void **)&Kb));
  PetscCall(KSPGetTolerances(inv->ksp, &rtol, &abstol, NULL, &maxits));
  if (form != PERMONHIP_KPLUS_PLAIN && inv->R) {
    const PetscScalar *r;
    Mat                Rloc;
    PetscCall(MatGetLocalSize(inv->R, &n, NULL));
    PetscCall(MatGetSize(inv->R, NULL, &kdim));
    PetscCall(MatDenseGetLocalMatrix(inv->R, &Rloc));
    PetscCall(MatDenseGetArrayRead(Rloc, &r));
    PetscCall(MatDenseGetLDA(Rloc, &k));
    PetscCheck(k == n, PETSC_COMM_SELF, PETSC_ERR_SUP, "R must be stored with lda = local rows");
    if (form == PERMONHIP_KPLUS_LEFT && kdim > 0) {
      /* the block with identity rows / columns on the fixing dofs: a second CSR + block-diagonal handle, owned by the MATINV */
      Mat_BlockDiag     *bd = (Mat_BlockDiag *)inv->A->data;
      const PetscInt    *ia, *ja;
      const PetscScalar *va;
      PetscInt           m, nz = 0, i, kk;
      PetscBool          done;
      int               *piv, *rp, *ci, rowstart[2];
      char              *isfix;
      double            *vv;
      pmh_csr            Kfix;
      pmh_blockdiag      Kfixb;
      PetscCall(PetscMalloc1(kdim, &piv));
      PMHCall(pmh_mat_regularize_pivots((int)n, (int)kdim, r, piv)); /* R: kdim columns of length n, column-major = the layout it takes */
      PetscCall(MatGetRowIJ(bd->localBlock, 0, PETSC_FALSE, PETSC_FALSE, &m, &ia, &ja, &done));
      PetscCheck(done && m == n, PETSC_COMM_SELF, PETSC_ERR_SUP, "the local block must be MATSEQAIJ of the kernel's row count");
      PetscCall(MatSeqAIJGetArrayRead(bd->localBlock, &va));
      PetscCall(PetscCalloc1(n, &isfix));
      for (kk = 0; kk < kdim; kk++) isfix[piv[kk]] = 1;
      PetscCall(PetscMalloc3(n + 1, &rp, ia[n] + n, &ci, ia[n] + n, &vv));
      rp[0] = 0;
      for (i = 0; i < n; i++) {
        if (isfix[i]) ci[nz] = (int)i, vv[nz] = 1.0, nz++;
        else
          for (kk = ia[i]; kk < ia[i + 1]; kk++)
            if (!isfix[ja[kk]]) ci[nz] = (int)ja[kk], vv[nz] = va[kk], nz++;
        rp[i + 1] = (int)nz;
      }
      PMHCall(pmh_csr_create(ctx, (int)n, (int)n, rp, ci, vv, &Kfix));
      rowstart[0] = 0, rowstart[1] = (int)n;
      PMHCall(pmh_blockdiag_create(ctx, 1, rowstart, Kfix, &Kfixb));
      PetscCall(PermonHipCompose((PetscObject)imat, "pmh_csr_fixed", Kfix, PermonHipCsrDestroy));
      PetscCall(PermonHipCompose((PetscObject)imat, "pmh_blockdiag_fixed", Kfixb, PermonHipBlockDiagDestroy));
      PMHCall(pmh_matinv_create(Kfixb, rtol, abstol, (int)maxits, 1, &Kp));
      PMHCall(pmh_matinv_set_nullspace(Kp, (int)kdim, r));
      PMHCall(pmh_matinv_set_left_inverse(Kp, (int)kdim, piv));
      PetscCall(PetscFree3(rp, ci, vv));
      PetscCall(PetscFree(isfix));
      PetscCall(PetscFree(piv));
      PetscCall(MatSeqAIJRestoreArrayRead(bd->localBlock, &va));
      PetscCall(MatRestoreRowIJ(bd->localBlock, 0, PETSC_FALSE, PETSC_FALSE, &m, &ia, &ja, &done));
    } else {
      PMHCall(pmh_matinv_create(Kb, rtol, abstol, (int)maxits, 1, &Kp));
      PMHCall(pmh_matinv_set_nullspace(Kp, (int)kdim, r)); /* kdim columns of length n = the layout pmh_matinv_set_nullspace takes */
    }
    PetscCall(MatDenseRestoreArrayRead(Rloc, &r));
  } else {
    PMHCall(pmh_matinv_create(Kb, rtol, abstol, (int)maxits, 1, &Kp));
  }
  PetscCall(PermonHipCompose((PetscObject)imat, "pmh_matinv", Kp, PermonHipMatInvDestroy));
  imat->ops->mult = MatMult_InvHIP; /* (with form 1 / 2 this slot already applies the wrapped inverse: use it on its own, not under the reference's MatProd wrapper) */
  PetscCall(MatSetVecType(imat, VECHIP));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PERMON_EXTERN PetscErrorCode MatInvAttachHIP(Mat imat, PetscBool moore_penrose)
{
  PetscFunctionBegin;
  PetscCall(MatInvAttachHIPForm(imat, moore_penrose ? PERMONHIP_KPLUS_MP : PERMONHIP_KPLUS_PLAIN));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* QPTDualize's "Kplus" as composed on F (qptransform.c:1116-1120) is the MATINV itself, or -- after the generalised-inverse wrapping (:1020-1062) -- the product
 * MatCreateProd(P_R, Kplus[, P_R]) that keeps the MATINV composed under "Kplus" (:1047).  Returns the MATINV and which form the wrapper stands for. */
static PetscErrorCode PermonHipUnwrapKplus(Mat Kplus, Mat *imat, PermonHipKplusForm *form)
{
  Mat       inner = NULL;
  PetscBool iscomposite;
  PetscInt  nmat = 0;
  PetscFunctionBegin;
  *imat = Kplus, *form = PERMONHIP_KPLUS_PLAIN;
  PetscCall(PetscObjectTypeCompare((PetscObject)Kplus, MATCOMPOSITE, &iscomposite));
  if (!iscomposite) PetscFunctionReturn(PETSC_SUCCESS);
  PetscCall(PetscObjectQuery((PetscObject)Kplus, "Kplus", (PetscObject *)&inner));
  PetscCheck(inner, PetscObjectComm((PetscObject)Kplus), PETSC_ERR_ARG_WRONGSTATE, "a composite K^+ that does not carry the MATINV it wraps");
  PetscCall(MatCompositeGetNumberMat(Kplus, &nmat));
  PetscCheck(nmat == 2 || nmat == 3, PetscObjectComm((PetscObject)Kplus), PETSC_ERR_SUP, "K^+ wrapper of %" PetscInt_FMT " factors", nmat);
  *imat = inner, *form = (nmat == 3) ? PERMONHIP_KPLUS_MP : PERMONHIP_KPLUS_LEFT;
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

/* ---------------------------------------------------------------------------------------------------
 * QPPF (src/qppf/interface/qppf.c): the projector factory on the device.  QPPFAttachHIP after QPPFSetUp (qppf.c:336-437): G as CSR
 * (replicated: every rank keeps the whole coarse matrix, the reference's -qppf_redundancy taken to its end), the small (G G')^{-1} is
 * formed and applied by the library.  orthonormal rows: 1 = explicitly (QPTOrthonormalizeEq formed T G), 2 = implicitly (the default form
 * -qp_E_orth_form implicit, qptransform.c:647: cp->G is a dummy that carries the un-orthonormalised G0 as "MatOrthColumns_Implicit_A",
 * permonmatorth.c:176-205; the library keeps G0 sparse and applies T = chol(G0 G0')^{-1} inside its G v kernels).
 * --------------------------------------------------------------------------------------------------- */
static int PermonHipQPPFDestroy(void *h) { return pmh_qppf_destroy((pmh_qppf)h); }

/* device CSR of a (possibly distributed) AIJ matrix, every rank holding all rows (MatCreateRedundantMatrix over the object's communicator) */
static PetscErrorCode PermonHipCsrFromMatReplicated(pmh_ctx ctx, Mat A, pmh_csr *out)
{
  PetscMPIInt size;
  Mat         Ared = A;
  PetscFunctionBegin;
  PetscCallMPI(MPI_Comm_size(PetscObjectComm((PetscObject)A), &size));
  if (size > 1) PetscCall(MatCreateRedundantMatrix(A, size, PETSC_COMM_SELF, MAT_INITIAL_MATRIX, &Ared));
  PetscCall(PermonHipCsrFromSeqAIJ(ctx, Ared, out));
  if (Ared != A) PetscCall(MatDestroy(&Ared));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PERMON_EXTERN PetscErrorCode QPPFAttachHIP(QPPF cp)
{
  pmh_ctx   ctx;
  pmh_csr   G;
  pmh_qppf  pf;
  Mat       G0 = NULL;
  PetscBool orth;
  int       mode;
  PetscFunctionBegin;
  PetscCall(PermonHipGetCtx(&ctx));
  PetscCall(QPPFSetUp(cp));
  PetscCall(PetscObjectQuery((PetscObject)cp->G, "MatOrthColumns_Implicit_A", (PetscObject *)&G0));
  PetscCall(QPPFGetGHasOrthonormalRows(cp, &orth));
  mode = G0 ? 2 : (orth ? 1 : 0);
  PetscCall(PermonHipCsrFromMatReplicated(ctx, G0 ? G0 : cp->G, &G));
  PMHCall(pmh_qppf_create(ctx, G, mode, &pf)); /* GG' on the fp64 matrix cores, Cholesky / inverse of the m x m matrix on the host (QPPFSetUpGGt/GGtinv_Private qppf.c:213-333) */
  PetscCall(PermonHipCompose((PetscObject)cp, "pmh_csr", G, PermonHipCsrDestroy));
  PetscCall(PermonHipCompose((PetscObject)cp, "pmh_qppf", pf, PermonHipQPPFDestroy));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* the handle, attached on first use */
static PetscErrorCode PermonHipQPPFHandle(QPPF cp, pmh_qppf *pf)
{
  PetscContainer c;
  PetscFunctionBegin;
  PetscCall(PetscObjectQuery((PetscObject)cp, "pmh_qppf", (PetscObject *)&c));
  if (!c) PetscCall(QPPFAttachHIP(cp));
  PetscCall(PermonHipQuery((PetscObject)cp, "pmh_qppf", (void **)pf));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* the right-hand side of the implicitly orthonormalised constraint: cE = T cE0 (QPTOrthonormalizeEq, qptransform.c:600-625; host vectors of length m) */
PERMON_EXTERN PetscErrorCode QPPFOrthRhs_HIP(QPPF cp, const PetscScalar e0[], PetscScalar e[])
{
  pmh_qppf pf;
  PetscFunctionBegin;
  PetscCall(PermonHipQPPFHandle(cp, &pf));
  PMHCall(pmh_qppf_orth_rhs(pf, e0, e));
  PetscFunctionReturn(PETSC_SUCCESS);
}

#define PMH_QPPF_XY(cp, x, y, CALL) \
  do { \
    pmh_qppf           h_; \
    const PetscScalar *x_; \
    PetscScalar       *y_; \
    PetscCall(PermonHipQPPFHandle(cp, &h_)); \
    PetscCall(VecHIPGetArrayRead(x, &x_)); \
    PetscCall(VecHIPGetArrayWrite(y, &y_)); \
    PMHCall(CALL); \
    PetscCall(VecHIPRestoreArrayWrite(y, &y_)); \
    PetscCall(VecHIPRestoreArrayRead(x, &x_)); \
  } while (0)

/* QPPFApplyQ qppf.c:454-503.  The reference's (v, state) -> Qv cache (:464-467, :495-498) stays where it is: a caller that wants it wraps this
   routine exactly as QPPFApplyQ wraps its MatMult sequence; the fused towers below reuse Q x structurally instead */
PERMON_EXTERN PetscErrorCode QPPFApplyQ_HIP(QPPF cp, Vec v, Vec Qv)
{
  PetscFunctionBegin;
  PMH_QPPF_XY(cp, v, Qv, pmh_qppf_apply_Q(h_, x_, y_));
  PetscFunctionReturn(PETSC_SUCCESS);
}
PERMON_EXTERN PetscErrorCode QPPFApplyP_HIP(QPPF cp, Vec v, Vec Pv) /* qppf.c:563-575 */
{
  PetscFunctionBegin;
  PMH_QPPF_XY(cp, v, Pv, pmh_qppf_apply_P(h_, x_, y_));
  PetscFunctionReturn(PETSC_SUCCESS);
}
PERMON_EXTERN PetscErrorCode QPPFApplyGtG_HIP(QPPF cp, Vec v, Vec GtGv) /* qppf.c:580-605 */
{
  PetscFunctionBegin;
  PMH_QPPF_XY(cp, v, GtGv, pmh_qppf_apply_GtG(h_, x_, y_));
  PetscFunctionReturn(PETSC_SUCCESS);
}
PERMON_EXTERN PetscErrorCode QPPFApplyCP_HIP(QPPF cp, Vec x, Vec y) /* qppf.c:610-645: y = (GG')^{-1} x */
{
  PetscFunctionBegin;
  PMH_QPPF_XY(cp, x, y, pmh_qppf_apply_CP(h_, x_, y_));
  PetscFunctionReturn(PETSC_SUCCESS);
}
PERMON_EXTERN PetscErrorCode QPPFApplyHalfQ_HIP(QPPF cp, Vec x, Vec y) /* qppf.c:507-527 */
{
  PetscFunctionBegin;
  PMH_QPPF_XY(cp, x, y, pmh_qppf_apply_halfQ(h_, x_, y_));
  PetscFunctionReturn(PETSC_SUCCESS);
}
PERMON_EXTERN PetscErrorCode QPPFApplyHalfQTranspose_HIP(QPPF cp, Vec x, Vec y) /* qppf.c:531-559 */
{
  PetscFunctionBegin;
  PMH_QPPF_XY(cp, x, y, pmh_qppf_apply_halfQ_transpose(h_, x_, y_));
  PetscFunctionReturn(PETSC_SUCCESS);
}
PERMON_EXTERN PetscErrorCode QPPFApplyG_HIP(QPPF cp, Vec v, Vec Gv) /* MatMult(cp->G, v, cp->G_left) qppf.c:475 */
{
  PetscFunctionBegin;
  PMH_QPPF_XY(cp, v, Gv, pmh_qppf_apply_G(h_, x_, y_));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* the MatShells QPPFCreateP / QPPFCreateGtG hand out (qppf.c:687-718; their MATOP_MULT is QPPFMatMult_P / _GtG): device slots */
static PetscErrorCode MatMult_QPPF_P_HIP(Mat P, Vec x, Vec y)
{
  QPPF cp;
  PetscFunctionBegin;
  PetscCall(MatShellGetContext(P, (void *)&cp));
  PetscCall(QPPFApplyP_HIP(cp, x, y));
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode MatMult_QPPF_GtG_HIP(Mat GtG, Vec x, Vec y)
{
  QPPF cp;
  PetscFunctionBegin;
  PetscCall(MatShellGetContext(GtG, (void *)&cp));
  PetscCall(QPPFApplyGtG_HIP(cp, x, y));
  PetscFunctionReturn(PETSC_SUCCESS);
}
/* after QPPFCreateP / QPPFCreateGtG: which == 0 the projector P, 1 the penalised term G'G */
PERMON_EXTERN PetscErrorCode QPPFShellAttachHIP(Mat shell, PetscInt which)
{
  PetscFunctionBegin;
  PetscCall(MatShellSetOperation(shell, MATOP_MULT, which ? (PetscErrorCodeFn *)MatMult_QPPF_GtG_HIP : (PetscErrorCodeFn *)MatMult_QPPF_P_HIP));
  PetscCall(MatSetVecType(shell, VECHIP));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ---------------------------------------------------------------------------------------------------
 * Operator towers.  The chain composes the Hessian of the QP the solver finally sees out of shells:
 *   F      = Timer(Prod(Timer(Bt), Timer(Kplus), Timer(B)))  with "B", "K", "Kplus", "Bt" composed on it     (QPTDualize, qptransform.c:1103-1128)
 *   P F P  = Prod(P, F, P) (box present) or P F = Prod(F, P)                                                 (QPTEnforceEqByProjector, :273-284; matprod.c:42-48)
 *   A_rho  = MatPenalized{A = P F P, BtB = QPPFCreateGtG, rho}                                               (QPTEnforceEqByPenalty -> MatCreatePenalized, matpenalized.c:212-243)
 * Applied slot by slot these cost one launch (and one VecHIPGet/Restore pair) per shell level; PermonHipOpFromMat recognises the
 * towers and builds ONE library operator for the whole thing -- pmh_op_create_penalized(pmh_op_create_projected(pmh_op_create_feti_dual(B, K^+)))
 * -- whose mult is the fused sequence bench.py measures (8 launches per A_rho x with the explicit dual operators attached to K^+: G0 x chunk sums,
 * k_gt_fused1d, B_c' -> X, k_fxo_gemm4, k_fxo_fin, B_c Y, G0 w chunk sums, k_gt_fused1d).
 * *op is borrowed when a handle is already composed on A; operators created here are appended to owned[] (the caller destroys them, outermost first).
 * --------------------------------------------------------------------------------------------------- */
typedef struct { /* matpenalized.c:4-8 (private to that file in the reference: a maintainer moves it to permonmatimpl.h) */
  Mat       A, BtB;
  PetscReal rho;
  Vec       xwork;
} Mat_Penalized;

typedef struct { /* an arbitrary Mat behind pmh_op_create_shell: two VECHIPs without storage of their own, placed on the device pointers of every call */
  Mat A;
  Vec x, y;
} PermonHipShell;

static int PermonHipShellMult(void *user, const double *x_dev, double *y_dev) /* pmh_shell_mult_fn */
{
  PermonHipShell *sh = (PermonHipShell *)user;
  if (VecHIPPlaceArray(sh->x, (PetscScalar *)x_dev)) return 1;
  if (VecHIPPlaceArray(sh->y, y_dev)) return 1;
  if (MatMult(sh->A, sh->x, sh->y)) return 1;
  if (VecHIPResetArray(sh->y)) return 1;
  if (VecHIPResetArray(sh->x)) return 1;
  return 0;
}

static PetscErrorCode PermonHipShellDestroy(void **ctx)
{
  PermonHipShell *sh = (PermonHipShell *)*ctx;
  PetscFunctionBegin;
  PetscCall(VecDestroy(&sh->x));
  PetscCall(VecDestroy(&sh->y));
  PetscCall(PetscFree(sh));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PermonHipOpOwn(pmh_op op, PetscInt *nowned, pmh_op owned[])
{
  PetscFunctionBegin;
  PetscCheck(*nowned < PMH_MAX_TOWER, PETSC_COMM_SELF, PETSC_ERR_SUP, "operator tower deeper than %d levels", PMH_MAX_TOWER);
  /* outermost first: a level is created after the levels it is built on, and must be destroyed before them */
  for (PetscInt i = *nowned; i > 0; i--) owned[i] = owned[i - 1];
  owned[0] = op;
  (*nowned)++;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PermonHipOpFromMat(Mat A, pmh_op *op, PetscInt *nowned, pmh_op owned[])
{
  pmh_ctx        ctx;
  PetscContainer c;
  PetscBool      isaij, iscomposite, isshell;
  Mat            Bt = NULL, Kplus = NULL, inner = NULL;
  void (*fpen)(void) = NULL;

  PetscFunctionBegin;
  PetscCall(PermonHipGetCtx(&ctx));
  /* 0. a handle composed earlier (MatPenalizedAttachHIP, a previous set-up) */
  PetscCall(PetscObjectQuery((PetscObject)A, "pmh_op", (PetscObject *)&c));
  if (c) {
    PetscCall(PermonHipQuery((PetscObject)A, "pmh_op", (void **)op));
    PetscFunctionReturn(PETSC_SUCCESS);
  }
  /* 1. a sequential AIJ block: the CSR kernels with the fused MPGP epilogues (BASELINE configs[1]) */
  PetscCall(PetscObjectTypeCompareAny((PetscObject)A, &isaij, MATSEQAIJ, MATSEQAIJHIPSPARSE, ""));
  if (isaij) {
    pmh_csr csr;
    PetscCall(PermonHipCsrFromSeqAIJ(ctx, A, &csr)); /* host CSR of the local block -> device (once) */
    PetscCall(PermonHipCompose((PetscObject)A, "pmh_csr", csr, PermonHipCsrDestroy));
    PMHCall(pmh_op_create_csr(csr, op)); /* borrows csr, which lives with A */
    PetscCall(PermonHipOpOwn(*op, nowned, owned));
    PetscFunctionReturn(PETSC_SUCCESS);
  }
  /* 2. A_rho = A + rho B'B (MatCreatePenalized composes its accessors under these names, matpenalized.c:238-241) */
  PetscCall(PetscObjectQueryFunction((PetscObject)A, "MatPenalizedGetPenalizedTerm_Penalty_C", &fpen));
  if (fpen) {
    Mat_Penalized *pen;
    QPPF           cp = NULL;
    pmh_qppf       pf;
    pmh_op         opA;
    PetscCall(MatShellGetContext(A, (void *)&pen));
    PetscCall(PetscObjectQuery((PetscObject)pen->BtB, "qppf", (PetscObject *)&cp)); /* QPPFCreateGtG composes it (qppf.c:714) */
    PetscCheck(cp, PetscObjectComm((PetscObject)A), PETSC_ERR_ARG_WRONGSTATE, "the penalised term carries no QPPF");
    PetscCall(PermonHipQPPFHandle(cp, &pf));
    PetscCall(PermonHipOpFromMat(pen->A, &opA, nowned, owned));
    PMHCall(pmh_op_create_penalized(opA, pf, pen->rho, op));
    PetscCall(PermonHipOpOwn(*op, nowned, owned));
    PetscFunctionReturn(PETSC_SUCCESS);
  }
  /* 3. F = B K^+ B' of QPTDualize: recognised by what it composes on itself (qptransform.c:1116-1120), whatever Timer / Prod shells sit in between */
  PetscCall(PetscObjectQuery((PetscObject)A, "Bt", (PetscObject *)&Bt));
  PetscCall(PetscObjectQuery((PetscObject)A, "Kplus", (PetscObject *)&Kplus));
  if (Bt && Kplus) {
    pmh_gluing         B;
    pmh_matinv         Kp;
    Mat                imat;
    PermonHipKplusForm form;
    PetscCall(PermonHipUnwrapKplus(Kplus, &imat, &form)); /* the MATINV under QPTDualize's MatCreateProd(P_R, Kplus[, P_R]): its library handle carries the wrapping */
    PetscCall(PermonHipQuery((PetscObject)Bt, "pmh_gluing", (void **)&B));    /* MatGluingAttachHIP(Bt) */
    PetscCall(PermonHipQuery((PetscObject)imat, "pmh_matinv", (void **)&Kp)); /* MatInvAttachHIPForm(imat, form) [+ MatInvAttachExplicitHIP] */
    PMHCall(pmh_op_create_feti_dual(B, Kp, op));                               /* applies through the explicit dual operators when they are attached to Kp */
    PetscCall(PermonHipOpOwn(*op, nowned, owned));
    PetscFunctionReturn(PETSC_SUCCESS);
  }
  /* 4. P A P / P A: MATCOMPOSITE of type multiplicative, product = mat[n-1] ... mat[0] (matprod.c:34), with the shells of QPPFCreateP (they carry "qppf") */
  PetscCall(PetscObjectTypeCompare((PetscObject)A, MATCOMPOSITE, &iscomposite));
  if (iscomposite) {
    PetscInt nmat;
    Mat      M0, M1, M2 = NULL;
    QPPF     cp0 = NULL, cp1 = NULL, cp2 = NULL;
    PetscCall(MatCompositeGetNumberMat(A, &nmat));
    if (nmat == 2 || nmat == 3) {
      PetscCall(MatCompositeGetMat(A, 0, &M0));
      PetscCall(MatCompositeGetMat(A, 1, &M1));
      if (nmat == 3) PetscCall(MatCompositeGetMat(A, 2, &M2));
      PetscCall(PetscObjectQuery((PetscObject)M0, "qppf", (PetscObject *)&cp0));
      PetscCall(PetscObjectQuery((PetscObject)M1, "qppf", (PetscObject *)&cp1));
      if (M2) PetscCall(PetscObjectQuery((PetscObject)M2, "qppf", (PetscObject *)&cp2));
      if (nmat == 3 && cp0 && cp0 == cp2 && !cp1) { /* A_arr = {P, A, P}: P A P (qptransform.c:279-283) */
        pmh_qppf pf;
        pmh_op   opA;
        PetscCall(PermonHipQPPFHandle(cp0, &pf));
        PetscCall(PermonHipOpFromMat(M1, &opA, nowned, owned));
        PMHCall(pmh_op_create_projected(opA, pf, 1, op));
        PetscCall(PermonHipOpOwn(*op, nowned, owned));
        PetscFunctionReturn(PETSC_SUCCESS);
      }
      if (nmat == 2 && cp1 && !cp0) { /* A_arr = {A, P}: P A (qptransform.c:273-277) */
        pmh_qppf pf;
        pmh_op   opA;
        PetscCall(PermonHipQPPFHandle(cp1, &pf));
        PetscCall(PermonHipOpFromMat(M0, &opA, nowned, owned));
        PMHCall(pmh_op_create_projected(opA, pf, 0, op));
        PetscCall(PermonHipOpOwn(*op, nowned, owned));
        PetscFunctionReturn(PETSC_SUCCESS);
      }
    }
  }
  /* 5. a MatTimer (mattimer.c:80-104): look through it */
  PetscCall(PetscObjectTypeCompare((PetscObject)A, MATSHELL, &isshell));
  if (isshell) {
    PetscErrorCode (*mult)(Mat, Vec, Vec) = NULL;
    PetscCall(MatShellGetOperation(A, MATOP_MULT, (PetscErrorCodeFn **)&mult));
    if (mult == MatMult_Timer) {
      PetscCall(MatTimerGetMat(A, &inner));
      PetscCall(PermonHipOpFromMat(inner, op, nowned, owned));
      PetscFunctionReturn(PETSC_SUCCESS);
    }
  }
  /* 6. anything else: a shell operator whose mult calls MatMult on VECHIPs placed over the library's device pointers (MatCreateShellPermon's role, shell.c:5-31) */
  {
    PermonHipShell *sh;
    PetscContainer  holder;
    PetscInt        n;
    PetscCall(MatGetLocalSize(A, &n, NULL));
    PetscCall(PetscNew(&sh));
    sh->A = A; /* borrowed: the handle is composed on A and dies with it */
    PetscCall(VecCreateSeqHIPWithArray(PETSC_COMM_SELF, 1, n, NULL, &sh->x));
    PetscCall(VecCreateSeqHIPWithArray(PETSC_COMM_SELF, 1, n, NULL, &sh->y));
    PetscCall(PetscContainerCreate(PETSC_COMM_SELF, &holder));
    PetscCall(PetscContainerSetPointer(holder, sh));
    PetscCall(PetscContainerSetCtxDestroy(holder, PermonHipShellDestroy));
    PetscCall(PetscObjectCompose((PetscObject)A, "pmh_shell_ctx", (PetscObject)holder));
    PetscCall(PetscContainerDestroy(&holder));
    PMHCall(pmh_op_create_shell(ctx, (int)n, PermonHipShellMult, sh, op));
    PetscCall(PermonHipOpOwn(*op, nowned, owned));
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ---------------------------------------------------------------------------------------------------
 * MatPenalized (src/qp/utils/matpenalized.c): the four slots MatCreatePenalized registers (:232-235) and the penalty accessors SMALXE's
 * update uses (MatPenalizedUpdatePenalty, smalxe.c:391-393) on ONE library operator for the whole tower.
 * --------------------------------------------------------------------------------------------------- */
typedef struct {
  pmh_op   op;
  pmh_op   owned[PMH_MAX_TOWER];
  PetscInt nowned;
} PermonHipTower;

static int PermonHipTowerDestroy(void *h)
{
  PermonHipTower *t = (PermonHipTower *)h;
  for (PetscInt i = 0; i < t->nowned; i++) (void)pmh_op_destroy(t->owned[i]);
  (void)PetscFree(t);
  return 0;
}

static PetscErrorCode MatMult_PenalizedHIP(Mat Arho, Vec x, Vec y) /* matpenalized.c:12-22 */
{
  PetscFunctionBegin;
  PMH_MAT_XY(Arho, "pmh_tower", PermonHipTower *, x, y, pmh_op_mult(h_->op, x_, y_));
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode MatMultTranspose_PenalizedHIP(Mat Arho, Vec x, Vec y) /* matpenalized.c:26-36 */
{
  PetscFunctionBegin;
  PMH_MAT_XY(Arho, "pmh_tower", PermonHipTower *, x, y, pmh_op_mult_transpose(h_->op, x_, y_));
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode MatMultAdd_PenalizedHIP(Mat Arho, Vec x, Vec x2, Vec y) /* matpenalized.c:40-57 */
{
  PetscFunctionBegin;
  PMH_MAT_XY1Y(Arho, "pmh_tower", PermonHipTower *, x, x2, y, pmh_op_penalized_mult_add(h_->op, x_, y1_, y_));
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode MatMultTransposeAdd_PenalizedHIP(Mat Arho, Vec x, Vec x2, Vec y) /* matpenalized.c:61-78 */
{
  PetscFunctionBegin;
  PMH_MAT_XY1Y(Arho, "pmh_tower", PermonHipTower *, x, x2, y, pmh_op_penalized_mult_transpose_add(h_->op, x_, y1_, y_));
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode MatPenalizedSetPenalty_PenaltyHIP(Mat Arho, PetscReal rho) /* matpenalized.c:122-131 */
{
  Mat_Penalized  *pen;
  PermonHipTower *t;
  PetscFunctionBegin;
  PetscCall(MatShellGetContext(Arho, (void *)&pen));
  PetscCall(PermonHipQuery((PetscObject)Arho, "pmh_tower", (void **)&t));
  pen->rho = rho;
  PMHCall(pmh_op_penalized_set_penalty(t->op, rho));
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode MatPenalizedUpdatePenalty_PenaltyHIP(Mat Arho, PetscReal rho_update) /* matpenalized.c:135-144 */
{
  Mat_Penalized  *pen;
  PermonHipTower *t;
  double          rho;
  PetscFunctionBegin;
  PetscCall(MatShellGetContext(Arho, (void *)&pen));
  PetscCall(PermonHipQuery((PetscObject)Arho, "pmh_tower", (void **)&t));
  PMHCall(pmh_op_penalized_get_penalty(t->op, &rho));
  pen->rho = rho * rho_update;
  PMHCall(pmh_op_penalized_set_penalty(t->op, pen->rho));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* after MatCreatePenalized (called from QPTEnforceEqByPenalty, qptransform.c:329-410) */
PERMON_EXTERN PetscErrorCode MatPenalizedAttachHIP(Mat Arho)
{
  PermonHipTower *t;
  PetscFunctionBegin;
  PetscCall(PetscNew(&t));
  PetscCall(PermonHipOpFromMat(Arho, &t->op, &t->nowned, t->owned));
  PetscCall(PermonHipCompose((PetscObject)Arho, "pmh_tower", t, PermonHipTowerDestroy));
  PetscCall(PermonHipCompose((PetscObject)Arho, "pmh_op", t->op, NULL)); /* what QPSSetup_MPGPHIP / _SMALXEHIP pick up (borrowed from the tower) */
  PetscCall(MatShellSetOperation(Arho, MATOP_MULT, (PetscErrorCodeFn *)MatMult_PenalizedHIP));
  PetscCall(MatShellSetOperation(Arho, MATOP_MULT_ADD, (PetscErrorCodeFn *)MatMultAdd_PenalizedHIP));
  PetscCall(MatShellSetOperation(Arho, MATOP_MULT_TRANSPOSE, (PetscErrorCodeFn *)MatMultTranspose_PenalizedHIP));
  PetscCall(MatShellSetOperation(Arho, MATOP_MULT_TRANSPOSE_ADD, (PetscErrorCodeFn *)MatMultTransposeAdd_PenalizedHIP));
  PetscCall(PetscObjectComposeFunction((PetscObject)Arho, "MatPenalizedSetPenalty_Penalty_C", MatPenalizedSetPenalty_PenaltyHIP));
  PetscCall(PetscObjectComposeFunction((PetscObject)Arho, "MatPenalizedUpdatePenalty_Penalty_C", MatPenalizedUpdatePenalty_PenaltyHIP));
  PetscCall(MatSetVecType(Arho, VECHIP));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ---------------------------------------------------------------------------------------------------
 * MATINV set-up on the device (MatInvSetUp_Inv / MatInvCreateInnerObjects_Inv, matinv.c:435-590) and the explicit inverse
 * (MatInvExplicitly_Inv matinv.c:670-730: one KSPSolve per column of the identity) restricted to what F can see.
 * --------------------------------------------------------------------------------------------------- */
static int PermonHipMgDestroy(void *h) { return pmh_mg_destroy((pmh_mg)h); }
static int PermonHipFexplicitDestroy(void *h) { return pmh_fexplicit_destroy((pmh_fexplicit)h); }

/* lambda_max(D^-1 A) by a few power iterations on the host objects (set-up; PETSc's own estimate lives inside KSPCHEBYSHEV without a getter) */
static PetscErrorCode PermonHipJacobiLambdaMax(Mat A, PetscReal *lmax)
{
  Vec       d, v, w;
  PetscReal nrm, lam = 1.0;
  PetscFunctionBegin;
  PetscCall(MatCreateVecs(A, &v, &w));
  PetscCall(VecDuplicate(v, &d));
  PetscCall(MatGetDiagonal(A, d));
  PetscCall(VecSet(v, 1.0));
  for (PetscInt it = 0; it < 20; it++) {
    PetscCall(VecNorm(v, NORM_2, &nrm));
    PetscCall(VecScale(v, 1.0 / nrm));
    PetscCall(MatMult(A, v, w));
    PetscCall(VecPointwiseDivide(w, w, d));
    PetscCall(VecDot(w, v, &lam));
    PetscCall(VecCopy(w, v));
  }
  *lmax = lam;
  PetscCall(VecDestroy(&d));
  PetscCall(VecDestroy(&v));
  PetscCall(VecDestroy(&w));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* The PC of MATINV's inner KSP (MatInvGetKSP; -mat_inv_pc_type).  After MatInvAttachHIP:
   PCJACOBI / PCNONE -> the library's block CG as created; PCMG (Galerkin levels, Chebyshev smoothing) -> the level operators and interpolations
   go to pmh_mg_create, the coarsest level is inverted densely on the host (K_reg of -regularize 1 is SPD on every level); MATSEQBAIJ bs = 3 -> k_bsr3 */
PERMON_EXTERN PetscErrorCode MatInvSetUp_HIP(Mat imat)
{
  Mat_Inv   *inv = (Mat_Inv *)imat->data;
  pmh_ctx    ctx;
  pmh_matinv Kp;
  KSP        ksp;
  PC         pc;
  PetscBool  ismg, isbaij;
  PetscInt   bs;
  PetscFunctionBegin;
  PetscCall(PermonHipGetCtx(&ctx));
  PetscCall(PermonHipQuery((PetscObject)imat, "pmh_matinv", (void **)&Kp));
  PetscCall(MatInvGetKSP(imat, &ksp));
  PetscCall(KSPGetPC(ksp, &pc));
  /* a block-Jacobi outer PC holds the rank's sequential solver (matinv.c:514-540): descend to it */
  {
    PetscBool isbj;
    PetscCall(PetscObjectTypeCompare((PetscObject)pc, PCBJACOBI, &isbj));
    if (isbj) {
      KSP     *sub;
      PetscInt nloc;
      PetscCall(PCBJacobiGetSubKSP(pc, &nloc, NULL, &sub));
      PetscCall(KSPGetPC(sub[0], &pc));
    }
  }
  PetscCall(PetscObjectTypeCompare((PetscObject)pc, PCMG, &ismg));
  if (ismg) {
    PetscInt   nlev, l, nc;
    pmh_csr   *A, *P;
    double    *lmax, *inv_host;
    pmh_mg     mg;
    Mat        Al, Pl, Ac, Acd, I, X;
    KSP        smooth;
    int        crs[2];
    MatFactorInfo info;
    IS         rperm, cperm;
    const PetscScalar *xa;
    PetscCall(PCMGGetLevels(pc, &nlev));
    PetscCall(PetscMalloc3(nlev, &A, nlev, &P, nlev, &lmax));
    /* PETSc numbers the levels coarse (0) to fine (nlev-1); the library fine (0) to coarse */
    for (l = 0; l < nlev; l++) {
      PetscCall(PCMGGetSmoother(pc, nlev - 1 - l, &smooth));
      PetscCall(KSPGetOperators(smooth, &Al, NULL));
      PetscCall(PermonHipCsrFromSeqAIJ(ctx, Al, &A[l]));
      PetscCall(PermonHipJacobiLambdaMax(Al, &lmax[l]));
      if (l + 1 < nlev) {
        PetscCall(PCMGGetInterpolation(pc, nlev - 1 - l, &Pl)); /* from level nlev-2-l up to nlev-1-l: n_l x n_{l+1} */
        PetscCall(PermonHipCsrFromSeqAIJ(ctx, Pl, &P[l]));
      }
    }
    /* dense inverse of the coarsest operator (one block per rank: matblockdiag.c:787-788) */
    PetscCall(PCMGGetSmoother(pc, 0, &smooth));
    PetscCall(KSPGetOperators(smooth, &Ac, NULL));
    PetscCall(MatGetSize(Ac, &nc, NULL));
    PetscCall(MatConvert(Ac, MATSEQDENSE, MAT_INITIAL_MATRIX, &Acd));
    PetscCall(MatCreateSeqDense(PETSC_COMM_SELF, nc, nc, NULL, &I));
    PetscCall(MatShift(I, 1.0));
    PetscCall(MatDuplicate(I, MAT_DO_NOT_COPY_VALUES, &X));
    PetscCall(MatFactorInfoInitialize(&info));
    PetscCall(MatGetOrdering(Acd, MATORDERINGNATURAL, &rperm, &cperm));
    PetscCall(MatLUFactor(Acd, rperm, cperm, &info));
    PetscCall(MatMatSolve(Acd, I, X));
    PetscCall(PetscMalloc1((size_t)nc * nc, &inv_host));
    PetscCall(MatDenseGetArrayRead(X, &xa));
    for (PetscInt i = 0; i < nc; i++)
      for (PetscInt j = 0; j < nc; j++) inv_host[(size_t)i * nc + j] = xa[(size_t)j * nc + i]; /* column-major -> row-major */
    PetscCall(MatDenseRestoreArrayRead(X, &xa));
    crs[0] = 0, crs[1] = (int)nc;
    PMHCall(pmh_mg_create(ctx, (int)nlev, A, P, 2, lmax, 0.1, 1.1, 1, crs, inv_host, PMH_MG_FP64, &mg)); /* PETSc's Chebyshev window [0.1, 1.1] lambda_max, degree 2 */
    PMHCall(pmh_matinv_set_pc_mg(Kp, mg));
    PetscCall(PermonHipCompose((PetscObject)imat, "pmh_mg", mg, PermonHipMgDestroy));
    for (l = 0; l < nlev; l++) { /* the hierarchy borrows the CSR handles: they live with imat */
      char key[32];
      PetscCall(PetscSNPrintf(key, sizeof(key), "pmh_mg_A%d", (int)l));
      PetscCall(PermonHipCompose((PetscObject)imat, key, A[l], PermonHipCsrDestroy));
      if (l + 1 < nlev) {
        PetscCall(PetscSNPrintf(key, sizeof(key), "pmh_mg_P%d", (int)l));
        PetscCall(PermonHipCompose((PetscObject)imat, key, P[l], PermonHipCsrDestroy));
      }
    }
    PetscCall(PetscFree(inv_host));
    PetscCall(ISDestroy(&rperm));
    PetscCall(ISDestroy(&cperm));
    PetscCall(MatDestroy(&X));
    PetscCall(MatDestroy(&I));
    PetscCall(MatDestroy(&Acd));
    PetscCall(PetscFree3(A, P, lmax));
  }
  /* K x of the inner CG on 3x3 blocks when the local block has that structure (elasticity; PETSc's MATSEQBAIJ bs = 3 role) */
  {
    Mat_BlockDiag *bd;
    Mat            Kin = inv->A;
    PetscCall(PetscObjectTypeCompare((PetscObject)Kin, MATBLOCKDIAG, &isbaij));
    if (isbaij) {
      bd = (Mat_BlockDiag *)Kin->data;
      PetscCall(MatGetBlockSize(bd->localBlock, &bs));
      if (bs == 3) {
        int rc = pmh_matinv_enable_bsr3(Kp);
        PetscCheck(!rc || rc == PMH_ERR_SUP, PETSC_COMM_SELF, PETSC_ERR_LIB, "libpermonhip error %d: %s", rc, pmh_last_error());
      }
    }
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* The explicit local dual operators behind F: W_b = (K_b^+)[Gamma_b, Gamma_b] on the dofs Bt touches, assembled by K^+ solves of imat's own device
   solver and attached to it, so that every F built on imat (pmh_op_create_feti_dual: case 3 of PermonHipOpFromMat) applies as
   Bhat blockdiag(W_b) Bhat' -- two CSR launches and ONE dense kernel instead of an inner Krylov solve.
   storage: PMH_FX_SYM (any decomposition), or -- after pmh_csr_block_classes found congruent blocks -- PMH_FX_CLASS_SYM / PMH_FX_CLASS_ORBIT
   (dims != NULL: the blocks are boxes of dims[0] x dims[1] x dims[2] nodes with ndof dofs per node; the symmetries of the box that leave K
   invariant serve the set-up and, for the orbit storage, the apply: k_fxo_gemm16 on the fp64 matrix cores).  PETSC_DECIDE picks by that rule.
   Call after MatGluingAttachHIP(Bt), MatBlockDiagAttachHIP(K) and MatInvAttachHIP(imat) [+ MatInvSetUp_HIP]. */
PERMON_EXTERN PetscErrorCode MatInvAttachExplicitHIP(Mat imat, Mat Bt, PetscInt storage, const PetscInt dims[3], PetscInt ndof, PetscReal rtol)
{
  Mat_Inv       *inv = (Mat_Inv *)imat->data;
  Mat_BlockDiag *bd  = (Mat_BlockDiag *)inv->A->data;
  pmh_gluing     B;
  pmh_blockdiag  K;
  pmh_matinv     Kp;
  pmh_fexplicit  E;
  const PetscInt    *ia, *ja;
  const PetscScalar *va;
  PetscInt           n;
  PetscBool          done;
  int                rowstart[2], cls[1] = {0}, ncls = 1, nsym = 1, idims[3];
  PetscFunctionBegin;
  PetscCall(PermonHipQuery((PetscObject)Bt, "pmh_gluing", (void **)&B));
  PetscCall(PermonHipQuery((PetscObject)inv->A, "pmh_blockdiag", (void **)&K));
  PetscCall(PermonHipQuery((PetscObject)imat, "pmh_matinv", (void **)&Kp));
  PetscCall(MatGetRowIJ(bd->localBlock, 0, PETSC_FALSE, PETSC_FALSE, &n, &ia, &ja, &done));
  PetscCheck(done, PETSC_COMM_SELF, PETSC_ERR_SUP, "the local block must be MATSEQAIJ");
  PetscCall(MatSeqAIJGetArrayRead(bd->localBlock, &va));
  rowstart[0] = 0, rowstart[1] = (int)n;
  /* one sequential block per rank (matblockdiag.c:787-788): its class is trivially 0; several blocks per rank (BASELINE configs[3]) come as one
     concatenated block with its row starts, and pmh_csr_block_classes groups the bit-identical ones */
  PMHCall(pmh_csr_block_classes(1, rowstart, (const int *)ia, (const int *)ja, va, cls, &ncls));
  if (storage == PETSC_DECIDE) storage = dims ? PMH_FX_CLASS_ORBIT : PMH_FX_SYM;
  if (storage == PMH_FX_CLASS_ORBIT) {
    PMHCall(pmh_fexplicit_create_shared_orbit(B, K, cls, &E));
  } else if (storage == PMH_FX_CLASS_SYM) {
    PMHCall(pmh_fexplicit_create_shared_sym(B, K, cls, &E));
  } else if (storage == PMH_FX_CLASS) {
    PMHCall(pmh_fexplicit_create_shared(B, K, cls, &E));
  } else {
    PMHCall(pmh_fexplicit_create(B, K, (int)storage, &E));
  }
  if (dims && (storage == PMH_FX_CLASS_ORBIT || storage == PMH_FX_CLASS_SYM)) {
    idims[0] = (int)dims[0], idims[1] = (int)dims[1], idims[2] = (int)dims[2];
    PMHCall(pmh_fexplicit_set_box_symmetry(E, 0, idims, (int)ndof, (const int *)ia, (const int *)ja, va, &nsym));
    if (storage == PMH_FX_CLASS_ORBIT && nsym < 16) {
      /* ONE block per rank (matblockdiag.c:787-788) means one class of one block: its own touched set -- three interface faces, a Dirichlet or contact face -- is invariant under
         2 ... 8 of the box's operations only.  On the closure of that set under the group (the whole boundary of a cube) every operation survives: one K^+ solve per orbit, the GEMM apply */
      int  nc = 0, nclo = 0, eptr[2] = {0, 0};
      int *urel, *clo;
      PMHCall(pmh_fexplicit_class_union(E, 0, &nc, NULL));
      PetscCall(PetscMalloc2(nc + 1, &urel, n + 1, &clo));
      PMHCall(pmh_fexplicit_class_union(E, 0, &nc, urel));
      PMHCall(pmh_box_symmetry_closure(idims, (int)ndof, (const int *)ia, (const int *)ja, va, nc, urel, &nclo, clo, NULL));
      PMHCall(pmh_fexplicit_destroy(E));
      eptr[1] = nclo;
      PMHCall(pmh_fexplicit_create_shared_orbit_union(B, K, cls, eptr, clo, &E));
      PetscCall(PetscFree2(urel, clo));
      PMHCall(pmh_fexplicit_set_box_symmetry(E, 0, idims, (int)ndof, (const int *)ia, (const int *)ja, va, &nsym));
    }
    if (storage == PMH_FX_CLASS_ORBIT && nsym < 16) { /* still too few operations for the GEMM form to pay (a box with three different sides): the streaming kernel on the symmetric tiles */
      PMHCall(pmh_fexplicit_destroy(E));
      PMHCall(pmh_fexplicit_create_shared_sym(B, K, cls, &E));
      PMHCall(pmh_fexplicit_set_box_symmetry(E, 0, idims, (int)ndof, (const int *)ia, (const int *)ja, va, &nsym));
    }
  }
  PetscCall(MatSeqAIJRestoreArrayRead(bd->localBlock, &va));
  PetscCall(MatRestoreRowIJ(bd->localBlock, 0, PETSC_FALSE, PETSC_FALSE, &n, &ia, &ja, &done));
  PMHCall(pmh_fexplicit_assemble_auto(E, Kp, cls, cls, rtol > 0 ? rtol : 1e-12, 0, NULL)); /* MatInvExplicitly's loop of KSPSolves (matinv.c:640-665): 8 unit right-hand sides per block and pass on the multi-right-hand-side K^+ where it applies, else one */
  PMHCall(pmh_matinv_attach_explicit(Kp, E));
  PetscCall(PermonHipCompose((PetscObject)imat, "pmh_fexplicit", E, PermonHipFexplicitDestroy));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* Everything QPTDualize creates, in one call placed at its end (qptransform.c:1166-1174, child = the dual QP):
   B / Bt (MATGLUING or MATEXTENSION), K (MATBLOCKDIAG), Kplus (MATINV), the projector factory of the child, and -- with explicit_dual --
   the explicit local dual operators.  F itself needs no handle: PermonHipOpFromMat recognises it by the objects composed on it. */
PERMON_EXTERN PetscErrorCode QPTDualizeAttachHIP(QP child, PetscBool moore_penrose, PetscBool explicit_dual, const PetscInt dims[3], PetscInt ndof)
{
  Mat       F, B = NULL, Bt = NULL, K = NULL, Kplus = NULL;
  PetscBool isgluing;
  QPPF      pf;
  PetscFunctionBegin;
  PetscCall(QPGetOperator(child, &F));
  PetscCall(PetscObjectQuery((PetscObject)F, "B", (PetscObject *)&B));
  PetscCall(PetscObjectQuery((PetscObject)F, "Bt", (PetscObject *)&Bt));
  PetscCall(PetscObjectQuery((PetscObject)F, "K", (PetscObject *)&K));
  PetscCall(PetscObjectQuery((PetscObject)F, "Kplus", (PetscObject *)&Kplus));
  PetscCheck(B && Bt && K && Kplus, PetscObjectComm((PetscObject)child), PETSC_ERR_ARG_WRONGSTATE, "not the child QP of QPTDualize");
  PetscCall(PetscObjectTypeCompare((PetscObject)Bt, MATGLUING, &isgluing));
  PetscCheck(isgluing, PetscObjectComm((PetscObject)child), PETSC_ERR_SUP, "the fused dual operator needs -feti_gluing_mattype gluing (MATEXTENSION keeps its slot-wise device mults: MatExtensionAttachHIP)");
  PetscCall(MatGluingAttachHIP(Bt));
  PetscCall(MatBlockDiagAttachHIP(K));
  {
    /* which generalised inverse QPTDualize built (qptransform.c:997-1062) is read off the object it left behind; `moore_penrose` only matters for an unwrapped MATINV
       whose kernel the caller wants projected out (-qpt_dualize_Kplus_mp given to the library instead of to QPTDualize) */
    Mat                imat;
    PermonHipKplusForm form;
    PetscCall(PermonHipUnwrapKplus(Kplus, &imat, &form));
    if (form == PERMONHIP_KPLUS_PLAIN && moore_penrose) form = PERMONHIP_KPLUS_MP;
    PetscCheck(!(explicit_dual && form == PERMONHIP_KPLUS_LEFT), PetscObjectComm((PetscObject)child), PETSC_ERR_SUP, "the explicit local dual operators store symmetric blocks: K_reg^{-1} or -qpt_dualize_Kplus_mp, not the left generalised inverse");
    PetscCall(MatInvAttachHIPForm(imat, form));
    PetscCall(MatInvSetUp_HIP(imat));
    if (explicit_dual) PetscCall(MatInvAttachExplicitHIP(imat, Bt, PETSC_DECIDE, dims, ndof, 1e-12));
  }
  PetscCall(QPGetQPPF(child, &pf));
  if (pf) PetscCall(QPPFAttachHIP(pf));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ---------------------------------------------------------------------------------------------------
 * QPS type "smalxehip": QPSSetUp_SMALXE / QPSSolve_SMALXE (smalxe.c:772-997) as ONE device-resident loop (pmh_smalxe_*): the outer updates of
 * Bt mu, M1, rho, the inner MPGP with QPSConverged_Inner_SMALXE and its ||B u|| all run inside the library, on the operator tower of the QP.
 * The object keeps the reference's QPS_SMALXE (parameters, options, composed Get/Set methods, view): QPSCreate_SMALXE builds it, the two
 * numerical slots are replaced.  This is the path bench.py's headline measures:
 *   -qps_type smalxehip  ->  QPSSetUp_SMALXEHIP: PermonHipOpFromMat(qp->A = P F P) -> pmh_op_create_projected(pmh_op_create_feti_dual(B, K^+))
 *                            pmh_smalxe_create -> pmh_op_create_penalized(...), inner pmh_mpgp
 *                        ->  QPSSolve_SMALXEHIP: pmh_smalxe_solve -> per inner iteration A_rho p: ... pmh_fexplicit_mult -> k_fxo_gemm4 + k_fxo_fin
 * --------------------------------------------------------------------------------------------------- */
#include <../src/qps/impls/smalxe/smalxeimpl.h>
PERMON_EXTERN PetscErrorCode QPSCreate_SMALXE(QPS qps);

typedef struct {
  pmh_smalxe      solver;
  pmh_smalxe_opts opts;
  pmh_op          A;
  pmh_op          owned[PMH_MAX_TOWER];
  PetscInt        nowned;
} QPS_SMALXEHIP;

static int PermonHipSmalxeDestroy(void *h)
{
  QPS_SMALXEHIP *hip = (QPS_SMALXEHIP *)h;
  if (hip->solver) (void)pmh_smalxe_destroy(hip->solver);
  for (PetscInt i = 0; i < hip->nowned; i++) (void)pmh_op_destroy(hip->owned[i]);
  (void)PetscFree(hip);
  return 0;
}

static PetscErrorCode QPSSetUp_SMALXEHIP(QPS qps)
{
  QPS_SMALXE        *smalxe = (QPS_SMALXE *)qps->data;
  QPS_SMALXEHIP     *hip;
  QP                 qp;
  Mat                A, BE;
  Vec                cE, b, u, lb, ub;
  pmh_ctx            ctx;
  pmh_qppf           pf;
  const PetscScalar *b_d, *lb_d = NULL, *ub_d = NULL;
  PetscScalar       *u_d;
  const QPSType      innertype;
  pmh_mpgp           inner_solver;

  PetscFunctionBegin;
  PetscCall(PermonHipGetCtx(&ctx));
  qp = qps->solQP;
  if (qp->cE) { /* smalxe.c:782-787 */
    PetscCall(QPTHomogenizeEq(qp));
    PetscCall(QPChainGetLast(qp, &qps->solQP));
    qp = qps->solQP;
  }
  PetscCall(PetscNew(&hip));
  PMHCall(pmh_smalxe_default_opts(&hip->opts));
  /* the reference's parameters (set by its own options / setters on QPS_SMALXE) -> the library's struct */
  hip->opts.rtol = qps->rtol, hip->opts.atol = qps->atol, hip->opts.divtol = qps->divtol, hip->opts.max_it = (int)qps->max_it;
  hip->opts.M1_user = smalxe->M1_user, hip->opts.M1_direct = (smalxe->M1_type == QPS_ARG_DIRECT), hip->opts.M1_update = smalxe->M1_update;
  hip->opts.rtol_E = smalxe->rtol_E;
  hip->opts.rho_user = smalxe->rho_user, hip->opts.rho_direct = (smalxe->rho_type == QPS_ARG_DIRECT);
  hip->opts.rho_update = smalxe->rho_update, hip->opts.rho_update_late = smalxe->rho_update_late;
  hip->opts.eta_user = smalxe->eta_user, hip->opts.eta_direct = (smalxe->eta_type == QPS_ARG_DIRECT);
  hip->opts.update_threshold = smalxe->update_threshold;
  hip->opts.maxeig = smalxe->maxeig, hip->opts.maxeig_tol = smalxe->maxeig_tol, hip->opts.maxeig_iter = (int)smalxe->maxeig_iter;
  hip->opts.inject_maxeig = smalxe->inject_maxeig, hip->opts.inject_maxeig_set = smalxe->inject_maxeig_set;
  hip->opts.inner_iter_min = (int)smalxe->inner_iter_min, hip->opts.inner_no_gtol_stop = (int)smalxe->inner_no_gtol_stop;
  hip->opts.lag_enabled = smalxe->lag_enabled, hip->opts.lag_offset = (int)smalxe->norm_update_lag_offset;
  hip->opts.lag_start = (int)smalxe->Jstart, hip->opts.lag_step = (int)smalxe->Jstep, hip->opts.lag_end = (int)smalxe->Jend;
  hip->opts.lag_lower = smalxe->lower, hip->opts.lag_upper = smalxe->upper;
  hip->opts.knoll = smalxe->knoll;
  /* the inner solver object (prefix smalxe_, smalxe.c:492-507) carries the inner MPGP's parameters; its type is the device MPGP */
  PetscCall(QPSSMALXEGetInnerQPS(qps, &smalxe->inner));
  PetscCall(QPSGetType(smalxe->inner, &innertype));
  if (!innertype) PetscCall(QPSSetType(smalxe->inner, "mpgphip"));
  if (smalxe->setfromoptionscalled) PetscCall(QPSSetFromOptions(smalxe->inner));
  {
    PetscBool iship;
    PetscCall(PetscObjectTypeCompare((PetscObject)smalxe->inner, "mpgphip", &iship));
    PetscCheck(iship, PetscObjectComm((PetscObject)qps), PETSC_ERR_SUP, "smalxehip drives the device MPGP as its inner solver (-smalxe_qps_type mpgphip)");
    hip->opts.inner        = ((QPS_MPGPHIP *)smalxe->inner->data)->opts;
    hip->opts.inner.rtol   = smalxe->inner->rtol, hip->opts.inner.atol = smalxe->inner->atol;
    hip->opts.inner.max_it = (int)smalxe->inner->max_it;
  }
  /* operator, equality constraints (BE = G, cE homogenised away), box */
  PetscCall(QPGetOperator(qp, &A));
  PetscCall(QPGetEq(qp, &BE, &cE));
  hip->opts.be_implicit = BE->ops->mult ? 0 : 1; /* smalxe.c:878-886: only B'B available -> the SMALXEON updates of ||Bu|| */
  PetscCall(PermonHipQPPFHandle(qp->pf, &pf));
  PetscCall(PermonHipOpFromMat(A, &hip->A, &hip->nowned, hip->owned));
  PetscCall(QPGetRhs(qp, &b));
  PetscCall(QPGetSolutionVector(qp, &u));
  PetscCall(QPGetBox(qp, NULL, &lb, &ub));
  PetscCall(VecHIPGetArrayRead(b, &b_d));
  PetscCall(VecHIPGetArray(u, &u_d));
  if (lb) PetscCall(VecHIPGetArrayRead(lb, &lb_d));
  if (ub) PetscCall(VecHIPGetArrayRead(ub, &ub_d));
  /* eta, maxeig (power method on A), M1, rho, the penalised operator and the inner MPGP with the injected test: smalxe.c:806-875 */
  PMHCall(pmh_smalxe_create(ctx, hip->A, b_d, u_d, lb_d, ub_d, pf, &hip->opts, &hip->solver));
  {
    /* -qps_smalxehip_reuse_products: an extension of this back end, off by default (the reference forms A_rho u by a MatMult of its own in QPComputeObjective,
       smalxe.c:982, and at the start of every inner solve, mpgp.c:500; the library can carry it from the inner solve's last gradient) */
    PetscBool reuse = PETSC_FALSE;
    PetscCall(PetscOptionsGetBool(((PetscObject)qps)->options, ((PetscObject)qps)->prefix, "-qps_smalxehip_reuse_products", &reuse, NULL));
    PMHCall(pmh_smalxe_set_reuse_products(hip->solver, reuse ? 1 : 0));
  }
  if (ub) PetscCall(VecHIPRestoreArrayRead(ub, &ub_d));
  if (lb) PetscCall(VecHIPRestoreArrayRead(lb, &lb_d));
  PetscCall(VecHIPRestoreArray(u, &u_d));
  PetscCall(VecHIPRestoreArrayRead(b, &b_d));
  /* the inner QPS object views / reports the library's inner solver */
  PMHCall(pmh_smalxe_get_inner(hip->solver, &inner_solver));
  ((QPS_MPGPHIP *)smalxe->inner->data)->solver   = inner_solver;
  ((QPS_MPGPHIP *)smalxe->inner->data)->borrowed = PETSC_TRUE;
  smalxe->inner->setupcalled                     = PETSC_TRUE;
  PetscCall(PermonHipCompose((PetscObject)qps, "pmh_smalxe", hip, PermonHipSmalxeDestroy));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode QPSSolve_SMALXEHIP(QPS qps)
{
  QPS_SMALXE      *smalxe = (QPS_SMALXE *)qps->data;
  QPS_SMALXEHIP   *hip;
  QP               qp = qps->solQP;
  Vec              u;
  PetscScalar     *u_d;
  pmh_smalxe_stats st;

  PetscFunctionBegin;
  PetscCall(PermonHipQuery((PetscObject)qps, "pmh_smalxe", (void **)&hip));
  PetscCall(QPGetSolutionVector(qp, &u));
  PetscCall(VecHIPGetArray(u, &u_d)); /* held for the solve; the Restore bumps u's state */
  PMHCall(pmh_smalxe_solve(hip->solver));
  PMHCall(pmh_smalxe_get_stats(hip->solver, &st));
  PetscCall(VecHIPRestoreArray(u, &u_d));
  qps->iteration          = st.iteration;
  qps->rnorm              = st.rnorm;
  qps->reason             = (KSPConvergedReason)st.reason;
  smalxe->inner_iter_accu = st.inner_iter_accu;
  smalxe->state           = st.state;
  smalxe->M1              = st.M1;
  smalxe->M1_hits = st.M1_hits, smalxe->eta_hits = st.eta_hits, smalxe->M1_updates = st.M1_updates, smalxe->rho_updates = st.rho_updates;
  smalxe->normBu = st.normBu, smalxe->enorm = st.enorm, smalxe->maxeig = st.maxeig;
  smalxe->inner->iteration = st.inner.iteration, smalxe->inner->rnorm = st.inner.rnorm, smalxe->inner->reason = (KSPConvergedReason)st.inner.reason;
  /* the multipliers of the equality constraints stay in the library as Bt mu (smalxe.c:993-994: get_lambda / get_Bt_lambda are post-processing) */
  PetscFunctionReturn(PETSC_SUCCESS);
}

PERMON_EXTERN PetscErrorCode QPSCreate_SMALXEHIP(QPS qps)
{
  PetscFunctionBegin;
  PetscCall(QPSCreate_SMALXE(qps)); /* QPS_SMALXE, defaults (smalxe.c:1159-1207), options, view, reset, destroy, the composed Get/Set methods */
  qps->ops->setup = QPSSetUp_SMALXEHIP;
  qps->ops->solve = QPSSolve_SMALXEHIP;
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ---------------------------------------------------------------------------------------------------
 * QPS types "pcpghip" (QPSSolve_PCPG pcpg.c:51-134) and "ksphip" (QPSSolve_KSP qpsksp.c:127-143 with the KSP of QPSCreate_KSP :244-250:
 * CG, unpreconditioned norm, PCNONE unless the QP carries a PC).  The preconditioner: PCDUAL lumped -> pmh_pc_dual_lumped_apply behind a shell.
 * --------------------------------------------------------------------------------------------------- */
typedef struct {
  pmh_op   A, pc;
  pmh_op   owned[PMH_MAX_TOWER];
  PetscInt nowned;
  pmh_gluing    pcB;
  pmh_blockdiag pcK;
} QPS_CGHIP;

static int PermonHipLumpedMult(void *user, const double *x_dev, double *y_dev) /* pmh_shell_mult_fn: PCApply_Dual pcdual.c:63-78 */
{
  QPS_CGHIP *hip = (QPS_CGHIP *)user;
  return pmh_pc_dual_lumped_apply(hip->pcB, hip->pcK, x_dev, y_dev);
}

static PetscErrorCode QPSSetup_CGHIP(QPS qps)
{
  QPS_CGHIP *hip = (QPS_CGHIP *)qps->data;
  QP         qp  = qps->solQP;
  pmh_ctx    ctx;
  Mat        A;
  PC         pc;
  PetscBool  isdual;
  PetscFunctionBegin;
  PetscCall(PermonHipGetCtx(&ctx));
  if (qp->cE) { /* pcpg.c:37-41 */
    PetscCall(QPTHomogenizeEq(qp));
    PetscCall(QPChainGetLast(qp, &qps->solQP));
    qp = qps->solQP;
  }
  for (PetscInt i = 0; i < hip->nowned; i++) (void)pmh_op_destroy(hip->owned[i]);
  hip->nowned = 0, hip->pc = NULL;
  PetscCall(QPGetOperator(qp, &A));
  PetscCall(PermonHipOpFromMat(A, &hip->A, &hip->nowned, hip->owned));
  PetscCall(QPGetPC(qp, &pc));
  PetscCall(PetscObjectTypeCompare((PetscObject)pc, PCDUAL, &isdual));
  if (isdual) {
    PCDualType type;
    PetscCall(PCDualGetType(pc, &type));
    if (type == PC_DUAL_LUMPED) {
      PC_Dual *dual = (PC_Dual *)pc->data;
      PetscInt n;
      PetscCall(PCSetUp(pc)); /* PCSetUp_Dual queries "Bt" and "K" on F (pcdual.c:107-108) */
      PetscCall(PermonHipQuery((PetscObject)dual->At, "pmh_gluing", (void **)&hip->pcB));
      PetscCall(PermonHipQuery((PetscObject)dual->C_bb, "pmh_blockdiag", (void **)&hip->pcK));
      PetscCall(MatGetLocalSize(A, &n, NULL));
      PMHCall(pmh_op_create_shell(ctx, (int)n, PermonHipLumpedMult, hip, &hip->pc));
      PetscCall(PermonHipOpOwn(hip->pc, &hip->nowned, hip->owned));
    }
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode QPSSolve_CGHIP_Private(QPS qps, PetscBool projected)
{
  QPS_CGHIP         *hip = (QPS_CGHIP *)qps->data;
  QP                 qp  = qps->solQP;
  pmh_ctx            ctx;
  pmh_qppf           pf = NULL;
  Vec                b, x;
  const PetscScalar *b_d;
  PetscScalar       *x_d;
  pmh_pcpg_stats     st;
  PetscFunctionBegin;
  PetscCall(PermonHipGetCtx(&ctx));
  if (projected) PetscCall(PermonHipQPPFHandle(qp->pf, &pf));
  PetscCall(QPGetRhs(qp, &b));
  PetscCall(QPGetSolutionVector(qp, &x));
  PetscCall(VecHIPGetArrayRead(b, &b_d));
  PetscCall(VecHIPGetArray(x, &x_d));
  if (projected) PMHCall(pmh_pcpg_solve(ctx, hip->A, b_d, x_d, pf, hip->pc, qps->rtol, qps->atol, qps->divtol, (int)qps->max_it, &st));
  else PMHCall(pmh_ksp_cg_solve(ctx, hip->A, b_d, x_d, hip->pc, qps->rtol, qps->atol, qps->divtol, (int)qps->max_it, &st));
  PetscCall(VecHIPRestoreArray(x, &x_d));
  PetscCall(VecHIPRestoreArrayRead(b, &b_d));
  qps->iteration = st.iteration, qps->rnorm = st.rnorm, qps->reason = (KSPConvergedReason)st.reason;
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode QPSSolve_PCPGHIP(QPS qps)
{
  PetscFunctionBegin;
  PetscCall(QPSSolve_CGHIP_Private(qps, PETSC_TRUE));
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode QPSSolve_KSPHIP(QPS qps)
{
  PetscFunctionBegin;
  PetscCall(QPSSolve_CGHIP_Private(qps, PETSC_FALSE));
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode QPSIsQPCompatible_PCPGHIP(QPS qps, QP qp, PetscBool *flg) /* pcpg.c:13-21 */
{
  PetscFunctionBegin;
  *flg = (qp->qpc || qp->BI || !qp->BE) ? PETSC_FALSE : PETSC_TRUE;
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode QPSIsQPCompatible_KSPHIP(QPS qps, QP qp, PetscBool *flg) /* qpsksp.c: no constraint of any kind */
{
  PetscFunctionBegin;
  *flg = (qp->qpc || qp->BI || qp->BE) ? PETSC_FALSE : PETSC_TRUE;
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode QPSDestroy_CGHIP(QPS qps)
{
  QPS_CGHIP *hip = (QPS_CGHIP *)qps->data;
  PetscFunctionBegin;
  for (PetscInt i = 0; i < hip->nowned; i++) (void)pmh_op_destroy(hip->owned[i]);
  PetscCall(QPSDestroyDefault(qps));
  PetscFunctionReturn(PETSC_SUCCESS);
}
PERMON_EXTERN PetscErrorCode QPSCreate_PCPGHIP(QPS qps)
{
  QPS_CGHIP *hip;
  PetscFunctionBegin;
  PetscCall(PetscNew(&hip));
  qps->data                = (void *)hip;
  qps->ops->setup          = QPSSetup_CGHIP;
  qps->ops->solve          = QPSSolve_PCPGHIP;
  qps->ops->destroy        = QPSDestroy_CGHIP;
  qps->ops->isqpcompatible = QPSIsQPCompatible_PCPGHIP;
  PetscFunctionReturn(PETSC_SUCCESS);
}
PERMON_EXTERN PetscErrorCode QPSCreate_KSPHIP(QPS qps)
{
  QPS_CGHIP *hip;
  PetscFunctionBegin;
  PetscCall(PetscNew(&hip));
  qps->data                = (void *)hip;
  qps->ops->setup          = QPSSetup_CGHIP;
  qps->ops->solve          = QPSSolve_KSPHIP;
  qps->ops->destroy        = QPSDestroy_CGHIP;
  qps->ops->isqpcompatible = QPSIsQPCompatible_KSPHIP;
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ---------------------------------------------------------------------------------------------------
 * KSPFETI (src/ksp/impls/feti/feti.c): KSPSolve_FETI (:144-156) = KSPFETISetUp (:71-94: QPTMatISToBlockDiag, QPFetiSetUp, QPTFromOptions) +
 * QPSSolve.  Two bindings:
 *  (a) keep the reference's chain and let it run on the device: QPTDualizeAttachHIP at the end of QPTDualize, QPSSetDefaultType picking
 *      "ksphip" / "pcpghip" / "smalxehip" -- nothing KSPFETI-specific is needed;
 *  (b) the whole solve in one library call for a single-process run (every subdomain on this GPU): KSPSolve_FETIHIP below hands the MATIS
 *      pieces to pmh_kspfeti_solve (gluing by the QPFetiGetBgtSF rules, MatRegularize, K^+, the dual chain, CG on P F, primal recovery).
 * --------------------------------------------------------------------------------------------------- */
PERMON_EXTERN PetscErrorCode KSPSolve_FETIHIP(KSP ksp, IS isDir /* local numbering, enforced by B; or NULL */, Mat R /* dense, N x kdim, or NULL */)
{
  pmh_ctx                ctx;
  Mat                    Amat, Aloc;
  ISLocalToGlobalMapping l2gmap;
  const PetscInt        *l2g, *ia, *ja, *dir = NULL;
  const PetscScalar     *va, *bg, *r = NULL;
  PetscScalar           *f, *u, *xg;
  PetscInt               n, N, ndir = 0, kdim = 0;
  PetscBool              done;
  PetscMPIInt            size;
  Vec                    b, x;
  pmh_kspfeti_opts       o;
  pmh_kspfeti_stats      st;
  char                  *optstr = NULL, left[512];
  int                    rowstart[2];

  PetscFunctionBegin;
  PetscCall(PermonHipGetCtx(&ctx));
  PetscCallMPI(MPI_Comm_size(PetscObjectComm((PetscObject)ksp), &size));
  PetscCheck(size == 1, PetscObjectComm((PetscObject)ksp), PETSC_ERR_SUP, "KSPSolve_FETIHIP binds the single-process solve; with one rank per GPU use binding (a): QPTDualizeAttachHIP + -qps_type ksphip");
  PetscCall(KSPGetOperators(ksp, &Amat, NULL));
  PetscCall(KSPGetRhs(ksp, &b));
  PetscCall(KSPGetSolution(ksp, &x));
  PetscCall(MatISGetLocalMat(Amat, &Aloc));
  PetscCall(MatISGetLocalToGlobalMapping(Amat, &l2gmap, NULL));
  PetscCall(ISLocalToGlobalMappingGetIndices(l2gmap, &l2g));
  PetscCall(MatGetRowIJ(Aloc, 0, PETSC_FALSE, PETSC_FALSE, &n, &ia, &ja, &done));
  PetscCheck(done, PETSC_COMM_SELF, PETSC_ERR_SUP, "the local matrix of the MATIS must be MATSEQAIJ");
  PetscCall(MatSeqAIJGetArrayRead(Aloc, &va));
  PetscCall(MatGetSize(Amat, &N, NULL));
  PetscCall(PetscMalloc2(n, &f, n, &u));
  PetscCall(VecGetArrayRead(b, &bg));
  PMHCall(pmh_qpt_matis_split_rhs((int)n, (const int *)l2g, (int)N, bg, f)); /* QPTMatISToBlockDiag's vector part, qptransform.c:2095-2113 */
  PetscCall(VecRestoreArrayRead(b, &bg));
  if (isDir) {
    PetscCall(ISGetLocalSize(isDir, &ndir));
    PetscCall(ISGetIndices(isDir, &dir));
  }
  if (R) {
    PetscCall(MatGetSize(R, NULL, &kdim));
    PetscCall(MatDenseGetArrayRead(R, &r)); /* column-major N x kdim = kdim rows of length N, the layout pmh_kspfeti_solve takes */
  }
  PMHCall(pmh_kspfeti_default_opts(&o));
  PetscCall(PetscOptionsGetAll(NULL, &optstr)); /* -feti_gluing_type, -regularize, -dual_pc_dual_type, -qps_rtol ... (the keys of QPTFromOptions / QPSSetFromOptions) */
  PMHCall(pmh_kspfeti_set_from_options(optstr, &o, left, (int)sizeof(left)));
  PetscCall(PetscFree(optstr));
  rowstart[0] = 0, rowstart[1] = (int)n;
  PMHCall(pmh_kspfeti_solve(ctx, 1, rowstart, (const int *)ia, (const int *)ja, va, f, (const int *)l2g, (int)ndir, (const int *)dir, (int)kdim, r, &o, u, NULL, 0, &st));
  PetscCall(VecGetArray(x, &xg));
  PMHCall(pmh_qpt_matis_assemble_solution((int)n, (const int *)l2g, u, (int)N, xg)); /* the post-solve of QPTMatISToBlockDiag, qptransform.c:1945-1949 */
  PetscCall(VecRestoreArray(x, &xg));
  ksp->reason = (KSPConvergedReason)st.reason;
  ksp->its    = st.iteration;
  if (R) PetscCall(MatDenseRestoreArrayRead(R, &r));
  if (isDir) PetscCall(ISRestoreIndices(isDir, &dir));
  PetscCall(PetscFree2(f, u));
  PetscCall(MatSeqAIJRestoreArrayRead(Aloc, &va));
  PetscCall(MatRestoreRowIJ(Aloc, 0, PETSC_FALSE, PETSC_FALSE, &n, &ia, &ja, &done));
  PetscCall(ISLocalToGlobalMappingRestoreIndices(l2gmap, &l2g));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* registration: called from PermonInitialize next to QPSRegisterAll (src/sys/permoninit.c:86-88) */
PERMON_EXTERN PetscErrorCode PermonHipRegisterAll(void)
{
  PetscFunctionBegin;
  PetscCall(QPSRegister("mpgphip", QPSCreate_MPGPHIP));     /* -qps_type mpgphip, or register as QPSMPGP to override */
  PetscCall(QPSRegister("smalxehip", QPSCreate_SMALXEHIP)); /* -qps_type smalxehip: the device-resident SMALXE + MPGP loop on the operator tower (the path bench.py measures) */
  PetscCall(QPSRegister("pcpghip", QPSCreate_PCPGHIP));     /* -qps_type pcpghip: projected preconditioned CG */
  PetscCall(QPSRegister("ksphip", QPSCreate_KSPHIP));       /* -qps_type ksphip: the CG of QPSKSP on the projected dual of a linear problem */
  PetscCall(QPCRegister("boxhip", QPCCreate_BoxHIP));       /* QPCSetType(qpc, "boxhip"), or register as QPCBOX to override (qpcreg.c:23-28) */
  /* the Mat / PC / QPPF slots are attached per object after its set-up: MatBlockDiagAttachHIP, MatGluingAttachHIP, MatExtensionAttachHIP,
     MatInvAttachHIP, MatInvSetUp_HIP, MatInvAttachExplicitHIP, QPPFAttachHIP, MatPenalizedAttachHIP, PCDualAttachHIP -- QPTDualizeAttachHIP does the first
     group in one call at the end of QPTDualize, MatPenalizedAttachHIP goes after MatCreatePenalized in QPTEnforceEqByPenalty, PCDualAttachHIP into PCSetUp_Dual */
  PetscFunctionReturn(PETSC_SUCCESS);
}
