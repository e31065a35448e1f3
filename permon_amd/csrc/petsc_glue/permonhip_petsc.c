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
 *   Mat mult slots (matblockdiag.c:742-746, gluing.c:280-284)  -> MatMult_*HIP below
 */
#include <permon/private/qpsimpl.h>
#include <permon/private/qpcimpl.h>
#include <permonmat.h>
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

/* ---------------------------------------------------------------------------------------------------
 * QPS type "mpgphip": replaces QPSSolve_MPGP / QPSSetup_MPGP (src/qps/impls/mpgp/mpgp.c:359-650)
 * --------------------------------------------------------------------------------------------------- */
typedef struct {
  pmh_csr       A;
  pmh_op        op;
  pmh_mpgp      solver;
  pmh_mpgp_opts opts;
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

static PetscErrorCode QPSSetup_MPGPHIP(QPS qps)
{
  QPS_MPGPHIP       *hip = (QPS_MPGPHIP *)qps->data;
  pmh_ctx            ctx;
  Mat                A;
  Vec                b, x, lb, ub;
  const PetscInt    *ia, *ja;
  const PetscScalar *va, *b_d, *lb_d = NULL, *ub_d = NULL;
  PetscScalar       *x_d;
  PetscInt           n;
  PetscBool          done;

  PetscFunctionBegin;
  PetscCall(PermonHipGetCtx(&ctx));
  PetscCall(QPGetOperator(qps->solQP, &A));
  PetscCall(QPGetRhs(qps->solQP, &b));
  PetscCall(QPGetSolutionVector(qps->solQP, &x));
  PetscCall(QPGetBox(qps->solQP, NULL, &lb, &ub));
  /* host CSR of the local SeqAIJ block -> device (once) */
  PetscCall(MatGetRowIJ(A, 0, PETSC_FALSE, PETSC_FALSE, &n, &ia, &ja, &done));
  PetscCall(MatSeqAIJGetArrayRead(A, &va));
  PMHCall(pmh_csr_create(ctx, (int)n, (int)n, (const int *)ia, (const int *)ja, va, &hip->A));
  PetscCall(MatSeqAIJRestoreArrayRead(A, &va));
  PetscCall(MatRestoreRowIJ(A, 0, PETSC_FALSE, PETSC_FALSE, &n, &ia, &ja, &done));
  PMHCall(pmh_op_create_csr(hip->A, &hip->op));
  /* device pointers of the PETSc Vecs (VECHIP) */
  PetscCall(VecHIPGetArrayRead(b, &b_d));
  PetscCall(VecHIPGetArray(x, &x_d));
  if (lb) PetscCall(VecHIPGetArrayRead(lb, &lb_d));
  if (ub) PetscCall(VecHIPGetArrayRead(ub, &ub_d));
  hip->opts.rtol = qps->rtol, hip->opts.atol = qps->atol, hip->opts.divtol = qps->divtol, hip->opts.max_it = (int)qps->max_it;
  PMHCall(pmh_mpgp_create(ctx, hip->op, b_d, x_d, lb_d, ub_d, &hip->opts, &hip->solver));
  PMHCall(pmh_mpgp_set_convergence_test(hip->solver, QPSMPGPHIPConverged, qps));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode QPSSolve_MPGPHIP(QPS qps)
{
  QPS_MPGPHIP   *hip = (QPS_MPGPHIP *)qps->data;
  pmh_mpgp_stats st;
  Vec            x;

  PetscFunctionBegin;
  PMHCall(pmh_mpgp_solve(hip->solver));
  PMHCall(pmh_mpgp_get_stats(hip->solver, &st));
  qps->iteration = st.iteration;
  qps->rnorm     = st.rnorm;
  qps->reason    = (KSPConvergedReason)st.reason;
  PetscCall(QPGetSolutionVector(qps->solQP, &x));
  PetscCall(PetscObjectStateIncrease((PetscObject)x)); /* x was written outside PETSc accessors (state stamps) */
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

static PetscErrorCode QPSDestroy_MPGPHIP(QPS qps)
{
  QPS_MPGPHIP *hip = (QPS_MPGPHIP *)qps->data;
  PetscFunctionBegin;
  pmh_mpgp_destroy(hip->solver);
  pmh_op_destroy(hip->op);
  pmh_csr_destroy(hip->A);
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
  qps->ops->setup   = QPSSetup_MPGPHIP;
  qps->ops->solve   = QPSSolve_MPGPHIP;
  qps->ops->destroy = QPSDestroy_MPGPHIP;
  PetscCall(PetscObjectComposeFunction((PetscObject)qps, "QPSMPGPSetOperatorMaxEigenvalue_MPGP_C", QPSMPGPSetOperatorMaxEigenvalue_MPGPHIP));
  PetscCall(PetscObjectComposeFunction((PetscObject)qps, "QPSMPGPUpdateMaxEigenvalue_MPGP_C", QPSMPGPUpdateMaxEigenvalue_MPGPHIP));
  PetscCall(PetscObjectComposeFunction((PetscObject)qps, "QPSMPGPGetCurrentStepType_MPGP_C", QPSMPGPGetCurrentStepType_MPGPHIP));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ---------------------------------------------------------------------------------------------------
 * QPC box ops on the device: the four _QPCOps slots called by the wrappers of src/qpc/interface/qpc.c
 * --------------------------------------------------------------------------------------------------- */
#include <../src/qpc/impls/box/qpcboximpl.h>
#define BOXPTRS \
  QPC_Box           *ctx = (QPC_Box *)qpc->data; \
  const PetscScalar *lb_d = NULL, *ub_d = NULL; \
  pmh_ctx            h; \
  PetscInt           n; \
  PetscCall(PermonHipGetCtx(&h)); \
  if (ctx->lb) PetscCall(VecHIPGetArrayRead(ctx->lb, &lb_d)); \
  if (ctx->ub) PetscCall(VecHIPGetArrayRead(ctx->ub, &ub_d));

static PetscErrorCode QPCGrads_BoxHIP(QPC qpc, Vec x, Vec g, Vec gf, Vec gc)
{
  const PetscScalar *x_d, *g_d;
  PetscScalar       *gf_d, *gc_d;
  PetscFunctionBegin;
  BOXPTRS;
  PetscCall(VecGetLocalSize(x, &n));
  PetscCall(VecHIPGetArrayRead(x, &x_d));
  PetscCall(VecHIPGetArrayRead(g, &g_d));
  PetscCall(VecHIPGetArrayWrite(gf, &gf_d));
  PetscCall(VecHIPGetArrayWrite(gc, &gc_d));
  PMHCall(pmh_qpc_box_grads(h, (int)n, x_d, g_d, lb_d, ub_d, qpc->astol, gf_d, gc_d)); /* includes gf=g, gc=0 of qpc.c:551-552 */
  PetscCall(VecHIPRestoreArrayWrite(gc, &gc_d));
  PetscCall(VecHIPRestoreArrayWrite(gf, &gf_d));
  PetscCall(VecHIPRestoreArrayRead(g, &g_d));
  PetscCall(VecHIPRestoreArrayRead(x, &x_d));
  PetscFunctionReturn(PETSC_SUCCESS);
}
/* QPCGradReduced_BoxHIP, QPCFeas_BoxHIP, QPCProject_BoxHIP: same pattern over pmh_qpc_box_gradreduced /
   pmh_qpc_box_feas (local min; the wrapper qpc.c:521 adds MPI_Allreduce(MIN)) / pmh_qpc_box_project. */

/* ---------------------------------------------------------------------------------------------------
 * Mat mult slots: MATBLOCKDIAG local block and MATGLUING on the device
 * --------------------------------------------------------------------------------------------------- */
static PetscErrorCode MatMult_GluingHIP(Mat mat, Vec right, Vec left) /* replaces gluing.c:47-81 */
{
  pmh_gluing         B;
  const PetscScalar *l_d;
  PetscScalar       *x_d;
  PetscFunctionBegin;
  PetscCall(PetscObjectQuery((PetscObject)mat, "pmh_gluing", (PetscObject *)&B)); /* container set at MatCreateGluing time */
  PetscCall(VecHIPGetArrayRead(right, &l_d));
  PetscCall(VecHIPGetArrayWrite(left, &x_d));
  PMHCall(pmh_gluing_mult(B, l_d, x_d));
  PetscCall(VecHIPRestoreArrayWrite(left, &x_d));
  PetscCall(VecHIPRestoreArrayRead(right, &l_d));
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
  /* PetscInt must be 32 bit (the library's index type), as for pmh_csr_create */
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
  PetscFunctionReturn(PETSC_SUCCESS);
}
