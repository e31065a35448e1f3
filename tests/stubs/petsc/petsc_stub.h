/* PETSc STAND-IN for ONE purpose: `gcc -fsyntax-only` of permon_amd/csrc/petsc_glue/permonhip_petsc.c against PERMON's OWN headers (/root/reference/include) on an image that
 * has no PETSc (tests/test_glue_syntax.py).  Types, macros and prototypes only -- nothing here is ever compiled into an object, linked or shipped, and no reference source is
 * built with it.  Written from the PETSc manual pages' signatures (3.21+); opaque handles, no struct layouts beyond what PERMON's private headers dereference.
 * What the check buys: typos, undeclared identifiers, wrong arity / argument types against PERMON's own prototypes (include/permonqps.h, permon/private/qpsimpl.h:12-24,
 * qpcimpl.h:8-25 ...) and against these signatures.  What it cannot: PETSc's real struct layouts and macro expansions, linking, running. */
#pragma once
#include <math.h>
#include <stdarg.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define PETSC_VERSION_MAJOR 3
#define PETSC_VERSION_MINOR 23
#define PETSC_VERSION_SUBMINOR 0
#define PETSC_VERSION_RELEASE 1
#define PETSC_VERSION_LT(a, b, c) (PETSC_VERSION_MAJOR < (a) || (PETSC_VERSION_MAJOR == (a) && (PETSC_VERSION_MINOR < (b) || (PETSC_VERSION_MINOR == (b) && PETSC_VERSION_SUBMINOR < (c)))))
#define PETSC_VERSION_GE(a, b, c) (!PETSC_VERSION_LT(a, b, c))
#define PETSC_VERSION_LE(a, b, c) (PETSC_VERSION_LT(a, b, c) || (PETSC_VERSION_MAJOR == (a) && PETSC_VERSION_MINOR == (b) && PETSC_VERSION_SUBMINOR == (c)))
#define PETSC_VERSION_GT(a, b, c) (!PETSC_VERSION_LE(a, b, c))
#define PETSC_EXTERN extern
#define PETSC_INTERN extern
#define PETSC_SINGLE_LIBRARY_INTERN extern
#define PETSC_UNUSED __attribute__((unused))
#define PETSC_HAVE_HIP 1
#define PETSC_MAX_PATH_LEN 4096
#define PETSC_MACHINE_EPSILON 2.2204460492503131e-16
#define PETSC_MAX_REAL 1.7976931348623157e+308
#define PETSC_MIN_REAL (-PETSC_MAX_REAL)
#define PETSC_INFINITY (PETSC_MAX_REAL / 4)
#define PETSC_NINFINITY (-PETSC_INFINITY)
#define PETSC_SQRT_MACHINE_EPSILON 1.490116119384766e-08
#define PETSC_SMALL 1.e-10
#define PETSC_MAX_INT 2147483647
#define PETSC_INT_MAX 2147483647

typedef int       PetscErrorCode;
typedef int       PetscInt;
typedef int       PetscMPIInt;
typedef int       PetscBLASInt;
typedef int64_t   PetscInt64;
typedef int64_t   PetscCount;
typedef double    PetscReal;
typedef double    PetscScalar;
typedef double    MatScalar;
typedef double    MatReal;
typedef double    PetscLogDouble;
typedef int       PetscClassId;
typedef int       PetscLogEvent;
typedef int       PetscLogStage;
typedef int64_t   PetscObjectState;
typedef int64_t   PetscObjectId;
typedef size_t    PetscSizeT;
typedef short     PetscShort;
typedef void    (*PetscVoidFn)(void);
typedef PetscVoidFn *PetscVoidFunction;
typedef PetscErrorCode PetscErrorCodeFn(void);
typedef enum { PETSC_FALSE, PETSC_TRUE } PetscBool;
typedef enum { PETSC_BOOL3_FALSE, PETSC_BOOL3_TRUE, PETSC_BOOL3_UNKNOWN = -1 } PetscBool3;
typedef enum { PETSC_COPY_VALUES, PETSC_OWN_POINTER, PETSC_USE_POINTER } PetscCopyMode;
typedef enum { NOT_SET_VALUES, INSERT_VALUES, ADD_VALUES, MAX_VALUES, MIN_VALUES, INSERT_ALL_VALUES, ADD_ALL_VALUES, INSERT_BC_VALUES, ADD_BC_VALUES } InsertMode;
typedef enum { SCATTER_FORWARD = 0, SCATTER_REVERSE = 1, SCATTER_FORWARD_LOCAL = 2, SCATTER_REVERSE_LOCAL = 3 } ScatterMode;
typedef enum { NORM_1 = 0, NORM_2 = 1, NORM_FROBENIUS = 2, NORM_INFINITY = 3, NORM_1_AND_2 = 4 } NormType;
typedef enum { PETSC_OFFLOAD_UNALLOCATED = 0x0, PETSC_OFFLOAD_CPU = 0x1, PETSC_OFFLOAD_GPU = 0x2, PETSC_OFFLOAD_BOTH = 0x3, PETSC_OFFLOAD_KOKKOS = 0x100 } PetscOffloadMask;
typedef enum { PETSC_MEMTYPE_HOST = 0, PETSC_MEMTYPE_DEVICE = 1, PETSC_MEMTYPE_HIP = 5 } PetscMemType;
typedef enum { PETSC_INT = 16, PETSC_DOUBLE = 1, PETSC_BOOL = 14 } PetscDataType;
typedef const char *VecType, *MatType, *KSPType, *PCType, *ISType, *TaoType, *MatSolverType, *MatOrderingType, *PetscViewerType, *PetscRandomType, *ISLocalToGlobalMappingType, *PetscSFType, *VecScatterType;
#define PETSC_SUCCESS 0
#define PETSC_ERR_MEM 55
#define PETSC_ERR_SUP 56
#define PETSC_ERR_SUP_SYS 57
#define PETSC_ERR_ORDER 58
#define PETSC_ERR_SIG 59
#define PETSC_ERR_FP 72
#define PETSC_ERR_COR 74
#define PETSC_ERR_LIB 76
#define PETSC_ERR_PLIB 77
#define PETSC_ERR_MEMC 78
#define PETSC_ERR_CONV_FAILED 82
#define PETSC_ERR_USER 83
#define PETSC_ERR_SYS 88
#define PETSC_ERR_POINTER 70
#define PETSC_ERR_MPI_LIB_INCOMP 87
#define PETSC_ERR_ARG_SIZ 60
#define PETSC_ERR_ARG_IDN 61
#define PETSC_ERR_ARG_WRONG 62
#define PETSC_ERR_ARG_CORRUPT 64
#define PETSC_ERR_ARG_OUTOFRANGE 63
#define PETSC_ERR_ARG_BADPTR 68
#define PETSC_ERR_ARG_NOTSAMETYPE 69
#define PETSC_ERR_ARG_NOTSAMECOMM 80
#define PETSC_ERR_ARG_WRONGSTATE 73
#define PETSC_ERR_ARG_TYPENOTSET 89
#define PETSC_ERR_ARG_INCOMP 75
#define PETSC_ERR_ARG_NULL 85
#define PETSC_ERR_ARG_UNKNOWN_TYPE 86
#define PETSC_ERR_FILE_OPEN 65
#define PETSC_ERR_NOT_CONVERGED 91
#define PETSC_ERR_MAT_LU_ZRPVT 71
#define PETSC_DECIDE (-1)
#define PETSC_DETERMINE PETSC_DECIDE
#define PETSC_DEFAULT (-2)
#define PETSC_CURRENT (-2)
#define PETSC_UNLIMITED (-3)
#define PETSC_IGNORE NULL
#define PETSC_NULLPTR NULL
#define PetscInt_FMT "d"
#define PetscInt64_FMT "ld"
#define PetscCount_FMT "ld"
#define PetscBLASInt_FMT "d"
#define MPIU_SCALAR MPI_DOUBLE
#define MPIU_REAL MPI_DOUBLE
#define MPIU_INT MPI_INT
#define MPIU_BOOL MPI_INT
#define MPIU_SUM MPI_SUM
#define MPIU_MAX MPI_MAX
#define MPIU_MIN MPI_MIN
#define PetscUnlikely(c) __builtin_expect(!!(c), 0)
#define PetscLikely(c) __builtin_expect(!!(c), 1)
#define PetscDefined(x) 0
#define PetscUnlikelyDebug(c) 0
#define PetscAbsScalar(a) fabs(a)
#define PetscAbsReal(a) fabs(a)
#define PetscAbsInt(a) (((a) < 0) ? -(a) : (a))
#define PetscSqrtReal(a) sqrt(a)
#define PetscSqrtScalar(a) sqrt(a)
#define PetscRealPart(a) (a)
#define PetscImaginaryPart(a) 0.0
#define PetscPowReal(a, b) pow(a, b)
#define PetscPowScalar(a, b) pow(a, b)
#define PetscLog10Real(a) log10(a)
#define PetscMax(a, b) (((a) < (b)) ? (b) : (a))
#define PetscMin(a, b) (((a) < (b)) ? (a) : (b))
#define PetscSqr(a) ((a) * (a))
#define PetscIsInfOrNanReal(a) (isinf(a) || isnan(a))
#define PetscIsInfOrNanScalar(a) (isinf(a) || isnan(a))
#define PetscIsNanReal(a) isnan(a)
#define PetscIsInfReal(a) isinf(a)

/* MPI */
typedef int MPI_Comm, MPI_Datatype, MPI_Op, MPI_Request;
typedef struct { int MPI_SOURCE, MPI_TAG, MPI_ERROR; } MPI_Status;
#define MPI_COMM_WORLD 1
#define MPI_COMM_SELF 2
#define MPI_COMM_NULL 0
#define MPI_DOUBLE 11
#define MPI_INT 6
#define MPI_BYTE 2
#define MPI_CHAR 1
#define MPI_SUM 3
#define MPI_MAX 1
#define MPI_MIN 2
#define MPI_IN_PLACE ((void *)1)
#define MPI_SUCCESS 0
int MPI_Comm_rank(MPI_Comm, int *);
int MPI_Comm_size(MPI_Comm, int *);
int MPI_Bcast(void *, int, MPI_Datatype, int, MPI_Comm);
int MPI_Allreduce(const void *, void *, int, MPI_Datatype, MPI_Op, MPI_Comm);
int MPI_Barrier(MPI_Comm);
int MPI_Comm_compare(MPI_Comm, MPI_Comm, int *);
int MPI_Comm_dup(MPI_Comm, MPI_Comm *);
#define MPIU_Allreduce(a, b, c, d, e, f) MPI_Allreduce(a, b, c, d, e, f)
#define MPIU_Allreduce_Private MPI_Allreduce
extern MPI_Comm PETSC_COMM_WORLD;
#define PETSC_COMM_SELF MPI_COMM_SELF

/* objects: opaque handles */
typedef struct _p_PetscObject *PetscObject;
typedef struct _p_Vec *Vec;
typedef struct _p_Mat *Mat;
typedef struct _p_KSP *KSP;
typedef struct _p_PC *PC;
typedef struct _p_IS *IS;
typedef struct _p_Tao *Tao;
typedef struct _p_PetscViewer *PetscViewer;
typedef struct _p_PetscSF *PetscSF;
typedef PetscSF VecScatter;
typedef struct _p_PetscRandom *PetscRandom;
typedef struct _p_PetscContainer *PetscContainer;
typedef struct _p_ISLocalToGlobalMapping *ISLocalToGlobalMapping;
typedef struct _p_PetscLayout *PetscLayout;
typedef struct _p_MatNullSpace *MatNullSpace;
typedef struct _p_PetscOptions *PetscOptions;
typedef struct _p_PetscOptionItems *PetscOptionItems;
typedef struct _p_PetscDeviceContext *PetscDeviceContext;
typedef struct _p_PetscDevice *PetscDevice;
typedef struct _p_DM *DM;
typedef struct _p_PetscDraw *PetscDraw;
typedef struct _p_PetscDrawLG *PetscDrawLG;
typedef struct _p_PetscFunctionList *PetscFunctionList;
typedef struct _p_PetscObjectList *PetscObjectList;
typedef struct _p_MatCoarsen *MatCoarsen;
typedef struct _p_PetscHMapI *PetscHMapI;
typedef struct _p_PetscHMapIJV *PetscHMapIJV;
typedef struct _p_PetscHSetI *PetscHSetI;
typedef struct _p_PetscTable *PetscTable;
typedef struct _p_MatPartitioning *MatPartitioning;
typedef struct _p_KSPGuess *KSPGuess;
typedef struct _p_TaoLineSearch *TaoLineSearch;
typedef struct _p_VecTagger *VecTagger;
typedef struct _n_PetscBT *PetscBT_;
typedef char *PetscBT;
typedef struct { PetscInt rank, index; } PetscSFNode;
typedef struct { PetscInt dummy[8]; } MatStash;
typedef struct { PetscReal diagonal_fill, usedt, dt, dtcol, dtcount, fill, levels, pivotinblocks, zeropivot, shifttype, shiftamount; PetscBool factoronhost, solveonhost; } MatFactorInfo;
typedef struct { PetscLogDouble block_size, nz_allocated, nz_used, nz_unneeded, memory, assemblies, mallocs, fill_ratio_given, fill_ratio_needed, factor_mallocs; } MatInfo;
typedef struct { PetscInt nzerorows; } MatFactorError_;
typedef enum { MAT_FACTOR_NONE, MAT_FACTOR_LU, MAT_FACTOR_CHOLESKY, MAT_FACTOR_ILU, MAT_FACTOR_ICC, MAT_FACTOR_ILUDT, MAT_FACTOR_QR, MAT_FACTOR_NUM_TYPES } MatFactorType;
typedef enum { MAT_INITIAL_MATRIX, MAT_REUSE_MATRIX, MAT_IGNORE_MATRIX, MAT_INPLACE_MATRIX } MatReuse;
typedef enum { MAT_DO_NOT_COPY_VALUES, MAT_COPY_VALUES, MAT_SHARE_NONZERO_PATTERN } MatDuplicateOption;
typedef enum { DIFFERENT_NONZERO_PATTERN, SUBSET_NONZERO_PATTERN, SAME_NONZERO_PATTERN, UNKNOWN_NONZERO_PATTERN } MatStructure;
typedef enum { MAT_FLUSH_ASSEMBLY = 1, MAT_FINAL_ASSEMBLY = 0 } MatAssemblyType;
typedef enum { MAT_SYMMETRIC = 4, MAT_SPD = 18, MAT_NEW_NONZERO_ALLOCATION_ERR = 16, MAT_SYMMETRY_ETERNAL = 14, MAT_STRUCTURALLY_SYMMETRIC = 5, MAT_ROW_ORIENTED = 1 } MatOption;
typedef enum { MAT_COMPOSITE_ADDITIVE, MAT_COMPOSITE_MULTIPLICATIVE } MatCompositeType;
typedef enum { MATOP_MULT = 3, MATOP_MULT_ADD = 4, MATOP_MULT_TRANSPOSE = 5, MATOP_MULT_TRANSPOSE_ADD = 6, MATOP_DESTROY = 60, MATOP_GET_DIAGONAL = 17, MATOP_DUPLICATE = 34, MATOP_VIEW = 59 } MatOperation;
typedef enum { MATPRODUCT_UNSPECIFIED = 0, MATPRODUCT_AB, MATPRODUCT_AtB, MATPRODUCT_ABt, MATPRODUCT_PtAP, MATPRODUCT_RARt, MATPRODUCT_ABC } MatProductType;
typedef enum { KSP_NORM_DEFAULT = -1, KSP_NORM_NONE = 0, KSP_NORM_PRECONDITIONED = 1, KSP_NORM_UNPRECONDITIONED = 2, KSP_NORM_NATURAL = 3 } KSPNormType;
typedef enum { PC_SIDE_DEFAULT = -1, PC_LEFT, PC_RIGHT, PC_SYMMETRIC } PCSide;
typedef enum {
  KSP_CONVERGED_RTOL_NORMAL = 1, KSP_CONVERGED_ATOL_NORMAL = 9, KSP_CONVERGED_RTOL = 2, KSP_CONVERGED_ATOL = 3, KSP_CONVERGED_ITS = 4, KSP_CONVERGED_NEG_CURVE = 5, KSP_CONVERGED_STEP_LENGTH = 6,
  KSP_CONVERGED_HAPPY_BREAKDOWN = 7, KSP_DIVERGED_NULL = -2, KSP_DIVERGED_ITS = -3, KSP_DIVERGED_DTOL = -4, KSP_DIVERGED_BREAKDOWN = -5, KSP_DIVERGED_BREAKDOWN_BICG = -6, KSP_DIVERGED_NONSYMMETRIC = -7,
  KSP_DIVERGED_INDEFINITE_PC = -8, KSP_DIVERGED_NANORINF = -9, KSP_DIVERGED_INDEFINITE_MAT = -10, KSP_DIVERGED_PC_FAILED = -11, KSP_CONVERGED_ITERATING = 0
} KSPConvergedReason;
typedef enum { PETSC_VIEWER_DEFAULT, PETSC_VIEWER_ASCII_INFO = 4, PETSC_VIEWER_ASCII_INFO_DETAIL = 5 } PetscViewerFormat;
typedef enum { PETSC_DEVICE_HOST, PETSC_DEVICE_CUDA, PETSC_DEVICE_HIP, PETSC_DEVICE_SYCL } PetscDeviceType;
typedef enum { IS_GTOLM_MASK, IS_GTOLM_DROP } ISGlobalToLocalMappingMode;
typedef struct { PetscInt dummy; } PetscViewerAndFormat;
extern const char *const KSPConvergedReasons_Shifted[];
extern const char *const *KSPConvergedReasons;
extern PetscClassId MAT_CLASSID, VEC_CLASSID, KSP_CLASSID, PC_CLASSID, IS_CLASSID, PETSC_VIEWER_CLASSID, PETSC_OBJECT_CLASSID;

/* the object header PERMON's own classes embed (petsc/private/petscimpl.h); only the fields PERMON and the glue touch */
typedef struct _p_PetscObject {
  PetscClassId     classid;
  MPI_Comm         comm;
  PetscObjectId    id;
  PetscInt         refct;
  char            *type_name, *name, *prefix, *class_name, *description, *mansec;
  PetscObjectState state;
  PetscObjectList  olist;
  PetscFunctionList qlist;
  PetscOptions     options;
  PetscBool        optionsprinted;
  void            *python_context;
} _p_PetscObject;
#define PETSCHEADER(ObjectOps) \
  _p_PetscObject hdr; \
  ObjectOps      ops[1]
#define PetscHeaderCreate(h, classid, class_name, descr, mansec, comm, destroy, view) PetscHeaderCreate_Private((PetscObject *)&(h), sizeof(*(h)), (classid), (class_name), (descr), (mansec), (comm), (PetscErrorCode(*)(PetscObject *))(destroy), (PetscErrorCode(*)(PetscObject, PetscViewer))(view))
PetscErrorCode PetscHeaderCreate_Private(PetscObject *, size_t, PetscClassId, const char[], const char[], const char[], MPI_Comm, PetscErrorCode (*)(PetscObject *), PetscErrorCode (*)(PetscObject, PetscViewer));
#define PetscHeaderDestroy(h) PetscHeaderDestroy_Private((PetscObject *)(h))
PetscErrorCode PetscHeaderDestroy_Private(PetscObject *);

/* error handling / control-flow macros (simplified: the real ones also maintain PETSc's stack) */
#define PetscFunctionBegin do { } while (0)
#define PetscFunctionBeginUser do { } while (0)
#define PetscFunctionBeginHot do { } while (0)
#define PetscFunctionReturn(...) return __VA_ARGS__
#define PetscFunctionReturnVoid() return
PetscErrorCode PetscError(MPI_Comm, int, const char *, const char *, PetscErrorCode, int, const char *, ...) __attribute__((format(printf, 7, 8)));
#define SETERRQ(comm, ierr, ...) return PetscError(comm, __LINE__, __func__, __FILE__, ierr, 0, __VA_ARGS__)
#define PetscCheck(cond, comm, ierr, ...) \
  do { \
    if (PetscUnlikely(!(cond))) SETERRQ(comm, ierr, __VA_ARGS__); \
  } while (0)
#define PetscAssert(cond, comm, ierr, ...) PetscCheck(cond, comm, ierr, __VA_ARGS__)
#define PetscCall(...) \
  do { \
    PetscErrorCode ierr_petsc_call_q_ = (__VA_ARGS__); \
    if (PetscUnlikely(ierr_petsc_call_q_ != PETSC_SUCCESS)) return PetscError(PETSC_COMM_SELF, __LINE__, __func__, __FILE__, ierr_petsc_call_q_, 1, " "); \
  } while (0)
#define PetscCallMPI(...) \
  do { \
    int ierr_mpi_ = (__VA_ARGS__); \
    if (PetscUnlikely(ierr_mpi_ != MPI_SUCCESS)) return PetscError(PETSC_COMM_SELF, __LINE__, __func__, __FILE__, PETSC_ERR_MPI_LIB_INCOMP, 1, " "); \
  } while (0)
#define PetscCallAbort(comm, ...) do { (void)(__VA_ARGS__); } while (0)
#define PetscCallVoid(...) do { (void)(__VA_ARGS__); } while (0)
#define CHKERRQ(ierr) PetscCall(ierr)
#define PetscValidHeaderSpecific(h, ck, arg) do { (void)(h); } while (0)
#define PetscValidHeader(h, arg) do { (void)(h); } while (0)
#define PetscValidLogicalCollectiveReal(h, v, arg) do { (void)(h); } while (0)
#define PetscValidLogicalCollectiveInt(h, v, arg) do { (void)(h); } while (0)
#define PetscValidLogicalCollectiveBool(h, v, arg) do { (void)(h); } while (0)
#define PetscValidLogicalCollectiveEnum(h, v, arg) do { (void)(h); } while (0)
#define PetscValidLogicalCollectiveScalar(h, v, arg) do { (void)(h); } while (0)
#define PetscValidType(h, arg) do { (void)(h); } while (0)
#define PetscCheckSameComm(a, arga, b, argb) do { (void)(a); (void)(b); } while (0)
#define PetscCheckSameTypeAndComm(a, arga, b, argb) do { (void)(a); (void)(b); } while (0)
#define PetscAssertPointer(p, arg) do { (void)(p); } while (0)
#define PetscValidPointer(p, arg) do { (void)(p); } while (0)
#define PetscValidRealPointer(p, arg) do { (void)(p); } while (0)
#define PetscValidIntPointer(p, arg) do { (void)(p); } while (0)
#define PetscValidBoolPointer(p, arg) do { (void)(p); } while (0)
#define PetscValidScalarPointer(p, arg) do { (void)(p); } while (0)
#define PetscValidCharPointer(p, arg) do { (void)(p); } while (0)
#define PetscValidFunction(p, arg) do { (void)(p); } while (0)

/* memory */
PetscErrorCode PetscMallocA(int, PetscBool, int, const char *, const char *, size_t, void *, ...);
PetscErrorCode PetscFreeA(int, int, const char *, const char *, void *, ...);
#define PetscNew(b) PetscMallocA(1, PETSC_TRUE, __LINE__, __func__, __FILE__, sizeof(**(b)), (b))
#define PetscMalloc1(m1, r1) PetscMallocA(1, PETSC_FALSE, __LINE__, __func__, __FILE__, (size_t)(m1) * sizeof(**(r1)), (r1))
#define PetscCalloc1(m1, r1) PetscMallocA(1, PETSC_TRUE, __LINE__, __func__, __FILE__, (size_t)(m1) * sizeof(**(r1)), (r1))
#define PetscMalloc2(m1, r1, m2, r2) PetscMallocA(2, PETSC_FALSE, __LINE__, __func__, __FILE__, (size_t)(m1) * sizeof(**(r1)), (r1), (size_t)(m2) * sizeof(**(r2)), (r2))
#define PetscMalloc3(m1, r1, m2, r2, m3, r3) PetscMallocA(3, PETSC_FALSE, __LINE__, __func__, __FILE__, (size_t)(m1) * sizeof(**(r1)), (r1), (size_t)(m2) * sizeof(**(r2)), (r2), (size_t)(m3) * sizeof(**(r3)), (r3))
#define PetscMalloc4(m1, r1, m2, r2, m3, r3, m4, r4) \
  PetscMallocA(4, PETSC_FALSE, __LINE__, __func__, __FILE__, (size_t)(m1) * sizeof(**(r1)), (r1), (size_t)(m2) * sizeof(**(r2)), (r2), (size_t)(m3) * sizeof(**(r3)), (r3), (size_t)(m4) * sizeof(**(r4)), (r4))
#define PetscFree(a) ((PetscErrorCode)((a) ? (PetscFreeA(1, __LINE__, __func__, __FILE__, &(a))) : 0))
#define PetscFree2(m1, m2) PetscFreeA(2, __LINE__, __func__, __FILE__, &(m1), &(m2))
#define PetscFree3(m1, m2, m3) PetscFreeA(3, __LINE__, __func__, __FILE__, &(m1), &(m2), &(m3))
#define PetscFree4(m1, m2, m3, m4) PetscFreeA(4, __LINE__, __func__, __FILE__, &(m1), &(m2), &(m3), &(m4))
PetscErrorCode PetscArraycpy_(void *, const void *, size_t);
#define PetscArraycpy(a, b, n) PetscArraycpy_((a), (b), (size_t)(n) * sizeof(*(a)))
#define PetscArrayzero(a, n) PetscArraycpy_((a), (a), 0 * (size_t)(n))
PetscErrorCode PetscMemcpy(void *, const void *, size_t);
PetscErrorCode PetscMemzero(void *, size_t);
PetscErrorCode PetscStrallocpy(const char[], char *[]);
PetscErrorCode PetscStrcmp(const char[], const char[], PetscBool *);
PetscErrorCode PetscStrlen(const char[], size_t *);
PetscErrorCode PetscSNPrintf(char *, size_t, const char[], ...) __attribute__((format(printf, 3, 4)));
PetscErrorCode PetscPrintf(MPI_Comm, const char[], ...) __attribute__((format(printf, 2, 3)));
PetscErrorCode PetscInfo_Private(const char[], PetscObject, const char[], ...);
#define PetscInfo(A, ...) PetscInfo_Private(__func__, ((PetscObject)A), __VA_ARGS__)
PetscErrorCode PetscTime(PetscLogDouble *);
PetscErrorCode PetscLogFlops(PetscLogDouble);
PetscErrorCode PetscLogEventBegin(PetscLogEvent, ...);
PetscErrorCode PetscLogEventEnd(PetscLogEvent, ...);
PetscErrorCode PetscCitationsRegister(const char[], PetscBool *);

/* ---- more types PERMON's copies of PETSc's private Mat structs (include/permon/private/petsc/*.h) name ---- */
#include <sys/types.h>
typedef enum { PETSC_SUBCOMM_GENERAL = 0, PETSC_SUBCOMM_CONTIGUOUS = 1, PETSC_SUBCOMM_INTERLACED = 2 } PetscSubcommType;
typedef struct _n_PetscSubcomm *PetscSubcomm;
typedef struct _p_MatTransposeColoring *MatTransposeColoring;
typedef struct _p_MatColoring *MatColoring;
typedef struct _p_MatFDColoring *MatFDColoring;
typedef struct _n_PetscEventRegLog *PetscEventRegLog;
typedef PetscErrorCode PetscCtxDestroyFn(void **);
typedef struct _p_PetscHSetIJ *PetscHSetIJ;
typedef struct { PetscBool use, check; PetscInt nrows, *i, *rindex; } Mat_CompressedRow;
#define MPI_C_BOOL 30
#define MPI_LOR 7
#define MPI_LAND 6
#define MPIU_2INT 31

/* ---- the private object structs as far as PERMON's classes and the glue dereference them (petsc/private/{matimpl,pcimpl,kspimpl}.h): ops tables with the mult / apply
 * slots the glue fills, the `data` pointer, KSP's iteration state ---- */
typedef unsigned int PetscEnum;
typedef enum { MAT_SHIFT_NONE, MAT_SHIFT_NONZERO, MAT_SHIFT_POSITIVE_DEFINITE, MAT_SHIFT_INBLOCKS } MatFactorShiftType;
struct _MatOps {
  PetscErrorCode (*setvalues)(Mat, PetscInt, const PetscInt[], PetscInt, const PetscInt[], const PetscScalar[], InsertMode);
  PetscErrorCode (*getrow)(Mat, PetscInt, PetscInt *, PetscInt *[], PetscScalar *[]);
  PetscErrorCode (*restorerow)(Mat, PetscInt, PetscInt *, PetscInt *[], PetscScalar *[]);
  PetscErrorCode (*mult)(Mat, Vec, Vec);
  PetscErrorCode (*multadd)(Mat, Vec, Vec, Vec);
  PetscErrorCode (*multtranspose)(Mat, Vec, Vec);
  PetscErrorCode (*multtransposeadd)(Mat, Vec, Vec, Vec);
  PetscErrorCode (*solve)(Mat, Vec, Vec);
  PetscErrorCode (*getdiagonal)(Mat, Vec);
  PetscErrorCode (*duplicate)(Mat, MatDuplicateOption, Mat *);
  PetscErrorCode (*destroy)(Mat);
  PetscErrorCode (*view)(Mat, PetscViewer);
  PetscErrorCode (*setfromoptions)(Mat, PetscOptionItems);
  PetscErrorCode (*assemblybegin)(Mat, MatAssemblyType);
  PetscErrorCode (*assemblyend)(Mat, MatAssemblyType);
  PetscErrorCode (*setup)(Mat);
  PetscErrorCode (*createvecs)(Mat, Vec *, Vec *);
  PetscErrorCode (*matmult)(Mat, Mat, Mat);
  PetscErrorCode (*productsetfromoptions)(Mat);
  PetscErrorCode (*scale)(Mat, PetscScalar);
  PetscErrorCode (*shift)(Mat, PetscScalar);
  PetscErrorCode (*zeroentries)(Mat);
  PetscErrorCode (*transpose)(Mat, MatReuse, Mat *);
  PetscErrorCode (*getinfo)(Mat, int, MatInfo *);
  PetscErrorCode (*convert)(Mat, MatType, MatReuse, Mat *);
  PetscErrorCode (*axpy)(Mat, PetscScalar, Mat, MatStructure);
  PetscErrorCode (*copy)(Mat, Mat, MatStructure);
  PetscErrorCode (*createsubmatrix)(Mat, IS, IS, MatReuse, Mat *);
  PetscErrorCode (*getlocaltoglobalmapping)(Mat, ISLocalToGlobalMapping *, ISLocalToGlobalMapping *);
};
struct _p_Mat {
  PETSCHEADER(struct _MatOps);
  PetscLayout      rmap, cmap;
  void            *data;
  MatFactorType    factortype;
  PetscBool        assembled, was_assembled;
  PetscInt         num_ass;
  PetscObjectState nonzerostate;
  MatInfo          info;
  InsertMode       insertmode;
  MatStash         stash, bstash;
  MatNullSpace     nullsp, transnullsp, nearnullsp;
  PetscBool        preallocated;
  PetscBool3       symmetric, hermitian, structurally_symmetric, spd;
  PetscBool        symmetry_eternal, structural_symmetry_eternal, spd_eternal;
  PetscBool        nooffprocentries, nooffproczerorows, assembly_subset, submat_singleis;
  void            *spptr;
  char            *solvertype;
  PetscBool        checksymmetryonassembly, checknullspaceonassembly;
  PetscReal        checksymmetrytol;
  Mat              schur;
  VecType          defaultvectype;
  PetscBool        boundtocpu, bindingpropagates;
  PetscOffloadMask offloadmask;
  void            *product;
};
struct _PCOps {
  PetscErrorCode (*setup)(PC);
  PetscErrorCode (*apply)(PC, Vec, Vec);
  PetscErrorCode (*matapply)(PC, Mat, Mat);
  PetscErrorCode (*applyrichardson)(PC, Vec, Vec, Vec, PetscReal, PetscReal, PetscReal, PetscInt, PetscBool, PetscInt *, int *);
  PetscErrorCode (*applyBA)(PC, PCSide, Vec, Vec, Vec);
  PetscErrorCode (*applytranspose)(PC, Vec, Vec);
  PetscErrorCode (*applyBAtranspose)(PC, PetscInt, Vec, Vec, Vec);
  PetscErrorCode (*setfromoptions)(PC, PetscOptionItems);
  PetscErrorCode (*presolve)(PC, KSP, Vec, Vec);
  PetscErrorCode (*postsolve)(PC, KSP, Vec, Vec);
  PetscErrorCode (*getfactoredmatrix)(PC, Mat *);
  PetscErrorCode (*applysymmetricleft)(PC, Vec, Vec);
  PetscErrorCode (*applysymmetricright)(PC, Vec, Vec);
  PetscErrorCode (*setuponblocks)(PC);
  PetscErrorCode (*destroy)(PC);
  PetscErrorCode (*view)(PC, PetscViewer);
  PetscErrorCode (*reset)(PC);
  PetscErrorCode (*load)(PC, PetscViewer);
};
struct _p_PC {
  PETSCHEADER(struct _PCOps);
  DM               dm;
  PetscInt         setupcalled;
  PetscObjectState matstate, matnonzerostate;
  MatStructure     flag;
  Mat              mat, pmat;
  Vec              diagonalscaleright, diagonalscaleleft;
  PetscBool        diagonalscale, useAmat, setfromoptionscalled, erroriffailure;
  PetscInt         reusepreconditioner;
  void            *data;
};
struct _KSPOps {
  PetscErrorCode (*buildsolution)(KSP, Vec, Vec *);
  PetscErrorCode (*buildresidual)(KSP, Vec, Vec, Vec *);
  PetscErrorCode (*matsolve)(KSP, Mat, Mat);
  PetscErrorCode (*solve)(KSP);
  PetscErrorCode (*setup)(KSP);
  PetscErrorCode (*setfromoptions)(KSP, PetscOptionItems);
  PetscErrorCode (*publishoptions)(KSP);
  PetscErrorCode (*computeextremesingularvalues)(KSP, PetscReal *, PetscReal *);
  PetscErrorCode (*computeeigenvalues)(KSP, PetscInt, PetscReal *, PetscReal *, PetscInt *);
  PetscErrorCode (*computeritz)(KSP, PetscBool, PetscBool, PetscInt *, Vec[], PetscReal *, PetscReal *);
  PetscErrorCode (*destroy)(KSP);
  PetscErrorCode (*view)(KSP, PetscViewer);
  PetscErrorCode (*reset)(KSP);
  PetscErrorCode (*load)(KSP, PetscViewer);
};
struct _p_KSP {
  PETSCHEADER(struct _KSPOps);
  DM                 dm;
  PetscBool          dmAuto, dmActive;
  PetscInt           max_it, min_it;
  KSPGuess           guess;
  PetscBool          guess_zero, guess_not_read, calc_sings, calc_ritz;
  PCSide             pc_side;
  PetscReal          rtol, abstol, ttol, divtol, rnorm0, rnorm;
  KSPConvergedReason reason;
  PetscBool          errorifnotconverged;
  Vec                vec_sol, vec_rhs;
  PetscReal         *res_hist;
  PetscInt           res_hist_len, res_hist_max;
  PetscBool          res_hist_reset;
  PetscInt           chknorm;
  PetscBool          lagnorm;
  PetscInt           numbermonitors;
  PetscErrorCode (*converged)(KSP, PetscInt, PetscReal, KSPConvergedReason *, void *);
  PetscErrorCode (*convergeddestroy)(void *);
  PetscErrorCode (*user_convergeddestroy)(void *);
  void              *cnvP, *user;
  PC                 pc;
  void              *data;
  PetscBool          view, viewPre, viewRate, viewMat, viewPMat, viewRhs, viewSol, viewMatExp, viewEV, viewSV, viewEVExp, viewFinalRes, viewPOpExp, viewDScale;
  PetscInt           setupstage, setupnewmatrix;
  PetscInt           its, totalits;
  PetscBool          transpose_solve;
  KSPNormType        normtype;
  PCSide             pc_side_set;
  KSPNormType        normtype_set;
  Vec               *work;
  PetscInt           nwork;
  PetscInt           setfromoptionscalled;
  PetscBool          skippcsetfromoptions;
};

/* ---- type names ---- */
#define VECSEQ "seq"
#define VECMPI "mpi"
#define VECSTANDARD "standard"
#define VECHIP "hip"
#define VECSEQHIP "seqhip"
#define VECMPIHIP "mpihip"
#define MATSEQAIJ "seqaij"
#define MATMPIAIJ "mpiaij"
#define MATAIJ "aij"
#define MATSEQAIJHIPSPARSE "seqaijhipsparse"
#define MATAIJHIPSPARSE "aijhipsparse"
#define MATSEQDENSE "seqdense"
#define MATDENSE "dense"
#define MATSEQSBAIJ "seqsbaij"
#define MATSBAIJ "sbaij"
#define MATSEQBAIJ "seqbaij"
#define MATSHELL "shell"
#define MATCOMPOSITE "composite"
#define MATNEST "nest"
#define MATTRANSPOSEVIRTUAL "transpose"
#define MATIS "is"
#define MATORDERINGNATURAL "natural"
#define MATORDERINGND "nd"
#define MATSOLVERPETSC "petsc"
#define MATSOLVERMUMPS "mumps"
#define MATSOLVERSUPERLU "superlu"
#define MATSOLVERSUPERLU_DIST "superlu_dist"
#define PCNONE "none"
#define PCJACOBI "jacobi"
#define PCBJACOBI "bjacobi"
#define PCCHOLESKY "cholesky"
#define PCLU "lu"
#define PCMG "mg"
#define PCGAMG "gamg"
#define PCREDUNDANT "redundant"
#define PCSHELL "shell"
#define PCICC "icc"
#define KSPCG "cg"
#define KSPPREONLY "preonly"
#define KSPCHEBYSHEV "chebyshev"
#define KSPRICHARDSON "richardson"
#define KSPGMRES "gmres"
#define PETSCVIEWERASCII "ascii"
#define PETSCRAND48 "rand48"
PetscViewer PETSC_VIEWER_STDOUT_(MPI_Comm);
#define PETSC_VIEWER_STDOUT_WORLD PETSC_VIEWER_STDOUT_(PETSC_COMM_WORLD)
#define PETSC_VIEWER_STDOUT_SELF PETSC_VIEWER_STDOUT_(PETSC_COMM_SELF)

/* ---- PetscObject / options / viewer / container / device ---- */
MPI_Comm       PetscObjectComm(PetscObject);
PetscErrorCode PetscObjectGetComm(PetscObject, MPI_Comm *);
PetscErrorCode PetscObjectReference(PetscObject);
PetscErrorCode PetscObjectDereference(PetscObject);
PetscErrorCode PetscObjectCompose(PetscObject, const char[], PetscObject);
PetscErrorCode PetscObjectQuery(PetscObject, const char[], PetscObject *);
PetscErrorCode PetscObjectComposeFunction_Private(PetscObject, const char[], void (*)(void));
#define PetscObjectComposeFunction(a, b, ...) PetscObjectComposeFunction_Private((a), (b), (PetscVoidFn)(__VA_ARGS__))
PetscErrorCode PetscObjectQueryFunction_Private(PetscObject, const char[], void (**)(void));
#define PetscObjectQueryFunction(obj, name, fptr) PetscObjectQueryFunction_Private((obj), (name), (PetscVoidFn *)(fptr))
PetscErrorCode PetscObjectTypeCompare(PetscObject, const char[], PetscBool *);
PetscErrorCode PetscObjectTypeCompareAny(PetscObject, PetscBool *, const char[], ...);
PetscErrorCode PetscObjectBaseTypeCompare(PetscObject, const char[], PetscBool *);
PetscErrorCode PetscObjectChangeTypeName(PetscObject, const char[]);
PetscErrorCode PetscObjectStateIncrease(PetscObject);
PetscErrorCode PetscObjectStateGet(PetscObject, PetscObjectState *);
PetscErrorCode PetscObjectSetName(PetscObject, const char[]);
PetscErrorCode PetscObjectGetName(PetscObject, const char *[]);
PetscErrorCode PetscObjectGetOptionsPrefix(PetscObject, const char *[]);
PetscErrorCode PetscObjectSetOptionsPrefix(PetscObject, const char[]);
PetscErrorCode PetscObjectAppendOptionsPrefix(PetscObject, const char[]);
PetscErrorCode PetscObjectIncrementTabLevel(PetscObject, PetscObject, PetscInt);
PetscErrorCode PetscObjectGetType(PetscObject, const char *[]);
PetscErrorCode PetscObjectPrintClassNamePrefixType(PetscObject, PetscViewer);
#define PetscTryMethod(obj, A, B, C) \
  do { \
    PetscErrorCode(*_7_f) B; \
    PetscCall(PetscObjectQueryFunction((PetscObject)(obj), A, &_7_f)); \
    if (_7_f) PetscCall((*_7_f)C); \
  } while (0)
#define PetscUseMethod(obj, A, B, C) \
  do { \
    PetscErrorCode(*_7_f) B; \
    PetscCall(PetscObjectQueryFunction((PetscObject)(obj), A, &_7_f)); \
    PetscCheck(_7_f, PetscObjectComm((PetscObject)(obj)), PETSC_ERR_SUP, "Cannot locate function %s in object", A); \
    PetscCall((*_7_f)C); \
  } while (0)
#define PetscUseTypeMethod(obj, ...) PetscCall(0)
#define PetscTryTypeMethod(obj, ...) PetscCall(0)
PetscErrorCode PetscContainerCreate(MPI_Comm, PetscContainer *);
PetscErrorCode PetscContainerDestroy(PetscContainer *);
PetscErrorCode PetscContainerSetPointer(PetscContainer, void *);
PetscErrorCode PetscContainerGetPointer(PetscContainer, void **);
PetscErrorCode PetscContainerSetCtxDestroy(PetscContainer, PetscCtxDestroyFn *);
PetscErrorCode PetscContainerSetUserDestroy(PetscContainer, PetscErrorCode (*)(void *));
PetscErrorCode PetscObjectContainerCompose(PetscObject, const char *, void *, PetscCtxDestroyFn *);
PetscErrorCode PetscViewerASCIIPrintf(PetscViewer, const char[], ...) __attribute__((format(printf, 2, 3)));
PetscErrorCode PetscViewerASCIIPushTab(PetscViewer);
PetscErrorCode PetscViewerASCIIPopTab(PetscViewer);
PetscErrorCode PetscViewerASCIIGetStdout(MPI_Comm, PetscViewer *);
PetscErrorCode PetscViewerGetFormat(PetscViewer, PetscViewerFormat *);
PetscErrorCode PetscOptionsGetBool(PetscOptions, const char[], const char[], PetscBool *, PetscBool *);
PetscErrorCode PetscOptionsGetInt(PetscOptions, const char[], const char[], PetscInt *, PetscBool *);
PetscErrorCode PetscOptionsGetReal(PetscOptions, const char[], const char[], PetscReal *, PetscBool *);
PetscErrorCode PetscOptionsGetString(PetscOptions, const char[], const char[], char[], size_t, PetscBool *);
PetscErrorCode PetscOptionsHasName(PetscOptions, const char[], const char[], PetscBool *);
PetscErrorCode PetscOptionsGetAll(PetscOptions, char *[]);
PetscErrorCode PetscOptionsSetValue(PetscOptions, const char[], const char[]);
extern PetscOptionItems PetscOptionsObject; /* (the real macros thread a local of this name through PetscOptionsHeadBegin .. End) */
PetscErrorCode PetscOptionsHeadBegin_Private(PetscOptionItems, const char[]);
#define PetscOptionsHeadBegin(o, head) PetscCall(PetscOptionsHeadBegin_Private((o), (head)))
#define PetscOptionsHeadEnd() do { } while (0)
PetscErrorCode PetscOptionsReal_Private(PetscOptionItems, const char[], const char[], const char[], PetscReal, PetscReal *, PetscBool *);
PetscErrorCode PetscOptionsInt_Private(PetscOptionItems, const char[], const char[], const char[], PetscInt, PetscInt *, PetscBool *);
PetscErrorCode PetscOptionsBool_Private(PetscOptionItems, const char[], const char[], const char[], PetscBool, PetscBool *, PetscBool *);
PetscErrorCode PetscOptionsEnum_Private(PetscOptionItems, const char[], const char[], const char[], const char *const *, PetscEnum, PetscEnum *, PetscBool *);
PetscErrorCode PetscOptionsString_Private(PetscOptionItems, const char[], const char[], const char[], const char[], char *, size_t, PetscBool *);
#define PetscOptionsReal(a, b, c, d, e, f) PetscOptionsReal_Private(PetscOptionsObject, a, b, c, d, e, f)
#define PetscOptionsInt(a, b, c, d, e, f) PetscOptionsInt_Private(PetscOptionsObject, a, b, c, d, e, f)
#define PetscOptionsBool(a, b, c, d, e, f) PetscOptionsBool_Private(PetscOptionsObject, a, b, c, d, e, f)
#define PetscOptionsEnum(a, b, c, d, e, f, g) PetscOptionsEnum_Private(PetscOptionsObject, a, b, c, d, e, f, g)
#define PetscOptionsString(a, b, c, d, e, f, g) PetscOptionsString_Private(PetscOptionsObject, a, b, c, d, e, f, g)
#define PetscObjectOptionsBegin(obj) do { } while (0); do
#define PetscOptionsEnd() while (0)
PetscErrorCode PetscDeviceContextGetCurrentContext(PetscDeviceContext *);
PetscErrorCode PetscDeviceContextGetDevice(PetscDeviceContext, PetscDevice *);
PetscErrorCode PetscDeviceGetDeviceId(PetscDevice, PetscInt *);
PetscErrorCode PetscDeviceInitialize(PetscDeviceType);
PetscErrorCode PetscFunctionListAdd_Private(PetscFunctionList *, const char[], void (*)(void));
#define PetscFunctionListAdd(list, name, fptr) PetscFunctionListAdd_Private((list), (name), (PetscVoidFn)(fptr))
PetscErrorCode PetscFunctionListFind_Private(PetscFunctionList, const char[], void (**)(void));
#define PetscFunctionListFind(list, name, fptr) PetscFunctionListFind_Private((list), (name), (PetscVoidFn *)(fptr))
PetscErrorCode PetscRandomCreate(MPI_Comm, PetscRandom *);
PetscErrorCode PetscRandomDestroy(PetscRandom *);
PetscErrorCode PetscLayoutGetRanges(PetscLayout, const PetscInt *[]);

/* ---- IS / SF ---- */
PetscErrorCode ISCreateGeneral(MPI_Comm, PetscInt, const PetscInt[], PetscCopyMode, IS *);
PetscErrorCode ISCreateStride(MPI_Comm, PetscInt, PetscInt, PetscInt, IS *);
PetscErrorCode ISDestroy(IS *);
PetscErrorCode ISGetIndices(IS, const PetscInt *[]);
PetscErrorCode ISRestoreIndices(IS, const PetscInt *[]);
PetscErrorCode ISGetLocalSize(IS, PetscInt *);
PetscErrorCode ISGetSize(IS, PetscInt *);
PetscErrorCode ISLocalToGlobalMappingGetIndices(ISLocalToGlobalMapping, const PetscInt **);
PetscErrorCode ISLocalToGlobalMappingRestoreIndices(ISLocalToGlobalMapping, const PetscInt **);
PetscErrorCode ISLocalToGlobalMappingGetSize(ISLocalToGlobalMapping, PetscInt *);
PetscErrorCode PetscSFGetGraph(PetscSF, PetscInt *, PetscInt *, const PetscInt **, const PetscSFNode **);
PetscErrorCode PetscSFReduceBegin(PetscSF, MPI_Datatype, const void *, void *, MPI_Op);
PetscErrorCode PetscSFReduceEnd(PetscSF, MPI_Datatype, const void *, void *, MPI_Op);
PetscErrorCode PetscSFBcastBegin(PetscSF, MPI_Datatype, const void *, void *, MPI_Op);
PetscErrorCode PetscSFBcastEnd(PetscSF, MPI_Datatype, const void *, void *, MPI_Op);

/* ---- Vec ---- */
PetscErrorCode VecCreate(MPI_Comm, Vec *);
PetscErrorCode VecCreateSeq(MPI_Comm, PetscInt, Vec *);
PetscErrorCode VecCreateSeqWithArray(MPI_Comm, PetscInt, PetscInt, const PetscScalar[], Vec *);
PetscErrorCode VecCreateSeqHIPWithArray(MPI_Comm, PetscInt, PetscInt, const PetscScalar[], Vec *);
PetscErrorCode VecCreateMPIHIPWithArray(MPI_Comm, PetscInt, PetscInt, PetscInt, const PetscScalar[], Vec *);
PetscErrorCode VecDestroy(Vec *);
PetscErrorCode VecDuplicate(Vec, Vec *);
PetscErrorCode VecCopy(Vec, Vec);
PetscErrorCode VecSet(Vec, PetscScalar);
PetscErrorCode VecScale(Vec, PetscScalar);
PetscErrorCode VecAXPY(Vec, PetscScalar, Vec);
PetscErrorCode VecAYPX(Vec, PetscScalar, Vec);
PetscErrorCode VecWAXPY(Vec, PetscScalar, Vec, Vec);
PetscErrorCode VecDot(Vec, Vec, PetscScalar *);
PetscErrorCode VecNorm(Vec, NormType, PetscReal *);
PetscErrorCode VecPointwiseDivide(Vec, Vec, Vec);
PetscErrorCode VecPointwiseMult(Vec, Vec, Vec);
PetscErrorCode VecZeroEntries(Vec);
PetscErrorCode VecGetSize(Vec, PetscInt *);
PetscErrorCode VecGetLocalSize(Vec, PetscInt *);
PetscErrorCode VecGetArray(Vec, PetscScalar **);
PetscErrorCode VecRestoreArray(Vec, PetscScalar **);
PetscErrorCode VecGetArrayRead(Vec, const PetscScalar **);
PetscErrorCode VecRestoreArrayRead(Vec, const PetscScalar **);
PetscErrorCode VecGetArrayWrite(Vec, PetscScalar **);
PetscErrorCode VecRestoreArrayWrite(Vec, PetscScalar **);
PetscErrorCode VecHIPGetArray(Vec, PetscScalar **);
PetscErrorCode VecHIPRestoreArray(Vec, PetscScalar **);
PetscErrorCode VecHIPGetArrayRead(Vec, const PetscScalar **);
PetscErrorCode VecHIPRestoreArrayRead(Vec, const PetscScalar **);
PetscErrorCode VecHIPGetArrayWrite(Vec, PetscScalar **);
PetscErrorCode VecHIPRestoreArrayWrite(Vec, PetscScalar **);
PetscErrorCode VecHIPPlaceArray(Vec, const PetscScalar[]);
PetscErrorCode VecHIPResetArray(Vec);
PetscErrorCode VecSetType(Vec, VecType);
PetscErrorCode VecGetType(Vec, VecType *);
PetscErrorCode VecGetLocalVector(Vec, Vec);
PetscErrorCode VecRestoreLocalVector(Vec, Vec);
PetscErrorCode VecGetLocalVectorRead(Vec, Vec);
PetscErrorCode VecRestoreLocalVectorRead(Vec, Vec);
PetscErrorCode VecScatterBegin(VecScatter, Vec, Vec, InsertMode, ScatterMode);
PetscErrorCode VecScatterEnd(VecScatter, Vec, Vec, InsertMode, ScatterMode);
PetscErrorCode VecScatterDestroy(VecScatter *);

/* ---- Mat ---- */
PetscErrorCode MatCreate(MPI_Comm, Mat *);
PetscErrorCode MatDestroy(Mat *);
PetscErrorCode MatSetType(Mat, MatType);
PetscErrorCode MatGetType(Mat, MatType *);
PetscErrorCode MatSetVecType(Mat, VecType);
PetscErrorCode MatGetVecType(Mat, VecType *);
PetscErrorCode MatMult(Mat, Vec, Vec);
PetscErrorCode MatMultAdd(Mat, Vec, Vec, Vec);
PetscErrorCode MatMultTranspose(Mat, Vec, Vec);
PetscErrorCode MatMultTransposeAdd(Mat, Vec, Vec, Vec);
PetscErrorCode MatGetSize(Mat, PetscInt *, PetscInt *);
PetscErrorCode MatGetLocalSize(Mat, PetscInt *, PetscInt *);
PetscErrorCode MatGetBlockSize(Mat, PetscInt *);
PetscErrorCode MatGetOwnershipRange(Mat, PetscInt *, PetscInt *);
PetscErrorCode MatGetOwnershipRangeColumn(Mat, PetscInt *, PetscInt *);
PetscErrorCode MatGetOwnershipRanges(Mat, const PetscInt **);
PetscErrorCode MatGetOwnershipRangesColumn(Mat, const PetscInt **);
PetscErrorCode MatCreateVecs(Mat, Vec *, Vec *);
PetscErrorCode MatDuplicate(Mat, MatDuplicateOption, Mat *);
PetscErrorCode MatConvert(Mat, MatType, MatReuse, Mat *);
PetscErrorCode MatShift(Mat, PetscScalar);
PetscErrorCode MatScale(Mat, PetscScalar);
PetscErrorCode MatGetDiagonal(Mat, Vec);
PetscErrorCode MatAssemblyBegin(Mat, MatAssemblyType);
PetscErrorCode MatAssemblyEnd(Mat, MatAssemblyType);
PetscErrorCode MatSetOption(Mat, MatOption, PetscBool);
PetscErrorCode MatGetRowIJ(Mat, PetscInt, PetscBool, PetscBool, PetscInt *, const PetscInt *[], const PetscInt *[], PetscBool *);
PetscErrorCode MatRestoreRowIJ(Mat, PetscInt, PetscBool, PetscBool, PetscInt *, const PetscInt *[], const PetscInt *[], PetscBool *);
PetscErrorCode MatSeqAIJGetArrayRead(Mat, const PetscScalar **);
PetscErrorCode MatSeqAIJRestoreArrayRead(Mat, const PetscScalar **);
PetscErrorCode MatSeqAIJGetArray(Mat, PetscScalar **);
PetscErrorCode MatSeqAIJRestoreArray(Mat, PetscScalar **);
PetscErrorCode MatCreateSeqAIJWithArrays(MPI_Comm, PetscInt, PetscInt, PetscInt[], PetscInt[], PetscScalar[], Mat *);
PetscErrorCode MatCreateSeqDense(MPI_Comm, PetscInt, PetscInt, PetscScalar[], Mat *);
PetscErrorCode MatDenseGetArrayRead(Mat, const PetscScalar **);
PetscErrorCode MatDenseRestoreArrayRead(Mat, const PetscScalar **);
PetscErrorCode MatDenseGetArray(Mat, PetscScalar **);
PetscErrorCode MatDenseRestoreArray(Mat, PetscScalar **);
PetscErrorCode MatDenseGetLDA(Mat, PetscInt *);
PetscErrorCode MatDenseGetLocalMatrix(Mat, Mat *);
PetscErrorCode MatShellGetContext(Mat, void *);
PetscErrorCode MatShellSetContext(Mat, void *);
PetscErrorCode MatShellSetOperation(Mat, MatOperation, PetscErrorCodeFn *);
PetscErrorCode MatShellGetOperation(Mat, MatOperation, PetscErrorCodeFn **);
PetscErrorCode MatCreateShell(MPI_Comm, PetscInt, PetscInt, PetscInt, PetscInt, void *, Mat *);
PetscErrorCode MatCompositeGetMat(Mat, PetscInt, Mat *);
PetscErrorCode MatCompositeGetNumberMat(Mat, PetscInt *);
PetscErrorCode MatCompositeGetType(Mat, MatCompositeType *);
PetscErrorCode MatCreateRedundantMatrix(Mat, PetscInt, MPI_Comm, MatReuse, Mat *);
PetscErrorCode MatISGetLocalMat(Mat, Mat *);
PetscErrorCode MatISGetLocalToGlobalMapping(Mat, ISLocalToGlobalMapping *, ISLocalToGlobalMapping *);
PetscErrorCode MatGetLocalToGlobalMapping(Mat, ISLocalToGlobalMapping *, ISLocalToGlobalMapping *);
PetscErrorCode MatGetOrdering(Mat, MatOrderingType, IS *, IS *);
PetscErrorCode MatFactorInfoInitialize(MatFactorInfo *);
PetscErrorCode MatLUFactor(Mat, IS, IS, const MatFactorInfo *);
PetscErrorCode MatCholeskyFactor(Mat, IS, const MatFactorInfo *);
PetscErrorCode MatMatSolve(Mat, Mat, Mat);
PetscErrorCode MatSolve(Mat, Vec, Vec);
PetscErrorCode MatTranspose(Mat, MatReuse, Mat *);
PetscErrorCode MatCreateTranspose(Mat, Mat *);
PetscErrorCode MatTransposeGetMat(Mat, Mat *);
PetscErrorCode MatMatMult(Mat, Mat, MatReuse, PetscReal, Mat *);
PetscErrorCode MatMatTransposeMult(Mat, Mat, MatReuse, PetscReal, Mat *);
PetscErrorCode MatTransposeMatMult(Mat, Mat, MatReuse, PetscReal, Mat *);
PetscErrorCode MatGetNullSpace(Mat, MatNullSpace *);
PetscErrorCode MatSetNullSpace(Mat, MatNullSpace);
PetscErrorCode MatNullSpaceCreate(MPI_Comm, PetscBool, PetscInt, const Vec[], MatNullSpace *);
PetscErrorCode MatNullSpaceDestroy(MatNullSpace *);
PetscErrorCode MatIsSymmetricKnown(Mat, PetscBool *, PetscBool *);
PetscErrorCode MatView(Mat, PetscViewer);

/* ---- KSP / PC ---- */
PetscErrorCode KSPCreate(MPI_Comm, KSP *);
PetscErrorCode KSPDestroy(KSP *);
PetscErrorCode KSPSetType(KSP, KSPType);
PetscErrorCode KSPGetType(KSP, KSPType *);
PetscErrorCode KSPSolve(KSP, Vec, Vec);
PetscErrorCode KSPSetUp(KSP);
PetscErrorCode KSPGetPC(KSP, PC *);
PetscErrorCode KSPSetOperators(KSP, Mat, Mat);
PetscErrorCode KSPGetOperators(KSP, Mat *, Mat *);
PetscErrorCode KSPGetTolerances(KSP, PetscReal *, PetscReal *, PetscReal *, PetscInt *);
PetscErrorCode KSPSetTolerances(KSP, PetscReal, PetscReal, PetscReal, PetscInt);
PetscErrorCode KSPGetSolution(KSP, Vec *);
PetscErrorCode KSPGetRhs(KSP, Vec *);
PetscErrorCode KSPGetIterationNumber(KSP, PetscInt *);
PetscErrorCode KSPGetConvergedReason(KSP, KSPConvergedReason *);
PetscErrorCode KSPRegister(const char[], PetscErrorCode (*)(KSP));
PetscErrorCode KSPChebyshevEstEigGet(KSP, PetscReal *, PetscReal *);
PetscErrorCode PCCreate(MPI_Comm, PC *);
PetscErrorCode PCDestroy(PC *);
PetscErrorCode PCSetType(PC, PCType);
PetscErrorCode PCGetType(PC, PCType *);
PetscErrorCode PCSetUp(PC);
PetscErrorCode PCApply(PC, Vec, Vec);
PetscErrorCode PCGetOperators(PC, Mat *, Mat *);
PetscErrorCode PCSetOperators(PC, Mat, Mat);
PetscErrorCode PCRegister(const char[], PetscErrorCode (*)(PC));
PetscErrorCode PCMGGetLevels(PC, PetscInt *);
PetscErrorCode PCMGGetSmoother(PC, PetscInt, KSP *);
PetscErrorCode PCMGGetCoarseSolve(PC, KSP *);
PetscErrorCode PCMGGetInterpolation(PC, PetscInt, Mat *);
PetscErrorCode PCBJacobiGetSubKSP(PC, PetscInt *, PetscInt *, KSP *[]);
PetscErrorCode PCFactorGetMatrix(PC, Mat *);
PetscErrorCode PCFactorSetMatSolverType(PC, MatSolverType);
PetscErrorCode MatRegister(const char[], PetscErrorCode (*)(Mat));
