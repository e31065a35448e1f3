// Class-shared explicit local dual operators (fshared.hip), the PMH_FX_CLASS storage of pmh_fexplicit (internal)
#pragma once
#include "feti_internal.h"

struct fx_shared;
int       fxs_create(pmh_gluing B, pmh_blockdiag K, const int *block_class, int sym, fx_shared **out, const int *extra_ptr = nullptr, const int *extra_rel = nullptr); // sym 1: lower block-triangle in 16 x 16 tiles (PMH_FX_CLASS_SYM), 2: orbit representatives' rows (PMH_FX_CLASS_ORBIT)
void      fxs_destroy(fx_shared *S);
int       fxs_set_stripe(fx_shared *S, int rank, int size);
int       fxs_class_union(fx_shared *S, int c, int *n_c, int *urel_out);
int       fxs_set_symmetry(fx_shared *S, int c, int nsym, const int *posmap, const signed char *sign);
long long fxs_dense_bytes(fx_shared *S);
double    fxs_apply_bytes(fx_shared *S);
double    fxs_apply_flops(fx_shared *S);
void      fxs_apply_flops_detail(fx_shared *S, double *issued, double *dense);
int       fxs_assemble(fx_shared *S, pmh_matinv solver, int nslots, const int *slot_class, double rtol, int max_it, long long *n_solves);
int       fxs_apply(fx_shared *S, const double *lambda, double *y);
int       fxs_dense(fx_shared *S);
int       fxs_stages(fx_shared *S, pmh_csr *gather, double **mid_in, pmh_csr *scatter, const double **mid_out);
int       fxs_mid(fx_shared *S);
long long fxs_multivector_length(fx_shared *S);
double   *fxs_X(fx_shared *S);
double   *fxs_Y(fx_shared *S);
int       fxs_fill_pattern(fx_shared *S, int byte);
int       fxs_get_block(fx_shared *S, int b, int n, const int *gamma, double *out_host);
int       fxs_timing_enable(fx_shared *S, int max_launches);
int       fxs_timing_get(fx_shared *S, int *launches, double *total_ms, double *first_kernel_ms);
