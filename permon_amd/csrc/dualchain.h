// Fused dual-space chain of the penalised, projected FETI operator (dualchain.hip; internal)
#pragma once
#include "pmh_internal.h"

struct pmh_dualchain_s;
typedef pmh_dualchain_s *pmh_dualchain;

// F must offer pmh_op_s::stages; PMH_EPI_UNSUPPORTED (no error recorded) where the chain does not apply
int  pmh_dc_create(pmh_qppf pf, pmh_op F, pmh_dualchain *out);
void pmh_dc_destroy(pmh_dualchain dc);
// y = rho Q x + P F P x (+ the vector phase of epi in the last kernel); epi may be nullptr
int  pmh_dc_apply(pmh_dualchain dc, const double *x, double *y, double rho, const pmh_vec_epi *epi);
int  pmh_dc_emit_begin(pmh_dualchain dc, const double *x, const double *p, pmh_emit_args *ea);
void pmh_dc_invalidate(pmh_dualchain dc);
// SMALXE's ||B u||: every emission of the iterate also leaves T G0 u in Gu (m doubles) and its squared norm in d_scal / h_scal[slot]
int  pmh_dc_set_norm_target(pmh_dualchain dc, double *Gu, int slot);
bool pmh_dc_norm_ready(pmh_dualchain dc, const double *u);
// launches of the last application (tests, bench)
int  pmh_dc_last_launches(pmh_dualchain dc);
