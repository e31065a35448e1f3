// Options front end of the QPS classes: the reference configures its solvers from PETSc's options database
// (QPSSetFromOptions src/qps/interface/qps.c:860-900, QPSSetFromOptions_MPGP src/qps/impls/mpgp/mpgp.c:712-745,
// QPSSetFromOptions_SMALXE src/qps/impls/smalxe/smalxe.c:696-766).  This is the same key set over an option STRING
// (what a PETSc program gets on its command line or from ~/.permonrc), so the reference's own TEST blocks
// ("-qps_mpgp_expansion_type gf -qps_mpgp_expansion_length_type opt", src/tutorials/ex1.c:175) drive this library verbatim.
// Host code only; the argument checks of the reference's setters are reproduced (same conditions, same wording).
#include <cctype>
#include <string>

#include "pmh_internal.h"

namespace {
struct Token {
  std::string key, val;
  bool        has_val;
};

// PETSc's rule (PetscOptionsInsertString / PetscOptionsValidKey): a token starting with '-' followed by a non-digit is a key;
// the next token is its value unless it is a key itself ("-1e-3" is a value).
bool is_key(const std::string &t) { return t.size() >= 2 && t[0] == '-' && !(std::isdigit((unsigned char)t[1]) || t[1] == '.'); }

int tokenize(const char *s, std::vector<Token> &out)
{
  std::vector<std::string> w;
  for (const char *p = s; *p;) {
    while (*p && std::isspace((unsigned char)*p)) p++;
    if (!*p) break;
    const char *q = p;
    while (*q && !std::isspace((unsigned char)*q)) q++;
    w.emplace_back(p, q);
    p = q;
  }
  for (size_t i = 0; i < w.size(); i++) {
    if (!is_key(w[i])) return pmh_set_error(PMH_ERR_ARG, "options: expected an option key (-name), got \"%s\"", w[i].c_str());
    Token t;
    t.key     = w[i].substr(1);
    t.has_val = (i + 1 < w.size()) && !is_key(w[i + 1]);
    if (t.has_val) t.val = w[++i];
    out.push_back(t);
  }
  return PMH_SUCCESS;
}

std::string lower(std::string s)
{
  for (auto &c : s) c = (char)std::tolower((unsigned char)c);
  return s;
}

int get_real(const Token &t, double *v)
{
  if (!t.has_val) return pmh_set_error(PMH_ERR_ARG, "options: -%s needs a value", t.key.c_str());
  const std::string l = lower(t.val);
  if (l == "petsc_decide" || l == "decide") { // PetscOptionsStringToReal accepts these names
    *v = PMH_DECIDE;
    return PMH_SUCCESS;
  }
  if (l == "petsc_default" || l == "default") {
    *v = -2.0;
    return PMH_SUCCESS;
  }
  char *end = nullptr;
  *v        = strtod(t.val.c_str(), &end);
  if (end == t.val.c_str() || *end) return pmh_set_error(PMH_ERR_ARG, "options: -%s: \"%s\" is not a number", t.key.c_str(), t.val.c_str());
  return PMH_SUCCESS;
}

int get_int(const Token &t, int *v)
{
  if (!t.has_val) return pmh_set_error(PMH_ERR_ARG, "options: -%s needs a value", t.key.c_str());
  char *end = nullptr;
  long  l   = strtol(t.val.c_str(), &end, 10);
  if (end == t.val.c_str() || *end) return pmh_set_error(PMH_ERR_ARG, "options: -%s: \"%s\" is not an integer", t.key.c_str(), t.val.c_str());
  *v = (int)l;
  return PMH_SUCCESS;
}

// PetscOptionsStringToBool: no value = true; true/yes/1/on, false/no/0/off (any case)
int get_bool(const Token &t, int *v)
{
  if (!t.has_val) {
    *v = 1;
    return PMH_SUCCESS;
  }
  const std::string l = lower(t.val);
  if (l == "true" || l == "yes" || l == "1" || l == "on") *v = 1;
  else if (l == "false" || l == "no" || l == "0" || l == "off") *v = 0;
  else return pmh_set_error(PMH_ERR_ARG, "options: -%s: unknown logical value \"%s\"", t.key.c_str(), t.val.c_str());
  return PMH_SUCCESS;
}

// PetscOptionsEnum: case-insensitive name out of the list
int get_enum(const Token &t, const char *const *names, int n, int *v)
{
  if (!t.has_val) return pmh_set_error(PMH_ERR_ARG, "options: -%s needs a value", t.key.c_str());
  const std::string l = lower(t.val);
  for (int i = 0; i < n; i++)
    if (l == names[i]) {
      *v = i;
      return PMH_SUCCESS;
    }
  return pmh_set_error(PMH_ERR_ARG, "options: -%s: unknown value \"%s\"", t.key.c_str(), t.val.c_str());
}

const char *const kExpTypes[]    = {"std", "projcg", "gf", "g", "gfgr", "ggr"}; // QPSMPGPExpansionTypes mpgp.c:3
const char *const kExpLenTypes[] = {"fixed", "opt", "optapprox", "bb"};         // QPSMPGPExpansionLengthTypes mpgp.c:4

// one key of QPSSetFromOptions_MPGP; returns 1 if the key was consumed, 0 if it is not an MPGP key, < 0 on error
int mpgp_key(const Token &t, const std::string &k, pmh_mpgp_opts *m, int *alpha_bits)
{
  double r;
  int    i;
#define ERR(call) \
  do { \
    if (call) return -1; \
  } while (0)
  if (k == "qps_mpgp_alpha_direct") {
    ERR(get_bool(t, &m->alpha_direct));
    *alpha_bits |= 2;
  } else if (k == "qps_mpgp_alpha") {
    ERR(get_real(t, &m->alpha_user));
    *alpha_bits |= 1;
  } else if (k == "qps_mpgp_gamma") {
    ERR(get_real(t, &m->gamma));
  } else if (k == "qps_mpgp_maxeig") {
    ERR(get_real(t, &r));
    if (!(r >= 0 || r == PMH_DECIDE)) return pmh_set_error(PMH_ERR_ARG, "Argument must be nonnegative"), -1; // mpgp.c:995
    m->maxeig = r;
  } else if (k == "qps_mpgp_maxeig_tol") {
    ERR(get_real(t, &m->maxeig_tol));
  } else if (k == "qps_mpgp_maxeig_iter") {
    ERR(get_int(t, &i));
    if (!(i > 1)) return pmh_set_error(PMH_ERR_ARG, "Argument must be > 1"), -1; // mpgp.c:1088
    m->maxeig_iter = i;
  } else if (k == "qps_mpgp_btol") {
    ERR(get_real(t, &r)); // read into mpgp->btol, which no code path uses (mpgp.c:736): accepted, no effect
  } else if (k == "qps_mpgp_bound_chop_tol") {
    ERR(get_real(t, &m->bchop_tol));
  } else if (k == "qps_mpgp_expansion_type") {
    ERR(get_enum(t, kExpTypes, 6, &m->exptype));
  } else if (k == "qps_mpgp_expansion_length_type") {
    ERR(get_enum(t, kExpLenTypes, 4, &m->explengthtype));
  } else if (k == "qps_mpgp_alpha_reset") {
    ERR(get_bool(t, &m->resetalpha));
  } else if (k == "qps_mpgp_fallback") {
    ERR(get_bool(t, &m->fallback));
  } else if (k == "qps_mpgp_fallback2") {
    ERR(get_bool(t, &m->fallback2));
  } else {
    return 0;
  }
  return 1;
}

// QPSSetTolerances qps.c:905-930 (the keys -qps_rtol/-qps_atol/-qps_divtol/-qps_max_it of QPSSetFromOptions)
int tol_key(const Token &t, const std::string &k, double *rtol, double *atol, double *divtol, int *max_it, int *max_it_set)
{
  double r;
  int    i;
  if (k == "qps_rtol") {
    ERR(get_real(t, &r));
    if (r != -2.0) {
      if (!(r >= 0.0 && 1.0 > r)) return pmh_set_error(PMH_ERR_ARG, "Relative tolerance %g must be non-negative and less than 1.0", r), -1;
      *rtol = r;
    }
  } else if (k == "qps_atol") {
    ERR(get_real(t, &r));
    if (r != -2.0) {
      if (!(r >= 0.0)) return pmh_set_error(PMH_ERR_ARG, "Absolute tolerance %g must be non-negative", r), -1;
      *atol = r;
    }
  } else if (k == "qps_divtol") {
    ERR(get_real(t, &r));
    if (r != -2.0) {
      if (!(r >= 0.0)) return pmh_set_error(PMH_ERR_ARG, "Divergence tolerance %g must be larger than 1.0", r), -1;
      *divtol = r;
    }
  } else if (k == "qps_max_it") {
    ERR(get_int(t, &i));
    if (i != -2) {
      if (!(i >= 0)) return pmh_set_error(PMH_ERR_ARG, "Maximum number of iterations %d must be non-negative", i), -1;
      *max_it = i;
      if (max_it_set) *max_it_set = 1;
    }
  } else {
    return 0;
  }
  return 1;
}

int smalxe_key(const Token &t, const std::string &k, pmh_smalxe_opts *s)
{
  double r;
  int    i, b;
  if (k == "qps_smalxe_maxeig") {
    ERR(get_real(t, &r));
    if (!(r > 0 || r == PMH_DECIDE)) return pmh_set_error(PMH_ERR_ARG, "Argument must be positive"), -1; // smalxe.c:1240
    s->maxeig = r;
  } else if (k == "qps_smalxe_maxeig_tol") {
    ERR(get_real(t, &s->maxeig_tol));
  } else if (k == "qps_smalxe_maxeig_iter") {
    ERR(get_int(t, &i));
    if (!(i > 1)) return pmh_set_error(PMH_ERR_ARG, "Argument must be > 1"), -1; // smalxe.c:1407
    s->maxeig_iter = i;
  } else if (k == "qps_smalxe_maxeig_inject") {
    ERR(get_bool(t, &b));
    s->inject_maxeig = b, s->inject_maxeig_set = 1;
  } else if (k == "qps_smalxe_eta_direct") {
    ERR(get_bool(t, &s->eta_direct));
  } else if (k == "qps_smalxe_eta") {
    ERR(get_real(t, &r));
    if (!(r > 0)) return pmh_set_error(PMH_ERR_ARG, "Argument must be positive"), -1; // smalxe.c:1290
    s->eta_user = r;
  } else if (k == "qps_smalxe_rho_direct") {
    ERR(get_bool(t, &s->rho_direct));
  } else if (k == "qps_smalxe_rho") {
    ERR(get_real(t, &r));
    if (!(r > 0)) return pmh_set_error(PMH_ERR_ARG, "Argument must be positive"), -1; // smalxe.c:1315
    s->rho_user = r;
  } else if (k == "qps_smalxe_rho_update") {
    ERR(get_real(t, &r));
    if (!(r >= 1)) return pmh_set_error(PMH_ERR_ARG, "Argument must be >= 1"), -1; // smalxe.c:1361
    s->rho_update = r;
  } else if (k == "qps_smalxe_rho_update_late") {
    ERR(get_real(t, &r));
    if (!(r >= 1)) return pmh_set_error(PMH_ERR_ARG, "Argument must be >= 1"), -1; // smalxe.c:1384
    s->rho_update_late = r;
  } else if (k == "qps_smalxe_M1_direct") {
    ERR(get_bool(t, &s->M1_direct));
  } else if (k == "qps_smalxe_M1") {
    ERR(get_real(t, &r));
    if (!(r > 0)) return pmh_set_error(PMH_ERR_ARG, "Argument must be positive"), -1; // smalxe.c:1265
    s->M1_user = r;
  } else if (k == "qps_smalxe_M1_update") {
    ERR(get_real(t, &s->M1_update));
  } else if (k == "qps_smalxe_rtol_E") {
    ERR(get_real(t, &s->rtol_E));
  } else if (k == "qps_smalxe_inner_iter_min") {
    ERR(get_int(t, &s->inner_iter_min));
  } else if (k == "qps_smalxe_inner_no_gtol_stop") {
    ERR(get_int(t, &s->inner_no_gtol_stop));
  } else if (k == "qps_smalxe_update_threshold") {
    ERR(get_real(t, &s->update_threshold));
  } else if (k == "qps_smalxe_norm_update_lag") { // smalxe.c:754-762
    ERR(get_bool(t, &s->lag_enabled));
  } else if (k == "qps_smalxe_norm_update_lag_offset") {
    ERR(get_int(t, &s->lag_offset));
  } else if (k == "qps_smalxe_norm_update_lag_start") {
    ERR(get_int(t, &s->lag_start));
  } else if (k == "qps_smalxe_norm_update_lag_step") {
    ERR(get_int(t, &s->lag_step));
  } else if (k == "qps_smalxe_norm_update_lag_end") {
    ERR(get_int(t, &s->lag_end));
  } else if (k == "qps_smalxe_norm_update_lag_lower") {
    ERR(get_real(t, &s->lag_lower));
  } else if (k == "qps_smalxe_norm_update_lag_upper") {
    ERR(get_real(t, &s->lag_upper));
  } else if (k == "qps_smalxe_knoll") { // smalxe.c:764
    ERR(get_bool(t, &s->knoll));
  } else {
    return 0;
  }
  return 1;
#undef ERR
}
} // namespace

// options: PETSc-style option string.  prefix: the QPS object's options prefix ("" for the top solver; the reference appends
// "smalxe_" for SMALXE's inner MPGP, smalxe.c:500-502, so its keys read -smalxe_qps_mpgp_gamma ...).
// q / m / s: option structs already holding their defaults (pmh_*_default_opts); s may be NULL when the caller has no SMALXE.
// Keys of the inner solver are applied to s->inner.  unknown (optional, unknown_cap bytes) receives the space-separated keys
// nobody consumed (PETSc's -options_left report); they are not an error, as in PETSc.
extern "C" int pmh_qps_set_from_options(const char *options, const char *prefix, pmh_qps_opts *q, pmh_mpgp_opts *m, pmh_smalxe_opts *s, char *unknown, int unknown_cap)
{
  PMH_ARG(options && q && m);
  std::vector<Token> toks;
  PMH_CHK(tokenize(options, toks));
  const std::string pre = prefix ? prefix : "";
  const std::string inner_pre = pre + "smalxe_";
  std::string       left;
  int               alpha_bits = 0, inner_alpha_bits = 0;
  for (const Token &t : toks) {
    int rc = 0;
    if (t.key.compare(0, pre.size(), pre) == 0) {
      const std::string k = t.key.substr(pre.size());
      if (k == "qps_type") {
        if (!t.has_val) return pmh_set_error(PMH_ERR_ARG, "options: -%s needs a value", t.key.c_str());
        const std::string v = lower(t.val);
        if (v != "mpgp" && v != "smalxe" && v != "pcpg" && v != "ksp")
          return pmh_set_error(PMH_ERR_SUP, "Unable to find requested QPS type %s", t.val.c_str()); // QPSSetType qps.c:394 (tao: out of scope)
        snprintf(q->type, sizeof(q->type), "%s", v.c_str());
        rc = 1;
      } else if (k == "qps_monitor") {
        rc = get_bool(t, &q->monitor) ? -1 : 1;
      } else if (k == "qps_monitor_cost") {
        rc = get_bool(t, &q->monitor_cost) ? -1 : 1;
      } else if (k == "qps_monitor_cancel") {
        int b;
        rc = get_bool(t, &b) ? -1 : 1;
        if (rc == 1 && b) q->monitor = q->monitor_cost = 0;
      } else if (k == "qps_view_convergence") {
        q->view_convergence = 1, rc = 1; // PetscOptionsName: presence only
      } else if (k == "qps_view") {
        q->view = 1, rc = 1;
      } else if (k == "qps_auto_post_solve") {
        rc = get_bool(t, &q->auto_post_solve) ? -1 : 1;
      } else {
        rc = tol_key(t, k, &q->rtol, &q->atol, &q->divtol, &q->max_it, &q->max_it_set);
        if (!rc) rc = mpgp_key(t, k, m, &alpha_bits);
        if (!rc && s) rc = smalxe_key(t, k, s);
      }
    }
    if (!rc && s && t.key.compare(0, inner_pre.size(), inner_pre) == 0) { // the inner MPGP of SMALXE
      const std::string k = t.key.substr(inner_pre.size());
      rc = tol_key(t, k, &s->inner.rtol, &s->inner.atol, &s->inner.divtol, &s->inner.max_it, nullptr);
      if (!rc) rc = mpgp_key(t, k, &s->inner, &inner_alpha_bits);
    }
    if (rc < 0) return PMH_ERR_ARG;
    if (!rc) left += (left.empty() ? "-" : " -") + t.key;
  }
  // QPSMPGPSetAlpha(qps, alpha, alpha_direct) is called when either key is present, with alpha_direct = PETSC_FALSE unless
  // its own key is given (mpgp.c:723-726): -qps_mpgp_alpha alone means "multiple of 1/lambda_max"
  if (alpha_bits == 1) m->alpha_direct = 0;
  if (s && inner_alpha_bits == 1) s->inner.alpha_direct = 0;
  if (m->fallback2) m->fallback = 0; // mpgp.c:743
  if (s && s->inner.fallback2) s->inner.fallback = 0;
  if (unknown && unknown_cap > 0) snprintf(unknown, (size_t)unknown_cap, "%s", left.c_str());
  return PMH_SUCCESS;
}

// QPSCreate defaults (qps.c:73-76); type "" = QPSSetDefaultType decides at set-up (qps.c:422-455)
extern "C" int pmh_qps_default_opts(pmh_qps_opts *q)
{
  PMH_ARG(q);
  memset(q, 0, sizeof(*q));
  q->rtol = 1e-5, q->atol = 1e-50, q->divtol = 1e4, q->max_it = 10000;
  q->auto_post_solve = 1;
  return PMH_SUCCESS;
}

// Options of the FETI driver: the keys KSPFETI's chain reads from the options database -- QPFetiSetUp (qpfeti.c:340-341:
// -feti_gluing_type, -feti_gluing_exclude_dirichlet), QPFetiGetBgtSF (:757-758: -SCALE_ON), QPTFromOptions (qptransform.c:2220-2231:
// -regularize), QPTDualize (:1019: -qpt_dualize_Kplus_mp), PCDUAL with the dual QP's prefix (pcdual.c:170: -dual_pc_dual_type), the
// outer QPS tolerances (-qps_rtol ...) and the inner KSP of MATINV (-dual_mat_inv_ksp_rtol / _max_it) -- parsed into
// pmh_kspfeti_opts, so the reference's ex71 TEST blocks ("-feti_gluing_type orth", "-qps_rtol 1e-6 -dual_pc_dual_type lumped")
// configure pmh_kspfeti_solve verbatim.  Unknown keys are returned, not rejected (PETSc's -options_left).
extern "C" int pmh_kspfeti_set_from_options(const char *options, pmh_kspfeti_opts *o, char *unknown, int unknown_cap)
{
  PMH_ARG(options && o);
  std::vector<Token> toks;
  PMH_CHK(tokenize(options, toks));
  static const char *const gtypes[] = {"nonred", "full", "orth"};    // FetiGluingTypes
  static const char *const pctypes[] = {"none", "lumped"};           // PCDualTypes (pcdual.c)
  static const char *const orthtypes[] = {"none", "gs", "gslingen", "cholesky", "implicit", "inexact"}; // MatOrthTypes (permonmatorth.c:6)
  std::string left;
  int         inner_alpha_bits = 0;
  // resolved AFTER the loop: -qpt_dualize_Kplus_mp wins over -qpt_dualize_Kplus_left in whatever order they come (qptransform.c:1018-1019 reads _left only if
  // !true_mp)
  int         mp_given = 0, left_given = -1;
  for (const Token &t : toks) {
    const std::string &k = t.key;
    int                rc = 1, b = 0;
    if (k == "feti_gluing_type") rc = get_enum(t, gtypes, 3, &o->gluing_type) ? -1 : 1;
    else if (k == "feti_gluing_exclude_dirichlet") rc = get_bool(t, &o->exclude_dirichlet) ? -1 : 1;
    else if (k == "SCALE_ON") rc = get_bool(t, &o->scale) ? -1 : 1;
    // (without effect while kplus_left is on, as in the reference once it has computed the kernel)
    else if (k == "regularize") rc = get_bool(t, &o->regularize) ? -1 : 1;
    else if (k == "qpt_dualize_Kplus_mp") {
      rc = get_bool(t, &b) ? -1 : 1;
      if (rc == 1) mp_given = b; // the Moore-Penrose wrapping is this library's -regularize 0 path
    } else if (k == "dual_pc_dual_type") rc = get_enum(t, pctypes, 2, &o->lumped_pc) ? -1 : 1;
    else if (k == "dual_mat_inv_ksp_rtol") rc = get_real(t, &o->kplus_rtol) ? -1 : 1;
    else if (k == "dual_mat_inv_ksp_max_it") rc = get_int(t, &o->kplus_max_it) ? -1 : 1;
    else if (k == "dual_mat_inv_pc_type") { // the PC of MATINV's inner KSP (MatInvGetKSP; PETSc's PCType names): jacobi | gamg (mg: the same algebraic hierarchy)
      static const char *const pcs[] = {"jacobi", "gamg", "mg"};
      int v = 0;
      rc = get_enum(t, pcs, 3, &v) ? -1 : 1;
      if (rc == 1) o->kplus_pc = v ? 1 : 0;
    }
    else if (k == "feti") rc = get_bool(t, &b) ? -1 : 1; // the combination this driver always performs
    else if (k == "qp_chain_view_kkt") rc = get_bool(t, &o->view_kkt) ? -1 : 1;
    else if (k == "qps_view_convergence") rc = get_bool(t, &o->view_convergence) ? -1 : 1;
    else if (k == "qpt_matis_to_diag_norm") rc = get_bool(t, &o->matis_to_diag_norm) ? -1 : 1;
    else if (k == "qpt_dualize_Kplus_left") { // QPTDualize qptransform.c:1018; implies -regularize 0 as the reference's own switch does (:1003-1005)
      rc = get_bool(t, &b) ? -1 : 1;
      if (rc == 1) left_given = b;
    } else if (k == "project") rc = get_bool(t, &o->project) ? -1 : 1;                     // QPTFromOptions qptransform.c:2228
    else if (k == "dual_qp_E_orth_type") {                                                  // QPTOrthonormalizeEqFromOptions on the dual QP (prefix dual_)
      rc = get_enum(t, orthtypes, 6, &o->E_orth_type) ? -1 : 1;
      if (rc == 1 && o->E_orth_type == 5) return pmh_set_error(PMH_ERR_SUP, "options: -dual_qp_E_orth_type %s is not built (none, gs, gslingen, cholesky, implicit)", t.val.c_str());
    } else {
      rc = tol_key(t, k, &o->rtol, &o->atol, &o->divtol, &o->max_it, &o->max_it_set);
      if (!rc) rc = smalxe_key(t, k, &o->smalxe);                                           // -qps_smalxe_*: the solver of -project 0
      if (!rc && k.compare(0, 7, "smalxe_") == 0) {                                         // its inner solver (smalxe.c:500-502)
        const std::string ki = k.substr(7);
        rc = tol_key(t, ki, &o->smalxe.inner.rtol, &o->smalxe.inner.atol, &o->smalxe.inner.divtol, &o->smalxe.inner.max_it, nullptr);
        if (!rc) rc = mpgp_key(t, ki, &o->smalxe.inner, &inner_alpha_bits);
      }
    }
    if (rc < 0) return PMH_ERR_ARG;
    if (!rc) left += (left.empty() ? "-" : " -") + k;
  }
  if (inner_alpha_bits == 1) o->smalxe.inner.alpha_direct = 0;
  if (left_given >= 0) {
    o->kplus_left = left_given;
    if (left_given) o->regularize = 0; // implies -regularize 0 as the reference's own switch does (qptransform.c:1003-1005)
  }
  if (mp_given) o->regularize = 0, o->kplus_left = 0;
  if (unknown && unknown_cap > 0) snprintf(unknown, (size_t)unknown_cap, "%s", left.c_str());
  return PMH_SUCCESS;
}
