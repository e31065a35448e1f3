// Orbit storage (PMH_FX_CLASS_ORBIT) of the class-shared explicit local dual operators: the PLAN of the GEMM -- row tiles, column lists, k segments, split-K pieces, the
// work items of the launches and the tables of the finishing kernel.  Host code only (uploads its tables); the kernels are in fshared_kernels.h, the launches in fshared.hip.
#include <algorithm>
#include <chrono>
#include <map>
#include <cmath>

#include "fshared_types.h"

// 16-row instruction tiles (v_mfma_f64_16x16x4): the workgroup tile that pads least among 144, 128, 112, 96, 80 (ties: the larger tile).  (The 4x4x4_4b kernels of rounds
// 2-3, row tiles 128 ... 96, went at the end of round 6: docs/LAB_NOTEBOOK.md.)
int fxo_row_tile(int M)
{
  if (const char *e = getenv("PMH_FXO_TM")) { // tests: every row tile
    const int v = atoi(e);
    if (v == 144 || v == 128 || v == 112 || v == 96 || v == 80) return v;
  }
  int best = 144, pad = (M + 143) / 144 * 144;
  for (int tm : {128, 112, 96, 80})
    if ((M + tm - 1) / tm * tm < pad) pad = (M + tm - 1) / tm * tm, best = tm;
  return best;
}

// host helper (no device): the row tile fxo_prepare picks for a class with M orbit representatives and the padded row count of its GEMM
extern "C" int pmh_fexplicit_orbit_row_tile(int M, int *tm, int *Mp)
{
  PMH_ARG(M >= 1);
  const int t = fxo_row_tile(M);
  if (tm) *tm = t;
  if (Mp) *Mp = (M + t - 1) / t * t;
  return PMH_SUCCESS;
}

// ---- orbit storage: plan (after the symmetries and the stripe are known) ----------------------------------------------------------------------
struct fxo_unit { // a (group, row tile, k segment): its columns (a sub-list of the tile's), its chunks on this rank, its splits and partial tiles
  int       g = 0, mt = 0, seg = 0, coff = 0, nct = 0, listed = 0, lutoff = 0, kc0 = 0, kc1 = 0, S = 0;
  long long cbase = 0;
};
struct fxo_plan {
  std::vector<fxo_unit> units;
  std::vector<int>      coltab, fintab, lut, segc0;
};

int fxo_prepare(fx_shared *S)
{
  if (S->fxo_ready) return PMH_SUCCESS;
  pmh_ctx   ctx = S->ctx;
  long long atot = 0;
  std::vector<fxo_plan> plan; // per class with touched dofs, in class order
  std::vector<int>      tab_of(S->ncls, -1);
  for (int c = 0; c < S->ncls; c++) {
    fxs_class &C = S->C[c];
    if (C.nc == 0) continue;
    tab_of[c] = (int)plan.size();
    if (C.nsym < 1) return pmh_set_error(PMH_ERR_STATE, "PMH_FX_CLASS_ORBIT: block class %d has no symmetries (pmh_fexplicit_set_class_symmetry / _set_box_symmetry before the assembly)", c);
    // orbits of the rows: representative and operation of every row; rows fixed by several operations keep the first
    C.rep_of.assign((size_t)C.nc, -1), C.op_of.assign((size_t)C.nc, 0), C.reps.clear();
    for (int p = 0; p < C.nc; p++) {
      if (C.rep_of[p] >= 0) continue;
      C.reps.push_back(p);
      for (int g = 0; g < C.nsym; g++) {
        const int r = C.h_posmap[(size_t)g * C.nc + p];
        if (C.rep_of[r] < 0) C.rep_of[r] = p, C.op_of[r] = g;
      }
    }
    C.M_all = (int)C.reps.size();
    // several GPUs: every rank keeps ALL representatives' rows (0.19 GB for configs[2]; it solves for them itself) and multiplies a contiguous share of
    // the k range (the columns of W): full tiles at every N, and the partial Y are summed by the all-reduce that ends B Y anyway
    C.m0 = 0, C.m1 = C.M_all;
    const int M = C.m1 - C.m0;
    C.tm   = fxo_row_tile(M);
    C.tnw  = 0;
    if (C.S == 1 && C.nsym * C.S <= 48) {
      // a class of ONE block lists at most 48 columns: the 64-wide tile multiplies a quarter of zeros.  48 columns x (4 waves x NI x 16 rows): 192 rows unless
      // fewer pad less
      C.tnw = 48, C.tm = 192;
      for (int tm : {128, 64})
        if ((M + tm - 1) / tm * tm < (M + C.tm - 1) / C.tm * C.tm) C.tm = tm;
    }
    C.Mp   = std::max(1, (M + C.tm - 1) / C.tm) * C.tm;
    C.nsymp = (C.nsym + FXO_TN / 8 - 1) / (FXO_TN / 8) * (FXO_TN / 8);
    std::vector<int>         reppos((size_t)C.Mp, 0);
    std::vector<signed char> use((size_t)C.Mp * C.nsymp, 0), use_h((size_t)M * C.nsym, 0);
    for (int pl = 0; pl < M; pl++) {
      const int p = C.reps[C.m0 + pl];
      for (int g = 0; g < C.nsym; g++) {
        const int r = C.h_posmap[(size_t)g * C.nc + p];
        if (C.rep_of[r] == p && C.op_of[r] == g) use_h[(size_t)pl * C.nsym + g] = C.h_sign[(size_t)g * C.nc + p];
      }
    }
    // need pattern of a representative: bit ((group * nsym + g) * 8 + slot) = row g p is owned by (p, g) and block (group, slot) touches it.  The rows of A
    // follow the patterns (the widest first), so that a row tile holds few patterns and its column list stays short: a face-interior representative of a 2 x 2
    // x 2 decomposition needs 224 or 256 of the 384 columns
    const bool   prune = !getenv("PMH_FXO_NO_PRUNE") && !C.tmask.empty();
    const size_t nbits = (size_t)C.ngroups * C.nsym * FXS_S, nw = (nbits + 63) / 64;
    std::vector<unsigned long long> pat((size_t)M * nw, 0ULL);
    std::vector<int>                cnt((size_t)M, 0), rowrep((size_t)M);
    for (int pl = 0; pl < M; pl++) {
      const int p = C.reps[C.m0 + pl];
      for (int gr = 0; gr < C.ngroups; gr++)
        for (int g = 0; g < C.nsym; g++) {
          if (!use_h[(size_t)pl * C.nsym + g]) continue;
          const int r = C.h_posmap[(size_t)g * C.nc + p];
          for (int sl = 0; sl < FXS_S; sl++)
            if (!prune || C.tmask[((size_t)gr * C.nc + r) * FXS_S + sl]) {
              const size_t b = ((size_t)gr * C.nsym + g) * FXS_S + sl;
              pat[(size_t)pl * nw + b / 64] |= 1ULL << (b % 64), cnt[pl]++;
            }
        }
      rowrep[pl] = pl;
    }
    if (prune) {
      // rows with the same pattern together; the RARE patterns first (representatives on the cube's edges and corners need other columns than the face-interior
      // ones: they share the first row tile, whose list is the full one anyway), then the common ones, the wider first
      auto less_pat = [&](int a, int b) {
        return std::lexicographical_compare(pat.begin() + (size_t)a * nw, pat.begin() + (size_t)(a + 1) * nw, pat.begin() + (size_t)b * nw, pat.begin() + (size_t)(b + 1) * nw);
      };
      std::stable_sort(rowrep.begin(), rowrep.end(), less_pat);
      std::vector<int> gsize((size_t)M, 0); // size of the pattern group a representative belongs to
      for (int i = 0; i < M;) {
        int j = i + 1;
        while (j < M && !less_pat(rowrep[i], rowrep[j]) && !less_pat(rowrep[j], rowrep[i])) j++;
        for (int k = i; k < j; k++) gsize[rowrep[k]] = j - i;
        i = j;
      }
      std::stable_sort(rowrep.begin(), rowrep.end(), [&](int a, int b) {
        if (gsize[a] != gsize[b]) return gsize[a] < gsize[b];
        if (cnt[a] != cnt[b]) return cnt[a] > cnt[b];
        return less_pat(a, b);
      });
    }
    C.reprow.assign((size_t)M, 0);
    for (int row = 0; row < M; row++) {
      const int pl = rowrep[row];
      C.reprow[pl] = row, reppos[row] = C.reps[C.m0 + pl];
      for (int g = 0; g < C.nsym; g++) use[(size_t)row * C.nsymp + g] = use_h[(size_t)pl * C.nsym + g];
    }
    // column lists per (group, row tile): the columns some row of the tile needs (what k_fxo_fin walks)
    const int        ntile = C.Mp / C.tm, ncode = C.nsym * FXS_S, cw = (ncode + 63) / 64;
    std::vector<int> coltab, fintab((size_t)C.ngroups * (ntile + 1) * 4, 0);
    std::vector<unsigned long long> need((size_t)C.ngroups * ntile * cw, 0ULL); // the same lists as bit sets over the codes
    { // the class's column tile: 64 when no (group, row tile) lists more than 64 columns -- a class of ONE block lists at most its 48 operations, and a 128-wide tile would
      // multiply 80 columns of zeros (PMH_FXO_TN=128 keeps the wide tile for the A/B)
      int most = 0;
      for (int gr = 0; gr < C.ngroups; gr++)
        for (int mt = 0; mt < ntile; mt++) {
          int n = 0;
          for (int code = 0; code < ncode; code++) {
            const size_t b   = ((size_t)gr * C.nsym + (code >> 3)) * FXS_S + (code & 7);
            bool         any = false;
            for (int row = mt * C.tm; row < std::min(M, (mt + 1) * C.tm) && !any; row++) any = (pat[(size_t)rowrep[row] * nw + b / 64] >> (b % 64)) & 1ULL;
            n += any;
          }
          most = std::max(most, n);
        }
      // (only classes on the table-driven kernel: the single-class kernel of 8-block classes is left as it is)
      C.tn = (C.S != FXS_S && most <= 64) ? 64 : 128;
    }
    for (int gr = 0; gr < C.ngroups; gr++) {
      int elems = 0;
      for (int mt = 0; mt < ntile; mt++) {
        const int coff = (int)coltab.size();
        for (int code = 0; code < ncode; code++) {
          const size_t b   = ((size_t)gr * C.nsym + (code >> 3)) * FXS_S + (code & 7);
          bool         any = false;
          for (int row = mt * C.tm; row < std::min(M, (mt + 1) * C.tm) && !any; row++) any = (pat[(size_t)rowrep[row] * nw + b / 64] >> (b % 64)) & 1ULL;
          if (any) coltab.push_back(code), need[((size_t)gr * ntile + mt) * cw + code / 64] |= 1ULL << (code % 64);
        }
        while ((coltab.size() - coff) % C.tn) coltab.push_back(-1);
        int *ft = fintab.data() + ((size_t)gr * (ntile + 1) + mt) * 4;
        ft[0] = coff, ft[1] = (int)coltab.size() - coff, ft[2] = elems;
        elems += C.tm * ft[1];
      }
      fintab[((size_t)gr * (ntile + 1) + ntile) * 4 + 2] = elems;
      C.fin_elems = std::max(gr ? C.fin_elems : 0, elems);
    }
    // k segments: B[c][(g, slot)] = s_g(c) X[g c][slot] is structurally zero where block (group, slot) does not touch g c.  The signature of a position is the
    // set of columns that are NOT zero there; positions of one signature form a segment (the interior of a face of the cube with one dof component, ...), small
    // ones are pooled, and two segments are joined whenever that does not add column tiles (fewer, longer units split more evenly).  The k index of the product
    // runs segment after segment, each padded to whole chunks, and a (row tile, segment) unit multiplies only the columns of the tile's list that are non-zero
    // on the segment: for a 2 x 2 x 2 decomposition a face segment keeps 128 ... 256 of the 384 columns.  PMH_FXO_NO_KSEG=1: one segment (every listed column
    // over the whole k range).
    const size_t sw = (size_t)C.ngroups * cw;
    std::vector<std::vector<unsigned long long>> ssig;
    std::vector<std::vector<int>>                spos;
    if (prune && !getenv("PMH_FXO_NO_KSEG")) {
      std::map<std::vector<unsigned long long>, int> ids;
      std::vector<unsigned long long>                sg(sw);
      for (int cc = 0; cc < C.nc; cc++) {
        std::fill(sg.begin(), sg.end(), 0ULL);
        for (int gr = 0; gr < C.ngroups; gr++)
          for (int g = 0; g < C.nsym; g++) {
            const char *tm8 = &C.tmask[((size_t)gr * C.nc + C.h_posmap[(size_t)g * C.nc + cc]) * FXS_S];
            for (int sl = 0; sl < FXS_S; sl++)
              if (tm8[sl]) sg[(size_t)gr * cw + (g * FXS_S + sl) / 64] |= 1ULL << ((g * FXS_S + sl) % 64);
          }
        auto it = ids.find(sg);
        if (it == ids.end()) it = ids.emplace(sg, (int)ssig.size()).first, ssig.push_back(sg), spos.emplace_back();
        spos[it->second].push_back(cc);
      }
      const int minseg = getenv("PMH_FXO_SEGMIN") ? std::max(1, atoi(getenv("PMH_FXO_SEGMIN"))) : std::max(2 * FXO_TK, C.nc / 64);
      auto join = [&](size_t a, size_t b) { // b into a
        for (size_t w = 0; w < sw; w++) ssig[a][w] |= ssig[b][w];
        spos[a].insert(spos[a].end(), spos[b].begin(), spos[b].end());
        ssig.erase(ssig.begin() + b), spos.erase(spos.begin() + b);
      };
      long long pool = -1; // the small segments together
      for (size_t i = 0; i < spos.size();) {
        if ((int)spos[i].size() >= minseg) { i++; continue; }
        if (pool < 0) pool = (long long)i++;
        else join((size_t)pool, i);
      }
      auto cost = [&](const std::vector<unsigned long long> &sig, size_t npos) { // chunks x column tiles over the (group, row tile) pairs
        long long tiles = 0;
        for (int gr = 0; gr < C.ngroups; gr++)
          for (int mt = 0; mt < ntile; mt++) {
            int n = 0;
            for (int w = 0; w < cw; w++) n += __builtin_popcountll(need[((size_t)gr * ntile + mt) * cw + w] & sig[(size_t)gr * cw + w]);
            tiles += (n + C.tn - 1) / C.tn;
          }
        return (long long)((npos + FXO_TK - 1) / FXO_TK) * tiles;
      };
      for (;;) { // greedy: the pair whose union saves most (>= 0: equal cost still gives fewer, longer units)
        long long best = -1;
        size_t    ba = 0, bb = 0;
        std::vector<unsigned long long> un(sw);
        for (size_t a2 = 0; a2 < spos.size(); a2++)
          for (size_t b2 = a2 + 1; b2 < spos.size(); b2++) {
            for (size_t w = 0; w < sw; w++) un[w] = ssig[a2][w] | ssig[b2][w];
            const long long save = cost(ssig[a2], spos[a2].size()) + cost(ssig[b2], spos[b2].size()) - cost(un, spos[a2].size() + spos[b2].size());
            if (save > best) best = save, ba = a2, bb = b2;
          }
        if (best < 0) break;
        join(ba, bb);
      }
      for (auto &v : spos) std::sort(v.begin(), v.end());
      std::vector<size_t> order(spos.size());
      for (size_t i = 0; i < order.size(); i++) order[i] = i;
      // the long segments first
      std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return spos[x].size() != spos[y].size() ? spos[x].size() > spos[y].size() : spos[x][0] < spos[y][0]; });
      std::vector<std::vector<unsigned long long>> s2;
      std::vector<std::vector<int>>                p2;
      for (size_t i : order) s2.push_back(ssig[i]), p2.push_back(spos[i]);
      ssig.swap(s2), spos.swap(p2);
    } else {
      ssig.emplace_back(sw, ~0ULL), spos.emplace_back((size_t)C.nc);
      for (int cc = 0; cc < C.nc; cc++) spos[0][cc] = cc;
    }
    C.nseg = (int)spos.size();
    std::vector<int> segc0((size_t)C.nseg + 1, 0); // first chunk of every segment
    C.kinv.assign((size_t)C.nc, 0);
    for (int sg = 0; sg < C.nseg; sg++) {
      for (size_t i = 0; i < spos[sg].size(); i++) C.kinv[spos[sg][i]] = segc0[sg] * FXO_TK + (int)i;
      segc0[sg + 1] = segc0[sg] + ((int)spos[sg].size() + FXO_TK - 1) / FXO_TK;
    }
    C.nkc  = std::max(1, segc0[C.nseg]);
    C.ldk  = C.nkc * FXO_TK;
    C.aoff = atot;
    atot += (long long)C.Mp * C.ldk;
    // gather indices of B: (position of g c) << 1 | (s_g(c) < 0) at row kinv[c]; padded k and padded operations read the zero row nc of X
    // (one more row of gather indices, all on the zero row of X: what the padding columns of the lists below read)
    std::vector<int> gidx((size_t)(C.nsymp + 1) * C.ldk, C.nc << 1);
    for (int g = 0; g < C.nsym; g++)
      for (int cc = 0; cc < C.nc; cc++) gidx[(size_t)g * C.ldk + C.kinv[cc]] = (C.h_posmap[(size_t)g * C.nc + cc] << 1) | (C.h_sign[(size_t)g * C.nc + cc] < 0 ? 1 : 0);
    // the units: (group, row tile, segment) with the columns of the tile's list that are non-zero on the segment; look-up table from the tile's list
    fxo_plan P;
    P.segc0 = segc0;
    for (int gr = 0; gr < C.ngroups; gr++)
      for (int mt = 0; mt <= ntile; mt++) {
        fintab[((size_t)gr * (ntile + 1) + mt) * 4 + 3] = (int)P.units.size();
        if (mt == ntile) break;
        const int *ft = fintab.data() + ((size_t)gr * (ntile + 1) + mt) * 4;
        for (int sg = 0; sg < C.nseg; sg++) {
          fxo_unit U;
          U.g = gr, U.mt = mt, U.seg = sg, U.coff = (int)coltab.size(), U.lutoff = (int)P.lut.size();
          int n = 0;
          for (int j = 0; j < ft[1]; j++) {
            const int  code = coltab[(size_t)ft[0] + j];
            const bool in   = code >= 0 && ((ssig[sg][(size_t)gr * cw + code / 64] >> (code % 64)) & 1ULL);
            P.lut.push_back(in ? n : -1);
            if (in) coltab.push_back(code), n++;
          }
          U.listed = n;
          while ((coltab.size() - U.coff) % C.tn) coltab.push_back(-1);
          U.nct = (int)coltab.size() - U.coff;
          P.units.push_back(U);
        }
      }
    P.coltab = coltab, P.fintab = fintab;
    plan.push_back(P);
    if (C.d_coltab) pmh_free(ctx, C.d_coltab), pmh_free(ctx, C.d_fintab), C.d_coltab = nullptr;
    if (C.d_kinv) pmh_free(ctx, C.d_kinv), pmh_free(ctx, C.d_lut), C.d_kinv = nullptr;
    PMH_CHK(pmh_malloc(ctx, sizeof(int) * std::max<size_t>(1, coltab.size()), (void **)&C.d_coltab));
    PMH_CHK(pmh_malloc(ctx, sizeof(int) * fintab.size(), (void **)&C.d_fintab));
    PMH_CHK(pmh_malloc(ctx, sizeof(int) * std::max<size_t>(1, C.kinv.size()), (void **)&C.d_kinv));
    PMH_CHK(pmh_malloc(ctx, sizeof(int) * std::max<size_t>(1, P.lut.size()), (void **)&C.d_lut));
    if (!coltab.empty()) PMH_CHK(pmh_memcpy_h2d(ctx, C.d_coltab, coltab.data(), sizeof(int) * coltab.size()));
    PMH_CHK(pmh_memcpy_h2d(ctx, C.d_fintab, fintab.data(), sizeof(int) * fintab.size()));
    if (!C.kinv.empty()) PMH_CHK(pmh_memcpy_h2d(ctx, C.d_kinv, C.kinv.data(), sizeof(int) * C.kinv.size()));
    if (!P.lut.empty()) PMH_CHK(pmh_memcpy_h2d(ctx, C.d_lut, P.lut.data(), sizeof(int) * P.lut.size()));
    if (C.d_gidx) pmh_free(ctx, C.d_gidx), pmh_free(ctx, C.d_reppos), pmh_free(ctx, C.d_use);
    PMH_CHK(pmh_malloc(ctx, sizeof(int) * gidx.size(), (void **)&C.d_gidx));
    PMH_CHK(pmh_malloc(ctx, sizeof(int) * reppos.size(), (void **)&C.d_reppos));
    PMH_CHK(pmh_malloc(ctx, use.size(), (void **)&C.d_use));
    PMH_CHK(pmh_memcpy_h2d(ctx, C.d_gidx, gidx.data(), sizeof(int) * gidx.size()));
    PMH_CHK(pmh_memcpy_h2d(ctx, C.d_reppos, reppos.data(), sizeof(int) * reppos.size()));
    PMH_CHK(pmh_memcpy_h2d(ctx, C.d_use, use.data(), use.size()));
  }
  if (S->Afund) (void)hipFree(S->Afund);
  {
    // (k_fxo_gemm16 loads whole 4 KB pieces: up to one piece past the last chunk, never used)
    const size_t bytes = sizeof(double) * (size_t)std::max(32LL, atot) + 8192;
    hipError_t   e     = hipMalloc((void **)&S->Afund, bytes);
    if (e != hipSuccess) return pmh_set_error(PMH_ERR_HIP, "PMH_FX_CLASS_ORBIT: %.2f GB for the representatives' rows: %s", bytes / 1e9, hipGetErrorString(e));
    PMH_HIP(hipMemsetAsync(S->Afund, 0, bytes, ctx->stream));
    S->afund_tot = atot;
  }
  // GEMM work items: (unit, column tile, split of the unit's chunks).  Every unit is split so that no workgroup has more than T chunks, T the smallest for
  // which the class's workgroups still fit ONE round of the 2 resident per CU (measured: 507 workgroups 0.275 ms, 513: 0.34); the k range of a rank (several
  // GPUs) cuts the segments it crosses
  const int rank = S->stripe_size > 1 ? S->stripe_rank : 0, size = std::max(1, S->stripe_size);
  const int minch = getenv("PMH_FXO_MINCH") ? std::max(1, atoi(getenv("PMH_FXO_MINCH"))) : 8;
  // several classes on one row tile share ONE launch (fxo_gemm): the resident workgroups are divided among them
  int nplanned = 0, tm_first = 0, tn_first = 128, tnw_first = 0;
  bool one_tile = true;
  for (int c = 0; c < S->ncls; c++)
    if (tab_of[c] >= 0) {
      if (!nplanned) tm_first = S->C[c].tm, tn_first = S->C[c].tn, tnw_first = S->C[c].tnw;
      else if (S->C[c].tm != tm_first || S->C[c].tn != tn_first || S->C[c].tnw != tnw_first) one_tile = false;
      nplanned++;
    }
  bool small_records = false; // a class with fewer than 8 slots per record: only the table-driven kernel knows the record size
  for (int c = 0; c < S->ncls; c++)
    if (tab_of[c] >= 0 && S->C[c].S != FXS_S) small_records = true;
  // (classes on different row tiles: one table-driven launch per class)
  const bool merged = one_tile && (nplanned > 1 || small_records), tables = merged || small_records;
  const int slots_all = getenv("PMH_FXO_SLOTS") ? std::max(1, atoi(getenv("PMH_FXO_SLOTS"))) : 2 * ctx->num_cus;
  const int slots = merged ? std::max(16, slots_all / nplanned) : slots_all;
  for (int c = 0; c < S->ncls; c++) {
    if (tab_of[c] < 0) continue;
    fxo_plan &P = plan[tab_of[c]];
    const fxs_class &C = S->C[c];
    const int klo = (int)((long long)C.nkc * rank / size), khi = (int)((long long)C.nkc * (rank + 1) / size); // this rank's chunks
    for (fxo_unit &U : P.units) {
      U.kc0 = std::max(klo, P.segc0[U.seg]), U.kc1 = std::min(khi, P.segc0[U.seg + 1]);
      if (U.kc1 <= U.kc0 || !U.listed) U.kc0 = U.kc1 = 0;
    }
  }
  const int fixedS = getenv("PMH_FXO_SPLIT") ? std::max(1, atoi(getenv("PMH_FXO_SPLIT"))) : 0;
  std::vector<int>       items, vnkc(S->ncls, 1), vldk(S->ncls, 16), vncol(S->ncls, 128);
  std::vector<long long> iteml;
  std::vector<int>       wgfirst; // per class: first item of every workgroup (relative to the class's first item) + the end
  long long              ctot = 0;
  S->flops = 0.0, S->flops_issued = 0.0, S->flops_dense = 0.0, S->bytes = 0.0, S->owned_bytes = 0.0;
  int Smax = 1;
  for (int c = 0; c < S->ncls; c++) {
    fxs_class &C = S->C[c];
    if (!C.nc) continue;
    fxo_plan &P = plan[tab_of[c]];
    const int klo = (int)((long long)C.nkc * rank / size), khi = (int)((long long)C.nkc * (rank + 1) / size), nk = khi - klo;
    vnkc[c] = C.nkc, vldk[c] = C.ldk, vncol[c] = C.nsymp * 8;
    C.coff = ctot;
    const int               ntile = C.Mp / C.tm, Mrows = C.m1 - C.m0;
    const std::vector<int> &ftab = P.fintab, &ctab = P.coltab;
    std::vector<long long>  unitbase(P.units.size(), 0);
    std::vector<int>        unittab(P.units.size() * 4, 0);
    // (valid rows) x (listed columns) of the tiles; x chunks of the units; padded tile x chunks; partial tiles
    double                  prod = 0.0, uprod = 0.0, ctiles = 0.0, ptiles = 0.0;
    for (int g = 0; g < C.ngroups; g++)
      for (int mt = 0; mt < ntile; mt++) {
        const int *ft = ftab.data() + ((size_t)g * (ntile + 1) + mt) * 4;
        int        listed = 0;
        for (int j = 0; j < ft[1]; j++) listed += ctab[(size_t)ft[0] + j] >= 0;
        prod += (double)std::max(0, std::min(Mrows, (mt + 1) * C.tm) - mt * C.tm) * listed;
      }
    C.item_first = (int)(items.size() / 8);
    // Pieces: the units with the same number of column tiles form one sequence of chunks (unit after unit), cut into equal pieces of at most T chunks -- T the
    // smallest for which the class's workgroups (one per piece and column tile) still fit ONE round of the 2 resident per CU (measured: 507 workgroups 0.275
    // ms, 513: 0.34).  A piece may end one unit and begin the next (two items for its workgroups, two partial tiles): the kernel is bound by the latency of a
    // workgroup's own chunk loop, so what counts is the LONGEST workgroup, and unit-aligned splits (PMH_FXO_NO_STREAMK=1, or PMH_FXO_SPLIT) leave it at 48
    // chunks where the mean is 41.  Cuts closer than `snap` chunks to a unit's end move there.
    struct part { int u, k0, k1, sp; };
    struct piece { int ntl; std::vector<part> parts; };
    std::vector<piece> pieces;
    int                Tbest = 1, wmax = 0;
    const bool         aligned = fixedS || getenv("PMH_FXO_NO_STREAMK");
    for (fxo_unit &U : P.units) U.S = 0;
    int ntlmax = 0;
    for (const fxo_unit &U : P.units) ntlmax = std::max(ntlmax, U.nct / C.tn);
    if (aligned) {
      auto wgs = [&](int T) {
        long long n = 0;
        for (const fxo_unit &U : P.units) {
          const int nku = U.kc1 - U.kc0;
          if (nku > 0) n += (long long)(U.nct / C.tn) * std::max(1, std::min((nku + T - 1) / T, std::max(1, nku / minch)));
        }
        return n;
      };
      int lo = 1, hi = 1;
      for (const fxo_unit &U : P.units) hi = std::max(hi, U.kc1 - U.kc0);
      while (lo < hi) { // wgs does not grow with T
        const int mid = (lo + hi) / 2;
        if (wgs(mid) <= slots) hi = mid;
        else lo = mid + 1;
      }
      Tbest = lo;
      for (size_t ui = 0; ui < P.units.size(); ui++) {
        fxo_unit &U  = P.units[ui];
        const int nku = U.kc1 - U.kc0;
        U.S          = nku > 0 ? std::max(1, std::min(fixedS ? fixedS : (nku + Tbest - 1) / Tbest, std::max(1, nku / minch))) : 0;
        for (int sp = 0; sp < U.S; sp++) pieces.push_back({U.nct / C.tn, {{(int)ui, U.kc0 + (int)((long long)nku * sp / U.S), U.kc0 + (int)((long long)nku * (sp + 1) / U.S), sp}}});
      }
    } else {
      std::vector<std::vector<int>> seq((size_t)ntlmax + 1); // units by column tile count, in unit order (group, row tile, segment)
      std::vector<long long>        N((size_t)ntlmax + 1, 0);
      for (size_t ui = 0; ui < P.units.size(); ui++)
        if (P.units[ui].kc1 > P.units[ui].kc0) seq[P.units[ui].nct / C.tn].push_back((int)ui), N[P.units[ui].nct / C.tn] += P.units[ui].kc1 - P.units[ui].kc0;
      auto wgs = [&](long long T) {
        long long n = 0;
        for (int k = 1; k <= ntlmax; k++) n += (long long)k * ((N[k] + T - 1) / T);
        return n;
      };
      auto search = [&](int nslots) {
        long long lo = 1, hi = 1;
        for (int k = 1; k <= ntlmax; k++) hi = std::max(hi, N[k]);
        while (lo < hi) {
          const long long mid = (lo + hi) / 2;
          if (wgs(mid) <= nslots) hi = mid;
          else lo = mid + 1;
        }
        return (int)lo;
      };
      Tbest = search(slots);
      // short pieces (a rank's 1/8 share of configs[2]: 5 chunks): the launch is prologue / epilogue / partial tiles rather than products, and one workgroup
      // per CU with pieces twice as long is faster (measured at the 1/8 share: 0.066 -> 0.062 ms per dense apply; 384 slots 0.070, 192: 0.075)
      if (!getenv("PMH_FXO_SLOTS") && Tbest < 12) Tbest = search(ctx->num_cus);
      const int snap = std::max(0, std::min(minch / 4, Tbest / 8));
      for (int k = ntlmax; k >= 1; k--) {
        if (!N[k]) continue;
        const long long W = (N[k] + Tbest - 1) / Tbest;
        std::vector<long long> ends; // prefix sums: the units' ends in the sequence
        long long              acc = 0;
        for (int ui : seq[k]) acc += P.units[ui].kc1 - P.units[ui].kc0, ends.push_back(acc);
        std::vector<long long> cut((size_t)W + 1, 0);
        for (long long i = 1; i < W; i++) {
          long long cpos = N[k] * i / W;
          auto      itb  = std::lower_bound(ends.begin(), ends.end(), cpos);
          if (itb != ends.end() && *itb - cpos <= snap) cpos = *itb;
          else if (itb != ends.begin() && cpos - *(itb - 1) <= snap) cpos = *(itb - 1);
          cut[i] = std::max(cut[i - 1], cpos);
        }
        cut[W] = N[k];
        size_t    iu = 0;
        long long ubeg = 0; // start of unit seq[k][iu] in the sequence
        for (long long i = 0; i < W; i++) {
          if (cut[i + 1] <= cut[i]) continue;
          piece pc{k, {}};
          long long pos = cut[i];
          while (pos < cut[i + 1]) {
            while (ends[iu] <= pos) ubeg = ends[iu], iu++;
            fxo_unit       &U   = P.units[seq[k][iu]];
            const long long upto = std::min(cut[i + 1], ends[iu]);
            pc.parts.push_back({seq[k][iu], U.kc0 + (int)(pos - ubeg), U.kc0 + (int)(upto - ubeg), U.S++});
            pos = upto;
          }
          pieces.push_back(pc);
        }
      }
    }
    // k_fxo_fin walks the units that have partial tiles on this rank only (a rank's share of the k range crosses one to three segments)
    std::vector<int> before(P.units.size() + 1, 0);
    unitbase.clear(), unittab.clear();
    for (size_t ui = 0; ui < P.units.size(); ui++) {
      fxo_unit &U  = P.units[ui];
      const int nku = U.kc1 - U.kc0;
      before[ui]   = (int)unitbase.size();
      U.cbase      = ctot;
      if (U.S > 0) unitbase.push_back(ctot), unittab.insert(unittab.end(), {U.lutoff, U.nct, U.S, 0});
      Smax = std::max(Smax, U.S);
      ctot += (long long)U.S * C.tm * U.nct;
      const double rows = (double)std::max(0, std::min(Mrows, (U.mt + 1) * C.tm) - U.mt * C.tm);
      uprod += rows * U.listed * nku * FXO_TK, ctiles += (double)C.tm * U.nct * nku * FXO_TK, ptiles += (double)U.S * C.tm * U.nct;
    }
    before[P.units.size()] = (int)unitbase.size();
    {
      std::vector<int> ft2 = P.fintab;
      for (size_t i = 3; i < ft2.size(); i += 4) ft2[i] = before[(size_t)ft2[i]];
      PMH_CHK(pmh_memcpy_h2d(ctx, C.d_fintab, ft2.data(), sizeof(int) * ft2.size()));
    }
    int nitem2 = 0; // workgroups with more than one item
    {
      // The column tiles of one piece read the SAME chunks of A at the same pace.  Workgroups b and b + 8 run on one XCD (one L2: MI355X_MICROARCH.md,
      // workgroup dispatch; scripts/micro/census.hip), so the workgroups go out 8 pieces at a time, column tile after column tile: the nt-th tile of a piece
      // sits 8 nt workgroups after its first one and finds the chunk in the XCD's L2 instead of fetching it again from beyond (with default-policy loads of A:
      // scripts/micro/orbit_gemm.hip -DAPLAIN, OG_MAP=2: -8 % per GEMM).  The pieces are grouped by their number of column tiles, so that the groups of 8 are
      // uniform.  The partial sums stay indexed by (unit, split): the order of the workgroups changes nothing in the result.  PMH_FXO_NO_XCDMAP=1: piece after
      // piece, all column tiles each.
      const bool xcdmap = true;
      C.wgf_first = (int)wgfirst.size();
      auto emit = [&](const piece &pc, int nt) {
        wgfirst.push_back((int)(items.size() / 8) - C.item_first);
        int len = 0;
        for (const part &a : pc.parts) {
          const fxo_unit &U = P.units[a.u];
          items.insert(items.end(), {c, U.g, U.mt, nt, a.k0, a.k1, a.sp, U.nct});
          iteml.push_back(C.aoff);
          iteml.push_back(2 * (C.xoff + (long long)U.g * C.ld * C.S)); // in the signed multivector X2
          iteml.push_back(U.cbase + (long long)a.sp * C.tm * U.nct);
          iteml.push_back((long long)U.coff + (long long)nt * C.tn);
          len += a.k1 - a.k0;
        }
        wmax = std::max(wmax, len), nitem2 += pc.parts.size() > 1;
      };
      if (xcdmap) {
        std::stable_sort(pieces.begin(), pieces.end(), [](const piece &x, const piece &y) { return x.ntl > y.ntl; });
        for (size_t s0 = 0; s0 < pieces.size(); s0 += 8) {
          const size_t s1 = std::min(pieces.size(), s0 + 8);
          int          ntmax = 0;
          for (size_t i = s0; i < s1; i++) ntmax = std::max(ntmax, pieces[i].ntl);
          for (int nt = 0; nt < ntmax; nt++)
            for (size_t i = s0; i < s1; i++)
              if (nt < pieces[i].ntl) emit(pieces[i], nt);
        }
      } else {
        for (const piece &pc : pieces)
          for (int nt = 0; nt < pc.ntl; nt++) emit(pc, nt);
      }
      C.wg_count = (int)wgfirst.size() - C.wgf_first;
      wgfirst.push_back((int)(items.size() / 8) - C.item_first);
    }
    C.item_count = (int)(items.size() / 8) - C.item_first;
    if (getenv("PMH_FXO_VERBOSE")) {
      fprintf(stderr, "PMH_FX_CLASS_ORBIT class %d: %d representatives in %d row tiles of %d, %d k segments (chunks:", c, Mrows, ntile, C.tm, C.nseg);
      for (int sg = 0; sg < C.nseg; sg++) fprintf(stderr, " %d", P.segc0[sg + 1] - P.segc0[sg]);
      fprintf(stderr, "), %d workgroups (%d with two or more items) of at most %d chunks (limit %d of %d slots); padded columns per (group, row tile): unit by unit /", C.wg_count, nitem2, wmax, Tbest, slots);
      for (int g = 0; g < C.ngroups; g++)
        for (int mt = 0; mt < ntile; mt++) {
          for (const fxo_unit &U : P.units)
            if (U.g == g && U.mt == mt) fprintf(stderr, " %d", U.kc1 > U.kc0 ? U.nct : 0);
          fprintf(stderr, " of %d /", ftab[((size_t)g * (ntile + 1) + mt) * 4 + 1]);
        }
      fprintf(stderr, " (all: %d); listed x rows / all = %.3f, non-zero k of those = %.3f\n", C.nsym * 8, prod / std::max(1.0, (double)C.ngroups * Mrows * C.nsym * 8),
              uprod / std::max(1.0, prod * nk * FXO_TK));
    }
    if (C.d_finbase) pmh_free(ctx, C.d_finbase), pmh_free(ctx, C.d_unittab), C.d_finbase = nullptr;
    PMH_CHK(pmh_malloc(ctx, sizeof(long long) * std::max<size_t>(1, unitbase.size()), (void **)&C.d_finbase));
    PMH_CHK(pmh_malloc(ctx, sizeof(int) * std::max<size_t>(4, unittab.size()), (void **)&C.d_unittab));
    if (!unitbase.empty()) {
      PMH_CHK(pmh_memcpy_h2d(ctx, C.d_finbase, unitbase.data(), sizeof(long long) * unitbase.size()));
      PMH_CHK(pmh_memcpy_h2d(ctx, C.d_unittab, unittab.data(), sizeof(int) * unittab.size()));
    }
    const double M = Mrows, share = (double)nk / std::max(1, C.nkc);
    // the products of the listed columns with the tiles' rows over the rank's k range (padding rows and columns not counted; structural zeros of B counted)
    S->flops += 2.0 * C.nc * share * prod;
    S->flops_issued += 2.0 * ctiles, S->flops_dense += (double)C.ngroups * 2.0 * M * C.nc * share * 8.0 * C.nsym;
    S->owned_bytes += 8.0 * M * C.nc;
    // this rank's columns of A once + the gathered B + the split partial tiles written and read + Y
    S->bytes += (double)C.ngroups * (8.0 * M * C.nc * share + 8.0 * FXS_S * C.nc * share + 8.0 * FXS_S * C.nc) + 2.0 * 8.0 * ptiles;
  }
  S->fxo_S = Smax;
  S->nwg = 0;
  for (const fxs_class &C : S->C) S->nwg += C.wg_count;
  items.insert(items.end(), {0, 0, 0, 0, 0, 0, 0, 0});
  iteml.insert(iteml.end(), {0, 0, 0, 0});
  if (S->d_items) pmh_free(ctx, S->d_items);
  if (S->d_wgl) pmh_free(ctx, S->d_wgl);
  if (S->d_wg) pmh_free(ctx, S->d_wg);
  if (S->d_wgfirst) pmh_free(ctx, S->d_wgfirst);
  wgfirst.push_back(0);
  PMH_CHK(pmh_malloc(ctx, sizeof(int) * wgfirst.size(), (void **)&S->d_wgfirst));
  PMH_CHK(pmh_memcpy_h2d(ctx, S->d_wgfirst, wgfirst.data(), sizeof(int) * wgfirst.size()));
  PMH_CHK(pmh_malloc(ctx, sizeof(int) * items.size(), (void **)&S->d_items));
  PMH_CHK(pmh_memcpy_h2d(ctx, S->d_items, items.data(), sizeof(int) * items.size()));
  PMH_CHK(pmh_malloc(ctx, sizeof(long long) * iteml.size(), (void **)&S->d_wgl));
  PMH_CHK(pmh_memcpy_h2d(ctx, S->d_wgl, iteml.data(), sizeof(long long) * iteml.size()));
  // per class: nkc, ldk, ncol (ints) in d_wg
  std::vector<int> meta;
  meta.insert(meta.end(), vnkc.begin(), vnkc.end()), meta.insert(meta.end(), vldk.begin(), vldk.end()), meta.insert(meta.end(), vncol.begin(), vncol.end());
  PMH_CHK(pmh_malloc(ctx, sizeof(int) * meta.size(), (void **)&S->d_wg));
  PMH_CHK(pmh_memcpy_h2d(ctx, S->d_wg, meta.data(), sizeof(int) * meta.size()));
  if (ctot > S->cpart_cap) {
    if (S->cpart) pmh_free(ctx, S->cpart);
    PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)std::max(16LL, ctot), (void **)&S->cpart));
    S->cpart_cap = ctot;
  }
  // merged launch: workgroup -> items in the global item numbering, and the classes' own tables by class index
  if (S->d_wgfirst_all) pmh_free(ctx, S->d_wgfirst_all), S->d_wgfirst_all = nullptr;
  if (S->d_zrow_of) pmh_free(ctx, S->d_zrow_of), pmh_free(ctx, (void *)S->d_coltab_of), pmh_free(ctx, (void *)S->d_gidx_of), S->d_zrow_of = nullptr;
  S->nwg_all = 0, S->merged_tm = 0;
  if (tables) {
    std::vector<int>         wall, zr((size_t)S->ncls, 0), xs((size_t)S->ncls, 6);
    std::vector<const int *> ct((size_t)S->ncls, nullptr), gi((size_t)S->ncls, nullptr);
    for (int c = 0; c < S->ncls; c++) {
      const fxs_class &C = S->C[c];
      zr[c] = C.nsymp, ct[c] = C.d_coltab, gi[c] = C.d_gidx;
      for (xs[c] = 3; (1 << (xs[c] - 3)) < C.S; xs[c]++) {}
      // (a class's items are contiguous and the classes follow one another:
      for (int w = 0; w < C.wg_count; w++) wall.push_back(wgfirst[C.wgf_first + w] + C.item_first);
      // a workgroup ends where the next one, of whichever class, begins)
      S->nwg_all += C.wg_count;
    }
    int last_end = 0;
    for (int c = 0; c < S->ncls; c++)
      if (S->C[c].wg_count) last_end = wgfirst[S->C[c].wgf_first + S->C[c].wg_count] + S->C[c].item_first;
    wall.push_back(last_end);
    PMH_CHK(pmh_malloc(ctx, sizeof(int) * wall.size(), (void **)&S->d_wgfirst_all));
    PMH_CHK(pmh_memcpy_h2d(ctx, S->d_wgfirst_all, wall.data(), sizeof(int) * wall.size()));
    zr.insert(zr.end(), xs.begin(), xs.end()); // [zrow of the classes | record shifts of the classes]
    PMH_CHK(pmh_malloc(ctx, sizeof(int) * zr.size(), (void **)&S->d_zrow_of));
    PMH_CHK(pmh_memcpy_h2d(ctx, S->d_zrow_of, zr.data(), sizeof(int) * zr.size()));
    PMH_CHK(pmh_malloc(ctx, sizeof(const int *) * ct.size(), (void **)&S->d_coltab_of));
    PMH_CHK(pmh_memcpy_h2d(ctx, (void *)S->d_coltab_of, ct.data(), sizeof(const int *) * ct.size()));
    PMH_CHK(pmh_malloc(ctx, sizeof(const int *) * gi.size(), (void **)&S->d_gidx_of));
    PMH_CHK(pmh_memcpy_h2d(ctx, (void *)S->d_gidx_of, gi.data(), sizeof(const int *) * gi.size()));
    S->merged_tm = merged ? tm_first : 0, S->merged_tn = tn_first, S->merged_tnw = tnw_first;
  }
  if (S->d_fin_args) pmh_free(ctx, S->d_fin_args), S->d_fin_args = nullptr;
  S->fin_nbx = S->fin_ngroups = 0;
  if (S->ncls > 1) { // the classes' finishing kernels in one launch
    std::vector<fxo_fin_args> fa((size_t)S->ncls);
    for (int c = 0; c < S->ncls; c++) {
      const fxs_class &C = S->C[c];
      fxo_fin_args     &a = fa[c];
      memset(&a, 0, sizeof(a));
      if (!C.nc || C.fin_elems <= 0) continue; // nbx = 0: the class's workgroups return at once
      a.ntile = C.Mp / C.tm, a.tm = C.tm, a.nsymp = C.nsymp, a.nc = C.nc, a.ld = C.ld, a.nslot = C.S, a.nbx = (C.fin_elems + PMH_BLOCK - 1) / PMH_BLOCK, a.ngroups = C.ngroups;
      a.fintab = C.d_fintab, a.unittab = C.d_unittab, a.lut = C.d_lut, a.coltab = C.d_coltab, a.reppos = C.d_reppos, a.posmap = C.d_posmap, a.unitbase = C.d_finbase, a.use = C.d_use, a.xbase0 = C.xoff;
      S->fin_nbx = std::max(S->fin_nbx, a.nbx), S->fin_ngroups = std::max(S->fin_ngroups, a.ngroups);
    }
    PMH_CHK(pmh_malloc(ctx, sizeof(fxo_fin_args) * fa.size(), &S->d_fin_args));
    PMH_CHK(pmh_memcpy_h2d(ctx, S->d_fin_args, fa.data(), sizeof(fxo_fin_args) * fa.size()));
  }
  S->fxo_ready = 1;
  return PMH_SUCCESS;
}
