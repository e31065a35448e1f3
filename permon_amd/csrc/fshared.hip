// Explicit local dual operators SHARED by congruent blocks (storage PMH_FX_CLASS of pmh_fexplicit).
//
// Subdomains with bit-identical matrices (pmh_csr_block_classes: the 8 cubes of configs[2], the 64 of configs[3]) have the same K^+,
// so their dense operators W_b = (K^+)[Gamma_b, Gamma_b] are principal sub-matrices of ONE matrix W_c = (K^+)[U_c, U_c] on the union
// U_c of the dofs B touches in any block of the class (the whole boundary of the cube: 33 288 dofs for configs[2]).  Instead of one dense
// matrix per block (8 x 4 n_b^2 = 14.2 GB in symmetric storage) the class keeps W_c once in full (8 n_c^2 = 8.9 GB) and applies it to
// the blocks' vectors TOGETHER: X_c = [xhat_b scattered into U_c]_b is an n_c x 8 multivector (zero where a block does not touch a
// dof), Y_c = W_c X_c ONE pass over the matrix with eight right-hand sides (more blocks than 8: one pass per group of 8).  The matrix
// bytes per F apply drop by 1.6 x for configs[2] (by 2.9 x for the shape of configs[3]); the kernel walks down the rows with the lane
// owning its output columns (no reduction across lanes), and the rows of W_c deal over several GPUs as contiguous ranges.
// The gluing over the multivector numbering (index = (position in U_c) * 8 + slot of the block) is a pmh_gluing, so
//   F lambda = Bc' -> X,  Y = W_c X,  Bc Y (+ all-reduce)           stays three launches.
// Set-up: one K^+ solve per dof of U_c (what the per-block storage already did for congruent blocks), each giving one full row of W_c.
//
// Three storages of W_c live in this file (fx_shared::sym):
//   0  PMH_FX_CLASS        the full matrix, k_fxs_gemm8: 8 n_c^2 bytes per apply, HBM-bound
//   1 PMH_FX_CLASS_SYM its lower block-triangle in 16 x 16 tiles, k_fxs_symm8 (both products of a tile on the fp64 matrix instruction): 4 n_c^2 bytes,
//     HBM-bound
//   2 PMH_FX_CLASS_ORBIT only the rows of the orbit representatives under the class's symmetries, k_fxo_gemm16: a GEMM on the fp64 matrix
//     instruction,
//                          4 n_c^2 / 24 bytes for the cube's 48 operations, compute-bound (the default for congruent cubes)
// and the set-up by symmetry (fxs_set_symmetry: one K^+ solve per orbit of rows, self-checked against direct solves) serves 1 and 2.
// The file is split four ways (round 5): fshared_types.h (the storage structures), fshared_kernels.h (the device kernels), fshared_plan.hip (the orbit GEMM's plan, host only)
// and this file: creation, stripes, symmetries, the launches of the products, the assembly by K^+ solves.
#include "fshared_kernels.h"


static int fxs_build_launch(fx_shared *S)
{
  if (S->sym) {
    // work = (class, group, owned mega band m) one after the other, the longest first, each a run of steps (pairs of column tiles, 128 KB
    // when all four super bands reach that far); the run is cut into one equal share per CU
    struct band { int c, g, m, nj; };
    std::vector<band> bands;
    long long         steps = 0;
    for (int c = 0; c < S->ncls; c++) {
      fxs_class &C = S->C[c];
      for (int m = C.nmb - 1; m >= 0; m--) {
        if (!C.own[FXM_MB * m]) continue;
        for (int g = 0; g < C.ngroups; g++) {
          const int nj = std::min(C.nsb, FXM_MB * (m + 1)) * FXM_RT;
          bands.push_back({c, g, m, nj});
          steps += (nj + 1) / 2;
        }
      }
    }
    int nwg = (int)std::max(1LL, std::min((long long)S->ctx->num_cus, steps));
    if (const char *e = getenv("PMH_FXM_NWG")) nwg = std::max(1, atoi(e));
    std::vector<int>       first(1, 0), items;
    std::vector<long long> iteml;
    std::vector<std::vector<int>> nseg_of(S->ncls);
    for (int c = 0; c < S->ncls; c++) nseg_of[c].assign((size_t)std::max(1, S->C[c].ngroups * S->C[c].nmb), 0);
    {
      long long done = 0; // steps handed out so far
      size_t    bi = 0;
      long long boff = 0; // steps of band bi already handed out
      for (int k = 0; k < nwg; k++) {
        const long long upto = steps * (k + 1) / nwg;
        while (done < upto && bi < bands.size()) {
          const band     &b   = bands[bi];
          const long long bst = (b.nj + 1) / 2, take = std::min(bst - boff, upto - done);
          fxs_class      &C   = S->C[b.c];
          int            &ns  = nseg_of[b.c][(size_t)b.g * C.nmb + b.m];
          items.insert(items.end(), {b.c, b.g, b.m, (int)(2 * boff), (int)std::min<long long>(b.nj, 2 * (boff + take)), ns, 0, 0});
          iteml.push_back(C.woff);
          iteml.push_back(C.ptoff + (long long)b.g * C.ptsize + C.ptm[b.m]);
          ns++, done += take, boff += take;
          if (boff == bst) bi++, boff = 0;
        }
        first.push_back((int)(items.size() / 8));
      }
    }
    int nsegmax = 1;
    S->bytes = 0.0, S->owned_bytes = 0.0;
    for (int c = 0; c < S->ncls; c++) {
      fxs_class &C     = S->C[c];
      double     tiles = 0.0, parts = 0.0;
      std::vector<int> ownm, own_first((size_t)C.nmb + 1, 0);
      for (int m = 0; m < C.nmb; m++) {
        own_first[m] = (int)ownm.size();
        if (C.own[FXM_MB * m]) ownm.push_back(m);
      }
      C.nown = (int)ownm.size();
      std::vector<long long> ptoff_of((size_t)std::max(1, C.ngroups * C.nown), 0);
      for (int g = 0; g < C.ngroups; g++)
        for (int k = 0; k < C.nown; k++) ptoff_of[(size_t)g * C.nown + k] = C.ptoff + (long long)g * C.ptsize + C.ptm[ownm[k]];
      if (!C.d_ownfirst) PMH_CHK(pmh_malloc(S->ctx, sizeof(int) * own_first.size(), (void **)&C.d_ownfirst));
      PMH_CHK(pmh_memcpy_h2d(S->ctx, C.d_ownfirst, own_first.data(), sizeof(int) * own_first.size()));
      for (int m = 0; m < C.nmb; m++) {
        if (!C.own[FXM_MB * m]) continue;
        const int sb1 = std::min(C.nsb, FXM_MB * (m + 1));
        for (int sb = FXM_MB * m; sb < sb1; sb++) tiles += (double)(sb + 1) * FXM_RT * FXM_RT * 2048.0;
        int ns = 0;
        for (int g = 0; g < C.ngroups; g++) ns = std::max(ns, nseg_of[c][(size_t)g * C.nmb + m]);
        nsegmax = std::max(nsegmax, ns);
        // direct sums per item + transposed sums per column
        parts += (double)ns * (sb1 - FXM_MB * m) * FXM_RS * FXS_S * 8.0 + (double)sb1 * FXM_RS * FXS_S * 8.0;
      }
      S->owned_bytes += tiles;
      // the owned tiles once per group + X read (rows + columns) + the partial sums written and read back + Y written
      S->bytes += (double)C.ngroups * (tiles + 2.0 * parts + 2.0 * 8.0 * FXS_S * C.ld);
      if (!C.d_nseg) PMH_CHK(pmh_malloc(S->ctx, sizeof(int) * nseg_of[c].size(), (void **)&C.d_nseg));
      PMH_CHK(pmh_memcpy_h2d(S->ctx, C.d_nseg, nseg_of[c].data(), sizeof(int) * nseg_of[c].size()));
      if (C.d_ptoff) pmh_free(S->ctx, C.d_ptoff), C.d_ptoff = nullptr;
      if (!C.d_ptoff) PMH_CHK(pmh_malloc(S->ctx, sizeof(long long) * ptoff_of.size(), (void **)&C.d_ptoff));
      PMH_CHK(pmh_memcpy_h2d(S->ctx, C.d_ptoff, ptoff_of.data(), sizeof(long long) * ptoff_of.size()));
    }
    S->nwg = items.empty() ? 0 : nwg, S->nseg = nsegmax;
    items.insert(items.end(), {0, 0, 0, 0, 0, 0, 0, 0});
    iteml.insert(iteml.end(), {0, 0});
    if (S->d_wg) pmh_free(S->ctx, S->d_wg);
    if (S->d_items) pmh_free(S->ctx, S->d_items);
    if (S->d_wgl) pmh_free(S->ctx, S->d_wgl);
    PMH_CHK(pmh_malloc(S->ctx, sizeof(int) * first.size(), (void **)&S->d_wg));
    PMH_CHK(pmh_memcpy_h2d(S->ctx, S->d_wg, first.data(), sizeof(int) * first.size()));
    PMH_CHK(pmh_malloc(S->ctx, sizeof(int) * items.size(), (void **)&S->d_items));
    PMH_CHK(pmh_memcpy_h2d(S->ctx, S->d_items, items.data(), sizeof(int) * items.size()));
    PMH_CHK(pmh_malloc(S->ctx, sizeof(long long) * iteml.size(), (void **)&S->d_wgl));
    PMH_CHK(pmh_memcpy_h2d(S->ctx, S->d_wgl, iteml.data(), sizeof(long long) * iteml.size()));
    const long long need = (long long)nsegmax * std::max(16LL, S->nX);
    if (need > S->part_cap) {
      if (S->part) pmh_free(S->ctx, S->part);
      PMH_CHK(pmh_malloc(S->ctx, sizeof(double) * (size_t)need, (void **)&S->part));
      S->part_cap = need;
    }
    return PMH_SUCCESS; // k_fxs_symfin reads only what the items of a mega band wrote
  }
  // segments of the rank's rows: enough of them to give the chip >= ~4000 waves (n_c / 128 column chunks each), at most 32
  int maxrows = 0, chunks = 0;
  for (auto &C : S->C) maxrows = std::max(maxrows, C.r1 - C.r0), chunks += C.ngroups * (C.ld / 128);
  int nseg = std::max(1, std::min(32, (4096 + std::max(1, chunks) - 1) / std::max(1, chunks)));
  nseg     = std::max(1, std::min(nseg, maxrows / FXS_U));
  S->nseg = nseg;
  std::vector<int> wg;
  S->bytes = 0.0;
  for (int c = 0; c < S->ncls; c++) {
    fxs_class &C    = S->C[c];
    const int  rows = C.r1 - C.r0;
    for (int g = 0; g < C.ngroups; g++)
      for (int j = 0; j < nseg; j++) {
        const int lo = C.r0 + (int)((long long)rows * j / nseg), hi = C.r0 + (int)((long long)rows * (j + 1) / nseg);
        for (int c0 = 0; c0 < C.ld; c0 += 512) wg.insert(wg.end(), {c, g, c0, j, lo, hi});
      }
    // its rows once per group + X read + the segment sums written and read back + Y written
    S->bytes += (double)C.ngroups * (8.0 * (double)rows * C.ld + 8.0 * FXS_S * rows + (2.0 * nseg + 1.0) * 8.0 * FXS_S * C.ld);
  }
  S->nwg = (int)(wg.size() / 6);
  wg.insert(wg.end(), {0, 0, 0, 0, 0, 0});
  if (S->d_wg) pmh_free(S->ctx, S->d_wg);
  PMH_CHK(pmh_malloc(S->ctx, sizeof(int) * wg.size(), (void **)&S->d_wg));
  PMH_CHK(pmh_memcpy_h2d(S->ctx, S->d_wg, wg.data(), sizeof(int) * wg.size()));
  const long long need = (long long)nseg * std::max(16LL, S->nX);
  if (need > S->part_cap) {
    if (S->part) pmh_free(S->ctx, S->part);
    PMH_CHK(pmh_malloc(S->ctx, sizeof(double) * (size_t)need, (void **)&S->part));
    S->part_cap = need;
  }
  return pmh_memset(S->ctx, S->part, 0, sizeof(double) * (size_t)need); // column chunks beyond a class's ld / empty segments stay zero
}

// extra_ptr / extra_rel (optional): block-relative dofs ADDED to the touched set of class c (extra_rel[extra_ptr[c] .. extra_ptr[c + 1])): the closure of the
// touched set under the block's symmetry group (pmh_box_symmetry_closure), so that a class whose own touched set is not invariant keeps all its symmetries
int fxs_create(pmh_gluing B, pmh_blockdiag K, const int *block_class, int sym, fx_shared **out, const int *extra_ptr, const int *extra_rel)
{
  PMH_ARG(B && K && block_class && out && B->n_x == K->n);
  pmh_ctx    ctx = B->ctx;
  fx_shared *S   = new fx_shared();
  S->ctx = ctx, S->B = B, S->K = K, S->nb = K->nblocks, S->sym = sym;
  S->cls.assign(block_class, block_class + S->nb);
  S->ncls = 0;
  for (int b = 0; b < S->nb; b++) {
    PMH_ARG(block_class[b] >= 0);
    S->ncls = std::max(S->ncls, block_class[b] + 1);
  }
  S->C.resize(S->ncls);
  std::vector<int> slot(S->nb), group(S->nb);
  for (int b = 0; b < S->nb; b++) {
    fxs_class &C  = S->C[S->cls[b]];
    const int  nl = K->rowstart[b + 1] - K->rowstart[b];
    if (C.blocks.empty()) C.nloc = nl;
    else if (C.nloc != nl) return pmh_set_error(PMH_ERR_ARG, "pmh_fexplicit_create_shared: blocks of class %d differ in size", S->cls[b]);
    slot[b] = (int)C.blocks.size() % FXS_S, group[b] = (int)C.blocks.size() / FXS_S;
    C.blocks.push_back(b);
  }
  if (sym == 2)
    for (auto &C : S->C) {
      C.S = 1;
      while (C.S < FXS_S && C.S < (int)C.blocks.size()) C.S *= 2;
    }
  // union of the touched dofs per class
  for (int c = 0; c < S->ncls; c++) S->C[c].pos.assign((size_t)std::max(1, S->C[c].nloc), -1);
  auto block_of = [&](int i) { return (int)(std::upper_bound(K->rowstart.begin(), K->rowstart.end(), i) - K->rowstart.begin()) - 1; };
  std::vector<int> lb((size_t)std::max(1, B->n_leaves));
  for (int i = 0; i < B->n_leaves; i++) {
    lb[i] = block_of(B->h_row[i]);
    S->C[S->cls[lb[i]]].pos[B->h_row[i] - K->rowstart[lb[i]]] = 0;
  }
  if (extra_ptr && extra_rel)
    for (int c = 0; c < S->ncls; c++)
      for (int e = extra_ptr[c]; e < extra_ptr[c + 1]; e++) {
        if (extra_rel[e] < 0 || extra_rel[e] >= S->C[c].nloc) return pmh_set_error(PMH_ERR_ARG, "pmh_fexplicit_create_shared: extra dof %d of class %d is outside its blocks (%d rows)", extra_rel[e], c, S->C[c].nloc);
        S->C[c].pos[extra_rel[e]] = 0;
      }
  long long wtot = 0, xtot = 0, pttot = 0;
  for (int c = 0; c < S->ncls; c++) {
    fxs_class &C = S->C[c];
    for (int i = 0; i < C.nloc; i++)
      if (C.pos[i] == 0) C.pos[i] = (int)C.urel.size(), C.urel.push_back(i);
    C.nc      = (int)C.urel.size();
    const int pad = sym == 2 ? 32 : (sym ? FXM_RS : FXS_PAD);
    C.ld      = (C.nc + pad - 1) / pad * pad;
    C.ngroups = ((int)C.blocks.size() + FXS_S - 1) / FXS_S;
    C.r0 = 0, C.r1 = C.ld;
    C.woff = wtot, C.xoff = xtot;
    if (sym == 2) {
      if (C.ld == C.nc) C.ld += 32; // a zero row of X behind the touched dofs for the padded k range of the GEMM
    } else if (sym) {
      C.nsb = C.ld / FXM_RS, C.nmb = (C.nsb + FXM_MB - 1) / FXM_MB;
      C.own.assign((size_t)std::max(1, C.nsb), 1);
      C.ptm.assign((size_t)C.nmb + 1, 0);
      for (int m = 0; m < C.nmb; m++) C.ptm[m + 1] = C.ptm[m] + (long long)std::min(C.nsb, FXM_MB * (m + 1)) * FXM_RS * FXS_S;
      C.ptoff = pttot, C.ptsize = C.ptm[C.nmb];
      pttot += C.ngroups * C.ptsize;
      wtot += (long long)FXM_RS * FXM_RS * ((long long)C.nsb * (C.nsb + 1) / 2); // super band sb: (sb + 1) * 16 column tiles x 16 row tiles x 256 doubles
    } else
      wtot += (long long)C.ld * C.ld;
    xtot += (long long)C.ngroups * C.ld * C.S;
    PMH_CHK(pmh_malloc(ctx, sizeof(int) * (size_t)std::max(1, C.nc), (void **)&C.d_urel));
    if (C.nc) PMH_CHK(pmh_memcpy_h2d(ctx, C.d_urel, C.urel.data(), sizeof(int) * (size_t)C.nc));
  }
  if (xtot >= (1LL << 31)) return pmh_set_error(PMH_ERR_SUP, "pmh_fexplicit_create_shared: the multivector numbering exceeds 32-bit indices");
  S->nX = xtot, S->wtot = wtot;
  // gluing over the multivector numbering: leaf of block b at relative dof i -> (position of i in U_c) * 8 + slot(b), group by group
  std::vector<int> rows((size_t)std::max(1, B->n_leaves));
  for (int i = 0; i < B->n_leaves; i++) {
    const int        b = lb[i];
    const fxs_class &C = S->C[S->cls[b]];
    rows[i]            = (int)(C.xoff + (long long)group[b] * C.ld * C.S + (long long)C.pos[B->h_row[i] - K->rowstart[b]] * C.S + slot[b]);
  }
  if (sym == 2) {
    for (auto &C : S->C) C.tmask.assign((size_t)std::max(1, C.ngroups) * std::max(1, C.nc) * FXS_S, 0);
    for (int i = 0; i < B->n_leaves; i++) {
      const int  b = lb[i];
      fxs_class &C = S->C[S->cls[b]];
      C.tmask[((size_t)group[b] * C.nc + C.pos[B->h_row[i] - K->rowstart[b]]) * FXS_S + slot[b]] = 1;
    }
  }
  PMH_CHK(pmh_gluing_create(ctx, (int)std::max(1LL, xtot), B->n_lambda, B->n_leaves, rows.data(), B->h_root.data(), B->h_sign.data(), &S->Bc));
  if (sym == 2) {
    if (2 * xtot >= (1LL << 31)) return pmh_set_error(PMH_ERR_SUP, "pmh_fexplicit_create_shared_orbit: the signed multivector numbering exceeds 32-bit indices");
    std::vector<int>    rows2((size_t)std::max(1, 2 * B->n_leaves)), root2((size_t)std::max(1, 2 * B->n_leaves));
    std::vector<double> sign2((size_t)std::max(1, 2 * B->n_leaves));
    for (int i = 0; i < B->n_leaves; i++) {
      // entry (position, slot) -> (position, +, slot) and (position, -, slot)
      const int slot_i = slot[lb[i]], base = rows[i] - slot_i, Sc = S->C[S->cls[lb[i]]].S;
      rows2[2 * i] = 2 * base + slot_i, rows2[2 * i + 1] = 2 * base + Sc + slot_i;
      root2[2 * i] = root2[2 * i + 1] = B->h_root[i];
      sign2[2 * i] = B->h_sign[i], sign2[2 * i + 1] = -B->h_sign[i];
    }
    PMH_CHK(pmh_gluing_create(ctx, (int)std::max(1LL, 2 * xtot), B->n_lambda, 2 * B->n_leaves, rows2.data(), root2.data(), sign2.data(), &S->Bc2));
    PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)std::max(32LL, 2 * xtot), (void **)&S->X2));
    PMH_CHK(pmh_memset(ctx, S->X2, 0, sizeof(double) * (size_t)std::max(32LL, 2 * xtot)));
  }
  {
    const size_t bytes = sizeof(double) * (size_t)std::max(32LL, wtot);
    hipError_t   e     = hipMalloc((void **)&S->Wbase, bytes);
    if (e != hipSuccess) return pmh_set_error(PMH_ERR_HIP, "pmh_fexplicit_create_shared: %.2f GB for the shared explicit operators: %s", bytes / 1e9, hipGetErrorString(e));
    PMH_HIP(hipMemsetAsync(S->Wbase, 0, bytes, ctx->stream));
  }
  if (sym == 1) {
    S->pt_tot = pttot;
    PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)std::max(16LL, pttot), (void **)&S->pt));
  }
  PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)std::max(16LL, xtot), (void **)&S->X));
  PMH_CHK(pmh_malloc(ctx, sizeof(double) * (size_t)std::max(16LL, xtot), (void **)&S->Y));
  PMH_CHK(pmh_memset(ctx, S->X, 0, sizeof(double) * (size_t)std::max(16LL, xtot)));
  PMH_CHK(pmh_memset(ctx, S->Y, 0, sizeof(double) * (size_t)std::max(16LL, xtot))); // rows of other ranks' stripes stay zero
  std::vector<int>       ldv(S->ncls);
  std::vector<long long> wo(S->ncls), xo(S->ncls);
  for (int c = 0; c < S->ncls; c++) ldv[c] = S->C[c].ld, wo[c] = S->C[c].woff, xo[c] = S->C[c].xoff;
  PMH_CHK(pmh_malloc(ctx, sizeof(int) * S->ncls, (void **)&S->d_ld));
  PMH_CHK(pmh_malloc(ctx, sizeof(long long) * S->ncls, (void **)&S->d_woff));
  PMH_CHK(pmh_malloc(ctx, sizeof(long long) * S->ncls, (void **)&S->d_xoff));
  PMH_CHK(pmh_memcpy_h2d(ctx, S->d_ld, ldv.data(), sizeof(int) * S->ncls));
  PMH_CHK(pmh_memcpy_h2d(ctx, S->d_woff, wo.data(), sizeof(long long) * S->ncls));
  PMH_CHK(pmh_memcpy_h2d(ctx, S->d_xoff, xo.data(), sizeof(long long) * S->ncls));
  if (sym != 2) PMH_CHK(fxs_build_launch(S)); // orbit storage: planned once the symmetries are known (fxo_prepare)
  *out = S;
  return PMH_SUCCESS;
}

void fxs_destroy(fx_shared *S)
{
  if (!S) return;
  pmh_ctx ctx = S->ctx;
  for (auto &C : S->C) {
    if (C.d_urel) pmh_free(ctx, C.d_urel);
    if (C.d_nseg) pmh_free(ctx, C.d_nseg);
    if (C.d_ptoff) pmh_free(ctx, C.d_ptoff);
    if (C.d_ownfirst) pmh_free(ctx, C.d_ownfirst);
    if (C.d_posmap) pmh_free(ctx, C.d_posmap), pmh_free(ctx, C.d_sign);
    if (C.d_gidx) pmh_free(ctx, C.d_gidx), pmh_free(ctx, C.d_reppos), pmh_free(ctx, C.d_use);
    if (C.d_coltab) pmh_free(ctx, C.d_coltab), pmh_free(ctx, C.d_fintab), pmh_free(ctx, C.d_finbase);
    if (C.d_kinv) pmh_free(ctx, C.d_kinv), pmh_free(ctx, C.d_unittab), pmh_free(ctx, C.d_lut);
  }
  if (S->pt) pmh_free(ctx, S->pt);
  if (S->d_wgl) pmh_free(ctx, S->d_wgl);
  if (S->d_items) pmh_free(ctx, S->d_items);
  if (S->d_wgfirst) pmh_free(ctx, S->d_wgfirst);
  if (S->d_wgfirst_all) pmh_free(ctx, S->d_wgfirst_all);
  if (S->d_fin_args) pmh_free(ctx, S->d_fin_args);
  if (S->d_zrow_of) pmh_free(ctx, S->d_zrow_of), pmh_free(ctx, (void *)S->d_coltab_of), pmh_free(ctx, (void *)S->d_gidx_of);
  pmh_gluing_destroy(S->Bc);
  for (auto e : S->ev_mid) (void)hipEventDestroy(e);
  pmh_gluing_destroy(S->Bc2);
  pmh_free(ctx, S->X2);
  if (S->Wbase) (void)hipFree(S->Wbase);
  if (S->Afund) (void)hipFree(S->Afund);
  if (S->cpart) pmh_free(ctx, S->cpart);
  if (S->part) pmh_free(ctx, S->part);
  pmh_free(ctx, S->X), pmh_free(ctx, S->Y), pmh_free(ctx, S->d_wg), pmh_free(ctx, S->d_ld), pmh_free(ctx, S->d_woff), pmh_free(ctx, S->d_xoff);
  for (hipEvent_t e : S->ev) (void)hipEventDestroy(e);
  delete S;
}


// the dealing rule of the symmetric tile storage: mega band m of nmb (1024 rows; cost ~ m + 1) -> rank: from the longest down in snake order
static inline int fxm_owner(int m, int nmb, int size)
{
  const int i = nmb - 1 - m, round = i / size, k = i % size;
  return (round & 1) ? size - 1 - k : k;
}

// host helper (no device): the owner rank of every mega band of a class with n_c touched dofs and the tile bytes per rank under that rule
extern "C" int pmh_fexplicit_class_sym_plan(int n_c, int size, int *owner_out, double *bytes_per_rank)
{
  PMH_ARG(n_c >= 1 && size >= 1);
  const int nsb = (n_c + FXM_RS - 1) / FXM_RS, nmb = (nsb + FXM_MB - 1) / FXM_MB;
  if (bytes_per_rank)
    for (int r = 0; r < size; r++) bytes_per_rank[r] = 0.0;
  for (int m = 0; m < nmb; m++) {
    const int o = fxm_owner(m, nmb, size);
    if (owner_out) owner_out[m] = o;
    if (bytes_per_rank)
      for (int sb = FXM_MB * m; sb < std::min(nsb, FXM_MB * (m + 1)); sb++) bytes_per_rank[o] += (double)(sb + 1) * FXM_RT * FXM_RT * 2048.0;
  }
  return PMH_SUCCESS;
}

// several GPUs: rank r applies / assembles the rows [r0, r1) of every W_c, contiguous ranges of equal length (multiples of 32)
int fxs_set_stripe(fx_shared *S, int rank, int size)
{
  if (S->sym == 2) { // orbit storage: a contiguous range of the representatives per rank, planned by fxo_prepare
    S->stripe_rank = rank, S->stripe_size = size, S->fxo_ready = 0;
    return PMH_SUCCESS;
  }
  if (S->sym) {
    // whole mega bands of 1024 rows (a rank assembles exactly the rows it applies); mega band m costs ~ m + 1: dealt from the longest down in
    // snake order, so every rank gets the same number of long and short ones
    for (auto &C : S->C) {
      for (int m = 0; m < C.nmb; m++)
        for (int sb = FXM_MB * m; sb < std::min(C.nsb, FXM_MB * (m + 1)); sb++) C.own[sb] = fxm_owner(m, C.nmb, size) == rank;
    }
    return fxs_build_launch(S);
  }
  for (auto &C : S->C) {
    const int nrg = C.ld / 32; // row groups of 32
    C.r0 = (int)((long long)nrg * rank / size) * 32;
    C.r1 = (int)((long long)nrg * (rank + 1) / size) * 32;
  }
  return fxs_build_launch(S);
}

long long fxs_dense_bytes(fx_shared *S) { return S->sym ? (long long)S->owned_bytes : (long long)sizeof(double) * S->wtot; }
double    fxs_apply_flops(fx_shared *S) { return S->sym == 2 ? S->flops : 0.0; } // orbit storage: the GEMM's useful flops per apply
void      fxs_apply_flops_detail(fx_shared *S, double *issued, double *dense) { *issued = S->sym == 2 ? S->flops_issued : 0.0, *dense = S->sym == 2 ? S->flops_dense : 0.0; }
double    fxs_apply_bytes(fx_shared *S) { return S->bytes; }

// the touched dofs of class c, ascending, relative to the block start (the numbering of W_c's rows)
int fxs_class_union(fx_shared *S, int c, int *n_c, int *urel_out)
{
  PMH_ARG(S && c >= 0 && c < S->ncls);
  if (n_c) *n_c = S->C[c].nc;
  if (urel_out && S->C[c].nc) memcpy(urel_out, S->C[c].urel.data(), sizeof(int) * (size_t)S->C[c].nc);
  return PMH_SUCCESS;
}

// Set-up by symmetry: nsym signed permutations of U_c (posmap[g * n_c + c] = position of the image of the c-th touched dof, sign = +-1; operation 0 the
// identity) under which K_c, hence K_c^+, is invariant: W[g p][g c] = sign_g[p] sign_g[c] W[p][c], so ONE K^+ solve serves the whole orbit of a row
// (a cube of Q1 elasticity elements: the 48 signed coordinate permutations -> 48 x fewer solves).  The caller vouches for the invariance (permon_amd
// checks the generators against K); the assembly re-solves a handful of symmetry-filled rows directly and fails if they differ.
int fxs_set_symmetry(fx_shared *S, int c, int nsym, const int *posmap, const signed char *sign)
{
  PMH_ARG(S && c >= 0 && c < S->ncls && nsym >= 1 && posmap && sign);
  if (!S->sym) return pmh_set_error(PMH_ERR_SUP, "pmh_fexplicit_set_class_symmetry: needs the PMH_FX_CLASS_SYM or PMH_FX_CLASS_ORBIT storage");
  S->fxo_ready = 0;
  fxs_class &C  = S->C[c];
  const int  nc = C.nc;
  std::vector<char> seen((size_t)std::max(1, nc));
  for (int g = 0; g < nsym; g++) {
    std::fill(seen.begin(), seen.end(), 0);
    for (int i = 0; i < nc; i++) {
      const int t = posmap[(size_t)g * nc + i];
      if (t < 0 || t >= nc || seen[t] || (sign[(size_t)g * nc + i] != 1 && sign[(size_t)g * nc + i] != -1))
        return pmh_set_error(PMH_ERR_ARG, "pmh_fexplicit_set_class_symmetry: operation %d is not a signed permutation of the %d touched dofs", g, nc);
      if (g == 0 && (t != i || sign[i] != 1)) return pmh_set_error(PMH_ERR_ARG, "pmh_fexplicit_set_class_symmetry: operation 0 must be the identity");
      seen[t] = 1;
    }
  }
  C.nsym = nsym;
  C.h_posmap.assign(posmap, posmap + (size_t)nsym * nc);
  C.h_sign.assign(sign, sign + (size_t)nsym * nc);
  if (C.d_posmap) pmh_free(S->ctx, C.d_posmap), pmh_free(S->ctx, C.d_sign);
  PMH_CHK(pmh_malloc(S->ctx, sizeof(int) * (size_t)std::max(1, nsym * nc), (void **)&C.d_posmap));
  PMH_CHK(pmh_malloc(S->ctx, (size_t)std::max(1, nsym * nc), (void **)&C.d_sign));
  if (nc) {
    PMH_CHK(pmh_memcpy_h2d(S->ctx, C.d_posmap, C.h_posmap.data(), sizeof(int) * (size_t)nsym * nc));
    PMH_CHK(pmh_memcpy_h2d(S->ctx, C.d_sign, C.h_sign.data(), (size_t)nsym * nc));
  }
  return PMH_SUCCESS;
}


static int fxo_gemm(fx_shared *S)
{
  hipStream_t st = S->ctx->stream;
  // several classes on one row tile: one GEMM launch over all their items, then the classes' finishing launches
  const bool  merged = S->merged_tm > 0 && S->nwg_all > 0;
  if (merged) {
#define FXO_LAUNCH_ALL(NI, NWM, TNW)                                                                                                                                                                      \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fxo_gemm16<NI, NWM, true, TNW>), dim3(S->nwg_all), dim3(256), 0, st, (const int *)S->d_items, (const long long *)S->d_wgl, (const int *)S->d_wg,                 \
                     (const int *)(S->d_wg + S->ncls), (const int *)nullptr, 0, (const double *)S->Afund, (const int *)nullptr, (const double *)S->X2, S->cpart, (const int *)S->d_wgfirst_all, \
                     (const int *const *)S->d_coltab_of, (const int *)S->d_zrow_of, (const int *const *)S->d_gidx_of, (const int *)(S->d_zrow_of + S->ncls))
    switch (S->merged_tm + (S->merged_tn == 64 ? 1 : 0) + (S->merged_tnw == 48 ? 1 : 0)) {
    case 194: FXO_LAUNCH_ALL(3, 4, 48); break;
    case 130: FXO_LAUNCH_ALL(2, 4, 48); break;
    case 66: FXO_LAUNCH_ALL(1, 4, 48); break;
    case 144: FXO_LAUNCH_ALL(9, 1, 128); break;
    case 145: FXO_LAUNCH_ALL(9, 1, 64); break;
    case 128: FXO_LAUNCH_ALL(4, 2, 128); break;
    case 129: FXO_LAUNCH_ALL(4, 2, 64); break;
    case 112: FXO_LAUNCH_ALL(7, 1, 128); break;
    case 113: FXO_LAUNCH_ALL(7, 1, 64); break;
    case 96: FXO_LAUNCH_ALL(3, 2, 128); break;
    case 97: FXO_LAUNCH_ALL(3, 2, 64); break;
    case 80: FXO_LAUNCH_ALL(5, 1, 128); break;
    case 81: FXO_LAUNCH_ALL(5, 1, 64); break;
    default: return pmh_set_error(PMH_ERR_STATE, "PMH_FX_CLASS_ORBIT: row tile %d has no 16x16x4 kernel", S->merged_tm);
    }
#undef FXO_LAUNCH_ALL
    if (S->ev_mid_pending >= 0) PMH_HIP(hipEventRecord(S->ev_mid[S->ev_mid_pending], st));
  }
  for (int c = 0; c < S->ncls; c++) { // one launch per class (its own gather-index array and column lists); configs[2] / [3]: one class
    fxs_class &C = S->C[c];
    if (!C.nc) continue;
    const int first = C.item_first, count = C.wg_count; // the class's items are contiguous
    if (!count) continue;
#define FXO_LAUNCH(KERNEL)                                                                                                                                                                              \
  hipLaunchKernelGGL(KERNEL, dim3(count), dim3(256), 0, st, (const int *)(S->d_items + 8 * first), (const long long *)(S->d_wgl + 4 * first), (const int *)S->d_wg, (const int *)(S->d_wg + S->ncls), \
                     (const int *)C.d_coltab, C.nsymp, (const double *)S->Afund, (const int *)C.d_gidx, (const double *)S->X2, S->cpart, (const int *)(S->d_wgfirst + C.wgf_first))
#ifdef FXO_TRACE
    static unsigned long long *d_trace = nullptr;
    static int                 traced  = 0;
    if (!d_trace) {
      PMH_HIP(hipMalloc((void **)&d_trace, sizeof(unsigned long long) * 8 * 64 * 8));
      PMH_HIP(hipMemcpyToSymbol(HIP_SYMBOL(fxo_trace_buf), &d_trace, sizeof(d_trace)));
    }
    PMH_HIP(hipMemsetAsync(d_trace, 0, sizeof(unsigned long long) * 8 * 64 * 8, st));
#endif
    if (merged) {
      // (the class's products were part of the launch above)
    } else if (C.S != FXS_S) { // records of fewer than 8 slots: the table-driven kernel on this class's slice of the items
#define FXO_LAUNCH_T(NI, NWM, TNW)                                                                                                                                                                             \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fxo_gemm16<NI, NWM, true, TNW>), dim3(count), dim3(256), 0, st, (const int *)(S->d_items + 8 * first), (const long long *)(S->d_wgl + 4 * first), (const int *)S->d_wg, \
                     (const int *)(S->d_wg + S->ncls), (const int *)nullptr, 0, (const double *)S->Afund, (const int *)nullptr, (const double *)S->X2, S->cpart, (const int *)(S->d_wgfirst + C.wgf_first),     \
                     (const int *const *)S->d_coltab_of, (const int *)S->d_zrow_of, (const int *const *)S->d_gidx_of, (const int *)(S->d_zrow_of + S->ncls))
      switch (C.tm + (C.tn == 64 ? 1 : 0) + (C.tnw == 48 ? 1 : 0)) {
      case 194: FXO_LAUNCH_T(3, 4, 48); break;
      case 130: FXO_LAUNCH_T(2, 4, 48); break;
      case 66: FXO_LAUNCH_T(1, 4, 48); break;
      case 144: FXO_LAUNCH_T(9, 1, 128); break;
      case 145: FXO_LAUNCH_T(9, 1, 64); break;
      case 128: FXO_LAUNCH_T(4, 2, 128); break;
      case 129: FXO_LAUNCH_T(4, 2, 64); break;
      case 112: FXO_LAUNCH_T(7, 1, 128); break;
      case 113: FXO_LAUNCH_T(7, 1, 64); break;
      case 96: FXO_LAUNCH_T(3, 2, 128); break;
      case 97: FXO_LAUNCH_T(3, 2, 64); break;
      case 80: FXO_LAUNCH_T(5, 1, 128); break;
      case 81: FXO_LAUNCH_T(5, 1, 64); break;
      default: return pmh_set_error(PMH_ERR_STATE, "PMH_FX_CLASS_ORBIT: row tile %d has no 16x16x4 kernel", C.tm);
      }
#undef FXO_LAUNCH_T
    } else {
      switch (C.tm) {
      case 144: FXO_LAUNCH((k_fxo_gemm16<9, 1>)); break;
      case 128: FXO_LAUNCH((k_fxo_gemm16<4, 2>)); break;
      case 112: FXO_LAUNCH((k_fxo_gemm16<7, 1>)); break;
      case 96: FXO_LAUNCH((k_fxo_gemm16<3, 2>)); break;
      case 80: FXO_LAUNCH((k_fxo_gemm16<5, 1>)); break;
      default: return pmh_set_error(PMH_ERR_STATE, "PMH_FX_CLASS_ORBIT: row tile %d has no 16x16x4 kernel", C.tm);
      }
    }
#undef FXO_LAUNCH
    // (one class: configs[2] / [3]; several classes: after the last class's GEMM)
    if (!merged && S->ev_mid_pending >= 0 && c == S->ncls - 1) PMH_HIP(hipEventRecord(S->ev_mid[S->ev_mid_pending], st));
#ifdef FXO_TRACE
    if (++traced == 300) { // one launch in the steady state of the bench
      std::vector<unsigned long long> h(8 * 64 * 8);
      PMH_HIP(hipStreamSynchronize(st));
      PMH_HIP(hipMemcpy(h.data(), d_trace, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
      for (int w = 0; w < 8; w++) {
        fprintf(stderr, "FXO_TRACE workgroup %d (cycles of the shader clock; per chunk: loads issued | products | wait vmcnt | LDS store | barrier | total)\n", w * 37);
        for (int ch = 0; ch < 63; ch++) {
          const unsigned long long *q = &h[((size_t)w * 64 + ch) * 8];
          if (!q[5]) break;
#ifdef FXO_TRACE_FULL
          fprintf(stderr, "  chunk %2d: %6llu | %6llu | %6llu | %6llu | %6llu | %6llu\n", ch, q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], q[5] - q[4], q[5] - q[0]);
#else
          // loads + products | store + barrier
          fprintf(stderr, "  chunk %2d: %6llu | %6llu | %6llu | %6llu | %6llu | %6llu\n", ch, 0ULL, q[2] - q[0], 0ULL, 0ULL, q[5] - q[2], q[5] - q[0]);
#endif
        }
      }
    }
#endif
    if (C.fin_elems > 0 && !S->d_fin_args)
      hipLaunchKernelGGL(k_fxo_fin, dim3((unsigned)((C.fin_elems + PMH_BLOCK - 1) / PMH_BLOCK), C.ngroups), dim3(PMH_BLOCK), 0, st, C.Mp / C.tm, C.tm, C.nsymp, C.nc, (const int *)C.d_fintab,
                         (const int *)C.d_unittab, (const long long *)C.d_finbase, (const int *)C.d_lut, (const int *)C.d_coltab, (const double *)S->cpart, (const signed char *)C.d_use, (const int *)C.d_reppos, (const int *)C.d_posmap, C.xoff, C.ld,
                         S->Y, C.S);
  }
  if (S->d_fin_args && S->fin_nbx > 0) // after ALL classes' products
    hipLaunchKernelGGL(k_fxo_fin_all, dim3((unsigned)S->fin_nbx, (unsigned)S->fin_ngroups, (unsigned)S->ncls), dim3(PMH_BLOCK), 0, st, (const fxo_fin_args *)S->d_fin_args, (const double *)S->cpart, S->Y);
  PMH_HIP(hipGetLastError());
  return PMH_SUCCESS;
}

int fxs_assemble(fx_shared *S, pmh_matinv solver, int nslots, const int *slot_class, double rtol, int max_it, long long *n_solves)
{
  PMH_ARG(S && solver && nslots >= 1 && (solver->nblocks == nslots || solver->nblocks * PMH_MV_R == nslots) && slot_class);
  pmh_ctx                       ctx = S->ctx;
  if (S->sym == 2) PMH_CHK(fxo_prepare(S));
  pmh_asm_solver                A;
  struct closer {
    pmh_asm_solver &a;
    ~closer() { a.close(); }
  } closer_{A};
  PMH_CHK(A.open(solver, nslots));
  std::vector<std::vector<int>> cslots(S->ncls), todo(S->ncls);
  std::vector<std::vector<int>> rep_of(S->ncls), op_of(S->ncls), check(S->ncls);
  std::vector<std::map<int, std::vector<int>>> members(S->ncls); // representative row -> the owned rows of its orbit
  for (int s = 0; s < nslots; s++)
    if (slot_class[s] >= 0 && slot_class[s] < S->ncls) cslots[slot_class[s]].push_back(s);
  int nbatch = 0;
  for (int c = 0; c < S->ncls; c++) {
    fxs_class &C = S->C[c];
    if (C.nc == 0) continue;
    if (cslots[c].empty()) return pmh_set_error(PMH_ERR_ARG, "pmh_fexplicit_assemble: no solver slot for block class %d", c);
    for (int s : cslots[c])
      if (A.rows(s) != C.nloc) return pmh_set_error(PMH_ERR_ARG, "pmh_fexplicit_assemble: slot %d has %d rows, class %d blocks have %d", s, A.rows(s), c, C.nloc);
    if (S->sym == 2) {
      // orbit storage: the owned representatives; self-check: rows of their orbits reached by a non-trivial operation
      for (int pl = 0; pl < C.m1 - C.m0; pl++) todo[c].push_back(C.reps[C.m0 + pl]);
      std::vector<int> filled;
      for (int r = 0; r < C.nc; r++)
        if (C.op_of[r] != 0) {
          const int k = (int)(std::lower_bound(C.reps.begin(), C.reps.end(), C.rep_of[r]) - C.reps.begin());
          if (k >= C.m0 && k < C.m1) filled.push_back(r);
        }
      const int nchk = (int)std::min(filled.size(), cslots[c].size());
      for (int i = 0; i < nchk; i++) check[c].push_back(filled[(size_t)((long long)filled.size() * (2 * i + 1) / (2 * nchk))]);
    } else if (S->sym && C.nsym > 1) {
      // orbits of the rows under the class's symmetries: rep_of / op_of, one solve per orbit that has a row in this rank's super bands
      rep_of[c].assign((size_t)C.nc, -1), op_of[c].assign((size_t)C.nc, 0);
      for (int p = 0; p < C.nc; p++) {
        if (rep_of[c][p] >= 0) continue;
        for (int g = 0; g < C.nsym; g++) {
          const int r = C.h_posmap[(size_t)g * C.nc + p];
          if (rep_of[c][r] < 0) rep_of[c][r] = p, op_of[c][r] = g;
        }
      }
      std::vector<char> need((size_t)C.nc, 0);
      for (int r = 0; r < C.nc; r++)
        if (C.own[r / FXM_RS]) need[rep_of[c][r]] = 1, members[c][rep_of[c][r]].push_back(r);
      for (int p = 0; p < C.nc; p++)
        if (need[p]) todo[c].push_back(p);
      // self-check: up to one batch of symmetry-filled owned rows, spread over the range, solved directly at the end
      std::vector<int> filled;
      for (int r = 0; r < C.nc; r++)
        if (C.own[r / FXM_RS] && op_of[c][r] != 0) filled.push_back(r);
      const int nchk = (int)std::min(filled.size(), cslots[c].size());
      for (int i = 0; i < nchk; i++) check[c].push_back(filled[(size_t)((long long)filled.size() * (2 * i + 1) / (2 * nchk))]);
    } else if (S->sym) {
      for (int p = 0; p < C.nc; p++)
        if (C.own[p / FXM_RS]) todo[c].push_back(p); // the rows of this rank's super bands
    } else
      for (int p = C.r0; p < std::min(C.r1, C.nc); p++) todo[c].push_back(p); // the rows of this rank's stripe
    nbatch = std::max(nbatch, (int)((todo[c].size() + cslots[c].size() - 1) / cslots[c].size()));
  }
  double      *rhs, *sol;
  int         *d_idx, *h_idx;
  const size_t nsol = A.len();
  PMH_CHK(pmh_malloc(ctx, sizeof(double) * nsol, (void **)&rhs));
  PMH_CHK(pmh_malloc(ctx, sizeof(double) * nsol, (void **)&sol));
  PMH_CHK(pmh_malloc(ctx, sizeof(int) * nslots, (void **)&d_idx));
  PMH_HIP(hipHostMalloc((void **)&h_idx, sizeof(int) * nslots * 2, hipHostMallocDefault));
  PMH_CHK(pmh_memset(ctx, rhs, 0, sizeof(double) * nsol));
  double old_rtol, old_atol;
  int    old_maxit;
  PMH_CHK(pmh_matinv_get_tolerances(solver, &old_rtol, &old_atol, &old_maxit));
  PMH_CHK(pmh_matinv_set_tolerances(solver, rtol, 1e-300, max_it > 0 ? max_it : old_maxit));
  int              rc = PMH_SUCCESS;
  std::vector<int> prow(nslots);
  for (int k = 0; k < nbatch && !rc; k++) {
    int *hh = h_idx + (k & 1) * nslots;
    for (int s = 0; s < nslots; s++) hh[s] = -1, prow[s] = -1;
    for (int c = 0; c < S->ncls; c++)
      for (size_t t = 0; t < cslots[c].size(); t++) {
        const size_t j = (size_t)k * cslots[c].size() + t;
        if (j < todo[c].size()) {
          const int s = cslots[c][t];
          prow[s]     = todo[c][j];
          hh[s]       = A.rhs_index(s, S->C[c].urel[prow[s]]);
        }
      }
    if (hipMemcpyAsync(d_idx, hh, sizeof(int) * nslots, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
      rc = pmh_set_error(PMH_ERR_HIP, "pmh_fexplicit_assemble: index upload failed");
      break;
    }
    hipLaunchKernelGGL(k_fxs_set_entries, dim3((nslots + 63) / 64), dim3(64), 0, ctx->stream, nslots, (const int *)d_idx, 1.0, rhs);
    if ((rc = A.solve(rhs, sol))) break;
    if (A.hit_the_limit()) {
      rc = pmh_set_error(PMH_ERR_STATE, "pmh_fexplicit_assemble: a set-up solve of batch %d did not reach rtol %.1e within %d iterations of the inner KSP", k, rtol, solver->max_it);
      break;
    }
    hipLaunchKernelGGL(k_fxs_set_entries, dim3((nslots + 63) / 64), dim3(64), 0, ctx->stream, nslots, (const int *)d_idx, 0.0, rhs);
    for (int s = 0; s < nslots; s++) {
      if (prow[s] < 0) continue;
      (*n_solves)++;
      const fxs_class &C = S->C[slot_class[s]];
      if (S->sym == 2) {
        const int pl = (int)(std::lower_bound(C.reps.begin(), C.reps.end(), prow[s]) - C.reps.begin()) - C.m0;
        hipLaunchKernelGGL(k_fxo_store_row, dim3(std::max(1, std::min(64, (C.nc + PMH_BLOCK - 1) / PMH_BLOCK))), dim3(PMH_BLOCK), 0, ctx->stream, C.reprow[pl], C.tm, C.nc, C.nkc, (const int *)C.d_urel,
                           (const int *)C.d_kinv, A.sol_of(sol, s), S->Afund + C.aoff);
      } else if (S->sym && C.nsym > 1) {
        const int p = prow[s];
        for (int r : members[slot_class[s]][p]) {
          const int g = op_of[slot_class[s]][r], sb = r / FXM_RS;
          hipLaunchKernelGGL(k_fxs_extract_symg, dim3(std::max(1, std::min(64, (C.nc + PMH_BLOCK - 1) / PMH_BLOCK))), dim3(PMH_BLOCK), 0, ctx->stream, r, C.nc, (double)C.h_sign[(size_t)g * C.nc + p],
                             (const int *)C.d_urel, A.sol_of(sol, s), (const int *)(C.d_posmap + (size_t)g * C.nc), (const signed char *)(C.d_sign + (size_t)g * C.nc),
                             S->Wbase + C.woff + (long long)FXM_RS * FXM_RS * ((long long)sb * (sb + 1) / 2));
        }
      } else if (S->sym) {
        const int sb = prow[s] / FXM_RS;
        hipLaunchKernelGGL(k_fxs_extract_sym, dim3(std::max(1, std::min(64, (prow[s] + PMH_BLOCK) / PMH_BLOCK))), dim3(PMH_BLOCK), 0, ctx->stream, prow[s], (const int *)C.d_urel, A.sol_of(sol, s),
                           S->Wbase + C.woff + (long long)FXM_RS * FXM_RS * ((long long)sb * (sb + 1) / 2));
      } else
      hipLaunchKernelGGL(k_fxs_extract, dim3(std::max(1, std::min(64, (C.nc + PMH_BLOCK - 1) / PMH_BLOCK))), dim3(PMH_BLOCK), 0, ctx->stream, C.nc, (const int *)C.d_urel, A.sol_of(sol, s),
                         S->Wbase + C.woff + (long long)prow[s] * C.ld); // row p of W_c = column p (K^+ symmetric)
    }
    if (hipGetLastError() != hipSuccess) rc = pmh_set_error(PMH_ERR_HIP, "pmh_fexplicit_assemble: launch failed in batch %d", k);
  }
  // self-check of the set-up by symmetry: a few symmetry-filled rows against their direct solves
  bool any_check = false;
  for (int c = 0; c < S->ncls; c++) any_check = any_check || !check[c].empty();
  if (!rc && any_check) rc = pmh_sync(ctx); // the pinned index buffer is reused below
  if (!rc && any_check) {
    int *hh = h_idx;
    for (int s = 0; s < nslots; s++) hh[s] = -1, prow[s] = -1;
    for (int c = 0; c < S->ncls; c++)
      for (size_t t = 0; t < check[c].size(); t++) {
        const int s = cslots[c][t];
        prow[s] = check[c][t], hh[s] = A.rhs_index(s, S->C[c].urel[prow[s]]);
      }
    double *d_out = nullptr, h_out[2 * 64];
    rc = pmh_malloc(ctx, sizeof(double) * 2 * 64, (void **)&d_out);
    if (!rc && hipMemcpyAsync(d_idx, hh, sizeof(int) * nslots, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = pmh_set_error(PMH_ERR_HIP, "pmh_fexplicit_assemble: index upload failed");
    if (!rc) {
      hipLaunchKernelGGL(k_fxs_set_entries, dim3((nslots + 63) / 64), dim3(64), 0, ctx->stream, nslots, (const int *)d_idx, 1.0, rhs);
      rc = A.solve(rhs, sol);
    }
    for (int s = 0; s < nslots && !rc; s++) {
      if (prow[s] < 0) continue;
      (*n_solves)++;
      const fxs_class &C  = S->C[slot_class[s]];
      const int        r = prow[s], sb = r / FXM_RS;
      int              nb = std::max(1, std::min(64, (r + PMH_BLOCK) / PMH_BLOCK));
      if (S->sym == 2) {
        const int g = C.op_of[r], p = C.rep_of[r], pl = (int)(std::lower_bound(C.reps.begin(), C.reps.end(), p) - C.reps.begin()) - C.m0;
        nb          = std::max(1, std::min(64, (C.nc + PMH_BLOCK - 1) / PMH_BLOCK));
        hipLaunchKernelGGL(k_fxo_check_row, dim3(nb), dim3(PMH_BLOCK), 0, ctx->stream, C.reprow[pl], C.tm, C.nc, C.nkc, (double)C.h_sign[(size_t)g * C.nc + p], (const int *)C.d_urel, (const int *)C.d_kinv, A.sol_of(sol, s),
                           (const int *)(C.d_posmap + (size_t)g * C.nc), (const signed char *)(C.d_sign + (size_t)g * C.nc), (const double *)(S->Afund + C.aoff), d_out);
      } else
      hipLaunchKernelGGL(k_fxs_check_row, dim3(nb), dim3(PMH_BLOCK), 0, ctx->stream, r, (const int *)C.d_urel, A.sol_of(sol, s),
                         (const double *)(S->Wbase + C.woff + (long long)FXM_RS * FXM_RS * ((long long)sb * (sb + 1) / 2)), d_out);
      if ((rc = pmh_memcpy_d2h(ctx, h_out, d_out, sizeof(double) * 2 * nb))) break;
      double d = 0.0, m = 0.0;
      for (int i = 0; i < nb; i++) d = std::max(d, h_out[2 * i]), m = std::max(m, h_out[2 * i + 1]);
      if (!(d <= 1e-8 * m))
        rc = pmh_set_error(PMH_ERR_STATE, "pmh_fexplicit_assemble: row %d of class %d filled by symmetry differs from its direct solve by %.2e (relative to the row's largest entry): the operations handed to pmh_fexplicit_set_class_symmetry are not symmetries of K^+", r, slot_class[s], d / m);
    }
    if (d_out) pmh_free(ctx, d_out);
  }
  if (!rc) rc = pmh_sync(ctx);
  pmh_matinv_set_tolerances(solver, old_rtol, old_atol, old_maxit);
  pmh_free(ctx, rhs), pmh_free(ctx, sol), pmh_free(ctx, d_idx);
  (void)hipHostFree(h_idx);
  return rc;
}

static int fxs_gemm(fx_shared *S)
{
  if (S->sym == 2 && !S->fxo_ready) return pmh_set_error(PMH_ERR_STATE, "PMH_FX_CLASS_ORBIT: apply before the assembly");
  if (!S->nwg) return PMH_SUCCESS;
  hipStream_t st    = S->ctx->stream;
  const bool  timed = S->ev_on && (size_t)(2 * S->ev_used + 2) <= S->ev.size();
  if (timed) PMH_HIP(hipEventRecord(S->ev[2 * S->ev_used], st));
  const long long stride = std::max(16LL, S->nX);
  if (S->sym == 2) {
    S->ev_mid_pending = timed ? S->ev_used : -1;
    PMH_CHK(fxo_gemm(S));
    S->ev_mid_pending = -1;
  } else if (S->sym) {
    hipLaunchKernelGGL(k_fxs_symm8, dim3(S->nwg), dim3(FXM_THREADS), 0, st, (const int *)S->d_wg, (const int *)S->d_items, (const long long *)S->d_wgl, (const int *)S->d_ld, (const long long *)S->d_xoff,
                       (const double *)S->Wbase, (const double *)S->X, S->part, stride, S->pt);
    for (auto &C : S->C)
      hipLaunchKernelGGL(k_fxs_symfin, dim3((unsigned)(((long long)C.ld * FXS_S / 2 * 8 + PMH_BLOCK - 1) / PMH_BLOCK), C.ngroups), dim3(PMH_BLOCK), 0, st, C.ld, C.nmb, C.nown, (const int *)C.d_nseg, (const int *)C.d_ownfirst,
                         (const long long *)C.d_ptoff, C.xoff, stride, (const double *)S->part, (const double *)S->pt, S->Y);
  } else {
  hipLaunchKernelGGL(k_fxs_gemm8, dim3(S->nwg), dim3(PMH_BLOCK), 0, st, (const int *)S->d_wg, (const int *)S->d_ld, (const long long *)S->d_woff, (const long long *)S->d_xoff, (const double *)S->Wbase,
                     (const double *)S->X, S->part, stride);
  hipLaunchKernelGGL(k_fxs_fin, dim3((unsigned)((S->nX / 2 + PMH_BLOCK - 1) / PMH_BLOCK)), dim3(PMH_BLOCK), 0, st, S->nX, S->nseg, stride, (const double *)S->part, S->Y);
  }
  if (timed) {
    PMH_HIP(hipEventRecord(S->ev[2 * S->ev_used + 1], st));
    S->ev_used++;
  }
  PMH_HIP(hipGetLastError());
  return PMH_SUCCESS;
}

// X (position, slot) -> X2 (position, +-, slot): only for the dense kernel alone on a multivector handed in (pmh_fexplicit_dense_mult); the operator fills X2
// by its own gluing
__global__ __launch_bounds__(PMH_BLOCK) void k_fxo_signed_copy(long long n, int nslot, const double *__restrict__ X, double *__restrict__ X2)
{
  for (long long i = (long long)blockIdx.x * PMH_BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * PMH_BLOCK) {
    const long long p = i / nslot, sl = i % nslot;
    const double    v = X[i];
    X2[2 * p * nslot + sl] = v, X2[(2 * p + 1) * nslot + sl] = -v;
  }
}

int fxs_apply(fx_shared *S, const double *lambda, double *y)
{
  if (S->sym == 2) PMH_CHK(pmh_gluing_mult(S->Bc2, lambda, S->X2));
  else PMH_CHK(pmh_gluing_mult(S->Bc, lambda, S->X));
  PMH_CHK(fxs_gemm(S));
  return pmh_gluing_mult_transpose(S->Bc, S->Y, y); // ends with the all-reduce on several GPUs
}

int fxs_stages(fx_shared *S, pmh_csr *gather, double **mid_in, pmh_csr *scatter, const double **mid_out)
{
  if (S->sym == 2) *gather = S->Bc2->Bt, *mid_in = S->X2;
  else *gather = S->Bc->Bt, *mid_in = S->X;
  *scatter = S->Bc->B, *mid_out = S->Y;
  return PMH_SUCCESS;
}
int fxs_mid(fx_shared *S) { return fxs_gemm(S); }

// the dense kernel alone (tests, tuning): Y = blockdiag(W_c) X on the multivectors as they stand
int fxs_dense(fx_shared *S)
{
  if (S->sym == 2 && S->nX > 0) {
    for (const fxs_class &C : S->C) { // class by class: the records of a class have C.S slots
      const long long n = (long long)C.ngroups * C.ld * C.S;
      if (!n) continue;
      hipLaunchKernelGGL(k_fxo_signed_copy, dim3((unsigned)std::min<long long>(4096, (n + PMH_BLOCK - 1) / PMH_BLOCK)), dim3(PMH_BLOCK), 0, S->ctx->stream, n, C.S, (const double *)(S->X + C.xoff), S->X2 + 2 * C.xoff);
    }
    PMH_HIP(hipGetLastError());
  }
  return fxs_gemm(S);
}
long long fxs_multivector_length(fx_shared *S) { return S->nX; }
double   *fxs_X(fx_shared *S) { return S->X; }
double   *fxs_Y(fx_shared *S) { return S->Y; }

int fxs_fill_pattern(fx_shared *S, int byte)
{
  if (S->sym == 2) {
    PMH_CHK(fxo_prepare(S));
    PMH_HIP(hipMemsetAsync(S->Afund, byte, sizeof(double) * (size_t)S->afund_tot, S->ctx->stream));
    return pmh_sync(S->ctx);
  }
  PMH_HIP(hipMemsetAsync(S->Wbase, byte, sizeof(double) * (size_t)S->wtot, S->ctx->stream));
  return pmh_sync(S->ctx);
}

// W_b = W_c[pos_b, pos_b] on the host (tests); gamma: the block's touched dofs (rank-local primal indices, ascending)
int fxs_get_block(fx_shared *S, int b, int n, const int *gamma, double *out_host)
{
  const fxs_class    &C = S->C[S->cls[b]];
  if (S->sym == 2) {
    // orbit storage (tests: small classes, all representatives on this rank): row r = g p rebuilt from its representative's row
    if (!S->fxo_ready || C.m0 != 0 || C.m1 != C.M_all) return pmh_set_error(PMH_ERR_STATE, "pmh_fexplicit_get_block: the orbit storage holds only this rank's representatives");
    std::vector<double> T((size_t)C.Mp * C.ldk), row((size_t)C.nc);
    PMH_HIP(hipMemcpy(T.data(), S->Afund + C.aoff, sizeof(double) * T.size(), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) {
      const int r = C.pos[gamma[i] - S->K->rowstart[b]], p = C.rep_of[r], g = C.op_of[r], pl = (int)(std::lower_bound(C.reps.begin(), C.reps.end(), p) - C.reps.begin());
      const double  sp = (double)C.h_sign[(size_t)g * C.nc + p];
      const int     pr = C.reprow[pl]; // the representative's row of A
      const double *ab = T.data() + (size_t)(pr / C.tm) * C.nkc * (FXO_TK * C.tm) + pr % C.tm;
      for (int cc = 0; cc < C.nc; cc++) {
        const int k = C.kinv[cc];
        row[C.h_posmap[(size_t)g * C.nc + cc]] = sp * (double)C.h_sign[(size_t)g * C.nc + cc] * ab[(size_t)(k / FXO_TK) * (FXO_TK * C.tm) + (k % FXO_TK) * C.tm];
      }
      for (int k = 0; k < n; k++) out_host[(size_t)i * n + k] = row[C.pos[gamma[k] - S->K->rowstart[b]]];
    }
    return PMH_SUCCESS;
  }
  if (S->sym) {
    // the class's tiles on the host (tests: small classes), W[p][c] from the stored lower triangle
    const long long     len = (long long)FXM_RS * FXM_RS * ((long long)C.nsb * (C.nsb + 1) / 2);
    std::vector<double> T((size_t)len);
    PMH_HIP(hipMemcpy(T.data(), S->Wbase + C.woff, sizeof(double) * (size_t)len, hipMemcpyDeviceToHost));
    auto at = [&](int p, int c) {
      const int hi = std::max(p, c), lo = std::min(p, c), sb = hi / FXM_RS, Il = (hi % FXM_RS) / 16, r = hi & 15, q = r >> 2, l = 16 * (r & 3) + (lo & 15);
      const double v = T[(size_t)((long long)FXM_RS * FXM_RS * ((long long)sb * (sb + 1) / 2) + ((long long)(lo >> 4) * FXM_RT + Il) * 256 + (q >> 1) * 128 + 2 * l + (q & 1))];
      return hi == lo ? 2.0 * v : v;
    };
    for (int i = 0; i < n; i++)
      for (int k = 0; k < n; k++) out_host[(size_t)i * n + k] = at(C.pos[gamma[i] - S->K->rowstart[b]], C.pos[gamma[k] - S->K->rowstart[b]]);
    return PMH_SUCCESS;
  }
  std::vector<double> row((size_t)std::max(1, C.ld));
  for (int i = 0; i < n; i++) {
    const int p = C.pos[gamma[i] - S->K->rowstart[b]];
    PMH_HIP(hipMemcpy(row.data(), S->Wbase + C.woff + (long long)p * C.ld, sizeof(double) * (size_t)C.ld, hipMemcpyDeviceToHost));
    for (int k = 0; k < n; k++) out_host[(size_t)i * n + k] = row[C.pos[gamma[k] - S->K->rowstart[b]]];
  }
  return PMH_SUCCESS;
}

int fxs_timing_enable(fx_shared *S, int max_launches)
{
  while ((int)S->ev.size() < 2 * max_launches) {
    hipEvent_t e;
    PMH_HIP(hipEventCreate(&e));
    S->ev.push_back(e);
  }
  while ((int)S->ev_mid.size() < max_launches) {
    hipEvent_t e;
    PMH_HIP(hipEventCreate(&e));
    S->ev_mid.push_back(e);
  }
  S->ev_on = max_launches > 0, S->ev_used = 0;
  return PMH_SUCCESS;
}

int fxs_timing_get(fx_shared *S, int *launches, double *total_ms, double *first_kernel_ms)
{
  PMH_CHK(pmh_sync(S->ctx));
  double tot = 0.0, first = 0.0;
  for (int i = 0; i < S->ev_used; i++) {
    float ms = 0.f;
    PMH_HIP(hipEventElapsedTime(&ms, S->ev[2 * i], S->ev[2 * i + 1]));
    tot += ms;
    if (S->sym == 2) {
      PMH_HIP(hipEventElapsedTime(&ms, S->ev[2 * i], S->ev_mid[i]));
      first += ms;
    }
  }
  *launches = S->ev_used, *total_ms = tot;
  if (first_kernel_ms) *first_kernel_ms = S->sym == 2 ? first : tot; // orbit storage: the GEMM kernel alone (the rest is k_fxo_fin)
  return PMH_SUCCESS;
}
