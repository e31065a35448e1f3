// Class-shared explicit local dual operators: the storage structures of fshared.hip / fshared_plan.hip / fshared_kernels.h (internal)
#pragma once
#include <algorithm>
#include <chrono>
#include <map>
#include <cmath>

#include "feti_internal.h"
#include "fshared.h"
#include "pmh_internal.h"
#include "reduce.h"

typedef double dbl2 __attribute__((ext_vector_type(2)));

#define FXS_S 8    // right-hand sides per pass (blocks per group)
#define FXS_PAD 128

struct fxs_class {
  std::vector<int> blocks; // blocks of the class, ascending: slot = index % 8, group = index / 8
  std::vector<int> urel;   // sorted union of the touched dofs, relative to the block start
  std::vector<int> pos;    // relative dof -> position in urel (-1)
  int              nloc = 0, nc = 0, ld = 0, ngroups = 0, r0 = 0, r1 = 0;
  long long        woff = 0, xoff = 0;
  int             *d_urel = nullptr;
  // symmetric tile storage (fx_shared::sym): super bands of FXM_RS rows
  int              nsb = 0, nmb = 0; // mega bands of FXM_MB super bands
  std::vector<char> own;    // this rank applies / assembles super band sb (whole mega bands)
  long long        ptoff = 0, ptsize = 0; // transposed partial sums: ptoff + group * ptsize + ptm[mega band] + position * 8 + slot
  std::vector<long long> ptm;
  int             *d_nseg = nullptr;      // items (= segments of the direct sums) per (group, mega band) (0: not owned)
  long long       *d_ptoff = nullptr;
  int             *d_ownfirst = nullptr, nown = 0;
  // set-up by symmetry (fxs_set_symmetry): nsym signed permutations of U_c under which K_c^+ is invariant, op 0 = identity
  int                      nsym = 0;
  std::vector<int>         h_posmap; // [nsym][nc]: position of the image of the c-th touched dof
  std::vector<signed char> h_sign;   // [nsym][nc]: +-1
  int                     *d_posmap = nullptr;
  signed char             *d_sign = nullptr;
  // orbit storage (fx_shared::sym == 2): only the rows of W_c of the orbit representatives are kept, see the FXO section
  std::vector<int> reps, rep_of, op_of; // all representatives (positions, ascending); per row: its representative's position and the operation that reaches it
  // tm: row tile of the GEMM (fxo_row_tile); tnw = 48: the 48-column kernel (one-block classes)
  int              M_all = 0, m0 = 0, m1 = 0, Mp = 0, ldk = 0, nkc = 0, nsymp = 0, tm = 128, tnw = 0;
  long long        aoff = 0, coff = 0;  // offsets of the class in Afund / cpart
  int             *d_gidx = nullptr, *d_reppos = nullptr;
  // output pruning of the orbit GEMM: block (group, slot) touches only part of U_c, so row g p of Y is needed for the slots that touch it only.  Per (group,
  // row tile) the columns (operation << 3 | slot) some row of the tile needs, padded to 128 with -1; the representatives are ordered by their need pattern
  // (rows of A)
  std::vector<char> tmask;              // [ngroups][nc][8]
  std::vector<int>  reprow;             // representative index (in reps) -> row of A / cpart
  // fintab per (group, row tile): coltab offset, padded columns, first element of the tile in the group's numbering
  int              *d_coltab = nullptr, *d_fintab = nullptr;
  long long        *d_finbase = nullptr;                     // per (group, row tile): offset of split 0 in cpart
  // the class's items; its workgroups (slice of fx_shared::d_wgfirst)
  int               item_first = 0, item_count = 0, fin_elems = 0, wgf_first = 0, wg_count = 0;
  // column tile of the class's GEMM: 64 when no (group, row tile) lists more than 64 columns (a class of ONE block lists at most its 48 operations)
  int               tn = 128;
  // orbit storage: slots of a multivector record = the smallest power of two >= the class's blocks (<= 8): a class of ONE block gathers 8-byte records, not a
  // 64-byte line with seven zeros
  int               S = 8;
  signed char     *d_use = nullptr;
  // k segments of the orbit GEMM: the positions (= the k index of the product) are grouped by WHICH columns have a structural non-zero of B there (block
  // (group, slot) does not touch g c => B[c][(g, slot)] = 0), the rows of B are permuted segment after segment (each padded to whole chunks) and a (row tile,
  // segment) multiplies only the columns that are non-zero on the segment (fxo_prepare).  kinv: position -> row of B / column of the pre-tiled A
  std::vector<int> kinv;
  // unittab per (group, row tile, segment) unit: offset of its look-up table, padded columns, splits, 0
  int             *d_kinv = nullptr, *d_unittab = nullptr, *d_lut = nullptr;
  int              nseg = 1;
};

struct fx_shared {
  pmh_ctx                ctx;
  pmh_gluing             B;
  pmh_blockdiag          K;
  int                    nb, ncls;
  std::vector<int>       cls; // class of every block
  std::vector<fxs_class> C;
  pmh_gluing             Bc = nullptr;
  // orbit storage: the multivector the GEMM gathers from holds every entry TWICE, [position][+x | -x][slot] -- the gather index (position << 1 | negative)
  // addresses the signed value directly, no sign is applied to a loaded value inside the GEMM (a use of the loaded value in front of the products made every
  // wave wait for all of a chunk's global loads before its first MFMA).  Bc2: the gluing that fills it (every leaf of Bc twice, the second with the opposite
  // sign); Bc stays for B Y on the way back
  pmh_gluing             Bc2 = nullptr;
  double                *X2  = nullptr;
  double                *Wbase = nullptr, *X = nullptr, *Y = nullptr;
  long long              nX = 0, wtot = 0;
  int                   *d_wg = nullptr; // launch table: (class, group, first column, segment, first row, one-past-last row) per workgroup
  int                    nwg = 0, nseg = 0;
  double                *part = nullptr; // [nseg][nX] segment sums of k_fxs_gemm8
  long long              part_cap = 0;
  int                   *d_ld = nullptr;
  long long             *d_woff = nullptr, *d_xoff = nullptr;
  double                 bytes = 0.0;
  std::vector<hipEvent_t> ev;
  int                    ev_used = 0, ev_on = 0;
  std::vector<hipEvent_t> ev_mid; // orbit storage: after the GEMM kernel, before k_fxo_fin (the first kernel's own duration)
  int                    ev_mid_pending = -1;
  // symmetric tile storage (PMH_FX_CLASS_SYM): the lower block-triangle of W_c in 16 x 16 tiles, k_fxs_symm8 (fp64 MFMA) + k_fxs_symfin
  int                    sym = 0, segj = 0;
  long long             *d_wgl = nullptr; // per item: offset of its class's tiles, offset of its transposed partial sums
  int                   *d_items = nullptr, *d_wgfirst = nullptr;
  // several classes on the same row tile: ONE launch over all their work items (fxo_gemm): workgroup -> items with global item numbers, per-class pointer
  // tables
  int                   *d_wgfirst_all = nullptr, *d_zrow_of = nullptr, nwg_all = 0, merged_tm = 0, merged_tn = 128, merged_tnw = 0;
  const int            **d_coltab_of = nullptr, **d_gidx_of = nullptr;
  void                  *d_fin_args = nullptr; // fxo_fin_args per class: the classes' finishing launches as one (k_fxo_fin_all)
  int                    fin_nbx = 0, fin_ngroups = 0;
  double                *pt = nullptr;
  long long              pt_tot = 0;
  double                 owned_bytes = 0.0;
  // orbit storage
  double                *Afund = nullptr, *cpart = nullptr;
  long long              afund_tot = 0, cpart_cap = 0;
  int                    fxo_ready = 0, fxo_S = 1, stripe_rank = 0, stripe_size = 0;
  // listed columns x valid rows; padded tiles; every (representative, operation, block)
  double                 flops = 0.0, flops_issued = 0.0, flops_dense = 0.0;
};

// the orbit GEMM's tiles (fshared_kernels.h) as the plan needs them
#define FXO_TM 128
#define FXO_TN 128
#define FXO_TK 16

// several classes in ONE launch (blockIdx.z = class; grid.x / grid.y = the largest class's): every class's parameters from a device table
struct fxo_fin_args {
  int              ntile, tm, nsymp, nc, ld, nslot, nbx, ngroups;
  const int       *fintab, *unittab, *lut, *coltab, *reppos, *posmap;
  const long long *unitbase;
  const signed char *use;
  long long        xbase0;
};

// fshared_plan.hip (host only)
int  fxo_row_tile(int M); // row tile of a class with M orbit representatives
int  fxo_prepare(fx_shared *S);
