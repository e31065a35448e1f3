// Feasibility micro-benchmark of the symmetry-reduced dense apply: C[p][(g,s)] = sum_c A[p][c] * sgn(g,c) * X[posmap_g[c]][s]
// (M orbit representatives x N = nsym * 8 columns x K = n_c) on the fp64 matrix instruction v_mfma_f64_4x4x4_4b_f64, split-K, B gathered from X.
// hipcc --offload-arch=gfx950 -O3 -DNWM=4 -DNWN=2 -DTK=16 -o /tmp/orbit_gemm scripts/micro/orbit_gemm.hip && /tmp/orbit_gemm [M N_ops K S]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef double dbl2 __attribute__((ext_vector_type(2)));
#ifndef NWM
#define NWM 4 // waves along M (64 rows each)
#endif
#ifndef NWN
#define NWN 2 // waves along N (64 columns each)
#endif
#ifndef TK
#define TK 16
#endif
#ifndef WTN
#define WTN 64 // columns of a wave's tile
#endif
#ifdef ORIENT4 // a wave's tile = WR rows (4 per A operand, the same in the 4 blocks of the instruction) x 64 columns (16 per B operand)
#ifndef WR
#define WR 60
#endif
#define TM (WR * NWM)
#else
#define TM (64 * NWM)
#endif
#define TN (WTN * NWN)
#define NT (64 * NWM * NWN)
#define LDA (TM + 16)
#ifdef ORIENT4
#define LDB (TN + 16)
#else
#define LDB (TN + 4)
#endif
static __device__ __forceinline__ double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

// A pre-tiled: block (mt, kc) = TK k x TM rows, k-major; gidx[g][k] = (position << 1) | negative; X[pos][8]; Cpart[split][Mp][N]
#ifdef NVGPR
#define NVGPR_ATTR __attribute__((amdgpu_num_vgpr(NVGPR)))
#else
#define NVGPR_ATTR
#endif
#ifndef MINB
#define MINB 1
#endif
__global__ __launch_bounds__(NT, MINB) NVGPR_ATTR void k_gemm(int Mt, int Nt, int S, int nkc, const double *__restrict__ A, const int *__restrict__ gidx, int ldk, const double *__restrict__ X,
                                                double *__restrict__ Cpart, int Mp, int N, const int *__restrict__ wmap)
{
  __shared__ double As[2][TK][LDA];
  __shared__ double Bs[2][TK][LDB];
  const int wgi = wmap ? __builtin_amdgcn_readfirstlane(wmap[blockIdx.x]) : blockIdx.x, s = wgi % S, nt = (wgi / S) % Nt, mt = wgi / (S * Nt);
  const int kc0 = (int)((long long)nkc * s / S), kc1 = (int)((long long)nkc * (s + 1) / S);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave / NWN, wn = wave % NWN;
  constexpr int NEA = (TK * TM / 2 + NT - 1) / NT; // dbl2 loads of A per thread and chunk
  constexpr int KPB = NT / TN;          // k rows of B staged per pass
  constexpr int NEB = TK / KPB;
  static_assert(NEA >= 1 && NEB >= 1 && TK % KPB == 0, "tile / thread shape");
  const int  col = t % TN, kb = t / TN;
  const int *gp = gidx + (size_t)(nt * (TN / 8) + (col >> 3)) * ldk;
  const int  sl = col & 7;
#ifdef ORIENT4
  constexpr int NA = WR / 4, NB = WTN / 16;
#else
  constexpr int NA = 4, NB = WTN / 4;
#endif
  double     acc[NA][NB];
#pragma unroll
  for (int i = 0; i < NA; i++)
#pragma unroll
    for (int j = 0; j < NB; j++) acc[i][j] = 0.0;
  dbl2   ar[NEA];
  double br[NEB];
  int    gn[NEB];
#if defined(NOALOAD) || defined(NOGATHER)
  for (int e = 0; e < NEA; e++) ar[e] = dbl2{1.0 + t, 2.0};
  for (int e = 0; e < NEB; e++) br[e] = 0.5 + t, gn[e] = 0;
#endif
  auto loadA = [&](int kc, dbl2 (&ar)[NEA]) {
#ifndef NOALOAD // knock-out: no HBM stream of A (timing only)
    const double *blk = A + ((size_t)mt * nkc + kc) * (TK * TM);
#pragma unroll
    for (int e = 0; e < NEA; e++)
#ifdef APLAIN // default cache policy: the workgroup that shares the CU (or the XCD) and reads the same chunk of A a moment later finds it in L1 / L2
      if ((TK * TM / 2) % NT == 0 || t + NT * e < TK * TM / 2) ar[e] = *(const dbl2 *)(blk + 2 * (t + NT * e));
#else
      if ((TK * TM / 2) % NT == 0 || t + NT * e < TK * TM / 2) ar[e] = __builtin_nontemporal_load((const dbl2 *)(blk + 2 * (t + NT * e)));
#endif
#endif
  };
  auto loadG = [&](int kc, int *g) {
#ifndef NOGATHER
#pragma unroll
    for (int e = 0; e < NEB; e++) g[e] = gp[kc * TK + kb + KPB * e];
#endif
  };
#ifdef DEFSIGN
  unsigned sg = 0; // the signs of the gathered values: applied when they are stored to LDS, so that nothing waits for the gather before the products
  auto gatherB = [&](const int *g) {
    sg = 0;
#pragma unroll
    for (int e = 0; e < NEB; e++) {
      br[e] = X[(size_t)(g[e] >> 1) * 8 + sl];
      sg |= (unsigned)(g[e] & 1) << e;
    }
  };
#else
  auto gatherB = [&](const int *g) {
#ifndef NOGATHER // knock-out: no index loads, no gathers (timing only)
#pragma unroll
    for (int e = 0; e < NEB; e++) {
      const double v = X[(size_t)(g[e] >> 1) * 8 + sl];
      br[e]          = (g[e] & 1) ? -v : v;
    }
#endif
  };
#endif
  auto store = [&](int buf, dbl2 (&ar)[NEA]) {
#pragma unroll
    for (int e = 0; e < NEA; e++) {
      const int q = t + NT * e, k = q / (TM / 2), r2 = (q % (TM / 2)) * 2;
      if ((TK * TM / 2) % NT == 0 || q < TK * TM / 2) *(dbl2 *)&As[buf][k][r2] = ar[e];
    }
#pragma unroll
#ifdef DEFSIGN
    for (int e = 0; e < NEB; e++) Bs[buf][kb + KPB * e][col] = (sg >> e & 1) ? -br[e] : br[e];
#else
    for (int e = 0; e < NEB; e++) Bs[buf][kb + KPB * e][col] = br[e];
#endif
  };
  if (kc0 < kc1) {
    loadG(kc0, gn);
    loadA(kc0, ar);
    gatherB(gn);
    if (kc0 + 1 < kc1) loadG(kc0 + 1, gn);
    store(0, ar);
#ifdef APF2 // A travels TWO chunks ahead: chunk kc0 + 1 is requested here and stored at the end of the first step
    if (kc0 + 1 < kc1) loadA(kc0 + 1, ar);
#endif
  }
  __syncthreads();
  const int ka = lane >> 4, ra = lane & 15, cb = lane & 3;
  auto body = [&](int kc, int buf, dbl2 (&arst)[NEA], dbl2 (&arld)[NEA]) { // arst: the registers stored at the end of this step (chunk kc + 1), arld: where this step's loads of A go
#ifdef APF2
    if (kc + 2 < kc1) loadA(kc + 2, arld);
    if (kc + 1 < kc1) {
      gatherB(gn);
      if (kc + 2 < kc1) loadG(kc + 2, gn);
    }
#else
    if (kc + 1 < kc1) {
#ifdef GFIRST // the gathers (and the next index loads) go out BEFORE the HBM loads of A: vmcnt retires in order, so a wait for a gather no longer waits for A
      gatherB(gn);
      if (kc + 2 < kc1) loadG(kc + 2, gn);
      loadA(kc + 1, arld);
#else
      loadA(kc + 1, arld);
      gatherB(gn);
      if (kc + 2 < kc1) loadG(kc + 2, gn);
#endif
    }
#endif
#if defined(PIPE) && !defined(ORIENT4)
    // half steps: the operands of the next half step (a: 4 values every other half step, b: 8 values) are read from LDS before the 32 products of this one are issued
    {
      constexpr int NH = NB / 2;
      double a0[4], a1[4], b0[NH], b1[NH];
#pragma unroll
      for (int i = 0; i < 4; i++) a0[i] = As[buf][ka][wm * 64 + i * 16 + ra];
#pragma unroll
      for (int j = 0; j < NH; j++) b0[j] = Bs[buf][ka][wn * WTN + j * 4 + cb];
#pragma unroll
      for (int k4 = 0; k4 < TK / 4; k4++) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NH; j++) b1[j] = Bs[buf][4 * k4 + ka][wn * WTN + (NH + j) * 4 + cb];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < NH; j++) acc[i][j] = mfma4(a0[i], b0[j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
        if (k4 + 1 < TK / 4) {
#pragma unroll
          for (int i = 0; i < 4; i++) a1[i] = As[buf][4 * (k4 + 1) + ka][wm * 64 + i * 16 + ra];
#pragma unroll
          for (int j = 0; j < NH; j++) b0[j] = Bs[buf][4 * (k4 + 1) + ka][wn * WTN + j * 4 + cb];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < NH; j++) acc[i][NH + j] = mfma4(a0[i], b1[j], acc[i][NH + j]);
        __builtin_amdgcn_sched_barrier(0);
        if (k4 + 1 < TK / 4) {
#pragma unroll
          for (int i = 0; i < 4; i++) a0[i] = a1[i];
        }
      }
    }
#else
#pragma unroll
    for (int k4 = 0; k4 < TK / 4; k4++) {
      double a[NA], b[NB];
#if defined(NOLDSREAD) // knock-out (timing only): the operands of every k step are the same registers -- no LDS reads, no waits for them
#pragma unroll
      for (int i = 0; i < NA; i++) a[i] = acc[i][0] * 1e-300 + 1.0;
#pragma unroll
      for (int j = 0; j < NB; j++) b[j] = acc[0][j] * 1e-300 + 0.5;
#elif defined(ORIENT4)
#pragma unroll
      for (int i = 0; i < NA; i++) a[i] = As[buf][4 * k4 + ka][wm * WR + i * 4 + cb];
#pragma unroll
      for (int j = 0; j < NB; j++) b[j] = Bs[buf][4 * k4 + ka][wn * WTN + j * 16 + ra];
#else
#pragma unroll
      for (int i = 0; i < NA; i++) a[i] = As[buf][4 * k4 + ka][wm * 64 + i * 16 + ra];
#pragma unroll
      for (int j = 0; j < NB; j++) b[j] = Bs[buf][4 * k4 + ka][wn * WTN + j * 4 + cb];
#endif
#ifdef SETPRIO
      __builtin_amdgcn_s_setprio(SETPRIO);
#endif
#pragma unroll
      for (int i = 0; i < NA; i++)
#pragma unroll
        for (int j = 0; j < NB; j++) acc[i][j] = mfma4(a[i], b[j], acc[i][j]);
#ifdef SETPRIO
      __builtin_amdgcn_s_setprio(0);
#endif
#ifdef SGB
      // interleave: 1 LDS read per 3 MFMAs (20 reads, 64 MFMAs per k4-step)
#pragma unroll
      for (int g = 0; g < 20; g++) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); // DS read
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); // MFMA
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
#endif
    }
#endif
#ifndef NOSTORE
    if (kc + 1 < kc1) store(buf ^ 1, arst);
#endif
#ifndef NOBAR
    __syncthreads();
#endif
  };
#ifdef APF2
  dbl2 ar2[NEA];
  for (int kc = kc0; kc < kc1; kc += 2) {
    body(kc, 0, ar, ar2);
    if (kc + 1 < kc1) body(kc + 1, 1, ar2, ar);
  }
#else
  for (int kc = kc0; kc < kc1; kc++) body(kc, (kc - kc0) & 1, ar, ar);
#endif
  // D lane l: row 4 ((l >> 2) & 3) + (l >> 4) of the 16, column l & 3 of the 4
  const int rr = 4 * ((lane >> 2) & 3) + (lane >> 4);
  double   *C  = Cpart + (size_t)s * Mp * N;
#ifdef ORIENT4
  // same A in the 4 blocks, B block b = columns 4 b .. 4 b + 3: D lane l = row l >> 4 of the 4, column l & 15 of the 16
  (void)rr;
#pragma unroll
  for (int i = 0; i < NA; i++)
#pragma unroll
    for (int j = 0; j < NB; j++) C[(size_t)(mt * TM + wm * WR + i * 4 + ka) * N + nt * TN + wn * WTN + j * 16 + ra] = acc[i][j];
#else
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < WTN / 4; j++) C[(size_t)(mt * TM + wm * 64 + i * 16 + rr) * N + nt * TN + wn * WTN + j * 4 + cb] = acc[i][j];
#endif
}

int main(int argc, char **argv)
{
  const int M = argc > 1 ? atoi(argv[1]) : 715, nops = argc > 2 ? atoi(argv[2]) : 48, K = argc > 3 ? atoi(argv[3]) : 33288;
  const int Mp = (M + TM - 1) / TM * TM, N = nops * 8, ldk = (K + TK - 1) / TK * TK, nkc = ldk / TK, Mt = Mp / TM, Nt = N / TN;
  int       S = std::max(1, std::min(nkc, (256 + Mt * Nt - 1) / (Mt * Nt)));
  if (argc > 4) S = atoi(argv[4]);
  printf("tile %d x %d x %d (%d threads): M %d (padded %d) N %d K %d (padded %d): %d x %d tiles, split-K %d -> %d workgroups\n", TM, TN, TK, NT, M, Mp, N, K, ldk, Mt, Nt, S, Mt * Nt * S);
  if (N % TN) return printf("N must be a multiple of %d\n", TN), 1;
  std::vector<double> hA((size_t)Mp * ldk, 0.0), hX((size_t)(ldk + 1) * 8, 0.0);
  std::vector<int>    hg((size_t)nops * ldk);
  srand(1);
  auto rnd = []() { return (rand() % 2001 - 1000) / 1000.0; };
  std::vector<double> Arow((size_t)M * K);
  for (auto &v : Arow) v = rnd();
  for (int p = 0; p < M; p++)
    for (int c = 0; c < K; c++) hA[((size_t)(p / TM) * nkc + c / TK) * (TK * TM) + (size_t)(c % TK) * TM + p % TM] = Arow[(size_t)p * K + c];
  for (int i = 0; i < K * 8; i++) hX[i] = rnd();
  for (int g = 0; g < nops; g++)
    for (int c = 0; c < ldk; c++) hg[(size_t)g * ldk + c] = c < K ? (((rand() % K) << 1) | (rand() & 1)) : (ldk << 1); // pad -> the zero row ldk
  double *dA, *dX, *dC;
  int    *dg;
  (void)hipMalloc(&dA, hA.size() * 8), (void)hipMalloc(&dX, hX.size() * 8), (void)hipMalloc(&dg, hg.size() * 4), (void)hipMalloc(&dC, (size_t)S * Mp * N * 8);
  (void)hipMemcpy(dA, hA.data(), hA.size() * 8, hipMemcpyHostToDevice), (void)hipMemcpy(dX, hX.data(), hX.size() * 8, hipMemcpyHostToDevice), (void)hipMemcpy(dg, hg.data(), hg.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  int *dmap = nullptr;
  const int mapmode = getenv("OG_MAP") ? atoi(getenv("OG_MAP")) : 0;
  if (mapmode) {
    // blocks b and b + 256 share a CU (scripts/micro/census.hip), blocks b and b + 8 an XCD.  mode 1: the column tiles 0 and 1 of one (row tile, split) on ONE CU (blocks b, b + 256),
    // the third column tile wherever there is room; mode 2: the column tiles of one (row tile, split) 8 blocks apart (one XCD)
    const int nwg = Mt * Nt * S;
    std::vector<int> map(nwg, -1), rest;
    if (mapmode == 1) {
      int p = 0;
      for (int mt = 0; mt < Mt; mt++)
        for (int sp = 0; sp < S; sp++) {
          if (p < 256 && p + 256 < nwg && Nt >= 2) map[p] = (mt * Nt + 0) * S + sp, map[p + 256] = (mt * Nt + 1) * S + sp, p++;
          else rest.push_back((mt * Nt + 0) * S + sp), rest.push_back((mt * Nt + 1) * S + sp);
          for (int nt = 2; nt < Nt; nt++) rest.push_back((mt * Nt + nt) * S + sp);
        }
      size_t r = 0;
      for (int b = 0; b < nwg; b++) if (map[b] < 0) map[b] = rest[r++];
    } else {
      int b = 0;
      for (int mt = 0; mt < Mt; mt++)
        for (int s0 = 0; s0 < S; s0 += 8) {
          const int w = std::min(8, S - s0);
          for (int nt = 0; nt < Nt; nt++)
            for (int j = 0; j < w; j++) map[b++] = (mt * Nt + nt) * S + s0 + j;
        }
    }
    (void)hipMalloc(&dmap, nwg * 4), (void)hipMemcpy(dmap, map.data(), nwg * 4, hipMemcpyHostToDevice);
    printf("work-item map mode %d\n", mapmode);
  }
  for (int it = 0; it < 3; it++) k_gemm<<<Mt * Nt * S, NT>>>(Mt, Nt, S, nkc, dA, dg, ldk, dX, dC, Mp, N, dmap);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  const int reps = 20;
  for (int it = 0; it < reps; it++) k_gemm<<<Mt * Nt * S, NT>>>(Mt, Nt, S, nkc, dA, dg, ldk, dX, dC, Mp, N, dmap);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  printf("%.4f ms per GEMM: %.1f TFLOP/s useful (M N K), %.1f TFLOP/s issued (padded M)\n", ms, 2.0 * M * N * (double)K / ms * 1e-9, 2.0 * Mp * N * (double)ldk / ms * 1e-9);
  std::vector<double> hC((size_t)S * Mp * N);
  (void)hipMemcpy(hC.data(), dC, hC.size() * 8, hipMemcpyDeviceToHost);
  double err = 0.0, ref_max = 0.0;
  for (int tcase = 0; tcase < 200; tcase++) {
    const int p = rand() % M, g = rand() % nops, sl = rand() % 8;
    double    ref = 0.0, got = 0.0;
    for (int c = 0; c < K; c++) {
      const int gi = hg[(size_t)g * ldk + c];
      ref += Arow[(size_t)p * K + c] * ((gi & 1) ? -1.0 : 1.0) * hX[(size_t)(gi >> 1) * 8 + sl];
    }
    for (int s = 0; s < S; s++) got += hC[((size_t)s * Mp + p) * N + g * 8 + sl];
    err = std::max(err, std::fabs(got - ref)), ref_max = std::max(ref_max, std::fabs(ref));
  }
  printf("max |C - ref| over 200 samples: %.3e (max |ref| %.3e) %s\n", err, ref_max, err <= 1e-10 * std::max(1.0, ref_max) ? "OK" : "WRONG");
  return 0;
}
