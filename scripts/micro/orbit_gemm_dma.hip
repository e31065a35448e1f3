// Orbit GEMM with both operands staged by LDS-DMA (global_load_lds_dwordx4): no VGPR round trip, no ds_write, no sign arithmetic.
//   A: the pre-tiled chunk [16 k][120 rows] is 15 360 contiguous bytes = 15 wave-instructions of 1 KB, written to LDS as it lies (row stride 240 dwords = 48 banks: conflict-free reads);
//   B: one wave-instruction per k-row of the 16 x 128 tile: lane l fetches the 16 bytes (slots 2 (l & 3), 2 (l & 3) + 1) of row gidx[op l >> 2][k] of X+- -- the multivector kept
//      TWICE, row 2 p = X[p], row 2 p + 1 = -X[p], so that the sign of the operation is part of the gather index and not a VALU instruction.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/og_dma scripts/micro/orbit_gemm_dma.hip && /tmp/og_dma [M N_ops K S]     (OG_MAP as orbit_gemm.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#ifndef NA
#define NA 15
#endif
#define WR (4 * NA)
#define TM (2 * WR)
#define TN 128
#define TK 16
#define LDB (TN + 16)
#define APIECES (TK * TM * 8 / 1024)
static_assert(TK * TM * 8 % 1024 == 0, "A chunk = whole 1 KB pieces");
static __device__ __forceinline__ double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
typedef __attribute__((address_space(3))) void       *lds_ptr;
typedef const __attribute__((address_space(1))) void *glb_ptr;
static __device__ __forceinline__ void glds16(const void *src, void *lds_base) { __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)lds_base, 16, 0, 0); }

__global__ __launch_bounds__(256, 1) void k_gemm_dma(int Mt, int Nt, int S, int nkc, const double *__restrict__ A, const int *__restrict__ gidx, int ldk, const double *__restrict__ Xpm, double *__restrict__ Cpart,
                                                    int Mp, int N, const int *__restrict__ wmap)
{
  __shared__ double smem[2 * TK * TM + 2 * TK * LDB];
  double *const     As = smem, *const Bs = smem + 2 * TK * TM;
  const int wgi = wmap ? __builtin_amdgcn_readfirstlane(wmap[blockIdx.x]) : blockIdx.x, s = wgi % S, nt = (wgi / S) % Nt, mt = wgi / (S * Nt);
  const int kc0 = (int)((long long)nkc * s / S), kc1 = (int)((long long)nkc * (s + 1) / S);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
  const int *gp = gidx + (size_t)(nt * (TN / 8) + (lane >> 2)) * ldk; // this lane's operation: 4 lanes per operation (slot pairs 0..3)
  double acc[NA][4];
#pragma unroll
  for (int i = 0; i < NA; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = 0.0;
  int idx[4];
  auto loadIdx = [&](int kc) {
#pragma unroll
    for (int e = 0; e < 4; e++) idx[e] = gp[kc * TK + wave * 4 + e];
  };
  auto issue = [&](int kc, int buf) {
    const char *ablk = (const char *)(A + ((size_t)mt * nkc + kc) * (TK * TM));
    char       *adst = (char *)(As + buf * TK * TM);
#pragma unroll
    for (int e = 0; e < (APIECES + 3) / 4; e++) {
      const int p = wave + 4 * e; // wave-uniform
      if (p < APIECES) glds16(ablk + p * 1024 + lane * 16, adst + p * 1024);
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int krow = wave * 4 + e;
      glds16((const char *)Xpm + (size_t)idx[e] * 64 + (lane & 3) * 16, (char *)(Bs + (buf * TK + krow) * LDB));
    }
  };
  if (kc0 < kc1) {
    loadIdx(kc0);
    issue(kc0, 0);
    if (kc0 + 1 < kc1) loadIdx(kc0 + 1);
  }
  __syncthreads();
  const int ka = lane >> 4, ra = lane & 15, cb = lane & 3;
  for (int kc = kc0; kc < kc1; kc++) {
    const int buf = (kc - kc0) & 1;
    if (kc + 1 < kc1) {
      issue(kc + 1, buf ^ 1);
      if (kc + 2 < kc1) loadIdx(kc + 2);
    }
    const double *Ab = As + buf * TK * TM, *Bb = Bs + buf * TK * LDB;
#pragma unroll
    for (int k4 = 0; k4 < TK / 4; k4++) {
      double a[NA], b[4];
#pragma unroll
      for (int i = 0; i < NA; i++) a[i] = Ab[(4 * k4 + ka) * TM + wm * WR + i * 4 + cb];
#pragma unroll
      for (int j = 0; j < 4; j++) b[j] = Bb[(4 * k4 + ka) * LDB + wn * 64 + j * 16 + ra];
#pragma unroll
      for (int i = 0; i < NA; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = mfma4(a[i], b[j], acc[i][j]);
    }
    __syncthreads(); // emits vmcnt(0): the DMA pieces of the next chunk have landed before anybody reads them
  }
  double *C = Cpart + (size_t)s * Mp * N;
#pragma unroll
  for (int i = 0; i < NA; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) C[(size_t)(mt * TM + wm * WR + i * 4 + ka) * N + nt * TN + wn * 64 + j * 16 + ra] = acc[i][j];
}

// Hybrid: A by LDS-DMA as above; B by per-COLUMN 8-byte gathers from X+- through registers (arbitrary column lists, as the library's pruned lists are), no sign arithmetic
__global__ __launch_bounds__(256, 1) void k_gemm_hyb(int Mt, int Nt, int S, int nkc, const double *__restrict__ A, const int *__restrict__ gidx, int ldk, const double *__restrict__ Xpm, double *__restrict__ Cpart,
                                                    int Mp, int N, const int *__restrict__ wmap)
{
  __shared__ double smem[2 * TK * TM + 2 * TK * LDB];
  double *const     As = smem, *const Bs = smem + 2 * TK * TM;
  const int wgi = wmap ? __builtin_amdgcn_readfirstlane(wmap[blockIdx.x]) : blockIdx.x, s = wgi % S, nt = (wgi / S) % Nt, mt = wgi / (S * Nt);
  const int kc0 = (int)((long long)nkc * s / S), kc1 = (int)((long long)nkc * (s + 1) / S);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
  const int col = t % TN, kb = t / TN; // this thread's column of the tile, k rows kb, kb + 2, ...
  const int *gp = gidx + (size_t)(nt * (TN / 8) + (col >> 3)) * ldk;
  const double *xs = Xpm + (col & 7);
  double acc[NA][4];
#pragma unroll
  for (int i = 0; i < NA; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = 0.0;
  int    gn[8];
  double br[8];
  auto loadG = [&](int kc) {
#pragma unroll
    for (int e = 0; e < 8; e++) gn[e] = gp[kc * TK + kb + 2 * e];
  };
  auto gatherB = [&]() {
#pragma unroll
    for (int e = 0; e < 8; e++) br[e] = xs[(size_t)gn[e] * 8];
  };
  auto issueA = [&](int kc, int buf) {
    const char *ablk = (const char *)(A + ((size_t)mt * nkc + kc) * (TK * TM));
    char       *adst = (char *)(As + buf * TK * TM);
#pragma unroll
    for (int e = 0; e < (APIECES + 3) / 4; e++) {
      const int p = wave + 4 * e;
      if (p < APIECES) glds16(ablk + p * 1024 + lane * 16, adst + p * 1024);
    }
  };
  auto storeB = [&](int buf) {
#pragma unroll
    for (int e = 0; e < 8; e++) Bs[(buf * TK + kb + 2 * e) * LDB + col] = br[e];
  };
  if (kc0 < kc1) {
    loadG(kc0);
    issueA(kc0, 0);
    gatherB();
    if (kc0 + 1 < kc1) loadG(kc0 + 1);
    storeB(0);
  }
  __syncthreads();
  const int ka = lane >> 4, ra = lane & 15, cb = lane & 3;
  for (int kc = kc0; kc < kc1; kc++) {
    const int buf = (kc - kc0) & 1;
    if (kc + 1 < kc1) {
      gatherB();
      if (kc + 2 < kc1) loadG(kc + 2);
      issueA(kc + 1, buf ^ 1);
    }
    const double *Ab = As + buf * TK * TM, *Bb = Bs + buf * TK * LDB;
#pragma unroll
    for (int k4 = 0; k4 < TK / 4; k4++) {
      double a[NA], b[4];
#pragma unroll
      for (int i = 0; i < NA; i++) a[i] = Ab[(4 * k4 + ka) * TM + wm * WR + i * 4 + cb];
#pragma unroll
      for (int j = 0; j < 4; j++) b[j] = Bb[(4 * k4 + ka) * LDB + wn * 64 + j * 16 + ra];
#pragma unroll
      for (int i = 0; i < NA; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = mfma4(a[i], b[j], acc[i][j]);
    }
    if (kc + 1 < kc1) storeB(buf ^ 1);
    __syncthreads();
  }
  double *C = Cpart + (size_t)s * Mp * N;
#pragma unroll
  for (int i = 0; i < NA; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) C[(size_t)(mt * TM + wm * WR + i * 4 + ka) * N + nt * TN + wn * 64 + j * 16 + ra] = acc[i][j];
}

int main(int argc, char **argv)
{
  const int M = argc > 1 ? atoi(argv[1]) : 715, nops = argc > 2 ? atoi(argv[2]) : 48, K = argc > 3 ? atoi(argv[3]) : 33288;
  const int Mp = (M + TM - 1) / TM * TM, N = nops * 8, ldk = (K + TK - 1) / TK * TK, nkc = ldk / TK, Mt = Mp / TM, Nt = N / TN;
  int       S = argc > 4 ? atoi(argv[4]) : 28;
  printf("DMA tile %d x %d x %d: M %d (padded %d) N %d K %d: %d x %d tiles, split-K %d -> %d workgroups\n", TM, TN, TK, M, Mp, N, K, Mt, Nt, S, Mt * Nt * S);
  if (N % TN) return printf("N must be a multiple of %d\n", TN), 1;
  std::vector<double> hA((size_t)Mp * ldk, 0.0), hX((size_t)(ldk + 1) * 8, 0.0), hXpm((size_t)(ldk + 1) * 16, 0.0);
  std::vector<int>    hg((size_t)nops * ldk);
  srand(1);
  auto rnd = []() { return (rand() % 2001 - 1000) / 1000.0; };
  std::vector<double> Arow((size_t)M * K);
  for (auto &v : Arow) v = rnd();
  for (int p = 0; p < M; p++)
    for (int c = 0; c < K; c++) hA[((size_t)(p / TM) * nkc + c / TK) * (TK * TM) + (size_t)(c % TK) * TM + p % TM] = Arow[(size_t)p * K + c];
  for (int i = 0; i < K * 8; i++) hX[i] = rnd();
  for (int p = 0; p <= ldk; p++)
    for (int sl = 0; sl < 8; sl++) hXpm[(size_t)(2 * p) * 8 + sl] = hX[(size_t)p * 8 + sl], hXpm[(size_t)(2 * p + 1) * 8 + sl] = -hX[(size_t)p * 8 + sl];
  for (int g = 0; g < nops; g++)
    for (int c = 0; c < ldk; c++) hg[(size_t)g * ldk + c] = c < K ? (((rand() % K) << 1) | (rand() & 1)) : (ldk << 1);
  double *dA, *dX, *dC;
  int    *dg;
  (void)hipMalloc(&dA, hA.size() * 8), (void)hipMalloc(&dX, hXpm.size() * 8), (void)hipMalloc(&dg, hg.size() * 4), (void)hipMalloc(&dC, (size_t)S * Mp * N * 8);
  (void)hipMemcpy(dA, hA.data(), hA.size() * 8, hipMemcpyHostToDevice), (void)hipMemcpy(dX, hXpm.data(), hXpm.size() * 8, hipMemcpyHostToDevice), (void)hipMemcpy(dg, hg.data(), hg.size() * 4, hipMemcpyHostToDevice);
  int *dmap = nullptr;
  if (getenv("OG_MAP") && atoi(getenv("OG_MAP")) == 2) {
    const int        nwg = Mt * Nt * S;
    std::vector<int> map(nwg, -1);
    int              b = 0;
    for (int mt = 0; mt < Mt; mt++)
      for (int s0 = 0; s0 < S; s0 += 8)
        for (int nt = 0; nt < Nt; nt++)
          for (int j = 0; j < std::min(8, S - s0); j++) map[b++] = (mt * Nt + nt) * S + s0 + j;
    (void)hipMalloc(&dmap, nwg * 4), (void)hipMemcpy(dmap, map.data(), nwg * 4, hipMemcpyHostToDevice);
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  const bool hyb = getenv("OG_HYB") != nullptr;
  printf("%s\n", hyb ? "hybrid: A by LDS-DMA, B by 8-byte gathers from X+-" : "both operands by LDS-DMA");
  for (int it = 0; it < 3; it++) {
    if (hyb) k_gemm_hyb<<<Mt * Nt * S, 256>>>(Mt, Nt, S, nkc, dA, dg, ldk, dX, dC, Mp, N, dmap);
    else k_gemm_dma<<<Mt * Nt * S, 256>>>(Mt, Nt, S, nkc, dA, dg, ldk, dX, dC, Mp, N, dmap);
  }
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  const int reps = 20;
  for (int it = 0; it < reps; it++) {
    if (hyb) k_gemm_hyb<<<Mt * Nt * S, 256>>>(Mt, Nt, S, nkc, dA, dg, ldk, dX, dC, Mp, N, dmap);
    else k_gemm_dma<<<Mt * Nt * S, 256>>>(Mt, Nt, S, nkc, dA, dg, ldk, dX, dC, Mp, N, dmap);
  }
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
