// Micro-benchmark: issue rate and lane maps of the two fp64 MFMA shapes of gfx950 (v_mfma_f64_16x16x4_f64, v_mfma_f64_4x4x4_4b_f64).
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f64 scripts/micro/mfma_f64.hip && /tmp/mfma_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_rate16(double *out, int n)
{
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  for (int i = 0; i < n; i++) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  d4 s = c0 + c1 + c2 + c3;
  out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
}
__global__ __launch_bounds__(256) void k_rate4(double *out, int n)
{
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
  for (int i = 0; i < n; i++) {
    c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
    c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0);
    c5 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c5, 0, 0, 0);
    c6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c6, 0, 0, 0);
    c7 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c7, 0, 0, 0);
  }
  out[blockIdx.x * 256 + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}
// fp64 FMA on the vector ALU for comparison (8 independent chains)
__global__ __launch_bounds__(256) void k_rate_valu(double *out, int n)
{
  double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9 * threadIdx.x;
  double c[8] = {0, 1, 2, 3, 4, 5, 6, 7};
  for (int i = 0; i < n; i++)
#pragma unroll
    for (int j = 0; j < 8; j++) c[j] = __builtin_fma(c[j], a, b);
  out[blockIdx.x * 256 + threadIdx.x] = c[0] + c[1] + c[2] + c[3] + c[4] + c[5] + c[6] + c[7];
}
// lane maps: A one-hot on lane la, B one-hot on lane lb => which (lane, register) of D is 1
__global__ void k_map16(int *where)
{
  int lane = threadIdx.x;
  for (int la = 0; la < 64; la++)
    for (int lb = 0; lb < 64; lb++) {
      d4 c = {0, 0, 0, 0};
      c = __builtin_amdgcn_mfma_f64_16x16x4f64(lane == la ? 1.0 : 0.0, lane == lb ? 1.0 : 0.0, c, 0, 0, 0);
      for (int r = 0; r < 4; r++)
        if (c[r] != 0.0) where[la * 64 + lb] = lane * 4 + r;
    }
}
__global__ void k_map4(int *where)
{
  int lane = threadIdx.x;
  for (int la = 0; la < 64; la++)
    for (int lb = 0; lb < 64; lb++) {
      double c = __builtin_amdgcn_mfma_f64_4x4x4f64(lane == la ? 1.0 : 0.0, lane == lb ? 1.0 : 0.0, 0.0, 0, 0, 0);
      if (c != 0.0) where[la * 64 + lb] = lane;
    }
}
int main()
{
  double *out;
  hipMalloc(&out, 1024 * 256 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  int n = 20000, nwg = 1024; // 4 workgroups of 4 waves per CU: one or more waves per SIMD
  for (int wpc = 1; wpc <= 2; wpc++) {
    nwg = 256 * wpc;
    float ms;
    k_rate16<<<nwg, 256>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_rate16<<<nwg, 256>>>(out, n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)nwg * 4 * n * 4 * 2048.0;
    printf("16x16x4 f64: %d workgroups x 4 waves: %.3f ms  %.1f TFLOP/s  (%.1f clk per MFMA per SIMD at 2.4 GHz, %d waves/SIMD)\n", nwg, ms, fl / ms * 1e-9, ms * 1e-3 * 2.4e9 / (n * 4.0 * wpc), wpc);
    k_rate4<<<nwg, 256>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_rate4<<<nwg, 256>>>(out, n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    fl = (double)nwg * 4 * n * 8 * 512.0;
    printf("4x4x4_4b f64: %d workgroups x 4 waves: %.3f ms  %.1f TFLOP/s  (%.1f clk per MFMA per SIMD at 2.4 GHz)\n", nwg, ms, fl / ms * 1e-9, ms * 1e-3 * 2.4e9 / (n * 8.0 * wpc));
    k_rate_valu<<<nwg, 256>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_rate_valu<<<nwg, 256>>>(out, n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    fl = (double)nwg * 256 * n * 8 * 2.0;
    printf("v_fma_f64: %d workgroups x 4 waves: %.3f ms  %.1f TFLOP/s\n", nwg, ms, fl / ms * 1e-9);
  }
  int *w;
  hipMalloc(&w, 4096 * 4);
  std::vector<int> h(4096);
  hipMemset(w, 0xff, 4096 * 4);
  k_map16<<<1, 64>>>(w);
  hipMemcpy(h.data(), w, 4096 * 4, hipMemcpyDeviceToHost);
  // A lane la = (m, k), B lane lb = (k', n): non-zero iff k == k'
  printf("16x16x4 map: A lane la, B lane lb -> D (lane, reg); shown for la in {0,1,16,17}, lb in {0,1,16,17}\n");
  for (int la : {0, 1, 16, 17, 33})
    for (int lb : {0, 1, 16, 17, 33}) printf("  la %2d lb %2d -> %s lane %d reg %d\n", la, lb, h[la * 64 + lb] < 0 ? "none" : "D", h[la * 64 + lb] / 4, h[la * 64 + lb] % 4);
  int ok = 1;
  for (int la = 0; la < 64; la++)
    for (int lb = 0; lb < 64; lb++) {
      int m = la & 15, k = la >> 4, k2 = lb >> 4, nn = lb & 15;
      int exp = (k == k2) ? ((nn + 16 * (m & 3)) * 4 + (m >> 2)) : -1; // col = lane&15 = n, row = (lane>>4) + 4 reg = m
      if (h[la * 64 + lb] != exp) ok = 0;
    }
  printf("16x16x4: A[m=l&15][k=l>>4], B[k=l>>4][n=l&15], D col=lane&15,row=(lane>>4)+4*reg : %s\n", ok ? "CONFIRMED" : "WRONG");
  hipMemset(w, 0xff, 4096 * 4);
  k_map4<<<1, 64>>>(w);
  hipMemcpy(h.data(), w, 4096 * 4, hipMemcpyDeviceToHost);
  printf("4x4x4_4b map (la, lb -> D lane), non-zero pairs for la = 0, 1, 4, 5, 16:\n");
  for (int la : {0, 1, 4, 5, 16}) {
    printf("  la %2d:", la);
    for (int lb = 0; lb < 64; lb++)
      if (h[la * 64 + lb] >= 0) printf(" (lb %d -> lane %d)", lb, h[la * 64 + lb]);
    printf("\n");
  }
  return 0;
}
