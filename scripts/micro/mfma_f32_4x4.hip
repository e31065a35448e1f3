// Round 6 probe: lane maps of v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4 x 4 outer products, K = 1): which (lane, register) of D receives A[lane la] * B[lane lb].
// hipcc --offload-arch=gfx950 -O2 scripts/micro/mfma_f32_4x4.hip -o scripts/micro/bin/mfma_f32_4x4 && scripts/micro/bin/mfma_f32_4x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(int *out)
{
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; la++)
    for (int lb = 0; lb < 64; lb++) {
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
      c       = __builtin_amdgcn_mfma_f32_4x4x1f32(lane == la ? 1.f : 0.f, lane == lb ? 1.f : 0.f, c, 0, 0, 0);
      for (int v = 0; v < 4; v++)
        if (c[v] != 0.f) out[la * 64 + lb] = lane * 4 + v; // (at most one (lane, register) per pair)
    }
}
int main()
{
  int *d;
  hipMalloc(&d, sizeof(int) * 4096);
  hipMemset(d, 0xff, sizeof(int) * 4096);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  std::vector<int> h(4096);
  hipMemcpy(h.data(), d, sizeof(int) * 4096, hipMemcpyDeviceToHost);
  int hits = 0;
  for (int la = 0; la < 64; la++)
    for (int lb = 0; lb < 64; lb++)
      if (h[la * 64 + lb] >= 0) hits++;
  printf("pairs with a product: %d of 4096\n", hits);
  for (int la = 0; la < 12; la++) {
    printf("A lane %2d:", la);
    for (int lb = 0; lb < 64; lb++)
      if (h[la * 64 + lb] >= 0) printf("  B%-2d->D(lane %d, reg %d)", lb, h[la * 64 + lb] / 4, h[la * 64 + lb] % 4);
    printf("\n");
  }
  // hypothesis: block = lane / 4, A row i = la % 4, B column j = lb % 4, D[lane = 4 blk + j][reg = i]
  int bad = 0;
  for (int la = 0; la < 64; la++)
    for (int lb = 0; lb < 64; lb++) {
      const int want = (la / 4 == lb / 4) ? ((4 * (la / 4) + lb % 4) * 4 + la % 4) : -1;
      if (h[la * 64 + lb] != want) bad++;
    }
  printf("hypothesis D[lane 4 blk + j][reg i] = A[4 blk + i] B[4 blk + j]: %d mismatches\n", bad);
  return 0;
}
