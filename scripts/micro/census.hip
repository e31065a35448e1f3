// Which workgroups of a one-round grid share a CU?  507 workgroups of 256 threads with 72 KB of LDS (2 per CU, as k_fxo_gemm4), each records
// HW_REG_HW_ID / HW_REG_XCC_ID and its start time, then spins ~40 us so that the whole grid is resident together.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/census scripts/micro/census.hip && /tmp/census [nwg]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void k_census(unsigned *out, long long *t0)
{
  __shared__ double pad[9216]; // 72 KB
  pad[threadIdx.x] = 0.0;
  if (threadIdx.x == 0) {
    unsigned hw  = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));  // HW_REG_HW_ID, 32 bits
    unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)); // HW_REG_XCC_ID
    out[2 * blockIdx.x]     = hw;
    out[2 * blockIdx.x + 1] = xcc;
    t0[blockIdx.x]          = __builtin_amdgcn_s_memrealtime();
  }
  long long t = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t < 4000) {} // 100 MHz ticks: 40 us
  if (pad[threadIdx.x] != 0.0) out[0] = 0;
}
int main(int argc, char **argv)
{
  int nwg = argc > 1 ? atoi(argv[1]) : 507;
  unsigned *d; long long *dt;
  hipMalloc(&d, nwg * 8); hipMalloc(&dt, nwg * 8);
  for (int rep = 0; rep < 2; rep++) {
    k_census<<<nwg, 256>>>(d, dt);
    hipDeviceSynchronize();
  }
  std::vector<unsigned> h(2 * nwg); std::vector<long long> t(nwg);
  hipMemcpy(h.data(), d, nwg * 8, hipMemcpyDeviceToHost); hipMemcpy(t.data(), dt, nwg * 8, hipMemcpyDeviceToHost);
  std::map<unsigned long long, std::vector<int>> cu;
  long long tmin = t[0];
  for (int b = 0; b < nwg; b++) tmin = std::min(tmin, t[b]);
  for (int b = 0; b < nwg; b++) {
    unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
    unsigned cu_id = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    cu[((unsigned long long)xcc << 32) | (se << 8) | (sh << 4) | cu_id].push_back(b);
  }
  printf("%d workgroups on %zu distinct (xcc, se, sh, cu)\n", nwg, cu.size());
  std::map<int, int> diffs;
  int shown = 0;
  for (auto &kv : cu) {
    if (shown < 24) {
      printf("xcc %llu se/sh/cu %03llx:", kv.first >> 32, kv.first & 0xfff);
      for (int b : kv.second) printf(" %d(+%lld)", b, t[b] - tmin);
      printf("\n"), shown++;
    }
    if (kv.second.size() == 2) diffs[kv.second[1] - kv.second[0]]++;
  }
  printf("difference of the two block indices sharing a CU: ");
  for (auto &d2 : diffs) printf("%d x%d  ", d2.first, d2.second);
  printf("\nraw hw_id of blocks 0..3: %08x %08x %08x %08x\n", h[0], h[2], h[4], h[6]);
  return 0;
}
