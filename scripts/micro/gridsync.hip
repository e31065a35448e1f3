// micro-benchmark: cost of a grid-wide barrier on MI355X for small cooperative grids (is a fused multi-phase coarse-level kernel
// cheaper than one launch per phase?).  hipcc --offload-arch=gfx950 -O3 gridsync.hip -o gridsync
#include <hip/hip_cooperative_groups.h>
#include <hip/hip_runtime.h>

#include <cstdio>
namespace cg = cooperative_groups;

__global__ __launch_bounds__(256) void k_sync(int nsync, float *x, int n)
{
  cg::grid_group g = cg::this_grid();
  for (int s = 0; s < nsync; s++) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) x[i] = x[(i + 977) % n] * 0.5f + 1.0f; // a little cross-workgroup traffic per phase
    g.sync();
  }
}

// hand-made barrier: one atomic counter per phase (monotone target), agent-scope fences, bounded spin
__global__ __launch_bounds__(256) void k_sync_atomic(int nsync, float *x, int n, unsigned *ctr, int *timeout)
{
  for (int s = 0; s < nsync; s++) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) x[i] = x[(i + 977) % n] * 0.5f + 1.0f;
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      atomicAdd(ctr, 1u);
      const unsigned target = (unsigned)(s + 1) * gridDim.x;
      long long      spins  = 0;
      while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++spins > 20000000LL) {
          *timeout = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      __threadfence();
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_phase(float *x, int n)
{
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) x[i] = x[(i + 977) % n] * 0.5f + 1.0f;
}

int main()
{
  const int n = 36501;
  float    *x;
  unsigned *ctr;
  int      *to;
  hipMalloc(&x, n * sizeof(float));
  hipMalloc(&ctr, 4);
  hipMalloc(&to, 4);
  hipMemset(x, 0, n * sizeof(float));
  hipMemset(to, 0, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int wgs : {32, 64, 128, 256}) {
    for (int nsync : {1, 101}) {
      int   ns = nsync, nn = n;
      void *args[] = {&ns, &x, &nn};
      hipLaunchCooperativeKernel((void *)k_sync, dim3(wgs), dim3(256), args, 0, 0);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int r = 0; r < 10; r++) hipLaunchCooperativeKernel((void *)k_sync, dim3(wgs), dim3(256), args, 0, 0);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("grid.sync   wgs %3d nsync %3d : %8.2f us per kernel\n", wgs, nsync, ms * 100.0f);
      hipMemset(ctr, 0, 4);
      hipLaunchKernelGGL(k_sync_atomic, dim3(wgs), dim3(256), 0, 0, ns, x, nn, ctr, to);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int r = 0; r < 10; r++) {
        hipMemsetAsync(ctr, 0, 4);
        hipLaunchKernelGGL(k_sync_atomic, dim3(wgs), dim3(256), 0, 0, ns, x, nn, ctr, to);
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      int hto = 0;
      hipMemcpy(&hto, to, 4, hipMemcpyDeviceToHost);
      printf("atomic      wgs %3d nsync %3d : %8.2f us per kernel (incl. memset)  timeout=%d\n", wgs, nsync, ms * 100.0f, hto);
    }
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 1000; r++) hipLaunchKernelGGL(k_phase, dim3(wgs), dim3(256), 0, 0, x, n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("launches    wgs %3d            : %8.2f us per launch (back to back in one stream)\n", wgs, ms);
  }
  return 0;
}
