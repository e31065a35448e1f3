// What does a "last workgroup done" ticket cost on MI355X?  400 workgroups of 256 threads write one value each and one partial per workgroup; variants of the tail:
//  0 nothing   1 __threadfence only   2 relaxed atomic only   3 fence + acq_rel atomic (the textbook ticket)   4 agent-scope stores, s_waitcnt, relaxed atomic
//  5 as 4 with a two-level ticket (16 counters, then one)   6 as 4, last workgroup also sums the partials with agent-scope loads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int V>
__global__ __launch_bounds__(256) void k(int n, const double *x, double *y, double *part, unsigned *ticket, double *out)
{
  __shared__ int last;
  const int i = blockIdx.x * 256 + threadIdx.x;
  double v = i < n ? x[i] * 1.0001 : 0.0;
  if (i < n) y[i] = v;
  // block partial
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  __shared__ double w[4];
  if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double s = (w[0] + w[1]) + (w[2] + w[3]);
    if (V >= 4) __hip_atomic_store(&part[blockIdx.x], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else part[blockIdx.x] = s;
  }
  if (V == 0) return;
  if (V == 1 || V == 3) __threadfence();
  if (V >= 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (V == 1) return;
  if (threadIdx.x == 0) {
    if (V == 5) {
      const unsigned g = blockIdx.x & 15, ng = (gridDim.x + 15 - g) / 16;
      unsigned a = __hip_atomic_fetch_add(ticket + 1 + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = 0;
      if (a == ng - 1) {
        __hip_atomic_store(ticket + 1 + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned b = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (b == 15) last = 1, __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      unsigned a = (V == 3) ? __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) : __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = (a == gridDim.x - 1);
      if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  if (!last) return;
  if (V == 3) __threadfence();
  if (V == 3 || V == 6) {
    double s = 0.0;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += 256) s += (V == 6) ? __hip_atomic_load(&part[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : part[b];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *out = (w[0] + w[1]) + (w[2] + w[3]);
  } else if (threadIdx.x == 0) *out = 1.0;
}

template <int V> float run(int n, int reps, const double *x, double *y, double *part, unsigned *ticket, double *out, hipStream_t st)
{
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const int nb = (n + 255) / 256;
  for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k<V>, dim3(nb), dim3(256), 0, st, n, x, y, part, ticket, out);
  CHK(hipEventRecord(e0, st));
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k<V>, dim3(nb), dim3(256), 0, st, n, x, y, part, ticket, out);
  CHK(hipEventRecord(e1, st));
  CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / reps;
}

int main(int argc, char **argv)
{
  const int n = argc > 1 ? atoi(argv[1]) : 102268, reps = 200;
  double *x, *y, *part, *out; unsigned *ticket;
  CHK(hipMalloc(&x, 8 * (size_t)n)); CHK(hipMalloc(&y, 8 * (size_t)n)); CHK(hipMalloc(&part, 8 * 4096)); CHK(hipMalloc(&out, 64)); CHK(hipMalloc(&ticket, 256));
  CHK(hipMemset(x, 0, 8 * (size_t)n)); CHK(hipMemset(ticket, 0, 256));
  hipStream_t st; CHK(hipStreamCreate(&st));
  printf("n = %d (%d workgroups), us per launch back to back:\n", n, (n + 255) / 256);
  printf(" 0 no tail                         %7.2f\n", run<0>(n, reps, x, y, part, ticket, out, st));
  printf(" 1 __threadfence                   %7.2f\n", run<1>(n, reps, x, y, part, ticket, out, st));
  printf(" 2 relaxed atomic                  %7.2f\n", run<2>(n, reps, x, y, part, ticket, out, st));
  printf(" 3 fence + acq_rel atomic + sum    %7.2f\n", run<3>(n, reps, x, y, part, ticket, out, st));
  printf(" 4 sc1 stores + waitcnt + relaxed  %7.2f\n", run<4>(n, reps, x, y, part, ticket, out, st));
  printf(" 5 as 4, two-level ticket          %7.2f\n", run<5>(n, reps, x, y, part, ticket, out, st));
  printf(" 6 as 4 + sum with sc1 loads       %7.2f\n", run<6>(n, reps, x, y, part, ticket, out, st));
  double h; CHK(hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost));
  printf("out %g\n", h);
  return 0;
}
