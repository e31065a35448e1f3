// Round 5: what the host's one synchronisation per MPGP step costs on this chip, and what the alternatives would cost.  kernel1 (a stand-in for k_dc_final: 400 workgroups,
// ~10 us) -> the host learns a value -> kernel2 (stand-in for the step's vector kernel) with an argument that depends on it.  Measured: the GPU-side gap between the end of
// kernel1 and the start of kernel2 (wall_clock64 stamps written by the kernels themselves, 100 MHz).
//   A  hipStreamSynchronize, then launch kernel2                                  (what solve_fused does)
//   B  kernel1's LAST workgroup (ticket) writes a flag to pinned host memory; the host spins on it, then launches kernel2
//   C  as B, but a one-thread gate kernel and kernel2 are ALREADY enqueued: the gate spins on a pinned host word the host sets after it has computed the argument,
//      kernel2 reads the argument from pinned memory
// build: hipcc --offload-arch=gfx950 -O3 -o roundtrip roundtrip.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k1(double *v, int n, unsigned long long *t_end, unsigned int *count, volatile int *h_flag, int seq)
{
  double s = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) s += v[i];
  if (s == 12345.678) v[0] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const unsigned int c = atomicAdd(count, 1u);
    if (c == gridDim.x - 1) {
      *count = 0;
      *t_end = wall_clock64();
      if (h_flag) { __threadfence_system(); *h_flag = seq; }
    }
  }
}
__global__ void k2(double *v, int n, unsigned long long *t_start, double arg, const volatile double *h_arg)
{
  if (blockIdx.x == 0 && threadIdx.x == 0) *t_start = wall_clock64();
  const double a = h_arg ? *h_arg : arg;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) v[i] = v[i] * a;
}
__global__ void k_gate(const volatile int *h_go, int seq, int *timed_out)
{
  for (int spin = 0; spin < 20000000; spin++) {
    if (*h_go == seq) return;
    __builtin_amdgcn_s_sleep(2);
  }
  *timed_out = 1;
}

int main()
{
  const int n = 100000, reps = 300;
  double *v;
  unsigned long long *t, h_t[2];
  unsigned int *count;
  int *h_flag, *h_go, *d_to;
  double *h_arg;
  CHK(hipMalloc(&v, n * sizeof(double)));
  CHK(hipMemset(v, 0, n * sizeof(double)));
  CHK(hipMalloc(&t, 2 * sizeof(unsigned long long)));
  CHK(hipMalloc(&count, sizeof(unsigned int)));
  CHK(hipMemset(count, 0, sizeof(unsigned int)));
  CHK(hipMalloc(&d_to, sizeof(int)));
  CHK(hipMemset(d_to, 0, sizeof(int)));
  CHK(hipHostMalloc((void **)&h_flag, sizeof(int), hipHostMallocMapped));
  CHK(hipHostMalloc((void **)&h_go, sizeof(int), hipHostMallocMapped));
  CHK(hipHostMalloc((void **)&h_arg, sizeof(double), hipHostMallocMapped));
  *h_flag = 0, *h_go = 0, *h_arg = 1.0;
  hipStream_t st;
  CHK(hipStreamCreate(&st));
  for (int mode = 0; mode < 3; mode++) {
    std::vector<double> gaps;
    for (int r = 1; r <= reps; r++) {
      const int seq = mode * 100000 + r;
      if (mode == 0) {
        hipLaunchKernelGGL(k1, dim3(400), dim3(256), 0, st, v, n, t, count, (volatile int *)nullptr, seq);
        CHK(hipStreamSynchronize(st));
        hipLaunchKernelGGL(k2, dim3(400), dim3(256), 0, st, v, n, t + 1, 1.0, (const volatile double *)nullptr);
      } else if (mode == 1) {
        hipLaunchKernelGGL(k1, dim3(400), dim3(256), 0, st, v, n, t, count, (volatile int *)h_flag, seq);
        while (*(volatile int *)h_flag != seq) {}
        hipLaunchKernelGGL(k2, dim3(400), dim3(256), 0, st, v, n, t + 1, 1.0, (const volatile double *)nullptr);
      } else {
        hipLaunchKernelGGL(k1, dim3(400), dim3(256), 0, st, v, n, t, count, (volatile int *)h_flag, seq);
        hipLaunchKernelGGL(k_gate, dim3(1), dim3(1), 0, st, (const volatile int *)h_go, seq, d_to);
        hipLaunchKernelGGL(k2, dim3(400), dim3(256), 0, st, v, n, t + 1, 1.0, (const volatile double *)h_arg);
        while (*(volatile int *)h_flag != seq) {}
        *(volatile double *)h_arg = 1.0; // "the host computed the step length"
        __sync_synchronize();
        *(volatile int *)h_go = seq;
      }
      CHK(hipStreamSynchronize(st));
      CHK(hipMemcpy(h_t, t, sizeof(h_t), hipMemcpyDeviceToHost));
      if (r > 20) gaps.push_back((double)(h_t[1] - h_t[0]) * 0.01); // 100 MHz -> us
    }
    std::sort(gaps.begin(), gaps.end());
    double s = 0;
    for (double g : gaps) s += g;
    printf("mode %c: gap end(kernel1) -> start(kernel2): mean %.2f us, median %.2f, p10 %.2f, p90 %.2f\n", "ABC"[mode], s / gaps.size(), gaps[gaps.size() / 2], gaps[gaps.size() / 10], gaps[gaps.size() * 9 / 10]);
  }
  int to = 0;
  CHK(hipMemcpy(&to, d_to, sizeof(int), hipMemcpyDeviceToHost));
  printf("gate time-outs: %d\n", to);
  return 0;
}
