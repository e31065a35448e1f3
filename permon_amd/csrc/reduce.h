// Deterministic block-level reductions for 64-wide wavefronts (gfx950).
// Every reduction has a fixed tree shape: lane shuffle tree inside a wave, then the 4 wave
// results through LDS in wave order; block partials are combined by pmh_finalize_partials in a
// fixed order too, so a given (n, grid) always sums in the same order (no atomics).
#pragma once
#include <hip/hip_runtime.h>

#include "pmh_internal.h"

__device__ __forceinline__ double pmh_wave_sum(double v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

__device__ __forceinline__ double pmh_wave_min(double v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_down(v, o, 64));
  return v;
}

// result valid in thread 0; `lds` needs PMH_BLOCK/64 doubles; includes the barriers it needs
template <int OP>
__device__ __forceinline__ double pmh_block_reduce(double v, double *lds)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = (OP == PMH_RED_SUM) ? pmh_wave_sum(v) : pmh_wave_min(v);
  __syncthreads(); // protect lds reuse across consecutive calls
  if (lane == 0) lds[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double r = lds[0];
#pragma unroll
    for (int w = 1; w < PMH_BLOCK / 64; w++) r = (OP == PMH_RED_SUM) ? (r + lds[w]) : fmin(r, lds[w]);
    v = r;
  }
  return v;
}
