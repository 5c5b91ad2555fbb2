// Device-side helpers shared by the kernels of libmpsengine.so (64-lane wavefront reductions).
#pragma once
#include <hip/hip_runtime.h>

constexpr int RED_THREADS = 256;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// Block-wide (256 threads) sum of two values in a fixed order; the totals are returned to
// EVERY thread.  Contains two barriers; safe to call repeatedly.
__device__ __forceinline__ void block_allsum2(double& a, double& b) {
  __shared__ double s_red[2 * (RED_THREADS / 64) + 2];
  a = wave_sum(a);
  b = wave_sum(b);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();  // previous use of s_red fully consumed
  if (lane == 0) {
    s_red[2 * w] = a;
    s_red[2 * w + 1] = b;
  }
  __syncthreads();
  double x = 0, y = 0;
#pragma unroll
  for (int i = 0; i < RED_THREADS / 64; ++i) {
    x += s_red[2 * i];
    y += s_red[2 * i + 1];
  }
  a = x;
  b = y;
}
