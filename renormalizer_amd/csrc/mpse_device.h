// Device-side helpers shared by the kernels of libmpsengine.so (64-lane wavefront reductions).
#pragma once
#include <hip/hip_runtime.h>

constexpr int RED_THREADS = 256;

// One DPP move of a 64-bit value (two 32-bit v_mov_dpp); lanes without a source receive 0.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi2, lo2);
}

// Sum over the 64 lanes of a wave, returned to every lane.  Data-parallel-primitive (DPP) moves run on the
// VALU - unlike __shfl_*, which lowers to ds_bpermute and queues on the CU's single LDS crossbar: with 16 waves
// reducing a dozen values each, that crossbar was the bottleneck of the QR panel kernel.
//   quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_shr:4, row_shr:8, row_bcast:15, row_bcast:31 -> total in lane 63
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_move<0xb1>(v);
  v += dpp_move<0x4e>(v);
  v += dpp_move<0x114>(v);
  v += dpp_move<0x118>(v);
  v += dpp_move<0x142>(v);
  v += dpp_move<0x143>(v);
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
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
