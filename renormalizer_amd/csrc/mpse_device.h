// Device-side helpers shared by the kernels of libmpsengine.so (64-lane wavefront reductions).
#pragma once
#include <hip/hip_runtime.h>

constexpr int RED_THREADS = 256;

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global load and
// store of the wave (s_waitcnt vmcnt(0)): prefetches would be drained and the write latency of results that only
// later kernels read would sit on the critical path of a loop.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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

// DPP move with a bank mask: only lanes of the selected banks (4-lane groups of a 16-lane row) receive data, the
// others keep ``old``.  Two of these build a lane-xor-4 / lane-xor-8 exchange out of row shifts.
template <int CTRL, int BANK>
__device__ __forceinline__ double dpp_move_banks(double old, double v) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, 0xf, BANK, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, 0xf, BANK, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_xor4(double v) {   // row_shl:4 into banks 0,2 ; row_shr:4 into banks 1,3
  return dpp_move_banks<0x114, 0xA>(dpp_move_banks<0x104, 0x5>(0.0, v), v);
}
__device__ __forceinline__ double lane_xor8(double v) {   // row_shl:8 into banks 0,1 ; row_shr:8 into banks 2,3
  return dpp_move_banks<0x118, 0xC>(dpp_move_banks<0x108, 0x3>(0.0, v), v);
}

// Eight values summed over the 16 lanes of each row in ~60 VALU instructions instead of 8 x 4 butterfly steps:
// at every halving stage a lane keeps half of its values and trades the other half with its partner (lane xor 1,
// 2, 4), so the work halves per stage; one plain xor-8 step finishes the row.  Returns, in lane l, the row sum of
// v[rowsum8_index(l)].  (The naive per-value butterfly costs ~200 VALU cycles per value and wave and was the
// largest item of the QR panel kernel, tools/ubench/sync_cost.hip.)
__device__ __forceinline__ int rowsum8_index(int lane) { return ((lane & 1) << 2) | (lane & 2) | ((lane >> 2) & 1); }
__device__ __forceinline__ double wave_rowsum8(const double (&v)[8], int lane) {
  const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
  double w[4], u[2];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const double keep = b0 ? v[t + 4] : v[t], send = b0 ? v[t] : v[t + 4];
    w[t] = keep + dpp_move<0xb1>(send);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const double keep = b1 ? w[t + 2] : w[t], send = b1 ? w[t] : w[t + 2];
    u[t] = keep + dpp_move<0x4e>(send);
  }
  const double keep = b2 ? u[1] : u[0], send = b2 ? u[0] : u[1];
  double x = keep + lane_xor4(send);
  x += lane_xor8(x);
  return x;
}

// Cross-row steps of a wave reduction with the gfx950 row-swap instructions: v_permlane16_swap exchanges the odd
// rows of one register with the even rows of another, so swap(x, x) yields the two halves of a lane-xor-16 pair.
__device__ __forceinline__ double lane_xor16_sum(double v) {
  const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(v), __double2loint(v), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(v), __double2hiint(v), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double lane_xor32_sum(double v) {
  const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(v), __double2loint(v), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(v), __double2hiint(v), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
// wave_rowsum8 continued over the four rows: every lane ends with the full 64-lane sum of v[rowsum8_index(lane)]
__device__ __forceinline__ double wave_sum8(const double (&v)[8], int lane) {
  return lane_xor32_sum(lane_xor16_sum(wave_rowsum8(v, lane)));
}

// 1/x and 1/sqrt(x) to double precision from the hardware estimates and two Newton steps (the IEEE division /
// square root sequences are ~25 dependent instructions each; the QR panel kernel does three per column on its
// critical path).  x must be finite, non-zero (and positive for rsqrt).
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = r * (2.0 - x * r);
  r = r * (2.0 - x * r);
  return r;
}
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - 0.5 * x * y * y);
  y = y * (1.5 - 0.5 * x * y * y);
  return y;
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
