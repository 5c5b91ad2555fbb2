// Attainable FP64 MFMA rate on this box: every SIMD of every CU issues v_mfma_f64_16x16x4_f64 back to back on
// NACC independent accumulators, W waves per SIMD.  Prints TFLOP/s and the implied clock.
//   hipcc -O3 --offload-arch=gfx950 mfma_f64_peak.hip -o mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_peak(double* out, int iters, double a0, double b0) {
  v4d acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = v4d{0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)   // inline asm: the accumulators stay where they are (the builtin made hipcc shuttle
                                      // them between VGPRs and AGPRs around the loop back-edge: 35 instead of ~peak)
      asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int wg_per_cu, int ncu) {
  const int iters = 4000;
  const int blocks = ncu * wg_per_cu;
  double* out;
  hipMalloc(&out, sizeof(double) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_peak<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-3);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = double(blocks) * 4 /*waves*/ * iters * NACC * 2048.0;
    const double tf = flops / (ms * 1e-3) / 1e12;
    // per SIMD: 2048 flop per 64 cycles = 32 flop/clk
    printf("NACC %d  %d WG/CU: %.3f ms  %.1f TFLOP/s  (implied clock %.2f GHz at 32 flop/clk/SIMD)\n", NACC, wg_per_cu, ms,
           tf, tf * 1e12 / (ncu * 4 * 32.0) / 1e9);
  }
  hipFree(out);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("%s  CUs %d  clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  run<1>(1, p.multiProcessorCount);
  run<4>(1, p.multiProcessorCount);
  run<12>(1, p.multiProcessorCount);
  run<12>(2, p.multiProcessorCount);
  run<4>(4, p.multiProcessorCount);
  return 0;
}
