// Micro-benchmark: cost of the building blocks of a latency-bound single-workgroup loop on MI355X.
// hipcc --offload-arch=gfx950 -O3 sync_cost.hip -o sync_cost && ./sync_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../renormalizer_amd/csrc/mpse_device.h"

template <int MODE, int NT>
__global__ __launch_bounds__(NT) void k(double* out, double* gbuf, int iters, long long* cyc) {
  __shared__ double s[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double acc = tid * 1e-3;
  double2 x[4] = {{acc, 1.0}, {acc, 2.0}, {acc, 3.0}, {acc, 4.0}};
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 2) {
      double v[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = wave_sum(acc * (t + 1) + x[t & 3].x);
      if (lane == 0) {
#pragma unroll
        for (int t = 0; t < 8; ++t) s[(wave * 8 + t) & 63] = v[t];
      }
    }
    __syncthreads();
    if (MODE >= 3 && wave == 0) {
      double a = s[lane & 7] + 1.5;
      double nrm = sqrt(a * a + 2.0);
      double b = (nrm - a) / nrm;
      double c = 1.0 / (a - nrm * 1.0001);
      s[8 + (lane & 7)] = b + c;
    }
    __syncthreads();
    acc += s[8 + (lane & 7)] * 1e-9;
    if (MODE >= 4) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        x[q].x -= acc * 1e-3;
        reinterpret_cast<double2*>(gbuf)[(size_t)(it & 7) * NT * 4 + q * NT + tid] = x[q];
      }
    }
    if (MODE >= 5) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        double2 y = reinterpret_cast<const double2*>(gbuf)[(size_t)((it + 3) & 7) * NT * 4 + q * NT + tid + (size_t)NT * 64];
        acc += y.x * 1e-12;
      }
    }
  }
  long long t1 = clock64();
  out[tid] = acc + x[0].x;
  if (tid == 0) cyc[0] = t1 - t0;
}

template <int MODE, int NT>
void run(const char* name) {
  double *out, *g; long long* cyc;
  hipMalloc(&out, NT * 8); hipMalloc(&g, (size_t)NT * 4 * 16 * 8 * 80); hipMalloc(&cyc, 8);
  hipMemset(g, 0, (size_t)NT * 4 * 16 * 8 * 80);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  k<MODE, NT><<<1, NT>>>(out, g, iters, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE, NT><<<1, NT>>>(out, g, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-44s NT=%4d  %.3f us/iter  %.0f cycles/iter  (eff clock %.2f GHz)\n", name, NT, ms * 1e3 / iters,
         (double)c / iters, (double)c / (ms * 1e6));
  hipFree(out); hipFree(g); hipFree(cyc);
}

int main() {
  run<1, 1024>("2 barriers");
  run<2, 1024>("+ 8 DPP wave sums + LDS");
  run<3, 1024>("+ wave-0 sqrt / divisions");
  run<4, 1024>("+ 4 global stores (16 B) per thread");
  run<5, 1024>("+ 4 global loads (16 B) per thread");
  run<1, 256>("2 barriers");
  run<2, 256>("+ 8 DPP wave sums + LDS");
  run<3, 256>("+ wave-0 sqrt / divisions");
  run<4, 256>("+ 4 global stores (16 B) per thread");
  run<5, 256>("+ 4 global loads (16 B) per thread");
  return 0;
}
