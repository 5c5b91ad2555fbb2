// Hot-path contractions: environment update and effective-Hamiltonian matvec.
// The index algebra lives in mpse_plans.h (a list of strided-GEMM steps); this file
// executes a plan on the device with the FP64-MFMA contraction kernel.
#include "mpse_internal.h"
#include "mpse_plans.h"

using namespace mpse_plan;

namespace {

__device__ __forceinline__ long long off2(const mpse_index m, long long i) {
  if (m.lo_ext >= m.ext) return i * m.s_lo;
  const long long hi = i / m.lo_ext;
  return hi * m.s_hi + (i - hi * m.lo_ext) * m.s_lo;
}

// dst(i,j) = src(i,j) through two-level row / column maps; one 16 B (complex) or 8 B element per thread,
// threads run along j so contiguous columns coalesce
template <typename T>
__global__ __launch_bounds__(256) void k_copy_strided(T* dst, const T* src, mpse_index mi, mpse_index ni, mpse_index mo,
                                                      mpse_index no, const int* skip) {
  if (skip && *skip) return;
  const long long nn = ni.ext;
  const long long total = mi.ext * nn;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const long long i = t / nn, j = t - i * nn;
    dst[off2(mo, i) + off2(no, j)] = src[off2(mi, i) + off2(ni, j)];
  }
}

// dev[b] = max_{a,c} | E[a,b,c] - delta(a,c) | for an environment E (D, w, D); doubles are non-negative, so
// their bit patterns order like integers and atomicMax on the 64-bit pattern is an exact max
template <bool CPLX>
__global__ __launch_bounds__(256) void k_unit_deviation(const double* env, int D, int w, unsigned long long* dev) {
  constexpr int E = CPLX ? 2 : 1;
  const int a = blockIdx.x;
  __shared__ double red[4];
  for (int b = 0; b < w; ++b) {
    const double* row = env + ((long long)a * w + b) * D * E;
    double m = 0.0;
    for (int c = threadIdx.x; c < D; c += 256) {
      double re = row[c * E] - (c == a ? 1.0 : 0.0);
      double v = fabs(re);
      if (CPLX) v = fmax(v, fabs(row[c * E + 1]));
      if (!(v <= 1e300)) v = 1e300;  // NaN / inf never pass for a unit channel
      m = fmax(m, v);
    }
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      m = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
      atomicMax(dev + b, (unsigned long long)__double_as_longlong(m));
    }
    __syncthreads();
  }
}

int copy_call(mpse_ctx* ctx, int dtype, const void* src, void* dst, mpse_index mi, mpse_index ni, mpse_index mo,
              mpse_index no) {
  const long long total = mi.ext * ni.ext;
  if (total <= 0) return MPSE_OK;
  long long nb = (total + 255) / 256;
  if (nb > 65536) nb = 65536;
  if (dtype == MPSE_C128)
    hipLaunchKernelGGL((k_copy_strided<double2>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, (double2*)dst,
                       (const double2*)src, mi, ni, mo, no, ctx->skip_flag);
  else
    hipLaunchKernelGGL((k_copy_strided<double>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, (double*)dst,
                       (const double*)src, mi, ni, mo, no, ctx->skip_flag);
  MPSE_HIP(ctx, hipGetLastError());
  return MPSE_OK;
}

}  // namespace

static int run_plan(mpse_ctx* ctx, int dtype, const Plan& p, const void* bufs_in[B_COUNT]) {
  if (p.error) return mpse_fail(ctx, MPSE_ERR_SHAPE, "%s", p.error);
  const void* bufs[B_COUNT];
  for (int i = 0; i < B_COUNT; ++i) bufs[i] = bufs_in[i];
  TmpBuf t1(ctx), t2(ctx), t3(ctx);
  const size_t es = dtype_size(dtype);
  if (p.tmp_elems[0]) {
    MPSE_TRY(t1.alloc(size_t(p.tmp_elems[0]) * es));
    bufs[B_T1] = t1.p;
  }
  if (p.tmp_elems[1]) {
    MPSE_TRY(t2.alloc(size_t(p.tmp_elems[1]) * es));
    bufs[B_T2] = t2.p;
  }
  if (p.tmp_elems[2]) {
    MPSE_TRY(t3.alloc(size_t(p.tmp_elems[2]) * es));
    bufs[B_T3] = t3.p;
  }
  for (const Step& s : p.steps) {
    const char* a = (const char*)bufs[s.a] + size_t(s.a_off) * dtype_size(s.dta);
    const char* b = (const char*)bufs[s.b] + size_t(s.b_off) * dtype_size(s.dtb);
    const int dtc = (s.dta == MPSE_C128 || s.dtb == MPSE_C128) ? MPSE_C128 : MPSE_F64;
    if (dtc != dtype)  // every step of a plan must produce the working dtype
      return mpse_fail(ctx, MPSE_ERR_ARG, "plan step dtype mismatch (got %d want %d)", dtc, dtype);
    char* c = (char*)const_cast<void*>(bufs[s.c]) + size_t(s.c_off) * dtype_size(dtc);
    if (!bufs[s.a] || !bufs[s.b] || !bufs[s.c]) return mpse_fail(ctx, MPSE_ERR_ARG, "plan: missing buffer");
    if (s.kind == K_COPY) {
      MPSE_TRY(copy_call(ctx, dtc, a, c, s.ma, s.ka, s.mc, s.nc));
      continue;
    }
    MPSE_TRY(gemm_call(ctx, s.dta, s.dtb, s.conja, s.conjb, s.ma, s.ka, s.kb, s.nb, s.mc, s.nc, s.batch, s.sba,
                       s.sbb, s.sbc, a, b, c, 1.0, s.beta, s.skip_zero));
  }
  return MPSE_OK;
}

extern "C" int mpse_heff_apply(mpse_ctx* ctx, int dtype, const mpse_heff* h, const void* C, void* out) {
  if (!ctx || !h || !C || !out || !h->L || !h->R) return MPSE_ERR_ARG;
  MPSE_BIND(ctx);
  if (dtype != MPSE_C128 && (h->l_dtype == MPSE_C128 || h->r_dtype == MPSE_C128 || h->w_dtype == MPSE_C128))
    return mpse_fail(ctx, MPSE_ERR_ARG, "heff_apply: real centre with complex operator parts");
  Plan p = plan_heff(dtype, *h);
  const void* bufs[B_COUNT] = {nullptr};
  bufs[B_L] = h->L;
  bufs[B_R] = h->R;
  bufs[B_W0] = h->W0;
  bufs[B_W1] = h->W1;
  bufs[B_C] = C;
  bufs[B_OUT] = out;
  return run_plan(ctx, dtype, p, bufs);
}

extern "C" int mpse_env_update(mpse_ctx* ctx, int dtype, int domain, const mpse_dims* dims, const void* env,
                               int env_dtype, const void* ket, const void* bra, int bra_conj, const void* W,
                               int w_dtype, void* out) {
  if (!ctx || !dims || !env || !ket || !W || !out) return MPSE_ERR_ARG;
  MPSE_BIND(ctx);
  if (dtype != MPSE_C128 && (env_dtype == MPSE_C128 || w_dtype == MPSE_C128))
    return mpse_fail(ctx, MPSE_ERR_ARG, "env_update: real sites with complex env/mpo");
  if (!bra) {
    bra = ket;
    if (dims->Dl_bra != dims->Dl_ket || dims->Dr_bra != dims->Dr_ket)
      return mpse_fail(ctx, MPSE_ERR_SHAPE, "env_update: bra==NULL needs equal bonds");
  }
  Plan p = plan_env(dtype, domain, *dims, env_dtype, w_dtype, bra_conj);
  const void* bufs[B_COUNT] = {nullptr};
  bufs[B_L] = env;
  bufs[B_W0] = W;
  bufs[B_C] = ket;
  bufs[B_BRA] = bra;
  bufs[B_OUT] = out;
  return run_plan(ctx, dtype, p, bufs);
}

extern "C" int mpse_env_update_multi(mpse_ctx* ctx, int dtype, int domain, const mpse_dims* dims, int n_mpo,
                                     const int64_t* wl, const int64_t* wr, const void* env, int env_dtype,
                                     const void* ket, const void* bra, int bra_conj, const void* const* W, int w_dtype,
                                     void* out) {
  if (!ctx || !dims || !env || !ket || !W || !wl || !wr || !out) return MPSE_ERR_ARG;
  MPSE_BIND(ctx);
  if (n_mpo < 1 || n_mpo > 4) return mpse_fail(ctx, MPSE_ERR_ARG, "env_update_multi: 1 to 4 MPO layers, got %d", n_mpo);
  if (dtype != MPSE_C128 && (env_dtype == MPSE_C128 || w_dtype == MPSE_C128))
    return mpse_fail(ctx, MPSE_ERR_ARG, "env_update_multi: real sites with complex env/mpo");
  if (!bra) {
    bra = ket;
    if (dims->Dl_bra != dims->Dl_ket || dims->Dr_bra != dims->Dr_ket)
      return mpse_fail(ctx, MPSE_ERR_SHAPE, "env_update_multi: bra==NULL needs equal bonds");
  }
  Plan p = plan_env_multi(dtype, domain, *dims, n_mpo, wl, wr, env_dtype, w_dtype, bra_conj);
  const void* bufs[B_COUNT] = {nullptr};
  bufs[B_L] = env;
  for (int i = 0; i < n_mpo; ++i) {
    if (!W[i]) return MPSE_ERR_ARG;
    bufs[w_buf(i)] = W[i];
  }
  bufs[B_C] = ket;
  bufs[B_BRA] = bra;
  bufs[B_OUT] = out;
  return run_plan(ctx, dtype, p, bufs);
}

extern "C" int mpse_heff_apply2(mpse_ctx* ctx, int dtype, const mpse_heff* h, const void* C, void* out) {
  if (!ctx || !h || !C || !out || !h->L || !h->R || !h->W0) return MPSE_ERR_ARG;
  MPSE_BIND(ctx);
  if (dtype != MPSE_C128 && (h->l_dtype == MPSE_C128 || h->r_dtype == MPSE_C128 || h->w_dtype == MPSE_C128))
    return mpse_fail(ctx, MPSE_ERR_ARG, "heff_apply2: real centre with complex operator parts");
  Plan p = plan_heff2(dtype, *h);
  const void* bufs[B_COUNT] = {nullptr};
  bufs[B_L] = h->L;
  bufs[B_R] = h->R;
  bufs[B_W0] = h->W0;
  bufs[B_W1] = h->W1;
  bufs[B_C] = C;
  bufs[B_OUT] = out;
  return run_plan(ctx, dtype, p, bufs);
}

extern "C" int mpse_env_unit_channel(mpse_ctx* ctx, int dtype, const void* env, int64_t D, int64_t w, double tol,
                                     int64_t* unit_host) {
  if (!ctx || !env || !unit_host) return MPSE_ERR_ARG;
  MPSE_BIND(ctx);
  *unit_host = 0;
  if (D <= 0 || w <= 0 || w > 2048) return MPSE_OK;
  unsigned long long* dev = reinterpret_cast<unsigned long long*>(ctx->dscratch);
  MPSE_HIP(ctx, hipMemsetAsync(dev, 0, size_t(w) * sizeof(double), ctx->stream));
  if (dtype == MPSE_C128)
    hipLaunchKernelGGL((k_unit_deviation<true>), dim3((unsigned)D), dim3(256), 0, ctx->stream, (const double*)env, (int)D,
                       (int)w, dev);
  else
    hipLaunchKernelGGL((k_unit_deviation<false>), dim3((unsigned)D), dim3(256), 0, ctx->stream, (const double*)env,
                       (int)D, (int)w, dev);
  MPSE_HIP(ctx, hipGetLastError());
  if (w <= 1024) {
    MPSE_TRY(publish_and_wait(ctx, reinterpret_cast<const double*>(dev), (int)w, 32));
  } else {
    MPSE_HIP(ctx, hipMemcpyAsync(ctx->pinned + 32, dev, size_t(w) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    MPSE_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  for (int64_t b = 0; b < w; ++b)
    if (ctx->pinned[32 + b] <= tol) {
      *unit_host = b + 1;
      break;
    }
  return MPSE_OK;
}
