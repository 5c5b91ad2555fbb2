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

// ---- MPO step of the masked one-site chain (mpse_plans.h: K_WSTEP)
constexpr int WS_MAXE = 128;   // most (b, e) pairs feeding one (d, f): wl * d, checked on the host

struct WsEntry {
  int be;        // b * d + e
  int pad;
  double val;
};

// sparse form of the MPO site: for every (d, f) the list of (b, e) with W[b, d, e, f] != 0, stored compactly:
// cnt[o], ptr[o] (o = d * wr + f) and ent[ptr[o] ..]; cnt[d * wr] = total number of entries.  One workgroup.
__global__ __launch_bounds__(256) void k_w_csr(const double* __restrict__ W, int wl, int d, int wr, int* __restrict__ cnt,
                                               WsEntry* __restrict__ ent) {
  __shared__ int s_cnt[1024], s_ptr[1025];
  const int no = d * wr;
  for (int o = threadIdx.x; o < no; o += blockDim.x) {
    const int dd = o / wr, f = o - dd * wr;
    int c = 0;
    for (int b = 0; b < wl; ++b)
      for (int e = 0; e < d; ++e)
        if (W[(((long long)b * d + dd) * d + e) * wr + f] != 0.0) ++c;
    s_cnt[o] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int o = 0; o < no; ++o) {
      s_ptr[o] = acc;
      acc += s_cnt[o];
    }
    s_ptr[no] = acc;
  }
  __syncthreads();
  for (int o = threadIdx.x; o < no; o += blockDim.x) {
    const int dd = o / wr, f = o - dd * wr;
    int c = s_ptr[o];
    for (int b = 0; b < wl; ++b)
      for (int e = 0; e < d; ++e) {
        const double v = W[(((long long)b * d + dd) * d + e) * wr + f];
        if (v != 0.0) ent[c++] = WsEntry{b * d + e, 0, v};
      }
    cnt[o] = s_cnt[o];
    cnt[no + 1 + o] = s_ptr[o];
  }
  if (threadIdx.x == 0) cnt[no] = s_ptr[no];
}

// ---- MPO step of a small site (wl d <= 16 inputs, d wr <= 16 outputs per bond state and trailing index: the d = 2
// sites of a Holstein chain):  T2[a, dd, f, n] = sum_{b,e} W[b, dd, e, f] T1[b, a, e, n].  As a batched MFMA product
// this is 8 x 8 x 256 per bond state padded to 64 x 64 tiles (14 us); here one thread owns one (a, n), reads its
// inputs (coalesced along n), multiplies by W out of LDS and stores its outputs: two passes over 18 MB.
struct WsmArgs {
  const double* T1;
  double* T2;
  const double* W;
  int Da, d, wl, wr, N;
  const int* skip;
};

template <bool CPLX>
__global__ __launch_bounds__(256) void k_wsmall(const WsmArgs g) {
  if (g.skip && *g.skip) return;
  constexpr int E = CPLX ? 2 : 1;
  __shared__ double s_w[16][16];   // [o = dd * wr + f][be = b * d + e]
  const int nrow = g.wl * g.d, no = g.d * g.wr;
  {
    const int o = threadIdx.x >> 4, be = threadIdx.x & 15;
    double v = 0.0;
    if (o < no && be < nrow) {
      const int dd = o / g.wr, f = o - dd * g.wr, b = be / g.d, e = be - b * g.d;
      v = g.W[(((long long)b * g.d + dd) * g.d + e) * g.wr + f];
    }
    s_w[o][be] = v;
  }
  __syncthreads();
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)g.Da * g.N) return;
  const int a = (int)(idx / g.N), n = (int)(idx - (long long)a * g.N);
  double xr[16], xi[16];
#pragma unroll
  for (int be = 0; be < 16; ++be) {
    xr[be] = 0.0;
    xi[be] = 0.0;
    if (be < nrow) {
      const int b = be / g.d, e = be - b * g.d;
      const double* src = g.T1 + ((((long long)b * g.Da + a) * g.d + e) * g.N + n) * E;
      xr[be] = src[0];
      if constexpr (CPLX) xi[be] = src[1];
    }
  }
#pragma unroll
  for (int o = 0; o < 16; ++o) {
    if (o >= no) break;
    double ar = 0.0, ai = 0.0;
#pragma unroll
    for (int be = 0; be < 16; ++be) {
      const double w = s_w[o][be];   // zero beyond nrow
      ar += w * xr[be];
      if constexpr (CPLX) ai += w * xi[be];
    }
    double* dst = g.T2 + (((long long)a * no + o) * g.N + n) * E;
    dst[0] = ar;
    if constexpr (CPLX) dst[1] = ai;
  }
}

struct WsArgs {
  const double* T1;
  const double* X0;       // centre tensor (unit channel of the left environment), or null
  double* T2;
  const int* cnt;
  const WsEntry* ent;
  const unsigned char* m1_lo;   // tile flags of T1's producers: channels below / above the unit channel
  const unsigned char* m1_hi;
  unsigned char* m2_lo;         // tile masks of T2 for its consumers: f below / above the right unit channel
  unsigned char* m2_hi;
  int Da, d, wl, wr, Dk;
  int l_unit, r_unit;
  int t1_tiles_n, nkw_lo, nkw_hi;
  const int* skip;
};

// One workgroup per (64 rows of the (a, d) index = 64 / d values of a, 64 consecutive k); lane = k.
//   1. the producers' tile flags of all input rows (b, a, e) of the tile and the sparse MPO site go to LDS;
//   2. which f receive anything in this tile (flags only) -> consumer masks;
//   3. per a: the flagged input rows X[b][a][e][k0 .. k0 + 63] are staged in LDS, eight independent coalesced 1 KB
//      loads in flight per wave (the kernel must be bandwidth bound, not a chain of dependent loads), then every
//      wave forms its share of the d x wr output rows from LDS and stores them (1 KB each).
// A flagged f is stored for ALL rows of the tile, zeros included, so that a flagged tile is completely defined.
constexpr int WS_MAXROWS = 96;    // wl * d rows of 64 elements staged per a: 96 KB (complex) of LDS
constexpr int WS_LDS_ENT = 1024;  // sparse MPO entries kept in LDS (more: read from global memory)

template <bool CPLX>
__global__ __launch_bounds__(256) void k_wstep(const WsArgs g) {
  if (g.skip && *g.skip) return;
  constexpr int E = CPLX ? 2 : 1;
  extern __shared__ double s_x[];                   // [wl * d][64][E]
  __shared__ unsigned char s_flag[WS_MAXROWS * 64]; // [b * d + e][ai]  (ai = a - a0 < 64 / d)
  __shared__ WsEntry s_ent[WS_LDS_ENT];
  __shared__ int s_cnt[1024], s_ptr[1024];
  __shared__ int s_rows[WS_MAXROWS];
  __shared__ int s_n;
  __shared__ unsigned int s_any[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tid = threadIdx.x;
  const int k0 = blockIdx.x * 64, k = k0 + lane;
  const int tm = blockIdx.y;
  const int d = g.d, wr = g.wr, wl = g.wl;
  const int na = 64 / d;                          // d divides 64 (checked on the host)
  const int a0 = tm * na;
  const int nrow = wl * d, no = d * wr;
  // ---- 1. flags and sparse MPO site
  for (int t = tid; t < nrow * na; t += 256) {
    const int be = t / na, ai = t - be * na;
    const int b = be / d, e = be - b * d;
    const int a = a0 + ai;
    unsigned char f = 0;
    if (a < g.Da) {
      if (b == g.l_unit) {
        f = 1;
      } else {
        const bool low = g.l_unit < 0 || b < g.l_unit;
        const int b0 = low ? 0 : g.l_unit + 1;
        const long long row = (long long)(b - b0) * g.Da + a;
        const long long tn = ((long long)e * g.Dk + k0) >> 6;
        f = (low ? g.m1_lo : g.m1_hi)[(row >> 6) * g.t1_tiles_n + tn];
      }
    }
    s_flag[be * 64 + ai] = f;
  }
  const int nnz = g.cnt[no];
  const bool ent_lds = nnz <= WS_LDS_ENT;
  for (int o = tid; o < no; o += 256) {
    s_cnt[o] = g.cnt[o];
    s_ptr[o] = g.cnt[no + 1 + o];
  }
  if (ent_lds)
    for (int i = tid; i < nnz; i += 256) s_ent[i] = g.ent[i];
  const WsEntry* ents = ent_lds ? s_ent : g.ent;
  __syncthreads();
  // ---- 2. which f receive anything: outputs (ai, dd, f), one per thread and pass
  unsigned int anyf = 0;
  for (int t = tid; t < na * d * wr; t += 256) {
    const int f = t % wr, rest = t / wr;
    const int dd = rest % d, ai = rest / d;
    if (a0 + ai >= g.Da) continue;
    const int o = dd * wr + f, c = s_cnt[o];
    const WsEntry* en = ents + s_ptr[o];
    for (int i = 0; i < c; ++i)
      if (s_flag[en[i].be * 64 + ai]) {
        anyf |= 1u << f;
        break;
      }
  }
  for (int o = 32; o > 0; o >>= 1) anyf |= __shfl_xor(anyf, o, 64);
  if (lane == 0) s_any[wave] = anyf;
  __syncthreads();
  anyf = s_any[0] | s_any[1] | s_any[2] | s_any[3];
  if (tid < wr * 4) {   // consumer masks: one byte per (f, 16 k); this workgroup owns the four bytes of its 64 k
    const int f = tid >> 2, sub = tid & 3;
    if (f != g.r_unit) {
      const bool low = g.r_unit < 0 || f < g.r_unit;
      const int f0 = low ? 0 : g.r_unit + 1;
      const long long kt = ((long long)(f - f0) * g.Dk + k0) / 16 + sub;
      unsigned char* m = low ? g.m2_lo : g.m2_hi;
      const long long nkw = low ? g.nkw_lo : g.nkw_hi;
      m[(long long)tm * nkw * 8 + kt] = (anyf >> f) & 1u;
    }
  }
  if (g.r_unit >= 0) anyf |= 1u << g.r_unit;
  if (anyf == 0) return;
  // ---- 3. per a: stage flagged rows, then compute
  for (int ai = 0; ai < na; ++ai) {
    const int a = a0 + ai;
    if (a >= g.Da) break;
    if (tid == 0) s_n = 0;
    __syncthreads();                                // previous a's rows fully consumed; counter reset
    if (tid < nrow && s_flag[tid * 64 + ai]) s_rows[atomicAdd(&s_n, 1)] = tid;
    __syncthreads();
    const int nr = s_n;
    for (int base = wave * 8; base < nr; base += 32) {
      double2 v[8];
      int bes[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u;
        bes[u] = idx < nr ? s_rows[idx] : -1;
        v[u] = make_double2(0.0, 0.0);
        if (bes[u] >= 0) {
          const int b = bes[u] / d, e = bes[u] - b * d;
          const double* src = (b == g.l_unit) ? g.X0 + (((long long)a * d + e) * g.Dk + k) * E
                                              : g.T1 + ((((long long)b * g.Da + a) * d + e) * g.Dk + k) * E;
          if (CPLX)
            v[u] = *reinterpret_cast<const double2*>(src);
          else
            v[u].x = src[0];
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (bes[u] >= 0) {
          if (CPLX)
            reinterpret_cast<double2*>(s_x)[bes[u] * 64 + lane] = v[u];
          else
            s_x[bes[u] * 64 + lane] = v[u].x;
        }
    }
    __syncthreads();
    for (int t = wave; t < no; t += 4) {
      const int f = t % wr;
      if (!((anyf >> f) & 1u)) continue;
      const int dd = t / wr;
      const int c = s_cnt[t];
      const WsEntry* en = ents + s_ptr[t];
      double xr = 0.0, xi = 0.0;
      for (int i = 0; i < c; ++i) {
        const WsEntry w = en[i];
        if (!s_flag[w.be * 64 + ai]) continue;
        if (CPLX) {
          const double2 x = reinterpret_cast<const double2*>(s_x)[w.be * 64 + lane];
          xr += w.val * x.x;
          xi += w.val * x.y;
        } else {
          xr += w.val * s_x[w.be * 64 + lane];
        }
      }
      double* dst = g.T2 + ((((long long)a * d + dd) * wr + f) * g.Dk + k) * E;
      if (CPLX)
        *reinterpret_cast<double2*>(dst) = make_double2(xr, xi);
      else
        dst[0] = xr;
    }
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
  TmpBuf msk[M_COUNT] = {TmpBuf(ctx), TmpBuf(ctx), TmpBuf(ctx), TmpBuf(ctx)};
  for (int i = 0; i < M_COUNT; ++i)
    if (p.mask_bytes[i] > 0) MPSE_TRY(msk[i].alloc(size_t(p.mask_bytes[i]) + 64));
  // MPSE_WSMALL=0: the MPO step of small sites as a batched MFMA product like the large ones
  static const bool wsmall_on = [] {
    const char* e = getenv("MPSE_WSMALL");
    return !(e && e[0] == '0');
  }();
  for (const Step& s : p.steps) {
    if (s.kind == K_GEMM && s.is_wstep && wsmall_on && s.dta == MPSE_F64 && s.a_off == 0 && s.b_off == 0 &&
        s.c_off == 0 && s.beta == 0.0 && s.w_wl * s.w_d <= 16 && s.w_d * s.w_wr <= 16) {
      if (!bufs[s.a] || !bufs[s.b] || !bufs[s.c]) return mpse_fail(ctx, MPSE_ERR_ARG, "plan: missing buffer");
      WsmArgs g;
      g.T1 = (const double*)bufs[s.b];
      g.T2 = (double*)const_cast<void*>(bufs[s.c]);
      g.W = (const double*)bufs[s.a];
      g.Da = (int)s.w_Da, g.d = (int)s.w_d, g.wl = (int)s.w_wl, g.wr = (int)s.w_wr, g.N = (int)s.w_N;
      g.skip = ctx->skip_flag;
      const long long total = s.w_Da * s.w_N;
      const dim3 grid((unsigned)((total + 255) / 256));
      if (s.dtb == MPSE_C128)
        hipLaunchKernelGGL((k_wsmall<true>), grid, dim3(256), 0, ctx->stream, g);
      else
        hipLaunchKernelGGL((k_wsmall<false>), grid, dim3(256), 0, ctx->stream, g);
      MPSE_HIP(ctx, hipGetLastError());
      continue;
    }
    if (s.kind == K_WSTEP) {
      const WStepDesc& w = s.ws;
      if (w.wl * w.d > WS_MAXROWS || w.wr > 32 || 64 % w.d != 0 || w.d * w.wr > 1024)
        return mpse_fail(ctx, MPSE_ERR_SHAPE, "masked MPO step: MPO site outside the supported range");
      // sparse form of the MPO site: built once per Krylov solve (the site does not change), else per call
      TmpBuf CNT(ctx), ENT(ctx);
      int* cntp = nullptr;
      WsEntry* entp = nullptr;
      if (ctx->occ_cache_on)
        for (const auto& e : ctx->wcsr_cache)
          if (e.w == bufs[s.b] && e.wl == w.wl && e.d == w.d && e.wr == w.wr) {
            cntp = static_cast<int*>(e.cnt);
            entp = static_cast<WsEntry*>(e.ent);
          }
      if (!cntp) {
        const size_t cb = size_t(2 * w.d * w.wr + 2) * sizeof(int), eb = size_t(w.wl * w.d * w.d * w.wr) * sizeof(WsEntry);
        if (ctx->occ_cache_on) {
          void *pc = nullptr, *pe = nullptr;
          MPSE_TRY(mpse_malloc(ctx, cb, &pc));
          MPSE_TRY(mpse_malloc(ctx, eb, &pe));
          ctx->wcsr_cache.push_back({bufs[s.b], w.wl, w.d, w.wr, pc, pe});
          cntp = static_cast<int*>(pc), entp = static_cast<WsEntry*>(pe);
        } else {
          MPSE_TRY(CNT.alloc(cb));
          MPSE_TRY(ENT.alloc(eb));
          cntp = CNT.as<int>(), entp = ENT.as<WsEntry>();
        }
        // (not subject to the skip flag: a cached table must be complete whenever it is used)
        hipLaunchKernelGGL(k_w_csr, dim3(1), dim3(256), 0, ctx->stream, (const double*)bufs[s.b], (int)w.wl, (int)w.d,
                           (int)w.wr, cntp, entp);
      }
      WsArgs g;
      g.T1 = (const double*)bufs[B_T1];
      g.X0 = (const double*)bufs[s.a];
      g.T2 = (double*)const_cast<void*>(bufs[s.c]);
      g.cnt = cntp;
      g.ent = entp;
      g.m1_lo = msk[M_T1_LO].as<unsigned char>();
      g.m1_hi = msk[M_T1_HI].as<unsigned char>();
      g.m2_lo = msk[M_T2_LO].as<unsigned char>();
      g.m2_hi = msk[M_T2_HI].as<unsigned char>();
      g.Da = (int)w.Da, g.d = (int)w.d, g.wl = (int)w.wl, g.wr = (int)w.wr, g.Dk = (int)w.Dk;
      g.l_unit = (int)w.l_unit, g.r_unit = (int)w.r_unit;
      g.t1_tiles_n = (int)w.t1_tiles_n, g.nkw_lo = (int)w.nkw_lo, g.nkw_hi = (int)w.nkw_hi;
      g.skip = ctx->skip_flag;
      const dim3 grid((unsigned)(w.Dk / 64), (unsigned)((w.Da * w.d + 63) / 64));
      const size_t lds = size_t(w.wl * w.d) * 64 * dtype_size(dtype);
      static const bool lds_attr = [] {   // beyond the default 64 KB of dynamic LDS (the CU has 160 KB)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wstep<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  WS_MAXROWS * 64 * 16);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wstep<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  WS_MAXROWS * 64 * 16);
        (void)hipGetLastError();
        return true;
      }();
      (void)lds_attr;
      if (dtype == MPSE_C128)
        hipLaunchKernelGGL((k_wstep<true>), grid, dim3(256), lds, ctx->stream, g);
      else
        hipLaunchKernelGGL((k_wstep<false>), grid, dim3(256), lds, ctx->stream, g);
      MPSE_HIP(ctx, hipGetLastError());
      continue;
    }
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
    // the last step completes the result: it takes the caller's dot request along when it is a plain product into
    // the whole of `out`
    if (ctx->dot_req.y && &s == &p.steps.back() && s.c == B_OUT && s.c_off == 0 && s.batch == 1 && s.cmask_slot < 0)
      ctx->dot_now = true;
    if (s.cin >= 0) {
      if (!bufs[s.cin]) return mpse_fail(ctx, MPSE_ERR_ARG, "plan: missing beta source");
      ctx->cin_req.ptr = (const char*)bufs[s.cin] + size_t(s.cin_off) * dtype_size(dtc);
      ctx->cin_req.m = s.mcin;
      ctx->cin_req.n = s.ncin;
    }
    const int st = gemm_call(ctx, s.dta, s.dtb, s.conja, s.conjb, s.ma, s.ka, s.kb, s.nb, s.mc, s.nc, s.batch, s.sba,
                             s.sbb, s.sbc, a, b, c, 1.0, s.beta, s.skip_zero,
                             s.amask_slot >= 0 ? msk[s.amask_slot].p : nullptr,
                             s.cmask_slot >= 0 ? msk[s.cmask_slot].p : nullptr);
    // requests the call did not take (degenerate product, error) must not reach a later one
    ctx->cin_req = mpse_ctx::CinReq();
    ctx->dot_now = false;
    MPSE_TRY(st);
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
  if (MPSE_RECORDING(ctx)) {
    const mpse_dims dc = *dims;
    ctx->defer_ops[ctx->defer_recording].push_back([=] {
      return mpse_env_update(ctx, dtype, domain, &dc, env, env_dtype, ket, bra, bra_conj, W, w_dtype, out);
    });
    return MPSE_OK;
  }
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
