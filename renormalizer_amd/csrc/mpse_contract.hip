// Hot-path contractions: environment update and effective-Hamiltonian matvec.
// The index algebra lives in mpse_plans.h (a list of strided-GEMM steps); this file
// executes a plan on the device with the FP64-MFMA contraction kernel.
#include <deque>

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
  // channels [b_lo, b_hi) of T1 arrive as `nslices` K slices of the product that forms them (slice s of channel b, row
  // a at slices + s * slice_stride + (((b - b_lo) * Da + a) * d + e) * N + n): added here instead of by a launch of
  // their own (nslices == 0: everything is read from T1)
  const double* slices;
  long long slice_stride;
  int nslices, b_lo, b_hi;
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
      if (g.nslices > 0 && b >= g.b_lo && b < g.b_hi) {
        const double* src = g.slices + ((((long long)(b - g.b_lo) * g.Da + a) * g.d + e) * g.N + n) * E;
        for (int sl = 0; sl < g.nslices; ++sl) {      // slice order: the same sum as the reduction launch forms
          xr[be] += src[0];
          if constexpr (CPLX) xi[be] += src[1];
          src += g.slice_stride * E;
        }
      } else {
        const double* src = g.T1 + ((((long long)b * g.Da + a) * g.d + e) * g.N + n) * E;
        xr[be] = src[0];
        if constexpr (CPLX) xi[be] = src[1];
      }
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

// ---- elementwise MPO step of the folded one-site matvec (mpse_plans.h: K_WMIX):
//   dst_j[a, dd, k] = sum_terms sum_e W[b_t, dd, e, f_t] src_t[a, e, k]
// A thread owns one (a, k) and WM_CHUNK consecutive dd (lanes run along k: every load / store of a wave is 1 KB
// contiguous; the chunks of one (a, k) sit in neighbouring waves and share their loads in L2); it reads only the
// columns e its rows touch ([e_lo, e_hi) per term and chunk, from the host copy of the site: a tridiagonal block
// costs six loads per chunk, an identity block is a plain add).  Non-identity blocks sit in LDS as dense d x d matrices.
struct WmixTermDev {
  const double* src;
  long long s_a, s_d;
  int b, f, ident, slot;
  unsigned char e_lo[WM_MAXCHUNKS], e_hi[WM_MAXCHUNKS];
};
struct WmixDstDev {
  double* dst;
  long long s_a, s_d;
  int nterm, pad;
  WmixTermDev term[WM_MAXTERM];
};
struct WmixArgs {
  WmixDstDev dst[WM_MAXDST];
  const double* W;
  int ndst, Da, d, wr, Dk, nchunk, nslot;
  const int* skip;
};

template <bool CPLX>
__global__ __launch_bounds__(256) void k_wmix(const WmixArgs g) {
  if (g.skip && *g.skip) return;
  constexpr int E = CPLX ? 2 : 1;
  extern __shared__ double s_blk[];               // [slot][dd][e]
  const int d = g.d, dd2 = d * d;
  for (int j = 0; j < g.ndst; ++j)
    for (int t = 0; t < g.dst[j].nterm; ++t) {
      const WmixTermDev& tm = g.dst[j].term[t];
      if (tm.ident) continue;
      for (int r = threadIdx.x; r < dd2; r += 256) {
        const int x = r / d, e = r - x * d;
        s_blk[tm.slot * dd2 + r] = g.W[(((long long)tm.b * d + x) * d + e) * g.wr + tm.f];
      }
    }
  __syncthreads();
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)g.Da * g.nchunk * g.Dk) return;
  const int k = (int)(idx % g.Dk);
  const long long rest = idx / g.Dk;
  const int q = (int)(rest % g.nchunk), a = (int)(rest / g.nchunk);
  const int x0 = q * WM_CHUNK;
  for (int j = 0; j < g.ndst; ++j) {
    const WmixDstDev& D = g.dst[j];
    double ar[WM_CHUNK], ai[WM_CHUNK];
#pragma unroll
    for (int i = 0; i < WM_CHUNK; ++i) ar[i] = 0.0, ai[i] = 0.0;
    for (int t = 0; t < D.nterm; ++t) {
      const WmixTermDev& tm = D.term[t];
      const double* src = tm.src + ((long long)a * tm.s_a + k) * E;
      if (tm.ident) {
#pragma unroll
        for (int i = 0; i < WM_CHUNK; ++i)
          if (x0 + i < d) {
            const double* p = src + (long long)(x0 + i) * tm.s_d * E;
            ar[i] += p[0];
            if constexpr (CPLX) ai[i] += p[1];
          }
        continue;
      }
      const double* blk = s_blk + tm.slot * dd2;
      const int lo = tm.e_lo[q], hi = tm.e_hi[q];
      for (int e0 = lo; e0 < hi; e0 += 8) {
        double vr[8], vi[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {      // the loads of a batch are issued together
          vr[u] = 0.0, vi[u] = 0.0;
          if (e0 + u < hi) {
            const double* p = src + (long long)(e0 + u) * tm.s_d * E;
            vr[u] = p[0];
            if constexpr (CPLX) vi[u] = p[1];
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (e0 + u < hi) {
#pragma unroll
            for (int i = 0; i < WM_CHUNK; ++i)
              if (x0 + i < d) {
                const double w = blk[(x0 + i) * d + e0 + u];
                ar[i] += w * vr[u];
                if constexpr (CPLX) ai[i] += w * vi[u];
              }
          }
      }
    }
    double* dst = D.dst + ((long long)a * D.s_a + k) * E;
#pragma unroll
    for (int i = 0; i < WM_CHUNK; ++i)
      if (x0 + i < d) {
        double* p = dst + (long long)(x0 + i) * D.s_d * E;
        if constexpr (CPLX)
          *reinterpret_cast<double2*>(p) = make_double2(ar[i], ai[i]);
        else
          p[0] = ar[i];
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
  // The elementwise MPO step of a small site adds the K slices of the ONE product that forms its input channels itself
  // (no reduction launch between them): which step that is, and where its slices go
  auto small_wstep = [](const Step& s) {
    return s.kind == K_GEMM && s.is_wstep && s.dta == MPSE_F64 && s.a_off == 0 && s.b_off == 0 && s.c_off == 0 &&
           s.beta == 0.0 && s.w_wl * s.w_d <= 16 && s.w_d * s.w_wr <= 16;
  };
  const Step* slice_producer = nullptr;
  const Step* slice_consumer = nullptr;
  {
    int n_t1 = 0;
    const Step* prod = nullptr;
    const Step* cons = nullptr;
    for (const Step& s : p.steps) {
      if (small_wstep(s) && s.b == B_T1 && !cons) cons = &s;
      if (!cons && s.kind == K_GEMM && !s.is_wstep && s.c == B_T1) ++n_t1, prod = &s;
    }
    // (the product's rows are (channel | bra bond) over whole channels of the consumer's input, compactly stored)
    if (cons && n_t1 == 1 && prod->batch == 1 && prod->beta == 0.0 && prod->cin < 0 && cons->w_Da > 0 &&
        prod->nc.ext == cons->w_d * cons->w_N && prod->mc.ext % cons->w_Da == 0 &&
        prod->c_off % (cons->w_Da * cons->w_d * cons->w_N) == 0)
      slice_producer = prod, slice_consumer = cons;
  }
  TmpBuf SLC(ctx);
  int slices_used = 0;
  long long slice_elems = 0, slice_b_lo = 0, slice_b_hi = 0;
  for (const Step& s : p.steps) {
    if (small_wstep(s)) {
      if (!bufs[s.a] || !bufs[s.b] || !bufs[s.c]) return mpse_fail(ctx, MPSE_ERR_ARG, "plan: missing buffer");
      WsmArgs g;
      g.T1 = (const double*)bufs[s.b];
      g.T2 = (double*)const_cast<void*>(bufs[s.c]);
      g.W = (const double*)bufs[s.a];
      g.Da = (int)s.w_Da, g.d = (int)s.w_d, g.wl = (int)s.w_wl, g.wr = (int)s.w_wr, g.N = (int)s.w_N;
      g.skip = ctx->skip_flag;
      g.slices = slices_used > 0 ? SLC.as<const double>() : nullptr;
      g.nslices = slices_used, g.slice_stride = slice_elems, g.b_lo = (int)slice_b_lo, g.b_hi = (int)slice_b_hi;
      slices_used = 0;
      const long long total = s.w_Da * s.w_N;
      const dim3 grid((unsigned)((total + 255) / 256));
      if (s.dtb == MPSE_C128)
        hipLaunchKernelGGL((k_wsmall<true>), grid, dim3(256), 0, ctx->stream, g);
      else
        hipLaunchKernelGGL((k_wsmall<false>), grid, dim3(256), 0, ctx->stream, g);
      MPSE_HIP(ctx, hipGetLastError());
      continue;
    }
    if (s.kind == K_WMIX) {
      if (!bufs[s.b]) return mpse_fail(ctx, MPSE_ERR_ARG, "plan: missing buffer");
      WmixArgs g;
      memset(&g, 0, sizeof(g));
      g.W = (const double*)bufs[s.b];
      g.ndst = (int)s.mix.size();
      g.Da = (int)s.wp_Da, g.d = (int)s.wp_d, g.wr = (int)s.wp_wr, g.Dk = (int)s.wp_Dk;
      g.nchunk = (g.d + WM_CHUNK - 1) / WM_CHUNK;
      g.skip = ctx->skip_flag;
      int nslot = 0;
      for (int j = 0; j < g.ndst; ++j) {
        const WMixDst& q = s.mix[j];
        if (!bufs[q.dst]) return mpse_fail(ctx, MPSE_ERR_ARG, "plan: missing buffer");
        g.dst[j].dst = (double*)((char*)const_cast<void*>(bufs[q.dst]) + size_t(q.dst_off) * es);
        g.dst[j].s_a = q.s_a, g.dst[j].s_d = q.s_d, g.dst[j].nterm = q.nterm;
        for (int t = 0; t < q.nterm; ++t) {
          const WMixTerm& tm = q.term[t];
          if (!bufs[tm.src]) return mpse_fail(ctx, MPSE_ERR_ARG, "plan: missing buffer");
          WmixTermDev& o = g.dst[j].term[t];
          o.src = (const double*)((const char*)bufs[tm.src] + size_t(tm.src_off) * es);
          o.s_a = tm.s_a, o.s_d = tm.s_d, o.b = tm.b, o.f = tm.f, o.ident = tm.ident ? 1 : 0;
          o.slot = tm.ident ? 0 : nslot++;
          memcpy(o.e_lo, tm.e_lo, sizeof(o.e_lo));
          memcpy(o.e_hi, tm.e_hi, sizeof(o.e_hi));
        }
      }
      g.nslot = nslot;
      const size_t lds = size_t(nslot > 0 ? nslot : 1) * g.d * g.d * sizeof(double);
      if (lds > 64 * 1024) return mpse_fail(ctx, MPSE_ERR_SHAPE, "plan: MPO blocks of the elementwise step exceed the LDS");
      const long long total = (long long)s.wp_Da * g.nchunk * s.wp_Dk;
      const dim3 grid((unsigned)((total + 255) / 256));
      if (dtype == MPSE_C128)
        hipLaunchKernelGGL((k_wmix<true>), grid, dim3(256), lds, ctx->stream, g);
      else
        hipLaunchKernelGGL((k_wmix<false>), grid, dim3(256), lds, ctx->stream, g);
      MPSE_HIP(ctx, hipGetLastError());
      continue;
    }
    if (s.kind == K_GGEMM) {
      GroupedDesc gd;
      gd.dta = s.dta, gd.dtb = s.dtb;
      gd.ma = s.ma, gd.ka = s.ka, gd.kb = s.kb, gd.nb = s.nb, gd.mc = s.mc, gd.nc = s.nc;
      gd.ngrp = (int)s.groups.size();
      // occupancy of the operands: the environments by a scan that the solve keeps (one scan serves every channel),
      // the centre tensor by the structural mask of the solve - pushed through the MPO block for prepared operands
      TmpBuf MA(ctx), MB(ctx);
      const unsigned char *fa = nullptr, *fb = nullptr;
      bool stable = true;
      if (s.scan_a.on && bufs[s.scan_a.buf]) {
        bool st = false;
        MPSE_TRY(occ_mask_get(ctx, (const char*)bufs[s.scan_a.buf] + size_t(s.scan_a.off) * dtype_size(s.scan_a.dt),
                              s.scan_a.dt, s.scan_a.r, s.scan_a.k, MA, &fa, &gd.am_pitch, &st));
        stable = stable && st;
      }
      if (s.scan_b.on && bufs[s.scan_b.buf]) {
        bool st = false;
        MPSE_TRY(occ_mask_get(ctx, (const char*)bufs[s.scan_b.buf] + size_t(s.scan_b.off) * dtype_size(s.scan_b.dt),
                              s.scan_b.dt, s.scan_b.r, s.scan_b.k, MB, &fb, &gd.bm_pitch, &st));
        stable = stable && st;
      }
      // structural mask of the centre (B_C) as operand B: K = its left bond, columns = (e, k)
      const unsigned char* cm = nullptr;
      int cm_pitch = 0;
      {
        const char* pc = (const char*)bufs[B_C];
        const int64_t K = s.kb.ext, N = s.nb.ext;
        const int64_t nkw = ((K + 15) / 16 + 7) / 8, ntn = (N + 63) / 64;
        if (ctx->cmask.ptr && pc && pc >= ctx->cmask.lo && pc < ctx->cmask.hi && ctx->cmask.bytes == ntn * nkw * 8) {
          cm = static_cast<const unsigned char*>(ctx->cmask.ptr);
          cm_pitch = (int)(nkw * 8);
        }
      }
      for (int i = 0; i < gd.ngrp; ++i) {
        const GGroupPlan& g = s.groups[i];
        GroupedGrp& o = gd.grp[i];
        if (!bufs[g.cbuf]) return mpse_fail(ctx, MPSE_ERR_ARG, "plan: missing buffer");
        o.C = (char*)const_cast<void*>(bufs[g.cbuf]) + size_t(g.c_off) * es;
        o.nseg = g.nseg;
        o.beta = g.beta;
        for (int q = 0; q < g.nseg; ++q) {
          const GSegPlan& sg = g.seg[q];
          if (!bufs[sg.abuf] || !bufs[sg.bbuf]) return mpse_fail(ctx, MPSE_ERR_ARG, "plan: missing buffer");
          o.seg[q].A = (const char*)bufs[sg.abuf] + size_t(sg.a_off) * dtype_size(s.dta);
          o.seg[q].B = (const char*)bufs[sg.bbuf] + size_t(sg.b_off) * dtype_size(s.dtb);
          if (fa && sg.am_row0 >= 0) o.seg[q].am = fa + size_t(sg.am_row0) * gd.am_pitch;
          if (sg.bm_kind == BM_SCAN && fb) {
            o.seg[q].bm = fb + sg.bm_kt0;
          } else if (sg.bm_kind == BM_CENTRE && cm) {
            o.seg[q].bm = cm;
            gd.bm_pitch = cm_pitch;
          }
        }
      }
      gd.masks_stable = stable;
      if (gd.ngrp == 1 && s.groups[0].split2) {
        if (!bufs[B_OUT2]) return mpse_fail(ctx, MPSE_ERR_ARG, "plan: missing second result");
        gd.split2 = true;
        gd.c2 = const_cast<void*>(bufs[B_OUT2]);
      }
      // the last step completes the result: it carries the caller's dot request
      if (ctx->dot_req.y && &s == &p.steps.back() && gd.ngrp == 1 && s.groups[0].cbuf == B_OUT && s.groups[0].c_off == 0)
        ctx->dot_now = true;
      const int st = gemm_grouped(ctx, gd);
      ctx->dot_now = false;
      MPSE_TRY(st);
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
    if (ctx->dot_req.y && &s == &p.steps.back() && s.c == B_OUT && s.c_off == 0 && s.batch == 1)
      ctx->dot_now = true;
    if (s.cin >= 0) {
      if (!bufs[s.cin]) return mpse_fail(ctx, MPSE_ERR_ARG, "plan: missing beta source");
      ctx->cin_req.ptr = (const char*)bufs[s.cin] + size_t(s.cin_off) * dtype_size(dtc);
      ctx->cin_req.m = s.mcin;
      ctx->cin_req.n = s.ncin;
    }
    if (&s == slice_producer) {
      // room for four slices of this product's result (more: the product reduces them itself, as usual)
      const size_t blk = size_t(s.mc.ext) * size_t(s.nc.ext) * dtype_size(dtc);
      if (SLC.alloc(4 * blk) == MPSE_OK) {
        ctx->slices_req.ptr = SLC.p;
        ctx->slices_req.cap_bytes = 4 * blk;
        ctx->slices_req.used = 0;
      }
    }
    const int st = gemm_call(ctx, s.dta, s.dtb, s.conja, s.conjb, s.ma, s.ka, s.kb, s.nb, s.mc, s.nc, s.batch, s.sba,
                             s.sbb, s.sbc, a, b, c, 1.0, s.beta, s.skip_zero);
    if (&s == slice_producer) {
      slices_used = ctx->slices_req.used;
      ctx->slices_req = mpse_ctx::SlicesReq();
      if (slices_used > 0) {
        // rows of the product = (channel - b_lo | bra bond): which channels of T1 it forms
        slice_elems = (long long)s.mc.ext * s.nc.ext;
        slice_b_lo = s.c_off / (slice_consumer->w_Da * slice_consumer->w_d * slice_consumer->w_N);
        slice_b_hi = slice_b_lo + s.mc.ext / slice_consumer->w_Da;
      }
    }
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
  std::shared_ptr<void> wi_keep;     // (the map is also touched by mpse_free, possibly from another thread)
  if (h->nsite == 1 && h->W0) {
    std::lock_guard<std::mutex> lock(ctx->pool_mu);
    auto it = ctx->wsite_info.find(h->W0);
    if (it != ctx->wsite_info.end()) wi_keep = it->second.info;
  }
  // a caller that takes the result in two parts (mpse_ctx::y2_req, the Lanczos solve) lets the last product run as
  // halved tiles; `used` tells it whether the second part holds anything
  mpse_ctx::PartsReq& pr = ctx->parts_req;
  {
    bool taken = false;
    MPSE_TRY(heff_small_try(ctx, dtype, h, C, out, &taken));
    if (taken) return MPSE_OK;     // (pr.used is set there: the result may be a sum of parts)
    const WSiteInfo* wi0 = static_cast<const WSiteInfo*>(wi_keep.get());
    const bool wi_ok = wi0 && h->nsite == 1 && wi0->wl == h->dims.wl && wi0->d == h->dims.d0 && wi0->wr == h->dims.wr;
    MPSE_TRY(heff0_fused_try(ctx, dtype, h, C, wi_ok ? wi0->w.data() : nullptr, &taken));
    if (taken) return MPSE_OK;     // (tile-masked parts: pr.used, pr.mask)
  }
  const bool two_ok = pr.ptr != nullptr && pr.cap_elems >= 2 * pr.n;
  Plan p = plan_heff(dtype, *h, static_cast<const WSiteInfo*>(wi_keep.get()), two_ok);
  pr.used = 0;
  const void* bufs[B_COUNT] = {nullptr};
  bufs[B_L] = h->L;
  bufs[B_R] = h->R;
  bufs[B_W0] = h->W0;
  bufs[B_W1] = h->W1;
  bufs[B_C] = C;
  bufs[B_OUT] = out;
  // halved tiles: part 0 is `out` itself (it holds the beta term), part 1 the second slot of the caller's buffer;
  // split products: all parts in the caller's buffer (mpse_gemm.hip sets `used`)
  if (p.two_results) bufs[B_OUT2] = static_cast<char*>(pr.ptr) + size_t(pr.n) * dtype_size(dtype);
  const int st = run_plan(ctx, dtype, p, bufs);
  if (st == MPSE_OK && p.two_results) pr.used = -2;     // (out, part 1)
  return st;
}

extern "C" int mpse_mpo_site_hint(mpse_ctx* ctx, const void* W_dev, const double* W_host, int64_t wl, int64_t d,
                                  int64_t wr) {
  if (!ctx || !W_dev) return MPSE_ERR_ARG;
  if (!W_host) {
    std::lock_guard<std::mutex> lock(ctx->pool_mu);
    ctx->wsite_info.erase(W_dev);
    return MPSE_OK;
  }
  if (wl <= 0 || d <= 0 || wr <= 0) return mpse_fail(ctx, MPSE_ERR_SHAPE, "mpo_site_hint: empty MPO site");
  std::shared_ptr<void> info = std::make_shared<WSiteInfo>(analyse_mpo_site(W_host, wl, d, wr));
  std::lock_guard<std::mutex> lock(ctx->pool_mu);
  ctx->wsite_info[W_dev] = mpse_ctx::WSiteEntry{info, size_t(wl) * size_t(d) * size_t(d) * size_t(wr) * sizeof(double)};
  return MPSE_OK;
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
  std::shared_ptr<void> wi_keep;     // the site as the caller described it (mpse_mpo_site_hint): elementwise MPO step
  {
    static const bool fold_on = [] {
      const char* e = getenv("MPSE_ENV_WFOLD");
      return !(e && e[0] == '0');
    }();
    std::lock_guard<std::mutex> lock(ctx->pool_mu);
    auto it = ctx->wsite_info.find(W);
    if (fold_on && it != ctx->wsite_info.end()) wi_keep = it->second.info;
  }
  Plan p = plan_env(dtype, domain, *dims, env_dtype, w_dtype, bra_conj, static_cast<const WSiteInfo*>(wi_keep.get()));
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
