// The 0-site effective Hamiltonian (bond matvec of the TDVP back-step, mps/hop_expr.py:63-67:
//   out[a, l] = sum_{b, c, k} L[a, b, c] C[c, k] R[l, b, k])
// as ONE launch on FP64 MFMA for large complex bond matrices, with the result handed to the Lanczos update as a sum
// of tile-masked parts.
//
// Why: through the contraction plans this matvec is (L.C) as a split-K product + its reduction launch + (.R) as a
// split-K product + its reduction launch - four dependent launches, ~60 us at D = 256, w = 5, 850 times per evolve of
// the headline run (a quarter of the step).  The chain needs no exchange between workgroups
// when it is cut along (bra rows, MPO channel, ket chunk):
//   workgroup (at, b, kc): 16 bra rows a, one channel b, 64 ket columns k
//     step 1  T[a, k] = sum_c L[a, b, c] C[c, k]          16 x Dl x 64, one 16-column tile of T per wave
//     step 2  part_{(b, kc)}[a, l] = sum_k T[a, k] R[l, b, k]   16 x 64 x Dr, the l tiles dealt to the four waves
//   T goes from the accumulators to the A-operand layout through LDS; MFMA operands come straight from global memory
//   (L: 64-byte row segments; C and the transposed right environment Rt[b, k, l]: 256-byte rows).
// The w * ceil(Dr / 64) parts of an output tile are NOT reduced by a launch of their own: a 64-bit word per 16 x 16
// output tile says which parts hold it (the others were never written), and k_lanczos_update_u adds exactly those while
// it reads - in part order, so the result is bitwise reproducible.
// Block sparsity (quantum numbers; the identity channels of canonical environments) is exploited at 16 x 16 granularity:
// byte flags of L and R tiles are computed once per solve (the environments are constant), the flags of C are the
// caller's structural centre mask (mpse_expm_centre_mask; without one every tile of C counts as occupied).  A workgroup
// whose T is structurally zero returns at once; step 1 visits the c tiles where L AND C hold data, step 2 the
// (l tile, k tile) pairs where R does.
#include "mpse_device.h"
#include "mpse_internal.h"

typedef double v4d __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ v4d mfma(double a, double b, v4d c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

struct F0Args {
  const double* L;        // (Dl, w, Dl)
  const double* Rt;       // (w, Dr, Dr): Rt[b, k, l] = R[l, b, k]
  const double* C;        // (Dl, Dr)
  double* parts;          // part s at parts + s * n (complex elements), laid out like out (Dl, Dr)
  const double* y;        // optional: dot partner laid out like out
  double* dot_part;       // one (re, im) per workgroup
  const unsigned char* FL;   // [(at * w + b) * ntl + ct]
  const unsigned char* FC;   // [kc * fc_pitch + ct] (centre mask), or null
  const unsigned char* FR;   // [(lt * w + b) * ntr + kt]
  const unsigned long long* mask;   // [at * ntr + lt]: bit s = part s holds this tile
  const int* skip;
  long long n;
  int Dl, Dr, w, ntl, ntr, nkc, fc_pitch;
};

// Rt[(b, k), l] = R[l, b, k]
__global__ __launch_bounds__(256) void k_f0_transpose(double2* __restrict__ rt, const double2* __restrict__ r, int D, int w,
                                                       const int* __restrict__ skip) {
  if (skip && *skip) return;
  const long long n = (long long)D * w * D;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const int l = (int)(i % D);
    const long long bk = i / D;
    rt[i] = r[(long long)l * w * D + bk];
  }
}

// flags of the 16 x 16 tiles of an environment E (D, w, D) viewed per channel: F[(rt * w + b) * nt + ct] = any non-zero in
// rows [16 rt, 16 rt + 16), channel b, columns [16 ct, 16 ct + 16).  grid (nt, w), 256 threads = one tile row.
__global__ __launch_bounds__(256) void k_f0_flags(const double2* __restrict__ E, int D, int w, int nt,
                                                   unsigned char* __restrict__ F, const int* __restrict__ skip) {
  if (skip && *skip) return;
  __shared__ int s_f[64];
  const int rt = blockIdx.x, b = blockIdx.y, row = threadIdx.x >> 4, col = threadIdx.x & 15;
  if (threadIdx.x < 64) s_f[threadIdx.x] = 0;
  __syncthreads();
  const double2* base = E + ((long long)(16 * rt + row) * w + b) * D;
  for (int ct = 0; ct < nt; ++ct) {           // (no barrier inside: the loads of all tiles are in flight together)
    const double2 v = base[16 * ct + col];
    if (v.x != 0.0 || v.y != 0.0) s_f[ct] = 1;
  }
  __syncthreads();
  if ((int)threadIdx.x < nt) F[((long long)rt * w + b) * nt + threadIdx.x] = s_f[threadIdx.x] ? 1 : 0;
}

// which parts hold which output tile: bit s = b * nkc + kc of mask[at * ntr + lt] is set when step 1 of workgroup
// (at, b, kc) has a c tile with data on both sides AND R has data in rows lt, channel b, k chunk kc.  One workgroup.
__global__ __launch_bounds__(1024) void k_f0_valid(const unsigned char* __restrict__ FL, const unsigned char* __restrict__ FC,
                                                    const unsigned char* __restrict__ FR, int w, int ntl, int ntr, int nkc,
                                                    int fc_pitch, unsigned long long* __restrict__ mask,
                                                    const int* __restrict__ skip) {
  if (skip && *skip) return;
  __shared__ unsigned long long s1[64], s2[64];    // per bra tile row / per l tile: bit s
  const int nparts = w * nkc, tid = threadIdx.x;
  if (tid < 64) s1[tid] = s2[tid] = 0;
  __syncthreads();
  for (int t = tid; t < ntl * nparts; t += 1024) {
    const int at = t / nparts, s = t - at * nparts, b = s / nkc, kc = s - b * nkc;
    bool any = false;
    for (int ct = 0; ct < ntl; ++ct) any = any || (FL[((long long)at * w + b) * ntl + ct] && (!FC || FC[kc * fc_pitch + ct]));
    if (any) atomicOr(&s1[at], 1ull << s);
  }
  for (int t = tid; t < ntr * nparts; t += 1024) {
    const int lt = t / nparts, s = t - lt * nparts, b = s / nkc, kc = s - b * nkc;
    bool any = false;
    for (int j = 0; j < 4; ++j) {
      const int kt = 4 * kc + j;
      if (kt < ntr) any = any || FR[((long long)lt * w + b) * ntr + kt];
    }
    if (any) atomicOr(&s2[lt], 1ull << s);
  }
  __syncthreads();
  for (int t = tid; t < ntl * ntr; t += 1024) {
    const int at = t / ntr, lt = t - at * ntr;
    mask[t] = s1[at] & s2[lt];
  }
}

__global__ __launch_bounds__(256) void k_heff0_fused(const F0Args g) {
  constexpr int NW = 4;   // waves per workgroup (eight - two groups splitting the c tiles of step 1, added up in LDS -
                          // spilled registers and lost: 495 against 510 site-updates/s)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), x = lane & 15,
            kq = lane >> 4;
  __shared__ double sTr[16 * 65], sTi[16 * 65];
  __shared__ double s_dot[8];
  if (g.skip && *g.skip) return;
  // Launch position -> (bra tile row, part), plain order.  (Measured and dropped: a die - launch position mod 8, one L2
  // each - taking whole parts, so that its L2 holds only the panels of C and Rt its workgroups share.  Dealt channel-major
  // the heavy parts - ket chunks that straddle two quantum-number sectors - piled up on two dies: 39 us against 33; dealt
  // chunk-major 32 against 29.)
  const int nparts = g.w * g.nkc;
  const int wg = blockIdx.x;
  const int at = wg / nparts, s = wg - at * nparts, b = s / g.nkc, kc = s - b * g.nkc;
  // which output tiles this workgroup holds (the mask is the single statement of that rule): lane lt looks at tile lt
  const unsigned long long lts =
      __ballot(lane < g.ntr && ((g.mask[(long long)at * g.ntr + min(lane, g.ntr - 1)] >> s) & 1ull));
  if (lts == 0) {
    if (g.dot_part && tid == 0) {
      g.dot_part[2 * wg] = 0.0;
      g.dot_part[2 * wg + 1] = 0.0;
    }
    return;
  }
  // ---- step 1: T[a, k] for this wave's 16 columns k of the chunk
  const int wcolt = wave;
  const int a0 = 16 * at, k0 = 64 * kc + 16 * wcolt;
  const bool wcol = k0 < g.Dr;                       // (last chunk of a Dr that is not a multiple of 64)
  // (every flag this workgroup will consult is requested here, together: a dependent trip to memory costs ~1.5 us, and
  // a workgroup has no neighbour on its compute unit to hide it behind)
  unsigned frn = 0;          // lane lt: bit j = R has data in rows lt, channel b, k tile 4 kc + j
  if (lane < g.ntr) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int kt = 4 * kc + jj;
      if (kt < g.ntr && g.FR[((long long)lane * g.w + b) * g.ntr + kt]) frn |= 1u << jj;
    }
  }
  const int lc = min(lane, g.ntl - 1);
  const unsigned long long cts = __ballot(lane < g.ntl && g.FL[((long long)at * g.w + b) * g.ntl + lc] &&
                                          (!g.FC || g.FC[kc * g.fc_pitch + lc]));
  const double2* La = reinterpret_cast<const double2*>(g.L) + ((long long)(a0 + x) * g.w + b) * g.Dl + kq;   // + c
  const double2* Cb = reinterpret_cast<const double2*>(g.C) + (long long)kq * g.Dr + (wcol ? k0 : 0) + x;     // + c * Dr
  v4d tr = {0, 0, 0, 0}, ti = {0, 0, 0, 0};
  // operands of three c tiles in flight (the tensors come from the memory-side cache: ~1.5 us away, and a workgroup has
  // no neighbour on its compute unit to hide that behind)
  double2 av[3][4], bv[3][4];
  auto load1 = [&](int slot, int ct) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int c = 16 * ct + 4 * kk;
      av[slot][kk] = La[c];
      bv[slot][kk] = Cb[(long long)c * g.Dr];
    }
  };
  auto mul1 = [&](int slot) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      tr = mfma(av[slot][kk].x, bv[slot][kk].x, tr);
      tr = mfma(-av[slot][kk].y, bv[slot][kk].y, tr);
      ti = mfma(av[slot][kk].x, bv[slot][kk].y, ti);
      ti = mfma(av[slot][kk].y, bv[slot][kk].x, ti);
    }
  };
  const unsigned long long mine = cts;
  if (wcol && mine) {
    // Three c tiles in flight, slots used in a fixed rotation.  The compiler barriers keep the order "request the tile
    // after next, THEN multiply the oldest": without them the scheduler sinks every batch of loads to just before its
    // use (fewer live registers) and each tile waits for its own trip to memory - measured: 3 600 cycles per tile against
    // the 1 024 of its sixteen MFMAs.
    unsigned long long m = mine;       // tiles still to load
    auto next_ct = [&]() {
      const int ct = m ? (int)__builtin_ctzll(m) : -1;
      m &= m - (m ? 1 : 0);
      return ct;
    };
    int c0 = next_ct(), c1 = next_ct(), c2 = next_ct();
    if (c0 >= 0) load1(0, c0);
    if (c1 >= 0) load1(1, c1);
    if (c2 >= 0) load1(2, c2);
    while (c0 >= 0) {
      asm volatile("" ::: "memory");
      mul1(0);
      c0 = next_ct();
      if (c0 >= 0) load1(0, c0);
      asm volatile("" ::: "memory");
      if (c1 < 0) break;
      mul1(1);
      c1 = next_ct();
      if (c1 >= 0) load1(1, c1);
      asm volatile("" ::: "memory");
      if (c2 < 0) break;
      mul1(2);
      c2 = next_ct();
      if (c2 >= 0) load1(2, c2);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    sTr[(kq + 4 * r) * 65 + 16 * wcolt + x] = tr[r];
    sTi[(kq + 4 * r) * 65 + 16 * wcolt + x] = ti[r];
  }
  __syncthreads();
  // ---- step 2: the l tiles this workgroup holds, dealt to the waves in order; the A operand (T) of the whole chunk
  // stays in registers
  double xr[16], xi[16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    xr[ks] = sTr[x * 65 + 4 * ks + kq];
    xi[ks] = sTi[x * 65 + 4 * ks + kq];
  }
  double dre = 0.0, dim = 0.0;
  const double2* Rb = reinterpret_cast<const double2*>(g.Rt) + ((long long)b * g.Dr + 64 * kc + kq) * g.Dr + x;
  double2* part = reinterpret_cast<double2*>(g.parts) + (long long)s * g.n;
  // this wave's tiles: the (wave)-th, (wave + 8)-th, .. set bit of lts; two of them in flight
  auto nth_tile = [&](unsigned long long m, int n) {      // position of the n-th set bit, or -1
    for (int i = 0; i < n && m; ++i) m &= m - 1;
    return m ? (int)__builtin_ctzll(m) : -1;
  };
  auto k_tiles = [&](int lt) {                             // k tiles of the chunk where R has data for l tile lt
    return (unsigned)__builtin_amdgcn_readlane((int)frn, lt);
  };
  double2 rv[2][16], yv2[2][4];
  auto load2 = [&](int slot, int lt, unsigned) {
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      // (unconditional: a load guarded by the tile flag becomes load + select, and the select waits for the load on the
      // spot; k tiles without data are loaded - rows clamped into the tensor - and never multiplied)
      const long long krow = min((long long)(4 * ks), (long long)(g.Dr - 1 - 64 * kc - kq));
      rv[slot][ks] = Rb[krow * g.Dr + 16 * lt];
    }
    if (g.y) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        yv2[slot][r] = reinterpret_cast<const double2*>(g.y)[(long long)(a0 + kq + 4 * r) * g.Dr + 16 * lt + x];
    }
  };
  auto mul2 = [&](int slot, int lt, unsigned kts) {
    v4d orr = {0, 0, 0, 0}, oi = {0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      if ((kts >> (ks >> 2)) & 1u) {      // (uniform)
        orr = mfma(xr[ks], rv[slot][ks].x, orr);
        orr = mfma(-xi[ks], rv[slot][ks].y, orr);
        oi = mfma(xr[ks], rv[slot][ks].y, oi);
        oi = mfma(xi[ks], rv[slot][ks].x, oi);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long long e = (long long)(a0 + kq + 4 * r) * g.Dr + 16 * lt + x;
      part[e] = make_double2(orr[r], oi[r]);
      if (g.y) {
        const double2 yv = yv2[slot][r];
        dre += orr[r] * yv.x + oi[r] * yv.y;      // conj(o) y
        dim += orr[r] * yv.y - oi[r] * yv.x;
      }
    }
  };
  int idx = wave;
  int lt0 = nth_tile(lts, idx), lt1 = -1;
  unsigned kt0 = 0, kt1 = 0;
  if (lt0 >= 0) {
    kt0 = k_tiles(lt0);
    load2(0, lt0, kt0);
  }
  while (lt0 >= 0) {
    idx += NW;
    lt1 = nth_tile(lts, idx);
    if (lt1 >= 0) {
      kt1 = k_tiles(lt1);
      load2(1, lt1, kt1);
    }
    asm volatile("" ::: "memory");     // (the next tile's operands are requested before this tile is multiplied)
    mul2(0, lt0, kt0);
    if (lt1 < 0) break;
    idx += NW;
    lt0 = nth_tile(lts, idx);
    if (lt0 >= 0) {
      kt0 = k_tiles(lt0);
      load2(0, lt0, kt0);
    }
    asm volatile("" ::: "memory");
    mul2(1, lt1, kt1);
  }
  if (g.dot_part) {
    dre = wave_sum(dre);
    dim = wave_sum(dim);
    if (lane == 0) {
      s_dot[2 * wave] = dre;
      s_dot[2 * wave + 1] = dim;
    }
    __syncthreads();
    if (tid == 0) {
      double ar = 0.0, ai = 0.0;
#pragma unroll
      for (int w8 = 0; w8 < NW; ++w8) {
        ar += s_dot[2 * w8];
        ai += s_dot[2 * w8 + 1];
      }
      g.dot_part[2 * wg] = ar;
      g.dot_part[2 * wg + 1] = ai;
    }
  }
}

}  // namespace

// Number of parts the fused 0-site matvec would deliver for this operator (0: not eligible).  Eligible: complex bond
// matrix between complex environments with equal bra / ket bonds, both bond dimensions multiples of 16 and large enough
// that the plans' four launches are not already cheap, at most 64 parts.  MPSE_HEFF0=0 switches the path off.
int heff0_fused_parts(const mpse_heff* h, int dtype) {
  static const int mode = [] {
    const char* e = getenv("MPSE_HEFF0");
    return e ? atoi(e) : 1;
  }();
  if (mode == 0 || h->nsite != 0 || dtype != MPSE_C128 || h->l_dtype != MPSE_C128 || h->r_dtype != MPSE_C128) return 0;
  const mpse_dims& d = h->dims;
  const int64_t Dl = d.Dl_ket, Dr = d.Dr_ket, w = d.wl;
  if ((d.Dl_bra > 0 && d.Dl_bra != Dl) || (d.Dr_bra > 0 && d.Dr_bra != Dr) || d.wr != w || d.danc > 1) return 0;
  if (Dl % 16 || Dr % 16 || Dl > 1024 || Dr > 1024 || w < 1) return 0;     // (tile rows / columns index 64-bit words)
  const int64_t min_d = mode >= 2 ? 16 : 128;            // (MPSE_HEFF0=2: every eligible shape - tests)
  if (Dl < min_d || Dr < min_d) return 0;
  const int64_t nparts = w * ((Dr + 63) / 64);
  if (nparts > 64) return 0;
  return (int)nparts;
}

void heff0_drop_cache(mpse_ctx* ctx) {
  if (ctx->f0.buf) mpse_free(ctx, ctx->f0.buf);
  ctx->f0 = mpse_ctx::F0Cache();
}

// Runs the fused matvec when the caller offered masked parts (mpse_ctx::parts_req.masked_ok) with room for all of them.
int heff0_fused_try(mpse_ctx* ctx, int dtype, const mpse_heff* h, const void* C, bool* taken) {
  *taken = false;
  mpse_ctx::PartsReq& pr = ctx->parts_req;
  const int nparts = heff0_fused_parts(h, dtype);
  if (nparts == 0 || !pr.ptr || !pr.masked_ok) return MPSE_OK;
  const int Dl = (int)h->dims.Dl_ket, Dr = (int)h->dims.Dr_ket, w = (int)h->dims.wl;
  const long long n = (long long)Dl * Dr;
  if (pr.n != n || pr.cap_elems < (long long)nparts * n) return MPSE_OK;
  const int ntl = Dl / 16, ntr = Dr / 16, nkc = (Dr + 63) / 64;
  const int nwg = ntl * nparts;
  if (ctx->dot_req.y && nwg > ctx->dot_req.cap) return MPSE_OK;
  // the centre mask applies when it describes this shape: rows of C in 16s, columns in 64s
  const unsigned char* FC = nullptr;
  int fc_pitch = 0;
  {
    const long long nkw = (ntl + 7) / 8;
    const char* pc = static_cast<const char*>(C);
    if (ctx->cmask.ptr && pc >= ctx->cmask.lo && pc < ctx->cmask.hi && ctx->cmask.bytes == (long long)nkc * nkw * 8) {
      FC = static_cast<const unsigned char*>(ctx->cmask.ptr);
      fc_pitch = (int)(nkw * 8);
    }
  }
  // per-solve data: transposed right environment, tile flags of L and R, the part mask
  const size_t rt_bytes = size_t(w) * Dr * Dr * 16;
  const size_t fl_bytes = (size_t(ntl) * w * ntl + 15) & ~size_t(15), fr_bytes = (size_t(ntr) * w * ntr + 15) & ~size_t(15);
  const size_t mk_bytes = size_t(ntl) * ntr * 8;
  const bool keep = ctx->occ_cache_on || ctx->small_rt_scope;
  if (!keep) return MPSE_OK;     // (outside a solve nothing would own the flags and the mask until the consumer has run)
  mpse_ctx::F0Cache& fc = ctx->f0;
  char* base = nullptr;
  const bool hit = keep && fc.buf && fc.L == h->L && fc.R == h->R && fc.cmask == (const void*)FC && fc.Dl == Dl && fc.Dr == Dr &&
                   fc.w == w;
  if (hit) {
    base = static_cast<char*>(fc.buf);
  } else {
    void* p = nullptr;
    heff0_drop_cache(ctx);
    MPSE_TRY(mpse_malloc(ctx, rt_bytes + fl_bytes + fr_bytes + mk_bytes, &p));
    fc.buf = p, fc.L = h->L, fc.R = h->R, fc.cmask = FC, fc.Dl = Dl, fc.Dr = Dr, fc.w = w;
    base = static_cast<char*>(p);
    const long long nel = (long long)w * Dr * Dr;
    int nb = (int)((nel + 255) / 256);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_f0_transpose, dim3(nb), dim3(256), 0, ctx->stream, reinterpret_cast<double2*>(base),
                       static_cast<const double2*>(h->R), Dr, w, ctx->skip_flag);
    hipLaunchKernelGGL(k_f0_flags, dim3(ntl, w), dim3(256), 0, ctx->stream, static_cast<const double2*>(h->L), Dl, w, ntl,
                       reinterpret_cast<unsigned char*>(base + rt_bytes), ctx->skip_flag);
    hipLaunchKernelGGL(k_f0_flags, dim3(ntr, w), dim3(256), 0, ctx->stream, static_cast<const double2*>(h->R), Dr, w, ntr,
                       reinterpret_cast<unsigned char*>(base + rt_bytes + fl_bytes), ctx->skip_flag);
    hipLaunchKernelGGL(k_f0_valid, dim3(1), dim3(1024), 0, ctx->stream,
                       reinterpret_cast<const unsigned char*>(base + rt_bytes), FC,
                       reinterpret_cast<const unsigned char*>(base + rt_bytes + fl_bytes), w, ntl, ntr, nkc, fc_pitch,
                       reinterpret_cast<unsigned long long*>(base + rt_bytes + fl_bytes + fr_bytes), ctx->skip_flag);
  }
  F0Args g{};
  g.L = static_cast<const double*>(h->L);
  g.Rt = reinterpret_cast<const double*>(base);
  g.C = static_cast<const double*>(C);
  g.parts = static_cast<double*>(pr.ptr);
  g.FL = reinterpret_cast<const unsigned char*>(base + rt_bytes);
  g.FC = FC;
  g.FR = reinterpret_cast<const unsigned char*>(base + rt_bytes + fl_bytes);
  g.mask = reinterpret_cast<const unsigned long long*>(base + rt_bytes + fl_bytes + fr_bytes);
  g.skip = ctx->skip_flag;
  g.n = n;
  g.Dl = Dl, g.Dr = Dr, g.w = w, g.ntl = ntl, g.ntr = ntr, g.nkc = nkc, g.fc_pitch = fc_pitch;
  if (ctx->dot_req.y) {
    g.y = static_cast<const double*>(ctx->dot_req.y);
    g.dot_part = ctx->dot_req.part;
    ctx->dot_req.nb_out = nwg;
  }
  hipLaunchKernelGGL(k_heff0_fused, dim3(nwg), dim3(256), 0, ctx->stream, g);
  MPSE_HIP(ctx, hipGetLastError());
  pr.used = nparts;
  pr.mask = g.mask;
  pr.mask_row = Dr;
  pr.mask_tiles = ntr;
  *taken = true;
  return MPSE_OK;
}
