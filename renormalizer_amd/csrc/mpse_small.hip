// Effective-Hamiltonian matvec of SMALL centres (0- and 1-site, mps/hop_expr.py:63-79) as ONE launch.
//
// At bond dimensions of a few tens the three-step chain of mpse_plans.h (two strided GEMMs around the MPO step, plus
// split-K reductions) is 3-6 launches of a few microseconds each: the Krylov solve of such a site is bound by launches
// and kernel boundaries, not by arithmetic (profiles/r04_traj_scaling.jsonl).  The chain
//   out[a,x,l] = sum_{b,c,e,f,k} L[a,b,c] W[b,x,e,f] C[c,e,k] R[l,f,k]
// needs no exchange between workgroups when it is cut along the bra bond a of L: the workgroup of row a forms
//   T1[b,(e,k)] = sum_c L[a,b,c] C[c,(e,k)]          (wl x d Dr, in LDS)
//   P[x,(f,k)]  = sum_{b,e} W[b,x,e,f] T1[b,e,k]     (d x wr Dr, in LDS; W as a sparse list built in LDS)
//   out[a,x,l]  = sum_{(f,k)} P[x,(f,k)] Rt[(f,k),l]
// with plain FP64 vector FMAs (the FP64 vector and matrix peaks of this part are equal; the products here are a few
// rows tall, far below an MFMA tile).  Rt[(f,k),l] = R[l,f,k] is a transposed copy of the right environment, made
// once per Krylov solve (the environments are constant over a solve), so that the last step reads coalesced rows.
// The partial sums of <result, y> that the Lanczos update needs ride on the same launch (mpse_ctx::dot_req), one
// (re, im) pair per workgroup in a fixed order: results are bitwise reproducible run to run.
//
// When the caller takes the result as a sum of parts (mpse_ctx::parts_req: the Lanczos update adds them while it reads),
// the ket bond k of R is cut into up to four slices over blockIdx.y: a workgroup then works on the columns (e, k) of C and
// the rows (f, k) of Rt of its slice only and writes its own partial result - four times the compute units, a quarter of
// the bytes per workgroup (27 -> 11.5 us per matvec at D = 64 together with the prefetch pipeline).
//
// The 0-site matvec (abc,lbk,ck->al) is the same kernel with d = 1 and no MPO step (P = T1, wl == wr).
#include <algorithm>
#include <cstdlib>

#include "mpse_device.h"
#include "mpse_internal.h"

namespace {

constexpr int SM_THREADS = 256;
constexpr int SM_WMAX = 8;     // MPO bond channels
constexpr int SM_DMAX = 16;    // physical dimension

struct SmallArgs {
  const double* L;     // (Dl, wl, Dl)
  const double* Rt;    // (wr, Dr, Dr): [f][k][l]
  const double* W;     // (wl, d, d, wr) real, null for the 0-site matvec
  const double* C;     // (Dl, d, Dr)
  double* out;         // (Dl, d, Dr)
  const double* y;     // dot request: vector laid out like out (null: none)
  double* part;        // (re, im) per workgroup
  const int* skip;     // non-zero word: the launch is a no-op (asynchronous Lanczos past its convergence)
  int Dl, Dr, d, wl, wr;
  int kh;              // ket bond states of R per workgroup: blockIdx.y = slice h of the bond, k in [h kh, (h + 1) kh);
                       // slice h writes its own partial result at out + h * part_stride (the caller adds the parts)
  long long part_stride;   // elements of the working type
  int kg;              // K groups of the first step (threads = kg x min(d kh, 256 / kg ..))
  int off_L, off_T1, off_X, off_red;   // LDS offsets in elements of the working type
  int csr_pitch;       // entries per (x, f) row of the sparse W list
  int cnt_dbl, idx_dbl, csr_dbl;   // LDS doubles taken by the counts, the indices, the whole sparse list (even)
};

typedef const __attribute__((address_space(4))) double* ConstD;

template <bool CPLX>
struct Elem;
template <>
struct Elem<true> {
  using T = double2;
  static __device__ __forceinline__ T zero() { return make_double2(0.0, 0.0); }
  static __device__ __forceinline__ T ld_const(ConstD p, int i) { return make_double2(p[2 * i], p[2 * i + 1]); }
  static __device__ __forceinline__ void mad(T& acc, const T a, const T b) {
    acc.x = fma(a.x, b.x, acc.x);
    acc.x = fma(-a.y, b.y, acc.x);
    acc.y = fma(a.x, b.y, acc.y);
    acc.y = fma(a.y, b.x, acc.y);
  }
  static __device__ __forceinline__ void mad_real(T& acc, const double w, const T b) {
    acc.x = fma(w, b.x, acc.x);
    acc.y = fma(w, b.y, acc.y);
  }
  static __device__ __forceinline__ void add(T& acc, const T b) { acc.x += b.x, acc.y += b.y; }
  static __device__ __forceinline__ void dot(double& re, double& im, const T c, const T y) {   // conj(c) * y
    re += c.x * y.x + c.y * y.y;
    im += c.x * y.y - c.y * y.x;
  }
};
template <>
struct Elem<false> {
  using T = double;
  static __device__ __forceinline__ T zero() { return 0.0; }
  static __device__ __forceinline__ T ld_const(ConstD p, int i) { return p[i]; }
  static __device__ __forceinline__ void mad(T& acc, const T a, const T b) { acc = fma(a, b, acc); }
  static __device__ __forceinline__ void mad_real(T& acc, const double w, const T b) { acc = fma(w, b, acc); }
  static __device__ __forceinline__ void add(T& acc, const T b) { acc += b; }
  static __device__ __forceinline__ void dot(double& re, double& im, const T c, const T y) { re += c * y; }
};

// Rt[(f, k), l] = R[l, f, k]
template <bool CPLX>
__global__ __launch_bounds__(SM_THREADS) void k_env_transpose(double* __restrict__ rt, const double* __restrict__ r, int D,
                                                               int w, const int* __restrict__ skip) {
  using E = Elem<CPLX>;
  using T = typename E::T;
  if (skip && *skip) return;
  const long long n = (long long)D * w * D;
  const long long stride = (long long)gridDim.x * SM_THREADS;
  for (long long i = (long long)blockIdx.x * SM_THREADS + threadIdx.x; i < n; i += stride) {
    const int l = (int)(i % D);
    const long long fk = i / D;       // f * D + k
    reinterpret_cast<T*>(rt)[i] = reinterpret_cast<const T*>(r)[(long long)l * w * D + fk];
  }
}

constexpr int SM_U = 8;   // loads in flight per column and lane

// dst[b * bstride + lcols[p]] = sum over c = c0, c0 + cstep, .. < Dl of L[b, c] C[c, cols[p]], b < WL, for the columns
// p < ncol of this thread; the loads of C run one batch of rows ahead of the arithmetic.  The row of L sits in LDS: every
// value is a broadcast read that two columns share (with one column per thread the LDS pipe, not the FP64 pipe, bounds
// the step; scalar loads of L were tried - forty live values per batch spill the scalar registers).
template <bool CPLX, int NC, int WL>
__device__ __forceinline__ void small_t1(const typename Elem<CPLX>::T* __restrict__ Cm, const typename Elem<CPLX>::T* sL, int Dl, int N1,
                                         int c0, int cstep, const int (&cols)[NC], const int (&lcols)[NC], int ncol,
                                         typename Elem<CPLX>::T* dst, int bstride) {
  using E = Elem<CPLX>;
  using T = typename E::T;
  constexpr int U = SM_U / NC;   // the same number of loads in flight per lane whatever the column count
  T acc[NC][WL];
#pragma unroll
  for (int p = 0; p < NC; ++p)
#pragma unroll
    for (int b = 0; b < WL; ++b) acc[p][b] = E::zero();
  const int nc = Dl > c0 ? (Dl - c0 + cstep - 1) / cstep : 0;
  const int nbatch = nc / U;
  T cur[NC][U], nxt[NC][U];
  int c = c0;
  if (nbatch > 0) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int p = 0; p < NC; ++p) cur[p][u] = Cm[(long long)(c + u * cstep) * N1 + cols[p]];
  }
  for (int bi = 0; bi < nbatch; ++bi) {
    const bool more = bi + 1 < nbatch;
    if (more) {
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int p = 0; p < NC; ++p) nxt[p][u] = Cm[(long long)(c + (U + u) * cstep) * N1 + cols[p]];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int b = 0; b < WL; ++b) {
        const T lv = sL[b * Dl + c + u * cstep];
#pragma unroll
        for (int p = 0; p < NC; ++p) E::mad(acc[p][b], lv, cur[p][u]);
      }
    }
    if (more) {
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int p = 0; p < NC; ++p) cur[p][u] = nxt[p][u];
    }
    c += U * cstep;
  }
  for (; c < Dl; c += cstep) {
#pragma unroll
    for (int b = 0; b < WL; ++b) {
      const T lv = sL[b * Dl + c];
#pragma unroll
      for (int p = 0; p < NC; ++p) E::mad(acc[p][b], lv, Cm[(long long)c * N1 + cols[p]]);
    }
  }
#pragma unroll
  for (int p = 0; p < NC; ++p)
    if (p < ncol) {
#pragma unroll
      for (int b = 0; b < WL; ++b) dst[b * bstride + lcols[p]] = acc[p][b];
    }
}

template <bool CPLX, int NC>
__device__ __forceinline__ void small_t1_wl(int wl, const typename Elem<CPLX>::T* __restrict__ Cm, const typename Elem<CPLX>::T* Lrow, int Dl,
                                            int N1, int c0, int cstep, const int (&cols)[NC], const int (&lcols)[NC], int ncol,
                                            typename Elem<CPLX>::T* dst, int bstride) {
  switch (wl) {   // (uniform)
    case 1: small_t1<CPLX, NC, 1>(Cm, Lrow, Dl, N1, c0, cstep, cols, lcols, ncol, dst, bstride); break;
    case 2: small_t1<CPLX, NC, 2>(Cm, Lrow, Dl, N1, c0, cstep, cols, lcols, ncol, dst, bstride); break;
    case 3: small_t1<CPLX, NC, 3>(Cm, Lrow, Dl, N1, c0, cstep, cols, lcols, ncol, dst, bstride); break;
    case 4: small_t1<CPLX, NC, 4>(Cm, Lrow, Dl, N1, c0, cstep, cols, lcols, ncol, dst, bstride); break;
    case 5: small_t1<CPLX, NC, 5>(Cm, Lrow, Dl, N1, c0, cstep, cols, lcols, ncol, dst, bstride); break;
    case 6: small_t1<CPLX, NC, 6>(Cm, Lrow, Dl, N1, c0, cstep, cols, lcols, ncol, dst, bstride); break;
    case 7: small_t1<CPLX, NC, 7>(Cm, Lrow, Dl, N1, c0, cstep, cols, lcols, ncol, dst, bstride); break;
    default: small_t1<CPLX, NC, 8>(Cm, Lrow, Dl, N1, c0, cstep, cols, lcols, ncol, dst, bstride); break;
  }
}

// acc[x] = sum over q = q0, q0 + G, .. < K3 of P[x, q] Rt[q, l], x < DX (P rows beyond the physical dimension are padding
// of the LDS area: their sums are never stored); loads of Rt SM_U deep, one batch ahead of the arithmetic
template <bool CPLX, int DX>
__device__ __forceinline__ void small_t3(const typename Elem<CPLX>::T* __restrict__ Rt, const typename Elem<CPLX>::T* sP,
                                         int K3, int Dr, int kh, int k0, int G, int grp, int l,
                                         typename Elem<CPLX>::T* sRed, int d) {
  using E = Elem<CPLX>;
  using T = typename E::T;
  constexpr int U = DX <= 4 ? SM_U : (DX == 8 ? 4 : 2);   // DX x U broadcast reads of P per batch stay in registers
  T acc[DX];
#pragma unroll
  for (int x = 0; x < DX; ++x) acc[x] = E::zero();
  const bool on = grp < G;
  if (on) {
    // q = (f, kk) of this slice -> row f * Dr + k0 + kk of Rt
    auto rt_at = [&](int q) -> T {
      const int f = q / kh, kk = q - f * kh;
      return Rt[(long long)(f * Dr + k0 + kk) * Dr + l];
    };
    const int ni = (K3 - grp + G - 1) / G;      // my q values
    const int nbatch = ni / U;
    T cur[U], nxt[U];
    int q = grp;
    if (nbatch > 0) {
#pragma unroll
      for (int u = 0; u < U; ++u) cur[u] = rt_at(q + u * G);
    }
    for (int bi = 0; bi < nbatch; ++bi) {
      const bool more = bi + 1 < nbatch;
      if (more) {
#pragma unroll
        for (int u = 0; u < U; ++u) nxt[u] = rt_at(q + (U + u) * G);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int x = 0; x < DX; ++x) E::mad(acc[x], sP[x * K3 + q + u * G], cur[u]);
      }
      if (more) {
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
      }
      q += U * G;
    }
    for (; q < K3; q += G) {
      const T r0 = rt_at(q);
#pragma unroll
      for (int x = 0; x < DX; ++x) E::mad(acc[x], sP[x * K3 + q], r0);
    }
  }
  __syncthreads();   // every read of P is done: the reduction area may overlay it
  if (on) {
#pragma unroll
    for (int x = 0; x < DX; ++x)
      if (x < d) sRed[(grp * d + x) * Dr + l] = acc[x];
  }
}

template <bool CPLX>
__global__ __launch_bounds__(SM_THREADS) void k_heff_small(const SmallArgs g) {
  using E = Elem<CPLX>;
  using T = typename E::T;
  if (g.skip && *g.skip) return;
  extern __shared__ __attribute__((aligned(16))) double smem_raw[];
  const int tid = threadIdx.x;
  const int a = blockIdx.x;
  const int Dl = g.Dl, Dr = g.Dr, d = g.d, wl = g.wl, wr = g.wr;
  const int kh = g.kh, k0 = blockIdx.y * kh;     // this workgroup's slice of the ket bond of R
  const int N1 = d * Dr;                          // a row of C / of the result
  const int N1h = d * kh;                         // columns (e, kk) of this slice
  const bool has_w = g.W != nullptr;
  // LDS: [sparse W: counts (int), indices (int), values (double)] [L row] [T1] [X = K-group partials, then P]
  const int nrow_w = has_w ? d * wr : 0;
  int* s_cnt = reinterpret_cast<int*>(smem_raw);
  int* s_idx = reinterpret_cast<int*>(smem_raw + g.cnt_dbl);
  double* s_val = smem_raw + g.cnt_dbl + g.idx_dbl;
  T* s_el = reinterpret_cast<T*>(smem_raw + g.csr_dbl);
  T* sL = s_el + g.off_L;
  T* sT1 = s_el + g.off_T1;
  T* sX = s_el + g.off_X;
  T* sRed = s_el + g.off_red;

  // ---- the row of L, and the sparse form of W: row (x, f) lists its non-zero (b, e) as b * d + e
  {
    const T* Lrow = reinterpret_cast<const T*>(g.L) + (long long)a * wl * Dl;
    for (int i = tid; i < wl * Dl; i += SM_THREADS) sL[i] = Lrow[i];
    if (has_w) {
      for (int r = tid; r < nrow_w; r += SM_THREADS) {
        const int x = r / wr, f = r - x * wr;
        int cnt = 0;
        for (int b = 0; b < wl; ++b)
          for (int e = 0; e < d; ++e) {
            const double v = g.W[(((long long)b * d + x) * d + e) * wr + f];
            if (v != 0.0) {
              s_idx[r * g.csr_pitch + cnt] = b * d + e;
              s_val[r * g.csr_pitch + cnt] = v;
              ++cnt;
            }
          }
        s_cnt[r] = cnt;
      }
    }
  }
  __syncthreads();

  // ---- T1[b, (e, kk)] = sum_c L[a, b, c] C[c, e, k0 + kk]
  {
    const T* __restrict__ Cm = reinterpret_cast<const T*>(g.C);
    const int kg = g.kg;
    auto gcol = [&](int lc) {   // column of C behind column lc = e * kh + kk of the slice
      const int e = lc / kh;
      return e * Dr + k0 + (lc - e * kh);
    };
    if (kg == 1) {
      // two columns per thread and pass
      for (int col = tid; col < N1h; col += 2 * SM_THREADS) {
        const bool two = col + SM_THREADS < N1h;
        const int lcols[2] = {col, two ? col + SM_THREADS : col};
        const int cols[2] = {gcol(lcols[0]), gcol(lcols[1])};
        small_t1_wl<CPLX, 2>(wl, Cm, sL, Dl, N1, 0, 1, cols, lcols, two ? 2 : 1, sT1, N1h);
      }
    } else {
      // fewer columns than threads: kg groups of threads share the K range (c = grp, grp + kg, ..), partials through LDS
      const int grp = tid / N1h;
      const int col = tid - grp * N1h;
      if (grp < kg) {
        const int lcols[1] = {col};
        const int cols[1] = {gcol(col)};
        small_t1_wl<CPLX, 1>(wl, Cm, sL, Dl, N1, grp, kg, cols, lcols, 1, sX + grp * wl * N1h, N1h);
      }
      __syncthreads();
      for (int i = tid; i < wl * N1h; i += SM_THREADS) {
        T s = sX[i];
        for (int q = 1; q < kg; ++q) E::add(s, sX[q * wl * N1h + i]);
        sT1[i] = s;
      }
    }
  }
  __syncthreads();

  // ---- P[x, (f, kk)] = sum_{b, e} W[b, x, e, f] T1[b, e, kk]   (0-site: P = T1)
  const int K3 = wr * kh;
  const T* sP = sT1;
  if (has_w) {
    for (int i = tid; i < d * K3; i += SM_THREADS) {
      const int k = i % kh;
      const int r = i / kh;            // x * wr + f
      const int cnt = s_cnt[r];
      T acc = E::zero();
      for (int q = 0; q < cnt; ++q) E::mad_real(acc, s_val[r * g.csr_pitch + q], sT1[s_idx[r * g.csr_pitch + q] * kh + k]);
      sX[i] = acc;
    }
    sP = sX;
    __syncthreads();
  }

  // ---- part[a, x, l] = sum_q P[x, q] Rt[row(q), l], q = (f, kk); thread = (K group, l): group grp takes q = grp, grp + G, ..
  {
    const T* __restrict__ Rt = reinterpret_cast<const T*>(g.Rt);
    int G = SM_THREADS / Dr;
    if (G > K3) G = K3;
    const int grp = tid / Dr, l = tid - grp * Dr;
    if (d == 1)
      small_t3<CPLX, 1>(Rt, sP, K3, Dr, kh, k0, G, grp, l, sRed, d);
    else if (d == 2)
      small_t3<CPLX, 2>(Rt, sP, K3, Dr, kh, k0, G, grp, l, sRed, d);
    else if (d <= 4)
      small_t3<CPLX, 4>(Rt, sP, K3, Dr, kh, k0, G, grp, l, sRed, d);
    else if (d <= 8)
      small_t3<CPLX, 8>(Rt, sP, K3, Dr, kh, k0, G, grp, l, sRed, d);
    else
      small_t3<CPLX, 16>(Rt, sP, K3, Dr, kh, k0, G, grp, l, sRed, d);
    __syncthreads();
    double dre = 0.0, dim = 0.0;
    T* orow = reinterpret_cast<T*>(g.out) + (long long)blockIdx.y * g.part_stride + (long long)a * N1;
    const T* yrow = g.y ? reinterpret_cast<const T*>(g.y) + (long long)a * N1 : nullptr;
    for (int i = tid; i < N1; i += SM_THREADS) {
      T s = sRed[i];
      for (int q = 1; q < G; ++q) E::add(s, sRed[q * N1 + i]);
      orow[i] = s;
      if (yrow) E::dot(dre, dim, s, yrow[i]);
    }
    if (g.y) {   // grid-uniform; the dot product is linear in the parts: one partial per workgroup
      block_allsum2(dre, dim);
      if (tid == 0) {
        const int slot = a * gridDim.y + blockIdx.y;
        g.part[2 * slot] = dre;
        g.part[2 * slot + 1] = dim;
      }
    }
  }
}

// largest centre (elements) that takes this path; MPSE_SMALL=0 switches it off, MPSE_SMALL=<n> moves the limit
long long small_limit() {
  static const long long v = [] {
    const char* e = getenv("MPSE_SMALL");
    if (!e || !e[0]) return 32768ll;
    return atoll(e);
  }();
  return v;
}

int lds_limit_bytes() {
  static const int v = [] {
    int dev = 0, lim = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 65536;
    if (hipDeviceGetAttribute(&lim, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || lim <= 0) return 65536;
    return lim > 65536 ? 65536 : lim;    // (a second workgroup per compute unit stays possible)
  }();
  return v;
}

}  // namespace

void heff_small_drop_cache(mpse_ctx* ctx) {
  heff0_drop_cache(ctx);       // (the fused 0-site matvec keeps its per-solve data for the same span)
  if (ctx->small_rt.rt) mpse_free(ctx, ctx->small_rt.rt);
  ctx->small_rt = mpse_ctx::SmallRt();
}

int heff_small_try(mpse_ctx* ctx, int dtype, const mpse_heff* h, const void* C, void* out, bool* taken) {
  *taken = false;
  const mpse_dims& s = h->dims;
  if (h->nsite != 0 && h->nsite != 1) return MPSE_OK;
  const long long lim = small_limit();
  if (lim <= 0) return MPSE_OK;
  const int64_t Dl = s.Dl_ket, Dr = s.Dr_ket, wl = s.wl, wr = s.wr;
  const int64_t d = h->nsite == 1 ? s.d0 : 1;
  if ((s.Dl_bra > 0 && s.Dl_bra != Dl) || (s.Dr_bra > 0 && s.Dr_bra != Dr) || s.danc > 1) return MPSE_OK;
  if (h->l_dtype != dtype || h->r_dtype != dtype) return MPSE_OK;
  if (h->nsite == 1 && (!h->W0 || h->w_dtype != MPSE_F64)) return MPSE_OK;
  if (h->nsite == 0 && wl != wr) return MPSE_OK;
  if (Dl < 1 || Dr < 1 || Dl > 4096 || Dr > SM_THREADS || wl < 1 || wr < 1 || wl > SM_WMAX || wr > SM_WMAX || d < 1 ||
      d > SM_DMAX)
    return MPSE_OK;
  if (Dl * d * Dr > lim) return MPSE_OK;
  const bool cplx = dtype == MPSE_C128;
  const size_t es = dtype_size(dtype);
  // Slices of the ket bond of R: when the caller takes the result as a sum of parts (the Lanczos update adds them while
  // it reads), a row of L is worked on by KH workgroups, each with 1 / KH of the centre's columns and of R - the launch
  // covers KH times as many compute units and every workgroup streams 1 / KH of the bytes
  mpse_ctx::PartsReq& pr = ctx->parts_req;
  int64_t KH = 1;
  if (pr.ptr && pr.n == Dl * d * Dr) {
    for (int64_t c : {4, 2}) {
      if (Dr % c == 0 && Dr / c >= 16 && Dl * c <= 512 && pr.cap_elems >= c * pr.n &&
          (!ctx->dot_req.y || Dl * c <= ctx->dot_req.cap)) {
        KH = c;
        break;
      }
    }
  }
  const int64_t kh = Dr / KH;
  // LDS layout (mirrors the kernel)
  const int64_t N1 = d * kh, K3 = wr * kh;
  int kg = 1;
  if (N1 < SM_THREADS) {
    kg = int(SM_THREADS / N1);
    if (kg > Dl) kg = int(Dl);
    if (kg < 1) kg = 1;
  }
  const bool has_w = h->nsite == 1;
  const int64_t nrow_w = has_w ? d * wr : 0, pitch = has_w ? wl * d : 0;
  const int64_t cnt_dbl = (nrow_w + 1) / 2, idx_dbl = (nrow_w * pitch + 1) / 2;
  const int64_t csr_doubles = (cnt_dbl + idx_dbl + nrow_w * pitch + 1) & ~int64_t(1);
  int64_t G = SM_THREADS / Dr;
  if (G > K3) G = K3;
  const int64_t d_pad = d <= 2 ? d : (d <= 4 ? 4 : (d <= 8 ? 8 : 16));   // rows of P the last step reads (small_t3)
  const int64_t off_L = 0, off_T1 = wl * Dl, off_X = off_T1 + std::max<int64_t>(wl * N1, has_w ? 0 : d_pad * K3);
  const int64_t x_len = std::max<int64_t>(kg > 1 ? kg * wl * N1 : 0, has_w ? d_pad * K3 : 0);
  const int64_t red_len = G * d * Dr;
  const int64_t el = std::max<int64_t>(off_X + x_len, red_len);
  const int64_t lds = csr_doubles * 8 + el * int64_t(es);
  if (lds + 256 > lds_limit_bytes()) return MPSE_OK;     // (+ the static words of the block reduction)

  // transposed right environment: once per solve (the cache lives as long as the solve's occupancy caches)
  const size_t rbytes = size_t(Dr) * wr * Dr * es;
  TmpBuf rt_tmp(ctx);
  const double* rt = nullptr;
  const bool keep = ctx->occ_cache_on || ctx->small_rt_scope;
  if (keep && ctx->small_rt.src == h->R && ctx->small_rt.bytes == rbytes && ctx->small_rt.rt) {
    rt = static_cast<const double*>(ctx->small_rt.rt);
  } else {
    void* dst = nullptr;
    if (keep) {
      heff_small_drop_cache(ctx);
      MPSE_TRY(mpse_malloc(ctx, rbytes, &dst));
      ctx->small_rt.rt = dst;
      ctx->small_rt.src = h->R;
      ctx->small_rt.bytes = rbytes;
    } else {
      MPSE_TRY(rt_tmp.alloc(rbytes));
      dst = rt_tmp.p;
    }
    const long long n = (long long)Dr * wr * Dr;
    int nb = int((n + SM_THREADS - 1) / SM_THREADS);
    if (nb > 1024) nb = 1024;
    if (cplx)
      hipLaunchKernelGGL((k_env_transpose<true>), dim3(nb), dim3(SM_THREADS), 0, ctx->stream, (double*)dst,
                         (const double*)h->R, (int)Dr, (int)wr, ctx->skip_flag);
    else
      hipLaunchKernelGGL((k_env_transpose<false>), dim3(nb), dim3(SM_THREADS), 0, ctx->stream, (double*)dst,
                         (const double*)h->R, (int)Dr, (int)wr, ctx->skip_flag);
    rt = static_cast<const double*>(dst);
  }

  SmallArgs g{};
  g.L = static_cast<const double*>(h->L);
  g.Rt = rt;
  g.W = has_w ? static_cast<const double*>(h->W0) : nullptr;
  g.C = static_cast<const double*>(C);
  g.out = KH > 1 ? static_cast<double*>(pr.ptr) : static_cast<double*>(out);
  g.kh = (int)kh;
  g.part_stride = pr.n;
  g.skip = ctx->skip_flag;
  g.Dl = (int)Dl, g.Dr = (int)Dr, g.d = (int)d, g.wl = (int)wl, g.wr = (int)wr;
  g.kg = kg;
  g.off_L = (int)off_L, g.off_T1 = (int)off_T1, g.off_X = (int)off_X, g.off_red = 0;
  g.csr_pitch = (int)pitch;
  g.cnt_dbl = (int)cnt_dbl, g.idx_dbl = (int)idx_dbl, g.csr_dbl = (int)csr_doubles;
  if (ctx->dot_req.y && Dl * KH <= ctx->dot_req.cap) {
    g.y = static_cast<const double*>(ctx->dot_req.y);
    g.part = ctx->dot_req.part;
    ctx->dot_req.nb_out = (int)(Dl * KH);
  }
  if (cplx)
    hipLaunchKernelGGL((k_heff_small<true>), dim3((unsigned)Dl, (unsigned)KH), dim3(SM_THREADS), (size_t)lds, ctx->stream, g);
  else
    hipLaunchKernelGGL((k_heff_small<false>), dim3((unsigned)Dl, (unsigned)KH), dim3(SM_THREADS), (size_t)lds, ctx->stream, g);
  MPSE_HIP(ctx, hipGetLastError());
  pr.used = KH > 1 ? (int)KH : 0;
  *taken = true;
  return MPSE_OK;
}
