// Davidson eigensolver on device-resident vectors (replaces lib/davidson/davidson.py:154-441 as called at
// mps/gs.py:533-538) and the integer basis selection of mps/lib.py:253-322.
//
// The iteration is driven from the host side of the engine, not from Python: per cycle the subspace matrix grows
// by one column obtained from ONE batched reduction launch (all <V_i, W_j> at once), the Ritz vector, its image and
// the residual come from one fused pass over the basis, Gram-Schmidt coefficients never leave the device.  The host
// sees two small read-backs per cycle (the new column of the subspace matrix, the residual norm).
#include <algorithm>
#include <cmath>
#include <complex>
#include <limits>
#include <vector>

#include "mpse_device.h"
#include "mpse_internal.h"

typedef std::complex<double> cd;

namespace {

constexpr int DV_MAX = 48;          // most basis vectors ever held
constexpr int DV_RED_MAX_BLOCKS = 256;

inline int dv_blocks(int64_t n_doubles) {
  int64_t b = (n_doubles + RED_THREADS * 8 - 1) / (RED_THREADS * 8);
  if (b < 1) b = 1;
  if (b > DV_RED_MAX_BLOCKS) b = DV_RED_MAX_BLOCKS;
  return (int)b;
}

// partial[(v * gridDim.x + blockIdx.x) * 2 ..] = sum over this block's elements of conj(X_v) * y, v = blockIdx.y
template <bool CPLX>
__global__ __launch_bounds__(RED_THREADS) void k_multi_dot(const double* __restrict__ X, long long stride_x,
                                                           const double* __restrict__ y, long long n,
                                                           double* __restrict__ partial) {
  const double* x = X + (long long)blockIdx.y * stride_x * (CPLX ? 2 : 1);
  double re = 0, im = 0;
  const long long step = (long long)gridDim.x * RED_THREADS;
  if (CPLX) {
    const double2* x2 = reinterpret_cast<const double2*>(x);
    const double2* y2 = reinterpret_cast<const double2*>(y);
    for (long long i = (long long)blockIdx.x * RED_THREADS + threadIdx.x; i < n; i += step) {
      const double2 a = x2[i], b = y2[i];
      re += a.x * b.x + a.y * b.y;
      im += a.x * b.y - a.y * b.x;
    }
  } else {
    for (long long i = (long long)blockIdx.x * RED_THREADS + threadIdx.x; i < n; i += step) re += x[i] * y[i];
  }
  block_allsum2(re, im);
  if (threadIdx.x == 0) {
    double* p = partial + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 2;
    p[0] = re;
    p[1] = im;
  }
}

// out[2 v ..] = sum of the nb partials of value v (fixed order); one block per value
__global__ __launch_bounds__(RED_THREADS) void k_multi_reduce(const double* __restrict__ partial, int nb,
                                                              double* __restrict__ out) {
  const double* p = partial + (long long)blockIdx.x * nb * 2;
  double re = 0, im = 0;
  for (int i = threadIdx.x; i < nb; i += RED_THREADS) {
    re += p[2 * i];
    im += p[2 * i + 1];
  }
  block_allsum2(re, im);
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = re;
    out[2 * blockIdx.x + 1] = im;
  }
}

struct DvCoefs {
  double re[DV_MAX];
  double im[DV_MAX];
};

// x = sum_i c_i V_i ; hx = sum_i c_i W_i ; r = hx - e x ; partial |r|^2 per block
template <bool CPLX>
__global__ __launch_bounds__(RED_THREADS) void k_ritz(double* __restrict__ x, double* __restrict__ hx,
                                                      double* __restrict__ r, const double* __restrict__ V,
                                                      const double* __restrict__ W, long long n, int m, DvCoefs c,
                                                      double e, double* __restrict__ partial) {
  const long long step = (long long)gridDim.x * RED_THREADS;
  double s = 0, zero = 0;
  for (long long i = (long long)blockIdx.x * RED_THREADS + threadIdx.x; i < n; i += step) {
    if (CPLX) {
      double xr = 0, xi = 0, hr = 0, hi = 0;
      for (int j = 0; j < m; ++j) {
        const double2 v = reinterpret_cast<const double2*>(V)[(long long)j * n + i];
        const double2 w = reinterpret_cast<const double2*>(W)[(long long)j * n + i];
        xr += c.re[j] * v.x - c.im[j] * v.y;
        xi += c.re[j] * v.y + c.im[j] * v.x;
        hr += c.re[j] * w.x - c.im[j] * w.y;
        hi += c.re[j] * w.y + c.im[j] * w.x;
      }
      const double rr = hr - e * xr, ri = hi - e * xi;
      reinterpret_cast<double2*>(x)[i] = make_double2(xr, xi);
      reinterpret_cast<double2*>(hx)[i] = make_double2(hr, hi);
      reinterpret_cast<double2*>(r)[i] = make_double2(rr, ri);
      s += rr * rr + ri * ri;
    } else {
      double xr = 0, hr = 0;
      for (int j = 0; j < m; ++j) {
        xr += c.re[j] * V[(long long)j * n + i];
        hr += c.re[j] * W[(long long)j * n + i];
      }
      const double rr = hr - e * xr;
      x[i] = xr;
      hx[i] = hr;
      r[i] = rr;
      s += rr * rr;
    }
  }
  block_allsum2(s, zero);
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = s;
    partial[2 * blockIdx.x + 1] = 0.0;
  }
}

// t -= sum_i ov_i X_i (and optionally u -= sum_i ov_i Y_i) with ov read from device memory (re, im pairs)
template <bool CPLX>
__global__ void k_project_out(double* __restrict__ t, const double* __restrict__ X, long long n, int m,
                              const double* __restrict__ ov, double* __restrict__ u, const double* __restrict__ Y) {
  const long long step = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    if (CPLX) {
      double2 a = reinterpret_cast<double2*>(t)[i];
      double2 b = u ? reinterpret_cast<double2*>(u)[i] : make_double2(0.0, 0.0);
      for (int j = 0; j < m; ++j) {
        const double cr = ov[2 * j], ci = ov[2 * j + 1];
        const double2 v = reinterpret_cast<const double2*>(X)[(long long)j * n + i];
        a.x -= cr * v.x - ci * v.y;
        a.y -= cr * v.y + ci * v.x;
        if (u) {
          const double2 w = reinterpret_cast<const double2*>(Y)[(long long)j * n + i];
          b.x -= cr * w.x - ci * w.y;
          b.y -= cr * w.y + ci * w.x;
        }
      }
      reinterpret_cast<double2*>(t)[i] = a;
      if (u) reinterpret_cast<double2*>(u)[i] = b;
    } else {
      double a = t[i], b = u ? u[i] : 0.0;
      for (int j = 0; j < m; ++j) {
        a -= ov[2 * j] * X[(long long)j * n + i];
        if (u) b -= ov[2 * j] * Y[(long long)j * n + i];
      }
      t[i] = a;
      if (u) u[i] = b;
    }
  }
}

// dst = src / sqrt(*n2) (and dst2 = src2 / sqrt(*n2)); a vanishing norm leaves zeros
__global__ void k_scale_rsqrt(double* dst, const double* __restrict__ src, double* dst2, const double* __restrict__ src2,
                              long long n_doubles, const double* __restrict__ n2) {
  const double v = *n2;
  const double s = v > 0.0 ? 1.0 / sqrt(v) : 0.0;
  const long long step = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_doubles; i += step) {
    dst[i] = src[i] * s;
    if (dst2) dst2[i] = src2[i] * s;
  }
}

inline int ew_blocks(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

// Hermitian eigenproblem of the subspace matrix (row-major a[m*m]) by cyclic Jacobi rotations; eigenvalues ascending,
// eigenvectors are the columns of u.
void herm_eig(int m, std::vector<cd> a, std::vector<double>& w, std::vector<cd>& u) {
  u.assign((size_t)m * m, cd(0));
  for (int i = 0; i < m; ++i) u[(size_t)i * m + i] = 1.0;
  for (int sweep = 0; sweep < 80; ++sweep) {
    double off = 0, diag = 0;
    for (int i = 0; i < m; ++i) {
      diag += std::norm(a[(size_t)i * m + i]);
      for (int j = i + 1; j < m; ++j) off += std::norm(a[(size_t)i * m + j]);
    }
    if (off == 0.0 || off <= 1e-34 * (diag + off)) break;
    for (int p = 0; p < m - 1; ++p)
      for (int q = p + 1; q < m; ++q) {
        const cd apq = a[(size_t)p * m + q];
        const double g = std::abs(apq);
        if (g == 0.0) continue;
        const double app = a[(size_t)p * m + p].real(), aqq = a[(size_t)q * m + q].real();
        const cd ph = apq / g;                       // a_pq = g e^{i phi}
        const double theta = (aqq - app) / (2.0 * g);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        // columns: col_p' = c col_p - s conj(ph) col_q ; col_q' = s ph col_p + c col_q  (unitary G), A <- G^H A G
        for (int k = 0; k < m; ++k) {
          const cd akp = a[(size_t)k * m + p], akq = a[(size_t)k * m + q];
          a[(size_t)k * m + p] = c * akp - s * std::conj(ph) * akq;
          a[(size_t)k * m + q] = s * ph * akp + c * akq;
        }
        for (int k = 0; k < m; ++k) {
          const cd apk = a[(size_t)p * m + k], aqk = a[(size_t)q * m + k];
          a[(size_t)p * m + k] = c * apk - s * ph * aqk;
          a[(size_t)q * m + k] = s * std::conj(ph) * apk + c * aqk;
        }
        for (int k = 0; k < m; ++k) {
          const cd ukp = u[(size_t)k * m + p], ukq = u[(size_t)k * m + q];
          u[(size_t)k * m + p] = c * ukp - s * std::conj(ph) * ukq;
          u[(size_t)k * m + q] = s * ph * ukp + c * ukq;
        }
      }
  }
  std::vector<int> order(m);
  for (int i = 0; i < m; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return a[(size_t)x * m + x].real() < a[(size_t)y * m + y].real(); });
  w.resize(m);
  std::vector<cd> us((size_t)m * m);
  for (int j = 0; j < m; ++j) {
    w[j] = a[(size_t)order[j] * m + order[j]].real();
    for (int k = 0; k < m; ++k) us[(size_t)k * m + j] = u[(size_t)k * m + order[j]];
  }
  u.swap(us);
}

struct Dav {
  mpse_ctx* ctx;
  int dtype;
  bool cplx;
  const mpse_heff* h;
  int twolayer;
  const void* mask;
  int64_t n;
  size_t es;
  int nb;
  char* V;        // basis vectors, n elements each
  char* W;        // their images
  double* part;   // partial sums
  double* vals;   // reduced values (re, im pairs)
  std::vector<cd> hsub;   // DV_MAX x DV_MAX
  int nmatvec = 0;

  char* v(int i) { return V + size_t(i) * n * es; }
  char* w(int i) { return W + size_t(i) * n * es; }

  int apply(const void* x, void* y) {
    ++nmatvec;
    if (twolayer)
      MPSE_TRY(mpse_heff_apply2(ctx, dtype, h, x, y));
    else
      MPSE_TRY(mpse_heff_apply(ctx, dtype, h, x, y));
    if (mask) MPSE_TRY(mpse_mul_real(ctx, dtype, y, mask, n));
    return MPSE_OK;
  }
  // vals[2 v ..] = <X_v, y>, v < nx  (device resident)
  int dots(const void* X, int nx, const void* y, double* dst) {
    if (nx <= 0) return MPSE_OK;
    if (cplx)
      hipLaunchKernelGGL((k_multi_dot<true>), dim3(nb, nx), dim3(RED_THREADS), 0, ctx->stream, (const double*)X,
                         (long long)n, (const double*)y, (long long)n, part);
    else
      hipLaunchKernelGGL((k_multi_dot<false>), dim3(nb, nx), dim3(RED_THREADS), 0, ctx->stream, (const double*)X,
                         (long long)n, (const double*)y, (long long)n, part);
    hipLaunchKernelGGL(k_multi_reduce, dim3(nx), dim3(RED_THREADS), 0, ctx->stream, (const double*)part, nb, dst);
    MPSE_HIP(ctx, hipGetLastError());
    return MPSE_OK;
  }
  int fetch(const double* dsrc, int count, std::vector<double>& out) {
    MPSE_TRY(publish_and_wait(ctx, dsrc, count, 64));
    out.assign(ctx->pinned + 64, ctx->pinned + 64 + count);
    return MPSE_OK;
  }
  // column j of the subspace matrix: h[i][j] = <V_i, W_j>, i <= j  (one launch + one read-back)
  int gram_column(int j) {
    MPSE_TRY(dots(V, j + 1, w(j), vals));
    std::vector<double> g;
    MPSE_TRY(fetch(vals, 2 * (j + 1), g));
    for (int i = 0; i <= j; ++i) {
      const cd x(g[2 * i], cplx ? g[2 * i + 1] : 0.0);
      hsub[(size_t)i * DV_MAX + j] = x;
      hsub[(size_t)j * DV_MAX + i] = std::conj(x);
    }
    hsub[(size_t)j * DV_MAX + j] = hsub[(size_t)j * DV_MAX + j].real();
    return MPSE_OK;
  }
  void sub_eig(int m, std::vector<double>& ew, std::vector<cd>& ev) {
    std::vector<cd> a((size_t)m * m);
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < m; ++j) a[(size_t)i * m + j] = hsub[(size_t)i * DV_MAX + j];
    herm_eig(m, a, ew, ev);
  }
  // x = V c, hx = W c, r = hx - e x ; returns |r|
  int ritz(int m, const std::vector<cd>& ev, int col, double e, void* x, void* hx, void* r, double* rnorm) {
    DvCoefs c;
    for (int i = 0; i < m; ++i) {
      c.re[i] = ev[(size_t)i * m + col].real();
      c.im[i] = ev[(size_t)i * m + col].imag();
    }
    if (cplx)
      hipLaunchKernelGGL((k_ritz<true>), dim3(nb), dim3(RED_THREADS), 0, ctx->stream, (double*)x, (double*)hx,
                         (double*)r, (const double*)V, (const double*)W, (long long)n, m, c, e, part);
    else
      hipLaunchKernelGGL((k_ritz<false>), dim3(nb), dim3(RED_THREADS), 0, ctx->stream, (double*)x, (double*)hx,
                         (double*)r, (const double*)V, (const double*)W, (long long)n, m, c, e, part);
    hipLaunchKernelGGL(k_multi_reduce, dim3(1), dim3(RED_THREADS), 0, ctx->stream, (const double*)part, nb, vals);
    MPSE_HIP(ctx, hipGetLastError());
    std::vector<double> g;
    MPSE_TRY(fetch(vals, 2, g));
    *rnorm = std::sqrt(g[0] > 0 ? g[0] : 0.0);
    return MPSE_OK;
  }
  // Gram-Schmidt (twice) of t (and, with it, of its image u) against X_0..X_{m-1} / Y_0..; then t /= |t| (u /= |t|)
  // into dst (dst2).  The squared norm before scaling is left at vals + 2 * DV_MAX (read later by the host).
  int orthonormalise(void* t, void* u, const void* X, const void* Y, int m, void* dst, void* dst2) {
    for (int pass = 0; pass < 2 && m > 0; ++pass) {
      MPSE_TRY(dots(X, m, t, vals));
      if (cplx)
        hipLaunchKernelGGL((k_project_out<true>), dim3(ew_blocks(n)), dim3(256), 0, ctx->stream, (double*)t,
                           (const double*)X, (long long)n, m, (const double*)vals, (double*)u, (const double*)Y);
      else
        hipLaunchKernelGGL((k_project_out<false>), dim3(ew_blocks(n)), dim3(256), 0, ctx->stream, (double*)t,
                           (const double*)X, (long long)n, m, (const double*)vals, (double*)u, (const double*)Y);
    }
    MPSE_TRY(dots(t, 1, t, vals + 2 * DV_MAX));
    hipLaunchKernelGGL(k_scale_rsqrt, dim3(ew_blocks(n * (cplx ? 2 : 1))), dim3(256), 0, ctx->stream, (double*)dst,
                       (const double*)t, (double*)dst2, (const double*)u, (long long)(n * (cplx ? 2 : 1)),
                       (const double*)(vals + 2 * DV_MAX));
    MPSE_HIP(ctx, hipGetLastError());
    return MPSE_OK;
  }
  int norm2_host(double* out) {
    std::vector<double> g;
    MPSE_TRY(fetch(vals + 2 * DV_MAX, 1, g));
    *out = g[0];
    return MPSE_OK;
  }
};

}  // namespace

extern "C" int mpse_davidson(mpse_ctx* ctx, int dtype, const mpse_heff* h, int twolayer, const void* hdiag_f64,
                             const void* mask_f64, int nroots, int nguess, const void* guess, double tol, int max_cycle,
                             int max_space, double lindep, double shift, double* e_host, void* x_out, int* ncycle,
                             int* nmatvec) {
  if (!ctx || !h || !hdiag_f64 || !guess || !e_host || !x_out) return MPSE_ERR_ARG;
  MPSE_BIND(ctx);
  if (dtype != MPSE_F64 && dtype != MPSE_C128) return mpse_fail(ctx, MPSE_ERR_ARG, "davidson: unknown dtype");
  if (nroots < 1 || nguess < 1 || nroots > 16) return mpse_fail(ctx, MPSE_ERR_ARG, "davidson: 1 <= nroots <= 16 and at least one guess");
  const mpse_dims& s = h->dims;
  if ((s.Dl_bra > 0 && s.Dl_bra != s.Dl_ket) || (s.Dr_bra > 0 && s.Dr_bra != s.Dr_ket))
    return mpse_fail(ctx, MPSE_ERR_SHAPE, "davidson: the effective Hamiltonian must be square (bra bonds == ket bonds)");
  if (h->nsite != 1 && h->nsite != 2) return mpse_fail(ctx, MPSE_ERR_ARG, "davidson: one- or two-site centres");
  SmallRtScope small_rt_scope(ctx);
  const int64_t anc = s.danc > 0 ? s.danc : 1;
  int64_t n = s.Dl_ket * s.Dr_ket * s.d0 * anc;
  if (h->nsite == 2) n *= s.d1 * (s.danc1 > 0 ? s.danc1 : anc);
  if (n <= 0) return mpse_fail(ctx, MPSE_ERR_SHAPE, "davidson: empty centre tensor");
  if (max_space <= 0) max_space = 12 + (nroots - 1) * 3;
  if (max_space + nroots + 1 > DV_MAX) return mpse_fail(ctx, MPSE_ERR_ARG, "davidson: max_space too large");
  if (max_cycle <= 0) max_cycle = 100;
  const bool cplx = dtype == MPSE_C128;
  const size_t es = dtype_size(dtype);
  // tol > 0: PySCF's rule (|de| < tol and |r| < sqrt(tol)); tol < 0: the residual alone decides, |r| < -tol (PRIMME's
  // convergence test, mps/gs.py:552-569 of the reference)
  const bool res_only = tol < 0.0;
  const double toloose = res_only ? -tol : std::sqrt(tol);
  const int cap = max_space + nroots + 1;

  TmpBuf VB(ctx), WB(ctx), PB(ctx), SB(ctx), XB(ctx), HXB(ctx), RB(ctx), TB(ctx);
  MPSE_TRY(VB.alloc(size_t(cap) * n * es));
  MPSE_TRY(WB.alloc(size_t(cap) * n * es));
  Dav d;
  d.ctx = ctx, d.dtype = dtype, d.cplx = cplx, d.h = h, d.twolayer = twolayer, d.mask = mask_f64, d.n = n, d.es = es;
  d.nb = dv_blocks(n * (cplx ? 2 : 1));
  MPSE_TRY(PB.alloc(size_t(DV_MAX) * d.nb * 2 * sizeof(double)));
  MPSE_TRY(SB.alloc(size_t(4 * DV_MAX) * sizeof(double)));
  d.V = VB.as<char>(), d.W = WB.as<char>(), d.part = PB.as<double>(), d.vals = SB.as<double>();
  d.hsub.assign(size_t(DV_MAX) * DV_MAX, cd(0));
  // per root: Ritz vector, its image, residual
  MPSE_TRY(XB.alloc(size_t(nroots) * n * es));
  MPSE_TRY(HXB.alloc(size_t(nroots) * n * es));
  MPSE_TRY(RB.alloc(size_t(nroots) * n * es));
  MPSE_TRY(TB.alloc(size_t(n) * es));
  auto xs = [&](int r) { return XB.as<char>() + size_t(r) * n * es; };
  auto hxs = [&](int r) { return HXB.as<char>() + size_t(r) * n * es; };
  auto rs = [&](int r) { return RB.as<char>() + size_t(r) * n * es; };
  void* t = TB.p;

  // ---- initial basis: masked guesses, orthonormalised; dependent / vanishing ones are dropped
  int m = 0;
  for (int g = 0; g < nguess && m < cap - 1; ++g) {
    MPSE_TRY(mpse_memcpy_d2d(ctx, t, (const char*)guess + size_t(g) * n * es, size_t(n) * es));
    if (mask_f64) MPSE_TRY(mpse_mul_real(ctx, dtype, t, mask_f64, n));
    double n0 = 0;
    MPSE_TRY(d.dots(t, 1, t, d.vals + 2 * DV_MAX));
    MPSE_TRY(d.norm2_host(&n0));
    if (!(n0 > 0)) continue;
    MPSE_TRY(mpse_scal(ctx, dtype, t, n, 1.0 / std::sqrt(n0), 0.0));
    double n2 = 1.0;
    if (m > 0) {
      MPSE_TRY(d.orthonormalise(t, nullptr, d.V, nullptr, m, d.v(m), nullptr));
      MPSE_TRY(d.norm2_host(&n2));
      if (n2 < lindep) continue;
    } else {
      MPSE_TRY(mpse_memcpy_d2d(ctx, d.v(0), t, size_t(n) * es));
    }
    ++m;
  }
  if (m == 0) return mpse_fail(ctx, MPSE_ERR_ARG, "davidson: zero initial guess");
  int nw = 0;                      // images / subspace columns available
  std::vector<double> ew, e_last, es_out(nroots, 0.0);
  std::vector<cd> ev;
  int cyc = 0, k = 0;
  bool pending_check = false;      // the last appended direction has not been checked against lindep yet
  for (cyc = 1; cyc <= max_cycle; ++cyc) {
    if (pending_check) {
      double n2 = 0;
      MPSE_TRY(d.norm2_host(&n2));
      pending_check = false;
      if (n2 < lindep) {           // what was appended is rounding noise: drop it and stop (davidson.py:405-409)
        --m;
        --cyc;
        break;
      }
    }
    for (; nw < m; ++nw) {
      MPSE_TRY(d.apply(d.v(nw), d.w(nw)));
      MPSE_TRY(d.gram_column(nw));
    }
    d.sub_eig(m, ew, ev);
    k = nroots < m ? nroots : m;
    std::vector<double> rn(k, 0.0);
    std::vector<char> conv(k, 0);
    for (int r = 0; r < k; ++r) {
      MPSE_TRY(d.ritz(m, ev, r, ew[r], xs(r), hxs(r), rs(r), &rn[r]));
      const double de = (r < (int)e_last.size()) ? ew[r] - e_last[r] : std::numeric_limits<double>::infinity();
      conv[r] = (res_only ? rn[r] < toloose : ((std::fabs(de) < tol && rn[r] < toloose) || rn[r] < 1e-14)) ? 1 : 0;
      es_out[r] = ew[r];
    }
    e_last.assign(ew.begin(), ew.begin() + k);
    bool all = (k == nroots);
    for (int r = 0; r < k; ++r) all = all && conv[r];
    if (all) break;
    if (nroots > 1 && m >= n) break;   // the basis spans the whole space
    std::vector<int> todo;
    for (int r = 0; r < k; ++r)
      if (!conv[r]) todo.push_back(r);
    if (todo.empty())
      for (int r = 0; r < k; ++r) todo.push_back(r);
    // preconditioned residuals first (they only need r, e): the restart below rewrites the basis
    // restart from the current Ritz vectors when the space is full (davidson.py:423-427)
    const bool restart = (nroots == 1) ? (m >= max_space || m >= n) : (m + (int)todo.size() > max_space);
    if (restart) {
      int mm = 0;
      for (int r = 0; r < k; ++r) {
        // Ritz vectors are orthonormal up to rounding; re-orthonormalise them (and their images with the same
        // coefficients) so that the restarted basis is clean
        if (mm == 0) {
          MPSE_TRY(d.dots(xs(r), 1, xs(r), d.vals + 2 * DV_MAX));
          hipLaunchKernelGGL(k_scale_rsqrt, dim3(ew_blocks(n * (cplx ? 2 : 1))), dim3(256), 0, ctx->stream,
                             (double*)d.v(0), (const double*)xs(r), (double*)d.w(0), (const double*)hxs(r),
                             (long long)(n * (cplx ? 2 : 1)), (const double*)(d.vals + 2 * DV_MAX));
          MPSE_HIP(ctx, hipGetLastError());
        } else {
          MPSE_TRY(d.orthonormalise(xs(r), hxs(r), d.V, d.W, mm, d.v(mm), d.w(mm)));
        }
        ++mm;
      }
      m = mm;
      nw = m;
      // subspace matrix of the restarted basis: columns recomputed (k <= nroots launches)
      for (int j = 0; j < m; ++j) MPSE_TRY(d.gram_column(j));
    }
    int added = 0;
    for (int r : todo) {
      if (m >= cap - 1) break;
      MPSE_TRY(mpse_davidson_precond(ctx, dtype, t, rs(r), hdiag_f64, mask_f64, n, ew[r], shift));
      MPSE_TRY(d.orthonormalise(t, nullptr, d.V, nullptr, m, d.v(m), nullptr));
      if (nroots == 1) {
        pending_check = true;      // verified at the top of the next cycle, together with its other read-backs
        ++m;
        ++added;
      } else {
        double n2 = 0;
        MPSE_TRY(d.norm2_host(&n2));
        if (n2 >= lindep) {
          ++m;
          ++added;
        }
      }
    }
    if (added == 0) break;
  }
  if (cyc > max_cycle) cyc = max_cycle;
  for (int r = 0; r < nroots; ++r) e_host[r] = r < k ? es_out[r] : std::numeric_limits<double>::quiet_NaN();
  MPSE_TRY(mpse_memcpy_d2d(ctx, x_out, XB.p, size_t(k) * n * es));
  if (k < nroots) MPSE_TRY(mpse_memset_zero(ctx, (char*)x_out + size_t(k) * n * es, size_t(nroots - k) * n * es));
  if (ncycle) *ncycle = cyc;
  if (nmatvec) *nmatvec = d.nmatvec;
  return MPSE_OK;
}

// Indices of the renormalised basis states to keep, replaces select_basis of mps/lib.py:253-322: an equal quota of
// int(m_max * percent / nblocks) states for every quantum-number block (blocks in ascending id order, inside a block
// by descending weight), the remaining slots by descending weight over everything that is left; ties keep the
// original order (stable).  block_id_host[i] = rank of state i's quantum number among the distinct ones.  Host only.
extern "C" int mpse_truncate_select(const double* sigma_host, const int64_t* block_id_host, int64_t count,
                                    int64_t m_max, double percent, int64_t* picked_host, int64_t* npicked) {
  if (count < 0 || (count && (!sigma_host || !picked_host)) || !npicked) return MPSE_ERR_ARG;
  const int64_t nbasis = std::min<int64_t>(count, m_max < 0 ? 0 : m_max);
  std::vector<int64_t> remaining(count), picked;
  for (int64_t i = 0; i < count; ++i) remaining[i] = i;
  auto by_weight = [&](int64_t a, int64_t b) { return sigma_host[a] > sigma_host[b]; };
  if (percent != 0 && block_id_host && count > 0) {
    std::vector<int64_t> ids(block_id_host, block_id_host + count);
    std::sort(ids.begin(), ids.end());
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    const int64_t per_block = (int64_t)((double)nbasis * percent / (double)ids.size());
    std::vector<char> taken(count, 0);
    for (int64_t b : ids) {
      std::vector<int64_t> members;
      for (int64_t i : remaining)
        if (block_id_host[i] == b) members.push_back(i);
      std::stable_sort(members.begin(), members.end(), by_weight);
      const int64_t take = std::min<int64_t>(per_block, (int64_t)members.size());
      for (int64_t j = 0; j < take; ++j) {
        picked.push_back(members[j]);
        taken[members[j]] = 1;
      }
      std::vector<int64_t> rest;
      for (int64_t i : remaining)
        if (!taken[i]) rest.push_back(i);
      remaining.swap(rest);
    }
  }
  const int64_t rest = nbasis - (int64_t)picked.size();
  std::stable_sort(remaining.begin(), remaining.end(), by_weight);
  for (int64_t j = 0; j < rest && j < (int64_t)remaining.size(); ++j) picked.push_back(remaining[j]);
  for (size_t j = 0; j < picked.size(); ++j) picked_host[j] = picked[j];
  *npicked = (int64_t)picked.size();
  return MPSE_OK;
}
