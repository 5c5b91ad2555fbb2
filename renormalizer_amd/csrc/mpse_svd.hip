// Quantum-number blocked economic SVD by one-sided (Hestenes) Jacobi on the device.
// Replaces scipy.linalg.svd(gesdd) per block in mps/svd_qn.py:12-49,177-213.
//
// Per block (mm >= nn; wide blocks are processed as their adjoint):
//   * the block is gathered column-major, V = I;
//   * round-robin sweeps: every launch orthogonalises nn/2 disjoint column pairs, one
//     workgroup per pair (Gram entries by wavefront-shuffle reductions, then the plane
//     rotation on the A and V columns); a sweep without rotations ends the iteration;
//   * sigma_j = |a_j| are read back, sorted on the host (descending, like LAPACK);
//   * columns are normalised; numerically null columns are zeroed and the left basis is
//     completed to an exact isometry by a Householder QR of the normalised matrix
//     (Q R with |R_jj| = 1 on the non-null columns), which is what LAPACK's economic U
//     guarantees and what the sweep algorithms above this layer rely on.
#include <algorithm>
#include <cmath>
#include <numeric>

#include "mpse_device.h"
#include "mpse_internal.h"

namespace {

template <bool CPLX>
struct Cx;
template <>
struct Cx<true> {
  static constexpr int E = 2;
  __device__ static double2 ld(const double* p, long long i) { return reinterpret_cast<const double2*>(p)[i]; }
  __device__ static void st(double* p, long long i, double2 v) { reinterpret_cast<double2*>(p)[i] = v; }
};
template <>
struct Cx<false> {
  static constexpr int E = 1;
  __device__ static double2 ld(const double* p, long long i) { return make_double2(p[i], 0.0); }
  __device__ static void st(double* p, long long i, double2 v) { p[i] = v.x; }
};

inline int ew_blocks(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}

template <bool CPLX>
__global__ void k_gather_block(double* ws, const double* __restrict__ coef, long long ncol,
                               const long long* __restrict__ rows, const long long* __restrict__ cols, int mm, int nn,
                               int herm) {
  const long long total = (long long)mm * nn;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    if (!herm) {
      const int c = (int)(t % nn), r = (int)(t / nn);
      Cx<CPLX>::st(ws, r + (long long)c * mm, Cx<CPLX>::ld(coef, rows[r] * ncol + cols[c]));
    } else {
      const int r = (int)(t % mm), c = (int)(t / mm);
      double2 v = Cx<CPLX>::ld(coef, rows[c] * ncol + cols[r]);
      v.y = -v.y;
      Cx<CPLX>::st(ws, r + (long long)c * mm, v);
    }
  }
}

template <bool CPLX>
__global__ void k_set_identity(double* v, int n) {
  const long long total = (long long)n * n;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride)
    Cx<CPLX>::st(v, t, make_double2((t % n) == (t / n) ? 1.0 : 0.0, 0.0));
}

// one round-robin step: workgroup b handles the pair (p,q) of step `step` (circle method on N = even(nn))
template <bool CPLX>
__global__ __launch_bounds__(RED_THREADS) void k_jacobi_step(double* a, double* v, int mm, int nn, int N, int step,
                                                             double tol, double null2, int* nrot) {
  constexpr int E = Cx<CPLX>::E;
  const int kk = blockIdx.x;
  int p, q;
  if (kk == 0) {
    p = step % (N - 1);
    q = N - 1;
  } else {
    p = (step + kk) % (N - 1);
    q = (step - kk + (N - 1)) % (N - 1);
  }
  if (p > q) {
    const int t = p;
    p = q;
    q = t;
  }
  if (q >= nn) return;  // padding column of an odd nn
  double* ap = a + (long long)p * mm * E;
  double* aq = a + (long long)q * mm * E;
  double alpha = 0, beta = 0, gr = 0, gi = 0;
  for (int r = threadIdx.x; r < mm; r += RED_THREADS) {
    const double2 x = Cx<CPLX>::ld(ap, r), y = Cx<CPLX>::ld(aq, r);
    alpha += x.x * x.x + x.y * x.y;
    beta += y.x * y.x + y.y * y.y;
    gr += x.x * y.x + x.y * y.y;  // conj(x) * y
    gi += x.x * y.y - x.y * y.x;
  }
  block_allsum2(alpha, beta);
  block_allsum2(gr, gi);
  const double g = sqrt(gr * gr + gi * gi);
  // A column whose norm is below (largest column norm) * eps * m can only belong to singular values that are
  // numerically zero; it is zeroed and replaced by the null-space completion afterwards, so rotating it
  // (endlessly, at rounding level) is pointless.  Also avoids forming alpha*beta, which underflows.
  if (g == 0.0 || alpha <= null2 || beta <= null2) return;
  if (!(g > tol * sqrt(alpha) * sqrt(beta))) return;  // already orthogonal (block-uniform decision)
  // phase of gamma and the real Jacobi rotation for [[alpha, g], [g, beta]]
  const double pr = gr / g, pi = gi / g;
  const double zeta = (beta - alpha) / (2.0 * g);
  const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
  const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
  // a_p' = c a_p - s e^{-i phi} a_q ; a_q' = s a_p + c e^{-i phi} a_q     (e^{-i phi} = (pr, -pi))
  for (int r = threadIdx.x; r < mm; r += RED_THREADS) {
    const double2 x = Cx<CPLX>::ld(ap, r), y0 = Cx<CPLX>::ld(aq, r);
    const double2 y = make_double2(y0.x * pr + y0.y * pi, y0.y * pr - y0.x * pi);
    Cx<CPLX>::st(ap, r, make_double2(c * x.x - s * y.x, c * x.y - s * y.y));
    Cx<CPLX>::st(aq, r, make_double2(s * x.x + c * y.x, s * x.y + c * y.y));
  }
  double* vp = v + (long long)p * nn * E;
  double* vq = v + (long long)q * nn * E;
  for (int r = threadIdx.x; r < nn; r += RED_THREADS) {
    const double2 x = Cx<CPLX>::ld(vp, r), y0 = Cx<CPLX>::ld(vq, r);
    const double2 y = make_double2(y0.x * pr + y0.y * pi, y0.y * pr - y0.x * pi);
    Cx<CPLX>::st(vp, r, make_double2(c * x.x - s * y.x, c * x.y - s * y.y));
    Cx<CPLX>::st(vq, r, make_double2(s * x.x + c * y.x, s * x.y + c * y.y));
  }
  if (threadIdx.x == 0) atomicAdd(nrot, 1);
}

// sig[c] = |a_c|
template <bool CPLX>
__global__ __launch_bounds__(RED_THREADS) void k_col_norms(const double* __restrict__ a, int mm, double* sig) {
  const double* col = a + (long long)blockIdx.x * mm * Cx<CPLX>::E;
  double s = 0, z = 0;
  for (int r = threadIdx.x; r < mm; r += RED_THREADS) {
    const double2 x = Cx<CPLX>::ld(col, r);
    s += x.x * x.x + x.y * x.y;
  }
  block_allsum2(s, z);
  if (threadIdx.x == 0) sig[blockIdx.x] = sqrt(s);
}

// dst[:, j] = src[:, perm[j]] / sig[perm[j]]  (zero when sig <= thresh): columns in descending-sigma
// order, so that numerically null columns come last and the completing QR has a diagonal R on the rest
template <bool CPLX>
__global__ void k_normalise_perm(double* dst, const double* __restrict__ src, int mm, int nn,
                                 const double* __restrict__ sig, const long long* __restrict__ perm, double thresh) {
  const long long total = (long long)mm * nn;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int j = (int)(t / mm), r = (int)(t % mm);
    const int pj = (int)perm[j];
    const double sg = sig[pj];
    double2 x = Cx<CPLX>::ld(src, r + (long long)pj * mm);
    if (sg > thresh) {
      x.x /= sg;
      x.y /= sg;
    } else {
      x = make_double2(0.0, 0.0);
    }
    Cx<CPLX>::st(dst, t, x);
  }
}

// scatter the factors of one block.  un: normalised, sigma-ordered and factored workspace (R_jj on its
// diagonal), q: completed isometry (mm x nn col-major, sigma-ordered), vm: V (nn x nn col-major, Jacobi order),
// perm: column order (descending sigma).
//   !herm: U[rows[r], koff+j] = q[r,j] * d_j ; Vt[koff+j, cols[c]] = conj(vm[c,pj])
//    herm: Vt[koff+j, cols[r]] = conj(q[r,j] * d_j) ; U[rows[c], koff+j] = vm[c,pj]
//   d_j = R_jj (unit modulus) for a regular column, 1 for a completed null column
template <bool CPLX>
__global__ void k_scatter_svd(double* U, double* Vt, const double* __restrict__ un, const double* __restrict__ q,
                              const double* __restrict__ vm, const long long* __restrict__ perm, long long K,
                              long long ncol, const long long* __restrict__ rows, const long long* __restrict__ cols,
                              int mm, int nn, long long koff, int herm) {
  // K is the leading dimension of U (number of columns of the output U)
  const long long totq = (long long)mm * nn, totv = (long long)nn * nn;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < totq + totv; t += stride) {
    if (t < totq) {
      int j, r;
      if (!herm) {
        j = (int)(t % nn);
        r = (int)(t / nn);
      } else {
        r = (int)(t % mm);
        j = (int)(t / mm);
      }
      double2 d = Cx<CPLX>::ld(un, j + (long long)j * mm);
      if (d.x == 0.0 && d.y == 0.0) d = make_double2(1.0, 0.0);
      const double2 x = Cx<CPLX>::ld(q, r + (long long)j * mm);
      double2 y = make_double2(x.x * d.x - x.y * d.y, x.x * d.y + x.y * d.x);
      if (!herm) {
        Cx<CPLX>::st(U, rows[r] * K + koff + j, y);
      } else {
        y.y = -y.y;
        Cx<CPLX>::st(Vt, (koff + j) * ncol + cols[r], y);
      }
    } else {
      const long long t2 = t - totq;
      int j, c;
      if (!herm) {
        c = (int)(t2 % nn);
        j = (int)(t2 / nn);
      } else {
        j = (int)(t2 % nn);
        c = (int)(t2 / nn);
      }
      const int pj = (int)perm[j];
      double2 x = Cx<CPLX>::ld(vm, c + (long long)pj * nn);
      if (!herm) {
        x.y = -x.y;
        Cx<CPLX>::st(Vt, (koff + j) * ncol + cols[c], x);
      } else {
        Cx<CPLX>::st(U, rows[c] * K + koff + j, x);
      }
    }
  }
}

// null-space vectors of the tall side: columns nn .. nn+extra-1 of the completed isometry q
//   !herm: U[rows[r], uoff+e] = q[r, nn+e]   ;   herm: Vt[voff+e, cols[r]] = conj(q[r, nn+e])
template <bool CPLX>
__global__ void k_scatter_null(double* U, double* Vt, const double* __restrict__ q, long long ldU, long long ncol,
                               const long long* __restrict__ rows, const long long* __restrict__ cols, int mm, int nn,
                               int extra, long long off, int herm) {
  const long long total = (long long)mm * extra;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    if (!herm) {
      const int e = (int)(t % extra), r = (int)(t / extra);
      Cx<CPLX>::st(U, rows[r] * ldU + off + e, Cx<CPLX>::ld(q, r + (long long)(nn + e) * mm));
    } else {
      const int r = (int)(t % mm), e = (int)(t / mm);
      double2 x = Cx<CPLX>::ld(q, r + (long long)(nn + e) * mm);
      x.y = -x.y;
      Cx<CPLX>::st(Vt, (off + e) * ncol + cols[r], x);
    }
  }
}

template <bool CPLX>
int block_svd_batched(mpse_ctx* ctx, const void* coef, int64_t nrow, int64_t ncol, int nblocks, const int64_t* row_idx,
                      const int64_t* row_off, const int64_t* col_idx, const int64_t* col_off, void* U, void* Vt,
                      double* S_host, int64_t K, const int64_t* extra_host, int64_t KU, int64_t KV);

template <bool CPLX>
int block_svd_impl(mpse_ctx* ctx, const void* coef, int64_t nrow, int64_t ncol, int nblocks, const int64_t* row_idx,
                   const int64_t* row_off, const int64_t* col_idx, const int64_t* col_off, void* U, void* Vt,
                   double* S_host, int64_t K, const int64_t* extra_host, int64_t KU, int64_t KV) {
  constexpr size_t es = CPLX ? 16 : 8;
  int64_t ktot = 0, maxws = 0, maxk = 0, maxq = 0, nu = 0, nv = 0, maxrows = 0;
  for (int b = 0; b < nblocks; ++b) {
    const int64_t m = row_off[b + 1] - row_off[b], n = col_off[b + 1] - col_off[b];
    if (m < 0 || n < 0) return mpse_fail(ctx, MPSE_ERR_SHAPE, "block_svd: negative block extent");
    const int64_t k = std::min(m, n);
    if (k > 0) maxrows = std::max(maxrows, std::max(m, n));
    ktot += k;
    maxws = std::max(maxws, m * n);
    maxk = std::max(maxk, k);
    int64_t ex = extra_host ? extra_host[b] : 0;
    if (ex < 0 || ex > std::max(m, n) - k) return mpse_fail(ctx, MPSE_ERR_SHAPE, "block_svd: bad extra count");
    if (k == 0) ex = 0;
    maxq = std::max(maxq, std::max(m, n) * (k + ex));
    if (m >= n) nu += ex; else nv += ex;
  }
  if (KU != ktot + nu || KV != ktot + nv)
    return mpse_fail(ctx, MPSE_ERR_SHAPE, "block_svd: KU/KV (%lld,%lld) do not match the blocks (%lld,%lld)",
                     (long long)KU, (long long)KV, (long long)(ktot + nu), (long long)(ktot + nv));
  if (ktot != K) return mpse_fail(ctx, MPSE_ERR_SHAPE, "block_svd: K=%lld but blocks give %lld", (long long)K, (long long)ktot);
  if (ktot == 0) return mpse_fail(ctx, MPSE_ERR_SHAPE, "Invalid quantum number");
  if (maxrows <= HH_BATCH_MAX_ROWS)   // every block fits the register-resident Householder kernels: batched path
    return block_svd_batched<CPLX>(ctx, coef, nrow, ncol, nblocks, row_idx, row_off, col_idx, col_off, U, Vt, S_host, K,
                                   extra_host, KU, KV);
  MPSE_TRY(mpse_memset_zero(ctx, U, size_t(nrow * KU) * es));
  MPSE_TRY(mpse_memset_zero(ctx, Vt, size_t(KV * ncol) * es));
  const int64_t nri = row_off[nblocks], nci = col_off[nblocks];
  long long uoff = K, voff = K;  // where the next block's null vectors go
  TmpBuf IDX(ctx), WS(ctx), Q(ctx), VM(ctx), PRM(ctx), SIG(ctx), PERM(ctx), CNT(ctx);
  MPSE_TRY(IDX.alloc(size_t(nri + nci) * 8));
  MPSE_TRY(WS.alloc(size_t(std::max(maxws, maxq)) * es));
  MPSE_TRY(Q.alloc(size_t(maxws) * es));
  MPSE_TRY(VM.alloc(size_t(maxk * maxk) * es));
  MPSE_TRY(PRM.alloc(size_t(maxk + 1) * sizeof(HhParam)));
  MPSE_TRY(SIG.alloc(size_t(maxk) * 8));
  MPSE_TRY(PERM.alloc(size_t(maxk) * 8));
  MPSE_TRY(CNT.alloc(64));
  MPSE_TRY(stage_h2d(ctx, IDX.p, row_idx, size_t(nri) * 8));
  MPSE_TRY(stage_h2d(ctx, IDX.as<char>() + size_t(nri) * 8, col_idx, size_t(nci) * 8));
  const long long* drows = IDX.as<long long>();
  const long long* dcols = IDX.as<long long>() + nri;
  std::vector<double> sig;
  std::vector<long long> perm;
  int64_t koff = 0;
  for (int b = 0; b < nblocks; ++b) {
    const int m = (int)(row_off[b + 1] - row_off[b]), n = (int)(col_off[b + 1] - col_off[b]);
    const int k = std::min(m, n);
    if (k == 0) continue;
    const int herm = m < n ? 1 : 0;
    const int mm = herm ? n : m, nn = herm ? m : n;  // mm >= nn == k
    const long long* rows = drows + row_off[b];
    const long long* cols = dcols + col_off[b];
    double* ws = WS.as<double>();
    double* vm = VM.as<double>();
    hipLaunchKernelGGL((k_gather_block<CPLX>), dim3(ew_blocks((int64_t)mm * nn)), dim3(256), 0, ctx->stream, ws,
                       (const double*)coef, (long long)ncol, rows, cols, mm, nn, herm);
    hipLaunchKernelGGL((k_set_identity<CPLX>), dim3(ew_blocks((int64_t)nn * nn)), dim3(256), 0, ctx->stream, vm, nn);
    if (nn > 1) {
      const int N = (nn + 1) & ~1;
      const double tol = 2.220446049250313e-16 * sqrt((double)mm);
      hipLaunchKernelGGL((k_col_norms<CPLX>), dim3(nn), dim3(RED_THREADS), 0, ctx->stream, (const double*)ws, mm,
                         SIG.as<double>());
      sig.resize(nn);
      MPSE_TRY(mpse_memcpy_d2h(ctx, sig.data(), SIG.p, size_t(nn) * 8));
      double cmax = 0.0;
      for (double x : sig) cmax = x > cmax ? x : cmax;
      const double nullnorm = cmax * 2.220446049250313e-16 * (double)mm;
      const double null2 = nullnorm * nullnorm;
      bool converged = false;
      for (int sweep = 0; sweep < 60 && !converged; ++sweep) {
        MPSE_HIP(ctx, hipMemsetAsync(CNT.p, 0, sizeof(int), ctx->stream));
        for (int step = 0; step < N - 1; ++step)
          hipLaunchKernelGGL((k_jacobi_step<CPLX>), dim3(N / 2), dim3(RED_THREADS), 0, ctx->stream, ws, vm, mm, nn, N,
                             step, tol, null2, CNT.as<int>());
        MPSE_HIP(ctx, hipGetLastError());
        MPSE_TRY(publish_and_wait(ctx, CNT.as<double>(), 1, 8));
        converged = (*reinterpret_cast<int*>(ctx->pinned + 8) == 0);
      }
      if (!converged) return mpse_fail(ctx, MPSE_ERR_NOCONV, "block_svd: Jacobi did not converge (block %d, %dx%d)", b, m, n);
    }
    hipLaunchKernelGGL((k_col_norms<CPLX>), dim3(nn), dim3(RED_THREADS), 0, ctx->stream, (const double*)ws, mm,
                       SIG.as<double>());
    sig.resize(nn);
    MPSE_TRY(mpse_memcpy_d2h(ctx, sig.data(), SIG.p, size_t(nn) * 8));
    perm.resize(nn);
    std::iota(perm.begin(), perm.end(), 0LL);
    std::stable_sort(perm.begin(), perm.end(), [&](long long x, long long y) { return sig[x] > sig[y]; });
    for (int j = 0; j < nn; ++j) S_host[koff + j] = sig[perm[j]];
    MPSE_TRY(stage_h2d(ctx, PERM.p, perm.data(), size_t(nn) * 8));
    const double smax = sig[perm[0]];
    const double thresh = smax * 2.220446049250313e-16 * (double)mm;
    double* un = Q.as<double>();
    hipLaunchKernelGGL((k_normalise_perm<CPLX>), dim3(ew_blocks((int64_t)mm * nn)), dim3(256), 0, ctx->stream, un,
                       (const double*)ws, mm, nn, SIG.as<const double>(), PERM.as<const long long>(), thresh);
    MPSE_TRY(hh_factor_colmajor(ctx, CPLX, un, mm, nn, nn, PRM.as<HhParam>()));
    const int extra = extra_host ? (int)extra_host[b] : 0;
    MPSE_TRY(hh_formq_colmajor(ctx, CPLX, ws, un, mm, nn, PRM.as<HhParam>(), nn + extra));
    hipLaunchKernelGGL((k_scatter_svd<CPLX>), dim3(ew_blocks((int64_t)mm * nn + (int64_t)nn * nn)), dim3(256), 0,
                       ctx->stream, (double*)U, (double*)Vt, (const double*)un, (const double*)ws,
                       (const double*)vm, PERM.as<const long long>(), (long long)KU, (long long)ncol, rows, cols, mm,
                       nn, (long long)koff, herm);
    if (extra > 0) {
      long long& off = herm ? voff : uoff;
      hipLaunchKernelGGL((k_scatter_null<CPLX>), dim3(ew_blocks((int64_t)mm * extra)), dim3(256), 0, ctx->stream,
                         (double*)U, (double*)Vt, (const double*)ws, (long long)KU, (long long)ncol, rows, cols, mm, nn,
                         extra, off, herm);
      off += extra;
    }
    MPSE_HIP(ctx, hipGetLastError());
    // PERM / SIG are rewritten for the next block only after this block's scatter: same stream, in order
    koff += k;
  }
  return MPSE_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Batched variant: all quantum-number blocks of a decomposition share every launch (blockIdx.y = block), the
// convergence of a sweep is read once for all blocks, column norms once for all blocks.  Same algorithm and
// thresholds as above.
struct SvdBlk {
  long long ws_off, v_off, q_off;   // element offsets: column-major block (mm x nn), V (nn x nn), completed Q (mm x nq)
  long long row_off, col_off;       // into the concatenated row / column index lists
  long long koff, null_off;         // where its singular triplets / null vectors go in U / Vt
  int mm, nn, N, herm, extra, sig_off;
  double tol;
};

template <bool CPLX>
__global__ void k_gather_blocks(double* ws, const double* __restrict__ coef, long long ncol,
                                const long long* __restrict__ rows_all, const long long* __restrict__ cols_all,
                                const SvdBlk* __restrict__ blks) {
  const SvdBlk B = blks[blockIdx.y];
  const long long* rows = rows_all + B.row_off;
  const long long* cols = cols_all + B.col_off;
  double* w = ws + B.ws_off * Cx<CPLX>::E;
  const int mm = B.mm, nn = B.nn;
  const long long total = (long long)mm * nn;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    if (!B.herm) {
      const int c = (int)(t % nn), r = (int)(t / nn);
      Cx<CPLX>::st(w, r + (long long)c * mm, Cx<CPLX>::ld(coef, rows[r] * ncol + cols[c]));
    } else {
      const int r = (int)(t % mm), c = (int)(t / mm);
      double2 v = Cx<CPLX>::ld(coef, rows[c] * ncol + cols[r]);
      v.y = -v.y;
      Cx<CPLX>::st(w, r + (long long)c * mm, v);
    }
  }
}

template <bool CPLX>
__global__ void k_identity_blocks(double* vbase, const SvdBlk* __restrict__ blks) {
  const SvdBlk B = blks[blockIdx.y];
  double* v = vbase + B.v_off * Cx<CPLX>::E;
  const long long total = (long long)B.nn * B.nn;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride)
    Cx<CPLX>::st(v, t, make_double2((t % B.nn) == (t / B.nn) ? 1.0 : 0.0, 0.0));
}

template <bool CPLX>
__global__ __launch_bounds__(RED_THREADS) void k_col_norms_b(const double* __restrict__ ws,
                                                             const SvdBlk* __restrict__ blks, double* sig) {
  const SvdBlk B = blks[blockIdx.y];
  if ((int)blockIdx.x >= B.nn) return;
  const double* col = ws + (B.ws_off + (long long)blockIdx.x * B.mm) * Cx<CPLX>::E;
  double s = 0, z = 0;
  for (int r = threadIdx.x; r < B.mm; r += RED_THREADS) {
    const double2 x = Cx<CPLX>::ld(col, r);
    s += x.x * x.x + x.y * x.y;
  }
  block_allsum2(s, z);
  if (threadIdx.x == 0) sig[B.sig_off + blockIdx.x] = sqrt(s);
}

// null2[b] = (largest column norm * eps * mm)^2 ; also resets the sweep state of the block
__global__ void k_null2(const SvdBlk* __restrict__ blks, const double* __restrict__ sig, double* null2, int* nrot,
                        int* done) {
  const SvdBlk B = blks[blockIdx.x];
  double m = 0.0;
  for (int j = threadIdx.x; j < B.nn; j += 64) m = fmax(m, sig[B.sig_off + j]);
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
  if (threadIdx.x == 0) {
    const double nn = m * 2.220446049250313e-16 * (double)B.mm;
    null2[blockIdx.x] = nn * nn;
    nrot[blockIdx.x] = 0;
    done[blockIdx.x] = B.nn > 1 ? 0 : 1;
  }
}

// after a sweep: a block without rotations is converged; counters restart
__global__ void k_sweep_end(int nblk, int* nrot, int* done) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblk) return;
  if (nrot[b] == 0) done[b] = 1;
  nrot[b] = 0;
}

template <bool CPLX>
__global__ __launch_bounds__(RED_THREADS) void k_jacobi_step_b(double* ws, double* vbase,
                                                               const SvdBlk* __restrict__ blks, int step,
                                                               const double* __restrict__ null2v, int* nrot,
                                                               const int* __restrict__ done) {
  constexpr int E = Cx<CPLX>::E;
  const int b = blockIdx.y;
  if (done[b]) return;
  const SvdBlk B = blks[b];
  const int N = B.N, nn = B.nn, mm = B.mm;
  const int kk = blockIdx.x;
  if (step >= N - 1 || kk >= N / 2) return;
  int p, q;
  if (kk == 0) {
    p = step % (N - 1);
    q = N - 1;
  } else {
    p = (step + kk) % (N - 1);
    q = (step - kk + (N - 1)) % (N - 1);
  }
  if (p > q) {
    const int t = p;
    p = q;
    q = t;
  }
  if (q >= nn) return;
  double* a = ws + B.ws_off * E;
  double* v = vbase + B.v_off * E;
  double* ap = a + (long long)p * mm * E;
  double* aq = a + (long long)q * mm * E;
  double alpha = 0, beta = 0, gr = 0, gi = 0;
  for (int r = threadIdx.x; r < mm; r += RED_THREADS) {
    const double2 x = Cx<CPLX>::ld(ap, r), y = Cx<CPLX>::ld(aq, r);
    alpha += x.x * x.x + x.y * x.y;
    beta += y.x * y.x + y.y * y.y;
    gr += x.x * y.x + x.y * y.y;
    gi += x.x * y.y - x.y * y.x;
  }
  block_allsum2(alpha, beta);
  block_allsum2(gr, gi);
  const double g = sqrt(gr * gr + gi * gi);
  const double null2 = null2v[b];
  if (g == 0.0 || alpha <= null2 || beta <= null2) return;
  if (!(g > B.tol * sqrt(alpha) * sqrt(beta))) return;
  const double pr = gr / g, pi = gi / g;
  const double zeta = (beta - alpha) / (2.0 * g);
  const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
  const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
  for (int r = threadIdx.x; r < mm; r += RED_THREADS) {
    const double2 x = Cx<CPLX>::ld(ap, r), y0 = Cx<CPLX>::ld(aq, r);
    const double2 y = make_double2(y0.x * pr + y0.y * pi, y0.y * pr - y0.x * pi);
    Cx<CPLX>::st(ap, r, make_double2(c * x.x - s * y.x, c * x.y - s * y.y));
    Cx<CPLX>::st(aq, r, make_double2(s * x.x + c * y.x, s * x.y + c * y.y));
  }
  double* vp = v + (long long)p * nn * E;
  double* vq = v + (long long)q * nn * E;
  for (int r = threadIdx.x; r < nn; r += RED_THREADS) {
    const double2 x = Cx<CPLX>::ld(vp, r), y0 = Cx<CPLX>::ld(vq, r);
    const double2 y = make_double2(y0.x * pr + y0.y * pi, y0.y * pr - y0.x * pi);
    Cx<CPLX>::st(vp, r, make_double2(c * x.x - s * y.x, c * x.y - s * y.y));
    Cx<CPLX>::st(vq, r, make_double2(s * x.x + c * y.x, s * x.y + c * y.y));
  }
  if (threadIdx.x == 0) atomicAdd(nrot + b, 1);
}

template <bool CPLX>
__global__ void k_normalise_perm_b(double* un_base, const double* __restrict__ ws, const SvdBlk* __restrict__ blks,
                                   const double* __restrict__ sig, const long long* __restrict__ perm,
                                   const double* __restrict__ thresh) {
  const SvdBlk B = blks[blockIdx.y];
  const int mm = B.mm, nn = B.nn;
  double* dst = un_base + B.ws_off * Cx<CPLX>::E;
  const double* src = ws + B.ws_off * Cx<CPLX>::E;
  const double th = thresh[blockIdx.y];
  const long long total = (long long)mm * nn;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int j = (int)(t / mm), r = (int)(t % mm);
    const int pj = (int)perm[B.sig_off + j];
    const double sg = sig[B.sig_off + pj];
    double2 x = Cx<CPLX>::ld(src, r + (long long)pj * mm);
    if (sg > th) {
      x.x /= sg;
      x.y /= sg;
    } else {
      x = make_double2(0.0, 0.0);
    }
    Cx<CPLX>::st(dst, t, x);
  }
}

template <bool CPLX>
__global__ void k_scatter_svd_b(double* U, double* Vt, const double* __restrict__ un_base,
                                const double* __restrict__ q_base, const double* __restrict__ v_base,
                                const long long* __restrict__ perm_all, long long K, long long ncol,
                                const long long* __restrict__ rows_all, const long long* __restrict__ cols_all,
                                const SvdBlk* __restrict__ blks) {
  constexpr int E = Cx<CPLX>::E;
  const SvdBlk B = blks[blockIdx.y];
  const int mm = B.mm, nn = B.nn, herm = B.herm;
  const double* un = un_base + B.ws_off * E;
  const double* q = q_base + B.q_off * E;
  const double* vm = v_base + B.v_off * E;
  const long long* perm = perm_all + B.sig_off;
  const long long* rows = rows_all + B.row_off;
  const long long* cols = cols_all + B.col_off;
  const long long koff = B.koff;
  const long long totq = (long long)mm * nn, totv = (long long)nn * nn;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < totq + totv; t += stride) {
    if (t < totq) {
      int j, r;
      if (!herm) {
        j = (int)(t % nn);
        r = (int)(t / nn);
      } else {
        r = (int)(t % mm);
        j = (int)(t / mm);
      }
      double2 d = Cx<CPLX>::ld(un, j + (long long)j * mm);
      if (d.x == 0.0 && d.y == 0.0) d = make_double2(1.0, 0.0);
      const double2 x = Cx<CPLX>::ld(q, r + (long long)j * mm);
      double2 y = make_double2(x.x * d.x - x.y * d.y, x.x * d.y + x.y * d.x);
      if (!herm) {
        Cx<CPLX>::st(U, rows[r] * K + koff + j, y);
      } else {
        y.y = -y.y;
        Cx<CPLX>::st(Vt, (koff + j) * ncol + cols[r], y);
      }
    } else {
      const long long t2 = t - totq;
      int j, c;
      if (!herm) {
        c = (int)(t2 % nn);
        j = (int)(t2 / nn);
      } else {
        j = (int)(t2 % nn);
        c = (int)(t2 / nn);
      }
      const int pj = (int)perm[j];
      double2 x = Cx<CPLX>::ld(vm, c + (long long)pj * nn);
      if (!herm) {
        x.y = -x.y;
        Cx<CPLX>::st(Vt, (koff + j) * ncol + cols[c], x);
      } else {
        Cx<CPLX>::st(U, rows[c] * K + koff + j, x);
      }
    }
  }
}

template <bool CPLX>
__global__ void k_scatter_null_b(double* U, double* Vt, const double* __restrict__ q_base, long long ldU,
                                 long long ncol, const long long* __restrict__ rows_all,
                                 const long long* __restrict__ cols_all, const SvdBlk* __restrict__ blks) {
  const SvdBlk B = blks[blockIdx.y];
  if (B.extra <= 0) return;
  const int mm = B.mm, nn = B.nn, extra = B.extra;
  const double* q = q_base + B.q_off * Cx<CPLX>::E;
  const long long* rows = rows_all + B.row_off;
  const long long* cols = cols_all + B.col_off;
  const long long total = (long long)mm * extra;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    if (!B.herm) {
      const int e = (int)(t % extra), r = (int)(t / extra);
      Cx<CPLX>::st(U, rows[r] * ldU + B.null_off + e, Cx<CPLX>::ld(q, r + (long long)(nn + e) * mm));
    } else {
      const int r = (int)(t % mm), e = (int)(t / mm);
      double2 x = Cx<CPLX>::ld(q, r + (long long)(nn + e) * mm);
      x.y = -x.y;
      Cx<CPLX>::st(Vt, (B.null_off + e) * ncol + cols[r], x);
    }
  }
}

template <bool CPLX>
int block_svd_batched(mpse_ctx* ctx, const void* coef, int64_t nrow, int64_t ncol, int nblocks, const int64_t* row_idx,
                      const int64_t* row_off, const int64_t* col_idx, const int64_t* col_off, void* U, void* Vt,
                      double* S_host, int64_t K, const int64_t* extra_host, int64_t KU, int64_t KV) {
  constexpr size_t es = CPLX ? 16 : 8;
  constexpr int E = CPLX ? 2 : 1;
  std::vector<SvdBlk> blks;
  long long ws_tot = 0, v_tot = 0, q_tot = 0, sig_tot = 0, koff = 0, uoff = K, voff = K;
  int maxN = 0, maxnn = 0;
  long long max_elems = 1;
  for (int b = 0; b < nblocks; ++b) {
    const int m = (int)(row_off[b + 1] - row_off[b]), n = (int)(col_off[b + 1] - col_off[b]);
    const int k = std::min(m, n);
    if (k == 0) continue;
    SvdBlk B;
    B.herm = m < n ? 1 : 0;
    B.mm = B.herm ? n : m;
    B.nn = B.herm ? m : n;
    B.N = (B.nn + 1) & ~1;
    B.extra = extra_host ? (int)extra_host[b] : 0;
    B.ws_off = ws_tot, B.v_off = v_tot, B.q_off = q_tot, B.sig_off = (int)sig_tot;
    B.row_off = row_off[b], B.col_off = col_off[b];
    B.koff = koff;
    B.null_off = B.herm ? voff : uoff;
    (B.herm ? voff : uoff) += B.extra;
    B.tol = 2.220446049250313e-16 * std::sqrt((double)B.mm);
    ws_tot += (long long)B.mm * B.nn;
    v_tot += (long long)B.nn * B.nn;
    q_tot += (long long)B.mm * (B.nn + B.extra);
    sig_tot += B.nn;
    koff += k;
    maxN = std::max(maxN, B.N);
    maxnn = std::max(maxnn, B.nn);
    max_elems = std::max(max_elems, (long long)B.mm * (B.nn + B.extra) + (long long)B.nn * B.nn);
    blks.push_back(B);
  }
  const int nblk = (int)blks.size();
  // whole-call timing for bench.py (variant 6): the algorithmic bytes depend on the sweep count, filled in below
  mpse_ctx::ProfRec srec;
  const bool spt = prof_begin(ctx, 6, 0.0, 0.0, &srec);
  int sweeps_done = 0;
  MPSE_TRY(mpse_memset_zero(ctx, U, size_t(nrow * KU) * es));
  MPSE_TRY(mpse_memset_zero(ctx, Vt, size_t(KV * ncol) * es));
  const int64_t nri = row_off[nblocks], nci = col_off[nblocks];
  TmpBuf IDX(ctx), DB(ctx), WS(ctx), UN(ctx), Q(ctx), VM(ctx), PRM(ctx), SIG(ctx), PERM(ctx), ST(ctx);
  MPSE_TRY(IDX.alloc(size_t(nri + nci) * 8));
  MPSE_TRY(DB.alloc(size_t(nblk) * sizeof(SvdBlk)));
  MPSE_TRY(WS.alloc(size_t(ws_tot) * es));
  MPSE_TRY(UN.alloc(size_t(ws_tot) * es));
  MPSE_TRY(Q.alloc(size_t(q_tot) * es));
  MPSE_TRY(VM.alloc(size_t(v_tot) * es));
  MPSE_TRY(PRM.alloc(size_t(sig_tot + 1) * sizeof(HhParam)));
  MPSE_TRY(SIG.alloc(size_t(sig_tot) * 8));
  MPSE_TRY(PERM.alloc(size_t(sig_tot) * 8));
  // per block: null2 (double), thresh (double), nrot (int), done (int)
  const int nblk_pad = (nblk + 1) & ~1;     // `done` is read back as doubles: keep it 8-byte aligned
  MPSE_TRY(ST.alloc(size_t(nblk_pad) * (8 + 8 + 4 + 4) + 64));
  double* null2 = ST.as<double>();
  double* thresh = null2 + nblk_pad;
  int* nrot = reinterpret_cast<int*>(thresh + nblk_pad);
  int* done = nrot + nblk_pad;
  MPSE_TRY(stage_h2d(ctx, IDX.p, row_idx, size_t(nri) * 8));
  MPSE_TRY(stage_h2d(ctx, IDX.as<char>() + size_t(nri) * 8, col_idx, size_t(nci) * 8));
  MPSE_TRY(stage_h2d(ctx, DB.p, blks.data(), size_t(nblk) * sizeof(SvdBlk)));
  const long long* drows = IDX.as<long long>();
  const long long* dcols = IDX.as<long long>() + nri;
  const SvdBlk* dblk = DB.as<SvdBlk>();
  double* ws = WS.as<double>();
  double* vm = VM.as<double>();
  double* sigd = SIG.as<double>();
  int gx = ew_blocks(max_elems);
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL((k_gather_blocks<CPLX>), dim3(gx, nblk), dim3(256), 0, ctx->stream, ws, (const double*)coef,
                     (long long)ncol, drows, dcols, dblk);
  hipLaunchKernelGGL((k_identity_blocks<CPLX>), dim3(gx, nblk), dim3(256), 0, ctx->stream, vm, dblk);
  hipLaunchKernelGGL((k_col_norms_b<CPLX>), dim3(maxnn, nblk), dim3(RED_THREADS), 0, ctx->stream, (const double*)ws,
                     dblk, sigd);
  hipLaunchKernelGGL(k_null2, dim3(nblk), dim3(64), 0, ctx->stream, dblk, (const double*)sigd, null2, nrot, done);
  MPSE_HIP(ctx, hipGetLastError());
  if (maxN > 1) {
    std::vector<int> hdone(nblk, 0);
    bool all = false;
    for (int sweep = 0; sweep < 60 && !all; ++sweep) {
      for (int step = 0; step < maxN - 1; ++step)
        hipLaunchKernelGGL((k_jacobi_step_b<CPLX>), dim3(maxN / 2, nblk), dim3(RED_THREADS), 0, ctx->stream, ws, vm,
                           dblk, step, (const double*)null2, nrot, (const int*)done);
      hipLaunchKernelGGL(k_sweep_end, dim3((nblk + 255) / 256), dim3(256), 0, ctx->stream, nblk, nrot, done);
      MPSE_HIP(ctx, hipGetLastError());
      // one read-back per sweep for all blocks
      if (nblk <= 2000) {
        MPSE_TRY(publish_and_wait(ctx, reinterpret_cast<const double*>(done), (nblk + 1) / 2, 8));
        memcpy(hdone.data(), ctx->pinned + 8, size_t(nblk) * sizeof(int));
      } else {
        MPSE_TRY(mpse_memcpy_d2h(ctx, hdone.data(), done, size_t(nblk) * sizeof(int)));
      }
      all = true;
      for (int b = 0; b < nblk; ++b) all = all && hdone[b];
      ++sweeps_done;
    }
    if (!all) return mpse_fail(ctx, MPSE_ERR_NOCONV, "block_svd: Jacobi did not converge within 60 sweeps");
  }
  hipLaunchKernelGGL((k_col_norms_b<CPLX>), dim3(maxnn, nblk), dim3(RED_THREADS), 0, ctx->stream, (const double*)ws,
                     dblk, sigd);
  std::vector<double> sig((size_t)sig_tot), th((size_t)nblk);
  MPSE_TRY(mpse_memcpy_d2h(ctx, sig.data(), sigd, size_t(sig_tot) * 8));
  std::vector<long long> perm((size_t)sig_tot);
  for (int b = 0; b < nblk; ++b) {
    const SvdBlk& B = blks[b];
    long long* p = perm.data() + B.sig_off;
    const double* sg = sig.data() + B.sig_off;
    std::iota(p, p + B.nn, 0LL);
    std::stable_sort(p, p + B.nn, [&](long long x, long long y) { return sg[x] > sg[y]; });
    for (int j = 0; j < B.nn; ++j) S_host[B.koff + j] = sg[p[j]];
    th[b] = sg[p[0]] * 2.220446049250313e-16 * (double)B.mm;
  }
  MPSE_TRY(stage_h2d(ctx, PERM.p, perm.data(), size_t(sig_tot) * 8));
  MPSE_TRY(stage_h2d(ctx, thresh, th.data(), size_t(nblk) * 8));
  double* un = UN.as<double>();
  hipLaunchKernelGGL((k_normalise_perm_b<CPLX>), dim3(gx, nblk), dim3(256), 0, ctx->stream, un, (const double*)ws, dblk,
                     (const double*)sigd, PERM.as<const long long>(), (const double*)thresh);
  MPSE_HIP(ctx, hipGetLastError());
  // Householder completion of every block's normalised left factor in the same launches
  std::vector<QrBlk> qb((size_t)nblk);
  for (int b = 0; b < nblk; ++b) {
    const SvdBlk& B = blks[b];
    qb[b] = QrBlk{B.ws_off, B.q_off, B.mm, B.nn, B.nn, B.sig_off, B.nn + B.extra};
  }
  MPSE_TRY(hh_qr_batched(ctx, CPLX, un, Q.as<double>(), PRM.as<HhParam>(), qb.data(), nblk, true));
  hipLaunchKernelGGL((k_scatter_svd_b<CPLX>), dim3(gx, nblk), dim3(256), 0, ctx->stream, (double*)U, (double*)Vt,
                     (const double*)un, (const double*)Q.as<double>(), (const double*)vm, PERM.as<const long long>(),
                     (long long)KU, (long long)ncol, drows, dcols, dblk);
  if (uoff > K || voff > K)
    hipLaunchKernelGGL((k_scatter_null_b<CPLX>), dim3(gx, nblk), dim3(256), 0, ctx->stream, (double*)U, (double*)Vt,
                       (const double*)Q.as<double>(), (long long)KU, (long long)ncol, drows, dcols, dblk);
  MPSE_HIP(ctx, hipGetLastError());
  if (spt) {
    // one sweep = nn (nn - 1) / 2 pairs; a pair reads its two columns of A (Gram entries), reads and writes them again
    // (rotation), and does the same on the nn-row columns of V: 3 passes over 2 (mm + nn) elements.  Rotations that are
    // skipped (converged pairs, null columns) make the real traffic smaller: an upper bound, like the flops
    // (dots 3 x 8 + rotation 2 x 12 real flops per complex row pair).
    double bytes = 0.0, flops = 0.0;
    for (const SvdBlk& B : blks) {
      const double pairs = 0.5 * B.nn * (B.nn - 1.0) * sweeps_done, rows = double(B.mm) + B.nn;
      bytes += pairs * 3.0 * 2.0 * rows * double(es);
      flops += pairs * rows * (CPLX ? 48.0 : 12.0);
    }
    srec.bytes = bytes;
    srec.flops = flops;
    ctx->prof_svd_sweeps += sweeps_done;
    prof_end(ctx, srec);
  }
  (void)E;
  return MPSE_OK;
}

}  // namespace

extern "C" int mpse_block_svd_full(mpse_ctx* ctx, int dtype, const void* coef, int64_t nrow, int64_t ncol, int nblocks,
                                   const int64_t* row_idx_host, const int64_t* row_off_host,
                                   const int64_t* col_idx_host, const int64_t* col_off_host,
                                   const int64_t* extra_host, void* U, int64_t KU, void* Vt, int64_t KV,
                                   double* S_host, int64_t K) {
  if (!ctx || !coef || !U || !Vt || !S_host || !row_idx_host || !row_off_host || !col_idx_host || !col_off_host)
    return MPSE_ERR_ARG;
  MPSE_BIND(ctx);
  if (nblocks <= 0) return mpse_fail(ctx, MPSE_ERR_SHAPE, "Invalid quantum number");
  if (dtype == MPSE_C128)
    return block_svd_impl<true>(ctx, coef, nrow, ncol, nblocks, row_idx_host, row_off_host, col_idx_host,
                                col_off_host, U, Vt, S_host, K, extra_host, KU, KV);
  if (dtype == MPSE_F64)
    return block_svd_impl<false>(ctx, coef, nrow, ncol, nblocks, row_idx_host, row_off_host, col_idx_host,
                                 col_off_host, U, Vt, S_host, K, extra_host, KU, KV);
  return mpse_fail(ctx, MPSE_ERR_ARG, "block_svd: unknown dtype");
}

extern "C" int mpse_block_svd(mpse_ctx* ctx, int dtype, const void* coef, int64_t nrow, int64_t ncol, int nblocks,
                              const int64_t* row_idx_host, const int64_t* row_off_host, const int64_t* col_idx_host,
                              const int64_t* col_off_host, void* U, void* Vt, double* S_host, int64_t K) {
  return mpse_block_svd_full(ctx, dtype, coef, nrow, ncol, nblocks, row_idx_host, row_off_host, col_idx_host,
                             col_off_host, nullptr, U, K, Vt, K, S_host, K);
}
