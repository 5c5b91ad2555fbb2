// Quantum-number blocked QR / RQ on the device (replaces scipy.linalg.qr / rq per block in
// mps/svd_qn.py:177-213 plus the blockrecover scatter, :89-96).
//
// Householder reflections (LAPACK ?geqr2/?ung2r conventions), because the centre matrices of
// a fixed-bond TDVP sweep are numerically rank deficient most of the time: Gram-matrix
// (Cholesky) schemes break down there, Householder always returns an exact isometry.
// Layout: each block is gathered into a column-major workspace so that the column-wise
// reductions are coalesced; one workgroup owns one column.
//   factorisation : one launch per reflector j - workgroup c applies H_j^H to column c > j
//                   (dot + update, column stays in L2) and the owner of column j+1 then
//                   derives the next reflector's parameters in the same launch;
//   Q formation   : ONE launch - column c of Q = H_0 ... H_c e_c is independent of all other
//                   columns, so workgroup c applies its reflectors back to back.
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

__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 cmulc(double2 a, double2 b) {  // conj(a) * b
  return make_double2(a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x);
}

// reflector parameters for column j from its current content (rows >= j); one workgroup.
template <bool CPLX>
__device__ void hh_make_params(double* a, int mm, int j, HhParam* prm) {
  double* col = a + (long long)j * mm * Cx<CPLX>::E;
  double s = 0, z = 0;
  for (int r = j + 1 + threadIdx.x; r < mm; r += RED_THREADS) {
    const double2 v = Cx<CPLX>::ld(col, r);
    s += v.x * v.x + v.y * v.y;
  }
  block_allsum2(s, z);
  if (threadIdx.x == 0) {
    const double2 alpha = Cx<CPLX>::ld(col, j);
    HhParam p;
    if (s == 0.0 && alpha.y == 0.0) {
      p.tau_re = p.tau_im = p.scale_re = p.scale_im = 0.0;  // H = I
    } else {
      const double nrm = sqrt(alpha.x * alpha.x + alpha.y * alpha.y + s);
      const double beta = alpha.x >= 0.0 ? -nrm : nrm;
      p.tau_re = (beta - alpha.x) / beta;
      p.tau_im = -alpha.y / beta;
      const double dr = alpha.x - beta, di = alpha.y;  // scale = 1 / (alpha - beta)
      const double den = dr * dr + di * di;
      p.scale_re = dr / den;
      p.scale_im = -di / den;
      Cx<CPLX>::st(col, j, make_double2(beta, 0.0));
    }
    prm[j] = p;
  }
}

template <bool CPLX>
__global__ __launch_bounds__(RED_THREADS) void k_hh_first(double* a, int mm, HhParam* prm) {
  hh_make_params<CPLX>(a, mm, 0, prm);
}

// apply H_j^H to column c = j + 1 + blockIdx.x ; then (c == j+1) derive reflector j+1
template <bool CPLX>
__global__ __launch_bounds__(RED_THREADS) void k_hh_apply(double* a, int mm, int kref, int j, HhParam* prm) {
  constexpr int E = Cx<CPLX>::E;
  const int c = j + 1 + blockIdx.x;
  const double* vj = a + (long long)j * mm * E;
  double* col = a + (long long)c * mm * E;
  const HhParam p = prm[j];
  const double2 tau = make_double2(p.tau_re, p.tau_im), scale = make_double2(p.scale_re, p.scale_im);
  if (tau.x != 0.0 || tau.y != 0.0) {
    double dr = 0, di = 0;
    for (int r = j + 1 + threadIdx.x; r < mm; r += RED_THREADS) {
      const double2 t = cmulc(Cx<CPLX>::ld(vj, r), Cx<CPLX>::ld(col, r));
      dr += t.x;
      di += t.y;
    }
    block_allsum2(dr, di);
    // dot = v^H col = col[j] + conj(scale) * sum conj(tail) col
    const double2 head = Cx<CPLX>::ld(col, j);
    const double2 sc = cmulc(scale, make_double2(dr, di));
    const double2 dot = make_double2(head.x + sc.x, head.y + sc.y);
    const double2 f = cmulc(tau, dot);          // conj(tau) * dot
    const double2 fs = cmul(f, scale);
    __syncthreads();  // everyone has read col[j] before thread 0 overwrites it
    for (int r = j + 1 + threadIdx.x; r < mm; r += RED_THREADS) {
      const double2 t = cmul(fs, Cx<CPLX>::ld(vj, r));
      double2 x = Cx<CPLX>::ld(col, r);
      x.x -= t.x;
      x.y -= t.y;
      Cx<CPLX>::st(col, r, x);
    }
    if (threadIdx.x == 0) Cx<CPLX>::st(col, j, make_double2(head.x - f.x, head.y - f.y));
  }
  if (blockIdx.x == 0 && j + 1 < kref) {
    __syncthreads();
    hh_make_params<CPLX>(a, mm, j + 1, prm);
  }
}

// column c of Q (mm x k, column-major) = H_0 H_1 ... H_c e_c
template <bool CPLX>
__global__ __launch_bounds__(RED_THREADS) void k_hh_formq(double* q, const double* __restrict__ a, int mm, int kref,
                                                          const HhParam* __restrict__ prm) {
  constexpr int E = Cx<CPLX>::E;
  const int c = blockIdx.x;
  double* col = q + (long long)c * mm * E;
  for (int r = threadIdx.x; r < mm; r += RED_THREADS) Cx<CPLX>::st(col, r, make_double2(r == c ? 1.0 : 0.0, 0.0));
  __syncthreads();
  // columns c >= kref (orthogonal complement, used for full_matrices SVD) receive all reflectors
  for (int j = (c < kref ? c : kref - 1); j >= 0; --j) {
    const HhParam p = prm[j];
    const double2 tau = make_double2(p.tau_re, p.tau_im), scale = make_double2(p.scale_re, p.scale_im);
    if (tau.x == 0.0 && tau.y == 0.0) continue;
    const double* vj = a + (long long)j * mm * E;
    double dr = 0, di = 0;
    for (int r = j + 1 + threadIdx.x; r < mm; r += RED_THREADS) {
      const double2 t = cmulc(Cx<CPLX>::ld(vj, r), Cx<CPLX>::ld(col, r));
      dr += t.x;
      di += t.y;
    }
    block_allsum2(dr, di);
    const double2 head = Cx<CPLX>::ld(col, j);
    const double2 sc = cmulc(scale, make_double2(dr, di));
    const double2 dot = make_double2(head.x + sc.x, head.y + sc.y);
    const double2 f = cmul(tau, dot);  // H (not H^H)
    const double2 fs = cmul(f, scale);
    __syncthreads();  // everyone has read col[j]
    for (int r = j + 1 + threadIdx.x; r < mm; r += RED_THREADS) {
      const double2 t = cmul(fs, Cx<CPLX>::ld(vj, r));
      double2 x = Cx<CPLX>::ld(col, r);
      x.x -= t.x;
      x.y -= t.y;
      Cx<CPLX>::st(col, r, x);
    }
    if (threadIdx.x == 0) Cx<CPLX>::st(col, j, make_double2(head.x - f.x, head.y - f.y));
    __syncthreads();
  }
}

// gather the blocks into their column-major workspaces, ws[r + c*mm] - all blocks of a decomposition in one launch
// (blockIdx.y = block, descriptors and index lists on the device):
//   !herm: ws[r,c] = coef[rows[r], cols[c]]          (mm = #rows)
//    herm: ws[r,c] = conj(coef[rows[c], cols[r]])    (mm = #cols)  -> QR of the adjoint gives RQ
template <bool CPLX>
__global__ void k_gather_blocks(double* ws_base, const double* __restrict__ coef, long long ncol,
                                const long long* __restrict__ drows, const long long* __restrict__ dcols,
                                const QrBlk* __restrict__ blks, int herm) {
  const QrBlk B = blks[blockIdx.y];
  double* ws = ws_base + B.ws_off * Cx<CPLX>::E;
  const long long* rows = drows + B.row_off;
  const long long* cols = dcols + B.col_off;
  const int mm = B.mm, nn = B.nn;
  const long long total = (long long)mm * nn;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    int r, c;
    if (!herm) {
      c = (int)(t % nn);
      r = (int)(t / nn);
      Cx<CPLX>::st(ws, r + (long long)c * mm, Cx<CPLX>::ld(coef, rows[r] * ncol + cols[c]));
    } else {
      r = (int)(t % mm);
      c = (int)(t / mm);
      double2 v = Cx<CPLX>::ld(coef, rows[c] * ncol + cols[r]);
      v.y = -v.y;
      Cx<CPLX>::st(ws, r + (long long)c * mm, v);
    }
  }
}

// scatter Q (mm x k col-major) and R (upper triangle of the workspace, k x nn) of every block to U (nrow x K) /
// Vt (K x ncol), one launch:
//   !herm: U[rows[r], koff+c] = Q[r,c] ; Vt[koff+i, cols[c]] = R[i,c]
//    herm: Vt[koff+c, cols[r]] = conj(Q[r,c]) ; U[rows[c], koff+i] = conj(R[i,c])
template <bool CPLX>
__global__ void k_scatter_blocks(double* U, double* Vt, const double* __restrict__ q_base,
                                 const double* __restrict__ ws_base, long long K, long long ncol,
                                 const long long* __restrict__ drows, const long long* __restrict__ dcols,
                                 const QrBlk* __restrict__ blks, int herm) {
  const QrBlk B = blks[blockIdx.y];
  const double* q = q_base + B.q_off * Cx<CPLX>::E;
  const double* a = ws_base + B.ws_off * Cx<CPLX>::E;
  const long long* rows = drows + B.row_off;
  const long long* cols = dcols + B.col_off;
  const int mm = B.mm, nn = B.nn, k = B.k;
  const long long koff = B.prm_off;
  const long long tq = (long long)mm * k, total = tq + (long long)k * nn;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    if (t < tq) {
      if (!herm) {
        const int c = (int)(t % k), r = (int)(t / k);
        Cx<CPLX>::st(U, rows[r] * K + koff + c, Cx<CPLX>::ld(q, r + (long long)c * mm));
      } else {
        const int r = (int)(t % mm), c = (int)(t / mm);
        double2 v = Cx<CPLX>::ld(q, r + (long long)c * mm);
        v.y = -v.y;
        Cx<CPLX>::st(Vt, (koff + c) * ncol + cols[r], v);
      }
    } else {
      const long long u = t - tq;
      const int c = (int)(u % nn), i = (int)(u / nn);
      double2 v = (i <= c) ? Cx<CPLX>::ld(a, i + (long long)c * mm) : make_double2(0.0, 0.0);
      if (!herm) {
        Cx<CPLX>::st(Vt, (koff + i) * ncol + cols[c], v);
      } else {
        v.y = -v.y;
        Cx<CPLX>::st(U, rows[c] * K + koff + i, v);
      }
    }
  }
}

inline int ew_blocks(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

// Householder factorisation of a column-major mm x nn workspace in place (k = min(mm,nn)
// reflectors; R ends up in the upper triangle, reflector tails below the diagonal).
int hh_factor_colmajor(mpse_ctx* ctx, bool cplx, double* ws, int mm, int nn, int k, HhParam* prm) {
  if (k <= 0) return MPSE_OK;
  if (mm <= HH_BATCH_MAX_ROWS) {
    QrBlk b{0, 0, mm, nn, k, 0};
    return hh_qr_batched(ctx, cplx, ws, nullptr, prm, &b, 1, false);
  }
  if (cplx)
    hipLaunchKernelGGL((k_hh_first<true>), dim3(1), dim3(RED_THREADS), 0, ctx->stream, ws, mm, prm);
  else
    hipLaunchKernelGGL((k_hh_first<false>), dim3(1), dim3(RED_THREADS), 0, ctx->stream, ws, mm, prm);
  for (int j = 0; j < k; ++j) {
    const int ncols_right = nn - j - 1;
    if (ncols_right <= 0) break;
    if (cplx)
      hipLaunchKernelGGL((k_hh_apply<true>), dim3(ncols_right), dim3(RED_THREADS), 0, ctx->stream, ws, mm, k, j, prm);
    else
      hipLaunchKernelGGL((k_hh_apply<false>), dim3(ncols_right), dim3(RED_THREADS), 0, ctx->stream, ws, mm, k, j, prm);
  }
  MPSE_HIP(ctx, hipGetLastError());
  return MPSE_OK;
}

int qr_words(mpse_ctx* ctx) {
  if (ctx->qr_words_dev) return MPSE_OK;
  void* p = nullptr;
  MPSE_TRY(mpse_malloc(ctx, 16, &p));
  ctx->qr_words_dev = static_cast<int*>(p);
  return device_zero(ctx, ctx->qr_words_dev, 16);
}

// explicit Q (mm x k, column-major) from a factored workspace
int hh_formq_colmajor(mpse_ctx* ctx, bool cplx, double* q, const double* ws, int mm, int k, const HhParam* prm,
                      int nq) {
  if (k <= 0) return MPSE_OK;
  if (nq < k) nq = k;
  if (cplx)
    hipLaunchKernelGGL((k_hh_formq<true>), dim3(nq), dim3(RED_THREADS), 0, ctx->stream, q, ws, mm, k, prm);
  else
    hipLaunchKernelGGL((k_hh_formq<false>), dim3(nq), dim3(RED_THREADS), 0, ctx->stream, q, ws, mm, k, prm);
  MPSE_HIP(ctx, hipGetLastError());
  return MPSE_OK;
}

namespace {

template <bool CPLX>
int block_qr_impl(mpse_ctx* ctx, const void* coef, int64_t nrow, int64_t ncol, int nblocks, const int64_t* row_idx,
                  const int64_t* row_off, const int64_t* col_idx, const int64_t* col_off, int herm, bool hh_only, void* U,
                  void* Vt,
                  int64_t K) {
  constexpr size_t es = CPLX ? 16 : 8;
  int64_t ktot = 0, ws_tot = 0, q_tot = 0;
  int max_mm = 0;
  std::vector<QrBlk> blks;
  std::vector<int> which;  // index of the source block
  for (int b = 0; b < nblocks; ++b) {
    const int64_t m = row_off[b + 1] - row_off[b], n = col_off[b + 1] - col_off[b];
    if (m < 0 || n < 0) return mpse_fail(ctx, MPSE_ERR_SHAPE, "block_qr: negative block extent");
    const int64_t k = m < n ? m : n;
    if (k == 0) continue;
    const int64_t mm = herm ? n : m, nn = herm ? m : n;
    blks.push_back(QrBlk{(long long)ws_tot, (long long)q_tot, (int)mm, (int)nn, (int)k, (int)ktot});
    which.push_back(b);
    ws_tot += mm * nn;
    q_tot += mm * k;
    ktot += k;
    if (mm > max_mm) max_mm = (int)mm;
  }
  if (ktot != K) return mpse_fail(ctx, MPSE_ERR_SHAPE, "block_qr: K=%lld but blocks give %lld", (long long)K, (long long)ktot);
  if (ktot == 0) return mpse_fail(ctx, MPSE_ERR_SHAPE, "Invalid quantum number");
  MPSE_TRY(device_zero2(ctx, U, size_t(nrow * K) * es, Vt, size_t(K * ncol) * es));
  const int64_t nri = row_off[nblocks], nci = col_off[nblocks];
  int64_t max_el = 0, max_sc = 0;
  for (size_t i = 0; i < blks.size(); ++i) {
    blks[i].row_off = row_off[which[i]];
    blks[i].col_off = col_off[which[i]];
    max_el = std::max<int64_t>(max_el, (int64_t)blks[i].mm * blks[i].nn);
    max_sc = std::max<int64_t>(max_sc, (int64_t)blks[i].mm * blks[i].k + (int64_t)blks[i].k * blks[i].nn);
  }
  // optional HIP-event sampling of the whole decomposition (mpse_prof_*, variant 5): Householder factorisation +
  // explicit economic Q, F = (c/2) (4 m n^2 - 4 n^3 / 3) per block with c = 8 complex / 2 real
  double qr_flops = 0.0, qr_bytes = 0.0;
  for (const QrBlk& B : blks) {
    const double m = B.mm, n = B.k;
    qr_flops += (CPLX ? 4.0 : 1.0) * (4.0 * m * n * n - 4.0 * n * n * n / 3.0);
    qr_bytes += double(es) * (2.0 * B.mm * B.nn + double(B.mm) * B.k);
  }
  ProfScope qprof(ctx, 5, qr_flops, qr_bytes);
  // one upload: row index lists | column index lists | block descriptors
  const size_t ib = size_t(nri + nci) * sizeof(int64_t), db = blks.size() * sizeof(QrBlk);
  static_assert(sizeof(QrBlk) % 8 == 0, "descriptors follow the 8-byte index lists");
  std::vector<char> host(ib + db);
  memcpy(host.data(), row_idx, size_t(nri) * sizeof(int64_t));
  memcpy(host.data() + size_t(nri) * sizeof(int64_t), col_idx, size_t(nci) * sizeof(int64_t));
  memcpy(host.data() + ib, blks.data(), db);
  // (replaying the launch sequence from a HIP graph measured no gain - 0.411 vs 0.414 ms per d = 2 decomposition: the
  // ~4 us between the dependent panel / update kernels are spent on the device, not by the host - and was removed)
  const bool batched = max_mm <= HH_BATCH_MAX_ROWS;
  TmpBuf IDX(ctx), WS(ctx), Q(ctx), PRM(ctx);
  MPSE_TRY(IDX.alloc(ib + db));
  MPSE_TRY(WS.alloc(size_t(ws_tot) * es));
  MPSE_TRY(stage_h2d(ctx, IDX.p, host.data(), ib + db));
  const long long* drows = static_cast<const long long*>(IDX.p);
  const long long* dcols = drows + nri;
  const QrBlk* dblk = reinterpret_cast<const QrBlk*>(static_cast<const char*>(IDX.p) + ib);
  double* ws = static_cast<double*>(WS.p);
  constexpr int E = CPLX ? 2 : 1;
  ++ctx->qr_calls;
  hipLaunchKernelGGL((k_gather_blocks<CPLX>), dim3(ew_blocks(max_el), (unsigned)blks.size()), dim3(256), 0, ctx->stream, ws,
                     (const double*)coef, (long long)ncol, drows, dcols, dblk, herm);
  // tall blocks of up to 256 columns: shifted Cholesky-QR on MFMA (mpse_cholqr.hip), all compute units instead of one
  // per block; a block it cannot decide (rank deficient, condition beyond ~1e15) raises a device flag and the whole
  // call is redone by the Householder kernels below on fresh copies of the blocks
  if (!hh_only && cholqr_eligible(ctx, blks.data(), (int)blks.size())) {
    bool ok = false;
    MPSE_TRY(cholqr_blocks(ctx, CPLX, ws, blks.data(), (int)blks.size(), drows, dcols, herm, U, Vt, (long long)K,
                           (long long)ncol, &ok));
    ++ctx->qr_chol_calls;
    if (ok) {
      qprof.end();
      return MPSE_OK;
    }
    ++ctx->qr_chol_fallbacks;
    hipLaunchKernelGGL((k_gather_blocks<CPLX>), dim3(ew_blocks(max_el), (unsigned)blks.size()), dim3(256), 0, ctx->stream,
                       ws, (const double*)coef, (long long)ncol, drows, dcols, dblk, herm);
  }
  MPSE_TRY(Q.alloc(size_t(q_tot) * es));
  MPSE_TRY(PRM.alloc(size_t(ktot + 1) * sizeof(HhParam)));
  double* q = static_cast<double*>(Q.p);
  HhParam* prm = static_cast<HhParam*>(PRM.p);
  if (batched) {
    MPSE_TRY(hh_qr_batched(ctx, CPLX, ws, q, prm, blks.data(), (int)blks.size(), true, dblk));
  } else {
    for (const QrBlk& B : blks) {
      MPSE_TRY(hh_factor_colmajor(ctx, CPLX, ws + B.ws_off * E, B.mm, B.nn, B.k, prm + B.prm_off));
      MPSE_TRY(hh_formq_colmajor(ctx, CPLX, q + B.q_off * E, ws + B.ws_off * E, B.mm, B.k, prm + B.prm_off, B.k));
    }
  }
  hipLaunchKernelGGL((k_scatter_blocks<CPLX>), dim3(ew_blocks(max_sc), (unsigned)blks.size()), dim3(256), 0, ctx->stream,
                     (double*)U, (double*)Vt, (const double*)q, (const double*)ws, (long long)K, (long long)ncol, drows, dcols,
                     dblk, herm);
  qprof.end();
  MPSE_HIP(ctx, hipGetLastError());
  return MPSE_OK;
}

template <bool CPLX>
__global__ void k_gather_cols(double* out, const double* __restrict__ in, long long nrow, long long ncol_in,
                              const long long* __restrict__ cols, const double* __restrict__ scale, long long ncol_out) {
  const long long total = nrow * ncol_out;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const long long j = t % ncol_out, r = t / ncol_out;
    double2 v = Cx<CPLX>::ld(in, r * ncol_in + cols[j]);
    if (scale) {
      v.x *= scale[j];
      v.y *= scale[j];
    }
    Cx<CPLX>::st(out, t, v);
  }
}

template <bool CPLX>
__global__ void k_gather_rows(double* out, const double* __restrict__ in, long long ncol,
                              const long long* __restrict__ rows, const double* __restrict__ scale, long long nrow_out) {
  const long long total = nrow_out * ncol;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const long long j = t % ncol, r = t / ncol;
    double2 v = Cx<CPLX>::ld(in, rows[r] * ncol + j);
    if (scale) {
      v.x *= scale[r];
      v.y *= scale[r];
    }
    Cx<CPLX>::st(out, t, v);
  }
}

}  // namespace

extern "C" {

int mpse_block_qr(mpse_ctx* ctx, int dtype, const void* coef, int64_t nrow, int64_t ncol, int nblocks,
                  const int64_t* row_idx_host, const int64_t* row_off_host, const int64_t* col_idx_host,
                  const int64_t* col_off_host, int system_is_R, void* U, void* Vt, int64_t K) {
  if (!ctx || !coef || !U || !Vt || !row_idx_host || !row_off_host || !col_idx_host || !col_off_host)
    return MPSE_ERR_ARG;
  if (MPSE_RECORDING(ctx) && nblocks > 0) {
    std::vector<int64_t> roff(row_off_host, row_off_host + nblocks + 1), coff(col_off_host, col_off_host + nblocks + 1);
    std::vector<int64_t> ridx(row_idx_host, row_idx_host + roff[nblocks]), cidx(col_idx_host, col_idx_host + coff[nblocks]);
    ctx->defer_ops[ctx->defer_recording].push_back(
        [ctx, dtype, coef, nrow, ncol, nblocks, ridx = std::move(ridx), roff = std::move(roff), cidx = std::move(cidx),
         coff = std::move(coff), system_is_R, U, Vt, K] {
          return mpse_block_qr(ctx, dtype, coef, nrow, ncol, nblocks, ridx.data(), roff.data(), cidx.data(), coff.data(),
                               system_is_R, U, Vt, K);
        });
    return MPSE_OK;
  }
  MPSE_BIND(ctx);
  if (nblocks <= 0) return mpse_fail(ctx, MPSE_ERR_SHAPE, "Invalid quantum number");
  if (system_is_R < 0 || system_is_R > 3) return mpse_fail(ctx, MPSE_ERR_ARG, "block_qr: system_is_R = %d", system_is_R);
  if (dtype == MPSE_C128)
    return block_qr_impl<true>(ctx, coef, nrow, ncol, nblocks, row_idx_host, row_off_host, col_idx_host, col_off_host,
                               system_is_R & 1, (system_is_R & 2) != 0, U, Vt, K);
  if (dtype == MPSE_F64)
    return block_qr_impl<false>(ctx, coef, nrow, ncol, nblocks, row_idx_host, row_off_host, col_idx_host,
                                col_off_host, system_is_R & 1, (system_is_R & 2) != 0, U, Vt, K);
  return mpse_fail(ctx, MPSE_ERR_ARG, "block_qr: unknown dtype");
}

int mpse_block_qr_optimistic(mpse_ctx* ctx, int on) {
  if (!ctx) return MPSE_ERR_ARG;
  MPSE_BIND(ctx);
  MPSE_TRY(qr_words(ctx));
  // the word is cleared on BOTH edges: after a step that tripped it, the verified repeat (mode off) and whatever runs
  // later on this context must not see a stale breakdown (mpse_block_qr_check reports 0 while the mode is off)
  MPSE_TRY(device_zero(ctx, ctx->qr_words_dev, 8));
  ctx->qr_optimistic = on != 0;
  return MPSE_OK;
}

int mpse_block_qr_scheme(mpse_ctx* ctx, int scheme) {
  if (!ctx || scheme < -1 || scheme > 2) return MPSE_ERR_ARG;
  ctx->qr_scheme = scheme;
  return MPSE_OK;
}

int mpse_block_qr_check(mpse_ctx* ctx, int* tripped) {
  if (!ctx || !tripped) return MPSE_ERR_ARG;
  MPSE_BIND(ctx);
  *tripped = 0;
  if (!ctx->qr_words_dev || !ctx->qr_optimistic) return MPSE_OK;   // (mode off: every decomposition was verified as it ran)
  int v[4] = {0, 0, 0, 0};
  MPSE_TRY(mpse_memcpy_d2h(ctx, v, ctx->qr_words_dev, 16));
  *tripped = v[0] != 0;
  return MPSE_OK;
}

int mpse_block_qr_pass_stats(mpse_ctx* ctx, int64_t* blocks, int64_t* two_pass) {
  if (!ctx) return MPSE_ERR_ARG;
  MPSE_BIND(ctx);
  int v[4] = {0, 0, 0, 0};
  if (ctx->qr_words_dev) MPSE_TRY(mpse_memcpy_d2h(ctx, v, ctx->qr_words_dev, 16));
  if (blocks) *blocks = v[2];
  if (two_pass) *two_pass = v[3];
  return MPSE_OK;
}

int mpse_block_qr_stats(mpse_ctx* ctx, int64_t* calls, int64_t* chol_calls, int64_t* chol_fallbacks) {
  if (!ctx) return MPSE_ERR_ARG;
  if (calls) *calls = ctx->qr_calls;
  if (chol_calls) *chol_calls = ctx->qr_chol_calls;
  if (chol_fallbacks) *chol_fallbacks = ctx->qr_chol_fallbacks;
  return MPSE_OK;
}

int mpse_gather_cols(mpse_ctx* ctx, int dtype, void* out, const void* in, int64_t nrow, int64_t ncol_in,
                     const int64_t* cols_host, const double* scale_host, int64_t ncol_out) {
  if (!ctx || !cols_host || (nrow * ncol_out && (!out || !in))) return MPSE_ERR_ARG;
  MPSE_BIND(ctx);
  if (nrow * ncol_out <= 0) return MPSE_OK;
  TmpBuf IDX(ctx);
  MPSE_TRY(IDX.alloc(size_t(ncol_out) * 16));
  MPSE_TRY(stage_h2d(ctx, IDX.p, cols_host, size_t(ncol_out) * 8));
  double* dscale = nullptr;
  if (scale_host) {
    dscale = IDX.as<double>() + ncol_out;
    MPSE_TRY(stage_h2d(ctx, dscale, scale_host, size_t(ncol_out) * 8));
  }
  const int nb = ew_blocks(nrow * ncol_out);
  if (dtype == MPSE_C128)
    hipLaunchKernelGGL((k_gather_cols<true>), dim3(nb), dim3(256), 0, ctx->stream, (double*)out, (const double*)in,
                       (long long)nrow, (long long)ncol_in, IDX.as<const long long>(), (const double*)dscale,
                       (long long)ncol_out);
  else
    hipLaunchKernelGGL((k_gather_cols<false>), dim3(nb), dim3(256), 0, ctx->stream, (double*)out, (const double*)in,
                       (long long)nrow, (long long)ncol_in, IDX.as<const long long>(), (const double*)dscale,
                       (long long)ncol_out);
  MPSE_HIP(ctx, hipGetLastError());
  return MPSE_OK;
}

int mpse_gather_rows(mpse_ctx* ctx, int dtype, void* out, const void* in, int64_t ncol, const int64_t* rows_host,
                     const double* scale_host, int64_t nrow_out) {
  if (!ctx || !rows_host || (nrow_out * ncol && (!out || !in))) return MPSE_ERR_ARG;
  MPSE_BIND(ctx);
  if (nrow_out * ncol <= 0) return MPSE_OK;
  TmpBuf IDX(ctx);
  MPSE_TRY(IDX.alloc(size_t(nrow_out) * 16));
  MPSE_TRY(stage_h2d(ctx, IDX.p, rows_host, size_t(nrow_out) * 8));
  double* dscale = nullptr;
  if (scale_host) {
    dscale = IDX.as<double>() + nrow_out;
    MPSE_TRY(stage_h2d(ctx, dscale, scale_host, size_t(nrow_out) * 8));
  }
  const int nb = ew_blocks(nrow_out * ncol);
  if (dtype == MPSE_C128)
    hipLaunchKernelGGL((k_gather_rows<true>), dim3(nb), dim3(256), 0, ctx->stream, (double*)out, (const double*)in,
                       (long long)ncol, IDX.as<const long long>(), (const double*)dscale, (long long)nrow_out);
  else
    hipLaunchKernelGGL((k_gather_rows<false>), dim3(nb), dim3(256), 0, ctx->stream, (double*)out, (const double*)in,
                       (long long)ncol, IDX.as<const long long>(), (const double*)dscale, (long long)nrow_out);
  MPSE_HIP(ctx, hipGetLastError());
  return MPSE_OK;
}

}  // extern "C"
