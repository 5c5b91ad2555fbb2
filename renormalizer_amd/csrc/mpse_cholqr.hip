// Shifted Cholesky-QR (three passes) of the quantum-number blocks of a centre matrix on FP64 MFMA - the fast path of
// mpse_block_qr (replaces scipy.linalg.qr / rq per block, mps/svd_qn.py:187-197) for tall blocks of up to 256 columns.
//
// Why: the Householder chain (mpse_qr2.hip) is 32 dependent panel rounds on ONE compute unit per block - 0.8 - 0.9 ms
// for a 4096 x 256 centre in two blocks, 1.2 % of the FP64 peak, a quarter of the kernel time of the headline run.
// Gram-matrix schemes put the m n^2 work on all compute units:
//   pass i:  G = A^H A            (k_cq_gram: MFMA, fragments straight from L2, row chunks -> partial sums;
//                                  k_cq_reduce: fixed-order sum of the chunks, per-tile |G - I| and trace)
//            R_i = chol(G [+ s I]) (one workgroup per block.  k_cq_chol_rl, up to 160 columns: right-looking, every
//                                  tile of the trailing matrix stays in the MFMA accumulators of one of 8 waves, the row
//                                  panel of a step goes through LDS; k_cq_chol, 161 - 256 columns: left-looking by
//                                  16-row panels.  Both eliminate a row panel - diagonal factor and forward
//                                  substitution at once - with a thread per column, panel_eliminate)
//            A <- A R_i^-1         (k_cq_trsm: one workgroup per 16 rows, right-looking over 16-column panels,
//                                  substitution inside a diagonal block by DPP row broadcasts - backward stable, an
//                                  explicit inverse would leave a residual u kappa(R))
//   R = R_3 R_2 R_1 (tile products riding on the TRSM launches of passes 2 and 3).
// Pass 1 is shifted (s = 11 (m n + n (n + 1)) u trace(G) >= the bound of Fukaya et al., SIAM J. Sci. Comput. 42 (2020)
// A477, which uses |A|_2^2): it cannot break down and leaves kappa(A_1) <~ 1e8 for kappa(A) up to ~1e15; passes 2 and 3
// are plain CholeskyQR2.  When the Gram matrix of pass 3 is the identity to first order (n max|G - I| <= 1e-9) its
// factor is written down (R = I + striu(E) + diag(E) / 2, error O(|E|^2)) and the triangular solve becomes a product.
//
// Rank-deficient / too ill-conditioned blocks (the 1e-10 padding of expand_bond_dimension in the first steps of a
// run: profiles/r05_qr_cond_step2.md - 4 of 98 decompositions; none of 98 at step 13) make a pivot of pass 2 / 3
// non-positive or leave the pivots of pass 3 outside [1/4, 4]: the kernels raise a device flag.  Either the host reads it
// after the last launch (one mapped-memory read-back) and mpse_block_qr runs the Householder path on those inputs, or -
// optimistic mode, mpse_block_qr_optimistic, for callers that can repeat a whole step - the flag is a sticky device word
// read once at the end of the step (the read-back idles the GPU for ~80 us per decomposition).
// Householder therefore still decides every case Cholesky cannot: results are an exact isometry either way.  The two
// schemes agree up to the phases of the columns where the block has full numerical rank; inside a numerical null space
// every QR is a different, equally valid completion (profiles/r05_qr_gauge.md).
#include <type_traits>

#include "mpse_device.h"
#include "mpse_internal.h"

typedef double v4d __attribute__((ext_vector_type(4)));

namespace {

struct CqBlk {
  long long ws_off;     // element offset of the mm x nn column-major block in the workspace
  long long part_off;   // tile-element offset ([chunk][T][256]) of its Gram partial sums
  long long t_off;      // tile offset (in tiles) of its G / R tiles and tile infos
  long long row_off, col_off;   // where the block's row / column index lists start (device lists of mpse_block_qr)
  int mm, nn, P, T;     // P = ceil(nn / 16) panels, T = P (P + 1) / 2 upper tiles
  int nchunk, rpc;      // row chunks of the Gram product, rows per chunk (multiple of 4)
  int koff, pad;        // first column of the block's factor in U / Vt
};

__device__ __forceinline__ v4d mfma(double a, double b, v4d c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ double readlane_d(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
// value of lane T of the own 16-lane row (DPP row_share)
template <int T>
__device__ __forceinline__ double row_share(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + T, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + T, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int tile_index(int q, int c, int P) { return q * P - (q * (q - 1)) / 2 + (c - q); }
__device__ __forceinline__ void tile_decode(int t, int P, int& q, int& c) {
  q = 0;
  while (t >= P - q) {
    t -= P - q;
    ++q;
  }
  c = q + t;
}

template <bool CPLX>
__device__ __forceinline__ void ld2(const double* p, long long i, double& re, double& im) {
  if constexpr (CPLX) {
    const double2 v = reinterpret_cast<const double2*>(p)[i];
    re = v.x;
    im = v.y;
  } else {
    re = p[i];
    im = 0.0;
  }
}
template <bool CPLX>
__device__ __forceinline__ void st2(double* p, long long i, double re, double im) {
  if constexpr (CPLX)
    reinterpret_cast<double2*>(p)[i] = make_double2(re, im);
  else
    p[i] = re;
}

// ---------------------------------------------------------------------------------------------------------- Gram
// grid (max over blocks of nmacro * T2, nblk).  A wave owns a 2 x 2 group of 16 x 16 tiles of the upper triangle of G
// (super-tile (Q, C), Q <= C: tiles (2Q + a, 2C + b)) over one row chunk; the four waves of a workgroup take four
// consecutive chunks of the SAME super-tile and add their results through LDS in wave order, so one partial sum per
// workgroup ("macro chunk") goes to memory - a quarter of the partial-sum traffic one partial per wave would cost.
// MFMA operands straight from global memory: lane (x = lane & 15, k = lane >> 4) reads A[r + k, 16 q + x] - four
// consecutive rows of a column per 16-lane group (64-byte segments; the workspace stays in L2 between the passes).
// Four operand loads feed sixteen MFMAs; the loads of two row steps are in flight ahead of the arithmetic.
template <bool CPLX>
__global__ __launch_bounds__(256) void k_cq_gram(const double* __restrict__ ws, double* __restrict__ part,
                                                  const CqBlk* __restrict__ blks, int* __restrict__ status, int nstat,
                                                  const int* __restrict__ done) {
  constexpr int E = CPLX ? 2 : 1;
  if (done && done[blockIdx.y]) return;     // (pass 3 of a block whose second pass was the last one)
  const CqBlk B = blks[blockIdx.y];
  if (nstat > 0 && blockIdx.x == 0 && blockIdx.y == 0)
    for (int i = threadIdx.x; i < nstat; i += 256) status[i] = 0;
  const int P2 = (B.P + 1) >> 1, T2 = P2 * (P2 + 1) / 2;
  const int nmacro = (B.nchunk + 3) >> 2;
  const int macro = blockIdx.x / T2, t2 = blockIdx.x - macro * T2;
  if (macro >= nmacro) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), x = lane & 15, kq = lane >> 4;
  int Q, C;
  tile_decode(t2, P2, Q, C);
  const int chunk = macro * 4 + wave;
  const int rb = min(chunk * B.rpc, B.mm), re = min(B.mm, rb + B.rpc);   // (chunks past the last one: empty range)
  const double* A = ws + B.ws_off * E;
  const long long mm = B.mm;
  // columns of the four operand fragments: a-side tiles 2Q, 2Q + 1, b-side tiles 2C, 2C + 1
  bool okc[4];
  long long fo[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int col = 16 * (2 * (i < 2 ? Q : C) + (i & 1)) + x;
    okc[i] = col < B.nn;
    fo[i] = (long long)min(col, B.nn - 1) * mm;
  }
  v4d gr[2][2], gi[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      gr[i][j] = v4d{0, 0, 0, 0};
      gi[i][j] = v4d{0, 0, 0, 0};
    }
  double vr[2][4], vi[2][4];   // [buffer][fragment]
#pragma unroll
  for (int i = 0; i < 4; ++i) vr[0][i] = vi[0][i] = vr[1][i] = vi[1][i] = 0.0;
  // (loads are unconditional - rows and columns clamped into the block - and masked where they are consumed: a select
  // right behind a load waits for it on the spot, and nothing would be in flight ahead of the arithmetic)
  auto load = [&](int buf, int r) {
    const long long rc = min(r + kq, B.mm - 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) ld2<CPLX>(A, fo[i] + rc, vr[buf][i], vi[buf][i]);
  };
  if (rb < re) load(0, rb);
  if (rb + 4 < re) load(1, rb + 4);
  for (int r = rb; r < re; r += 8) {
    double cr[4], ci[4], dr[4], di[4];
    const bool ok0 = r + kq < re, ok1 = r + 4 + kq < re;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      cr[i] = (ok0 && okc[i]) ? vr[0][i] : 0.0;
      ci[i] = (ok0 && okc[i]) ? vi[0][i] : 0.0;
      dr[i] = (ok1 && okc[i]) ? vr[1][i] : 0.0;
      di[i] = (ok1 && okc[i]) ? vi[1][i] : 0.0;
    }
    if (r + 8 < re) load(0, r + 8);
    if (r + 12 < re) load(1, r + 12);
    asm volatile("" ::: "memory");
    const bool second = r + 4 < re;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // conj(x) y = (xr yr + xi yi) + i (xr yi - xi yr)
        gr[i][j] = mfma(cr[i], cr[2 + j], gr[i][j]);
        if constexpr (CPLX) {
          gr[i][j] = mfma(ci[i], ci[2 + j], gr[i][j]);
          gi[i][j] = mfma(cr[i], ci[2 + j], gi[i][j]);
          gi[i][j] = mfma(-ci[i], cr[2 + j], gi[i][j]);
        }
      }
    if (second) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          gr[i][j] = mfma(dr[i], dr[2 + j], gr[i][j]);
          if constexpr (CPLX) {
            gr[i][j] = mfma(di[i], di[2 + j], gr[i][j]);
            gi[i][j] = mfma(dr[i], di[2 + j], gi[i][j]);
            gi[i][j] = mfma(-di[i], dr[2 + j], gi[i][j]);
          }
        }
    }
  }
  // the four waves' results added in wave order, one partial sum per workgroup
  __shared__ double2 s_acc[4][4][256];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) s_acc[wave][2 * i + j][(kq + 4 * r) * 16 + x] = make_double2(gr[i][j][r], gi[i][j][r]);
  __syncthreads();
#pragma unroll
  for (int ij = 0; ij < 4; ++ij) {
    const int q = 2 * Q + (ij >> 1), c = 2 * C + (ij & 1);
    if (q > c || c >= B.P) continue;   // (the lower tile of a diagonal group, tiles past the last panel)
    double2 v = s_acc[0][ij][tid];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      v.x += s_acc[w][ij][tid].x;
      v.y += s_acc[w][ij][tid].y;
    }
    st2<CPLX>(part, B.part_off + ((long long)macro * B.T + tile_index(q, c, B.P)) * 256 + tid, v.x, v.y);
  }
}

// one workgroup per tile: G tile = sum of the chunks' partial sums in chunk order; tinfo[tile] = (max |G - delta| over
// the tile's valid entries, sum of the real diagonal of a diagonal tile)
template <bool CPLX>
__global__ __launch_bounds__(256) void k_cq_reduce(const double* __restrict__ part, double* __restrict__ G,
                                                    double* __restrict__ tinfo, const CqBlk* __restrict__ blks,
                                                    const int* __restrict__ done) {
  constexpr int E = CPLX ? 2 : 1;
  if (done && done[blockIdx.y]) return;
  const CqBlk B = blks[blockIdx.y];
  const int t = blockIdx.x;
  if (t >= B.T) return;
  const int e = threadIdx.x, row = e >> 4, col = e & 15;
  double sr = 0.0, si = 0.0;
  const double* p = part + (B.part_off + (long long)t * 256) * E;
  const long long stride = (long long)B.T * 256;
  const int nmacro = (B.nchunk + 3) >> 2;
  for (int ch = 0; ch < nmacro; ch += 8) {   // eight loads in flight, added in chunk order
    double a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a[u] = b[u] = 0.0;
      if (ch + u < nmacro) ld2<CPLX>(p, (ch + u) * stride + e, a[u], b[u]);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      sr += a[u];
      si += b[u];
    }
  }
  st2<CPLX>(G, (B.t_off + t) * 256 + e, sr, si);
  int q, c;
  tile_decode(t, B.P, q, c);
  const bool valid = 16 * q + row < B.nn && 16 * c + col < B.nn;
  const bool diag = q == c && row == col;
  double dev = 0.0, tr = 0.0;
  if (valid) {
    const double dr = sr - (diag ? 1.0 : 0.0);
    dev = fmax(fabs(dr), fabs(si));
    if (diag) tr = sr;
  }
  // block max / sum in a fixed order
  __shared__ double s_m[256], s_t[256];
  s_m[e] = dev;
  s_t[e] = tr;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if (e < h) {
      s_m[e] = fmax(s_m[e], s_m[e + h]);
      s_t[e] += s_t[e + h];
    }
    __syncthreads();
  }
  if (e == 0) {
    tinfo[2 * (B.t_off + t)] = s_m[0];
    tinfo[2 * (B.t_off + t) + 1] = s_t[0];
  }
}

// Row panel p of the factor from the Schur-complement tiles S(p, p .. P-1) (16 rows x W = 16 (P - p) columns in LDS,
// tile 0 = the diagonal tile): the Cholesky elimination of the diagonal tile applied to the WHOLE panel, one thread per
// column with its 16 rows in registers.  Step k: every column subtracts conj(s_ki) s_kj / s_kk from its rows i > k;
// what the step needs from other columns is row k of the diagonal tile alone, which the 16 threads of that tile
// publish in place one step ahead.  After 15 steps row i holds S_i(final); R(p, :) = rows scaled by 1 / sqrt(s_ii).
// This replaces "factor the diagonal tile on one wave, then forward-substitute the other tiles" (4.4 + 2.5 us per panel,
// measured by switching the phases off: a dependent chain of FP64 operations on a single wave) by 15 short steps on
// all columns at once.  Called by every thread of the workgroup (the barriers are workgroup barriers); threads past
// the panel width only keep the barriers.  Pivots are checked at the end: status |= 1 (through *bad).
//
// Pivot shift (pass 1 with a per-pivot shift, `thr` != nullptr): a pivot that has lost all but a fraction theta of its
// diagonal entry of G (d_k <= thr[k] = theta G_kk) counts as d_k + shift (shift = sDinv[16]) - the factor is that of
// G + D with a diagonal 0 <= D <= shift I chosen on the way.  The eigenvalues of Q1^H Q1 = R1^-H G R1^-1 then still lie
// in [lambda_min / (lambda_min + shift), 1] (D <= shift I is all the bound of the fully shifted scheme uses), so
// passes 2 and 3 see nothing worse than before, while a block that is well conditioned up to column scaling is not
// shifted at all and leaves pass 1 orthogonal to ~u kappa^2: its second pass can be the last (cholqr_run).
// Every thread applies the rule to the pivot it reads (the published tile keeps the unshifted value): no thread is
// singled out inside the steps.  sDinv: [0, 16) 1 / sqrt(pivot), [16] the shift, [17, 33) sqrt(pivot).
template <bool CPLX>
__device__ __forceinline__ void panel_eliminate(double2 (*tiles)[256], int W, int p, int P, int nn_unused, double* Rt,
                                                double* sDinv, bool window, int tid, int* bad, const double* thr) {
  (void)nn_unused;
  constexpr int E = CPLX ? 2 : 1;
  const bool act = tid < W;
  const int ct = tid >> 4, col = tid & 15;
  const bool dyn = thr != nullptr;
  double cr[16], ci[16];
  if (act) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const double2 v = tiles[ct][i * 16 + col];
      cr[i] = v.x;
      ci[i] = v.y;
    }
  }
  auto step = [&](auto kc) {
    constexpr int k = decltype(kc)::value;
    if (act) {
      double piv = tiles[0][k * 16 + k].x;                       // (a bad pivot makes garbage here; flagged below)
      if (dyn) piv = piv > thr[k] ? piv : piv + sDinv[16];
      const double inv = fast_rcp(piv);
      const double tr = cr[k] * inv, ti = ci[k] * inv;
#pragma unroll
      for (int i = k + 1; i < 16; ++i) {
        const double2 e = tiles[0][k * 16 + i];   // s(k, i); rows i > k lose conj(s_ki) s_kj / s_kk
        cr[i] -= e.x * tr + e.y * ti;
        if constexpr (CPLX) ci[i] -= e.x * ti - e.y * tr;
      }
      if (ct == 0) tiles[0][(k + 1) * 16 + col] = make_double2(cr[k + 1], ci[k + 1]);
    }
    lds_barrier();
  };
  step(std::integral_constant<int, 0>{});
  step(std::integral_constant<int, 1>{});
  step(std::integral_constant<int, 2>{});
  step(std::integral_constant<int, 3>{});
  step(std::integral_constant<int, 4>{});
  step(std::integral_constant<int, 5>{});
  step(std::integral_constant<int, 6>{});
  step(std::integral_constant<int, 7>{});
  step(std::integral_constant<int, 8>{});
  step(std::integral_constant<int, 9>{});
  step(std::integral_constant<int, 10>{});
  step(std::integral_constant<int, 11>{});
  step(std::integral_constant<int, 12>{});
  step(std::integral_constant<int, 13>{});
  step(std::integral_constant<int, 14>{});
  // pivots s_ii (row i as published, shifted by the same rule): checks, 1 / sqrt and sqrt
  if (tid < 16) {
    double d = tiles[0][tid * 16 + tid].x;
    if (dyn) d = d > thr[tid] ? d : d + sDinv[16];
    bool ok = d > 0.0 && d < 1e300;
    if (window && !(d >= 0.25 && d <= 4.0)) ok = false;         // (last pass: the Gram matrix was the identity to O(0.1))
    if (!(d > 0.0 && d < 1e300)) d = 1.0;
    if (!ok) *bad = 1;
    const double rs = fast_rsqrt(d);
    sDinv[tid] = rs;
    sDinv[17 + tid] = d * rs;
  }
  lds_barrier();
  if (act) {
    double* td = Rt + (long long)tile_index(p, p + ct, P) * 256 * E;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const double rs = sDinv[i];
      double vr = cr[i] * rs, vi = ci[i] * rs;
      if (ct == 0) {
        if (i > col) vr = vi = 0.0;
        if (i == col) {
          vr = sDinv[17 + i];
          vi = 0.0;
        }
      }
      tiles[ct][i * 16 + col] = make_double2(vr, vi);
      st2<CPLX>(td, i * 16 + col, vr, vi);
    }
  }
}

// ------------------------------------------------------------------------------------------------------ Cholesky
// One workgroup (8 waves) per block: R^H R = G (+ shift), R upper triangular, stored as 16 x 16 tiles (q <= c), row
// major inside a tile.  Left-looking by row panels: S(p, c) = G(p, c) - sum_{q<p} R(q, p)^H R(q, c) on MFMA with the
// operands read from the tiles already written (global memory / L2, same workgroup); the diagonal tile is factorised
// in the registers of wave 0 (lane j = column j, rows by v_readlane broadcasts), the rest of the row panel solved
// by forward substitution, one thread per column.
// status[0] |= 1 on a breakdown; status[1 + block] = 1 when the first-order factor was taken (pass 3 only).
template <bool CPLX>
__global__ __launch_bounds__(512) void k_cq_chol(const double* __restrict__ G, const double* __restrict__ tinfo,
                                                  double* __restrict__ R, const CqBlk* __restrict__ blks,
                                                  int* __restrict__ status, int pass, int* __restrict__ gflag, int nblk,
                                                  double theta, double tau) {
  constexpr int E = CPLX ? 2 : 1;
  int* const donep = status + 1 + nblk + blockIdx.x;
  if (pass == 3 && *donep) return;      // the second pass of this block was the last one
  const CqBlk B = blks[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), x = lane & 15, kq = lane >> 4;
  const int P = B.P, T = B.T, nn = B.nn;
  const double* Gt = G + B.t_off * 256 * E;
  double* Rt = R + B.t_off * 256 * E;
  __shared__ double2 sS[16][256];
  __shared__ double sDinv[33];     // [0, 16) 1 / sqrt(pivot) of the panel, [16] the shift of pass 1, [17, 33) sqrt(pivot)
  __shared__ double s_red[2][512];
  __shared__ double sThr[256];
  const bool dyn = pass == 1 && theta > 0.0;
  if (dyn)     // pivot thresholds theta G_kk (padding rows: none)
    for (int i = tid; i < 16 * P; i += 512)
      sThr[i] = i < nn ? theta * Gt[((long long)tile_index(i >> 4, i >> 4, P) * 256 + (i & 15) * 17) * E] : -1.0;
  // trace and max |G - I| of the block (fixed order)
  {
    double m = 0.0, tr = 0.0;
    for (int t = tid; t < T; t += 512) {
      m = fmax(m, tinfo[2 * (B.t_off + t)]);
      tr += tinfo[2 * (B.t_off + t) + 1];
    }
    s_red[0][tid] = m;
    s_red[1][tid] = tr;
    __syncthreads();
    for (int h = 256; h > 0; h >>= 1) {
      if (tid < h) {
        s_red[0][tid] = fmax(s_red[0][tid], s_red[0][tid + h]);
        s_red[1][tid] += s_red[1][tid + h];
      }
      __syncthreads();
    }
  }
  const double maxdev = s_red[0][0], trace = s_red[1][0];
  const double shift = pass == 1 ? 11.0 * ((double)B.mm * nn + (double)nn * (nn + 1)) * 1.1102230246251565e-16 * trace : 0.0;
  if (tid == 0) sDinv[16] = shift;     // (visible after the barriers ahead of the first panel_eliminate)
  // adaptive pass count: when the Gram matrix of pass 2 is the identity to within tau, the factor of this pass leaves an
  // isometry to rounding (u kappa(Q1)^2 with kappa(Q1)^2 <= (1 + tau) / (1 - tau)) and pass 3 is skipped for the block
  const bool last = pass == 3 || (pass == 2 && (double)nn * maxdev <= tau);
  if (pass == 2 && tid == 0) *donep = last ? 1 : 0;
  if (pass >= 2 && (double)nn * maxdev <= 1e-9) {
    // first-order factor of G = I + E:  R = I + striu(E) + diag(E) / 2   (error O(|E|^2) <= 1e-18)
    for (int t = 0; t < T; ++t) {
      int q, c;
      tile_decode(t, P, q, c);
      if (tid < 256) {
        const int row = tid >> 4, col = tid & 15;
        double gr, gi;
        ld2<CPLX>(Gt, (long long)t * 256 + tid, gr, gi);
        if (q == c) {
          if (row > col) gr = gi = 0.0;
          if (row == col) {
            gr = 16 * q + row < nn ? 0.5 * (1.0 + gr) : 1.0;
            gi = 0.0;
          }
        }
        st2<CPLX>(Rt, (long long)t * 256 + tid, gr, gi);
      }
    }
    if (tid == 0) status[1 + blockIdx.x] = 1;
    return;
  }
  if (tid == 0) status[1 + blockIdx.x] = 0;
  int bad = 0;
  for (int p = 0; p < P; ++p) {
    // ---- S tiles of row panel p: wave w takes the column tiles c = p + w, p + w + 8
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int c = p + wave + 8 * s;
      if (c >= P) break;
      v4d ar, ai = {0, 0, 0, 0};
      {
        const double* g0 = Gt + (long long)tile_index(p, c, P) * 256 * E;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = kq + 4 * r;
          double gr, gi;
          ld2<CPLX>(g0, row * 16 + x, gr, gi);
          if (c == p && row == x) {
            gi = 0.0;
            gr = 16 * p + row < nn ? (dyn ? gr : gr + shift) : 1.0;
          }
          ar[r] = gr;
          ai[r] = gi;
        }
      }
      for (int q = 0; q < p; ++q) {
        const double* ta = Rt + (long long)tile_index(q, p, P) * 256 * E;
        const double* tb = Rt + (long long)tile_index(q, c, P) * 256 * E;
        double xr[4], xi[4], yr[4], yi[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          ld2<CPLX>(ta, (4 * kk + kq) * 16 + x, xr[kk], xi[kk]);
          ld2<CPLX>(tb, (4 * kk + kq) * 16 + x, yr[kk], yi[kk]);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          // S -= conj(x) y
          ar = mfma(-xr[kk], yr[kk], ar);
          if constexpr (CPLX) {
            ar = mfma(-xi[kk], yi[kk], ar);
            ai = mfma(-xr[kk], yi[kk], ai);
            ai = mfma(xi[kk], yr[kk], ai);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) sS[c - p][(kq + 4 * r) * 16 + x] = make_double2(ar[r], ai[r]);
    }
    __syncthreads();
    // ---- the row panel of the factor (diagonal factor + forward substitution in one elimination, see panel_eliminate)
    {
      int badp = 0;
      panel_eliminate<CPLX>(sS, 16 * (P - p), p, P, nn, Rt, sDinv, last, tid, &badp, dyn ? sThr + 16 * p : nullptr);
      if (badp) bad = 1;
    }
    __syncthreads();   // the tiles of row panel p are visible to the MFMA loads of the next panels
  }
  if (bad) {
    atomicOr(status, 1);
    if (gflag) atomicOr(gflag, 1);   // optimistic mode: the context's sticky flag, read at the end of the sweep
  }
}

// Right-looking form for blocks of up to 160 columns (P <= 10, T <= 55): every tile of the trailing matrix lives in
// the MFMA accumulators of one of the 8 waves for the whole factorisation (tile t -> wave t mod 8, slot t / 8), the row
// panel of a step goes through LDS (two buffers by parity) and is the operand of all updates of that step - no global
// load sits inside the loop (the left-looking form above read its operands back from L2 in a loop of dependent round
// trips: 100 us per factorisation at n = 145 against the ~35 us the chain diagonal factor -> row solve -> update needs).
// 8 waves with 7 accumulator slots each hold the 55 tiles of P <= 10 (160 columns); ten slots per wave, or 16 waves of
// five, spilled and lost to the left-looking kernel, which keeps the blocks of 161 - 256 columns.
template <bool CPLX>
__global__ __launch_bounds__(512) void k_cq_chol_rl(const double* __restrict__ G, const double* __restrict__ tinfo,
                                                     double* __restrict__ R, const CqBlk* __restrict__ blks,
                                                     int* __restrict__ status, int pass, int* __restrict__ gflag, int nblk,
                                                     double theta, double tau) {
  constexpr int E = CPLX ? 2 : 1, PMAX = 10, NW = 8, NS = 7, NT = 64 * NW;
  int* const donep = status + 1 + nblk + blockIdx.x;
  if (pass == 3 && *donep) return;      // the second pass of this block was the last one
  const CqBlk B = blks[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), x = lane & 15, kq = lane >> 4;
  const int P = B.P, T = B.T, nn = B.nn;
  const double* Gt = G + B.t_off * 256 * E;
  double* Rt = R + B.t_off * 256 * E;
  __shared__ double2 sRow[2][PMAX][256];
  __shared__ double sDinv[33];     // [0, 16) 1 / sqrt(pivot) of the panel, [16] the shift of pass 1, [17, 33) sqrt(pivot)
  __shared__ double sThr[16 * PMAX];
  double* s_red = reinterpret_cast<double*>(&sRow[0][0][0]);   // scratch of the reductions before the loop
  const bool dyn = pass == 1 && theta > 0.0;
  if (dyn)     // pivot thresholds theta G_kk (padding rows: none)
    for (int i = tid; i < 16 * P; i += NT)
      sThr[i] = i < nn ? theta * Gt[((long long)tile_index(i >> 4, i >> 4, P) * 256 + (i & 15) * 17) * E] : -1.0;
  {
    double m = 0.0, tr = 0.0;
    for (int t = tid; t < T; t += NT) {
      m = fmax(m, tinfo[2 * (B.t_off + t)]);
      tr += tinfo[2 * (B.t_off + t) + 1];
    }
    s_red[tid] = m;
    s_red[NT + tid] = tr;
    __syncthreads();
    for (int h = NT / 2; h > 0; h >>= 1) {
      if (tid < h) {
        s_red[tid] = fmax(s_red[tid], s_red[tid + h]);
        s_red[NT + tid] += s_red[NT + tid + h];
      }
      __syncthreads();
    }
  }
  const double maxdev = s_red[0], trace = s_red[NT];
  __syncthreads();
  const double shift = pass == 1 ? 11.0 * ((double)B.mm * nn + (double)nn * (nn + 1)) * 1.1102230246251565e-16 * trace : 0.0;
  if (tid == 0) sDinv[16] = shift;     // (visible after the barriers ahead of the first panel_eliminate)
  const bool last = pass == 3 || (pass == 2 && (double)nn * maxdev <= tau);     // (adaptive pass count: see k_cq_chol)
  if (pass == 2 && tid == 0) *donep = last ? 1 : 0;
  if (pass >= 2 && (double)nn * maxdev <= 1e-9) {
    for (int t = 0; t < T; ++t) {
      int q, c;
      tile_decode(t, P, q, c);
      if (tid < 256) {
        const int row = tid >> 4, col = tid & 15;
        double gr, gi;
        ld2<CPLX>(Gt, (long long)t * 256 + tid, gr, gi);
        if (q == c) {
          if (row > col) gr = gi = 0.0;
          if (row == col) {
            gr = 16 * q + row < nn ? 0.5 * (1.0 + gr) : 1.0;
            gi = 0.0;
          }
        }
        st2<CPLX>(Rt, (long long)t * 256 + tid, gr, gi);
      }
    }
    if (tid == 0) status[1 + blockIdx.x] = 1;
    return;
  }
  if (tid == 0) status[1 + blockIdx.x] = 0;
  // this wave's tiles
  int tq[NS], tc[NS];
  v4d ar[NS], ai[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int t = wave + NW * s;
    tq[s] = tc[s] = -1;
    ar[s] = v4d{0, 0, 0, 0};
    ai[s] = v4d{0, 0, 0, 0};
    if (t < T) {
      tile_decode(t, P, tq[s], tc[s]);
      const double* g0 = Gt + (long long)t * 256 * E;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = kq + 4 * r;
        double gr, gi;
        ld2<CPLX>(g0, row * 16 + x, gr, gi);
        if (tq[s] == tc[s] && row == x) {
          gi = 0.0;
          gr = 16 * tq[s] + row < nn ? (dyn ? gr : gr + shift) : 1.0;
        }
        ar[s][r] = gr;
        ai[s][r] = gi;
      }
    }
  }
  int bad = 0;
  for (int p = 0; p < P; ++p) {
    const int buf = p & 1;
    // ---- the row panel of this step leaves the accumulators
#pragma unroll
    for (int s = 0; s < NS; ++s)
      if (tq[s] == p) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sRow[buf][tc[s] - p][(kq + 4 * r) * 16 + x] = make_double2(ar[s][r], ai[s][r]);
      }
    __syncthreads();
    // ---- the row panel of the factor (diagonal factor + forward substitution in one elimination, see panel_eliminate)
    int badp = 0;
    panel_eliminate<CPLX>(sRow[buf], 16 * (P - p), p, P, nn, Rt, sDinv, last, tid, &badp, dyn ? sThr + 16 * p : nullptr);
    if (badp) bad = 1;
    __syncthreads();
    // ---- updates: S(q, c) -= R(p, q)^H R(p, c) for this wave's tiles below the panel
#pragma unroll
    for (int s = 0; s < NS; ++s)
      if (tq[s] > p) {
        const double2* ta = sRow[buf][tq[s] - p];
        const double2* tb = sRow[buf][tc[s] - p];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const double2 xv = ta[(4 * kk + kq) * 16 + x], yv = tb[(4 * kk + kq) * 16 + x];
          ar[s] = mfma(-xv.x, yv.x, ar[s]);
          if constexpr (CPLX) {
            ar[s] = mfma(-xv.y, yv.y, ar[s]);
            ai[s] = mfma(-xv.x, yv.y, ai[s]);
            ai[s] = mfma(xv.y, yv.x, ai[s]);
          }
        }
      }
  }
  if (bad) {
    atomicOr(status, 1);
    if (gflag) atomicOr(gflag, 1);   // optimistic mode: the context's sticky flag, read at the end of the sweep
  }
}

// --------------------------------------------------------------------------------------- triangular solve + R product
// grid (ceil(max mm / 16) + (rmul ? max T : 0), nblk), 4 waves.
//   x < ceil(mm / 16): rows 16 x .. 16 x + 15 of the block:  X R = A in place.  Wave w owns the column panels
//     p = w, w + 4, .. as MFMA accumulators; for q = 0 .. P-1 the owner solves X_q R_qq = A_q by substitution inside the
//     accumulator layout (lane = (row group, column): the finished column t of X_q is broadcast along the 16-lane rows
//     by DPP row_share), stores X_q and publishes it through LDS as the A operand of the updates A_p -= X_q R(q, p).
//     mode 1 (first-order factor R = I + U of pass 3): X_p = A_p - sum_{q<=p} A_q U(q, p), no dependent chain.
//   the other workgroups: one tile each of Rout = Rcur . Rprev (the accumulated triangular factor).
template <bool CPLX>
__global__ __launch_bounds__(256) void k_cq_trsm(double* __restrict__ ws, const double* __restrict__ R,
                                                  const double* __restrict__ Rprev, double* __restrict__ Rout,
                                                  const CqBlk* __restrict__ blks, const int* __restrict__ status, int nrb_max,
                                                  int rmul, const int* __restrict__ done) {
  constexpr int E = CPLX ? 2 : 1;
  if (done && done[blockIdx.y]) return;     // (pass 3 of a block whose second pass was the last one)
  const CqBlk B = blks[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), x = lane & 15, kq = lane >> 4;
  const int P = B.P, nn = B.nn;
  const double* Rt = R + B.t_off * 256 * E;
  __shared__ double sXr[2][16 * 17], sXi[2][16 * 17];
  __shared__ double2 sAcc[4][256];
  // which part of the launch: blockIdx.x < nrb_max = row blocks (of the tallest block), then the product tiles
  if ((int)blockIdx.x >= nrb_max) {
    if (!rmul) return;
    const int t = (int)blockIdx.x - nrb_max;
    if (t >= B.T) return;
    int q, c;
    tile_decode(t, P, q, c);
    const double* Rp = Rprev + B.t_off * 256 * E;
    v4d ar = {0, 0, 0, 0}, ai = {0, 0, 0, 0};
    for (int s = q + wave; s <= c; s += 4) {
      const double* ta = Rt + (long long)tile_index(q, s, P) * 256 * E;
      const double* tb = Rp + (long long)tile_index(s, c, P) * 256 * E;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        double xr, xi, yr, yi;
        ld2<CPLX>(ta, x * 16 + 4 * kk + kq, xr, xi);
        ld2<CPLX>(tb, (4 * kk + kq) * 16 + x, yr, yi);
        ar = mfma(xr, yr, ar);
        if constexpr (CPLX) {
          ar = mfma(-xi, yi, ar);
          ai = mfma(xr, yi, ai);
          ai = mfma(xi, yr, ai);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) sAcc[wave][(kq + 4 * r) * 16 + x] = make_double2(ar[r], ai[r]);
    __syncthreads();
    double2 v = sAcc[0][tid];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      v.x += sAcc[w][tid].x;
      v.y += sAcc[w][tid].y;
    }
    st2<CPLX>(Rout + B.t_off * 256 * E, (long long)t * 256 + tid, v.x, v.y);
    return;
  }
  const int r0 = 16 * blockIdx.x;
  if (r0 >= B.mm) return;
  double* A = ws + B.ws_off * E;
  const long long mm = B.mm;
  const int mode = status[1 + blockIdx.y];
  // this wave's panels as accumulators: slot s <-> panel p = wave + 4 s; lane holds rows kq + 4 r, column x
  v4d ar[4], ai[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int p = wave + 4 * s;
    ar[s] = v4d{0, 0, 0, 0};
    ai[s] = v4d{0, 0, 0, 0};
    if (p < P) {      // (unconditional loads, clamped into the block; masked below, once all are on their way)
      const int col = min(16 * p + x, nn - 1);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = min(r0 + kq + 4 * r, B.mm - 1);
        double vr, vi;
        ld2<CPLX>(A, (long long)col * mm + row, vr, vi);
        ar[s][r] = vr;
        ai[s][r] = vi;
      }
    }
  }
  asm volatile("" ::: "memory");
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int p = wave + 4 * s;
    if (p < P) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (!(r0 + kq + 4 * r < B.mm && 16 * p + x < nn)) ar[s][r] = ai[s][r] = 0.0;
    }
  }
  auto store_panel = [&](int p, const v4d& vr, const v4d& vi) {
    const int col = 16 * p + x;
    if (col < nn) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = r0 + kq + 4 * r;
        if (row < B.mm) st2<CPLX>(A, (long long)col * mm + row, vr[r], vi[r]);
      }
    }
  };
  if (mode == 1) {
    // X_p = A_p - sum_{q <= p} A_q U(q, p), U = R - I; A operands straight from global memory (rows contiguous)
    const int arow = min(r0 + x, B.mm - 1);
    const bool rok = r0 + x < B.mm;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int p = wave + 4 * s;
      if (p >= P) break;
      for (int q = 0; q <= p; ++q) {
        const double* tb = Rt + (long long)tile_index(q, p, P) * 256 * E;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int acol = 16 * q + 4 * kk + kq;
          double xr = 0.0, xi = 0.0, yr, yi;
          ld2<CPLX>(A, (long long)min(acol, nn - 1) * mm + arow, xr, xi);
          if (!(rok && acol < nn)) xr = xi = 0.0;
          ld2<CPLX>(tb, (4 * kk + kq) * 16 + x, yr, yi);
          if (q == p && 4 * kk + kq == x) yr -= 1.0;
          // acc -= x y
          ar[s] = mfma(-xr, yr, ar[s]);
          if constexpr (CPLX) {
            ar[s] = mfma(xi, yi, ar[s]);
            ai[s] = mfma(-xr, yi, ai[s]);
            ai[s] = mfma(-xi, yr, ai[s]);
          }
        }
      }
    }
    __syncthreads();   // every wave has read the panels it needs before any is overwritten
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (wave + 4 * s < P) store_panel(wave + 4 * s, ar[s], ai[s]);
    return;
  }
  // general mode.  Every operand is loaded where it is used: requesting the tile of the critical update (panel q + 1)
  // or the next diagonal column ahead of the barrier cost more registers (one workgroup per SIMD) than it hid latency
  double rr[16], ri[16];
  auto load_rcol = [&](int q) {
    const double* td = Rt + (long long)tile_index(q, q, P) * 256 * E;
#pragma unroll
    for (int t = 0; t < 16; ++t) ld2<CPLX>(td, t * 16 + x, rr[t], ri[t]);
  };
  for (int q = 0; q < P; ++q) {
    const int owner = q & 3, slot = q >> 2, buf = q & 1;
    if (wave == owner) {
      load_rcol(q);
      asm volatile("" ::: "memory");   // (all sixteen loads are out before the first is consumed: left alone the
                                         // scheduler pairs each load with its use - sixteen dependent trips to L2)
      double dj = rr[0];
#pragma unroll
      for (int t = 1; t < 16; ++t) dj = x == t ? rr[t] : dj;
      const double dinv = fast_rcp(dj);
      // (only rows t < x of column x take part: zeros elsewhere make the updates unconditional)
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        if (t >= x) rr[t] = ri[t] = 0.0;
      }
      v4d vr, vi;
      vr = slot == 0 ? ar[0] : slot == 1 ? ar[1] : slot == 2 ? ar[2] : ar[3];
      vi = slot == 0 ? ai[0] : slot == 1 ? ai[1] : slot == 2 ? ai[2] : ai[3];
      // column t of X is final once the columns before it have been subtracted: x_t = v_t / r_tt, read from lane t of
      // the 16-lane row together with 1 / r_tt; every lane scales its own column at the end
      auto step = [&](auto tc) {
        constexpr int t = decltype(tc)::value;
        const double dt = row_share<t>(dinv);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double br = row_share<t>(vr[r]) * dt;
          const double bi = CPLX ? row_share<t>(vi[r]) * dt : 0.0;
          vr[r] -= br * rr[t] - bi * ri[t];       // (br + i bi)(rr + i ri)
          if constexpr (CPLX) vi[r] -= br * ri[t] + bi * rr[t];
        }
      };
      step(std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 2>{});
      step(std::integral_constant<int, 3>{});
      step(std::integral_constant<int, 4>{});
      step(std::integral_constant<int, 5>{});
      step(std::integral_constant<int, 6>{});
      step(std::integral_constant<int, 7>{});
      step(std::integral_constant<int, 8>{});
      step(std::integral_constant<int, 9>{});
      step(std::integral_constant<int, 10>{});
      step(std::integral_constant<int, 11>{});
      step(std::integral_constant<int, 12>{});
      step(std::integral_constant<int, 13>{});
      step(std::integral_constant<int, 14>{});
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        vr[r] *= dinv;
        vi[r] *= dinv;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sXr[buf][(kq + 4 * r) * 17 + x] = vr[r];
        sXi[buf][(kq + 4 * r) * 17 + x] = vi[r];
      }
      store_panel(q, vr, vi);
    }
    lds_barrier();
    // updates of this wave's later panels: A_p -= X_q R(q, p)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int p = wave + 4 * s;
      if (p > q && p < P) {
        const double* tb = Rt + (long long)tile_index(q, p, P) * 256 * E;
        double yr[4], yi[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) ld2<CPLX>(tb, (4 * kk + kq) * 16 + x, yr[kk], yi[kk]);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const double xr = sXr[buf][x * 17 + 4 * kk + kq], xi = sXi[buf][x * 17 + 4 * kk + kq];
          ar[s] = mfma(-xr, yr[kk], ar[s]);
          if constexpr (CPLX) {
            ar[s] = mfma(xi, yi[kk], ar[s]);
            ai[s] = mfma(-xr, yi[kk], ai[s]);
            ai[s] = mfma(-xi, yr[kk], ai[s]);
          }
        }
      }
    }
  }
}

// scatter: U[rows[r], koff + c] = Q[r, c] (Q = the workspace after the last pass), Vt[koff + i, cols[c]] = R[i, c];
// herm: Vt[koff + c, cols[r]] = conj(Q[r, c]), U[rows[c], koff + i] = conj(R[i, c]).  The last workgroup-independent
// job: status word as a double for the host (dstat[0] = status[0]).
template <bool CPLX>
__global__ void k_cq_scatter(double* U, double* Vt, const double* __restrict__ ws, const double* __restrict__ R,
                             const double* __restrict__ R2, long long K,
                             long long ncol, const long long* __restrict__ drows, const long long* __restrict__ dcols,
                             const CqBlk* __restrict__ blks, int herm, const int* __restrict__ status, double* dstat,
                             const int* __restrict__ done, int* __restrict__ words) {
  constexpr int E = CPLX ? 2 : 1;
  const CqBlk B = blks[blockIdx.y];
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    dstat[0] = (double)status[0];
    int n2 = 0;                                      // blocks that ended after two passes (diagnostics)
    for (int b = 0; b < (int)gridDim.y; ++b) n2 += done[b] != 0;
    dstat[1] = (double)n2;
    if (words) {       // the context's running counts (mpse_block_qr_pass_stats): blocks factorised, of them in two passes
      atomicAdd(words + 2, (int)gridDim.y);
      atomicAdd(words + 3, n2);
    }
  }
  const double* q = ws + B.ws_off * E;
  const double* Rt = (done[blockIdx.y] ? R2 : R) + B.t_off * 256 * E;      // R2 R1, or R3 R2 R1
  const long long* rows = drows + B.row_off;
  const long long* cols = dcols + B.col_off;
  const int mm = B.mm, nn = B.nn, k = B.nn;
  const long long koff = B.koff;
  const long long tq = (long long)mm * k, total = tq + (long long)k * nn;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    if (t < tq) {
      double vr, vi;
      if (!herm) {
        const int c = (int)(t % k), r = (int)(t / k);
        ld2<CPLX>(q, r + (long long)c * mm, vr, vi);
        st2<CPLX>(U, rows[r] * K + koff + c, vr, vi);
      } else {
        const int r = (int)(t % mm), c = (int)(t / mm);
        ld2<CPLX>(q, r + (long long)c * mm, vr, vi);
        st2<CPLX>(Vt, (koff + c) * ncol + cols[r], vr, -vi);
      }
    } else {
      const long long u = t - tq;
      const int c = (int)(u % nn), i = (int)(u / nn);
      double vr = 0.0, vi = 0.0;
      if (i <= c) ld2<CPLX>(Rt, (long long)tile_index(i >> 4, c >> 4, B.P) * 256 + (i & 15) * 16 + (c & 15), vr, vi);
      if (!herm)
        st2<CPLX>(Vt, (koff + i) * ncol + cols[c], vr, vi);
      else
        st2<CPLX>(U, rows[c] * K + koff + i, vr, -vi);
    }
  }
}

template <bool CPLX>
int cholqr_run(mpse_ctx* ctx, double* ws, const QrBlk* blks, int nblk, const long long* drows, const long long* dcols,
               int herm, void* U, void* Vt, long long K, long long ncol, bool* ok) {
  // optimistic mode (mpse_block_qr_optimistic): no read-back here - a breakdown sets the context's sticky device flag,
  // which the caller of the sweep reads once at its end (and then repeats the step on the Householder kernels)
  MPSE_TRY(qr_words(ctx));
  int* gflag = ctx->qr_optimistic ? ctx->qr_words_dev : nullptr;
  constexpr size_t es = CPLX ? 16 : 8;
  std::vector<CqBlk> cb(nblk);
  long long part_tot = 0, t_tot = 0;
  int max_gram = 1, max_T = 1, max_nrb = 1, max_P = 1;
  long long max_sc = 1;
  long long groups = 0;
  for (int b = 0; b < nblk; ++b) {
    const int P2 = ((blks[b].nn + 15) / 16 + 1) / 2;
    groups += P2 * (P2 + 1) / 2;      // super-tiles = waves per row chunk
  }
  for (int b = 0; b < nblk; ++b) {
    CqBlk& c = cb[b];
    c.ws_off = blks[b].ws_off;
    c.mm = blks[b].mm;
    c.nn = blks[b].nn;
    c.P = (c.nn + 15) / 16;
    c.T = c.P * (c.P + 1) / 2;
    c.row_off = blks[b].row_off;
    c.col_off = blks[b].col_off;
    c.koff = blks[b].prm_off;
    c.pad = 0;
    // row chunks (one per wave): ~8 waves per compute unit over all blocks, at least 32 rows and at most 64 chunks
    int want = (int)((8LL * ctx->n_cu + groups - 1) / groups);
    if (want < 1) want = 1;
    if (want > 64) want = 64;
    int rpc = (c.mm + want - 1) / want;
    if (rpc < 32) rpc = 32;
    rpc = (rpc + 3) & ~3;
    c.rpc = rpc;
    c.nchunk = (c.mm + rpc - 1) / rpc;
    c.part_off = part_tot;
    c.t_off = t_tot;
    part_tot += (long long)((c.nchunk + 3) / 4) * c.T * 256;
    t_tot += c.T;
    {
      const int P2 = (c.P + 1) / 2, T2 = P2 * (P2 + 1) / 2;
      max_gram = std::max(max_gram, ((c.nchunk + 3) / 4) * T2);
    }
    max_T = std::max(max_T, c.T);
    max_P = std::max(max_P, c.P);
    max_nrb = std::max(max_nrb, (c.mm + 15) / 16);
    max_sc = std::max<long long>(max_sc, (long long)c.mm * c.nn + (long long)c.nn * c.nn);
  }
  // one allocation: descriptors | partial sums | G | five sets of R tiles (R1, R2, R3, R2 R1, R3 R2 R1) | tile infos | status
  const size_t db = (size_t(nblk) * sizeof(CqBlk) + 15) & ~size_t(15);
  const size_t pb = size_t(part_tot) * es, tb = size_t(t_tot) * 256 * es, ib = size_t(t_tot) * 2 * sizeof(double);
  const size_t sb = ((2 * size_t(nblk) + 1) * sizeof(int) + 15) & ~size_t(15);   // flag | trsm mode per block | done per block
  TmpBuf M(ctx);
  MPSE_TRY(M.alloc(db + pb + 6 * tb + ib + sb + 16));
  char* base = static_cast<char*>(M.p);
  MPSE_TRY(stage_h2d(ctx, base, cb.data(), size_t(nblk) * sizeof(CqBlk)));
  const CqBlk* dblk = reinterpret_cast<const CqBlk*>(base);
  double* part = reinterpret_cast<double*>(base + db);
  double* G = reinterpret_cast<double*>(base + db + pb);
  double* Rb[5];
  for (int i = 0; i < 5; ++i) Rb[i] = reinterpret_cast<double*>(base + db + pb + (1 + i) * tb);
  double* tinfo = reinterpret_cast<double*>(base + db + pb + 6 * tb);
  int* status = reinterpret_cast<int*>(base + db + pb + 6 * tb + ib);
  const int* done = status + 1 + nblk;
  double* dstat = reinterpret_cast<double*>(base + db + pb + 6 * tb + ib + sb);
  // Adaptive pass count (round 6).  theta: pass 1 shifts a pivot only when it has lost all but theta of its diagonal
  // entry (0 = the shift on the whole diagonal, the round-5 scheme); tau: pass 2 is the last pass of a block whose Gram
  // matrix deviates from the identity by n max|G - I| <= tau (0 = always three passes).  Both are decided on the device,
  // per block, from the data of the call alone; pass 3's launches return at once for the blocks that are done.
  static const double theta = [] {
    const char* e = getenv("MPSE_CHOLQR_THETA");
    return e ? atof(e) : 1e-12;
  }();
  static const double tau = [] {
    const char* e = getenv("MPSE_CHOLQR_TAU");
    return e ? atof(e) : 0.1;
  }();
  const double* racc = nullptr;
  for (int pass = 1; pass <= 3; ++pass) {
    double* Rcur = Rb[pass - 1];
    const int* dn = pass == 3 ? done : nullptr;
    hipLaunchKernelGGL((k_cq_gram<CPLX>), dim3(max_gram, nblk), dim3(256), 0, ctx->stream, (const double*)ws, part, dblk,
                       status, pass == 1 ? 2 * nblk + 1 : 0, dn);
    hipLaunchKernelGGL((k_cq_reduce<CPLX>), dim3(max_T, nblk), dim3(256), 0, ctx->stream, (const double*)part, G, tinfo, dblk,
                       dn);
    if (max_P <= 10)
      hipLaunchKernelGGL((k_cq_chol_rl<CPLX>), dim3(nblk), dim3(512), 0, ctx->stream, (const double*)G,
                         (const double*)tinfo, Rcur, dblk, status, pass, gflag, nblk, theta, tau);
    else
      hipLaunchKernelGGL((k_cq_chol<CPLX>), dim3(nblk), dim3(512), 0, ctx->stream, (const double*)G, (const double*)tinfo,
                         Rcur, dblk, status, pass, gflag, nblk, theta, tau);
    double* Rout = pass == 2 ? Rb[3] : Rb[4];
    const int rmul = pass >= 2 ? 1 : 0;
    const dim3 tg(max_nrb + (rmul ? max_T : 0), nblk);
    hipLaunchKernelGGL((k_cq_trsm<CPLX>), tg, dim3(256), 0, ctx->stream, ws, (const double*)Rcur, racc, Rout, dblk,
                       (const int*)status, max_nrb, rmul, dn);
    racc = pass == 1 ? Rcur : Rout;
  }
  int nb = (int)((max_sc + 255) / 256);
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL((k_cq_scatter<CPLX>), dim3(nb, nblk), dim3(256), 0, ctx->stream, (double*)U, (double*)Vt, (const double*)ws,
                     (const double*)Rb[4], (const double*)Rb[3], K, ncol, drows, dcols, dblk, herm, (const int*)status, dstat,
                     done, ctx->qr_words_dev);
  MPSE_HIP(ctx, hipGetLastError());
  if (gflag) {
    *ok = true;
    return MPSE_OK;
  }
  MPSE_TRY(publish_and_wait(ctx, dstat, 1, 3990));
  *ok = ctx->pinned[3990] == 0.0;
  return MPSE_OK;
}

}  // namespace

// Eligible: every block at least as tall as wide (k = nn), at most 256 columns, and the tallest block of at least 256
// rows and the widest of at least 96 columns (inside a sweep the scheme loses to the Householder chain at 64 columns:
// spin-boson chain, D = 64, 512 x 64 centres: 24.0 against 22.5 ms per evolve, profiles/r05_vs_r4_side_configs.md): the cost of the scheme is ~14 launches whose Cholesky / triangular-solve chains depend on the
// COLUMN count only (0.28 - 0.40 ms at 100 - 180 columns), the Householder chain grows with the rows (0.19 ms at 512 x 64,
// 0.32 - 0.37 ms at 256 rows x 150 columns, 0.78 ms at 2 800 rows: tools/qr_bench.py, profiles/r05_qr_cholqr.md).
// MPSE_CHOLQR=0 switches the path off, MPSE_CHOLQR=2 takes every eligible shape (tests); MPSE_CHOLQR_MINROWS moves the
// row threshold; mpse_block_qr_scheme overrides MPSE_CHOLQR per context.
bool cholqr_eligible(const mpse_ctx* ctx, const QrBlk* blks, int nblk) {
  static const int env_mode = [] {
    const char* e = getenv("MPSE_CHOLQR");
    return e ? atoi(e) : 1;
  }();
  const int mode = ctx->qr_scheme >= 0 ? ctx->qr_scheme : env_mode;
  static const int min_rows = [] {
    const char* e = getenv("MPSE_CHOLQR_MINROWS");
    return e ? atoi(e) : 256;
  }();
  if (mode == 0 || nblk <= 0) return false;
  int max_mm = 0, max_nn = 0;
  for (int b = 0; b < nblk; ++b) {
    if (blks[b].mm < blks[b].nn || blks[b].k != blks[b].nn || blks[b].nn > 256) return false;
    max_mm = std::max(max_mm, blks[b].mm);
    max_nn = std::max(max_nn, blks[b].nn);
  }
  if (mode >= 2) return true;
  // (96 columns: with the two-pass scheme of round 6 a threshold of 48 was measured again on the spin-boson chain, the
  // D = 64 Holstein chain and the 497-site FMO chain - within the scatter of the box either way, profiles/r06_small_mincols.txt)
  return max_mm >= min_rows && max_nn >= 96;
}

int cholqr_blocks(mpse_ctx* ctx, bool cplx, double* ws, const QrBlk* blks, int nblk, const long long* drows,
                  const long long* dcols, int herm, void* U, void* Vt, long long K, long long ncol, bool* ok) {
  *ok = false;
  if (cplx) return cholqr_run<true>(ctx, ws, blks, nblk, drows, dcols, herm, U, Vt, K, ncol, ok);
  return cholqr_run<false>(ctx, ws, blks, nblk, drows, dcols, herm, U, Vt, K, ncol, ok);
}
