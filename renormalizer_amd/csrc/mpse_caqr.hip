// Communication-avoiding Householder QR (CAQR) of the quantum-number blocks: tall-skinny panels of NB = 16 columns are
// factorised by a two-level TSQR tree instead of one workgroup walking every row of the block.
//
//   level 0: the active rows of a panel (rows >= j0 of the block) are cut into chunks of 256 rows; every chunk is
//            factorised by its OWN workgroup on its own CU (thread = row, the 16 panel columns of the row in
//            registers): Householder, one reduction round per column;
//   level 1: the <= 16 stacked 16 x 16 triangles (<= 4096 rows per block) are factorised by one workgroup with the
//            SAME code; its R is the panel's R.
//   Q_panel = diag(Q_chunk) . Q_root, each factor in compact-WY form  Q = I - V T V^H  (V: 256 x 16, T: 16 x 16 upper
//   triangular, LAPACK ?larft forward / columnwise): the trailing matrix receives Q_panel^H with two launches (chunk
//   level, then the top 16 rows of every chunk through the root factor), Q is formed by applying the panels in reverse
//   to the identity.  Householder throughout: exact isometries for the numerically rank-deficient centres of a
//   fixed-bond TDVP sweep (Gram / Cholesky schemes break down there).
//
// Per 16 columns the dependent chain is 2 x 16 column steps of a 256-row workgroup plus two updates, instead of 4 x
// (panel over ALL rows in one workgroup + update) launches; the block's rows are spread over up to 16 CUs.
//
// Column step (k_caqr_factor): the 16 register slots of a row form a ring [a_j .. a_15 | v_0 .. v_{j-1}]: slot 0 is the
// pivot column, the finished reflectors follow the remaining columns.  ONE reduction round (32 doubles through a
// halving butterfly: permlane32 / permlane16 swaps, then DPP inside the 16-lane rows; one barrier for the four waves)
// delivers the tail norm of the pivot, its inner products with the remaining columns (the update coefficients) AND
// with the finished reflectors (column j of V^H V, from which T follows without a second pass).  Reference semantics:
// scipy.linalg.qr / rq per block in mps/svd_qn.py:187-204.
#include <cstdlib>

#include "mpse_device.h"
#include "mpse_internal.h"

namespace {

constexpr int NB = 16;     // panel width
constexpr int CH = 256;    // rows per chunk = threads per workgroup

struct CaqrArgs {
  double* ws;              // workspaces of the blocks (column-major mm x nn each)
  double* q;               // Q buffers (mm x nq), formq only
  const QrBlk* blks;
  double2* vn;             // node storage: V (CH x NB, column-major, always complex) per node
  double2* tn;             // T (NB x NB, row-major [i][j]) per node
  double2* rs;             // stacked triangles of the chunks, per block: CH x NB column-major
  int pmax, nodes_pp;      // panels per block (capacity), nodes per panel (capacity: max chunks + 1; the root is last)
};

__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 cmulc(double2 a, double2 b) {  // conj(a) * b
  return make_double2(a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ double2 cconj(double2 a) { return make_double2(a.x, -a.y); }
// fused accumulations: four dependent-pair FMAs per complex multiply-add (a product followed by an add costs six
// FP64 issue slots and a longer dependency chain)
__device__ __forceinline__ void cmac(double2& acc, double2 a, double2 b) {     // acc += a * b
  acc.x = fma(a.x, b.x, acc.x);
  acc.y = fma(a.x, b.y, acc.y);
  acc.x = fma(-a.y, b.y, acc.x);
  acc.y = fma(a.y, b.x, acc.y);
}
__device__ __forceinline__ void cmacc(double2& acc, double2 a, double2 b) {    // acc += conj(a) * b
  acc.x = fma(a.x, b.x, acc.x);
  acc.y = fma(a.x, b.y, acc.y);
  acc.x = fma(a.y, b.y, acc.x);
  acc.y = fma(-a.y, b.x, acc.y);
}
__device__ __forceinline__ void cnmac(double2& acc, double2 a, double2 b) {    // acc -= a * b
  acc.x = fma(-a.x, b.x, acc.x);
  acc.y = fma(-a.x, b.y, acc.y);
  acc.x = fma(a.y, b.y, acc.x);
  acc.y = fma(-a.y, b.x, acc.y);
}

template <bool CPLX>
__device__ __forceinline__ double2 ld_el(const double* p, long long i) {
  if constexpr (CPLX) return reinterpret_cast<const double2*>(p)[i];
  return make_double2(p[i], 0.0);
}
template <bool CPLX>
__device__ __forceinline__ void st_el(double* p, long long i, double2 v) {
  if constexpr (CPLX)
    reinterpret_cast<double2*>(p)[i] = v;
  else
    p[i] = v.x;
}

// LDS traffic between lanes of ONE wave: order the accesses without a workgroup barrier
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Sum of 32 values over the 64 lanes of a wave by halving: lanes trade half of their values across the wave halves
// (permlane32 swap), then across the row pairs (permlane16 swap), then the 16-lane rows finish with wave_rowsum8.
// Returns in lane l the wave-wide sum of v[sum32_index(l)] (every value ends up in two lanes).
__device__ __forceinline__ int sum32_index(int lane) {
  return rowsum8_index(lane & 15) + (((lane >> 4) & 1) << 3) + ((lane >> 5) << 4);
}
__device__ __forceinline__ double swap32_sum(double a, double b) {   // lanes < 32: sum of a over {l, l+32}; others: of b
  const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double swap16_sum(double a, double b) {   // even rows: sum of a over the row pair; odd: of b
  const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double wave_sum32(const double (&v)[32], int lane) {
  double u[16], w[8];
#pragma unroll
  for (int t = 0; t < 16; ++t) u[t] = swap32_sum(v[t], v[t + 16]);
#pragma unroll
  for (int t = 0; t < 8; ++t) w[t] = swap16_sum(u[t], u[t + 8]);
  return wave_rowsum8(w, lane);
}

__device__ __forceinline__ void make_reflector(double2 alpha, double s, double2* tau, double2* scale, double* beta_out) {
  const double n2 = alpha.x * alpha.x + alpha.y * alpha.y + s;
  if ((s == 0.0 && alpha.y == 0.0) || n2 < 1e-280) {   // H = I (see mpse_qr2.hip: columns below 1e-140 are noise)
    *tau = make_double2(0.0, 0.0);
    *scale = make_double2(0.0, 0.0);
    *beta_out = alpha.x;
    return;
  }
  const bool fast = n2 < 1e280;
  const double nrm = fast ? n2 * fast_rsqrt(n2) : sqrt(n2);
  const double beta = alpha.x >= 0.0 ? -nrm : nrm;
  const double ibeta = fast ? fast_rcp(beta) : 1.0 / beta;
  *tau = make_double2((beta - alpha.x) * ibeta, -alpha.y * ibeta);
  const double dr = alpha.x - beta, di = alpha.y;
  const double den = dr * dr + di * di;
  const double iden = fast ? fast_rcp(den) : 1.0 / den;
  *scale = make_double2(dr * iden, -di * iden);
  *beta_out = beta;
}

__device__ __forceinline__ long long node_index(const CaqrArgs& a, int blk, int p, int c) {
  return ((long long)blk * a.pmax + p) * a.nodes_pp + c;
}

// ---- factorisation of one node: level 0 = chunk blockIdx.x of panel p, level 1 = the stacked triangles of the chunks
template <bool CPLX>
__global__ __launch_bounds__(CH) void k_caqr_factor(const CaqrArgs a, int p, int level) {
  constexpr int E = CPLX ? 2 : 1;
  __shared__ double s_part[2][4][32];   // double buffered by column parity: one barrier per column
  __shared__ double2 s_head[2][NB];
  __shared__ double2 s_coef[4][NB];     // wave private: fs per slot
  __shared__ double2 s_fc[4][NB];       // wave private: fc per slot (diagonal row)
  __shared__ double2 s_G[NB][NB];       // [l][j]: v_l^H v_j for l < j
  __shared__ double2 s_R[NB][NB];       // [row][col]
  __shared__ double2 s_tau[NB];
  const int blk = blockIdx.y;
  const QrBlk B = a.blks[blk];
  const int j0 = p * NB;
  if (j0 >= B.k) return;
  const int mm = B.mm;
  const int nch = (mm - j0 + CH - 1) / CH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int c;
  if (level == 0) {
    c = blockIdx.x;
    if (c >= nch) return;
  } else {
    if (nch <= 1) return;
    c = a.nodes_pp - 1;
  }
  double* ws = a.ws + B.ws_off * E;
  double2* rs = a.rs + (long long)blk * CH * NB;

  // ---- the row of this thread: 16 panel columns
  double2 ring[NB];
  if (level == 0) {
    const int r = j0 + c * CH + tid;
#pragma unroll
    for (int t = 0; t < NB; ++t)
      ring[t] = (r < mm && j0 + t < B.nn) ? ld_el<CPLX>(ws, r + (long long)(j0 + t) * mm) : make_double2(0.0, 0.0);
  } else {
#pragma unroll
    for (int t = 0; t < NB; ++t) ring[t] = tid < nch * NB ? rs[tid + t * CH] : make_double2(0.0, 0.0);
  }
  if (tid < NB * NB) {
    (&s_G[0][0])[tid] = make_double2(0.0, 0.0);
    (&s_R[0][0])[tid] = make_double2(0.0, 0.0);
  }
  __syncthreads();

#pragma unroll 1
  for (int j = 0; j < NB; ++j) {
    const int pb = j & 1;
    // --- one reduction round: |tail|^2 of the pivot and its inner products with the other 15 slots
    const bool tail = tid > j;
    const double2 pv = make_double2(tail ? ring[0].x : 0.0, tail ? ring[0].y : 0.0);
    double val[32];
    val[0] = pv.x * pv.x + pv.y * pv.y;
    val[1] = 0.0;
#pragma unroll
    for (int t = 1; t < NB; ++t) {
      const double2 d = cmulc(pv, ring[t]);
      val[2 * t] = d.x;
      val[2 * t + 1] = d.y;
    }
    if (tid == j) {   // the diagonal row publishes the heads of all slots
#pragma unroll
      for (int t = 0; t < NB; ++t) s_head[pb][t] = ring[t];
    }
    const double wsum = wave_sum32(val, lane);
    if ((lane & 8) == 0) s_part[pb][wave][sum32_index(lane)] = wsum;
    __syncthreads();
    // --- every wave derives the reflector and the coefficients on its own (no second barrier)
    double2 d = make_double2(0.0, 0.0);
    if (lane < NB) {
      const double2 p0 = reinterpret_cast<const double2*>(s_part[pb][0])[lane];
      const double2 p1 = reinterpret_cast<const double2*>(s_part[pb][1])[lane];
      const double2 p2 = reinterpret_cast<const double2*>(s_part[pb][2])[lane];
      const double2 p3 = reinterpret_cast<const double2*>(s_part[pb][3])[lane];
      d = make_double2((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y));
    }
    const double ssq = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(d.x), 0),
                                        __builtin_amdgcn_readlane(__double2loint(d.x), 0));
    const double2 alpha = s_head[pb][0];
    double2 tau, scale;
    double beta;
    make_reflector(alpha, ssq, &tau, &scale, &beta);
    if (lane >= 1 && lane < NB) {
      const double2 head = s_head[pb][lane];
      double2 fs = make_double2(0.0, 0.0), fc = make_double2(0.0, 0.0);
      if (lane <= NB - 1 - j) {          // a remaining column: coefficients of its update by H_j^H
        const double2 sc = cmulc(scale, d);
        fc = cmulc(tau, make_double2(head.x + sc.x, head.y + sc.y));
        fs = cmul(fc, scale);
      } else if (wave == 0) {            // a finished reflector v_l: g = v_l^H v_j = conj(head) + scale * conj(d)
        const double2 sd = cmul(scale, cconj(d));
        s_G[lane - (NB - j)][j] = make_double2(head.x + sd.x, sd.y - head.y);
      }
      s_coef[wave][lane] = fs;
      s_fc[wave][lane] = fc;             // (only the diagonal row reads it)
    }
    if (wave == 0 && lane == 0) s_tau[j] = tau;
    wave_lds_sync();
    // --- update of the remaining columns (finished reflectors carry a zero coefficient)
    const bool diag = tid == j;
    const double2 sv = cmul(scale, pv);                                  // v_j below the diagonal
#pragma unroll
    for (int t = 1; t < NB; ++t) cnmac(ring[t], s_coef[wave][t], pv);
    if (diag) {     // one branch, the 15 loads in flight together (slots of finished reflectors hold a zero fc)
      double2 fcv[NB];
#pragma unroll
      for (int t = 1; t < NB; ++t) fcv[t] = s_fc[wave][t];
#pragma unroll
      for (int t = 1; t < NB; ++t) {
        ring[t].x -= fcv[t].x;
        ring[t].y -= fcv[t].y;
      }
    }
    // --- the pivot column is final: R above / on the diagonal, v_j below
    if (tid <= j) s_R[tid][j] = diag ? make_double2(beta, (tau.x == 0.0 && tau.y == 0.0) ? alpha.y : 0.0) : ring[0];
    const double2 vnew = tail ? sv : make_double2(diag ? 1.0 : 0.0, 0.0);
#pragma unroll
    for (int t = 1; t < NB; ++t) ring[t - 1] = ring[t];
    ring[NB - 1] = vnew;
    // no second barrier: the next column writes the other halves of s_part / s_head, and a wave reaches the barrier
    // after that only when it is done reading this column's; s_coef / s_fc are private to a wave
    wave_lds_sync();
  }
  __syncthreads();     // s_G / s_R / s_tau complete

  // ---- T = (strict_upper(V^H V) + diag(1 / tau))^-1 by the column recurrence of ?larft; lane i owns row i
  const long long node = node_index(a, blk, p, c);
  if (wave == 0 && lane < NB) {
    double2 trow[NB];
#pragma unroll
    for (int jj = 0; jj < NB; ++jj) trow[jj] = make_double2(0.0, 0.0);
#pragma unroll
    for (int jj = 0; jj < NB; ++jj) {
      const double2 tj = s_tau[jj];
      double2 acc = make_double2(0.0, 0.0);
#pragma unroll
      for (int l = 0; l < jj; ++l) {
        const double2 u = cmul(trow[l], s_G[l][jj]);
        acc.x += u.x;
        acc.y += u.y;
      }
      const double2 m = cmul(tj, acc);
      trow[jj] = lane == jj ? tj : (lane < jj ? make_double2(-m.x, -m.y) : make_double2(0.0, 0.0));
    }
    double2* T = a.tn + node * NB * NB;
#pragma unroll
    for (int jj = 0; jj < NB; ++jj) T[lane * NB + jj] = trow[jj];
  }
  // ---- V (the ring now holds v_0 .. v_15 of this row)
  double2* V = a.vn + node * CH * NB;
#pragma unroll
  for (int t = 0; t < NB; ++t) V[t * CH + tid] = ring[t];
  // ---- R: the panel's R (a single chunk, or the root) goes to the block, a chunk's triangle to the stack
  if (tid < NB * NB) {
    const int i = tid / NB, t = tid % NB;
    const double2 rv = s_R[i][t];
    if (level == 1 || nch == 1) {
      if (j0 + i < mm && j0 + t < B.nn && i <= t) st_el<CPLX>(ws, (j0 + i) + (long long)(j0 + t) * mm, rv);
    } else {
      rs[(c * NB + i) + t * CH] = rv;
    }
  }
}

// ---- application of one node to a tile of 16 columns of X:  X <- (I - V T^H V^H) X  (adjoint: the trailing matrix
// during the factorisation) or (I - V T V^H) X (forward: formation of Q).  Level 0: X = the chunk's rows; level 1:
// X = the top 16 rows of every chunk.  Thread = (column = tid / 16, row segment = tid % 16), rows seg + 16 q.
template <bool CPLX>
__global__ __launch_bounds__(CH) void k_caqr_apply(const CaqrArgs a, int p, int level, int adjoint, int to_q, int col0) {
  constexpr int E = CPLX ? 2 : 1;
  extern __shared__ double2 s_dyn[];
  double2* sV = s_dyn;                         // [i][r]: NB x CH
  double2* sT = sV + NB * CH;                  // [i][j]
  double2* sW = sT + NB * NB;                  // [col][i]
  double2* sZ = sW + NB * NB;                  // [col][i]
  const int blk = blockIdx.z;
  const QrBlk B = a.blks[blk];
  const int j0 = p * NB;
  if (j0 >= B.k) return;
  const int mm = B.mm;
  const int nch = (mm - j0 + CH - 1) / CH;
  int c;
  if (level == 0) {
    c = blockIdx.y;
    if (c >= nch) return;
  } else {
    if (nch <= 1) return;
    c = a.nodes_pp - 1;
  }
  const int ncols = to_q ? (B.nq > B.k ? B.nq : B.k) : B.nn;
  const int cbeg = (to_q ? j0 : j0 + NB) + (col0 + (int)blockIdx.x) * NB;
  if (cbeg >= ncols) return;
  const int tid = threadIdx.x, seg = tid & 15, cl = tid >> 4;
  double* X = to_q ? a.q + B.q_off * E : a.ws + B.ws_off * E;
  const long long node = node_index(a, blk, p, c);
  const double2* V = a.vn + node * CH * NB;
  const double2* T = a.tn + node * NB * NB;
#pragma unroll
  for (int t = 0; t < NB; ++t) sV[t * CH + tid] = V[t * CH + tid];
  sT[tid] = T[tid];
  // rows of this thread
  const int col = cbeg + cl;
  const bool col_ok = col < ncols;
  double2 x[NB];
  auto row_offset = [&](int q) -> long long {      // element offset of row slot q of this thread, -1: not there
    int gr;
    bool ok;
    if (level == 0) {
      gr = j0 + c * CH + seg + 16 * q;
      ok = gr < mm;
    } else {
      gr = j0 + q * CH + seg;                      // stacked row (chunk q, row seg)
      ok = q < nch && gr < mm;
    }
    return ok && col_ok ? gr + (long long)col * mm : -1;
  };
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    const long long ro = row_offset(q);
    x[q] = ro >= 0 ? ld_el<CPLX>(X, ro) : make_double2(0.0, 0.0);
  }
  __syncthreads();
  // ---- W = V^H X: partial sums of this thread's 16 rows, then over the 16 segments of the column (one DPP row).
  // Four reflectors per pass of a rolled loop (a fully unrolled 16 x 16 body makes the compiler hoist all 256 LDS
  // loads and spill).
  const int lane = tid & 63;
#pragma unroll 1
  for (int g = 0; g < 4; ++g) {
    double2 acc[4];
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) acc[ii] = make_double2(0.0, 0.0);
    const double2* vcol = sV + g * 4 * CH + seg;
#pragma unroll
    for (int q = 0; q < NB; ++q) {          // four independent accumulation chains, four LDS loads per row slot
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) cmacc(acc[ii], vcol[ii * CH + 16 * q], x[q]);
    }
    double v8[8];
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      v8[2 * ii] = acc[ii].x;
      v8[2 * ii + 1] = acc[ii].y;
    }
    const double s = wave_rowsum8(v8, lane);
    if ((lane & 8) == 0) reinterpret_cast<double*>(sW + cl * NB)[g * 8 + rowsum8_index(lane & 15)] = s;
  }
  wave_lds_sync();
  // ---- Z = T^H W (adjoint) or T W: segment s computes z_s
  {
    double2 z = make_double2(0.0, 0.0);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const double2 wi = sW[cl * NB + i];
      if (adjoint)
        cmacc(z, sT[i * NB + seg], wi);
      else
        cmac(z, sT[seg * NB + i], wi);
    }
    sZ[cl * NB + seg] = z;
  }
  wave_lds_sync();
  // ---- X -= V Z, one reflector per pass
#pragma unroll 2
  for (int i = 0; i < NB; ++i) {
    const double2 zi = sZ[cl * NB + i];
    const double2* vcol = sV + i * CH + seg;
#pragma unroll
    for (int q = 0; q < NB; ++q) cnmac(x[q], vcol[16 * q], zi);
  }
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    const long long ro = row_offset(q);
    if (ro >= 0) st_el<CPLX>(X, ro, x[q]);
  }
}

// Q buffers <- leading columns of the identity (mm x nq per block)
template <bool CPLX>
__global__ __launch_bounds__(256) void k_caqr_eye(const CaqrArgs a) {
  constexpr int E = CPLX ? 2 : 1;
  const QrBlk B = a.blks[blockIdx.y];
  const int ncols = B.nq > B.k ? B.nq : B.k;
  double* q = a.q + B.q_off * E;
  const long long total = (long long)B.mm * ncols;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int r = (int)(t % B.mm), c = (int)(t / B.mm);
    st_el<CPLX>(q, t, make_double2(r == c ? 1.0 : 0.0, 0.0));
  }
}

template <bool CPLX>
int run_caqr(mpse_ctx* ctx, double* ws, double* q, const QrBlk* blks_host, const QrBlk* blks_dev, int nblk, bool form_q) {
  int max_mm = 0, max_nn = 0, max_k = 0, max_q = 0;
  long long max_q_el = 0;
  for (int b = 0; b < nblk; ++b) {
    const QrBlk& B = blks_host[b];
    max_mm = B.mm > max_mm ? B.mm : max_mm;
    max_nn = B.nn > max_nn ? B.nn : max_nn;
    max_k = B.k > max_k ? B.k : max_k;
    const int nq = B.nq > B.k ? B.nq : B.k;
    max_q = nq > max_q ? nq : max_q;
    max_q_el = (long long)B.mm * nq > max_q_el ? (long long)B.mm * nq : max_q_el;
  }
  CaqrArgs a;
  a.ws = ws;
  a.q = q;
  a.blks = blks_dev;
  a.pmax = (max_k + NB - 1) / NB;
  a.nodes_pp = (max_mm + CH - 1) / CH + 1;
  const size_t nnodes = size_t(nblk) * a.pmax * a.nodes_pp;
  TmpBuf VN(ctx), TN(ctx), RS(ctx);
  MPSE_TRY(VN.alloc(nnodes * CH * NB * sizeof(double2)));
  MPSE_TRY(TN.alloc(nnodes * NB * NB * sizeof(double2)));
  MPSE_TRY(RS.alloc(size_t(nblk) * CH * NB * sizeof(double2)));
  a.vn = VN.as<double2>();
  a.tn = TN.as<double2>();
  a.rs = RS.as<double2>();
  constexpr size_t lds = (size_t(NB) * CH + 3 * NB * NB) * sizeof(double2);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_caqr_apply<CPLX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  for (int p = 0; p < a.pmax; ++p) {
    const int j0 = p * NB;
    const int nchp = (max_mm - j0 + CH - 1) / CH;     // upper bound over the blocks
    if (nchp <= 0) break;
    hipLaunchKernelGGL((k_caqr_factor<CPLX>), dim3(nchp, nblk), dim3(CH), 0, ctx->stream, a, p, 0);
    if (nchp > 1) hipLaunchKernelGGL((k_caqr_factor<CPLX>), dim3(1, nblk), dim3(CH), 0, ctx->stream, a, p, 1);
    const int trailing = max_nn - j0 - NB;
    if (trailing > 0) {
      const int ntiles = (trailing + NB - 1) / NB;
      hipLaunchKernelGGL((k_caqr_apply<CPLX>), dim3(ntiles, nchp, nblk), dim3(CH), lds, ctx->stream, a, p, 0, 1, 0, 0);
      if (nchp > 1)
        hipLaunchKernelGGL((k_caqr_apply<CPLX>), dim3(ntiles, 1, nblk), dim3(CH), lds, ctx->stream, a, p, 1, 1, 0, 0);
    }
  }
  if (form_q && max_q > 0) {
    long long eb = (max_q_el + 255) / 256;
    if (eb > 4096) eb = 4096;
    hipLaunchKernelGGL((k_caqr_eye<CPLX>), dim3((unsigned)eb, nblk), dim3(256), 0, ctx->stream, a);
    for (int p = a.pmax - 1; p >= 0; --p) {
      const int j0 = p * NB;
      const int nchp = (max_mm - j0 + CH - 1) / CH;
      if (nchp <= 0 || max_q <= j0) continue;
      const int ntiles = (max_q - j0 + NB - 1) / NB;
      if (nchp > 1)
        hipLaunchKernelGGL((k_caqr_apply<CPLX>), dim3(ntiles, 1, nblk), dim3(CH), lds, ctx->stream, a, p, 1, 0, 1, 0);
      hipLaunchKernelGGL((k_caqr_apply<CPLX>), dim3(ntiles, nchp, nblk), dim3(CH), lds, ctx->stream, a, p, 0, 0, 1, 0);
    }
  }
  MPSE_HIP(ctx, hipGetLastError());
  return MPSE_OK;
}

}  // namespace

bool caqr_enabled() {   // read at every call: tests flip it inside one process
  const char* e = getenv("MPSE_QR_CAQR");
  return e && e[0] == '1';
}

// Same contract as hh_qr_batched (mpse_qr2.hip) without the per-reflector parameters: on return the upper triangle of
// every workspace holds R (k x nn) and, with form_q, q holds the mm x max(nq, k) leading columns of Q.  What is left
// below the diagonal of the workspaces is not meaningful.  Blocks of at most CAQR_MAX_ROWS rows.
int caqr_batched(mpse_ctx* ctx, bool cplx, double* ws, double* q, const QrBlk* blks_host, int nblk, bool form_q,
                 const QrBlk* blks_dev) {
  if (nblk <= 0) return MPSE_OK;
  for (int b = 0; b < nblk; ++b)
    if (blks_host[b].mm > CAQR_MAX_ROWS) return mpse_fail(ctx, MPSE_ERR_SHAPE, "caqr_batched: block too tall");
  TmpBuf DB(ctx);
  if (!blks_dev) {
    MPSE_TRY(DB.alloc(size_t(nblk) * sizeof(QrBlk)));
    MPSE_TRY(stage_h2d(ctx, DB.p, blks_host, size_t(nblk) * sizeof(QrBlk)));
    blks_dev = DB.as<QrBlk>();
  }
  if (cplx) return run_caqr<true>(ctx, ws, q, blks_host, blks_dev, nblk, form_q);
  return run_caqr<false>(ctx, ws, q, blks_host, blks_dev, nblk, form_q);
}
