// Batched, panel-blocked Householder QR on column-major workspaces.
//
// All quantum-number blocks of a decomposition are factorised by the SAME launches (grid.y =
// block), and the dependent chain of reflectors is cut into panels of NB columns:
//   k_hh_panel       : ONE workgroup (256 or 512 threads by block height) per block keeps the
//                      m x NB panel in registers and factorises it: one block-wide reduction per
//                      column delivers the column norm and all inner products with the remaining
//                      panel columns at once (eight values through a halving DPP butterfly), the
//                      dot and update phases are branch free;
//   k_hh_apply_panel : one 256-thread workgroup per trailing column keeps that column in
//                      registers and applies the NB new reflectors back to back (reflector tails
//                      stream from L2);
//   k_hh_formq_b     : one workgroup per column of Q, column in registers, reflectors applied
//                      in reverse.
// Per NB columns this costs 2 launches instead of NB, and the trailing matrix is read and written
// once per panel instead of once per reflector.  The default path applies the panels in compact-WY form
// (k_hh_apply_wy / k_hh_formq_wy: one reduction round per panel instead of one per reflector); the inner products
// of the reflector tails that T needs fall out of the panel kernel's own reduction rounds (retired columns stay in
// its register ring), so the applying workgroups only reduce V^H x.  Blocks of at most 1024 rows run one launch
// per panel (k_hh_step: the panel's workgroup first gives its columns the previous panel's reflectors, the trailing
// update of that panel runs beside it); taller blocks keep the two launches: there the update of the four panel
// columns alone fills its compute unit's FP64 pipe for longer than the second launch costs.  Conventions are LAPACK's (?geqr2 / ?ung2r):
// H_j = I - tau_j v_j v_j^H, v_j = (0.., 1, scale_j * tail_j), tails stored UNSCALED below the
// diagonal, R on and above it.
#include <cstdlib>

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

__device__ __forceinline__ void make_reflector(double2 alpha, double s, HhParam* p, double* beta_out) {
  const double n2 = alpha.x * alpha.x + alpha.y * alpha.y + s;
  if ((s == 0.0 && alpha.y == 0.0) || n2 < 1e-280) {
    // H = I.  Besides the exact case this covers columns whose squared norm underflows (|column| < 1e-140): ?larfg
    // would rescale and iterate; here the tail is dropped, an absolute perturbation below 1e-140 - the sweeps
    // only ever see such columns as padding noise next to O(1) data - and no 0/0 can arise.
    p->tau_re = p->tau_im = p->scale_re = p->scale_im = 0.0;
    *beta_out = alpha.x;
    return;
  }
  double nrm, ibeta, iden;
  const bool fast = n2 < 1e280;
  // hardware estimates + Newton (mpse_device.h) instead of three IEEE sequences on the critical path of a column
  nrm = fast ? n2 * fast_rsqrt(n2) : sqrt(n2);
  const double beta = alpha.x >= 0.0 ? -nrm : nrm;
  ibeta = fast ? fast_rcp(beta) : 1.0 / beta;
  p->tau_re = (beta - alpha.x) * ibeta;
  p->tau_im = -alpha.y * ibeta;
  const double dr = alpha.x - beta, di = alpha.y;
  const double den = dr * dr + di * di;
  iden = fast ? fast_rcp(den) : 1.0 / den;
  p->scale_re = dr * iden;
  p->scale_im = -di * iden;
  *beta_out = beta;
}

// ---- panel factorisation: NT threads, thread owns rows tid + NT q (q < RPT), NB columns.
// The column loop is NOT unrolled (an 8x unrolled body is ~90 KB of code and the kernel becomes
// instruction-fetch bound): the pivot is always register column 0 and the panel is rotated by one
// column after each reflector, so every iteration runs the same code with static register indices.
template <int NT, int NV8>
__device__ __forceinline__ void wy_reduce(double (&vals)[NV8 * 8], double (*s_part)[48], int tid);
template <bool CPLX, int NT, int RPT>
__device__ __forceinline__ void wy_load_v(const double* a, const HhParam* prm, int mm, int j0, int nbb, int tid,
                                          double2 (&u)[4][RPT], double2 (&tau)[4], double2 (&g)[6]);
__device__ __forceinline__ void wy_load_g(const HhParam* prm, int j0, int nbb, double2 (&g)[6]);
__device__ __forceinline__ double2 wy_g(const double2 (&g)[6], int i, int l);

// LOOK: the panel first receives the reflectors of the PREVIOUS panel (compact-WY, one reduction round for its NB
// columns) - the trailing update of that panel runs in other workgroups of the same launch and skips these columns.
template <bool CPLX, int NT, int RPT, int NB, bool LOOK>
__device__ __forceinline__ void hh_panel_role(double* ws_base, const QrBlk& B, HhParam* prm_base, int j0) {
  constexpr int E = Cx<CPLX>::E;
  constexpr int NV = 2 * NB;  // [0] = |tail|^2, [1] unused, then (re, im) of the dot with panel column t >= 1
  // partial sums in LDS: one row of NV values per 16-lane row of every wave (NB == 4: the halving butterfly
  // stops at the row, measured cheaper than finishing the wave with row swaps / bpermutes) or per wave
  constexpr bool ROWPART = (NB == 4) && (NT <= 256);   // one wave per SIMD: stop the butterfly at the 16-lane row
  constexpr int NPART = ROWPART ? NT / 16 : NT / 64;
  __shared__ double s_part[NPART][NV];
  __shared__ double s_f[NB][4];
  __shared__ double s_par[2];
  __shared__ double s_head[2 * NB];
  __shared__ double s_scale[2 * NB];
  if (j0 >= B.k) return;
  const int nbb = min(NB, B.k - j0);
  const int mm = B.mm;
  double* a = ws_base + B.ws_off * E;
  HhParam* prm = prm_base + B.prm_off;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int jp = j0 - NB;                              // previous panel: complete (NB reflectors) whenever j0 < k
  const int rlo = (LOOK && jp >= 0) ? jp : j0;

  double2 x[RPT][NB];
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int r = tid + NT * q;
#pragma unroll
    for (int c = 0; c < NB; ++c)
      x[q][c] = (r >= rlo && r < mm && c < nbb) ? Cx<CPLX>::ld(a, r + (long long)(j0 + c) * mm) : make_double2(0.0, 0.0);
  }
  if constexpr (LOOK) {
    static_assert(NB == 4, "the compact-WY helpers hold four reflectors");
    if (jp >= 0) {
      // eight waves (two per SIMD, 256 registers each) cannot hold the previous panel's reflectors next to their
      // own columns and the 48 partial sums: they stream the reflector rows twice (L2) instead
      constexpr bool STREAM = NT > 256 && RPT >= 6;
      __shared__ double s_wy[NT / 64][32];
      __shared__ double s_z[4][8];
      double2 u[STREAM ? 1 : 4][STREAM ? 1 : RPT], tau[4], sc[4], g[6];
      if constexpr (!STREAM) {
        wy_load_v<CPLX, NT, RPT>(a, prm, mm, jp, 4, tid, u, tau, g);
      } else {
        wy_load_g(prm, jp, 4, g);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          tau[i] = make_double2(prm[jp + i].tau_re, prm[jp + i].tau_im);
          sc[i] = make_double2(prm[jp + i].scale_re, prm[jp + i].scale_im);
        }
      }
      auto row_u = [&](int q, double2 (&ur)[4]) {
        const int r = tid + NT * q;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if constexpr (!STREAM) {
            ur[i] = u[i][q];
          } else {
            double2 t = (r > jp + i && r < mm) ? Cx<CPLX>::ld(a, r + (long long)(jp + i) * mm) : make_double2(0.0, 0.0);
            t = cmul(sc[i], t);
            if (r == jp + i) t = make_double2(1.0, 0.0);
            ur[i] = t;
          }
        }
      };
      double vals[32];
#pragma unroll
      for (int t = 0; t < 32; ++t) vals[t] = 0.0;
#pragma unroll
      for (int q = 0; q < RPT; ++q) {
        double2 ur[4];
        row_u(q, ur);
#pragma unroll
        for (int c = 0; c < NB; ++c)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const double2 t = cmulc(ur[i], x[q][c]);
            vals[8 * c + 2 * i] += t.x;
            vals[8 * c + 2 * i + 1] += t.y;
          }
      }
      // wave partials -> LDS; wave 0 sums them (lane t owns value t), lane c < 4 solves z = T^H w for panel column c
      // with the totals of lanes 8c .. 8c+7, and the 16 coefficients go back through LDS: the other waves read 32
      // doubles instead of all partial sums, and nobody repeats the triangular solves
      {
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          double v8[8];
#pragma unroll
          for (int t = 0; t < 8; ++t) v8[t] = vals[gq * 8 + t];
          const double w = wave_sum8(v8, lane);
          if (lane < 8) s_wy[wave][gq * 8 + rowsum8_index(lane)] = w;
        }
        lds_barrier();
        if (wave == 0) {
          double tot = 0.0;
          if (lane < 32) {
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) tot += s_wy[w][lane];
          }
          const int c = lane & 3;
          double2 zz[4];
#pragma unroll
          for (int l = 0; l < 4; ++l) {
            double2 w = make_double2(__shfl(tot, 8 * c + 2 * l, 64), __shfl(tot, 8 * c + 2 * l + 1, 64));
#pragma unroll
            for (int i = 0; i < l; ++i) {
              const double2 t = cmulc(wy_g(g, i, l), zz[i]);
              w.x -= t.x;
              w.y -= t.y;
            }
            zz[l] = cmulc(tau[l], w);
          }
          if (lane < 4) {
#pragma unroll
            for (int l = 0; l < 4; ++l) {
              s_z[c][2 * l] = zz[l].x;
              s_z[c][2 * l + 1] = zz[l].y;
            }
          }
        }
        lds_barrier();
      }
      double2 z[NB][4];
#pragma unroll
      for (int c = 0; c < NB; ++c)
#pragma unroll
        for (int l = 0; l < 4; ++l) z[c][l] = make_double2(s_z[c][2 * l], s_z[c][2 * l + 1]);
#pragma unroll
      for (int q = 0; q < RPT; ++q) {
        const int r = tid + NT * q;
        double2 ur[4];
        row_u(q, ur);
#pragma unroll
        for (int c = 0; c < NB; ++c) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const double2 t = cmul(ur[i], z[c][i]);
            x[q][c].x -= t.x;
            x[q][c].y -= t.y;
          }
          // rows of the previous panel are final entries of R
          if (r >= jp && r < j0 && c < nbb) Cx<CPLX>::st(a, r + (long long)(j0 + c) * mm, x[q][c]);
        }
      }
    }
  }

#pragma unroll 1
  for (int jj = 0; jj < nbb; ++jj) {
    const int j = j0 + jj;
    // --- one reduction: tail norm of the pivot column and its inner products with the other panel columns
    double val[NV];
#pragma unroll
    for (int t = 0; t < NV; ++t) val[t] = 0.0;
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      const int r = tid + NT * q;
      // branch-free: rows outside the tail contribute zeros (a divergent region per row costs more than the flops)
      const bool tail = (r > j) && (r < mm);
      const double2 v = make_double2(tail ? x[q][0].x : 0.0, tail ? x[q][0].y : 0.0);
      val[0] += v.x * v.x + v.y * v.y;
#pragma unroll
      for (int t = 1; t < NB; ++t) {
        const double2 t2 = cmulc(v, x[q][t]);
        val[2 * t] += t2.x;
        val[2 * t + 1] += t2.y;
      }
      if (r == j) {  // owner of the diagonal row publishes alpha and the heads of the other columns
#pragma unroll
        for (int t = 0; t < NB; ++t) {
          s_head[2 * t] = x[q][t].x;
          s_head[2 * t + 1] = x[q][t].y;
        }
      }
    }
    if constexpr (ROWPART) {
      // all eight values reduced together over each 16-lane row (halving butterfly), one partial per row
      const double w = wave_rowsum8(val, lane);
      if ((lane & 8) == 0) s_part[wave * 4 + (lane >> 4)][rowsum8_index(lane)] = w;
    } else if constexpr (NB == 4) {
      // several waves per SIMD: finish the wave with the row-swap steps, wave 0 then sums NT/64 partials only
      const double w = wave_sum8(val, lane);
      if (lane < 8) s_part[wave][rowsum8_index(lane)] = w;
    } else {
      // wave-level sums on the VALU (DPP), one partial per wave and value
#pragma unroll
      for (int t = 0; t < NV; ++t) {
        const double w = wave_sum(val[t]);
        if (lane == 0) s_part[wave][t] = w;
      }
    }
    lds_barrier();  // (A) partial sums and the diagonal row are in LDS (the column stores stay in flight)
    // --- the scalar work (f64 sqrt / divisions) is done once, by wave 0
    if (wave == 0) {
      double tot = 0.0;
      if (lane < NV) {
        double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;  // four chains: the adds are latency bound
        static_assert(NPART % 4 == 0, "partial rows are summed four at a time");
#pragma unroll
        for (int w = 0; w < NPART; w += 4) {
          t0 += s_part[w][lane];
          t1 += s_part[w + 1][lane];
          t2 += s_part[w + 2][lane];
          t3 += s_part[w + 3][lane];
        }
        tot = (t0 + t1) + (t2 + t3);
      }
      const double ssq = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(tot), 0),
                                          __builtin_amdgcn_readlane(__double2loint(tot), 0));
      HhParam p;
      double beta;
      const double2 alpha = make_double2(s_head[0], s_head[1]);
      make_reflector(alpha, ssq, &p, &beta);
      const double2 tau = make_double2(p.tau_re, p.tau_im), scale = make_double2(p.scale_re, p.scale_im);
      // lane t (1 <= t < NB) owns ring column t = panel column (jj + t) mod NB.  A column still to be factorised
      // turns its inner product into its update coefficients; a retired one (the tail of reflector i < jj, kept in
      // the ring) turns it into u_i^H u_jj = conj(scale_i) (conj(v_i[j]) + scale conj(v^H v_i)) for the compact-WY
      // applications; columns past the block's rank stay zero
      const int t = (lane >= 1 && lane < NB) ? lane : 1;
      const double2 d = make_double2(__shfl(tot, 2 * t, 64), __shfl(tot, 2 * t + 1, 64));
      if (lane >= 1 && lane < NB) {
        const double2 head = make_double2(s_head[2 * t], s_head[2 * t + 1]);
        double2 fc = make_double2(0.0, 0.0), fsc = make_double2(0.0, 0.0);
        if (t < nbb - jj) {
          const double2 sc = cmulc(scale, d);
          fc = cmulc(tau, make_double2(head.x + sc.x, head.y + sc.y));
          fsc = cmul(fc, scale);
        } else if (t >= NB - jj) {
          const int i = t - (NB - jj);
          const double2 sci = make_double2(s_scale[2 * i], s_scale[2 * i + 1]);
          const double2 sd = cmul(scale, make_double2(d.x, -d.y));
          const double2 g = cmulc(sci, make_double2(head.x + sd.x, sd.y - head.y));
          prm[j].g[2 * i] = g.x;
          prm[j].g[2 * i + 1] = g.y;
        }
        s_f[t][0] = fc.x;
        s_f[t][1] = fc.y;
        s_f[t][2] = fsc.x;
        s_f[t][3] = fsc.y;
      }
      if (lane == 0) {
        prm[j].tau_re = p.tau_re;
        prm[j].tau_im = p.tau_im;
        prm[j].scale_re = p.scale_re;
        prm[j].scale_im = p.scale_im;
        s_scale[2 * jj] = p.scale_re;     // read by later columns only, i.e. after the barriers below
        s_scale[2 * jj + 1] = p.scale_im;
        s_par[0] = beta;
        s_par[1] = (p.tau_re == 0.0 && p.tau_im == 0.0) ? alpha.y : 0.0;
      }
    }
    lds_barrier();  // (B) coefficients published
    const double beta = s_par[0], diag_im = s_par[1];
    double2 fc[NB], fs[NB];  // all coefficients fetched from LDS once, before any use
#pragma unroll
    for (int t = 1; t < NB; ++t) {
      fc[t] = make_double2(s_f[t][0], s_f[t][1]);
      fs[t] = make_double2(s_f[t][2], s_f[t][3]);
    }
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      const int r = tid + NT * q;
      // tail rows: -= (f scale) v, done for every row with v zeroed outside the tail (one select per row instead of
      // one divergent region per row and column); the single diagonal row is the only real branch left
      const bool tail = (r > j) && (r < mm);
      const double2 vt = make_double2(tail ? x[q][0].x : 0.0, tail ? x[q][0].y : 0.0);
#pragma unroll
      for (int t = 1; t < NB; ++t) {
        const double2 t2 = cmul(fs[t], vt);
        x[q][t].x -= t2.x;
        x[q][t].y -= t2.y;
      }
      if (r == j) {
        x[q][0] = make_double2(beta, diag_im);
#pragma unroll
        for (int t = 1; t < NB; ++t) {
          x[q][t].x -= fc[t].x;
          x[q][t].y -= fc[t].y;
        }
      }
      // the pivot column is final: store it, then rotate the ring by one column (the retired column stays in
      // registers: its tail meets the later pivots of the panel in the same reduction round)
      if (r >= j0 && r < mm) Cx<CPLX>::st(a, r + (long long)j * mm, x[q][0]);
      const double2 done = x[q][0];
#pragma unroll
      for (int t = 1; t < NB; ++t) x[q][t - 1] = x[q][t];
      x[q][NB - 1] = done;
    }
    // no barrier here: the next column writes s_part / s_head, which nobody reads after (B); s_f / s_par are
    // rewritten only after the next (A), which every thread reaches after finishing its update above
  }
}

template <bool CPLX, int NT, int RPT, int NB>
__global__ __launch_bounds__(NT) void k_hh_panel(double* ws_base, const QrBlk* __restrict__ blks, HhParam* prm_base,
                                                   int j0) {
  const QrBlk B = blks[blockIdx.x];
  hh_panel_role<CPLX, NT, RPT, NB, false>(ws_base, B, prm_base, j0);
}

// ---- apply reflectors j0 .. j0+nbb-1 (H^H, in order) to trailing column c; 256 threads, rows tid + 256 q
template <bool CPLX, int RPT>
__global__ __launch_bounds__(256) void k_hh_apply_panel(double* ws_base, const QrBlk* __restrict__ blks,
                                                        const HhParam* __restrict__ prm_base, int j0, int nb) {
  constexpr int E = Cx<CPLX>::E;
  __shared__ double s_head[2];
  const QrBlk B = blks[blockIdx.y];
  if (j0 >= B.k) return;
  const int nbb = min(nb, B.k - j0);
  const int c = j0 + nbb + blockIdx.x;
  if (c >= B.nn) return;
  const int mm = B.mm;
  double* a = ws_base + B.ws_off * E;
  const HhParam* prm = prm_base + B.prm_off;
  double* col = a + (long long)c * mm * E;
  const int tid = threadIdx.x;
  double2 x[RPT];
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int r = tid + 256 * q;
    x[q] = (r >= j0 && r < mm) ? Cx<CPLX>::ld(col, r) : make_double2(0.0, 0.0);
  }
  // reflector tails stream from L2: the loads for reflector jj+1 are issued before the reduction of jj
  double2 v[RPT], vn[RPT];
  auto load_v = [&](int jj, double2* dst) {
    const int j = j0 + jj;
    const double* vj = a + (long long)j * mm * E;
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      const int r = tid + 256 * q;
      dst[q] = (r > j && r < mm) ? Cx<CPLX>::ld(vj, r) : make_double2(0.0, 0.0);
    }
  };
  load_v(0, v);
  for (int jj = 0; jj < nbb; ++jj) {
    const int j = j0 + jj;
    if (jj + 1 < nbb) load_v(jj + 1, vn);
    const HhParam p = prm[j];
    const double2 tau = make_double2(p.tau_re, p.tau_im), scale = make_double2(p.scale_re, p.scale_im);
    if (tau.x != 0.0 || tau.y != 0.0) {  // block-uniform
      double dr = 0, di = 0;
#pragma unroll
      for (int q = 0; q < RPT; ++q) {
        const int r = tid + 256 * q;
        const double2 t2 = cmulc(v[q], x[q]);
        dr += t2.x;
        di += t2.y;
        if (r == j) {
          s_head[0] = x[q].x;
          s_head[1] = x[q].y;
        }
      }
      block_allsum2(dr, di);  // two barriers inside: s_head is visible afterwards
      const double2 head = make_double2(s_head[0], s_head[1]);
      const double2 sc = cmulc(scale, make_double2(dr, di));
      const double2 f = cmulc(tau, make_double2(head.x + sc.x, head.y + sc.y));
      const double2 fs = cmul(f, scale);
#pragma unroll
      for (int q = 0; q < RPT; ++q) {
        const int r = tid + 256 * q;
        const double2 t2 = cmul(fs, v[q]);  // v is zero outside the tail
        x[q].x -= t2.x;
        x[q].y -= t2.y;
        if (r == j) {
          x[q].x -= f.x;
          x[q].y -= f.y;
        }
      }
      __syncthreads();  // s_head reused by the next reflector
    }
#pragma unroll
    for (int q = 0; q < RPT; ++q) v[q] = vn[q];
  }
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int r = tid + 256 * q;
    if (r >= j0 && r < mm) Cx<CPLX>::st(col, r, x[q]);
  }
}

// ---- column c of Q = H_0 ... H_c e_c ; 256 threads, column in registers
template <bool CPLX, int RPT>
__global__ __launch_bounds__(256) void k_hh_formq_b(double* q_base, const double* __restrict__ ws_base,
                                                    const QrBlk* __restrict__ blks,
                                                    const HhParam* __restrict__ prm_base) {
  constexpr int E = Cx<CPLX>::E;
  __shared__ double s_head[2];
  const QrBlk B = blks[blockIdx.y];
  // A workgroup runs on die (launch position mod 8).  Column c starts at panel c / 4 and walks down: neighbouring
  // columns are at the same panel at the same time.  Every die therefore takes a contiguous range of columns, so
  // that its workgroups read the same reflector panel while it is in that die's L2 (with the columns dealt round
  // robin, a die held workgroups at 16 different panels and streamed the reflectors 456 MB per launch through).
  int c = blockIdx.x;
  if ((gridDim.x & 7) == 0) c = (c & 7) * (gridDim.x >> 3) + (c >> 3);
  if (c >= (B.nq > B.k ? B.nq : B.k)) return;
  const int mm = B.mm;
  const double* a = ws_base + B.ws_off * E;
  const HhParam* prm = prm_base + B.prm_off;
  double* col = q_base + (B.q_off + (long long)c * mm) * E;
  const int tid = threadIdx.x;
  double2 x[RPT];
#pragma unroll
  for (int q = 0; q < RPT; ++q) x[q] = make_double2((tid + 256 * q) == c ? 1.0 : 0.0, 0.0);
  // columns c >= k (orthogonal complement) receive all reflectors
  for (int j = (c < B.k ? c : B.k - 1); j >= 0; --j) {
    const HhParam p = prm[j];
    const double2 tau = make_double2(p.tau_re, p.tau_im), scale = make_double2(p.scale_re, p.scale_im);
    if (tau.x == 0.0 && tau.y == 0.0) continue;
    const double* vj = a + (long long)j * mm * E;
    double2 v[RPT];
    double dr = 0, di = 0;
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      const int r = tid + 256 * q;
      v[q] = (r > j && r < mm) ? Cx<CPLX>::ld(vj, r) : make_double2(0.0, 0.0);
      const double2 t2 = cmulc(v[q], x[q]);
      dr += t2.x;
      di += t2.y;
      if (r == j) {
        s_head[0] = x[q].x;
        s_head[1] = x[q].y;
      }
    }
    block_allsum2(dr, di);
    const double2 head = make_double2(s_head[0], s_head[1]);
    const double2 sc = cmulc(scale, make_double2(dr, di));
    const double2 f = cmul(tau, make_double2(head.x + sc.x, head.y + sc.y));  // H, not H^H
    const double2 fs = cmul(f, scale);
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      const int r = tid + 256 * q;
      const double2 t2 = cmul(fs, v[q]);
      x[q].x -= t2.x;
      x[q].y -= t2.y;
      if (r == j) {
        x[q].x -= f.x;
        x[q].y -= f.y;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int r = tid + 256 * q;
    if (r < mm) Cx<CPLX>::st(col, r, x[q]);
  }
}

// ---- compact-WY forms: the NB (<= 4) reflectors of a panel act through  Q_p = H_1 .. H_nb = I - V T V^H  with
// T^{-1} = strict_upper(V^H V) + diag(1 / tau)  (LAPACK ?larft, forward / columnwise).  One reduction round delivers
// V^H x for the columns a workgroup owns; the six inner products of V^H V come from the panel kernel, which meets
// every pair of tails in its own reduction rounds (HhParam::g); the triangular solves
// need no division: z = T^H w is  z_l = conj(tau_l) (w_l - sum_{i<l} conj(G_il) z_i),  z = T w is
// z_i = tau_i (w_i - sum_{l>i} G_il z_l).  Per panel this is one block-wide round instead of one per reflector.
template <int NT, int NV8>
__device__ __forceinline__ void wy_reduce(double (&vals)[NV8 * 8], double (*s_part)[48], int tid) {
  // vals: NV8 groups of eight partial sums per thread -> totals of all 8 NV8 values in every thread
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int g = 0; g < NV8; ++g) {
    double v8[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) v8[t] = vals[g * 8 + t];
    const double w = wave_sum8(v8, lane);
    if (lane < 8) s_part[wave][g * 8 + rowsum8_index(lane)] = w;
  }
  lds_barrier();
#pragma unroll
  for (int t = 0; t < NV8 * 8; ++t) {
    double a = 0.0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) a += s_part[w][t];
    vals[t] = a;
  }
}

// G_il = u_i^H u_l (i < l) of the panel at j0, left in HhParam::g of reflector l by the panel kernel; order
// 01, 02, 03, 12, 13, 23.  Reflectors past the block's rank have tau = 0 and never meet their G.
__device__ __forceinline__ void wy_load_g(const HhParam* prm, int j0, int nbb, double2 (&g)[6]) {
#pragma unroll
  for (int l = 1; l < 4; ++l) {
    const bool on = l < nbb;
    const HhParam* p = prm + (on ? j0 + l : j0);
#pragma unroll
    for (int i = 0; i < l; ++i) {
      const int o = i == 0 ? l - 1 : i == 1 ? l + 1 : 5;
      g[o] = on ? make_double2(p->g[2 * i], p->g[2 * i + 1]) : make_double2(0.0, 0.0);
    }
  }
}
__device__ __forceinline__ double2 wy_g(const double2 (&g)[6], int i, int l) {  // i < l
  return g[i == 0 ? l - 1 : i == 1 ? l + 1 : 5];
}

// full reflector vectors of a panel for the rows of this thread: u_i[r] = scale_i * tail_i[r] (r > j_i), 1 (r == j_i), 0.
// Two halves, so that the tails of the next panel can be in flight while the current one is applied.
template <bool CPLX, int NT, int RPT>
__device__ __forceinline__ void wy_load_raw(const double* a, int mm, int j0, int nbb, int tid, double2 (&u)[4][RPT]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bool on = i < nbb;
    const double* vj = a + (long long)(j0 + (on ? i : 0)) * mm * Cx<CPLX>::E;
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      const int r = tid + NT * q;
      u[i][q] = (on && r > j0 + i && r < mm) ? Cx<CPLX>::ld(vj, r) : make_double2(0.0, 0.0);
    }
  }
}
template <int NT, int RPT>
__device__ __forceinline__ void wy_finish(const HhParam* prm, int j0, int nbb, int tid, double2 (&u)[4][RPT],
                                          double2 (&tau)[4], double2 (&g)[6]) {
  wy_load_g(prm, j0, nbb, g);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bool on = i < nbb;
    const HhParam* p = prm + (on ? j0 + i : j0);
    tau[i] = on ? make_double2(p->tau_re, p->tau_im) : make_double2(0.0, 0.0);
    const double2 sc = make_double2(p->scale_re, p->scale_im);
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      const int r = tid + NT * q;
      double2 t = cmul(sc, u[i][q]);
      if (on && r == j0 + i) t = make_double2(1.0, 0.0);
      u[i][q] = t;
    }
  }
}
template <bool CPLX, int NT, int RPT>
__device__ __forceinline__ void wy_load_v(const double* a, const HhParam* prm, int mm, int j0, int nbb, int tid,
                                          double2 (&u)[4][RPT], double2 (&tau)[4], double2 (&g)[6]) {
  wy_load_raw<CPLX, NT, RPT>(a, mm, j0, nbb, tid, u);
  wy_finish<NT, RPT>(prm, j0, nbb, tid, u, tau, g);
}

// trailing update: columns c0 .. c0+NC-1 receive Q_p^H = I - V T^H V^H
template <bool CPLX, int NT, int RPT, int NC>
__device__ __forceinline__ void hh_apply_role(double* ws_base, const QrBlk& B, const HhParam* __restrict__ prm_base,
                                              int j0, int nbb, int c0) {
  constexpr int E = Cx<CPLX>::E;
  __shared__ double s_part[NT / 64][48];
  if (c0 >= B.nn) return;
  const int mm = B.mm, tid = threadIdx.x;
  double* a = ws_base + B.ws_off * E;
  double2 u[4][RPT], tau[4], g[6];
  wy_load_v<CPLX, NT, RPT>(a, prm_base + B.prm_off, mm, j0, nbb, tid, u, tau, g);
  double2 x[NC][RPT];
#pragma unroll
  for (int cc = 0; cc < NC; ++cc) {
    const double* col = a + (long long)(c0 + cc) * mm * E;
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      const int r = tid + NT * q;
      x[cc][q] = (c0 + cc < B.nn && r >= j0 && r < mm) ? Cx<CPLX>::ld(col, r) : make_double2(0.0, 0.0);
    }
  }
  // one round: w_i = u_i^H x per column (8 doubles each)
  constexpr int NV8 = NC;
  double vals[NV8 * 8];
#pragma unroll
  for (int t = 0; t < NV8 * 8; ++t) vals[t] = 0.0;
#pragma unroll
  for (int cc = 0; cc < NC; ++cc)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < RPT; ++q) {
        const double2 t = cmulc(u[i][q], x[cc][q]);
        vals[8 * cc + 2 * i] += t.x;
        vals[8 * cc + 2 * i + 1] += t.y;
      }
  wy_reduce<NT, NV8>(vals, s_part, tid);
#pragma unroll
  for (int cc = 0; cc < NC; ++cc) {
    if (c0 + cc >= B.nn) continue;
    double2 z[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      double2 w = make_double2(vals[8 * cc + 2 * l], vals[8 * cc + 2 * l + 1]);
#pragma unroll
      for (int i = 0; i < l; ++i) {
        const double2 t = cmulc(wy_g(g, i, l), z[i]);
        w.x -= t.x;
        w.y -= t.y;
      }
      z[l] = cmulc(tau[l], w);
    }
    double* col = a + (long long)(c0 + cc) * mm * E;
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      const int r = tid + NT * q;
      double2 y = x[cc][q];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const double2 t = cmul(u[i][q], z[i]);
        y.x -= t.x;
        y.y -= t.y;
      }
      if (r >= j0 && r < mm) Cx<CPLX>::st(col, r, y);
    }
  }
}

template <bool CPLX, int NT, int RPT, int NC>
__global__ __launch_bounds__(NT) void k_hh_apply_wy(double* ws_base, const QrBlk* __restrict__ blks,
                                                      const HhParam* __restrict__ prm_base, int j0, int nb) {
  const QrBlk B = blks[blockIdx.y];
  if (j0 >= B.k) return;
  const int nbb = min(nb, B.k - j0);
  hh_apply_role<CPLX, NT, RPT, NC>(ws_base, B, prm_base, j0, nbb, j0 + nbb + blockIdx.x * NC);
}

// One launch per panel with look-ahead: workgroup 0 of a block brings the panel at j0 up to date with the previous
// panel's reflectors and factorises it, the others give the previous panel's update to the columns right of the
// panel.  Both read only what the previous launch wrote, so the trailing update leaves the dependent chain: a QR of
// k columns is k / 4 launches as long as its slower role (the panel) instead of k / 2 alternating ones.
template <bool CPLX, int NT, int RPT, int NC>
__global__ __launch_bounds__(NT) void k_hh_step(double* ws_base, const QrBlk* __restrict__ blks, HhParam* prm_base,
                                                  int j0) {
  const QrBlk B = blks[blockIdx.y];
  if (blockIdx.x == 0) {
    hh_panel_role<CPLX, NT, RPT, 4, true>(ws_base, B, prm_base, j0);
    return;
  }
  const int jp = j0 - 4;
  if (jp < 0 || jp >= B.k) return;
  const int nbb = min(4, B.k - jp);
  const int cstart = j0 < B.k ? j0 + min(4, B.k - j0) : jp + nbb;
  hh_apply_role<CPLX, NT, RPT, NC>(ws_base, B, prm_base, jp, nbb, cstart + (blockIdx.x - 1) * NC);
}

// column c of Q = Q_0 Q_1 .. e_c, panels applied in reverse, Q_p = I - V T V^H; columns c >= k complete the basis
template <bool CPLX, int NT, int RPT>
__global__ __launch_bounds__(NT) void k_hh_formq_wy(double* q_base, const double* __restrict__ ws_base,
                                                      const QrBlk* __restrict__ blks,
                                                      const HhParam* __restrict__ prm_base) {
  constexpr int E = Cx<CPLX>::E;
  __shared__ double s_part[2][NT / 64][48];
  const QrBlk B = blks[blockIdx.y];
  // A workgroup runs on die (launch position mod 8).  Column c starts at panel c / 4 and walks down: neighbouring
  // columns are at the same panel at the same time.  Every die therefore takes a contiguous range of columns, so
  // that its workgroups read the same reflector panel while it is in that die's L2 (with the columns dealt round
  // robin, a die held workgroups at 16 different panels and streamed the reflectors 456 MB per launch through).
  int c = blockIdx.x;
  if ((gridDim.x & 7) == 0) c = (c & 7) * (gridDim.x >> 3) + (c >> 3);
  if (c >= (B.nq > B.k ? B.nq : B.k)) return;
  const int mm = B.mm, tid = threadIdx.x;
  const double* a = ws_base + B.ws_off * E;
  double* col = q_base + (B.q_off + (long long)c * mm) * E;
  double2 x[RPT];
#pragma unroll
  for (int q = 0; q < RPT; ++q) x[q] = make_double2((tid + NT * q) == c ? 1.0 : 0.0, 0.0);
  const int jtop = c < B.k ? c : B.k - 1;   // reflectors above jtop leave e_c alone
  int buf = 0;
  // the tails of the next panel are requested before the reduction round of the current one where two sets fit the
  // registers (one wave per SIMD)
  constexpr bool PF = NT <= 256;
  double2 u[4][RPT], un[PF ? 4 : 1][PF ? RPT : 1];
  if constexpr (PF) {
    const int jl = (jtop / 4) * 4;
    wy_load_raw<CPLX, NT, RPT>(a, mm, jl, min(4, B.k - jl), tid, u);
  }
  for (int j0 = (jtop / 4) * 4; j0 >= 0; j0 -= 4) {
    const int nbb = min(4, B.k - j0);
    double2 tau[4], g[6];
    if constexpr (!PF) wy_load_raw<CPLX, NT, RPT>(a, mm, j0, nbb, tid, u);
    wy_finish<NT, RPT>(prm_base + B.prm_off, j0, nbb, tid, u, tau, g);
    if constexpr (PF) {
      if (j0 >= 4) wy_load_raw<CPLX, NT, RPT>(a, mm, j0 - 4, 4, tid, un);
    }
    double vals[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) vals[t] = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < RPT; ++q) {
        const double2 t = cmulc(u[i][q], x[q]);
        vals[2 * i] += t.x;
        vals[2 * i + 1] += t.y;
      }
    wy_reduce<NT, 1>(vals, s_part[buf], tid);   // double buffered: one barrier per panel
    buf ^= 1;
    double2 z[4];
#pragma unroll
    for (int i = 3; i >= 0; --i) {
      double2 w = make_double2(vals[2 * i], vals[2 * i + 1]);
#pragma unroll
      for (int l = i + 1; l < 4; ++l) {
        const double2 t = cmul(wy_g(g, i, l), z[l]);
        w.x -= t.x;
        w.y -= t.y;
      }
      z[i] = cmul(tau[i], w);
    }
#pragma unroll
    for (int q = 0; q < RPT; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const double2 t = cmul(u[i][q], z[i]);
        x[q].x -= t.x;
        x[q].y -= t.y;
      }
    if constexpr (PF) {
      if (j0 >= 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int q = 0; q < RPT; ++q) u[i][q] = un[i][q];
      }
    }
  }
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int r = tid + NT * q;
    if (r < mm) Cx<CPLX>::st(col, r, x[q]);
  }
}

inline bool qr_use_wy() { return true; }

template <bool CPLX>
int run_batched(mpse_ctx* ctx, double* ws, double* q, HhParam* prm, const QrBlk* dblk, int nblk, int max_mm,
                int max_nn, int max_k, int max_q, int max_tail, bool form_q) {
  // Register-resident configurations by block height: 256 threads x 4 rows, 512 x 4, 512 x 8 (4 panel columns
  // each).  Per column the kernel pays one reduction round (eight values through the halving butterfly), the
  // scalar reflector set-up in wave 0 and the branch-free update; the waves of one SIMD serialise on the VALU, so
  // tall blocks prefer more rows per thread over more waves (cycle breakdown by s_memtime).
  // Rows per thread follow the tallest block (256 x 1..4 up to 1024 rows, 512 x 5..8 above 2048): the panel is bound
  // by the FP64 vector work of its CU, and row slots that do not exist still cost their share of it (headline: 400
  // rows -> 256 x 2 instead of 256 x 4: -16 % per d = 2 QR; 3200 rows -> 512 x 7).
  constexpr bool fit = true;
  const int cfg = max_mm <= 1024 ? 0 : max_mm <= 2048 ? 1 : 2;
  const int rpt = !fit ? (cfg == 2 ? 8 : 4) : cfg == 0 ? (max_mm + 255) / 256 : cfg == 2 ? (max_mm + 511) / 512 : 4;
  const int nb = 4;
  const bool wy = qr_use_wy();
#define MPSE_QR_CASES_256(KERNEL, GRID, ...)                                                                        \
  switch (rpt) {                                                                                                    \
    case 1: hipLaunchKernelGGL((KERNEL<CPLX, 256, 1 __VA_OPT__(,) __VA_ARGS__>), GRID, dim3(256), 0, ctx->stream, ARGS); break; \
    case 2: hipLaunchKernelGGL((KERNEL<CPLX, 256, 2 __VA_OPT__(,) __VA_ARGS__>), GRID, dim3(256), 0, ctx->stream, ARGS); break; \
    case 3: hipLaunchKernelGGL((KERNEL<CPLX, 256, 3 __VA_OPT__(,) __VA_ARGS__>), GRID, dim3(256), 0, ctx->stream, ARGS); break; \
    default: hipLaunchKernelGGL((KERNEL<CPLX, 256, 4 __VA_OPT__(,) __VA_ARGS__>), GRID, dim3(256), 0, ctx->stream, ARGS);       \
  }
#define MPSE_QR_CASES_512(KERNEL, GRID, ...)                                                                        \
  switch (rpt) {                                                                                                    \
    case 5: hipLaunchKernelGGL((KERNEL<CPLX, 512, 5 __VA_OPT__(,) __VA_ARGS__>), GRID, dim3(512), 0, ctx->stream, ARGS); break; \
    case 6: hipLaunchKernelGGL((KERNEL<CPLX, 512, 6 __VA_OPT__(,) __VA_ARGS__>), GRID, dim3(512), 0, ctx->stream, ARGS); break; \
    case 7: hipLaunchKernelGGL((KERNEL<CPLX, 512, 7 __VA_OPT__(,) __VA_ARGS__>), GRID, dim3(512), 0, ctx->stream, ARGS); break; \
    default: hipLaunchKernelGGL((KERNEL<CPLX, 512, 8 __VA_OPT__(,) __VA_ARGS__>), GRID, dim3(512), 0, ctx->stream, ARGS);       \
  }
  constexpr bool look = true;
  if (look && wy && cfg == 0) {
    // the extra step after the last panel only carries the trailing update of blocks wider than their rank
    for (int j0 = 0; j0 < max_k + (max_tail > 0 ? nb : 0); j0 += nb) {
      const int cols = j0 == 0 ? 0 : max_nn - (j0 - nb) - 1;   // upper bound on columns the update role can own
#define ARGS ws, dblk, prm, j0
      MPSE_QR_CASES_256(k_hh_step, dim3(1 + (cols + 1) / 2, nblk), 2)
#undef ARGS
    }
  } else
  for (int j0 = 0; j0 < max_k; j0 += nb) {
#define ARGS ws, dblk, prm, j0
    if (cfg == 0) {
      MPSE_QR_CASES_256(k_hh_panel, dim3(nblk), 4)
    } else if (cfg == 1) {
      hipLaunchKernelGGL((k_hh_panel<CPLX, 512, 4, 4>), dim3(nblk), dim3(512), 0, ctx->stream, ws, dblk, prm, j0);
    } else {
      MPSE_QR_CASES_512(k_hh_panel, dim3(nblk), 4)
    }
#undef ARGS
    const int trailing = max_nn - j0 - 1;  // upper bound on columns to the right of any block's panel
    if (trailing > 0 && wy) {
      // two columns per workgroup share the loads of V where the registers allow (the tallest configuration holds
      // 4 reflector tails x 8 rows per thread: one column)
      const dim3 grid2((trailing + 1) / 2, nblk), grid1(trailing, nblk);
#define ARGS ws, dblk, prm, j0, nb
      if (cfg == 0) {
        MPSE_QR_CASES_256(k_hh_apply_wy, grid2, 2)
      } else if (cfg == 1) {
        hipLaunchKernelGGL((k_hh_apply_wy<CPLX, 256, 8, 2>), grid2, dim3(256), 0, ctx->stream, ws, dblk, prm, j0, nb);
      } else {
        MPSE_QR_CASES_512(k_hh_apply_wy, grid1, 1)
      }
#undef ARGS
    } else if (trailing > 0) {
      dim3 grid(trailing, nblk);
      switch (cfg) {
        case 0:
          hipLaunchKernelGGL((k_hh_apply_panel<CPLX, 4>), grid, dim3(256), 0, ctx->stream, ws, dblk, prm, j0, nb);
          break;
        case 1:
          hipLaunchKernelGGL((k_hh_apply_panel<CPLX, 8>), grid, dim3(256), 0, ctx->stream, ws, dblk, prm, j0, nb);
          break;
        default:
          hipLaunchKernelGGL((k_hh_apply_panel<CPLX, 16>), grid, dim3(256), 0, ctx->stream, ws, dblk, prm, j0, nb);
      }
    }
  }
  if (form_q && max_q > 0 && wy) {
    dim3 grid(max_q, nblk);
#define ARGS q, ws, dblk, prm
    if (cfg == 0) {
      MPSE_QR_CASES_256(k_hh_formq_wy, grid)
    } else if (cfg == 1) {
      hipLaunchKernelGGL((k_hh_formq_wy<CPLX, 256, 8>), grid, dim3(256), 0, ctx->stream, q, ws, dblk, prm);
    } else {
      MPSE_QR_CASES_512(k_hh_formq_wy, grid)
    }
#undef ARGS
  } else if (form_q && max_q > 0) {
    dim3 grid(max_q, nblk);
    switch (cfg) {
      case 0:
        hipLaunchKernelGGL((k_hh_formq_b<CPLX, 4>), grid, dim3(256), 0, ctx->stream, q, ws, dblk, prm);
        break;
      case 1:
        hipLaunchKernelGGL((k_hh_formq_b<CPLX, 8>), grid, dim3(256), 0, ctx->stream, q, ws, dblk, prm);
        break;
      default:
        hipLaunchKernelGGL((k_hh_formq_b<CPLX, 16>), grid, dim3(256), 0, ctx->stream, q, ws, dblk, prm);
    }
  }
#undef MPSE_QR_CASES_256
#undef MPSE_QR_CASES_512
  MPSE_HIP(ctx, hipGetLastError());
  return MPSE_OK;
}

}  // namespace

// Factorise (and optionally form Q for) nblk column-major blocks living in one workspace.
// Requires max mm <= HH_BATCH_MAX_ROWS; callers fall back to the unblocked kernels otherwise.
int hh_qr_batched(mpse_ctx* ctx, bool cplx, double* ws, double* q, HhParam* prm, const QrBlk* blks_host, int nblk,
                  bool form_q, const QrBlk* blks_dev) {
  if (nblk <= 0) return MPSE_OK;
  int max_mm = 0, max_nn = 0, max_k = 0, max_q = 0, max_tail = 0;
  for (int b = 0; b < nblk; ++b) {
    max_tail = blks_host[b].nn - blks_host[b].k > max_tail ? blks_host[b].nn - blks_host[b].k : max_tail;
    max_mm = blks_host[b].mm > max_mm ? blks_host[b].mm : max_mm;
    max_nn = blks_host[b].nn > max_nn ? blks_host[b].nn : max_nn;
    max_k = blks_host[b].k > max_k ? blks_host[b].k : max_k;
    const int nq = blks_host[b].nq > blks_host[b].k ? blks_host[b].nq : blks_host[b].k;
    max_q = nq > max_q ? nq : max_q;
  }
  if (max_mm > HH_BATCH_MAX_ROWS) return mpse_fail(ctx, MPSE_ERR_SHAPE, "hh_qr_batched: block too tall");
  TmpBuf DB(ctx);
  if (!blks_dev) {
    MPSE_TRY(DB.alloc(size_t(nblk) * sizeof(QrBlk)));
    MPSE_TRY(stage_h2d(ctx, DB.p, blks_host, size_t(nblk) * sizeof(QrBlk)));
    blks_dev = DB.as<QrBlk>();
  }
  if (cplx) return run_batched<true>(ctx, ws, q, prm, blks_dev, nblk, max_mm, max_nn, max_k, max_q, max_tail, form_q);
  return run_batched<false>(ctx, ws, q, prm, blks_dev, nblk, max_mm, max_nn, max_k, max_q, max_tail, form_q);
}
