// FP64-MFMA strided tensor-contraction kernel (the one dense kernel every hot-path
// contraction is built from) and its launcher mpse_gemm.
//
//   C[b](i,j) = alpha * sum_k opA(A[b](i,k)) opB(B[b](k,j)) + beta * C[b](i,j)
//
// Design for gfx950 (MI355X):
//   * v_mfma_f64_16x16x4_f64 (64 cycles/SIMD): a wave owns a 32x32 output tile = 2x2 MFMA tiles on planar
//     (re / im) operand fragments.  complex x complex uses the 3M scheme (three real products per complex
//     product: Ar Br, Ai Bi, (Ar + Ai)(Br + Bi)), complex x real = 2 MFMAs, real x real = 1.
//   * a 256-thread workgroup (4 waves, 2x2) owns a 64x64 tile; K advances 16 at a time:
//     global -> registers (issued one tile ahead through running pointers when both K maps are single level)
//     -> LDS as planar (re / im) panels whose layout follows the operand's memory order ([i][k] rows of 17
//     doubles when k is the contiguous index, [k][i] rows of 80 doubles otherwise, so that both the staging
//     ds_write_b64 and the fragment ds_read_b64 are bank-conflict free) -> register fragments, double buffered
//     per k-group of 4.
//   * operands are read straight through two-level strides (no transpose copies: the reference's tensordot
//     materialises a transposed copy before every ?gemm); complex128 elements are 16-byte loads; the
//     thread->element map follows whichever logical index is contiguous in memory so wave loads coalesce.
//   * split-K with a fixed-order reduction kernel when there are fewer output tiles than CUs.
//   * optional structural-zero skipping: a scan kernel flags the 64 x 16 operand tiles that hold data, the K loop
//     visits only K tiles with data on both sides (block-sparse tensors of the quantum-number-conserving sweeps).
//   * launches with at most one workgroup per compute unit put eight waves on the tile (16 x 32 per wave) so that
//     every SIMD has a second wave to fill its MFMA gaps.
//   * workgroups are placed by die: the dispatcher deals launch positions to the eight dies (XCDs, one L2 each) round
//     robin - die = blockIdx mod 8 - so the position -> (tile, K slice) map decides what each L2 has to hold and how
//     evenly block-sparse work spreads.  Split products run slice-fastest (a die = a K range of both operands);
//     unsplit ones give a die whole tile rows or columns of the larger operand; block-sparse ones with more tiles than
//     slots are launched in a die-aware, heaviest-first order computed from the occupancy masks (k_tile_order), the
//     others with their tile columns skewed by the tile row.  (tools/gemm_balance.py on MPSE_GEMM_TRACE timelines.)
//   * accumulation order over k is fixed and independent of the launch order => bitwise reproducible results.
#include <cstdlib>

#include "mpse_device.h"
#include "mpse_internal.h"

typedef double v4d __attribute__((ext_vector_type(4)));

// debug timeline buffer of the contraction kernel: only inside the profiled (timed) region; MPSE_GEMM_TRACE_ONLY=f0
// leaves the buffer to the records of the fused bond / two-level-site matvec (mpse_heff0.hip)
static unsigned long long* gemm_trace_ptr(const mpse_ctx* ctx) {
  static const bool f0_only = [] {
    const char* e = getenv("MPSE_GEMM_TRACE_ONLY");
    return e && e[0] == 'f';
  }();
  return (ctx->prof_on && !f0_only) ? ctx->gemm_trace : nullptr;
}

namespace {

struct IdxMap {
  long long s_hi, s_lo;
  int ext, lo;
};

// Grouped launch (folded one-site matvec, mpse_plans.h): the tile rows of the launch are divided among up to GMAX_GRP
// groups of equal height; a group has its own result and up to GMAX_SEG (A, B) operand pairs whose products are summed
// - the K loop runs over the segments one after the other.  All pairs share the index maps of GemmArgs (M = rows of
// one group, K = length of one segment).  am / bm: byte flags of the 64 x 16 tiles of the segment's operands (row of
// tile t at t * pitch), null = all occupied.
constexpr int GMAX_GRP = 8, GMAX_SEG = 4;
struct GSeg {
  const double* A;
  const double* B;
  const unsigned char* am;
  const unsigned char* bm;
};
struct GGrp {
  GSeg seg[GMAX_SEG];
  double* C;
  int nseg;
  int use_beta;
};
struct GemmGroups {
  GGrp g[GMAX_GRP];
  int ngrp, tiles_m_grp, nkt_seg, am_pitch, bm_pitch;
  // split2: every output tile is computed by TWO workgroups, each over half of the tile's OCCUPIED K tiles: the first
  // half (with the beta term) is stored to C, the second to C2, laid out like C - the CONSUMER of the result adds the
  // two (the Lanczos update reads both; floating-point atomics onto one result measured slower than the launch they
  // were to speed up: 16 K eight-byte atomics per workgroup).  For block-sparse products with one tile per compute
  // unit: such a launch is as slow as its fullest tile, and a lone workgroup leaves the matrix pipe idle between its
  // MFMAs; the halves of heavy and light tiles are paired per compute unit through the launch order (cpd = compute
  // units per die, perm = tiles by decreasing weight).  DESIGN.md 4.1.
  int split2, cpd;
  double* C2;
  // optional: the flags of every tile's concatenated K range, assembled once (k_tile_order, kept with the launch order
  // for the solve): nkw words per tile, tile = (group's tile row, tile column) in launch-independent order
  const unsigned long long* flags;
};

struct GemmArgs {
  const double* A;
  const double* B;
  double* C;
  IdxMap mA, kA, kB, nB, mC, nC;
  long long sbA, sbB, sbC;  // batch strides in elements
  int M, N, K;
  int tiles_m, tiles_n;     // workgroup tiles
  int mtiles_m, mtiles_n;   // 64 x 64 tiles (granularity of the occupancy masks)
  int a_kfast, b_kfast;
  int conjA, conjB;
  int use_beta;
  double alpha_re, alpha_im, beta_re, beta_im;
  int skew;            // tile column = (column of the linear index + tile row) mod tiles_n
  int die_group;       // 0: off, 1: a die owns whole tile rows, 2: whole tile columns (see the kernel)
  int slice_fast;      // split-K, batch == 1: launch position = tile * ksplit + slice
  const int* perm;     // optional: linear tile index by launch position, tiles with the most K tiles first
  int ksplit;          // number of K slices (1 = none)
  int kt_per_split;    // k-tiles per slice
  double* ws;          // split-K partial sums: [batch][ksplit][M][N] compact, dtype of C
  // structural-zero skipping (optional, single-level K maps only): byte kt of amask[(b * tiles_m + tm) * nkw * 8 ..]
  // is 1 if tile (tm, k-tile kt) of A holds a non-zero element, 0 otherwise; bmask likewise per column tile.
  // (nkw 64-bit words of 8 flags per tile row.)  K tiles where either operand tile is entirely zero are never
  // loaded or multiplied.
  const unsigned long long* amask;
  const unsigned long long* bmask;
  int nkw;
  unsigned long long* kt_counter;   // profiling only: every workgroup adds the number of K tiles it multiplied
  const int* skip;                  // optional device flag: non-zero -> the launch does nothing (mpse_ctx::skip_flag)
  // optional (batch == 1): the kernel that stores the final values of C also accumulates sum conj(C) . y over its
  // share of C (y laid out like C) and stores one (re, im) partial per workgroup at dot_part - the Lanczos
  // coefficient alpha_j = <H v_j, v_j> without a pass of its own over the two vectors
  const double* dot_y;
  double* dot_part;
  // optional (batch == 1): the beta term is read from here (own index maps) instead of from C - a product that
  // accumulates onto a slice of another tensor needs no copy of that slice into C first
  const double* Cin;
  IdxMap mCin, nCin;
  unsigned long long* trace;        // debug timeline (mpse_ctx::gemm_trace), null normally
  GemmGroups gg;                    // GRP instantiations only
};


// (re, im) += conj(c) * y
__device__ __forceinline__ void dot_acc(double& re, double& im, double2 c, double2 y) {
  re += c.x * y.x + c.y * y.y;
  im += c.x * y.y - c.y * y.x;
}

__device__ __forceinline__ long long idx_off(const IdxMap& m, int i) {
  // single-level maps are canonicalised on the host to lo == INT_MAX: skip the integer division (uniform branch)
  if (m.lo == 0x7fffffff) return (long long)i * m.s_lo;
  const int hi = i / m.lo;
  const int l = i - hi * m.lo;
  return (long long)hi * m.s_hi + (long long)l * m.s_lo;
}

#ifndef MPSE_GEMM_3M
#define MPSE_GEMM_3M 1
#endif
constexpr int BM = 64, BN = 64, BK = 16, LDK = BK + 1;   // BM x BN: granularity of the tile-occupancy masks

// KS: both K maps are single level -> no integer division in the K loop
// WS: waves per side of the workgroup tile.  WS = 2: 256 threads own 64 x 64 (the workhorse); WS = 1: ONE wave owns
// 32 x 32 - for products whose 64 x 64 tiles cannot fill the chip, four times as many workgroups instead of slicing K
// into partial sums that a second kernel has to add up.
// WI: 16-row blocks of the output tile per wave.  2: four waves of 32 x 32 (the default).  1 (WS = 2 only): eight
// waves of 16 x 32 on the same 64 x 64 tile - for launches that put ONE workgroup on a compute unit, so that every
// SIMD still has two waves to overlap fragment reads, staging and address arithmetic with the other's MFMAs.
// GRP: grouped launch (GemmGroups; batch == 1, unsplit, KS, K a multiple of BK, M a multiple of the tile height).
template <bool CA, bool CB, bool KS, int WS, int WI = 2, bool GRP = false>
__global__ __launch_bounds__(64 * WS * WS * (2 / WI), WS == 1 ? 1 : WI == 1 ? 4 : ((CA && CB) ? 2 : 3)) void k_gemm(const GemmArgs g) {
  static_assert(!GRP || (KS && WS == 2), "grouped launches use the single-level 64 x 64 kernel");
  constexpr int TBM = 32 * WS, TBN = 32 * WS, NT = 64 * WS * WS * (2 / WI);
  static_assert(WI == 2 || (WI == 1 && WS == 2), "eight-wave form only for 64 x 64 tiles");
  constexpr int LD = WS == 2 ? 80 : 48;          // [k][i] panel rows; LD mod 32 == 16 keeps the fragment reads conflict free
  constexpr int NLD = TBM * BK / NT;             // staged elements per thread and operand (panel = BK*LD >= TBM*LDK doubles)
  constexpr bool CC = CA || CB;
  constexpr int EA = CA ? 2 : 1, EB = CB ? 2 : 1, EC = CC ? 2 : 1;
  // one LDS object: [Are | Aim? | Bre | Bim?], each BK x LD doubles
  __shared__ double smem[(2 + (CA ? 1 : 0) + (CB ? 1 : 0)) * BK * LD];
  if (g.skip && *g.skip) return;   // workgroup-uniform
  const unsigned long long tr0 = g.trace ? __builtin_readcyclecounter() : 0ull;
  const unsigned long long rt0 = g.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;   // device-wide 100 MHz clock
  unsigned long long tr1 = 0, tr2 = 0, tr3 = 0;
  double* sAr = smem;
  double* sAi = sAr + BK * LD;  // only meaningful if CA
  double* sBr = smem + (CA ? 2 : 1) * BK * LD;
  double* sBi = sBr + BK * LD;  // only meaningful if CB

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = WS == 2 ? wave >> 1 : 0, wn = WS == 2 ? wave & 1 : 0;   // WI == 1: wm = 0 .. 3
  constexpr int WROWS = 16 * WI;         // rows of the output tile per wave

  const int ntile = g.tiles_m * g.tiles_n;
  const int bid = blockIdx.x;
  // launch position -> (batch, K slice, tile): tile-fastest, or - split products of one batch element - slice-fastest:
  // the die of a workgroup is its launch position mod 8, so with the slice running fastest a die works on one (or a
  // few) K slices of every tile and its L2 sees 1/8 of both operands instead of most of them
  int bs = bid / ntile;                  // (batch, k-slice)
  int t = bid - bs * ntile;
  int half = 0;                          // grouped launch with split2: which half of the tile's K tiles
  if constexpr (GRP) {
    if (g.gg.split2) {
      // Launch position -> (tile, half).  Positions go to the dies round robin and, on a die, to its compute units in
      // order: the first cpd positions of a die get a unit each, the next cpd join them.  The 2 m halves of the die's m
      // tiles, by decreasing weight, are laid out so that the unit that receives the q-th heaviest half in the first
      // round receives the q-th lightest in the second.
      if ((ntile & 7) == 0) {
        const int m2 = 2 * (ntile >> 3), r = bid >> 3, cpd = g.gg.cpd;
        const int q = (r < cpd || m2 <= cpd) ? r : (m2 - 1) - (r - cpd);
        half = q & 1;
        t = ((q >> 1) << 3) | (bid & 7);
      } else {
        half = bid & 1;
        t = bid >> 1;
      }
      bs = half;       // (slot of the dot partials: half * tiles + tile)
    }
  }
  if (g.slice_fast) {
    t = bid / g.ksplit;
    bs = bid - t * g.ksplit;
  }
  const int b = bs / g.ksplit;
  const int ks_id = bs - b * g.ksplit;
  // heaviest tiles first (k_tile_order); the odd slices of a split product run through the order backwards, so that
  // the compute unit that receives a heavy tile's slice in one round receives a light one in the next
  if (g.perm) t = g.perm[(ks_id & 1) ? ntile - 1 - t : t];
  int tm = t / g.tiles_n;
  int tn = t - tm * g.tiles_n;
  // unsplit products without a sorted order: all tiles of a tile row (die_group 1) or tile column (2) on one die -
  // position p runs on die p mod 8, so die x takes rows x, x + 8, .. with every column: the larger operand's panels are
  // fetched into one L2 instead of into all eight
  if (g.die_group == 1) {
    const int x = t & 7, j = t >> 3;
    tm = x + 8 * (j / g.tiles_n);
    tn = j % g.tiles_n;
  } else if (g.die_group == 2) {
    const int x = t & 7, j = t >> 3;
    tn = x + 8 * (j / g.tiles_m);
    tm = j % g.tiles_m;
  }
  // Workgroups go to the eight dies round robin (die = blockIdx mod 8) and the number of tile columns is a multiple
  // of eight, so without the skew a die would own whole tile columns: with block-sparse operands (quantum-number
  // sectors are column ranges) some dies got three times the K tiles of others (tools/gemm_balance.py).
  if (g.skew && !g.perm && !g.die_group) tn = (tn + tm) % g.tiles_n;

  // grouped launch: the tile row picks the group, the rest of the kernel sees the group's own rows
  int grp = 0;
  if constexpr (GRP) {
    grp = tm / g.gg.tiles_m_grp;
    tm -= grp * g.gg.tiles_m_grp;
  }
  const GGrp& gp = g.gg.g[grp];          // (kernel-argument memory; touched by GRP instantiations only)
  const double* A = GRP ? gp.seg[0].A : g.A + (long long)b * g.sbA * EA;
  const double* B = GRP ? gp.seg[0].B : g.B + (long long)b * g.sbB * EB;
  // (halved tiles: the second half of every tile goes to the second result, without the beta term)
  double* C = GRP ? (half ? g.gg.C2 : gp.C) : g.C + (long long)b * g.sbC * EC;

  // ---- per-thread staging coordinates (4 elements of each operand tile)
  int ai[NLD], ak[NLD], bj[NLD], bk[NLD];
  long long aoff[NLD], boff[NLD];
#pragma unroll
  for (int r = 0; r < NLD; ++r) {
    if (g.a_kfast) {
      ak[r] = tid % BK;
      ai[r] = tid / BK + (NT / BK) * r;
    } else {
      ai[r] = tid % TBM;
      ak[r] = tid / TBM + (NT / TBM) * r;
    }
    if (g.b_kfast) {
      bk[r] = tid % BK;
      bj[r] = tid / BK + (NT / BK) * r;
    } else {
      bj[r] = tid % TBN;
      bk[r] = tid / TBN + (NT / TBN) * r;
    }
    // rows / columns past the edge read a clamped (valid) address: they only feed outputs that
    // are never stored, so no predication is needed and every load below is unconditional
    int gi = min(tm * TBM + ai[r], g.M - 1);
    int gj = min(tn * TBN + bj[r], g.N - 1);
    aoff[r] = idx_off(g.mA, gi);
    boff[r] = idx_off(g.nB, gj);
  }

  double2 ra[NLD], rb[NLD];
  bool ka_in[NLD], kb_in[NLD];
  // generic tile load: two-level K maps and the (only possibly partial) last K tile
  auto load_edge = [&](int kt) {
#pragma unroll
    for (int r = 0; r < NLD; ++r) {
      {
        const int k = kt * BK + ak[r];
        const bool kin = k < g.K;
        const int kc = kin ? k : g.K - 1;
        const long long ko = KS ? (long long)kc * g.kA.s_lo : idx_off(g.kA, kc);
        const double* p = A + (aoff[r] + ko) * EA;
        double2 v;
        if constexpr (CA) {
          v = *reinterpret_cast<const double2*>(p);
        } else {
          v.x = *p;
          v.y = 0.0;
        }
        ra[r] = v;       // zeroing of k >= K happens at LDS-write time: nothing here consumes the load,
        ka_in[r] = kin;  // so the waitcnt for it lands after the MFMA block of the current tile
      }
      {
        const int k = kt * BK + bk[r];
        const bool kin = k < g.K;
        const int kc = kin ? k : g.K - 1;
        const long long ko = KS ? (long long)kc * g.kB.s_lo : idx_off(g.kB, kc);
        const double* p = B + (boff[r] + ko) * EB;
        double2 v;
        if constexpr (CB) {
          v = *reinterpret_cast<const double2*>(p);
        } else {
          v.x = *p;
          v.y = 0.0;
        }
        rb[r] = v;
        kb_in[r] = kin;
      }
    }
  };
  const int nkt_seg = GRP ? g.gg.nkt_seg : 0;
  const int nkt_all = GRP ? gp.nseg * nkt_seg : (g.K + BK - 1) / BK;
  int kt_begin = GRP ? 0 : ks_id * g.kt_per_split;
  int kt_end = GRP ? nkt_all : min(nkt_all, kt_begin + g.kt_per_split);
  // fast path (single-level K maps; the launcher guarantees non-negative strides and operand spans below 4 GB):
  // a uniform tile base (scalar registers, recomputed from the tile number) plus one 32-bit byte offset per lane and
  // staged element - the global_load saddr + voffset form: no per-lane 64-bit pointers to keep and to advance
  unsigned int loa[NLD], lob[NLD];
  const char* a_row0 = nullptr;
  const char* b_col0 = nullptr;
  long long step_a = 0, step_b = 0;
  if constexpr (KS) {
    // tile bases = the smallest row / column offset of the tile.  Inside one group of a two-level map the offset
    // grows with the index (strides >= 0), so the minimum sits at the first row or at the first row of a later group:
    // a short uniform loop over the group starts inside the tile (scalar code, no LDS, no barrier)
    auto tile_min = [&](const IdxMap& m, int first, int count, int ext) -> long long {
      const int last = min(first + count, ext);          // exclusive
      long long best = idx_off(m, min(first, ext - 1));
      if (m.lo != 0x7fffffff) {
        for (int r = (first / m.lo + 1) * m.lo; r < last; r += m.lo) best = min(best, idx_off(m, r));
      }
      return best;
    };
    const long long a0 = tile_min(g.mA, tm * TBM, TBM, g.M), b0 = tile_min(g.nB, tn * TBN, TBN, g.N);
    a_row0 = reinterpret_cast<const char*>(A + a0 * EA);
    b_col0 = reinterpret_cast<const char*>(B + b0 * EB);
#pragma unroll
    for (int r = 0; r < NLD; ++r) {
      loa[r] = (unsigned int)((aoff[r] - a0 + (long long)ak[r] * g.kA.s_lo) * EA * 8);
      lob[r] = (unsigned int)((boff[r] - b0 + (long long)bk[r] * g.kB.s_lo) * EB * 8);
    }
    step_a = (long long)BK * g.kA.s_lo * EA * 8;   // bytes per K tile
    step_b = (long long)BK * g.kB.s_lo * EB * 8;
  }
  // grouped launch: tile bases of every segment's operands (uniform), selected by the segment a K tile falls into
  const char* seg_a[GMAX_SEG];
  const char* seg_b[GMAX_SEG];
  if constexpr (GRP) {
#pragma unroll
    for (int q = 0; q < GMAX_SEG; ++q) {
      const int qq = q < gp.nseg ? q : 0;
      seg_a[q] = a_row0 + (reinterpret_cast<const char*>(gp.seg[qq].A) - reinterpret_cast<const char*>(A));
      seg_b[q] = b_col0 + (reinterpret_cast<const char*>(gp.seg[qq].B) - reinterpret_cast<const char*>(B));
    }
  }
  auto seg_pick = [&](const char* const(&arr)[GMAX_SEG], int q) {
    return q == 0 ? arr[0] : q == 1 ? arr[1] : q == 2 ? arr[2] : arr[3];
  };
  // FULL: every k of the tile is < K.  Otherwise (at most the last K tile of a GEMM) lanes past K skip the
  // load and stage zeros.
  auto load_ks = [&](int kt, bool full) {
    const char* at = a_row0 + (long long)kt * step_a;
    const char* bt = b_col0 + (long long)kt * step_b;
#pragma unroll
    for (int r = 0; r < NLD; ++r) {
      const bool ina = full || kt * BK + ak[r] < g.K;
      const bool inb = full || kt * BK + bk[r] < g.K;
      ra[r] = make_double2(0.0, 0.0);
      rb[r] = make_double2(0.0, 0.0);
      if (ina) {
        if constexpr (CA)
          ra[r] = *reinterpret_cast<const double2*>(at + loa[r]);
        else
          ra[r].x = *reinterpret_cast<const double*>(at + loa[r]);
      }
      if (inb) {
        if constexpr (CB)
          rb[r] = *reinterpret_cast<const double2*>(bt + lob[r]);
        else
          rb[r].x = *reinterpret_cast<const double*>(bt + lob[r]);
      }
    }
  };
  auto load_full = [&](int kt) {
    const char* at = a_row0 + (long long)kt * step_a;
    const char* bt = b_col0 + (long long)kt * step_b;
    if constexpr (GRP) {
      const int q = kt / nkt_seg, l = kt - q * nkt_seg;
      at = seg_pick(seg_a, q) + (long long)l * step_a;
      bt = seg_pick(seg_b, q) + (long long)l * step_b;
    }
#pragma unroll
    for (int r = 0; r < NLD; ++r) {
      if constexpr (CA) {
        ra[r] = *reinterpret_cast<const double2*>(at + loa[r]);
      } else {
        ra[r].x = *reinterpret_cast<const double*>(at + loa[r]);
        ra[r].y = 0.0;
      }
      if constexpr (CB) {
        rb[r] = *reinterpret_cast<const double2*>(bt + lob[r]);
      } else {
        rb[r].x = *reinterpret_cast<const double*>(bt + lob[r]);
        rb[r].y = 0.0;
      }
    }
  };
  auto load_tile = [&](int kt) {
    if constexpr (GRP) {
      load_full(kt);
    } else if constexpr (KS) {
      if ((kt + 1) * BK <= g.K)
        load_full(kt);
      else
        load_ks(kt, false);
    } else {
      load_edge(kt);
    }
  };
  const double sgn_a = g.conjA ? -1.0 : 1.0, sgn_b = g.conjB ? -1.0 : 1.0;

  // complex x complex uses the 3M scheme (three real products per complex product instead of four):
  //   T1 = Ar Br, T2 = Ai Bi, T3 = (Ar + Ai)(Br + Bi)  =>  re = T1 - T2, im = T3 - T1 - T2
  // acc_re holds T1, acc_im holds T3, acc_t2 holds T2 until the epilogue combines them.
  constexpr bool M3 = CA && CB && MPSE_GEMM_3M;
  v4d acc_re[WI][2], acc_im[WI][2], acc_t2[WI][2];
#pragma unroll
  for (int i = 0; i < WI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      acc_re[i][j] = v4d{0, 0, 0, 0};
      acc_im[i][j] = v4d{0, 0, 0, 0};
      acc_t2[i][j] = v4d{0, 0, 0, 0};
    }

  // next K tile >= from (and < kt_end) whose A and B tiles both hold non-zeros; kt_end if there is none.
  // The flag words of this workgroup's tile row / column are fetched into LDS once (up to MASKW words each): read
  // from global memory inside the K loop they put an L2 round trip in front of every tile's loads.
  constexpr int MASKW = 64;
  __shared__ unsigned long long s_mask[2][MASKW];
  const unsigned long long* am = nullptr;
  const unsigned long long* bm = nullptr;
  bool mlds = false;   // the flag words are in LDS (read with ds_read: a generic pointer would make every look-up a
                       // flat load, whose wait drains the operand loads in flight as well)
  if constexpr (GRP) {
    // flags of the concatenated K range of this tile, assembled from the segments' operand masks (g.nkw = words of
    // that range when any segment has a mask, else 0)
    if (g.nkw > 0 && g.nkw <= MASKW && g.gg.flags) {
      if (tid < g.nkw) {
        s_mask[0][tid] = g.gg.flags[((long long)(grp * g.gg.tiles_m_grp + tm) * g.tiles_n + tn) * g.nkw + tid];
        s_mask[1][tid] = 0x0101010101010101ull;
      }
      __syncthreads();
      mlds = true;
      am = s_mask[0];
    } else if (g.nkw > 0 && g.nkw <= MASKW) {
      unsigned char* f0 = reinterpret_cast<unsigned char*>(s_mask[0]);
      for (int t = tid; t < g.nkw * 8; t += NT) {
        unsigned char v = 0;
        if (t < nkt_all) {
          const int q = t / nkt_seg, l = t - q * nkt_seg;
          const GSeg& sg = gp.seg[q];
          v = 1;
          if (sg.am) v &= sg.am[(long long)tm * g.gg.am_pitch + l];
          if (sg.bm) v &= sg.bm[(long long)tn * g.gg.bm_pitch + l];
        }
        f0[t] = v;
      }
      if (tid < g.nkw) s_mask[1][tid] = 0x0101010101010101ull;
      __syncthreads();
      mlds = true;
      am = s_mask[0];       // (non-null: look-ups go through the LDS copy)
    }
    if (g.gg.split2) {
      // this workgroup's half: occupied K tiles [n half / 2, n (half + 1) / 2) of the tile (every thread walks the
      // same flag words); without masks the K range itself is halved
      if (mlds) {
        auto word_of = [&](int w) {
          unsigned long long x = s_mask[0][w] & 0x0101010101010101ull;
          const int valid = nkt_all - 8 * w;
          if (valid < 8) x &= valid > 0 ? (1ull << (8 * valid)) - 1ull : 0ull;
          return x;
        };
        int n = 0;
        for (int w = 0; w < g.nkw; ++w) n += __popcll(word_of(w));
        auto pos_of = [&](int i) {      // position of the i-th occupied tile (i >= n: the end of the range)
          if (i >= n) return nkt_all;
          int seen = 0;
          for (int w = 0; w < g.nkw; ++w) {
            unsigned long long x = word_of(w);
            const int c = __popcll(x);
            if (seen + c > i) {
              for (int skip = i - seen; skip > 0; --skip) x &= x - 1;
              return (w << 3) + (__builtin_ctzll(x) >> 3);
            }
            seen += c;
          }
          return nkt_all;
        };
        kt_begin = half == 0 ? 0 : pos_of(n / 2);
        kt_end = half == 0 ? pos_of(n / 2) : nkt_all;
      } else {
        kt_begin = half == 0 ? 0 : nkt_all / 2;
        kt_end = half == 0 ? nkt_all / 2 : nkt_all;
      }
    }
  } else if constexpr (KS) {
    // masks are kept per 64 rows / columns whatever the workgroup tile
    if (g.amask) am = g.amask + ((long long)b * g.mtiles_m + (tm * TBM) / BM) * g.nkw;
    if (g.bmask) bm = g.bmask + ((long long)b * g.mtiles_n + (tn * TBN) / BN) * g.nkw;
    if ((am || bm) && g.nkw <= MASKW) {
      if (tid < g.nkw) {
        s_mask[0][tid] = am ? am[tid] : 0x0101010101010101ull;
        s_mask[1][tid] = bm ? bm[tid] : 0x0101010101010101ull;
      }
      __syncthreads();
      mlds = true;
    }
  }
  auto next_kt = [&](int from) -> int {
    if (!am && !bm) return from;
    while (from < kt_end) {
      const int w = from >> 3;                      // 8 byte flags per word
      unsigned long long word = 0x0101010101010101ull;
      if (mlds) {
        word &= s_mask[0][w] & s_mask[1][w];
      } else {
        if (am) word &= am[w];
        if (bm) word &= bm[w];
      }
      word &= ~0ull << ((from & 7) * 8);
      if (word) {
        const int kt = (w << 3) + (__builtin_ctzll(word) >> 3);
        return kt < kt_end ? kt : kt_end;
      }
      from = (w + 1) << 3;
    }
    return kt_end;
  };
  int kt = next_kt(kt_begin);
  if (g.trace) tr1 = __builtin_readcyclecounter();
  if (kt < kt_end) {
    load_tile(kt);
  }
  const int frow = lane & 15, fk = lane >> 4;
  // LDS strides (in doubles) of element (i, k) of each panel
  const int sai = g.a_kfast ? LDK : 1, sak = g.a_kfast ? 1 : LD;
  const int sbj = g.b_kfast ? LDK : 1, sbk = g.b_kfast ? 1 : LD;
  int wofa[NLD], wofb[NLD], rofa[WI], rofb[2];
#pragma unroll
  for (int r = 0; r < NLD; ++r) {
    wofa[r] = ai[r] * sai + ak[r] * sak;
    wofb[r] = bj[r] * sbj + bk[r] * sbk;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (i < WI) rofa[i] = (wm * WROWS + i * 16 + frow) * sai + fk * sak;
    rofb[i] = (wn * 32 + i * 16 + frow) * sbj + fk * sbk;
  }

  // operand fragments of one k-group (4 of K): double buffered so that the ds_reads of group kk+1 are in
  // flight under the 16 MFMAs of group kk
  double f_ar[2][WI], f_ai[2][WI], f_br[2][2], f_bi[2][2];
  constexpr int lds_set = 0;
  auto read_frag = [&](int buf, int kk) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i < WI) {
        f_ar[buf][i] = sAr[lds_set + rofa[i] + kk * 4 * sak];
        if constexpr (CA) f_ai[buf][i] = sAi[lds_set + rofa[i] + kk * 4 * sak];
      }
      f_br[buf][i] = sBr[lds_set + rofb[i] + kk * 4 * sbk];
      if constexpr (CB) f_bi[buf][i] = sBi[lds_set + rofb[i] + kk * 4 * sbk];
    }
  };
  // staged registers -> LDS panels of set `off`
  auto stage_to_lds = [&](int off) {
#pragma unroll
    for (int r = 0; r < NLD; ++r) {
      if constexpr (KS) {  // out-of-range k were staged as zeros by load_ks
        sAr[off + wofa[r]] = ra[r].x;
        if constexpr (CA) sAi[off + wofa[r]] = sgn_a * ra[r].y;
        sBr[off + wofb[r]] = rb[r].x;
        if constexpr (CB) sBi[off + wofb[r]] = sgn_b * rb[r].y;
      } else {
        sAr[off + wofa[r]] = ka_in[r] ? ra[r].x : 0.0;
        if constexpr (CA) sAi[off + wofa[r]] = ka_in[r] ? sgn_a * ra[r].y : 0.0;
        sBr[off + wofb[r]] = kb_in[r] ? rb[r].x : 0.0;
        if constexpr (CB) sBi[off + wofb[r]] = kb_in[r] ? sgn_b * rb[r].y : 0.0;
      }
    }
  };
  // the 16 (x3 / x4 / x2 / x1) MFMAs of k-group kk on fragment buffer cb
  auto mfma_group = [&](int cb) {
    if constexpr (M3) {
      double as[WI], bs[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (i < WI) as[i] = f_ar[cb][i] + f_ai[cb][i];
        bs[i] = f_br[cb][i] + f_bi[cb][i];
      }
#pragma unroll
      for (int i = 0; i < WI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc_re[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(f_ar[cb][i], f_br[cb][j], acc_re[i][j], 0, 0, 0);
          acc_t2[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(f_ai[cb][i], f_bi[cb][j], acc_t2[i][j], 0, 0, 0);
          acc_im[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(as[i], bs[j], acc_im[i][j], 0, 0, 0);
        }
    } else {
#pragma unroll
      for (int i = 0; i < WI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc_re[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(f_ar[cb][i], f_br[cb][j], acc_re[i][j], 0, 0, 0);
          if constexpr (CA && CB) {
            acc_re[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(-f_ai[cb][i], f_bi[cb][j], acc_re[i][j], 0, 0, 0);
            acc_im[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(f_ar[cb][i], f_bi[cb][j], acc_im[i][j], 0, 0, 0);
            acc_im[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(f_ai[cb][i], f_br[cb][j], acc_im[i][j], 0, 0, 0);
          } else if constexpr (CA) {
            acc_im[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(f_ai[cb][i], f_br[cb][j], acc_im[i][j], 0, 0, 0);
          } else if constexpr (CB) {
            acc_im[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(f_ar[cb][i], f_bi[cb][j], acc_im[i][j], 0, 0, 0);
          }
        }
    }
  };

  // the last k-group of a tile with the staging stores of the next tile between its MFMAs (one register of each
  // operand per output sub-tile; KS variants)
  auto mfma_group_staging = [&](int cb) {
    double as[WI], bs[2];
    if constexpr (M3) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (i < WI) as[i] = f_ar[cb][i] + f_ai[cb][i];
        bs[i] = f_br[cb][i] + f_bi[cb][i];
      }
    }
#pragma unroll
    for (int i = 0; i < WI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if constexpr (M3) {
          acc_re[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(f_ar[cb][i], f_br[cb][j], acc_re[i][j], 0, 0, 0);
          acc_t2[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(f_ai[cb][i], f_bi[cb][j], acc_t2[i][j], 0, 0, 0);
          acc_im[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(as[i], bs[j], acc_im[i][j], 0, 0, 0);
        } else {
          acc_re[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(f_ar[cb][i], f_br[cb][j], acc_re[i][j], 0, 0, 0);
          if constexpr (CA && CB) {
            acc_re[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(-f_ai[cb][i], f_bi[cb][j], acc_re[i][j], 0, 0, 0);
            acc_im[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(f_ar[cb][i], f_bi[cb][j], acc_im[i][j], 0, 0, 0);
            acc_im[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(f_ai[cb][i], f_br[cb][j], acc_im[i][j], 0, 0, 0);
          } else if constexpr (CA) {
            acc_im[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(f_ai[cb][i], f_br[cb][j], acc_im[i][j], 0, 0, 0);
          } else if constexpr (CB) {
            acc_im[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(f_ar[cb][i], f_bi[cb][j], acc_im[i][j], 0, 0, 0);
          }
        }
        constexpr int PER = NLD / 4 > 0 ? NLD / 4 : 1;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
          const int r = (2 * i + j) * PER + q;
          if (r < NLD) {
            sAr[wofa[r]] = ra[r].x;
            if constexpr (CA) sAi[wofa[r]] = sgn_a * ra[r].y;
            sBr[wofb[r]] = rb[r].x;
            if constexpr (CB) sBi[wofb[r]] = sgn_b * rb[r].y;
          }
        }
        // pin the interleave: the MFMAs of this sub-tile, then its staging stores (left alone the scheduler issues
        // the MFMAs first and the stores in a block behind them)
        constexpr int NM = M3 ? 3 : (CA && CB) ? 4 : (CA || CB) ? 2 : 1;
        __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, (2 + (CA ? 1 : 0) + (CB ? 1 : 0)) * PER, 0);
      }
  };

#ifndef MPSE_GEMM_PIPE
#define MPSE_GEMM_PIPE 1
#endif
  int kt_done = 0;
  if constexpr (KS && MPSE_GEMM_PIPE) {
    // Software-pipelined K loop.  A wave issues in order: every instruction that is not an MFMA and sits between the
    // last MFMA of one K tile and the first of the next (barrier, staging stores, barrier, mask look-up, pointer
    // updates, loads, first fragment reads) is time the matrix pipe idles unless a second workgroup on the CU fills it
    // - 1460 of 4530 cycles per tile for a workgroup alone on its CU (tools/gemm_trace.py).  So the last k-group of a
    // tile is multiplied AFTER the barrier that frees the panels, interleaved with the staging stores of the next
    // tile; the loads of the tile after that are issued under the first k-group of the next iteration.
    int nk = kt_end;
    if (kt < kt_end) {
      stage_to_lds(0);
      nk = next_kt(kt + 1);
      if (nk < kt_end) {
        load_tile(nk);
      }
      lds_barrier();
      if (g.trace) tr2 = __builtin_readcyclecounter();
      read_frag(0, 0);
    }
    const bool any = kt < kt_end;
    while (nk < kt_end) {          // tile kt has a successor (single-exit loop: an exit in the middle made the
      ++kt_done;                   // compiler keep two copies of the accumulators and spill)
      read_frag(1, 1);
      mfma_group(0);
      read_frag(0, 2);
      mfma_group(1);
      read_frag(1, 3);
      mfma_group(0);
      lds_barrier();               // every wave holds its last fragments: the panels may be overwritten
      mfma_group_staging(1);       // the MFMAs of the last k-group, the staging stores of tile nk between them
      // the tile after nk (mask words in LDS: no global load is waited for); its loads stay in flight across the
      // barrier (lds_barrier does not wait for them) and across the next iteration's MFMAs
      const int after = next_kt(nk + 1);
      if (after < kt_end) load_tile(after);
      kt = nk;
      nk = after;
      lds_barrier();               // panels of the next tile complete
      read_frag(0, 0);
    }
    if (any) {                     // last tile: nothing left to stage
      ++kt_done;
      read_frag(1, 1);
      mfma_group(0);
      read_frag(0, 2);
      mfma_group(1);
      read_frag(1, 3);
      mfma_group(0);
      mfma_group(1);
    }
  } else {
  while (kt < kt_end) {
    ++kt_done;
    __syncthreads();
    if (g.trace && kt_done == 1) tr2 = __builtin_readcyclecounter();
    stage_to_lds(0);
    __syncthreads();
    const int nk = next_kt(kt + 1);
    if (nk < kt_end) {
      load_tile(nk);
    }
    kt = nk;

    read_frag(0, 0);
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      const int cb = kk & 1;
      if (kk + 1 < BK / 4) read_frag(cb ^ 1, kk + 1);
      mfma_group(cb);
    }
  }
  }

  if (g.trace) tr3 = __builtin_readcyclecounter();
  struct TraceEnd {     // the record is written when the workgroup leaves the kernel, whichever way
    const GemmArgs& g;
    unsigned long long t0, rt0, &t1, &t2, &t3;
    int tid;
    int& ktd;
    __device__ ~TraceEnd() {
      if (g.trace && tid == 0) {
        const unsigned long long slot = atomicAdd(g.trace, 1ull);
        if (slot < GEMM_TRACE_CAP) {
          unsigned long long* r = g.trace + 1 + slot * GEMM_TRACE_WORDS;
          unsigned xcc = 0;
          r[0] = ((unsigned long long)gridDim.x << 32) | (unsigned long long)blockIdx.x;
          r[1] = ((unsigned long long)(CA ? 1 : 0) << 62) | ((unsigned long long)(CB ? 1 : 0) << 61) |
                 ((unsigned long long)g.ksplit << 40) | ((unsigned long long)(unsigned)g.K << 8) | xcc;
          // where the workgroup ran: HW_ID (wave / SIMD / CU / SH / SE fields) and the die (XCC_ID)
          const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
          r[2] = (unsigned long long)(unsigned)ktd | ((unsigned long long)(((xc & 0xf) << 16) | (hw & 0xffff)) << 32);
          r[3] = t0;
          r[4] = t1;
          r[5] = t2;
          r[6] = t3;
          r[7] = __builtin_readcyclecounter();
          r[8] = rt0;
          r[9] = __builtin_amdgcn_s_memrealtime();
        }
      }
    }
  } trace_end{g, tr0, rt0, tr1, tr2, tr3, tid, kt_done};
  if (g.kt_counter && tid == 0 && kt_done) atomicAdd(g.kt_counter, (unsigned long long)kt_done);
  if constexpr (M3) {
#pragma unroll
    for (int i = 0; i < WI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc_im[i][j] = acc_im[i][j] - acc_re[i][j] - acc_t2[i][j];
        acc_re[i][j] = acc_re[i][j] - acc_t2[i][j];
      }
  }
  // ---- epilogue: lane holds rows (lane>>4)+4r, column lane&15 of each 16x16 tile
  if (g.ksplit > 1) {
    // raw partial sums; alpha/beta and the strided store happen in k_splitk_reduce
    double* wsb = g.ws + (long long)bs * g.M * g.N * EC;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int gj = tn * TBN + wn * 32 + j * 16 + (lane & 15);
      if (gj >= g.N) continue;
#pragma unroll
      for (int i = 0; i < WI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gi = tm * TBM + wm * WROWS + i * 16 + (lane >> 4) + 4 * r;
          if (gi >= g.M) continue;
          double* p = wsb + ((long long)gi * g.N + gj) * EC;
          if constexpr (CC)
            *reinterpret_cast<double2*>(p) = make_double2(acc_re[i][j][r], acc_im[i][j][r]);
          else
            *p = acc_re[i][j][r];
        }
    }
    return;
  }
  double dre = 0.0, dim = 0.0;
  // Two passes: first every load of the beta term and of the dot partner is issued (16 elements per lane, nothing
  // stored yet, so the loads are in flight together), then the results are formed and stored.  Interleaved with the
  // stores, the loads went out one by one (they may alias C for all the compiler knows): 30 000 cycles of epilogue
  // on the C-step of the one-site matvec, a quarter of that workgroup's life (tools/gemm_trace.py).
  // (complex x complex only: the mixed variants run three waves per SIMD on 168 registers, where the 128 registers
  // of the preloads spill; their beta / dot terms are read inside the store loop as before)
  constexpr bool PRE = CA && CB;
  double2 pre_c[PRE ? WI : 1][2][4], pre_y[PRE ? WI : 1][2][4];
  const bool need_c = GRP ? (gp.use_beta != 0 && half == 0) : g.use_beta != 0, need_y = g.dot_y != nullptr;
  if (PRE && (need_c || need_y)) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int gj = min(tn * TBN + wn * 32 + j * 16 + (lane & 15), g.N - 1);     // clamped: never stored past the edge
      const long long coffn = idx_off(g.nC, gj);
      const long long cinn = g.Cin ? idx_off(g.nCin, gj) : 0;
#pragma unroll
      for (int i = 0; i < WI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gi = min(tm * TBM + wm * WROWS + i * 16 + (lane >> 4) + 4 * r, g.M - 1);
          const long long co = idx_off(g.mC, gi) + coffn;
          if (need_c) {
            const double* pin = g.Cin ? g.Cin + (idx_off(g.mCin, gi) + cinn) * EC : C + co * EC;
            if constexpr (CC)
              pre_c[i][j][r] = *reinterpret_cast<const double2*>(pin);
            else
              pre_c[i][j][r] = make_double2(*pin, 0.0);
          }
          if (need_y) {
            if constexpr (CC)
              pre_y[i][j][r] = reinterpret_cast<const double2*>(g.dot_y)[co];
            else
              pre_y[i][j][r] = make_double2(g.dot_y[co], 0.0);
          }
        }
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int gj = tn * TBN + wn * 32 + j * 16 + (lane & 15);
    if (gj >= g.N) continue;
    const long long coffn = idx_off(g.nC, gj);
#pragma unroll
    for (int i = 0; i < WI; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gi = tm * TBM + wm * WROWS + i * 16 + (lane >> 4) + 4 * r;
        if (gi >= g.M) continue;
        const long long co = idx_off(g.mC, gi) + coffn;
        double* p = C + co * EC;
        const double xr = acc_re[i][j][r];
        if constexpr (CC) {
          const double xi = acc_im[i][j][r];
          double2 o;
          o.x = g.alpha_re * xr - g.alpha_im * xi;
          o.y = g.alpha_re * xi + g.alpha_im * xr;
          if (need_c) {
            double2 c0;
            if constexpr (PRE)
              c0 = pre_c[i][j][r];
            else
              c0 = *reinterpret_cast<const double2*>(g.Cin ? g.Cin + (idx_off(g.mCin, gi) + idx_off(g.nCin, gj)) * EC : C + co * EC);
            o.x += g.beta_re * c0.x - g.beta_im * c0.y;
            o.y += g.beta_re * c0.y + g.beta_im * c0.x;
          }
          *reinterpret_cast<double2*>(p) = o;
          if (need_y) {
            if constexpr (PRE)
              dot_acc(dre, dim, o, pre_y[i][j][r]);
            else
              dot_acc(dre, dim, o, reinterpret_cast<const double2*>(g.dot_y)[co]);
          }
        } else {
          double o = g.alpha_re * xr;
          if (need_c) o += g.beta_re * (g.Cin ? g.Cin[idx_off(g.mCin, gi) + idx_off(g.nCin, gj)] : *p);
          *p = o;
          if (need_y) dre += o * g.dot_y[co];
        }
      }
    }
  }
  if constexpr (WS == 2) {
    if (g.dot_y) {   // workgroup-uniform
      // (block-wide sum for NT threads, totals to every thread; fixed order)
      __shared__ double s_dot[2 * (NT / 64)];
      dre = wave_sum(dre);
      dim = wave_sum(dim);
      if (lane == 0) {
        s_dot[2 * wave] = dre;
        s_dot[2 * wave + 1] = dim;
      }
      __syncthreads();
      dre = dim = 0.0;
#pragma unroll
      for (int w = 0; w < NT / 64; ++w) {
        dre += s_dot[2 * w];
        dim += s_dot[2 * w + 1];
      }
      if (tid == 0) {   // slot = the tile, not the launch position: the sum order does not depend on the tile order
        const long long slot = (long long)bs * ntile + (long long)tm * g.tiles_n + tn;
        g.dot_part[2 * slot] = dre;
        g.dot_part[2 * slot + 1] = dim;
      }
    }
  }
}

// Launch order of the output tiles of a block-sparse product.  The dispatcher hands workgroups to the eight dies
// round robin (die = launch position mod 8, checked with tools/gemm_balance.py) and to free slots in launch order.
//  * Tiles are sorted by the number of K tiles both operands occupy, most first: longest-processing-time-first
//    scheduling.  In storage order the full tiles of a quantum-number sector sit next to each other and pile up on
//    the same dies and compute units (30 K tiles on the fullest CU against 21 on average for the d = 16 A-step).
//  * Where the tile columns divide evenly among the dies, every die owns whole tile columns (chosen in a snake over
//    the columns sorted by weight, so that the dies carry equal work): its L2 then holds the column panels of B it
//    needs instead of streaming all of B (the plain sorted order raised the HBM-side traffic of the A-step by 2.5x).
// One workgroup, bitonic sorts of <= 2048 keys in LDS.
// (n: power of two <= 2048, the keys beyond the data are zero; 1024 threads, n / 2 compare-exchange pairs per step)
__device__ __forceinline__ void bitonic_desc(unsigned* key, int tid, int n) {
  for (int size = 2; size <= n; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      if (tid < (n >> 1)) {
        const int lo = 2 * tid - (tid & (stride - 1)), hi = lo + stride;
        const bool desc = (lo & size) == 0;
        const unsigned a = key[lo], b = key[hi];
        if ((a < b) == desc) {
          key[lo] = b;
          key[hi] = a;
        }
      }
      __syncthreads();
    }
}
__device__ __forceinline__ int pow2_at_least(int x) {
  int n = 2;
  while (n < x) n <<= 1;
  return n;
}
__global__ __launch_bounds__(1024) void k_tile_order(const unsigned long long* __restrict__ amask,
                                                       const unsigned long long* __restrict__ bmask, int nkw, int nkt,
                                                       int tiles_m, int tiles_n, int by_die, int* __restrict__ perm,
                                                       const int* __restrict__ skip, const GemmGroups gg,
                                                       unsigned char* __restrict__ flags_out, int flags_pitch) {
  if (skip && *skip) return;
  __shared__ unsigned key[2048], ckey[2048];
  __shared__ int colw[2048];
  __shared__ unsigned char die_of[2048];
  const int ntile = tiles_m * tiles_n, tid = threadIdx.x;
  const bool part = by_die && tiles_n % 8 == 0 && tiles_n <= 2048;
  for (int c = tid; c < 2048; c += 1024) colw[c] = 0;
  __syncthreads();
  for (int i = tid; i < 2048; i += 1024) {
    unsigned cnt = 0;
    if (i < ntile) {
      const int tm = i / tiles_n, tn = i - tm * tiles_n;
      if (gg.ngrp > 0) {   // grouped launch: the segments of the tile row's group, byte flags per operand
        const int grp = tm / gg.tiles_m_grp, tml = tm - grp * gg.tiles_m_grp;
        const GGrp& G = gg.g[grp];
        for (int q = 0; q < G.nseg; ++q)
          for (int l = 0; l < gg.nkt_seg; ++l) {
            unsigned v = 1;
            if (G.seg[q].am) v &= G.seg[q].am[(long long)tml * gg.am_pitch + l];
            if (G.seg[q].bm) v &= G.seg[q].bm[(long long)tn * gg.bm_pitch + l];
            cnt += v;
            if (flags_out) flags_out[(long long)i * flags_pitch + q * gg.nkt_seg + l] = (unsigned char)v;
          }
        if (flags_out)
          for (int l = G.nseg * gg.nkt_seg; l < flags_pitch; ++l) flags_out[(long long)i * flags_pitch + l] = 0;
      } else {
        for (int w = 0; w < nkw; ++w) {
          unsigned long long x = 0x0101010101010101ull;
          if (amask) x &= amask[(long long)tm * nkw + w];
          if (bmask) x &= bmask[(long long)tn * nkw + w];
          const int valid = nkt - 8 * w;                    // flag bytes past the last K tile are never written
          if (valid < 8) x &= (1ull << (8 * valid)) - 1ull;
          cnt += __popcll(x);
        }
      }
      if (part) atomicAdd(&colw[tn], (int)cnt);
    }
    key[i] = cnt;      // (<= 512 K tiles)
  }
  __syncthreads();
  if (part) {
    for (int c = tid; c < 2048; c += 1024) ckey[c] = c < tiles_n ? ((unsigned)colw[c] << 11) | (unsigned)(2047 - c) : 0u;
    __syncthreads();
    bitonic_desc(ckey, tid, pow2_at_least(tiles_n));
    for (int r = tid; r < tiles_n; r += 1024) {
      const int c = 2047 - (int)(ckey[r] & 2047u), r16 = r & 15;
      die_of[c] = (unsigned char)(r16 < 8 ? r16 : 15 - r16);          // snake over the columns by weight
    }
    __syncthreads();
  }
  for (int i = tid; i < 2048; i += 1024) {
    unsigned k = 0;
    if (i < ntile) {
      const unsigned die = part ? die_of[i % tiles_n] : 0u;
      k = (die << 28) | (key[i] << 16) | (unsigned)(0xFFFF - i);
    }
    key[i] = k;
  }
  __syncthreads();
  bitonic_desc(key, tid, pow2_at_least(ntile));
  const int per_die = ntile / 8;           // part: every die owns tiles_n / 8 columns of tiles_m tiles
  for (int i = tid; i < ntile; i += 1024) {
    const int s = part ? (7 - (i & 7)) * per_die + (i >> 3) : i;
    perm[i] = 0xFFFF - (int)(key[s] & 0xFFFFu);
  }
}

// C(i,j) = alpha * sum_s ws[b][s][i][j] + beta * C(i,j); slices summed in fixed order.
// One output row per blockIdx.y step (row index arithmetic is wave-uniform), threads run along j.
template <bool CC>
__global__ __launch_bounds__(256) void k_splitk_reduce(const GemmArgs g, int batch) {
  constexpr int EC = CC ? 2 : 1;
  if (g.skip && *g.skip) return;
  const long long mn = (long long)g.M * g.N;
  const int rows = g.M * batch;
  double dre = 0.0, dim = 0.0;
  for (int row = blockIdx.y; row < rows; row += gridDim.y) {
    const int b = row / g.M;
    const int i = row - b * g.M;
    const double* wrow = g.ws + ((long long)b * g.ksplit * mn + (long long)i * g.N) * EC;
    double* crow = g.C + ((long long)b * g.sbC + idx_off(g.mC, i)) * EC;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < g.N; j += gridDim.x * 256) {
      const double* w = wrow + (long long)j * EC;
      double xr = 0, xi = 0;
      int s = 0;
      // four slices in flight per thread (the loads are independent, the additions keep their fixed order)
      for (; s + 4 <= g.ksplit; s += 4) {
        if constexpr (CC) {
          const double2 v0 = *reinterpret_cast<const double2*>(w + (long long)s * mn * EC);
          const double2 v1 = *reinterpret_cast<const double2*>(w + (long long)(s + 1) * mn * EC);
          const double2 v2 = *reinterpret_cast<const double2*>(w + (long long)(s + 2) * mn * EC);
          const double2 v3 = *reinterpret_cast<const double2*>(w + (long long)(s + 3) * mn * EC);
          xr += v0.x;
          xi += v0.y;
          xr += v1.x;
          xi += v1.y;
          xr += v2.x;
          xi += v2.y;
          xr += v3.x;
          xi += v3.y;
        } else {
          const double v0 = w[(long long)s * mn], v1 = w[(long long)(s + 1) * mn];
          const double v2 = w[(long long)(s + 2) * mn], v3 = w[(long long)(s + 3) * mn];
          xr += v0;
          xr += v1;
          xr += v2;
          xr += v3;
        }
      }
      for (; s < g.ksplit; ++s) {
        if constexpr (CC) {
          const double2 v = *reinterpret_cast<const double2*>(w + (long long)s * mn * EC);
          xr += v.x;
          xi += v.y;
        } else {
          xr += w[(long long)s * mn];
        }
      }
      const long long co = idx_off(g.nC, j);
      double* p = crow + co * EC;
      const double* pin = g.Cin ? g.Cin + (idx_off(g.mCin, i) + idx_off(g.nCin, j)) * EC : p;
      if constexpr (CC) {
        double2 o = make_double2(g.alpha_re * xr - g.alpha_im * xi, g.alpha_re * xi + g.alpha_im * xr);
        if (g.use_beta) {
          const double2 c0 = *reinterpret_cast<const double2*>(pin);
          o.x += g.beta_re * c0.x - g.beta_im * c0.y;
          o.y += g.beta_re * c0.y + g.beta_im * c0.x;
        }
        *reinterpret_cast<double2*>(p) = o;
        if (g.dot_y) dot_acc(dre, dim, o, reinterpret_cast<const double2*>(g.dot_y)[idx_off(g.mC, i) + co]);
      } else {
        double o = g.alpha_re * xr;
        if (g.use_beta) o += g.beta_re * (*pin);
        *p = o;
        if (g.dot_y) dre += o * g.dot_y[idx_off(g.mC, i) + co];
      }
    }
  }
  if (g.dot_y) {   // grid-uniform; batch == 1
    block_allsum2(dre, dim);
    if (threadIdx.x == 0) {
      const long long bid = (long long)blockIdx.y * gridDim.x + blockIdx.x;
      g.dot_part[2 * bid] = dre;
      g.dot_part[2 * bid + 1] = dim;
    }
  }
}

// Tile occupancy of both GEMM operands in one launch: one wave per (k tile, 64-row tile, batch x operand); lane r
// scans the 16 k of its row and the wave stores the byte flag of the tile (1 = something non-zero).  "Rows" are
// the M index for A and the N index for B.  The sweeps' tensors are block sparse by quantum number and every
// kernel keeps their structural zeros exact, so whole tiles vanish.
struct OccOperand {
  const double* base;
  IdxMap rmap, kmap;
  int nrows, tiles, cplx, kfast;
  long long sb;
  unsigned char* flags;   // [batch][tiles][nkw * 8]
};
__global__ __launch_bounds__(256) void k_tile_occ(OccOperand oa, OccOperand ob, int K, int nkw, int batch,
                                                  const int* skip) {
  if (skip && *skip) return;
  // one wave per k tile (4 per workgroup); 16 independent loads per lane, lanes along whichever index is contiguous
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int kt = blockIdx.x * 4 + wave, t = blockIdx.y;
  const bool second = (int)blockIdx.z >= batch;
  const OccOperand& o = second ? ob : oa;
  const int b = second ? blockIdx.z - batch : blockIdx.z;
  if (t >= o.tiles || kt * BK >= K) return;
  const int E = o.cplx ? 2 : 1;
  const double* base = o.base + (long long)b * o.sb * E;
  bool nz = false;
  if (o.kfast) {   // k contiguous: 16 lanes span the 16 k of a row, 4 rows per pass
    const int k = kt * BK + (lane & 15);
    const bool kin = k < K;
    const long long ko = idx_off(o.kmap, kin ? k : K - 1);
    // all 16 loads are issued before the first compare (clamped addresses instead of predicated loads)
    double v0[16], v1[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = min(t * BM + i * 4 + (lane >> 4), o.nrows - 1);
      const double* q = base + (idx_off(o.rmap, r) + ko) * E;
      v0[i] = q[0];
      v1[i] = o.cplx ? q[1] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const bool in = (t * BM + i * 4 + (lane >> 4)) < o.nrows && kin;
      nz |= in && (v0[i] != 0.0 || v1[i] != 0.0);
    }
  } else {         // rows contiguous: one row per lane, 16 k per lane
    const int r = t * BM + lane;
    if (r < o.nrows) {
      const double* p = base + idx_off(o.rmap, r) * E;
      double v0[16], v1[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int k = min(kt * BK + i, K - 1);
        const double* q = p + idx_off(o.kmap, k) * E;
        v0[i] = q[0];
        v1[i] = o.cplx ? q[1] : 0.0;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) nz |= (kt * BK + i < K) && (v0[i] != 0.0 || v1[i] != 0.0);
    }
  }
  const bool any = __ballot(nz) != 0ull;
  if (lane == 0) o.flags[((long long)b * o.tiles + t) * nkw * 8 + kt] = any ? 1 : 0;
}

bool to_map(const mpse_index& s, IdxMap* m) {
  if (s.ext < 0 || s.ext > 0x7fffffffLL) return false;
  m->ext = (int)s.ext;
  long long lo = s.lo_ext <= 0 ? 1 : s.lo_ext;
  m->s_hi = s.s_hi;
  m->s_lo = s.s_lo;
  // canonical single-level forms: lo >= ext ; lo == 1 (only the hi level moves) ; contiguous levels
  if (lo >= s.ext) {
    m->s_hi = 0;
  } else if (lo == 1) {
    m->s_lo = s.s_hi;
    m->s_hi = 0;
    lo = s.ext;
  } else if (s.s_hi == lo * s.s_lo) {
    m->s_hi = 0;
    lo = s.ext;
  }
  if (lo >= s.ext) lo = 0x7fffffffLL;
  m->lo = (int)lo;
  return true;
}

inline bool is_single(const IdxMap& m) { return m.lo == 0x7fffffff; }

// ---- (d0,d1,d2) -> (d0,d2,d1) copy through a 32x32 LDS tile
template <bool CPLX>
__global__ __launch_bounds__(256) void k_transpose_inner(double* out, const double* in, long long d0, int d1,
                                                         int d2, int conj) {
  constexpr int E = CPLX ? 2 : 1;
  __shared__ double tile[32][33 * E];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int t1 = (d1 + 31) / 32, t2 = (d2 + 31) / 32;
  long long bid = blockIdx.x;
  const long long b0 = bid / ((long long)t1 * t2);
  const int rem = (int)(bid - b0 * t1 * t2);
  const int by = rem / t2, bx = rem - by * t2;
  const double* src = in + b0 * (long long)d1 * d2 * E;
  double* dst = out + b0 * (long long)d1 * d2 * E;
  for (int yy = ty; yy < 32; yy += 8) {
    int y = by * 32 + yy, x = bx * 32 + tx;
    if (y < d1 && x < d2) {
      const double* p = src + ((long long)y * d2 + x) * E;
      tile[yy][tx * E] = p[0];
      if (CPLX) tile[yy][tx * E + 1] = conj ? -p[1] : p[1];
    }
  }
  __syncthreads();
  for (int yy = ty; yy < 32; yy += 8) {
    int x = bx * 32 + yy, y = by * 32 + tx;  // output row = old column
    if (y < d1 && x < d2) {
      double* p = dst + ((long long)x * d1 + y) * E;
      p[0] = tile[tx][yy * E];
      if (CPLX) p[1] = tile[tx][yy * E + 1];
    }
  }
}

}  // namespace

static int gemm_impl(mpse_ctx* ctx, const mpse_gemm_desc* d, const void* A, const void* B, void* C, int skip_zero) {
  if (!ctx || !d) return MPSE_ERR_ARG;
  MPSE_BIND(ctx);
  if ((d->dtype_a != MPSE_F64 && d->dtype_a != MPSE_C128) || (d->dtype_b != MPSE_F64 && d->dtype_b != MPSE_C128))
    return mpse_fail(ctx, MPSE_ERR_ARG, "mpse_gemm: unknown dtype");
  if (d->m_a.ext != d->m_c.ext || d->n_b.ext != d->n_c.ext || d->k_a.ext != d->k_b.ext)
    return mpse_fail(ctx, MPSE_ERR_SHAPE, "mpse_gemm: extents disagree (M %lld/%lld N %lld/%lld K %lld/%lld)",
                     (long long)d->m_a.ext, (long long)d->m_c.ext, (long long)d->n_b.ext, (long long)d->n_c.ext,
                     (long long)d->k_a.ext, (long long)d->k_b.ext);
  if (d->m_a.ext == 0 || d->n_b.ext == 0 || d->batch <= 0) return MPSE_OK;
  if (!A || !B || !C) return mpse_fail(ctx, MPSE_ERR_ARG, "mpse_gemm: null operand");
  GemmArgs g;
  g.A = (const double*)A;
  g.B = (const double*)B;
  g.C = (double*)C;
  if (!to_map(d->m_a, &g.mA) || !to_map(d->k_a, &g.kA) || !to_map(d->k_b, &g.kB) || !to_map(d->n_b, &g.nB) ||
      !to_map(d->m_c, &g.mC) || !to_map(d->n_c, &g.nC))
    return mpse_fail(ctx, MPSE_ERR_SHAPE, "mpse_gemm: extent out of range");
  g.sbA = d->sb_a;
  g.sbB = d->sb_b;
  g.sbC = d->sb_c;
  g.M = g.mA.ext;
  g.N = g.nB.ext;
  g.K = g.kA.ext;
  g.mtiles_m = (g.M + BM - 1) / BM;
  g.mtiles_n = (g.N + BN - 1) / BN;
  // workgroup tile: 64 x 64.  (A 32 x 32 one-wave tile for products whose 64 x 64 tiles cannot fill the chip lost to
  // split-K by 17 % on the headline run - one wave walking the whole K is a longer latency chain than many workgroups
  // walking two K tiles each plus a reduction pass - and was removed.)
  g.tiles_m = g.mtiles_m;
  g.tiles_n = g.mtiles_n;
  auto fast = [](const IdxMap& m) { return m.ext <= 1 ? (long long)1 << 60 : (m.s_lo < 0 ? -m.s_lo : m.s_lo); };
  g.a_kfast = fast(g.kA) <= fast(g.mA);
  g.b_kfast = fast(g.kB) <= fast(g.nB);
  g.conjA = d->conj_a && d->dtype_a == MPSE_C128;
  g.conjB = d->conj_b && d->dtype_b == MPSE_C128;
  g.alpha_re = d->alpha_re;
  g.alpha_im = d->alpha_im;
  g.beta_re = d->beta_re;
  g.beta_im = d->beta_im;
  g.use_beta = (d->beta_re != 0.0 || d->beta_im != 0.0);
  // beta source of the caller (mpse_ctx::cin_req, set by run_plan for this one call)
  g.Cin = nullptr;
  if (ctx->cin_req.ptr) {
    const mpse_ctx::CinReq rq = ctx->cin_req;
    ctx->cin_req = mpse_ctx::CinReq();
    if (d->batch != 1 || !g.use_beta || !to_map(rq.m, &g.mCin) || !to_map(rq.n, &g.nCin) || g.mCin.ext != g.mC.ext ||
        g.nCin.ext != g.nC.ext)
      return mpse_fail(ctx, MPSE_ERR_ARG, "mpse_gemm: beta source needs batch == 1, beta != 0 and the extents of C");
    g.Cin = static_cast<const double*>(rq.ptr);
  }
  const bool ca = d->dtype_a == MPSE_C128, cb = d->dtype_b == MPSE_C128;
  // split-K when the output tiles alone cannot fill the 256 CUs (skinny results with long K)
  const long long base_blocks = (long long)g.tiles_m * g.tiles_n * d->batch;
  const int nkt_all = (g.K + BK - 1) / BK;
  g.ksplit = 1;
  g.kt_per_split = nkt_all > 0 ? nkt_all : 1;
  g.ws = nullptr;
  g.amask = g.bmask = nullptr;
  g.nkw = 0;
  g.skip = ctx->skip_flag;
  g.skew = 1;
  g.perm = nullptr;
  g.gg.ngrp = 0;
  g.slice_fast = 0;
  g.die_group = 0;
  g.dot_y = nullptr;
  g.dot_part = nullptr;
  if (!ctx->gemm_trace_checked) {
    ctx->gemm_trace_checked = true;
    if (getenv("MPSE_GEMM_TRACE")) {
      void* pt = nullptr;
      if (hipMalloc(&pt, (1 + GEMM_TRACE_CAP * GEMM_TRACE_WORDS) * sizeof(unsigned long long)) == hipSuccess) {
        (void)hipMemsetAsync(pt, 0, sizeof(unsigned long long), ctx->stream);
        ctx->gemm_trace = static_cast<unsigned long long*>(pt);
      }
    }
  }
  g.trace = gemm_trace_ptr(ctx);
  TmpBuf WSB(ctx), MSK(ctx);
  bool leave_slices = false;
  const int n_cu = ctx->n_cu > 0 ? ctx->n_cu : 256;
  const int tiles_limit = n_cu;
  const long long wg_target = n_cu;   // one workgroup per CU (policy sweeps of the headline run)
  if (base_blocks < tiles_limit && nkt_all >= 4) {
    // fewer output tiles than CUs: slice K until ~1 workgroup per CU exists (2 per CU: equal to 1.7 % slower on the
    // headline run depending on the box - the reduction pass reads twice the slices; 3 per CU: -3 %; 0.5 per CU: -7 %).  (One tile per CU runs as fast
    // unsplit as split in two + reduction pass since the K loop prefetches fragments: measured, 4096x256 C-step.)
    int want = (int)((wg_target + base_blocks - 1) / base_blocks);
    int maxs = nkt_all / 2;                                           // at least two k-tiles per slice
    int S = want < maxs ? want : maxs;
    if (S > 1) {
      g.kt_per_split = (nkt_all + S - 1) / S;
      g.ksplit = (nkt_all + g.kt_per_split - 1) / g.kt_per_split;
      // (Leaving the slices to the consumer of a matvec result - the Lanczos update adding them while it reads -
      // instead of the reduction launch measured 1.4 % slower on the headline run: every slice's workgroups then load
      // the dot partner, and the update kernel streams 16 slices with a fraction of the reduction kernel's blocks.)
      const size_t esz = (ca || cb) ? 16 : 8;
      const size_t ws_bytes = size_t(d->batch) * g.ksplit * size_t(g.M) * size_t(g.N) * esz;
      // slices for a consumer that adds them itself (mpse_ctx::slices_req): plain product into a compact result
      const bool compact = is_single(g.mC) && is_single(g.nC) && g.mC.s_lo == g.N && (g.nC.s_lo == 1 || g.N == 1);
      if (ctx->slices_req.ptr && !ctx->dot_now && d->batch == 1 && compact && !g.use_beta && d->alpha_re == 1.0 &&
          d->alpha_im == 0.0 && ws_bytes <= ctx->slices_req.cap_bytes) {
        g.ws = static_cast<double*>(ctx->slices_req.ptr);
        ctx->slices_req.used = g.ksplit;
        leave_slices = true;
      } else {
        MPSE_TRY(WSB.alloc(ws_bytes));
        g.ws = WSB.as<double>();
      }
    }
  }
  g.slice_fast = g.ksplit > 1 && d->batch == 1;
  long long nblk = base_blocks * g.ksplit;
  if (nblk > 0x7fffffffLL) return mpse_fail(ctx, MPSE_ERR_SHAPE, "mpse_gemm: grid too large");

  // dot request of the caller (mpse_ctx::dot_req, armed by run_plan for the step that completes the result)
  const bool want_dot = ctx->dot_now;
  ctx->dot_now = false;
  long long rgx = (g.N + 255) / 256 > 64 ? 64 : (g.N + 255) / 256;
  long long rgy = (long long)g.M * d->batch > 32768 ? 32768 : (long long)g.M * d->batch;
  if (want_dot && d->batch == 1) {
    long long producers = base_blocks;
    if (g.ksplit > 1) {
      const long long cap_y = ctx->dot_req.cap / rgx;
      if (cap_y >= 1 && rgy > cap_y) rgy = cap_y;
      producers = rgx * rgy;
    }
    if (producers >= 1 && producers <= ctx->dot_req.cap) {
      g.dot_y = static_cast<const double*>(ctx->dot_req.y);
      g.dot_part = ctx->dot_req.part;
      ctx->dot_req.nb_out = (int)producers;
    }
  }

  dim3 grid((unsigned)nblk), block(256);
  mpse_ctx::ProfRec rec;
  const int variant = (ca ? 1 : 0) + (cb ? 2 : 0);
  const double mnk = double(g.M) * double(g.N) * double(g.K) * double(d->batch);
  // algorithmic (dense-equivalent) flops; compulsory traffic: read A and B once, write C once (+ read C when beta != 0)
  const bool prof_this = prof_begin(
      ctx, variant, mnk * ((ca && cb) ? 8.0 : (ca || cb) ? 4.0 : 2.0),
      double(d->batch) * (double(g.M) * g.K * (ca ? 16 : 8) + double(g.K) * g.N * (cb ? 16 : 8) +
                          double(g.M) * g.N * ((ca || cb) ? 16 : 8) * (g.use_beta ? 2 : 1)),
      &rec);
  g.kt_counter = prof_this && ctx->prof_ktiles ? ctx->prof_ktiles + variant : nullptr;
  bool mask_a_stable = true, mask_b_stable = true;   // the mask (or its absence) outlives this call: cached or the solve's
  if (skip_zero && is_single(g.kA) && is_single(g.kB) && nkt_all >= 2 && d->batch <= 16384) {
    // tile occupancy of both operands (one small scan launch), then only K tiles with data on both sides are visited
    g.nkw = (nkt_all + 7) / 8;
    const size_t wa = size_t(d->batch) * g.mtiles_m * g.nkw, wb = size_t(d->batch) * g.mtiles_n * g.nkw;
    // skip_zero bit 0: scan A, bit 1: scan B (an operand that is as large as the product itself is not worth a pass)
    const bool sa = skip_zero & 1, sb_ = skip_zero & 2;
    // Inside a Krylov solve the environments do not change: their masks are computed once and kept (mpse_internal.h)
    auto cacheable = [&](const void* p) {
      if (!ctx->occ_cache_on) return false;
      const char* c = reinterpret_cast<const char*>(p);
      return (c >= ctx->occ_lo[0] && c < ctx->occ_hi[0]) || (c >= ctx->occ_lo[1] && c < ctx->occ_hi[1]);
    };
    auto make_key = [&](const void* p, const IdxMap& rm, const IdxMap& km, long long sb, int nrows, int tiles, int cplx) {
      mpse_ctx::OccKey k;
      memset(&k, 0, sizeof(k));
      k.ptr = p;
      k.r_ext = rm.ext, k.r_lo = rm.lo, k.r_shi = rm.s_hi, k.r_slo = rm.s_lo;
      k.k_ext = km.ext, k.k_lo = km.lo, k.k_shi = km.s_hi, k.k_slo = km.s_lo;
      k.sb = sb, k.nrows = nrows, k.tiles = tiles, k.nkw = g.nkw, k.batch = (int)d->batch, k.K = g.K, k.cplx = cplx;
      return k;
    };
    auto find = [&](const mpse_ctx::OccKey& k) -> void* {
      for (const auto& e : ctx->occ_cache)
        if (memcmp(&e.key, &k, sizeof(k)) == 0) return e.mask;
      return nullptr;
    };
    unsigned long long *am = nullptr, *bmk = nullptr;
    bool scan_a = sa, scan_b = sb_, b_structural = false;
    mpse_ctx::OccKey ka, kb;
    const bool ca_ok = sa && cacheable(g.A), cb_ok = sb_ && cacheable(g.B);
    if (ca_ok) {
      ka = make_key(g.A, g.mA, g.kA, g.sbA, g.M, g.mtiles_m, ca ? 1 : 0);
      if (void* hit = find(ka)) am = static_cast<unsigned long long*>(hit), scan_a = false;
    }
    if (cb_ok) {
      kb = make_key(g.B, g.nB, g.kB, g.sbB, g.N, g.mtiles_n, cb ? 1 : 0);
      if (void* hit = find(kb)) bmk = static_cast<unsigned long long*>(hit), scan_b = false;
    }
    // B is a Krylov vector of a solve whose caller supplied the structural mask of the centre tensor: no scan
    if (scan_b && ctx->cmask.ptr && d->batch == 1 && (long long)(wb * sizeof(unsigned long long)) == ctx->cmask.bytes) {
      const char* pb_ = reinterpret_cast<const char*>(g.B);
      if (pb_ >= ctx->cmask.lo && pb_ < ctx->cmask.hi) {
        bmk = static_cast<unsigned long long*>(const_cast<void*>(ctx->cmask.ptr));
        scan_b = false;
        b_structural = true;
      }
    }
    mask_a_stable = !sa || ca_ok;
    mask_b_stable = !sb_ || cb_ok || b_structural;
    // storage: cached masks live until the solve ends, the others in a temporary of this call
    size_t tmp_words = 0;
    if (scan_a && !ca_ok) tmp_words += wa;
    if (scan_b && !cb_ok) tmp_words += wb;
    if (tmp_words) MPSE_TRY(MSK.alloc(tmp_words * sizeof(unsigned long long)));
    unsigned long long* tmp = MSK.as<unsigned long long>();
    if (scan_a) {
      if (ca_ok) {
        void* pm = nullptr;
        MPSE_TRY(mpse_malloc(ctx, wa * sizeof(unsigned long long), &pm));
        ctx->occ_cache.push_back({ka, pm});
        am = static_cast<unsigned long long*>(pm);
      } else {
        am = tmp;
        tmp += wa;
      }
    }
    if (scan_b) {
      if (cb_ok) {
        void* pm = nullptr;
        MPSE_TRY(mpse_malloc(ctx, wb * sizeof(unsigned long long), &pm));
        ctx->occ_cache.push_back({kb, pm});
        bmk = static_cast<unsigned long long*>(pm);
      } else {
        bmk = tmp;
      }
    }
    // (flag bytes past the last k tile stay unwritten: next_kt never looks beyond kt_end)
    if (scan_a || scan_b) {
      OccOperand oa{g.A, g.mA, g.kA, g.M, scan_a ? g.mtiles_m : 0, ca ? 1 : 0, g.a_kfast, g.sbA, reinterpret_cast<unsigned char*>(am)};
      OccOperand ob{g.B, g.nB, g.kB, g.N, scan_b ? g.mtiles_n : 0, cb ? 1 : 0, g.b_kfast, g.sbB, reinterpret_cast<unsigned char*>(bmk)};
      const int tmax = (scan_a ? g.mtiles_m : 0) > (scan_b ? g.mtiles_n : 0) ? (scan_a ? g.mtiles_m : 0) : (scan_b ? g.mtiles_n : 0);
      const dim3 og((nkt_all + 3) / 4, tmax, (unsigned)(2 * d->batch));
      hipLaunchKernelGGL(k_tile_occ, og, dim3(256), 0, ctx->stream, oa, ob, g.K, g.nkw, (int)d->batch, ctx->skip_flag);
    }
    g.amask = sa ? am : nullptr;
    g.bmask = sb_ ? bmk : nullptr;
  }
  TmpBuf PERM(ctx);
  const long long ntile_all = (long long)g.tiles_m * g.tiles_n;
  if (d->batch == 1 && (g.amask || g.bmask) && ntile_all * g.ksplit > 2 * n_cu && ntile_all <= 2048) {
    // inside a Krylov solve both masks are the solve's (cached environment mask, structural centre mask): one sort
    // serves every matvec of the solve
    const bool keep = ctx->occ_cache_on && mask_a_stable && mask_b_stable;
    int* pp = nullptr;
    if (keep)
      for (const auto& e : ctx->perm_cache)
        if (e.amask == g.amask && e.bmask == g.bmask && e.tiles_m == g.tiles_m && e.tiles_n == g.tiles_n && e.nkt == nkt_all)
          pp = static_cast<int*>(e.perm);
    if (!pp) {
      if (keep) {
        void* pm = nullptr;
        MPSE_TRY(mpse_malloc(ctx, size_t(ntile_all) * sizeof(int), &pm));
        ctx->perm_cache.push_back({g.amask, g.bmask, g.tiles_m, g.tiles_n, nkt_all, pm});
        pp = static_cast<int*>(pm);
      } else {
        MPSE_TRY(PERM.alloc(size_t(ntile_all) * sizeof(int)));
        pp = PERM.as<int>();
      }
      hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, ctx->stream, g.amask, g.bmask, g.nkw, nkt_all, g.tiles_m,
                         g.tiles_n, 1, pp, ctx->skip_flag, GemmGroups(), (unsigned char*)nullptr, 0);
    }
    g.perm = pp;
  }
  // the fast kernel addresses its operands as uniform base + 32-bit lane offset: non-negative strides, spans < 4 GB
  auto span_ok = [](const IdxMap& m, const IdxMap& k, bool cplx) {
    if (m.s_hi < 0 || m.s_lo < 0 || k.s_hi < 0 || k.s_lo < 0) return false;
    auto span = [](const IdxMap& x) {
      if (x.ext <= 1) return 0.0;
      if (x.lo == 0x7fffffff) return double(x.ext - 1) * double(x.s_lo);
      return double((x.ext - 1) / x.lo) * double(x.s_hi) + double(x.lo - 1) * double(x.s_lo);
    };
    return (span(m) + span(k) + 1.0) * (cplx ? 16.0 : 8.0) < 4.0e9;
  };
  const bool ks = is_single(g.kA) && is_single(g.kB) && span_ok(g.mA, g.kA, ca) && span_ok(g.nB, g.kB, cb);
  if (!g.perm && g.ksplit == 1 && d->batch == 1 && (long long)g.tiles_m * g.tiles_n >= 64) {
    const double size_a = double(g.M) * (ca ? 2 : 1), size_b = double(g.N) * (cb ? 2 : 1);   // per unit of K
    if (size_a >= size_b && g.tiles_m % 8 == 0)
      g.die_group = 1;
    else if (g.tiles_n % 8 == 0)
      g.die_group = 2;
    else if (g.tiles_m % 8 == 0)
      g.die_group = 1;
  }
#define MPSE_LAUNCH(CA_, CB_)                                                                         \
  do {                                                                                                \
    if (ks && wide)                                                                                   \
      hipLaunchKernelGGL((k_gemm<CA_, CB_, true, 2, 1>), grid, dim3(512), 0, ctx->stream, g);         \
    else if (ks)                                                                                      \
      hipLaunchKernelGGL((k_gemm<CA_, CB_, true, 2>), grid, block, 0, ctx->stream, g);                \
    else                                                                                              \
      hipLaunchKernelGGL((k_gemm<CA_, CB_, false, 2>), grid, block, 0, ctx->stream, g);               \
  } while (0)
  // one workgroup per compute unit (or fewer): eight waves on the tile instead of four
  const bool wide = nblk <= n_cu && nkt_all >= 2;
  if (ca && cb)
    MPSE_LAUNCH(true, true);
  else if (ca)
    MPSE_LAUNCH(true, false);
  else if (cb)
    MPSE_LAUNCH(false, true);
  else
    MPSE_LAUNCH(false, false);
#undef MPSE_LAUNCH
  if (g.ksplit > 1 && !leave_slices) {
    const dim3 rgrid((unsigned)rgx, (unsigned)rgy);
    if (ca || cb)
      hipLaunchKernelGGL((k_splitk_reduce<true>), rgrid, dim3(256), 0, ctx->stream, g, (int)d->batch);
    else
      hipLaunchKernelGGL((k_splitk_reduce<false>), rgrid, dim3(256), 0, ctx->stream, g, (int)d->batch);
  }
  if (prof_this) prof_end(ctx, rec);
  MPSE_HIP(ctx, hipGetLastError());
  return MPSE_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Tile-occupancy flags of one operand (rows through `r`, K through `k`, both from `ptr`): from the cache of the
// running Krylov solve, else scanned now - into the cache when the operand lies inside the solve's environment
// ranges, into `tmp` otherwise.  flags[t * pitch + kt], 64 rows per tile t.
int occ_mask_get(mpse_ctx* ctx, const void* ptr, int dtype, mpse_index r, mpse_index k, TmpBuf& tmp,
                 const unsigned char** flags, int* pitch, bool* stable) {
  IdxMap rm, km;
  if (!to_map(r, &rm) || !to_map(k, &km) || !is_single(km))
    return mpse_fail(ctx, MPSE_ERR_SHAPE, "occupancy scan: K index must be single level");
  const int K = km.ext, nrows = rm.ext, nkt = (K + BK - 1) / BK, nkw = (nkt + 7) / 8, tiles = (nrows + BM - 1) / BM;
  const bool cplx = dtype == MPSE_C128;
  auto fastest = [](const IdxMap& m) { return m.ext <= 1 ? (long long)1 << 60 : (m.s_lo < 0 ? -m.s_lo : m.s_lo); };
  const int kfast = fastest(km) <= fastest(rm);
  mpse_ctx::OccKey key;
  memset(&key, 0, sizeof(key));
  key.ptr = ptr;
  key.r_ext = rm.ext, key.r_lo = rm.lo, key.r_shi = rm.s_hi, key.r_slo = rm.s_lo;
  key.k_ext = km.ext, key.k_lo = km.lo, key.k_shi = km.s_hi, key.k_slo = km.s_lo;
  key.sb = 0, key.nrows = nrows, key.tiles = tiles, key.nkw = nkw, key.batch = 1, key.K = K, key.cplx = cplx ? 1 : 0;
  const char* c = reinterpret_cast<const char*>(ptr);
  const bool cacheable = ctx->occ_cache_on && ((c >= ctx->occ_lo[0] && c < ctx->occ_hi[0]) ||
                                               (c >= ctx->occ_lo[1] && c < ctx->occ_hi[1]));
  *pitch = nkw * 8;
  *stable = cacheable;
  if (cacheable)
    for (const auto& e : ctx->occ_cache)
      if (memcmp(&e.key, &key, sizeof(key)) == 0) {
        *flags = static_cast<const unsigned char*>(e.mask);
        return MPSE_OK;
      }
  void* pm = nullptr;
  const size_t bytes = size_t(tiles) * nkw * 8;
  if (cacheable) {
    MPSE_TRY(mpse_malloc(ctx, bytes, &pm));
    ctx->occ_cache.push_back({key, pm});
  } else {
    MPSE_TRY(tmp.alloc(bytes));
    pm = tmp.p;
  }
  OccOperand oa{static_cast<const double*>(ptr), rm, km, nrows, tiles, cplx ? 1 : 0, kfast, 0, static_cast<unsigned char*>(pm)};
  OccOperand ob = oa;
  ob.tiles = 0;
  hipLaunchKernelGGL(k_tile_occ, dim3((nkt + 3) / 4, tiles, 2), dim3(256), 0, ctx->stream, oa, ob, K, nkw, 1, ctx->skip_flag);
  MPSE_HIP(ctx, hipGetLastError());
  *flags = static_cast<const unsigned char*>(pm);
  return MPSE_OK;
}

// Grouped launch of the contraction kernel (mpse_internal.h GroupedDesc; folded one-site matvec of mpse_plans.h).
int gemm_grouped(mpse_ctx* ctx, const GroupedDesc& d) {
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  if (d.ngrp < 1 || d.ngrp > GMAX_GRP) return mpse_fail(ctx, MPSE_ERR_ARG, "grouped product: 1 .. %d groups", GMAX_GRP);
  if (!to_map(d.ma, &g.mA) || !to_map(d.ka, &g.kA) || !to_map(d.kb, &g.kB) || !to_map(d.nb, &g.nB) ||
      !to_map(d.mc, &g.mC) || !to_map(d.nc, &g.nC))
    return mpse_fail(ctx, MPSE_ERR_SHAPE, "grouped product: extent out of range");
  g.M = g.mA.ext, g.N = g.nB.ext, g.K = g.kA.ext;
  if (g.M != g.mC.ext || g.N != g.nC.ext || g.K != g.kB.ext) return mpse_fail(ctx, MPSE_ERR_SHAPE, "grouped product: extents disagree");
  if (g.M == 0 || g.N == 0) return MPSE_OK;
  const bool ca = d.dta == MPSE_C128, cb = d.dtb == MPSE_C128;
  auto span_ok = [](const IdxMap& m, const IdxMap& k, bool cplx) {
    if (m.s_hi < 0 || m.s_lo < 0 || k.s_hi < 0 || k.s_lo < 0) return false;
    auto span = [](const IdxMap& x) {
      if (x.ext <= 1) return 0.0;
      if (x.lo == 0x7fffffff) return double(x.ext - 1) * double(x.s_lo);
      return double((x.ext - 1) / x.lo) * double(x.s_hi) + double(x.lo - 1) * double(x.s_lo);
    };
    return (span(m) + span(k) + 1.0) * (cplx ? 16.0 : 8.0) < 4.0e9;
  };
  if (!is_single(g.kA) || !is_single(g.kB) || !span_ok(g.mA, g.kA, ca) || !span_ok(g.nB, g.kB, cb) || g.K % BK != 0 ||
      g.K < BK || (d.ngrp > 1 && g.M % BM != 0) || !cb)
    return mpse_fail(ctx, MPSE_ERR_SHAPE, "grouped product: needs single-level K indices, K a multiple of %d, group "
                     "heights a multiple of %d and a complex second operand", BK, BM);
  g.mtiles_m = (g.M + BM - 1) / BM;
  g.mtiles_n = (g.N + BN - 1) / BN;
  g.tiles_m = g.mtiles_m * d.ngrp;
  g.tiles_n = g.mtiles_n;
  auto fast = [](const IdxMap& m) { return m.ext <= 1 ? (long long)1 << 60 : (m.s_lo < 0 ? -m.s_lo : m.s_lo); };
  g.a_kfast = fast(g.kA) <= fast(g.mA);
  g.b_kfast = fast(g.kB) <= fast(g.nB);
  g.alpha_re = 1.0, g.beta_re = 1.0;
  g.ksplit = 1;
  g.skip = ctx->skip_flag;
  g.skew = 1;
  g.trace = gemm_trace_ptr(ctx);
  GemmGroups& gg = g.gg;
  gg.ngrp = d.ngrp, gg.tiles_m_grp = g.mtiles_m, gg.nkt_seg = g.K / BK;
  gg.am_pitch = d.am_pitch, gg.bm_pitch = d.bm_pitch;
  gg.split2 = d.split2 ? 1 : 0;
  gg.C2 = static_cast<double*>(d.c2);
  gg.cpd = (ctx->n_cu > 0 ? ctx->n_cu : 256) / 8;
  if (d.split2 && (d.ngrp != 1 || !d.c2 || g.M % BM != 0))
    return mpse_fail(ctx, MPSE_ERR_ARG, "grouped product: halved tiles need one group, whole tile rows and a second result");
  int max_seg = 0;
  bool any_mask = false, any_beta = false;
  for (int i = 0; i < d.ngrp; ++i) {
    const GroupedGrp& s = d.grp[i];
    if (s.nseg < 1 || s.nseg > GMAX_SEG || !s.C) return mpse_fail(ctx, MPSE_ERR_ARG, "grouped product: bad group");
    gg.g[i].C = static_cast<double*>(s.C);
    gg.g[i].nseg = s.nseg;
    gg.g[i].use_beta = s.beta != 0.0 ? 1 : 0;
    if (s.beta != 0.0 && s.beta != 1.0) return mpse_fail(ctx, MPSE_ERR_ARG, "grouped product: beta must be 0 or 1");
    any_beta = any_beta || s.beta != 0.0;
    for (int q = 0; q < s.nseg; ++q) {
      if (!s.seg[q].A || !s.seg[q].B) return mpse_fail(ctx, MPSE_ERR_ARG, "grouped product: null operand");
      gg.g[i].seg[q] = GSeg{static_cast<const double*>(s.seg[q].A), static_cast<const double*>(s.seg[q].B), s.seg[q].am, s.seg[q].bm};
      any_mask = any_mask || s.seg[q].am || s.seg[q].bm;
    }
    max_seg = s.nseg > max_seg ? s.nseg : max_seg;
  }
  g.A = gg.g[0].seg[0].A, g.B = gg.g[0].seg[0].B, g.C = gg.g[0].C;
  g.use_beta = any_beta;
  const int nkt_max = max_seg * gg.nkt_seg;
  g.nkw = (any_mask && (nkt_max + 7) / 8 <= 64) ? (nkt_max + 7) / 8 : 0;
  const int n_cu = ctx->n_cu > 0 ? ctx->n_cu : 256;
  const long long ntile = (long long)g.tiles_m * g.tiles_n;
  if (ntile > 0x7fffffffLL) return mpse_fail(ctx, MPSE_ERR_SHAPE, "grouped product: grid too large");
  // the caller's dot request (the launch that completes a matvec result): one group only
  const bool want_dot = ctx->dot_now;
  ctx->dot_now = false;
  const long long nwg = ntile * (d.split2 ? 2 : 1);
  if (want_dot && d.ngrp == 1 && nwg >= 1 && nwg <= ctx->dot_req.cap) {
    g.dot_y = static_cast<const double*>(ctx->dot_req.y);
    g.dot_part = ctx->dot_req.part;
    ctx->dot_req.nb_out = (int)nwg;
  }
  mpse_ctx::ProfRec rec;
  const int variant = (ca ? 1 : 0) + (cb ? 2 : 0);
  double segs = 0;
  for (int i = 0; i < d.ngrp; ++i) segs += d.grp[i].nseg;
  const double mnk = double(g.M) * double(g.N) * double(g.K) * segs;
  const bool prof_this = prof_begin(ctx, variant, mnk * ((ca && cb) ? 8.0 : 4.0),
                                    segs * (double(g.M) * g.K * (ca ? 16 : 8) + double(g.K) * g.N * 16.0) +
                                        double(d.ngrp) * double(g.M) * g.N * 16.0 * (any_beta ? 2 : 1),
                                    &rec);
  g.kt_counter = prof_this && ctx->prof_ktiles ? ctx->prof_ktiles + variant : nullptr;
  TmpBuf PERM(ctx);
  // launch order by weight: products with more tiles than slots (heaviest first, die aware), and halved products (their
  // position -> (tile, half) map pairs heavy and light halves per compute unit through it; plain sorted order)
  if (g.nkw > 0 && (ntile > 2 * n_cu || d.split2) && ntile <= 2048) {
    int* pp = nullptr;
    const void *ka = gg.g[0].seg[0].am, *kb = gg.g[0].seg[0].bm;
    const int nkt_key = nkt_max + 1000 * d.ngrp;       // (grouped entries never collide with plain ones)
    const bool keep = ctx->occ_cache_on && d.masks_stable;
    if (keep)
      for (const auto& e : ctx->perm_cache)
        if (e.amask == ka && e.bmask == kb && e.tiles_m == g.tiles_m && e.tiles_n == g.tiles_n && e.nkt == nkt_key)
          pp = static_cast<int*>(e.perm);
    // (one buffer: the launch order, then the assembled flags of every tile - 8-byte aligned)
    const size_t perm_bytes = (size_t(ntile) * sizeof(int) + 7) & ~size_t(7), flag_pitch = size_t(g.nkw) * 8;
    if (!pp) {
      if (keep) {
        void* pm = nullptr;
        MPSE_TRY(mpse_malloc(ctx, perm_bytes + size_t(ntile) * flag_pitch, &pm));
        ctx->perm_cache.push_back({ka, kb, g.tiles_m, g.tiles_n, nkt_key, pm});
        pp = static_cast<int*>(pm);
      } else {
        MPSE_TRY(PERM.alloc(perm_bytes + size_t(ntile) * flag_pitch));
        pp = PERM.as<int>();
      }
      hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, ctx->stream, (const unsigned long long*)nullptr,
                         (const unsigned long long*)nullptr, 0, nkt_max, g.tiles_m, g.tiles_n, d.split2 ? 0 : 1, pp,
                         ctx->skip_flag, gg, reinterpret_cast<unsigned char*>(pp) + perm_bytes, (int)flag_pitch);
    }
    g.perm = pp;
    g.gg.flags = reinterpret_cast<const unsigned long long*>(reinterpret_cast<const char*>(pp) + perm_bytes);
  }
  if (!g.perm && ntile >= 64) {
    const double size_a = double(g.M) * d.ngrp * (ca ? 2 : 1), size_b = double(g.N) * 2;
    if (size_a >= size_b && g.tiles_m % 8 == 0)
      g.die_group = 1;
    else if (g.tiles_n % 8 == 0)
      g.die_group = 2;
    else if (g.tiles_m % 8 == 0)
      g.die_group = 1;
  }
  const dim3 grid((unsigned)nwg);
  const bool wide = nwg <= n_cu;
  if (ca) {
    if (wide)
      hipLaunchKernelGGL((k_gemm<true, true, true, 2, 1, true>), grid, dim3(512), 0, ctx->stream, g);
    else
      hipLaunchKernelGGL((k_gemm<true, true, true, 2, 2, true>), grid, dim3(256), 0, ctx->stream, g);
  } else {
    if (wide)
      hipLaunchKernelGGL((k_gemm<false, true, true, 2, 1, true>), grid, dim3(512), 0, ctx->stream, g);
    else
      hipLaunchKernelGGL((k_gemm<false, true, true, 2, 2, true>), grid, dim3(256), 0, ctx->stream, g);
  }
  if (prof_this) prof_end(ctx, rec);
  MPSE_HIP(ctx, hipGetLastError());
  return MPSE_OK;
}

extern "C" int mpse_gemm(mpse_ctx* ctx, const mpse_gemm_desc* d, const void* A, const void* B, void* C) {
  if (ctx && d && MPSE_RECORDING(ctx)) {
    const mpse_gemm_desc dc = *d;
    ctx->defer_ops[ctx->defer_recording].push_back([ctx, dc, A, B, C] { return mpse_gemm(ctx, &dc, A, B, C); });
    return MPSE_OK;
  }
  return gemm_impl(ctx, d, A, B, C, d ? (d->skip_zero_tiles & 3) : 0);
}

int gemm_call(mpse_ctx* ctx, int dta, int dtb, int conja, int conjb, mpse_index ma, mpse_index ka, mpse_index kb,
              mpse_index nb, mpse_index mc, mpse_index nc, int64_t batch, int64_t sba, int64_t sbb, int64_t sbc,
              const void* A, const void* B, void* C, double alpha, double beta, int skip_zero) {
  mpse_gemm_desc d;
  d.dtype_a = dta;
  d.dtype_b = dtb;
  d.conj_a = conja;
  d.conj_b = conjb;
  d.m_a = ma;
  d.k_a = ka;
  d.k_b = kb;
  d.n_b = nb;
  d.m_c = mc;
  d.n_c = nc;
  d.batch = batch;
  d.sb_a = sba;
  d.sb_b = sbb;
  d.sb_c = sbc;
  d.alpha_re = alpha;
  d.alpha_im = 0.0;
  d.beta_re = beta;
  d.beta_im = 0.0;
  d.skip_zero_tiles = skip_zero;
  return gemm_impl(ctx, &d, A, B, C, skip_zero);
}

extern "C" int mpse_transpose_inner(mpse_ctx* ctx, int dtype, void* out, const void* in, int64_t d0, int64_t d1,
                                    int64_t d2, int conj) {
  if (!ctx) return MPSE_ERR_ARG;
  MPSE_BIND(ctx);
  if (d0 <= 0 || d1 <= 0 || d2 <= 0) return MPSE_OK;
  if (!out || !in) return MPSE_ERR_ARG;
  long long nblk = d0 * ((d1 + 31) / 32) * ((d2 + 31) / 32);
  if (nblk > 0x7fffffffLL) return mpse_fail(ctx, MPSE_ERR_SHAPE, "transpose grid too large");
  if (dtype == MPSE_C128)
    hipLaunchKernelGGL((k_transpose_inner<true>), dim3((unsigned)nblk), dim3(256), 0, ctx->stream, (double*)out,
                       (const double*)in, (long long)d0, (int)d1, (int)d2, conj);
  else
    hipLaunchKernelGGL((k_transpose_inner<false>), dim3((unsigned)nblk), dim3(256), 0, ctx->stream, (double*)out,
                       (const double*)in, (long long)d0, (int)d1, (int)d2, 0);
  MPSE_HIP(ctx, hipGetLastError());
  return MPSE_OK;
}
