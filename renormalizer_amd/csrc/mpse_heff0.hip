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
//
// The same launch serves one-site centres with a two-level physical index (the electronic sites of the headline chain:
// abc,bdef,lfk,cek->adl, mps/hop_expr.py:75-79, W real).  The MPO site is a handful of numbers per (left channel, right
// channel) pair; a workgroup is (16 bra rows, RIGHT channel f, 64 ket columns), and step 1 builds
//   P_x[a, k] = sum over the non-zero W[b, x, e, f] of  W[b, x, e, f] . L[a, b, :] . C[:, e, k]      (x = 0, 1)
// as one MFMA chain per term with the scalar folded into the A operand, step 2 multiplies both P_x by the same rows of
// Rt.  The plans ran this matvec as product + reduction + elementwise MPO step + product + reduction.
#include "mpse_device.h"
#include "mpse_internal.h"

typedef double v4d __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ v4d mfma(double a, double b, v4d c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

constexpr int F0_TMAX = 16;     // terms (non-zero W[b, x, e, f]) per right channel f
struct F0Term {
  int b, e, x, pad;
  double w;
};
// What a launch position needs to start, in ONE read (k_f0_plan): before round 6 a workgroup went to memory four times in a
// row before its first operand load - part mask, term count, terms, tile flags, ~1.5 us each and nothing to hide them
// behind (the timeline of profiles/r06_f0_trace.md: 4 - 6 k cycles of prelude).  The compact form holds for bonds of up to
// 256 (16 tiles a side) and at most four terms per (channel, x); other shapes keep nt = -1 and read the tables.
struct F0Info {
  int unit;                // (at * nparts + s) * d + x: also the slot of the dot partials
  int nt;                  // terms of (f, x) listed below (0 .. 4), or -1
  unsigned long long lts;  // bit lt: the unit holds output tile lt
  unsigned long long frn;  // 4 bits per l tile: k tiles of the chunk where R has data
  unsigned long long cts;  // 16 bits per term: c tiles where L and C both hold data
  unsigned bpack, epack;   // 8 bits per term: left channel b, physical index e of the centre
  double w[4];
};
struct F0Args {
  const double* L;        // (Dl, wl, Dl)
  const double* Rt;       // (wr, Dr, Dr): Rt[f, k, l] = R[l, f, k]
  const double* C;        // (Dl, d, Dr)
  double* parts;          // part s at parts + s * n (complex elements), laid out like out (Dl, d, Dr)
  const double* y;        // optional: dot partner laid out like out
  double* dot_part;       // one (re, im) per workgroup
  const unsigned char* FL;   // [(at * wl + b) * ntl + ct]
  const unsigned char* FC;   // [(e * nkc + kc) * fc_pitch + ct] (centre mask), or null
  const unsigned char* FR;   // [(lt * wr + f) * ntr + kt]
  const unsigned long long* mask;   // [(at * d + x) * ntr + lt]: bit s = part s holds this tile (same word for all x)
  const F0Term* terms;    // [f * F0_TMAX + t]
  const int* nterm;       // [f]
  const int* skip;
  const F0Info* info;     // launch position -> unit, heaviest first, with what it needs to start (k_f0_plan); null: plain order
  long long n;
  int Dl, Dr, wl, wr, d, ntl, ntr, nkc, fc_pitch;
  unsigned long long* trace;   // debug timeline (mpse_ctx::gemm_trace, MPSE_GEMM_TRACE), null normally
};

// Rt[(b, k), l] = R[l, b, k]  (block `blk` of `nblk`)
__device__ __forceinline__ void f0_transpose(double2* __restrict__ rt, const double2* __restrict__ r, int D, int w, int blk,
                                             int nblk) {
  const long long n = (long long)D * w * D;
  const long long stride = (long long)nblk * 256;
  for (long long i = (long long)blk * 256 + threadIdx.x; i < n; i += stride) {
    const int l = (int)(i % D);
    const long long bk = i / D;
    rt[i] = r[(long long)l * w * D + bk];
  }
}

// flags of the 16 x 16 tiles of an environment E (D, w, D) viewed per channel: F[(rt * w + b) * nt + ct] = any non-zero in
// rows [16 rt, 16 rt + 16), channel b, columns [16 ct, 16 ct + 16).  One block of 256 threads = one tile row (rt, b).
__device__ __forceinline__ void f0_flags(const double2* __restrict__ E, int D, int w, int nt, unsigned char* __restrict__ F,
                                         int rt, int b) {
  __shared__ int s_f[64];
  const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
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

// Per-solve preparation in ONE launch (three before): blocks [0, ntl wl) flag the tiles of L, the next ntr wr blocks those
// of R, the rest transpose R - the three jobs do not depend on each other, and a launch at the head of a solve is ~8 us
// of latency in front of its first matvec.
__global__ __launch_bounds__(256) void k_f0_prepare(const double2* __restrict__ L, const double2* __restrict__ R,
                                                     double2* __restrict__ rt, unsigned char* __restrict__ FL,
                                                     unsigned char* __restrict__ FR, int Dl, int Dr, int wl, int wr, int ntl,
                                                     int ntr, const int* __restrict__ skip) {
  if (skip && *skip) return;
  int blk = blockIdx.x;
  if (blk < ntl * wl) {
    f0_flags(L, Dl, wl, ntl, FL, blk / wl, blk % wl);
    return;
  }
  blk -= ntl * wl;
  if (blk < ntr * wr) {
    f0_flags(R, Dr, wr, ntr, FR, blk / wr, blk % wr);
    return;
  }
  blk -= ntr * wr;
  f0_transpose(rt, R, Dr, wr, blk, (int)gridDim.x - ntl * wl - ntr * wr);
}

// which parts hold which output tile: bit s = f * nkc + kc of the word of (bra tile row at, l tile lt) is set when step 1
// of workgroup (at, f, kc) has, for one of the terms of f, a c tile with data on both sides AND R has data in rows lt,
// channel f, k chunk kc.  The word is stored for each of the d tile rows (at * d + x) of the result.  One workgroup.
__global__ __launch_bounds__(1024) void k_f0_valid(const unsigned char* __restrict__ FL, const unsigned char* __restrict__ FC,
                                                    const unsigned char* __restrict__ FR, const F0Term* __restrict__ terms,
                                                    const int* __restrict__ nterm, int wl, int wr, int d, int ntl, int ntr,
                                                    int nkc, int fc_pitch, unsigned long long* __restrict__ mask,
                                                    const int* __restrict__ skip) {
  if (skip && *skip) return;
  __shared__ unsigned long long s1[64], s2[64];    // per bra tile row / per l tile: bit s
  const int nparts = wr * nkc, tid = threadIdx.x;
  if (tid < 64) s1[tid] = s2[tid] = 0;
  __syncthreads();
  for (int t = tid; t < ntl * nparts; t += 1024) {
    const int at = t / nparts, s = t - at * nparts, f = s / nkc, kc = s - f * nkc;
    bool any = false;
    for (int q = 0; q < nterm[f]; ++q) {
      const F0Term tm = terms[f * F0_TMAX + q];
      for (int ct = 0; ct < ntl; ++ct)
        any = any || (FL[((long long)at * wl + tm.b) * ntl + ct] && (!FC || FC[(tm.e * nkc + kc) * fc_pitch + ct]));
    }
    if (any) atomicOr(&s1[at], 1ull << s);
  }
  for (int t = tid; t < ntr * nparts; t += 1024) {
    const int lt = t / nparts, s = t - lt * nparts, f = s / nkc, kc = s - f * nkc;
    bool any = false;
    for (int j = 0; j < 4; ++j) {
      const int kt = 4 * kc + j;
      if (kt < ntr) any = any || FR[((long long)lt * wr + f) * ntr + kt];
    }
    if (any) atomicOr(&s2[lt], 1ull << s);
  }
  __syncthreads();
  for (int t = tid; t < ntl * d * ntr; t += 1024) {
    const int row = t / ntr, lt = t - row * ntr;
    mask[t] = s1[row / d] & s2[lt];
  }
}

// Part mask AND launch order of the units of a solve in one launch (one workgroup, once per solve like the flags;
// MPSE_F0_ORDER=0: k_f0_valid alone, units in index order).
// Why an order: the dispatcher deals launch positions to the dies round robin and, on a die, gives every compute unit one
// workgroup before any gets a second; a third of the units have no work and return at once.  In plain order a bond launch of
// 320 units left 90 compute units with nothing but empty units while 31 held TWO working ones - and a working workgroup
// that shares its unit's MFMA pipes lives 25 us instead of 20 (profiles/r06_f0_trace.md).  Order: the n_cu heaviest units
// first, heaviest first (one compute unit each); then the other working units LIGHTEST first (position n_cu + k joins the
// unit of position k: the heaviest gets the lightest partner); the empty units last.  Weight = c tiles of step 1 + l tiles
// of step 2 (the fit of the timeline gives both ~0.6 us).  Only the schedule changes: every unit writes the same tiles of
// the same part and the same dot-partial slot as before.
// The tile flags and the term table are staged in LDS first (one trip to memory; the first version read them from global
// memory per unit, twice, and cost 30 us per solve - 2.3 % of the kernel time of a step, most of what the order gained).
__global__ __launch_bounds__(1024) void k_f0_plan(const unsigned char* __restrict__ FL, const unsigned char* __restrict__ FC,
                                                   const unsigned char* __restrict__ FR, const F0Term* __restrict__ terms,
                                                   const int* __restrict__ nterm, int wl, int wr, int d, int ntl, int ntr,
                                                   int nkc, int fc_pitch, unsigned long long* __restrict__ mask, int n_cu,
                                                   F0Info* __restrict__ info, int compact, const int* __restrict__ skip) {
  if (skip && *skip) return;
  // (the caller guarantees ntl, ntr <= 64, wl, wr <= 16 and at most MROWS flag rows of each kind: heff0_fused_parts)
  constexpr int WMAX = 256, MROWS = 1024;
  __shared__ unsigned long long s1[64], s2[64];    // per bra tile row / per l tile: bit s
  __shared__ int hist[WMAX], start[WMAX], s_nw;
  __shared__ unsigned long long mL[MROWS], mR[MROWS], mC[64];   // rows of FL / FR / FC as bit masks over their tiles
  __shared__ F0Term sT[16 * F0_TMAX];
  __shared__ int sNT[16];
  const int nparts = wr * nkc, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nu = ntl * nparts * d;
  // a flag row (nt <= 64 bytes) -> one word: a wave per row, a lane per tile, straight from global memory; the rows a wave
  // takes are requested together (four at a time), so that all three kinds of flags cost about one trip to memory
  auto rows_to_masks = [&](const unsigned char* F, int nrows, int nt, int pitch, unsigned long long* out) {
    for (int r0 = wave; r0 < nrows; r0 += 64) {
      unsigned char v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = r0 + 16 * j;
        v[j] = (r < nrows && lane < nt) ? F[(long long)r * pitch + lane] : (unsigned char)0;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = r0 + 16 * j;
        const unsigned long long m = __ballot(v[j] != 0);
        if (r < nrows && lane == 0) out[r] = m;
      }
    }
  };
  rows_to_masks(FL, ntl * wl, ntl, ntl, mL);
  rows_to_masks(FR, ntr * wr, ntr, ntr, mR);
  if (FC) rows_to_masks(FC, d * nkc, ntl, fc_pitch, mC);
  for (int i = tid; i < wr * F0_TMAX; i += 1024) sT[i] = terms[i];
  if (tid < wr) sNT[tid] = nterm[tid];
  if (tid < 64) s1[tid] = s2[tid] = 0;
  for (int i = tid; i < WMAX; i += 1024) hist[i] = 0;
  if (tid == 0) s_nw = 0;
  __syncthreads();
  const unsigned long long all_c = ntl >= 64 ? ~0ull : ((1ull << ntl) - 1);
  auto c_tiles = [&](int at, const F0Term& tm, int kc) {       // c tiles where L (rows at, channel b) and C (e, chunk kc) hold data
    return mL[at * wl + tm.b] & (FC ? mC[tm.e * nkc + kc] : all_c);
  };
  auto k_tiles = [&](int lt, int f, int kc) {                  // 4 bits: k tiles of chunk kc where R (rows lt, channel f) holds data
    return (unsigned)((mR[lt * wr + f] >> (4 * kc)) & 0xfull);
  };
  // ---- the part mask (as k_f0_valid)
  for (int t = tid; t < ntl * nparts; t += 1024) {
    const int at = t / nparts, sp = t - at * nparts, f = sp / nkc, kc = sp - f * nkc;
    bool any = false;
    for (int q = 0; q < sNT[f]; ++q) any = any || c_tiles(at, sT[f * F0_TMAX + q], kc) != 0;
    if (any) atomicOr(&s1[at], 1ull << sp);
  }
  for (int t = tid; t < ntr * nparts; t += 1024) {
    const int lt = t / nparts, sp = t - lt * nparts, f = sp / nkc, kc = sp - f * nkc;
    if (k_tiles(lt, f, kc)) atomicOr(&s2[lt], 1ull << sp);
  }
  __syncthreads();
  for (int t = tid; t < ntl * d * ntr; t += 1024) {
    const int row = t / ntr, lt = t - row * ntr;
    mask[t] = s1[row / d] & s2[lt];
  }
  // ---- a unit's record and weight (everything from LDS)
  auto record = [&](int u, F0Info& I) {
    const int x = u % d, wg0 = u / d, at = wg0 / nparts, sp = wg0 - at * nparts, f = sp / nkc, kc = sp - f * nkc;
    I = F0Info{};
    I.unit = u;
    I.nt = -1;
    int nlt = 0, nct = 0, nt = 0;
    if ((s1[at] >> sp) & 1ull)
      for (int lt = 0; lt < ntr; ++lt)
        if ((s2[lt] >> sp) & 1ull) {
          ++nlt;
          I.lts |= 1ull << lt;
          if (lt < 16) I.frn |= (unsigned long long)k_tiles(lt, f, kc) << (4 * lt);
        }
    if (nlt)
      for (int q = 0; q < sNT[f]; ++q) {
        const F0Term tm = sT[f * F0_TMAX + q];
        if (tm.x != x) continue;
        const unsigned long long c = c_tiles(at, tm, kc);
        nct += __builtin_popcountll(c);
        if (compact && nt < 4) {
          I.cts |= (c & 0xffffull) << (16 * nt);
          I.bpack |= (unsigned)tm.b << (8 * nt), I.epack |= (unsigned)tm.e << (8 * nt), I.w[nt] = tm.w;
        }
        ++nt;
      }
    if (compact) I.nt = nt;
    return nlt ? min(WMAX - 1, 1 + nct + nlt) : 0;
  };
  F0Info mine;                       // (launches of up to 1024 units: the record stays in registers across the sort)
  int myw = 0;
  for (int u = tid; u < nu; u += 1024) {
    F0Info I;
    const int w = record(u, I);
    if (u == tid) mine = I, myw = w;
    atomicAdd(&hist[w], 1);
    if (w) atomicAdd(&s_nw, 1);
  }
  __syncthreads();
  if (tid < WMAX) {                  // descending weights: start[w] = units heavier than w
    int acc = 0;
    for (int w = tid + 1; w < WMAX; ++w) acc += hist[w];
    start[tid] = acc;
  }
  __syncthreads();
  const int nw = s_nw;
  for (int u = tid; u < nu; u += 1024) {
    F0Info I;
    int w;
    if (u == tid)
      I = mine, w = myw;
    else
      w = record(u, I);
    const int rank = start[w] + atomicAdd(&hist[w], -1) - 1;       // (any order among equal weights: only the schedule)
    int pos = rank;
    if (rank >= n_cu && rank < nw) pos = n_cu + (nw - 1 - rank);
    info[pos] = I;
  }
}

// A unit = (bra tile row, part, value x of the physical index of the result - 0 for a bond matrix): the terms of the other
// x are another workgroup's - half the chain of the heaviest workgroups, which set the duration of the launch.  Launch
// positions map to units through g.info (k_f0_plan: heaviest first).
// Two workgroups per compute unit (256 registers per lane, 24 bytes of scratch): the launch of a two-level site has 640
// workgroups for 256 compute units and is as long as the sum of their chains, not as the longest one - a second
// workgroup's MFMAs fill the first one's waits (site launch 54 -> 45 us, 382 -> 375 ms of kernels per two steps).
__global__ __launch_bounds__(256, 2) void k_heff0_fused(const F0Args g) {
  constexpr int DX = 1;
  constexpr int NW = 4;   // waves per workgroup (eight - two groups splitting the c tiles of step 1, added up in LDS -
                          // spilled registers and lost: 495 against 510 site-updates/s)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), x = lane & 15,
            kq = lane >> 4;
  __shared__ double sTr[DX][16 * 65], sTi[DX][16 * 65];
  __shared__ double s_dot[8];
  // this position's record: the unit and - compact form - everything it needs before its first operand load.  Requested
  // ahead of the skip word of the solve: the two reads travel together instead of one after the other
  F0Info I;
  if (g.info) {
    I = g.info[blockIdx.x];
  } else {
    I.unit = (int)blockIdx.x;
    I.nt = -1;
  }
  asm volatile("" ::: "memory");
  if (g.skip && *g.skip) return;
  // debug timeline (tools/f0_trace.py): entry, flags in hand, end of step 1, end of step 2, exit - shader cycles
  unsigned long long tr[4] = {0, 0, 0, 0};
  const unsigned long long rt0 = g.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
  if (g.trace) tr[0] = __builtin_readcyclecounter();
  int tr_ct = 0, tr_lt = 0, tr_x = 0;
  auto trace_out = [&]() {
    if (g.trace && tid == 0) {
      const unsigned long long slot = atomicAdd(g.trace, 1ull);
      if (slot < GEMM_TRACE_CAP) {
        unsigned long long* r = g.trace + 1 + slot * GEMM_TRACE_WORDS;
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        r[0] = ((unsigned long long)(gridDim.x / g.d) << 32) | (unsigned long long)blockIdx.x;
        r[1] = (1ull << 63) | ((unsigned long long)g.d << 32) | (unsigned long long)tr_x;
        r[2] = (unsigned long long)(unsigned)tr_ct | ((unsigned long long)(unsigned)tr_lt << 16) |
               ((unsigned long long)(((xc & 0xf) << 16) | (hw & 0xffff)) << 32);
        r[3] = tr[0], r[4] = tr[1], r[5] = tr[2], r[6] = tr[3];
        r[7] = __builtin_readcyclecounter();
        r[8] = rt0;
        r[9] = __builtin_amdgcn_s_memrealtime();
      }
    }
  };
  // (Round 6, measured and dropped: several workgroups per unit, each forming the unit's T itself and multiplying every
  // n-th of its l tiles - the verdict's reading was that the launch lasts as long as the chain of its heaviest unit.  It
  // does, and splitting still lost: 546 -> 517 site-updates/s with two workgroups per unit, 458 with four, alternating runs
  // on one box, profiles/r06_ab_qr_f0split.txt - the extra workgroups land on compute units that already hold a working one,
  // and two working workgroups on a unit share its MFMA pipes (25 against 20 us of life, profiles/r06_f0_trace.md).  What
  // helped is WHERE the working workgroups run: k_f0_plan.)
  // Launch position -> (bra tile row, part), plain order.  (Measured and dropped: a die - launch position mod 8, one L2
  // each - taking whole parts, so that its L2 holds only the panels of C and Rt its workgroups share.  Dealt channel-major
  // the heavy parts - ket chunks that straddle two quantum-number sectors - piled up on two dies: 39 us against 33; dealt
  // chunk-major 32 against 29.)
  const int nparts = g.wr * g.nkc;
  const bool compact = I.nt >= 0;                    // (uniform)
  const int wg = I.unit;                             // also the slot of its dot partials
  const int wg0 = wg / g.d;
  const int xo_wg = wg - wg0 * g.d;                  // the value x of the physical index of the result this workgroup forms
  tr_x = xo_wg;
  const int at = wg0 / nparts, s = wg0 - at * nparts, f = s / g.nkc, kc = s - f * g.nkc;
  // which output tiles this workgroup holds (the mask is the single statement of that rule): lane lt looks at tile lt
  const unsigned long long lts =
      compact ? I.lts
              : __ballot(lane < g.ntr && ((g.mask[(long long)at * g.d * g.ntr + min(lane, g.ntr - 1)] >> s) & 1ull));
  if (lts == 0) {
    if (g.dot_part && tid == 0) {
      g.dot_part[2 * wg] = 0.0;
      g.dot_part[2 * wg + 1] = 0.0;
    }
    trace_out();
    return;
  }
  // ---- step 1: P_x[a, k] for this wave's 16 columns k of the chunk
  const int wcolt = wave;
  const int a0 = 16 * at, k0 = 64 * kc + 16 * wcolt;
  const bool wcol = k0 < g.Dr;                       // (last chunk of a Dr that is not a multiple of 64)
  // (every flag this workgroup will consult is requested here, together: a dependent trip to memory costs ~1.5 us, and
  // a workgroup has no neighbour on its compute unit to hide it behind)
  unsigned frn = 0;          // lane lt: bit j = R has data in rows lt, channel f, k tile 4 kc + j
  if (compact) {
    frn = lane < 16 ? (unsigned)((I.frn >> (4 * lane)) & 0xfull) : 0u;
  } else if (lane < g.ntr) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int kt = 4 * kc + jj;
      if (kt < g.ntr && g.FR[((long long)lane * g.wr + f) * g.ntr + kt]) frn |= 1u << jj;
    }
  }
  const int lc = min(lane, g.ntl - 1);
  const int nt = compact ? I.nt : g.nterm[f];
  // complex products as three real ones (3M): P1 = sum ar br, P2 = sum ai bi, P3 = sum (ar + ai)(br + bi);
  // re = P1 - P2, im = P3 - P1 - P2 - a quarter fewer MFMAs on the chain of the heaviest workgroups, which set the
  // duration of the launch (the contraction kernel of mpse_gemm.hip forms its complex products the same way)
  v4d t1[DX], t2[DX], t3[DX];
#pragma unroll
  for (int i = 0; i < DX; ++i) {
    t1[i] = v4d{0, 0, 0, 0};
    t2[i] = v4d{0, 0, 0, 0};
    t3[i] = v4d{0, 0, 0, 0};
  }
  // operands of three c tiles in flight (the tensors come from the memory-side cache: ~1.5 us away, and a workgroup has
  // no neighbour on its compute unit to hide that behind)
  double2 av[3][4], bv[3][4];
  for (int q = 0; q < nt; ++q) {
    int b, te;
    double wv;
    unsigned long long cts;
    if (compact) {                      // (the record lists the terms of this x only)
      b = (int)((I.bpack >> (8 * q)) & 0xffu), te = (int)((I.epack >> (8 * q)) & 0xffu);
      wv = q == 0 ? I.w[0] : q == 1 ? I.w[1] : q == 2 ? I.w[2] : I.w[3];
      cts = (I.cts >> (16 * q)) & 0xffffull;
    } else {
      const F0Term tm = g.terms[f * F0_TMAX + q];
      b = tm.b, te = tm.e, wv = tm.w;
      if (tm.x != xo_wg) continue;      // (uniform)
      cts = __ballot(lane < g.ntl && g.FL[((long long)at * g.wl + b) * g.ntl + lc] &&
                     (!g.FC || g.FC[(long long)(te * g.nkc + kc) * g.fc_pitch + lc]));
    }
    const double2* La = reinterpret_cast<const double2*>(g.L) + ((long long)(a0 + x) * g.wl + b) * g.Dl + kq;   // + c
    const double2* Cb = reinterpret_cast<const double2*>(g.C) + ((long long)kq * g.d + te) * g.Dr + (wcol ? k0 : 0) + x;
    const long long cstride = (long long)g.d * g.Dr;                                                    // + c * d * Dr
    auto load1 = [&](int slot, int ct) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int c = 16 * ct + 4 * kk;
        av[slot][kk] = La[c];
        bv[slot][kk] = Cb[(long long)c * cstride];
      }
    };
    auto mul1 = [&](int slot) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const double ar = av[slot][kk].x * wv, ai = av[slot][kk].y * wv;
#pragma unroll
        for (int i = 0; i < DX; ++i) {
          {
            t1[i] = mfma(ar, bv[slot][kk].x, t1[i]);
            t2[i] = mfma(ai, bv[slot][kk].y, t2[i]);
            t3[i] = mfma(ar + ai, bv[slot][kk].x + bv[slot][kk].y, t3[i]);
          }
        }
      }
    };
    if (g.trace) {
      if (!tr[1]) tr[1] = __builtin_readcyclecounter();
      tr_ct += __builtin_popcountll(cts);
    }
    if (wcol && cts) {
      // Three c tiles in flight, slots used in a fixed rotation.  The compiler barriers keep the order "request the tile
      // after next, THEN multiply the oldest": without them the scheduler sinks every batch of loads to just before its
      // use (fewer live registers) and each tile waits for its own trip to memory - measured: 3 600 cycles per tile
      // against the 1 024 of its sixteen MFMAs.
      unsigned long long m = cts;       // tiles still to load
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
  }
  if (g.trace) {
    tr[2] = __builtin_readcyclecounter();
    tr_lt = __builtin_popcountll(lts);
  }
  // ---- step 2: the l tiles this workgroup holds, dealt to the waves in order; the A operands (P_x) of the whole chunk
  // stay in registers
  // (the A operands of step 2 are read from LDS where they are used: sixty-four registers less per lane, which is what
  // lets two workgroups share a compute unit)
  double dre = 0.0, dim = 0.0;
  const double2* Rb = reinterpret_cast<const double2*>(g.Rt) + ((long long)f * g.Dr + 64 * kc + kq) * g.Dr + x;
  double2* part = reinterpret_cast<double2*>(g.parts) + (long long)s * g.n;
  // this wave's tiles: the (wave)-th, (wave + 4)-th, .. set bit of lts; two of them in flight
  auto nth_tile = [&](unsigned long long m, int n) {      // position of the n-th set bit, or -1
    for (int i = 0; i < n && m; ++i) m &= m - 1;
    return m ? (int)__builtin_ctzll(m) : -1;
  };
  auto k_tiles = [&](int lt) {                             // k tiles of the chunk where R has data for l tile lt
    return (unsigned)__builtin_amdgcn_readlane((int)frn, lt);
  };
  double2 rv[2][16], yv2[2][DX][4];
  auto load2 = [&](int slot, int lt) {
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      // (unconditional: a load guarded by the tile flag becomes load + select, and the select waits for the load on the
      // spot; k tiles without data are loaded - rows clamped into the tensor - and never multiplied)
      const long long krow = min((long long)(4 * ks), (long long)(g.Dr - 1 - 64 * kc - kq));
      rv[slot][ks] = Rb[krow * g.Dr + 16 * lt];
    }
    if (g.y) {
#pragma unroll
      for (int i = 0; i < DX; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          yv2[slot][i][r] =
              reinterpret_cast<const double2*>(g.y)[((long long)(a0 + kq + 4 * r) * g.d + xo_wg) * g.Dr + 16 * lt + x];
    }
  };
  auto mul2 = [&](int slot, int lt, unsigned kts) {
#pragma unroll
    for (int i = 0; i < DX; ++i) {
      v4d o1 = {0, 0, 0, 0}, o2 = {0, 0, 0, 0}, o3 = {0, 0, 0, 0};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        if ((kts >> (ks >> 2)) & 1u) {      // (uniform)
          const double xr = sTr[i][x * 65 + 4 * ks + kq], xi = sTi[i][x * 65 + 4 * ks + kq];
          o1 = mfma(xr, rv[slot][ks].x, o1);
          o2 = mfma(xi, rv[slot][ks].y, o2);
          o3 = mfma(xr + xi, rv[slot][ks].x + rv[slot][ks].y, o3);
        }
      }
      const v4d orr = o1 - o2, oi = o3 - o1 - o2;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long e = ((long long)(a0 + kq + 4 * r) * g.d + xo_wg) * g.Dr + 16 * lt + x;
        part[e] = make_double2(orr[r], oi[r]);
        if (g.y) {
          const double2 yv = yv2[slot][i][r];
          dre += orr[r] * yv.x + oi[r] * yv.y;      // conj(o) y
          dim += orr[r] * yv.y - oi[r] * yv.x;
        }
      }
    }
  };
  int idx = wave;
  int lt0 = nth_tile(lts, idx), lt1 = -1;
  unsigned kt0 = 0, kt1 = 0;
  if (lt0 >= 0) {
    kt0 = k_tiles(lt0);
    load2(0, lt0);
  }
  // The operands of this wave's first l tile are on their way BEFORE T goes through LDS and the workgroup meets at the
  // barrier: the trip to memory (~1.5 us) runs under the barrier instead of after it (step 2 of the heaviest workgroups
  // 22.5 - 25 k -> 21 - 22.4 k cycles in the timeline; neutral on the headline within the scatter of a box,
  // profiles/r06_ab_f0_compact_prefetch.txt).
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < DX; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sTr[i][(kq + 4 * r) * 65 + 16 * wcolt + x] = t1[i][r] - t2[i][r];
      sTi[i][(kq + 4 * r) * 65 + 16 * wcolt + x] = t3[i][r] - t1[i][r] - t2[i][r];
    }
  __syncthreads();
  while (lt0 >= 0) {
    idx += NW;
    lt1 = nth_tile(lts, idx);
    if (lt1 >= 0) {
      kt1 = k_tiles(lt1);
      load2(1, lt1);
    }
    asm volatile("" ::: "memory");     // (the next tile's operands are requested before this tile is multiplied)
    mul2(0, lt0, kt0);
    if (lt1 < 0) break;
    idx += NW;
    lt0 = nth_tile(lts, idx);
    if (lt0 >= 0) {
      kt0 = k_tiles(lt0);
      load2(0, lt0);
    }
    asm volatile("" ::: "memory");
    mul2(1, lt1, kt1);
  }
  if (g.trace) tr[3] = __builtin_readcyclecounter();
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
  trace_out();
}

}  // namespace

// Number of parts the fused matvec would deliver for this operator (0: not eligible).  Eligible: complex bond matrix
// (nsite = 0) or complex one-site centre with a two-level physical index and a real MPO site (nsite = 1, d = 2) between
// complex environments with equal bra / ket bonds, bond dimensions multiples of 16 (of 64 for the ket bond of a one-site
// centre: the chunks of the centre mask) and large enough that the plans' launches are not already cheap, at most 64
// parts.  MPSE_HEFF0=0 switches the path off.  (A one-site centre also needs the values of its MPO site on the host,
// mpse_mpo_site_hint: heff0_fused_try declines without them.)
int heff0_fused_parts(const mpse_heff* h, int dtype) {
  static const int mode = [] {
    const char* e = getenv("MPSE_HEFF0");
    return e ? atoi(e) : 1;
  }();
  if (mode == 0 || dtype != MPSE_C128 || h->l_dtype != MPSE_C128 || h->r_dtype != MPSE_C128) return 0;
  const mpse_dims& d = h->dims;
  const int64_t Dl = d.Dl_ket, Dr = d.Dr_ket;
  if (h->nsite == 0) {
    if (d.wr != d.wl) return 0;
  } else if (h->nsite == 1) {
    if (d.d0 != 2 || h->w_dtype != MPSE_F64 || !h->W0 || Dr % 64) return 0;
  } else {
    return 0;
  }
  if ((d.Dl_bra > 0 && d.Dl_bra != Dl) || (d.Dr_bra > 0 && d.Dr_bra != Dr) || d.danc > 1) return 0;
  if (Dl % 16 || Dr % 16 || Dl > 1024 || Dr > 1024 || d.wl < 1 || d.wr < 1 || d.wl > 16 || d.wr > 16) return 0;
  const int64_t min_d = mode >= 2 ? 16 : 128;            // (MPSE_HEFF0=2: every eligible shape - tests)
  if (Dl < min_d || Dr < min_d) return 0;
  const int64_t nparts = d.wr * ((Dr + 63) / 64);
  if (nparts > 64) return 0;
  return (int)nparts;
}

void heff0_drop_cache(mpse_ctx* ctx) {
  if (ctx->f0.buf) mpse_free(ctx, ctx->f0.buf);
  ctx->f0 = mpse_ctx::F0Cache();
}

// Runs the fused matvec when the caller offered masked parts (mpse_ctx::parts_req.masked_ok) with room for all of them.
// w_host: the MPO site (wl, d, d, wr) of a one-site centre as the host knows it (mpse_mpo_site_hint), else null.
int heff0_fused_try(mpse_ctx* ctx, int dtype, const mpse_heff* h, const void* C, const double* w_host, bool* taken) {
  *taken = false;
  mpse_ctx::PartsReq& pr = ctx->parts_req;
  const int nparts = heff0_fused_parts(h, dtype);
  if (nparts == 0 || !pr.ptr || !pr.masked_ok) return MPSE_OK;
  const int Dl = (int)h->dims.Dl_ket, Dr = (int)h->dims.Dr_ket, wl = (int)h->dims.wl, wr = (int)h->dims.wr;
  const int d = h->nsite == 1 ? (int)h->dims.d0 : 1;
  if (h->nsite == 1 && !w_host) return MPSE_OK;
  const long long n = (long long)Dl * d * Dr;
  if (pr.n != n || pr.cap_elems < (long long)nparts * n) return MPSE_OK;
  const int ntl = Dl / 16, ntr = Dr / 16, nkc = (Dr + 63) / 64;
  const int nwg = ntl * nparts;
  if (ctx->dot_req.y && nwg * d > ctx->dot_req.cap) return MPSE_OK;
  // the terms of every right channel: the non-zero entries W[b, x, e, f] (a bond matrix: the channel itself, factor 1)
  std::vector<F0Term> terms(size_t(wr) * F0_TMAX);
  std::vector<int> nterm(wr, 0);
  for (int f = 0; f < wr; ++f) {
    if (h->nsite == 0) {
      terms[size_t(f) * F0_TMAX] = F0Term{f, 0, 0, 0, 1.0};
      nterm[f] = 1;
      continue;
    }
    for (int b = 0; b < wl; ++b)
      for (int x = 0; x < d; ++x)
        for (int e = 0; e < d; ++e) {
          const double v = w_host[((size_t(b) * d + x) * d + e) * wr + f];
          if (v == 0.0) continue;
          if (nterm[f] == F0_TMAX) return MPSE_OK;      // (a denser site than this path is built for)
          terms[size_t(f) * F0_TMAX + nterm[f]++] = F0Term{b, e, x, 0, v};
        }
  }
  // the centre mask applies when it describes this shape: rows of C in 16s, columns (e, k) in 64s
  const unsigned char* FC = nullptr;
  int fc_pitch = 0;
  {
    const long long nkw = (ntl + 7) / 8;
    const char* pc = static_cast<const char*>(C);
    if (ctx->cmask.ptr && pc >= ctx->cmask.lo && pc < ctx->cmask.hi && ctx->cmask.bytes == (long long)d * nkc * nkw * 8) {
      FC = static_cast<const unsigned char*>(ctx->cmask.ptr);
      fc_pitch = (int)(nkw * 8);
    }
  }
  // per-solve data: transposed right environment, tile flags of L and R, the terms, the part mask
  const size_t rt_bytes = size_t(wr) * Dr * Dr * 16;
  const size_t fl_bytes = (size_t(ntl) * wl * ntl + 15) & ~size_t(15), fr_bytes = (size_t(ntr) * wr * ntr + 15) & ~size_t(15);
  const size_t tm_bytes = terms.size() * sizeof(F0Term), nt_bytes = (size_t(wr) * sizeof(int) + 15) & ~size_t(15);
  const size_t mk_bytes = size_t(ntl) * d * ntr * 8;
  const size_t nu = size_t(nwg) * d;
  const size_t od_bytes = (nu * sizeof(F0Info) + 15) & ~size_t(15);
  // compact records (F0Info): at most 16 tiles a side and four terms per (channel, x)
  bool compact = ntl <= 16 && ntr <= 16 && wl <= 255 && d <= 255;
  for (int f = 0; f < wr && compact; ++f)
    for (int x = 0; x < d; ++x) {
      int c = 0;
      for (int q = 0; q < nterm[f]; ++q) c += terms[size_t(f) * F0_TMAX + q].x == x;
      if (c > 4) compact = false;
    }
  const bool keep = ctx->occ_cache_on || ctx->small_rt_scope;
  if (!keep) return MPSE_OK;     // (outside a solve nothing would own the flags and the mask until the consumer has run)
  mpse_ctx::F0Cache& fc = ctx->f0;
  char* base = nullptr;
  const bool hit = fc.buf && fc.L == h->L && fc.R == h->R && fc.W == h->W0 && fc.cmask == (const void*)FC && fc.Dl == Dl &&
                   fc.Dr == Dr && fc.w == wr && fc.nsite == h->nsite;
  const size_t o_fl = rt_bytes, o_fr = o_fl + fl_bytes, o_tm = o_fr + fr_bytes, o_nt = o_tm + tm_bytes, o_mk = o_nt + nt_bytes;
  const size_t o_od = o_mk + mk_bytes, tot_bytes = o_od + od_bytes;
  // MPSE_F0_ORDER=0: launch positions in plain unit order (as before round 6)
  static const bool order_on = [] {
    const char* e = getenv("MPSE_F0_ORDER");
    return !(e && e[0] == '0');
  }();
  if (hit) {
    base = static_cast<char*>(fc.buf);
  } else {
    void* p = nullptr;
    heff0_drop_cache(ctx);
    MPSE_TRY(mpse_malloc(ctx, tot_bytes, &p));
    fc.buf = p, fc.L = h->L, fc.R = h->R, fc.W = h->W0, fc.cmask = FC, fc.Dl = Dl, fc.Dr = Dr, fc.w = wr, fc.nsite = h->nsite;
    base = static_cast<char*>(p);
    {   // terms | term counts: adjacent in the buffer, one upload
      std::vector<char> up(tm_bytes + nt_bytes, 0);
      memcpy(up.data(), terms.data(), tm_bytes);
      memcpy(up.data() + tm_bytes, nterm.data(), size_t(wr) * sizeof(int));
      MPSE_TRY(stage_h2d(ctx, base + o_tm, up.data(), up.size()));
    }
    const long long nel = (long long)wr * Dr * Dr;
    int nb = (int)((nel + 255) / 256);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_f0_prepare, dim3(ntl * wl + ntr * wr + nb), dim3(256), 0, ctx->stream,
                       static_cast<const double2*>(h->L), static_cast<const double2*>(h->R), reinterpret_cast<double2*>(base),
                       reinterpret_cast<unsigned char*>(base + o_fl), reinterpret_cast<unsigned char*>(base + o_fr), Dl, Dr, wl,
                       wr, ntl, ntr, ctx->skip_flag);
    if (order_on)
      hipLaunchKernelGGL(k_f0_plan, dim3(1), dim3(1024), 0, ctx->stream, reinterpret_cast<const unsigned char*>(base + o_fl), FC,
                         reinterpret_cast<const unsigned char*>(base + o_fr), reinterpret_cast<const F0Term*>(base + o_tm),
                         reinterpret_cast<const int*>(base + o_nt), wl, wr, d, ntl, ntr, nkc, fc_pitch,
                         reinterpret_cast<unsigned long long*>(base + o_mk), ctx->n_cu > 0 ? ctx->n_cu : 256,
                         reinterpret_cast<F0Info*>(base + o_od), compact ? 1 : 0, ctx->skip_flag);
    else
      hipLaunchKernelGGL(k_f0_valid, dim3(1), dim3(1024), 0, ctx->stream, reinterpret_cast<const unsigned char*>(base + o_fl), FC,
                         reinterpret_cast<const unsigned char*>(base + o_fr), reinterpret_cast<const F0Term*>(base + o_tm),
                         reinterpret_cast<const int*>(base + o_nt), wl, wr, d, ntl, ntr, nkc, fc_pitch,
                         reinterpret_cast<unsigned long long*>(base + o_mk), ctx->skip_flag);
  }
  F0Args g{};
  g.L = static_cast<const double*>(h->L);
  g.Rt = reinterpret_cast<const double*>(base);
  g.C = static_cast<const double*>(C);
  g.parts = static_cast<double*>(pr.ptr);
  g.FL = reinterpret_cast<const unsigned char*>(base + o_fl);
  g.FC = FC;
  g.FR = reinterpret_cast<const unsigned char*>(base + o_fr);
  g.terms = reinterpret_cast<const F0Term*>(base + o_tm);
  g.nterm = reinterpret_cast<const int*>(base + o_nt);
  g.mask = reinterpret_cast<const unsigned long long*>(base + o_mk);
  g.skip = ctx->skip_flag;
  g.info = order_on ? reinterpret_cast<const F0Info*>(base + o_od) : nullptr;
  g.trace = ctx->prof_on ? ctx->gemm_trace : nullptr;
  g.n = n;
  g.Dl = Dl, g.Dr = Dr, g.wl = wl, g.wr = wr, g.d = d, g.ntl = ntl, g.ntr = ntr, g.nkc = nkc, g.fc_pitch = fc_pitch;
  if (ctx->dot_req.y) {
    g.y = static_cast<const double*>(ctx->dot_req.y);
    g.dot_part = ctx->dot_req.part;
    ctx->dot_req.nb_out = nwg * d;
  }
  {
    // sampled HIP-event bracket (variant 7 of mpse_prof_get): algorithmic flops of SURVEY.md 8(d) for this matvec
    // (8 real flops per complex multiply-add; 4 for the real MPO site), bytes = operands read once + parts written
    const double fl = h->nsite == 0 ? 8.0 * wr * double(Dl) * Dr * (double(Dl) + Dr)
                                    : 8.0 * double(Dl) * Dl * wl * d * Dr + 4.0 * double(Dl) * Dr * wl * wr * d * d +
                                          8.0 * double(Dl) * Dr * Dr * wr * d;
    const double by = 16.0 * (double(Dl) * wl * Dl + double(Dr) * wr * Dr + 2.0 * double(n));
    ProfScope fprof(ctx, 7, fl, by);
    hipLaunchKernelGGL(k_heff0_fused, dim3(nwg * d), dim3(256), 0, ctx->stream, g);
    fprof.end();
  }
  MPSE_HIP(ctx, hipGetLastError());
  ++ctx->f0_launches[h->nsite == 1 ? 1 : 0];
  pr.used = nparts;
  pr.mask = g.mask;
  pr.mask_row = Dr;
  pr.mask_tiles = ntr;
  *taken = true;
  return MPSE_OK;
}

extern "C" int mpse_heff_fused_stats(mpse_ctx* ctx, int64_t* bond_launches, int64_t* site_launches) {
  if (!ctx) return MPSE_ERR_ARG;
  if (bond_launches) *bond_launches = ctx->f0_launches[0];
  if (site_launches) *site_launches = ctx->f0_launches[1];
  return MPSE_OK;
}
