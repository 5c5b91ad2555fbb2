// Contraction plans for the hot path: each environment update / effective-Hamiltonian
// matvec is a short list of strided-GEMM steps (see mpse_gemm.hip) over named buffers.
// Pure host C++ with no HIP dependency, so the index algebra can be exercised on a CPU
// (tests/host_emu) with a naive loop executor; the product executes the same plans
// with the FP64-MFMA kernel (mpse_contract.hip).
//
// Contraction order is (env . site) . W . (other side) - the order opt_einsum's
// optimal path picks for D >> w, d (mps/hop_expr.py via mps/oe_contract_wrap.py:24-35) -
// i.e. two large GEMMs around a small batched contraction with the MPO site tensor.
// All index permutations are absorbed in operand strides; nothing is transposed in memory.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mpsengine.h"

namespace mpse_plan {

enum Buf { B_L = 0, B_R, B_W0, B_W1, B_C, B_BRA, B_OUT, B_T1, B_T2, B_T3, B_W2, B_W3, B_OUT2, B_COUNT };
inline int w_buf(int layer) { return layer == 0 ? B_W0 : layer == 1 ? B_W1 : layer == 2 ? B_W2 : B_W3; }

enum Kind { K_GEMM = 0, K_COPY = 1, K_WMIX = 2, K_GGEMM = 3 };


// ---------------------------------------------------------------------------------------------------------------
// Folded one-site matvec: the MPO step shrinks to what the block structure of the MPO site needs, so most of the
// intermediates T1 = L . C and T2 = W T1 (Dl wl d Dr elements each) and the launches that only move them (MPO step as a
// batched product, unit-channel copy) disappear.
//   out[a,dd,l] = sum_f P_f[a,dd,k] R[l,f,k],   P_f = sum_b W[b,dd,:,f] (L[:,b,:] . C)
// An MPO site is a sparse matrix of d x d blocks W[b,:,:,f]; for sum-of-products Hamiltonians most non-zero blocks are
// the identity (channels that pass through the site).
//   * a channel b whose only block is an identity into a channel f that receives nothing else: the product
//     L[:,b,:] . C IS the plane P_f - written there directly;
//   * the other channels b write temporary planes T_b = L[:,b,:] . C, and one elementwise pass (K_WMIX) forms every
//     remaining plane P_f = sum_b W[b,:,:,f] T_b, with the centre tensor itself in the place of T_b for the unit channel
//     of L (d x d blocks out of LDS; an identity block is a plain add);
//   * a plane that is only the unit channel's identity block is the centre tensor itself (no copy), and the plane of
//     R's unit channel is `out`;
//   * all products L[:,b,:] . C of a site are ONE grouped launch (K_GGEMM: a group = the tile rows of one channel), and
//     out (+)= sum_{f != ru} P_f . R[:,f,:]^T is one more: one group whose K loop runs over the channels (segments).
// Every channel keeps its own product, so the quantum-number zero blocks of L[:,b,:] stay whole empty tiles.
constexpr int G_MAXGRP = 8, G_MAXSEG = 4, WM_MAXDST = 4, WM_MAXTERM = 4;

struct WBlock {
  int b, f;
  bool ident;   // W[b, :, :, f] is the d x d identity
};
struct WSiteInfo {
  int64_t wl = 0, d = 0, wr = 0;
  std::vector<WBlock> blocks;   // the non-zero blocks, ordered by (f, b)
  std::vector<double> w;        // the site itself (wl, d, d, wr), row major
};
inline WSiteInfo analyse_mpo_site(const double* w, int64_t wl, int64_t d, int64_t wr) {
  WSiteInfo r;
  r.wl = wl, r.d = d, r.wr = wr;
  r.w.assign(w, w + wl * d * d * wr);
  for (int64_t f = 0; f < wr; ++f)
    for (int64_t b = 0; b < wl; ++b) {
      bool any = false, ident = true;
      for (int64_t x = 0; x < d; ++x)
        for (int64_t e = 0; e < d; ++e) {
          const double v = w[((b * d + x) * d + e) * wr + f];
          if (v != 0.0) any = true;
          if (v != (x == e ? 1.0 : 0.0)) ident = false;
        }
      if (any) r.blocks.push_back(WBlock{(int)b, (int)f, ident});
    }
  return r;
}

// K_WMIX: dst[a, dd, k] = sum_terms sum_e W[b, dd, e, f] src[a, e, k]; element (a, x, k) of a tensor at off + a s_a + x s_d + k
constexpr int WM_CHUNK = 4, WM_MAXCHUNKS = 8;     // a thread forms WM_CHUNK consecutive dd; d <= 32
struct WMixTerm {
  int b = 0, f = 0;
  bool ident = false;
  int src = 0;
  int64_t src_off = 0, s_a = 0, s_d = 0;
  unsigned char e_lo[WM_MAXCHUNKS] = {0}, e_hi[WM_MAXCHUNKS] = {0};   // columns e the rows of a chunk need: [e_lo, e_hi)
};
struct WMixDst {
  int dst = 0;
  int64_t dst_off = 0, s_a = 0, s_d = 0;
  int nterm = 0;
  WMixTerm term[WM_MAXTERM];
};
enum BMaskKind { BM_NONE = 0, BM_SCAN = 1, BM_CENTRE = 2 };
struct GSegPlan {
  int abuf = 0;
  int64_t a_off = 0;
  int bbuf = 0;
  int64_t b_off = 0;
  int64_t am_row0 = -1;   // >= 0: occupancy of A from the step's scan_a: first tile row of this segment there
  int bm_kind = BM_NONE;  // BM_SCAN: from scan_b, K tiles from bm_kt0 on; BM_CENTRE: B is the centre tensor (structural
  int64_t bm_kt0 = 0;     // mask of the solve)
};
struct GGroupPlan {
  int nseg = 0;
  GSegPlan seg[G_MAXSEG];
  int cbuf = 0;
  int64_t c_off = 0;
  double beta = 0.0;
  // split2 (single-group steps): two workgroups per output tile, each over half of the tile's occupied K tiles; the
  // first half (+ beta C) is stored to C, the second to B_OUT2, laid out like C: the caller of the plan adds the two
  bool split2 = false;
};
struct ScanDesc {   // an operand whose tile occupancy is scanned once for all segments that read parts of it
  bool on = false;
  int buf = 0, dt = MPSE_F64;
  int64_t off = 0;
  mpse_index r{}, k{};
};

struct Step {
  int a, b, c;                 // buffer ids
  int64_t a_off, b_off, c_off; // element offsets into the buffers
  int dta, dtb;
  int conja, conjb;
  mpse_index ma, ka, kb, nb, mc, nc;
  int64_t batch, sba, sbb, sbc;
  int kind = K_GEMM;           // K_COPY: C(i,j) = A(i,j) with A indexed by (ma, ka), C by (mc, nc); b unused
  double beta = 0.0;           // K_GEMM: C = A.B + beta C
  int skip_zero = 0;           // K_GEMM: scan both operands for all-zero tiles and skip them (block-sparse sweeps)
  // K_GEMM steps written by push_w also carry the sizes of the MPO step they are (Da = batch of bond states, N =
  // trailing block): the executor runs small ones (d = 2 sites) through an elementwise kernel instead of MFMA tiles
  bool is_wstep = false;
  int64_t w_Da = 0, w_wl = 0, w_d = 0, w_wr = 0, w_N = 0;
  // K_WMIX / K_GGEMM (folded one-site matvec, above)
  std::vector<WMixDst> mix;    // K_WMIX: b = MPO site; tensors (wp_Da, wp_d, wp_Dk)
  int64_t wp_Da = 0, wp_d = 0, wp_Dk = 0, wp_wr = 0;
  std::vector<GGroupPlan> groups;   // K_GGEMM: ma .. nc describe ONE segment's product (rows of one group, K of one segment)
  ScanDesc scan_a, scan_b;
  int cin = -1;                // K_GEMM with beta != 0: buffer the beta term is read from (-1: C itself) ...
  int64_t cin_off = 0;         // ... at this element offset, through these index maps
  mpse_index mcin{}, ncin{};
};

struct Plan {
  std::vector<Step> steps;
  int64_t tmp_elems[3] = {0, 0, 0};  // T1, T2, T3 sizes (elements of the working dtype)
  bool two_results = false;          // the result is B_OUT + B_OUT2 (GGroupPlan::split2)
  const char* error = nullptr;
};

inline mpse_index i1(int64_t ext, int64_t stride) { return mpse_index{ext, ext > 0 ? ext : 1, 0, stride}; }
inline mpse_index i2(int64_t hi, int64_t lo, int64_t s_hi, int64_t s_lo) {
  return mpse_index{hi * lo, lo > 0 ? lo : 1, s_hi, s_lo};
}

inline void push(Plan& p, int a, int64_t ao, int dta, int conja, int b, int64_t bo, int dtb, int conjb, int c,
                 int64_t co, mpse_index ma, mpse_index ka, mpse_index kb, mpse_index nb, mpse_index mc,
                 mpse_index nc, int64_t batch = 1, int64_t sba = 0, int64_t sbb = 0, int64_t sbc = 0) {
  p.steps.push_back(Step{a, b, c, ao, bo, co, dta, dtb, conja, conjb, ma, ka, kb, nb, mc, nc, batch, sba, sbb, sbc});
}

// Skipping a unit channel trades a (m x k x n) slice of a GEMM for a strided copy kernel: below ~1e8 multiply-adds
// the slice costs less than the extra launch, so small centres keep the plain GEMM.
inline int64_t& unit_threshold() {   // multiply-adds; a test hook lowers it so that small cases take the copy path
  static int64_t v = int64_t(1) << 27;
  return v;
}
inline bool unit_pays(int64_t m, int64_t k, int64_t n) { return m * k * n >= unit_threshold(); }
inline bool& beta_source_flag() {
  static bool v = true;     // (test hook: the emulator also runs the plans with the copy instead)
  return v;
}
inline bool beta_source_on() { return beta_source_flag(); }
// Scanning the operands of a GEMM for structurally zero tiles costs two small launches and one pass over the
// operands; worth it from ~3e7 multiply-adds on.
// The result says which operands to scan: bit 0 = A, bit 1 = B.
inline int skip_pays(int64_t m, int64_t k, int64_t n, int which = 3) {
  return m * k * n >= (int64_t(1) << 28) ? which : 0;
}

inline void push_copy(Plan& p, int src, int64_t so, int dst, int64_t dof, int dtype, mpse_index mi, mpse_index ni,
                      mpse_index mo, mpse_index no) {
  Step s{src, src, dst, so, 0, dof, dtype, dtype, 0, 0, mi, ni, ni, no, mo, no, 1, 0, 0, 0};
  s.kind = K_COPY;
  p.steps.push_back(s);
}

// X[a,b,n] = sum_c E[a,b,c] K[c,n] for an environment E (rows, w, cols) and a matrix K (cols, N).
// `unit` (1-based, 0 = none) names an MPO-bond channel b along which E[:, b, :] is the identity matrix (what a
// canonical MPS gives for the channel in which no operator has acted yet): that slice of X is a copy of K and
// the GEMM runs over the remaining channels only.
// Layout of X: (rows, w, N) when b_outer == false; (w, rows, N) when b_outer == true.  With the channel as the
// outer index every 64-row tile of the GEMM belongs to one MPO channel and one run of bond states, so the
// quantum-number zero blocks of E show up as whole empty tiles (structural-zero skipping, mpse_gemm.hip).
inline void push_env_times(Plan& p, int ebuf, int e_dtype, int kbuf, int k_dtype, int xbuf, int64_t rows, int64_t w,
                           int64_t cols, int64_t N, int64_t unit, bool b_outer) {
  const int64_t u = (unit >= 1 && unit <= w && rows == cols && unit_pays(rows, cols, N)) ? unit - 1 : -1;
  if (u >= 0) {
    if (b_outer)
      push_copy(p, kbuf, 0, xbuf, u * rows * N, k_dtype, i1(rows, N), i1(N, 1), i1(rows, N), i1(N, 1));
    else
      push_copy(p, kbuf, 0, xbuf, u * N, k_dtype, i1(rows, N), i1(N, 1), i1(rows, w * N), i1(N, 1));
  }
  const int64_t lo[2] = {0, u + 1}, hi[2] = {u >= 0 ? u : w, u >= 0 ? w : 0};
  for (int r = 0; r < 2; ++r) {
    const int64_t b0 = lo[r], nb = hi[r] - lo[r];
    if (nb <= 0) continue;
    if (b_outer)   // GEMM row i = (b - b0) * rows + a
      push(p, ebuf, b0 * cols, e_dtype, 0, kbuf, 0, k_dtype, 0, xbuf, b0 * rows * N, i2(nb, rows, cols, w * cols),
           i1(cols, 1), i1(cols, N), i1(N, 1), i1(nb * rows, N), i1(N, 1));
    else           // GEMM row i = a * nb + (b - b0)
      push(p, ebuf, b0 * cols, e_dtype, 0, kbuf, 0, k_dtype, 0, xbuf, b0 * N, i2(rows, nb, w * cols, cols),
           i1(cols, 1), i1(cols, N), i1(N, 1), i2(rows, nb, w * N, N), i1(N, 1));
    p.steps.back().skip_zero = skip_pays(rows * nb, cols, N);
  }
}

// out[(m,g),l] = sum_{f,k} T[m,f,g,k] E[l,f,k] for T (M1, w, anc, Dk) and an environment E (Dout, w, Dk).
// `unit` as above: E[:, f, :] = identity contributes T[m,f,g,l] itself (needs Dout == Dk).
inline void push_times_env(Plan& p, int tbuf, int t_dtype, int ebuf, int e_dtype, int obuf, int64_t M1, int64_t anc,
                           int64_t w, int64_t Dk, int64_t Dout, int64_t unit) {
  const int64_t u = (unit >= 1 && unit <= w && Dout == Dk && unit_pays(M1 * anc, Dk, Dout)) ? unit - 1 : -1;
  double beta = 0.0;
  // The unit channel contributes T[m, u, g, l] itself: the first product reads it as its beta term straight from T
  // (no copy into `out` first), unless there is no product at all (w == 1) or MPSE_BETA_SOURCE=0.
  const bool direct = u >= 0 && w > 1 && beta_source_on();
  if (u >= 0 && !direct) {
    push_copy(p, tbuf, u * anc * Dk, obuf, 0, t_dtype, i2(M1, anc, w * anc * Dk, Dk), i1(Dk, 1), i1(M1 * anc, Dout),
              i1(Dout, 1));
    beta = 1.0;
  }
  bool first = true;
  const int64_t lo[2] = {0, u + 1}, hi[2] = {u >= 0 ? u : w, u >= 0 ? w : 0};
  for (int r = 0; r < 2; ++r) {
    const int64_t f0 = lo[r], nf = hi[r] - lo[r];
    if (nf <= 0) continue;
    push(p, tbuf, f0 * anc * Dk, t_dtype, 0, ebuf, f0 * Dk, e_dtype, 0, obuf, 0,
         /*A: m=(m1 | g)*/ i2(M1, anc, w * anc * Dk, Dk), /*k=(f | k)*/ i2(nf, Dk, anc * Dk, 1),
         /*B=E: k=(f,k), n=l*/ i1(nf * Dk, 1), i1(Dout, w * Dk), i1(M1 * anc, Dout), i1(Dout, 1));
    if (direct && first) {
      Step& s = p.steps.back();
      s.cin = tbuf;
      s.cin_off = u * anc * Dk;
      s.mcin = i2(M1, anc, w * anc * Dk, Dk);
      s.ncin = i1(Dk, 1);
      beta = 1.0;
    }
    first = false;
    p.steps.back().beta = beta;
    p.steps.back().skip_zero = skip_pays(M1 * anc, nf * Dk, Dout, 2);   // A = the big intermediate: environment side only
    beta = 1.0;
  }
}

// T_out[a, d, f, n] = sum_{b,e} W[b,d,e,f] T_in[b, a, e, n]   (W (wl,d,d,wr) row-major; batch over a;
// T_in in the channel-outer layout written by push_env_times(b_outer = true))
inline void push_w(Plan& p, int wbuf, int w_dtype, int tin, int tout, int t_dtype, int64_t na, int64_t wl, int64_t d,
                   int64_t wr, int64_t N) {
  push(p, wbuf, 0, w_dtype, 0, tin, 0, t_dtype, 0, tout, 0,
       /*A=W: i=(d | f), k=(b | e)*/ i2(d, wr, d * wr, 1), i2(wl, d, d * d * wr, wr),
       /*B=T_in[:, a]: k=(b | e), n*/ i2(wl, d, na * d * N, N), i1(N, 1),
       /*C=T_out[a]: (d,f), n*/ i1(d * wr, N), i1(N, 1), na, 0, d * N, d * wr * N);
  Step& s = p.steps.back();
  s.is_wstep = true;
  s.w_Da = na, s.w_wl = wl, s.w_d = d, s.w_wr = wr, s.w_N = N;
}

// Effective Hamiltonian matvec, mps/hop_expr.py:57-115.  The bra-side bonds (rows of L / R, bonds of `out`) may
// differ from the ket-side bonds (columns of L / R, bonds of C): that is the projection of H C onto another
// state's bond spaces used by the variational compression (mps/mp.py:513-650); the Krylov / Davidson drivers
// require them equal.
inline Plan plan_heff1_fold(int dtype, const mpse_heff& h, const WSiteInfo& wi, bool two_results);

// `wi`: block structure of the MPO site where the caller knows it (mpse_mpo_site_hint): large one-site centres then
// take the folded plan.  `two_results`: the caller accepts the result as the sum of B_OUT and B_OUT2 (Plan::two_results
// says whether the plan made use of it).
inline Plan plan_heff(int dtype, const mpse_heff& h, const WSiteInfo* wi = nullptr, bool two_results = false) {
  Plan p;
  const mpse_dims& s = h.dims;
  const int64_t Dl = s.Dl_ket, Dr = s.Dr_ket, wl = s.wl, wr = s.wr;
  const int64_t Dlb = s.Dl_bra > 0 ? s.Dl_bra : Dl, Drb = s.Dr_bra > 0 ? s.Dr_bra : Dr;
  const int64_t anc = s.danc > 0 ? s.danc : 1;
  if (h.nsite == 0) {
    // abc,lbk,ck->al (hop_expr.py:63-67): T[a,b,k] = L[(a,b),c] S[c,k] ; out[a,l] = T[a,(b,k)] R[l,(b,k)]
    if (wl != wr) {
      p.error = "heff(0-site): wl != wr";
      return p;
    }
    p.tmp_elems[0] = Dlb * wl * Dr;
    push_env_times(p, B_L, h.l_dtype, B_C, dtype, B_T1, Dlb, wl, Dl, Dr, h.l_unit, false);
    push_times_env(p, B_T1, dtype, B_R, h.r_dtype, B_OUT, Dlb, 1, wr, Dr, Drb, h.r_unit);
    return p;
  }
  if (h.nsite == 1) {
    if (wi) {
      Plan f = plan_heff1_fold(dtype, h, *wi, two_results);
      if (!f.error) return f;
    }
    // abc,bdef,lfk,cek->adl (hop_expr.py:75-79); ancilla cegk->adgl (87-91)
    const int64_t d = s.d0, N = d * anc * Dr, Nb = anc * Dr;
    p.tmp_elems[0] = Dlb * wl * N;
    p.tmp_elems[1] = Dlb * d * wr * Nb;
    // T1[a,b,(e,g,k)] = sum_c L[(a,b),c] C[c,(e,g,k)]
    push_env_times(p, B_L, h.l_dtype, B_C, dtype, B_T1, Dlb, wl, Dl, N, h.l_unit, true);
    // T2[a,d,f,(g,k)] = sum_{b,e} W[b,d,e,f] T1[a,b,e,(g,k)]
    push_w(p, B_W0, h.w_dtype, B_T1, B_T2, dtype, Dlb, wl, d, wr, Nb);
    // out[(a,d,g),l] = sum_{f,k} T2[a,d,f,g,k] R[l,f,k]
    push_times_env(p, B_T2, dtype, B_R, h.r_dtype, B_OUT, Dlb * d, anc, wr, Dr, Drb, h.r_unit);
    return p;
  }
  if (h.nsite == 2) {
    // abc,bdef,fghj,ljk,cehk->adgl (hop_expr.py:99-103); ancilla cemhnk->admgnl (111-115): the two ancilla legs m, n
    // have the sizes of their own sites (a0, a1)
    const int64_t d0 = s.d0, d1 = s.d1, wm = s.wm;
    const int64_t a0 = anc, a1 = s.danc1 > 0 ? s.danc1 : anc;
    const int64_t n2 = a1 * Dr;             // (n,k): trailing block after h
    const int64_t n1 = a0 * d1 * n2;        // (m,h,n,k): trailing block after e
    const int64_t N = d0 * n1;
    p.tmp_elems[0] = Dlb * wl * N;
    p.tmp_elems[1] = Dlb * d0 * wm * n1;
    p.tmp_elems[2] = Dlb * d0 * a0 * d1 * wr * n2;
    // T1[a,b,(e,m,h,n,k)]
    push_env_times(p, B_L, h.l_dtype, B_C, dtype, B_T1, Dlb, wl, Dl, N, h.l_unit, true);
    // T2[a,d,f,(m,h,n,k)]
    push_w(p, B_W0, h.w_dtype, B_T1, B_T2, dtype, Dlb, wl, d0, wm, n1);
    // T3[(a,d),m,g,j,(n,k)] = sum_{f,h} W1[f,g,h,j] T2[(a,d),f,m,h,(n,k)]   batch (a,d); one step per m
    for (int64_t m = 0; m < a0; ++m)
      push(p, B_W1, 0, h.w_dtype, 0, B_T2, m * d1 * n2, dtype, 0, B_T3, m * d1 * wr * n2,
           i2(d1, wr, d1 * wr, 1), i2(wm, d1, d1 * d1 * wr, wr),
           /*B: k=(f | h)*/ i2(wm, d1, a0 * d1 * n2, n2), i1(n2, 1),
           /*C: (g,j),(n,k)*/ i1(d1 * wr, n2), i1(n2, 1), Dlb * d0, 0, wm * a0 * d1 * n2, a0 * d1 * wr * n2);
    // out[(a,d,m,g,n),l] = sum_{j,k} T3[a,d,m,g,j,n,k] R[l,j,k]
    push_times_env(p, B_T3, dtype, B_R, h.r_dtype, B_OUT, Dlb * d0 * a0 * d1, a1, wr, Dr, Drb, h.r_unit);
    return p;
  }
  p.error = "heff: nsite must be 0, 1 or 2";
  return p;
}


// smallest Dl * d * Dr * Dl (multiply-adds of one channel's product) from which the folded plan is used
inline int64_t& fold_min() {
  static int64_t v = [] {
    const char* e = getenv("MPSE_WFOLD");      // MPSE_WFOLD=0: the three-step chain everywhere
    return (e && e[0] == '0') ? (int64_t(1) << 62) : (int64_t(1) << 28);
  }();
  return v;
}

inline int64_t& fold_cus() {    // compute units of the device (mpse_ctx_create sets it): products with at most one
  static int64_t v = 256;       // 64 x 64 tile per unit are halved along K (GGroupPlan::split2)
  return v;
}
inline bool& fold_split2() {
  static bool v = [] {
    const char* e = getenv("MPSE_SPLIT2");     // MPSE_SPLIT2=0: one workgroup per tile
    return !(e && e[0] == '0');
  }();
  return v;
}
inline int64_t& fold_split2_min_kt() {   // fewest K tiles of a product worth halving (test hook)
  static int64_t v = 8;
  return v;
}
inline int64_t& fold_align() {   // bra-bond multiple the device needs (a 64-row tile = one channel); a test hook lowers it
  static int64_t v = 64;
  return v;
}

// Folded one-site matvec (see above); p.error is set when the site does not qualify and the caller takes plan_heff.
inline Plan plan_heff1_fold(int dtype, const mpse_heff& h, const WSiteInfo& wi, bool two_results = false) {
  Plan p;
  const mpse_dims& s = h.dims;
  const int64_t Dl = s.Dl_ket, Dr = s.Dr_ket, wl = s.wl, wr = s.wr, d = s.d0;
  const int64_t Dlb = s.Dl_bra > 0 ? s.Dl_bra : Dl, Drb = s.Dr_bra > 0 ? s.Dr_bra : Dr;
  if (h.nsite != 1 || s.danc > 1 || h.w_dtype != MPSE_F64 || wi.wl != wl || wi.d != d || wi.wr != wr ||
      ((dtype != MPSE_C128 || h.r_dtype != MPSE_C128) && fold_align() == 64)) {   // (the grouped kernel is built for a
                                                    // complex second operand: centre and R; host emulation: any)
    p.error = "fold: not a plain complex one-site centre with a real MPO site";
    return p;
  }
  if (Dlb % fold_align() != 0 || Dl % std::min<int64_t>(16, fold_align()) != 0 ||
      Dr % std::min<int64_t>(16, fold_align()) != 0 || d > WM_CHUNK * WM_MAXCHUNKS || Dl * d * Dr * Dlb < fold_min()) {
    p.error = "fold: centre too small or not tile aligned";
    return p;
  }
  const int64_t lu = (h.l_unit >= 1 && h.l_unit <= wl && Dlb == Dl) ? h.l_unit - 1 : -1;
  const int64_t ru = (h.r_unit >= 1 && h.r_unit <= wr && Drb == Dr) ? h.r_unit - 1 : -1;
  const int64_t N = d * Dr, plane = Dlb * N;
  std::vector<std::vector<const WBlock*>> by_f(wr), by_b(wl);
  for (const WBlock& k : wi.blocks) by_f[k.f].push_back(&k), by_b[k.b].push_back(&k);
  // where the product of channel b goes: straight into the plane of its channel f, or into a temporary
  std::vector<int64_t> direct_f(wl, -1), tmp_of(wl, -1);
  int64_t ntmp = 0, ngemm = 0;
  for (int64_t b = 0; b < wl; ++b) {
    if (b == lu || by_b[b].empty()) continue;
    ++ngemm;
    if (by_b[b].size() == 1 && by_b[b][0]->ident && by_f[by_b[b][0]->f].size() == 1)
      direct_f[b] = by_b[b][0]->f;
    else
      tmp_of[b] = ntmp++;
  }
  if (ngemm > G_MAXGRP) {
    p.error = "fold: too many channels";
    return p;
  }
  // planes: storage for every channel f != ru that is not the centre tensor itself
  std::vector<int64_t> plane_of(wr, -1);
  std::vector<char> alias(wr, 0), present(wr, 0);
  int64_t nplanes = 0;
  for (int64_t f = 0; f < wr; ++f) {
    if (by_f[f].empty()) continue;
    present[f] = 1;
    if (f == ru) continue;
    if (by_f[f].size() == 1 && by_f[f][0]->b == lu && by_f[f][0]->ident)
      alias[f] = 1;
    else
      plane_of[f] = nplanes++;
  }
  // The products with R as halved tiles (split2) when the caller takes the result in two parts, they have at most one
  // tile per compute unit and fit one launch.
  int64_t ncseg = 0;
  for (int64_t f = 0; f < wr; ++f)
    if (present[f] && f != ru) ++ncseg;
  const int64_t ctiles = ((Dlb * d + 63) / 64) * ((Drb + 63) / 64);
  const bool split2 = two_results && fold_split2() && ncseg >= 1 && ncseg <= G_MAXSEG && ctiles <= fold_cus() && ncseg * ((Dr + 15) / 16) >= fold_split2_min_kt() &&
                      (Dlb * d) % 64 == 0 && Drb == Dr && dtype == MPSE_C128;
  auto dest = [&](int64_t f, int* buf, int64_t* off) {
    if (f == ru)
      *buf = B_OUT, *off = 0;
    else
      *buf = B_T2, *off = plane_of[f] * plane;
  };
  // A: all products L[:, b, :] . C in one grouped launch
  Step ga{};
  ga.kind = K_GGEMM;
  ga.dta = h.l_dtype, ga.dtb = dtype;
  ga.ma = i1(Dlb, wl * Dl), ga.ka = i1(Dl, 1), ga.kb = i1(Dl, N), ga.nb = i1(N, 1), ga.mc = i1(Dlb, N), ga.nc = i1(N, 1);
  ga.batch = 1;
  ga.scan_a.on = true;        // L as (channel | bra bond) rows: every 64-row tile belongs to one channel
  ga.scan_a.buf = B_L, ga.scan_a.dt = h.l_dtype, ga.scan_a.off = 0;
  ga.scan_a.r = i2(wl, Dlb, Dl, wl * Dl), ga.scan_a.k = i1(Dl, 1);
  for (int64_t b = 0; b < wl; ++b) {
    if (direct_f[b] < 0 && tmp_of[b] < 0) continue;
    GGroupPlan g;
    g.nseg = 1;
    g.seg[0].abuf = B_L, g.seg[0].a_off = b * Dl, g.seg[0].am_row0 = b * (Dlb / 64);
    g.seg[0].bbuf = B_C, g.seg[0].b_off = 0, g.seg[0].bm_kind = BM_CENTRE;
    if (direct_f[b] >= 0)
      dest(direct_f[b], &g.cbuf, &g.c_off);
    else
      g.cbuf = B_T1, g.c_off = tmp_of[b] * plane;
    ga.groups.push_back(g);
  }
  // W: the planes that are sums of blocks
  Step mix{};
  mix.kind = K_WMIX;
  mix.b = B_W0;
  mix.dta = dtype, mix.dtb = h.w_dtype;
  mix.wp_Da = Dlb, mix.wp_d = d, mix.wp_Dk = Dr, mix.wp_wr = wr;
  const int64_t nchunk = (d + WM_CHUNK - 1) / WM_CHUNK;
  for (int64_t f = 0; f < wr; ++f) {
    if (!present[f] || alias[f]) continue;
    bool is_direct = false;
    for (const WBlock* k : by_f[f])
      if (k->b != lu && direct_f[k->b] == f) is_direct = true;
    if (is_direct) continue;
    if ((int)mix.mix.size() >= WM_MAXDST || (int)by_f[f].size() > WM_MAXTERM) {
      p.error = "fold: too many blocks per channel";
      return p;
    }
    WMixDst q;
    dest(f, &q.dst, &q.dst_off);
    q.s_a = N, q.s_d = Dr;
    for (const WBlock* k : by_f[f]) {
      WMixTerm& t = q.term[q.nterm++];
      t.b = k->b, t.f = k->f, t.ident = k->ident;
      if (k->b == lu)
        t.src = B_C, t.src_off = 0;
      else
        t.src = B_T1, t.src_off = tmp_of[k->b] * plane;
      t.s_a = N, t.s_d = Dr;
      for (int64_t c = 0; c < nchunk; ++c) {
        int64_t lo = d, hi = 0;
        for (int64_t x = c * WM_CHUNK; x < std::min<int64_t>(d, (c + 1) * WM_CHUNK); ++x)
          for (int64_t e = 0; e < d; ++e)
            if (wi.w[((k->b * d + x) * d + e) * wr + k->f] != 0.0) lo = std::min(lo, e), hi = std::max(hi, e + 1);
        t.e_lo[c] = (unsigned char)(hi > lo ? lo : 0), t.e_hi[c] = (unsigned char)(hi > lo ? hi : 0);
      }
    }
    mix.mix.push_back(q);
  }
  {
    int64_t nslot = 0;
    for (const WMixDst& q : mix.mix)
      for (int t = 0; t < q.nterm; ++t) nslot += q.term[t].ident ? 0 : 1;
    if (nslot * d * d > 8192) {          // 64 KB of LDS for the dense blocks
      p.error = "fold: blocks too large for the elementwise pass";
      return p;
    }
  }
  // C: out (+)= sum_{f != ru} P_f . R[:, f, :]^T, one group, at most G_MAXSEG channels per launch
  Step gc{};
  gc.kind = K_GGEMM;
  gc.dta = dtype, gc.dtb = h.r_dtype;
  gc.ma = i1(Dlb * d, Dr), gc.ka = i1(Dr, 1), gc.kb = i1(Dr, 1), gc.nb = i1(Drb, wr * Dr), gc.mc = i1(Dlb * d, Drb),
  gc.nc = i1(Drb, 1);
  gc.batch = 1;
  gc.scan_b.on = true;        // R as (bra bond) x (channel | ket bond)
  gc.scan_b.buf = B_R, gc.scan_b.dt = h.r_dtype, gc.scan_b.off = 0;
  gc.scan_b.r = i1(Drb, wr * Dr), gc.scan_b.k = i1(wr * Dr, 1);
  std::vector<GSegPlan> csegs;
  for (int64_t f = 0; f < wr; ++f) {
    if (!present[f] || f == ru) continue;
    GSegPlan sg;
    sg.abuf = alias[f] ? B_C : B_T2, sg.a_off = alias[f] ? 0 : plane_of[f] * plane;
    sg.bbuf = B_R, sg.b_off = f * Dr;
    sg.bm_kind = BM_SCAN, sg.bm_kt0 = f * (Dr / 16);
    csegs.push_back(sg);
  }
  bool out_init = ru >= 0 && present[ru];
  if (csegs.empty() && !out_init) {
    p.error = "fold: the MPO site is zero";
    return p;
  }
  p.tmp_elems[0] = ntmp * plane;
  p.tmp_elems[1] = nplanes * plane;
  if (!ga.groups.empty()) p.steps.push_back(ga);
  if (!mix.mix.empty()) p.steps.push_back(mix);
  for (size_t i = 0; i < csegs.size(); i += G_MAXSEG) {
    Step st = gc;
    GGroupPlan g;
    g.cbuf = B_OUT, g.c_off = 0;
    g.beta = out_init ? 1.0 : 0.0;
    if (split2) g.split2 = true, p.two_results = true;
    for (size_t j = i; j < csegs.size() && j < i + G_MAXSEG; ++j) g.seg[g.nseg++] = csegs[j];
    st.groups.push_back(g);
    p.steps.push_back(st);
    out_init = true;
  }
  return p;
}

// Environment update, mps/lib.py:169-250.  Buffers: B_L = incoming environment (either
// domain), B_C = ket site, B_BRA = bra site, B_W0 = mpo site, B_OUT = new environment.
// MPO step of an environment update as the elementwise pass of the folded matvec (K_WMIX) instead of a batched
// real x complex product: for a large physical index and a site the caller has described (mpse_mpo_site_hint) the
// product moves ~130 MB to apply a handful of identity / diagonal / tridiagonal blocks.  in_left: the incoming channel is
// the LEFT index of W (domain L); otherwise the right one (domain R).  Tensors: src[a, (channel), e, k] / dst[a, (channel),
// dd, k] through the offsets and strides given per channel.  Returns false (nothing pushed) when the site does not fit the pass.
inline bool push_env_wmix(Plan& p, const WSiteInfo& wi, bool in_left, int dtype, int w_dtype, int tin, int tout, int64_t Da,
                          int64_t Dk, int64_t src_ch_stride, int64_t src_sa, int64_t src_sd, int64_t dst_ch_stride,
                          int64_t dst_sa, int64_t dst_sd) {
  const int64_t d = wi.d, wl = wi.wl, wr = wi.wr, nout = in_left ? wr : wl;
  const int64_t nchunk = (d + WM_CHUNK - 1) / WM_CHUNK;
  if (d > 32 || nchunk > WM_MAXCHUNKS) return false;
  std::vector<std::vector<const WBlock*>> by_out(nout);
  int64_t nslot = 0;
  for (const WBlock& k : wi.blocks) {
    by_out[in_left ? k.f : k.b].push_back(&k);
    nslot += k.ident ? 0 : 1;
  }
  for (auto& v : by_out)
    if ((int)v.size() > WM_MAXTERM) return false;
  if (nslot * d * d > 8192) return false;          // 64 KB of LDS for the dense blocks
  std::vector<Step> steps;
  Step mix{};
  auto fresh = [&] {
    mix = Step{};
    mix.kind = K_WMIX;
    mix.b = B_W0;
    mix.dta = dtype, mix.dtb = w_dtype;
    mix.wp_Da = Da, mix.wp_d = d, mix.wp_Dk = Dk, mix.wp_wr = wr;
  };
  fresh();
  for (int64_t o = 0; o < nout; ++o) {
    WMixDst q;
    q.dst = tout, q.dst_off = o * dst_ch_stride, q.s_a = dst_sa, q.s_d = dst_sd;
    for (const WBlock* k : by_out[o]) {
      WMixTerm& t = q.term[q.nterm++];
      t.b = k->b, t.f = k->f, t.ident = k->ident;
      t.src = tin, t.src_off = (in_left ? k->b : k->f) * src_ch_stride;
      t.s_a = src_sa, t.s_d = src_sd;
      for (int64_t c = 0; c < nchunk; ++c) {
        int64_t lo = d, hi = 0;
        for (int64_t x = c * WM_CHUNK; x < std::min<int64_t>(d, (c + 1) * WM_CHUNK); ++x)
          for (int64_t e = 0; e < d; ++e)
            if (wi.w[((k->b * d + x) * d + e) * wr + k->f] != 0.0) lo = std::min(lo, e), hi = std::max(hi, e + 1);
        t.e_lo[c] = (unsigned char)(hi > lo ? lo : 0), t.e_hi[c] = (unsigned char)(hi > lo ? hi : 0);
      }
    }
    mix.mix.push_back(q);          // (a channel without blocks: nterm = 0, the pass writes zeros)
    if ((int)mix.mix.size() == WM_MAXDST) {
      steps.push_back(mix);
      fresh();
    }
  }
  if (!mix.mix.empty()) steps.push_back(mix);
  for (Step& st : steps) p.steps.push_back(st);
  return true;
}

inline Plan plan_env(int dtype, int domain, const mpse_dims& s, int env_dtype, int w_dtype, int bra_conj,
                     const WSiteInfo* wi = nullptr) {
  Plan p;
  // the described site takes the elementwise MPO step where the batched product is the expensive way (large d, no ancilla)
  const bool wfold = wi && dtype == MPSE_C128 && w_dtype == MPSE_F64 && s.d0 >= 8 && (s.danc <= 1) && wi->wl == s.wl &&
                     wi->d == s.d0 && wi->wr == s.wr;
  const int64_t d = s.d0, wl = s.wl, wr = s.wr;
  const int64_t anc = s.danc > 0 ? s.danc : 1;
  const int64_t Dlb = s.Dl_bra, Dlk = s.Dl_ket, Drb = s.Dr_bra, Drk = s.Dr_ket;
  if (domain == MPSE_DOMAIN_L) {
    // env L[a,b,c] ; ket A[c,e,g,h] ; W[b,d,e,f] ; bra Ab[a,d,g,p] -> out[p,f,h]  (g = traced ancilla)
    const int64_t N = d * anc * Drk;
    p.tmp_elems[0] = Dlb * wl * N;
    p.tmp_elems[1] = Dlb * d * wr * anc * Drk;
    // X[a,b,(e,g,h)] = sum_c L[(a,b),c] A[c,(e,g,h)]
    push_env_times(p, B_L, env_dtype, B_C, dtype, B_T1, Dlb, wl, Dlk, N, s.env_unit, true);
    // Y[a,d,f,(g,h)] = sum_{b,e} W[b,d,e,f] X[a,b,e,(g,h)]
    if (!(wfold && push_env_wmix(p, *wi, true, dtype, w_dtype, B_T1, B_T2, Dlb, Drk, /*src X[b, a, e, h]: channel b*/ Dlb * d * Drk, d * Drk, Drk,
                                 /*dst: channel f*/ Drk, d * wr * Drk, wr * Drk)))
      push_w(p, B_W0, w_dtype, B_T1, B_T2, dtype, Dlb, wl, d, wr, anc * Drk);
    // out[p,(f,h)] = sum_{a,d,g} bra*[(a,d,g),p] Y[a,d,f,g,h]
    //   A = bra^T: m = p (stride 1), k = (a,d,g) (stride Drb)
    //   B = Y: k = ((a,d) | g): s_hi = wr*anc*Drk, s_lo = Drk ; n = (f | h): s_hi = anc*Drk, s_lo = 1
    push(p, B_BRA, 0, dtype, bra_conj, B_T2, 0, dtype, 0, B_OUT, 0, i1(Drb, 1), i1(Dlb * d * anc, Drb),
         i2(Dlb * d, anc, wr * anc * Drk, Drk), i2(wr, Drk, anc * Drk, 1), i1(Drb, wr * Drk), i1(wr * Drk, 1));
    p.steps.back().skip_zero = skip_pays(Drb, Dlb * d * anc, wr * Drk, 1);   // B = the big intermediate: bra side only
    return p;
  }
  if (domain == MPSE_DOMAIN_R) {
    // env R[a,b,c] ; ket A[h,e,g,c] ; W[q,d,e,b] ; bra Ab[p,d,g,a] -> out[p,q,h]
    p.tmp_elems[0] = Dlk * d * anc * wr * Drb;
    p.tmp_elems[1] = Dlk * wl * d * anc * Drb;
    // X[(h,e,g),(b,a)] = sum_c A[(h,e,g),c] R[a,b,c]   (n ordered (b | a) through the strides of R)
    {
      const int64_t M = Dlk * d * anc;
      const int64_t u = (s.env_unit >= 1 && s.env_unit <= wr && Drb == Drk && unit_pays(M, Drk, Drb)) ? s.env_unit - 1 : -1;
      // unit channel: R[a,u,c] = delta(a,c)  =>  X[m,u,a] = A[m,a]
      if (u >= 0) push_copy(p, B_C, 0, B_T1, u * Drb, dtype, i1(M, Drk), i1(Drk, 1), i1(M, wr * Drb), i1(Drb, 1));
      const int64_t lo[2] = {0, u + 1}, hi[2] = {u >= 0 ? u : wr, u >= 0 ? wr : 0};
      for (int r = 0; r < 2; ++r) {
        const int64_t b0 = lo[r], nb = hi[r] - lo[r];
        if (nb <= 0) continue;
        push(p, B_C, 0, dtype, 0, B_L, b0 * Drk, env_dtype, 0, B_T1, b0 * Drb, i1(M, Drk), i1(Drk, 1), i1(Drk, 1),
             i2(nb, Drb, Drk, wr * Drk), i1(M, wr * Drb), i1(nb * Drb, 1));
        p.steps.back().skip_zero = skip_pays(M, Drk, nb * Drb);
      }
    }
    // Y[h,q,d,g,a] = sum_{e,b} W[q,d,e,b] X[h,e,g,b,a] ; batch h ; one step per ancilla value g
    if (wfold && push_env_wmix(p, *wi, false, dtype, w_dtype, B_T1, B_T2, Dlk, Drb, /*src: channel b*/ Drb, d * wr * Drb, wr * Drb,
                               /*dst: channel q*/ d * Drb, wl * d * Drb, Drb)) {
    } else
    for (int64_t g = 0; g < anc; ++g)
      push(p, B_W0, 0, w_dtype, 0, B_T1, g * wr * Drb, dtype, 0, B_T2, g * Drb,
           /*A=W: m=(q,d), k=(e,b)*/ i1(wl * d, d * wr), i1(d * wr, 1),
           /*B: k=(e | b), n=a*/ i2(d, wr, anc * wr * Drb, Drb), i1(Drb, 1),
           /*C: m=(q,d), n=a*/ i1(wl * d, anc * Drb), i1(Drb, 1), Dlk, 0, d * anc * wr * Drb, wl * d * anc * Drb);
    // out[p,(q,h)] = sum_{d,g,a} bra*[p,(d,g,a)] Y[h,q,(d,g,a)]
    push(p, B_BRA, 0, dtype, bra_conj, B_T2, 0, dtype, 0, B_OUT, 0, i1(Dlb, d * anc * Drb), i1(d * anc * Drb, 1),
         i1(d * anc * Drb, 1), i2(wl, Dlk, d * anc * Drb, wl * d * anc * Drb), i1(Dlb, wl * Dlk), i1(wl * Dlk, 1));
    p.steps.back().skip_zero = skip_pays(Dlb, d * anc * Drb, wl * Dlk, 1);
    return p;
  }
  p.error = "env: bad domain";
  return p;
}

// ---------------------------------------------------------------------------------------------------------------
// Labelled views: the plans for stacked MPO layers (mps/lib.py:121-166, mps/hop_expr.py:24-52) are written as
// pairwise contractions of tensors whose dimensions carry one-letter labels; `contract` turns one contraction into
// strided-GEMM steps (labels listed outer -> inner; adjacent labels are merged when their strides are contiguous;
// an index may have at most two levels per operand, everything else goes to `loops`, one step per value).
struct View {
  int buf = 0, dt = MPSE_F64;
  int64_t off = 0;
  std::vector<std::pair<char, int64_t>> dims;   // memory order, row major
  View() {}
  View(int b, int d, std::initializer_list<std::pair<char, int64_t>> l) : buf(b), dt(d), dims(l) {}
  int64_t ext(char c) const {
    for (auto& p : dims)
      if (p.first == c) return p.second;
    return -1;
  }
  int64_t stride(char c) const {   // 0 if the label is absent
    int64_t s = 1;
    for (size_t i = dims.size(); i-- > 0;) {
      if (dims[i].first == c) return s;
      s *= dims[i].second;
    }
    return 0;
  }
  int64_t size() const {
    int64_t s = 1;
    for (auto& p : dims) s *= p.second;
    return s;
  }
};

inline bool labels_index(const View& v, const std::string& labels, mpse_index* out) {
  std::vector<std::pair<int64_t, int64_t>> lv;  // (ext, stride), outer -> inner, merged
  int64_t total = 1;
  for (char c : labels) {
    const int64_t e = v.ext(c), st = v.stride(c);
    if (e < 0) return false;
    total *= e;
    if (e == 1) continue;
    if (!lv.empty() && lv.back().second == e * st)
      lv.back().first *= e, lv.back().second = st;
    else
      lv.push_back({e, st});
  }
  if (lv.empty()) {
    *out = i1(total, 1);
    return true;
  }
  if (lv.size() == 1) {
    *out = i1(lv[0].first, lv[0].second);
    return true;
  }
  if (lv.size() == 2) {
    *out = i2(lv[0].first, lv[1].first, lv[0].second, lv[1].second);
    return true;
  }
  return false;
}

// C[batch][m, n] = sum_k A[batch][m, k] B[batch][k, n] for every value of the `loops` labels
inline void contract(Plan& p, const View& A, const View& B, const View& C, const std::string& m, const std::string& k,
                     const std::string& n, const std::string& batch = "", const std::string& loops = "",
                     int conja = 0) {
  if (p.error) return;
  mpse_index ma, ka, kb, nb, mc, nc, ba, bb, bc;
  if (!labels_index(A, m, &ma) || !labels_index(A, k, &ka) || !labels_index(B, k, &kb) || !labels_index(B, n, &nb) ||
      !labels_index(C, m, &mc) || !labels_index(C, n, &nc)) {
    p.error = "plan: an index of a stacked-MPO contraction needs more than two stride levels";
    return;
  }
  int64_t nbatch = 1, sba = 0, sbb = 0, sbc = 0;
  if (!batch.empty()) {
    // a batch label may be absent from A or B (stride 0) but must be a single level where present
    auto one = [&](const View& v, int64_t* st) {
      std::string present;
      for (char c : batch)
        if (v.ext(c) >= 0) present.push_back(c);
      if (present.empty()) {
        *st = 0;
        return true;
      }
      if (present.size() != batch.size()) return false;
      mpse_index ix;
      if (!labels_index(v, batch, &ix) || (ix.lo_ext < ix.ext && ix.ext > 1)) return false;
      *st = ix.s_lo;
      return true;
    };
    if (!one(A, &sba) || !one(B, &sbb) || !one(C, &sbc)) {
      p.error = "plan: batch labels of a stacked-MPO contraction do not form one level";
      return;
    }
    for (char c : batch) nbatch *= C.ext(c);
  }
  std::vector<int64_t> ext, cnt(loops.size(), 0);
  int64_t nloop = 1;
  for (char c : loops) {
    const int64_t e = C.ext(c) >= 0 ? C.ext(c) : (A.ext(c) >= 0 ? A.ext(c) : B.ext(c));
    ext.push_back(e);
    nloop *= e;
  }
  for (int64_t it = 0; it < nloop; ++it) {
    int64_t oa = A.off, ob = B.off, oc = C.off;
    for (size_t l = 0; l < loops.size(); ++l) {
      oa += cnt[l] * A.stride(loops[l]);
      ob += cnt[l] * B.stride(loops[l]);
      oc += cnt[l] * C.stride(loops[l]);
    }
    push(p, A.buf, oa, A.dt, conja, B.buf, ob, B.dt, 0, C.buf, oc, ma, ka, kb, nb, mc, nc, nbatch, sba, sbb, sbc);
    for (size_t l = loops.size(); l-- > 0;) {
      if (++cnt[l] < ext[l]) break;
      cnt[l] = 0;
    }
  }
}

// Environment update with a stack of n MPO sites (mps/lib.py:121-166 contract_one_site_multi_mpo), n >= 1:
//   L: env E[a, b_1..b_n, c], bra[a, x_0, g, p], W_i[b_i, x_{i-1}, x_i, f_i], ket[c, x_n, g, h] -> out[p, f_1..f_n, h]
//   R: env E[a, b_1..b_n, c], bra[p, x_0, g, a], W_i[q_i, x_{i-1}, x_i, b_i], ket[h, x_n, g, c] -> out[p, q_1..q_n, h]
// (layer 1 touches the bra, layer n the ket; g = traced ancilla).  Buffers: B_L env, B_C ket, B_BRA bra, B_OUT out,
// MPO sites w_buf(layer) = B_W0, B_W1, B_W2, B_W3; intermediates alternate between B_T1 and B_T2.
inline Plan plan_env_multi(int dtype, int domain, const mpse_dims& s, int n, const int64_t* wl, const int64_t* wr,
                           int env_dtype, int w_dtype, int bra_conj) {
  Plan p;
  if (n < 1 || n > 4) {
    p.error = "env_multi: 1 to 4 MPO layers";
    return p;
  }
  const int64_t d = s.d0, anc = s.danc > 0 ? s.danc : 1;
  const int64_t Dlb = s.Dl_bra, Dlk = s.Dl_ket, Drb = s.Dr_bra, Drk = s.Dr_ket;
  int64_t WL = 1, WR = 1;
  for (int i = 0; i < n; ++i) WL *= wl[i], WR *= wr[i];
  const bool left = domain == MPSE_DOMAIN_L;
  if (!left && domain != MPSE_DOMAIN_R) {
    p.error = "env: bad domain";
    return p;
  }
  // sizes of the intermediates: after the first GEMM the channels still carry the incoming bonds; layer by layer
  // (ket side first) they are replaced by the outgoing ones
  int64_t win[4], wout[4];
  for (int i = 0; i < n; ++i) win[i] = left ? wl[i] : wr[i], wout[i] = left ? wr[i] : wl[i];
  const int64_t Da = left ? Dlb : Drb;      // bond of the environment row that stays open until the last step
  const int64_t Dh = left ? Drk : Dlk;      // open ket bond
  const int64_t Din = left ? Dlk : Drk;     // contracted ket bond
  int64_t cur = Da * d * anc * Dh;
  for (int i = 0; i < n; ++i) cur *= win[i];
  int64_t maxsz = cur;
  {
    int64_t t = cur;
    for (int i = n - 1; i >= 0; --i) {
      t = t / win[i] * wout[i];
      if (t > maxsz) maxsz = t;
    }
  }
  p.tmp_elems[0] = p.tmp_elems[1] = maxsz;
  int tb = B_T1;
  if (left) {
    // T_n[(b_1..b_n), a, x, g, h] = sum_c E[a, (b), c] ket[c, (x, g, h)]   (channel-outer layout)
    push_env_times(p, B_L, env_dtype, B_C, dtype, tb, Dlb, WL, Dlk, d * anc * Drk, 0, true);
    // layer i = n..1: T_{i-1}[P, a, y, f_i, F, g, h] = sum_{b_i, x} W_i[b_i, y, x, f_i] T_i[P, b_i, a, x, F, g, h]
    int64_t P = WL, F = 1;
    for (int i = n - 1; i >= 0; --i) {
      P /= wl[i];
      View W(w_buf(i), w_dtype, {{'b', wl[i]}, {'y', d}, {'x', d}, {'f', wr[i]}});
      View Tin(tb, dtype, {{'P', P}, {'b', wl[i]}, {'a', Dlb}, {'x', d}, {'N', F * anc * Drk}});
      const int to = tb == B_T1 ? B_T2 : B_T1;
      View Tout(to, dtype, {{'P', P}, {'a', Dlb}, {'y', d}, {'f', wr[i]}, {'N', F * anc * Drk}});
      contract(p, W, Tin, Tout, "yf", "bx", "N", "a", "P");
      tb = to;
      F *= wr[i];
    }
    // out[p, (F, h)] = sum_{a, y, g} bra*[(a, y, g), p] T_0[(a, y), F, g, h]
    push(p, B_BRA, 0, dtype, bra_conj, tb, 0, dtype, 0, B_OUT, 0, i1(Drb, 1), i1(Dlb * d * anc, Drb),
         i2(Dlb * d, anc, WR * anc * Drk, Drk), i2(WR, Drk, anc * Drk, 1), i1(Drb, WR * Drk), i1(WR * Drk, 1));
    return p;
  }
  // R: T_n[h, x, g, (b_1..b_n), a] = sum_c ket[(h, x, g), c] E[a, (b), c]
  push(p, B_C, 0, dtype, 0, B_L, 0, env_dtype, 0, tb, 0, i1(Dlk * d * anc, Drk), i1(Drk, 1), i1(Drk, 1),
       i2(WR, Drb, Drk, WR * Drk), i1(Dlk * d * anc, WR * Drb), i1(WR * Drb, 1));
  // layer i = n..1: T_{i-1}[h, q_i, Q, y, g, P, a] = sum_{x, b_i} W_i[q_i, y, x, b_i] T_i[h, Q, x, g, P, b_i, a]
  int64_t P = WR, Q = 1;
  for (int i = n - 1; i >= 0; --i) {
    P /= wr[i];
    View W(w_buf(i), w_dtype, {{'q', wl[i]}, {'y', d}, {'x', d}, {'b', wr[i]}});
    View Tin(tb, dtype, {{'h', Dlk}, {'Q', Q}, {'x', d}, {'g', anc}, {'P', P}, {'b', wr[i]}, {'a', Drb}});
    const int to = tb == B_T1 ? B_T2 : B_T1;
    View Tout(to, dtype, {{'h', Dlk}, {'q', wl[i]}, {'Q', Q}, {'y', d}, {'g', anc}, {'P', P}, {'a', Drb}});
    contract(p, W, Tin, Tout, "qy", "xb", "Pa", "h", "Qg");
    tb = to;
    Q *= wl[i];
  }
  // out[p, (Q, h)] = sum_{y, g, a} bra*[p, (y, g, a)] T_0[h, Q, (y, g, a)]
  push(p, B_BRA, 0, dtype, bra_conj, tb, 0, dtype, 0, B_OUT, 0, i1(Dlb, d * anc * Drb), i1(d * anc * Drb, 1),
       i1(d * anc * Drb, 1), i2(WL, Dlk, d * anc * Drb, WL * d * anc * Drb), i1(Dlb, WL * Dlk), i1(WL * Dlk, 1));
  return p;
}

// Two-layer effective Hamiltonian of the (H - omega)^2 functional, mps/hop_expr.py:24-52 (same MPO sites in both
// layers, no ancilla):
//   1-site  abcd, befg, cfhi, jgik, aej -> dhk          L (Dl, wl, wl, Dl), R (Dr, wr, wr, Dr), W0 (wl, d0, d0, wr)
//   2-site  abcd, befg, cfhi, gjkl, ikmn, olnp, aejo -> dhmp         W0 (wl, d0, d0, wm), W1 (wm, d1, d1, wr)
// The centre enters through the FIRST bond index of L / R and the upper physical legs, the result leaves through
// the last one - exactly the reference's index placement.
inline Plan plan_heff2(int dtype, const mpse_heff& h) {
  Plan p;
  const mpse_dims& s = h.dims;
  const int64_t Dl = s.Dl_ket, Dr = s.Dr_ket, wl = s.wl, wr = s.wr, d0 = s.d0;
  if (h.nsite != 1 && h.nsite != 2) {
    p.error = "heff (two layers): one- or two-site centres only";
    return p;
  }
  if ((s.Dl_bra > 0 && s.Dl_bra != Dl) || (s.Dr_bra > 0 && s.Dr_bra != Dr)) {
    p.error = "heff (two layers): bra bonds must equal ket bonds";
    return p;
  }
  // z: a pure batch index next to the right bond (several centres at once - the columns of the identity when the
  // dense projected operator is wanted, mps/gs.py:307-369); 1-site: C (Dl, d0, z, Dr) with z = danc,
  // 2-site: C (Dl, d0, d1, z, Dr) with z = danc1 (danc must be 1)
  if (h.nsite == 1) {
    const int64_t nz = s.danc > 0 ? s.danc : 1;
    const int64_t big = std::max(wl * wl, std::max(wl * wr, wr * wr)) * Dl * d0 * nz * Dr;
    p.tmp_elems[0] = p.tmp_elems[1] = p.tmp_elems[2] = big;
    View L(B_L, h.l_dtype, {{'a', Dl}, {'b', wl}, {'c', wl}, {'d', Dl}});
    View R(B_R, h.r_dtype, {{'j', Dr}, {'g', wr}, {'i', wr}, {'k', Dr}});
    View W(B_W0, h.w_dtype, {{'b', wl}, {'e', d0}, {'f', d0}, {'g', wr}});       // layer 1 labels
    View W2(B_W0, h.w_dtype, {{'c', wl}, {'f', d0}, {'h', d0}, {'i', wr}});      // layer 2 labels
    View C(B_C, dtype, {{'a', Dl}, {'e', d0}, {'z', nz}, {'j', Dr}});
    View T1(B_T1, dtype, {{'b', wl}, {'c', wl}, {'d', Dl}, {'e', d0}, {'z', nz}, {'j', Dr}});
    contract(p, L, C, T1, "bcd", "a", "ezj");
    View T2(B_T2, dtype, {{'c', wl}, {'d', Dl}, {'f', d0}, {'g', wr}, {'z', nz}, {'j', Dr}});
    contract(p, W, T1, T2, "fg", "be", "zj", "cd");
    View T3(B_T3, dtype, {{'d', Dl}, {'h', d0}, {'z', nz}, {'j', Dr}, {'g', wr}, {'i', wr}});
    contract(p, W2, T2, T3, "hi", "cf", "zjg", "d");
    View O(B_OUT, dtype, {{'d', Dl}, {'h', d0}, {'z', nz}, {'k', Dr}});
    contract(p, T3, R, O, "dhz", "jgi", "k");
    return p;
  }
  if (s.danc > 1) {
    p.error = "heff (two layers, 2-site): the batch index is danc1, danc must be 1";
    return p;
  }
  const int64_t d1 = s.d1, wm = s.wm, nz = s.danc1 > 0 ? s.danc1 : 1;
  {
    int64_t big = 0;
    const int64_t cand[5] = {wl * wl, wl * wm, wm * wm, wm * wr, wr * wr};
    for (int64_t c : cand) big = c > big ? c : big;
    p.tmp_elems[0] = p.tmp_elems[1] = p.tmp_elems[2] = big * d0 * d1 * nz * Dl * Dr;
  }
  View L(B_L, h.l_dtype, {{'a', Dl}, {'b', wl}, {'c', wl}, {'d', Dl}});
  View R(B_R, h.r_dtype, {{'o', Dr}, {'l', wr}, {'n', wr}, {'p', Dr}});
  View C(B_C, dtype, {{'a', Dl}, {'e', d0}, {'j', d1}, {'z', nz}, {'o', Dr}});
  View Wa(B_W0, h.w_dtype, {{'b', wl}, {'e', d0}, {'f', d0}, {'g', wm}});
  View Wb(B_W0, h.w_dtype, {{'c', wl}, {'f', d0}, {'h', d0}, {'i', wm}});
  View Wc(B_W1, h.w_dtype, {{'g', wm}, {'j', d1}, {'k', d1}, {'l', wr}});
  View Wd(B_W1, h.w_dtype, {{'i', wm}, {'k', d1}, {'m', d1}, {'n', wr}});
  View T1(B_T1, dtype, {{'b', wl}, {'c', wl}, {'d', Dl}, {'e', d0}, {'j', d1}, {'z', nz}, {'o', Dr}});
  contract(p, L, C, T1, "bcd", "a", "ejzo");
  View T2(B_T2, dtype, {{'c', wl}, {'d', Dl}, {'f', d0}, {'g', wm}, {'j', d1}, {'z', nz}, {'o', Dr}});
  contract(p, Wa, T1, T2, "fg", "be", "jzo", "cd");
  View T3(B_T3, dtype, {{'d', Dl}, {'h', d0}, {'i', wm}, {'g', wm}, {'j', d1}, {'z', nz}, {'o', Dr}});
  contract(p, Wb, T2, T3, "hi", "cf", "gjzo", "d");
  View T4(B_T1, dtype, {{'d', Dl}, {'h', d0}, {'i', wm}, {'k', d1}, {'l', wr}, {'z', nz}, {'o', Dr}});
  contract(p, Wc, T3, T4, "kl", "gj", "zo", "dhi");
  View T5(B_T2, dtype, {{'d', Dl}, {'h', d0}, {'m', d1}, {'z', nz}, {'o', Dr}, {'l', wr}, {'n', wr}});
  contract(p, Wd, T4, T5, "mn", "ik", "zol", "dh");
  View O(B_OUT, dtype, {{'d', Dl}, {'h', d0}, {'m', d1}, {'z', nz}, {'p', Dr}});
  contract(p, T5, R, O, "dhmz", "oln", "p");
  return p;
}

}  // namespace mpse_plan
