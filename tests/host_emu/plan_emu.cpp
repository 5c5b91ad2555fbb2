// TEST INFRASTRUCTURE: executes the product's contraction plans (renormalizer_amd/csrc/
// mpse_plans.h) on HOST memory with a naive loop implementation of the strided-GEMM
// contract, so the index algebra of the hot-path contractions is checked on a CPU-only
// machine against numpy (tests/test_plans_host.py).  Never linked into libmpsengine.so.
#include <complex>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../renormalizer_amd/csrc/mpse_plans.h"

using namespace mpse_plan;
typedef std::complex<double> cd;

static inline int64_t off(const mpse_index& m, int64_t i) {
  if (m.lo_ext >= m.ext) return i * m.s_lo;
  return (i / m.lo_ext) * m.s_hi + (i % m.lo_ext) * m.s_lo;
}

static inline cd ld(const void* p, int dt, int64_t o, int conj) {
  if (dt == MPSE_C128) {
    cd v = ((const cd*)p)[o];
    return conj ? std::conj(v) : v;
  }
  return cd(((const double*)p)[o], 0.0);
}

static void naive_gemm(const Step& s, const void* A, const void* B, void* C, const void* Cin) {
  const int dtc = (s.dta == MPSE_C128 || s.dtb == MPSE_C128) ? MPSE_C128 : MPSE_F64;
  for (int64_t b = 0; b < s.batch; ++b)
    for (int64_t i = 0; i < s.ma.ext; ++i)
      for (int64_t j = 0; j < s.nb.ext; ++j) {
        cd acc = 0;
        for (int64_t k = 0; k < s.ka.ext; ++k)
          acc += ld(A, s.dta, b * s.sba + off(s.ma, i) + off(s.ka, k), s.conja) *
                 ld(B, s.dtb, b * s.sbb + off(s.kb, k) + off(s.nb, j), s.conjb);
        int64_t o = b * s.sbc + off(s.mc, i) + off(s.nc, j);
        // the beta term comes from C itself or, for steps with a beta source, from that tensor through its own maps
        const void* src = Cin ? Cin : C;
        const int64_t oi = Cin ? off(s.mcin, i) + off(s.ncin, j) : o;
        if (dtc == MPSE_C128)
          ((cd*)C)[o] = acc + (s.beta != 0.0 ? s.beta * ((const cd*)src)[oi] : cd(0));
        else
          ((double*)C)[o] = acc.real() + (s.beta != 0.0 ? s.beta * ((const double*)src)[oi] : 0.0);
      }
}

static void naive_copy(const Step& s, int dt, const void* A, void* C) {
  for (int64_t i = 0; i < s.ma.ext; ++i)
    for (int64_t j = 0; j < s.ka.ext; ++j) {
      const int64_t si = off(s.ma, i) + off(s.ka, j), di = off(s.mc, i) + off(s.nc, j);
      if (dt == MPSE_C128)
        ((cd*)C)[di] = ((const cd*)A)[si];
      else
        ((double*)C)[di] = ((const double*)A)[si];
    }
}

static int run(int dtype, const Plan& p, const void* bufs_in[B_COUNT]) {
  if (p.error) return MPSE_ERR_SHAPE;
  const void* bufs[B_COUNT];
  for (int i = 0; i < B_COUNT; ++i) bufs[i] = bufs_in[i];
  const size_t es = dtype == MPSE_C128 ? 16 : 8;
  std::vector<char> t[3];
  for (int i = 0; i < 3; ++i) {
    t[i].assign(size_t(p.tmp_elems[i]) * es + 16, 0x7f);  // poison: every element must be written before use
    bufs[B_T1 + i] = t[i].data();
  }
  for (const Step& s : p.steps) {
    const size_t ea = s.dta == MPSE_C128 ? 16 : 8, eb = s.dtb == MPSE_C128 ? 16 : 8;
    const int dtc = (s.dta == MPSE_C128 || s.dtb == MPSE_C128) ? MPSE_C128 : MPSE_F64;
    if (dtc != dtype && s.kind != K_WMIX) return MPSE_ERR_ARG;
    if (s.kind == K_WMIX) {   // dst[a, dd, k] = sum_terms sum_e W[b, dd, e, f] src[a, e, k]
      const double* W = (const double*)bufs[s.b];
      const int64_t nchunk = (s.wp_d + WM_CHUNK - 1) / WM_CHUNK;
      for (const WMixDst& q : s.mix)
        for (int64_t a = 0; a < s.wp_Da; ++a)
          for (int64_t dd = 0; dd < s.wp_d; ++dd)
            for (int64_t k = 0; k < s.wp_Dk; ++k) {
              cd acc = 0;
              for (int t = 0; t < q.nterm; ++t) {
                const WMixTerm& tm = q.term[t];
                const int64_t c = dd / WM_CHUNK;
                if (c >= nchunk) return MPSE_ERR_ARG;
                // only the columns the plan declares for this chunk are read - as on the device
                for (int64_t e = tm.e_lo[c]; e < tm.e_hi[c]; ++e) {
                  const double v = tm.ident ? (e == dd ? 1.0 : 0.0) : W[((tm.b * s.wp_d + dd) * s.wp_d + e) * s.wp_wr + tm.f];
                  if (v != 0.0) acc += v * ld(bufs[tm.src], dtype, tm.src_off + a * tm.s_a + e * tm.s_d + k, 0);
                }
              }
              const int64_t o = q.dst_off + a * q.s_a + dd * q.s_d + k;
              if (dtype == MPSE_C128)
                ((cd*)const_cast<void*>(bufs[q.dst]))[o] = acc;
              else
                ((double*)const_cast<void*>(bufs[q.dst]))[o] = acc.real();
            }
      continue;
    }
    if (s.kind == K_GGEMM) {   // every group: C = sum over its segments of A_seg . B_seg (+ beta C)
      for (const GGroupPlan& g : s.groups) {
        void* Cg = const_cast<void*>(bufs[g.cbuf]);
        for (int64_t i = 0; i < s.ma.ext; ++i)
          for (int64_t jn = 0; jn < s.nb.ext; ++jn) {
            cd acc = 0;
            for (int q = 0; q < g.nseg; ++q) {
              const GSegPlan& sg = g.seg[q];
              for (int64_t k = 0; k < s.ka.ext; ++k)
                acc += ld(bufs[sg.abuf], s.dta, sg.a_off + off(s.ma, i) + off(s.ka, k), 0) *
                       ld(bufs[sg.bbuf], s.dtb, sg.b_off + off(s.kb, k) + off(s.nb, jn), 0);
            }
            const int64_t o = g.c_off + off(s.mc, i) + off(s.nc, jn);
            // split2: the second result receives a part of the sum (here: nothing), the caller adds the two
            if (g.split2) {
              if (dtype == MPSE_C128)
                ((cd*)const_cast<void*>(bufs[B_OUT2]))[o] = 0.25 * acc, acc *= 0.75;
              else
                ((double*)const_cast<void*>(bufs[B_OUT2]))[o] = 0.25 * acc.real(), acc *= 0.75;
            }
            if (dtype == MPSE_C128)
              ((cd*)Cg)[o] = acc + (g.beta != 0.0 ? g.beta * ((const cd*)Cg)[o] : cd(0));
            else
              ((double*)Cg)[o] = acc.real() + (g.beta != 0.0 ? g.beta * ((const double*)Cg)[o] : 0.0);
          }
      }
      continue;
    }
    if (s.kind == K_COPY) {
      naive_copy(s, dtc, (const char*)bufs[s.a] + s.a_off * ea, (char*)const_cast<void*>(bufs[s.c]) + s.c_off * es);
      continue;
    }
    naive_gemm(s, (const char*)bufs[s.a] + s.a_off * ea, (const char*)bufs[s.b] + s.b_off * eb,
               (char*)const_cast<void*>(bufs[s.c]) + s.c_off * es,
               s.cin >= 0 ? (const char*)bufs[s.cin] + s.cin_off * es : nullptr);
  }
  return MPSE_OK;
}

extern "C" int emu_heff_apply(int dtype, const mpse_heff* h, const void* C, void* out) {
  Plan p = plan_heff(dtype, *h);
  const void* bufs[B_COUNT] = {nullptr};
  bufs[B_L] = h->L;
  bufs[B_R] = h->R;
  bufs[B_W0] = h->W0;
  bufs[B_W1] = h->W1;
  bufs[B_C] = C;
  bufs[B_OUT] = out;
  return run(dtype, p, bufs);
}

extern "C" int emu_env_update(int dtype, int domain, const mpse_dims* dims, const void* env, int env_dtype,
                              const void* ket, const void* bra, int bra_conj, const void* W, int w_dtype, void* out) {
  if (!bra) bra = ket;
  Plan p = plan_env(dtype, domain, *dims, env_dtype, w_dtype, bra_conj);
  const void* bufs[B_COUNT] = {nullptr};
  bufs[B_L] = env;
  bufs[B_W0] = W;
  bufs[B_C] = ket;
  bufs[B_BRA] = bra;
  bufs[B_OUT] = out;
  return run(dtype, p, bufs);
}

extern "C" int emu_env_update_multi(int dtype, int domain, const mpse_dims* dims, int n, const int64_t* wl,
                                    const int64_t* wr, const void* env, int env_dtype, const void* ket, const void* bra,
                                    int bra_conj, const void* const* W, int w_dtype, void* out) {
  if (!bra) bra = ket;
  Plan p = plan_env_multi(dtype, domain, *dims, n, wl, wr, env_dtype, w_dtype, bra_conj);
  const void* bufs[B_COUNT] = {nullptr};
  bufs[B_L] = env;
  for (int i = 0; i < n; ++i) bufs[w_buf(i)] = W[i];
  bufs[B_C] = ket;
  bufs[B_BRA] = bra;
  bufs[B_OUT] = out;
  return run(dtype, p, bufs);
}

extern "C" int emu_heff_apply2(int dtype, const mpse_heff* h, const void* C, void* out) {
  Plan p = plan_heff2(dtype, *h);
  const void* bufs[B_COUNT] = {nullptr};
  bufs[B_L] = h->L;
  bufs[B_R] = h->R;
  bufs[B_W0] = h->W0;
  bufs[B_W1] = h->W1;
  bufs[B_C] = C;
  bufs[B_OUT] = out;
  return run(dtype, p, bufs);
}

extern "C" void emu_set_unit_threshold(long long macs) { unit_threshold() = macs; }
// one-site matvec through the folded plan (the block structure of the MPO site read from the host copy of W);
// returns MPSE_ERR_SHAPE when the site does not qualify.  *nsteps receives the number of plan steps.
extern "C" int emu_heff_apply_fold(int dtype, const mpse_heff* h, const void* C, void* out, int* nsteps) {
  const WSiteInfo wi = analyse_mpo_site((const double*)h->W0, h->dims.wl, h->dims.d0, h->dims.wr);
  // the result in two parts where the plan makes use of it (complex centres whose rows fill whole tiles)
  Plan p = plan_heff1_fold(dtype, *h, wi, true);
  if (nsteps) *nsteps = (int)p.steps.size() + (p.two_results ? 100 : 0);
  const size_t es = dtype == MPSE_C128 ? 16 : 8;
  const int64_t n = (h->dims.Dl_bra > 0 ? h->dims.Dl_bra : h->dims.Dl_ket) * h->dims.d0 *
                    (h->dims.Dr_bra > 0 ? h->dims.Dr_bra : h->dims.Dr_ket);
  std::vector<char> out2(size_t(n) * es + 16, 0x7f);
  const void* bufs[B_COUNT] = {nullptr};
  bufs[B_L] = h->L;
  bufs[B_R] = h->R;
  bufs[B_W0] = h->W0;
  bufs[B_C] = C;
  bufs[B_OUT] = out;
  bufs[B_OUT2] = out2.data();
  const int st = run(dtype, p, bufs);
  if (st == MPSE_OK && p.two_results)
    for (int64_t i = 0; i < n * (int64_t)(es / 8); ++i) ((double*)out)[i] += ((const double*)out2.data())[i];
  return st;
}
extern "C" void emu_set_fold_min(long long macs, long long align, long long split2_min_kt) {
  fold_min() = macs;
  fold_align() = align;
  fold_split2_min_kt() = split2_min_kt;
}
extern "C" void emu_set_beta_source(int on) { beta_source_flag() = on != 0; }
