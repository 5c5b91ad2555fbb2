// Hot-path contractions: environment update and effective-Hamiltonian matvec.
// The index algebra lives in mpse_plans.h (a list of strided-GEMM steps); this file
// executes a plan on the device with the FP64-MFMA contraction kernel.
#include "mpse_internal.h"
#include "mpse_plans.h"

using namespace mpse_plan;

static int run_plan(mpse_ctx* ctx, int dtype, const Plan& p, const void* bufs_in[B_COUNT]) {
  if (p.error) return mpse_fail(ctx, MPSE_ERR_SHAPE, "%s", p.error);
  const void* bufs[B_COUNT];
  for (int i = 0; i < B_COUNT; ++i) bufs[i] = bufs_in[i];
  TmpBuf t1(ctx), t2(ctx), t3(ctx);
  const size_t es = dtype_size(dtype);
  if (p.tmp_elems[0]) {
    MPSE_TRY(t1.alloc(size_t(p.tmp_elems[0]) * es));
    bufs[B_T1] = t1.p;
  }
  if (p.tmp_elems[1]) {
    MPSE_TRY(t2.alloc(size_t(p.tmp_elems[1]) * es));
    bufs[B_T2] = t2.p;
  }
  if (p.tmp_elems[2]) {
    MPSE_TRY(t3.alloc(size_t(p.tmp_elems[2]) * es));
    bufs[B_T3] = t3.p;
  }
  for (const Step& s : p.steps) {
    const char* a = (const char*)bufs[s.a] + size_t(s.a_off) * dtype_size(s.dta);
    const char* b = (const char*)bufs[s.b] + size_t(s.b_off) * dtype_size(s.dtb);
    const int dtc = (s.dta == MPSE_C128 || s.dtb == MPSE_C128) ? MPSE_C128 : MPSE_F64;
    if (dtc != dtype)  // every step of a plan must produce the working dtype
      return mpse_fail(ctx, MPSE_ERR_ARG, "plan step dtype mismatch (got %d want %d)", dtc, dtype);
    char* c = (char*)const_cast<void*>(bufs[s.c]) + size_t(s.c_off) * dtype_size(dtc);
    if (!bufs[s.a] || !bufs[s.b] || !bufs[s.c]) return mpse_fail(ctx, MPSE_ERR_ARG, "plan: missing buffer");
    MPSE_TRY(gemm_call(ctx, s.dta, s.dtb, s.conja, s.conjb, s.ma, s.ka, s.kb, s.nb, s.mc, s.nc, s.batch, s.sba,
                       s.sbb, s.sbc, a, b, c));
  }
  return MPSE_OK;
}

extern "C" int mpse_heff_apply(mpse_ctx* ctx, int dtype, const mpse_heff* h, const void* C, void* out) {
  if (!ctx || !h || !C || !out || !h->L || !h->R) return MPSE_ERR_ARG;
  if (dtype != MPSE_C128 && (h->l_dtype == MPSE_C128 || h->r_dtype == MPSE_C128 || h->w_dtype == MPSE_C128))
    return mpse_fail(ctx, MPSE_ERR_ARG, "heff_apply: real centre with complex operator parts");
  Plan p = plan_heff(dtype, *h);
  const void* bufs[B_COUNT] = {nullptr};
  bufs[B_L] = h->L;
  bufs[B_R] = h->R;
  bufs[B_W0] = h->W0;
  bufs[B_W1] = h->W1;
  bufs[B_C] = C;
  bufs[B_OUT] = out;
  return run_plan(ctx, dtype, p, bufs);
}

extern "C" int mpse_env_update(mpse_ctx* ctx, int dtype, int domain, const mpse_dims* dims, const void* env,
                               int env_dtype, const void* ket, const void* bra, int bra_conj, const void* W,
                               int w_dtype, void* out) {
  if (!ctx || !dims || !env || !ket || !W || !out) return MPSE_ERR_ARG;
  if (dtype != MPSE_C128 && (env_dtype == MPSE_C128 || w_dtype == MPSE_C128))
    return mpse_fail(ctx, MPSE_ERR_ARG, "env_update: real sites with complex env/mpo");
  if (!bra) {
    bra = ket;
    if (dims->Dl_bra != dims->Dl_ket || dims->Dr_bra != dims->Dr_ket)
      return mpse_fail(ctx, MPSE_ERR_SHAPE, "env_update: bra==NULL needs equal bonds");
  }
  Plan p = plan_env(dtype, domain, *dims, env_dtype, w_dtype, bra_conj);
  const void* bufs[B_COUNT] = {nullptr};
  bufs[B_L] = env;
  bufs[B_W0] = W;
  bufs[B_C] = ket;
  bufs[B_BRA] = bra;
  bufs[B_OUT] = out;
  return run_plan(ctx, dtype, p, bufs);
}
