// Internal declarations shared by the translation units of libmpsengine.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdlib>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/mpsengine.h"

struct mpse_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int n_cu = 0;
  char err[512] = {0};
  char dev_name[128] = {0};

  // size-bucketed caching allocator: hipMalloc/hipFree synchronise the device, the
  // sweep allocates per site.  Single in-order stream => a freed block may be handed
  // out again immediately.
  std::mutex pool_mu;                       // mpse_malloc / mpse_free may be called from a GC pass on another thread
  std::multimap<size_t, void*> free_blocks;
  std::unordered_map<void*, size_t> live;   // ptr -> bucket size
  size_t pool_bytes = 0;
  size_t in_use_bytes = 0;
  unsigned long long n_device_allocs = 0;   // hipMalloc calls (pool misses)

  // optional HIP-event profiling of the contraction kernel (mpse_prof_*)
  struct ProfRec {
    hipEvent_t e0, e1;
    int variant;
    double flops, bytes;
  };
  bool prof_on = false;
  int prof_stride = 1;      // time every prof_stride-th contraction launch (sampling keeps the overhead small)
  long long prof_counter = 0;
  std::vector<ProfRec> prof_pending;
  std::vector<hipEvent_t> prof_free_events;
  static constexpr int PROF_NVAR = 8;     // 0-3: contraction kernel by operand types, 4: Lanczos vector kernels, 5: block QR, 6: block SVD,
                                          // 7: fused bond / two-level-site matvec (k_heff0_fused)
  int64_t prof_svd_sweeps = 0;            // Jacobi sweeps of the timed mpse_block_svd calls (largest block of each call)
  double prof_ms[PROF_NVAR] = {0};
  double prof_flops[PROF_NVAR] = {0};
  double prof_bytes[PROF_NVAR] = {0};
  int64_t prof_launches[PROF_NVAR] = {0};
  // K tiles (64 x 64 x 16 multiply-add blocks) actually multiplied by the timed contraction launches, per variant:
  // structural-zero skipping makes this smaller than the dense count (device counters, one atomic per workgroup)
  unsigned long long* prof_ktiles = nullptr;

  // pinned ring for small host->device uploads that must not stall the stream (index lists, descriptors)
  char* stage = nullptr;
  char* stage_dev = nullptr;   // the ring as the device sees it (mapped), null: copies go through the runtime
  size_t stage_size = 0, stage_pos = 0;

  // small pinned staging buffer for scalar read-backs
  double* pinned = nullptr;     // 4096 doubles
  double* pinned_dev = nullptr; // the same buffer as the device sees it (mapped, host coherent)
  unsigned long long publish_seq = 0;
  double* dscratch = nullptr;   // device scratch for reductions (1<<16 doubles); the last 8 hold flag words
  unsigned int flag_gen = 0;    // generation stamp of the Lanczos convergence flag (no per-check memset)
  // Tile-occupancy masks of operands that stay constant over one Krylov solve (the two environments): computed by
  // the first matvec, reused by the others.  Active only inside mpse_expm_lanczos, for operands inside the ranges
  // registered there.
  struct OccKey {
    const void* ptr;
    long long r_ext, r_lo, r_shi, r_slo, k_ext, k_lo, k_shi, k_slo, sb;
    int nrows, tiles, nkw, batch, K, cplx;
  };
  struct OccEntry {
    OccKey key;
    void* mask;
  };
  // Device word that turns every contraction launch into a no-op once it is non-zero: set for the duration of an
  // asynchronous Lanczos solve, whose iterations are enqueued ahead of the convergence decision (mpse_vec.hip)
  const int* skip_flag = nullptr;
  // Krylov dimension of the last solve per problem class (number of sites, vector length): how far to run ahead
  std::unordered_map<unsigned long long, int> lz_hint;
  // Block structure of MPO sites the caller has described (mpse_mpo_site_hint), by device pointer: large one-site
  // matvecs on such a site take the folded plan (mpse_plans.h).  Dropped when the buffer is freed, and when one of
  // the element-level entry points writes into it (wsite_written: copies, memset, scal, conj, axpy).
  struct WSiteEntry {
    std::shared_ptr<void> info;      // -> mpse_plan::WSiteInfo
    size_t bytes = 0;                // extent of the described site on the device
  };
  std::unordered_map<const void*, WSiteEntry> wsite_info;
  // Deferred calls (mpse_defer_*): mpse_gemm / mpse_block_qr / mpse_env_update issued while a list is being recorded
  // are stored with copies of their arguments; an armed list runs at the end of the next mpse_expm_lanczos, right
  // after the solve has been enqueued to its end.  While a list is recorded or waiting, freed device blocks are held
  // back (a recorded call may still read them).
  std::vector<std::function<int()>> defer_ops[2];
  int defer_recording = -1;
  int defer_armed = -1;
  bool defer_hold = false;                  // guarded by pool_mu
  std::vector<void*> defer_frees;           // guarded by pool_mu
  // Request to the contraction plan that produces a matvec result (mpse_heff_apply): also accumulate
  // sum conj(result) . y, as per-workgroup partials at `part` (room for `cap` of them); nb_out = number written
  // (0: the plan could not take it, the caller runs its own reduction)
  struct DotReq {
    const void* y = nullptr;
    double* part = nullptr;
    int cap = 0;
    int nb_out = 0;
  } dot_req;
  bool dot_now = false;   // set by run_plan for the one GEMM launch that completes the result
  // A plan step whose CONSUMER adds the K slices of a split product while it reads them (the elementwise MPO step of
  // the small sites): the product leaves its raw slices at ptr (slice s at ptr + s * M * N elements, compact like C)
  // and launches no reduction; `used` = number of slices (0: the product stored C as usual)
  struct SlicesReq {
    void* ptr = nullptr;
    size_t cap_bytes = 0;
    int used = 0;
  } slices_req;
  // A caller of mpse_heff_apply that can take the result as the SUM of several tensors (the Lanczos update adds them
  // while it reads) offers a buffer of cap_elems elements of the working dtype, n of them per part: the last product
  // of the plan may then leave its K slices there instead of reducing them (mpse_gemm.hip: split products, halved
  // tiles).  used = number of parts written at ptr, ptr + n, .. (0: the result is complete in `out`, as usual).
  struct PartsReq {
    void* ptr = nullptr;
    long long cap_elems = 0, n = 0;
    int used = 0;
    // The caller can also take parts that hold only SOME 16 x 16 tiles of the result (mpse_heff0.hip): the callee then
    // sets `mask` (device; one 64-bit word per tile of the result viewed as a matrix with rows of mask_row elements,
    // mask_tiles tiles per tile row: bit s = part s holds the tile) and the consumer adds exactly the parts named there
    bool masked_ok = false;
    const unsigned long long* mask = nullptr;
    int mask_row = 0, mask_tiles = 0;
  } parts_req;
  // Tile-occupancy mask of the centre tensor as operand B of the first products of a matvec, supplied by the caller
  // (mpse_expm_centre_mask: the structural pattern of the quantum numbers, the same for every Krylov vector).
  // `pending` holds what the caller set for the next solve; during that solve lo / hi delimit the Krylov vectors.
  struct CMask {
    const void* ptr = nullptr;
    long long bytes = 0;
    const char* lo = nullptr;
    const char* hi = nullptr;
  } cmask_pending, cmask;
  // beta source for the next GEMM call (consumed by it): C = A.B + beta * Cin(i, j) with Cin's own index maps
  struct CinReq {
    const void* ptr = nullptr;
    mpse_index m{}, n{};
  } cin_req;
  // Debug: per-workgroup timeline of the contraction kernel (MPSE_GEMM_TRACE=<file>): every workgroup appends one
  // record (grid, K tiles multiplied, s_memtime stamps of its phases); mpse_prof_get writes the file.
  unsigned long long* gemm_trace = nullptr;   // [1 + GEMM_TRACE_CAP * GEMM_TRACE_WORDS] words: counter, then records
  bool gemm_trace_checked = false;
  bool occ_cache_on = false;
  const char* occ_lo[2] = {nullptr, nullptr};
  const char* occ_hi[2] = {nullptr, nullptr};
  std::vector<OccEntry> occ_cache;
  // launch orders of block-sparse products (k_tile_order), kept like the masks they were computed from
  struct PermEntry {
    const void *amask, *bmask;
    int tiles_m, tiles_n, nkt;
    void* perm;
  };
  std::vector<PermEntry> perm_cache;
  // Transposed right environment of the small-centre matvec (mpse_small.hip), kept for the running solve like the masks
  struct SmallRt {
    const void* src = nullptr;
    void* rt = nullptr;
    size_t bytes = 0;
  } small_rt;
  bool small_rt_scope = false;   // a solver without occupancy caches (Davidson) keeps the transposed copy as well
  // Per-solve data of the fused 0-site matvec (mpse_heff0.hip): transposed right environment, tile flags, part mask
  struct F0Cache {
    void* buf = nullptr;
    const void *L = nullptr, *R = nullptr, *W = nullptr, *cmask = nullptr;
    int Dl = 0, Dr = 0, w = 0, nsite = -1;
  } f0;
  long long f0_launches[2] = {0, 0};   // fused matvec launches: bond matrices, two-level sites
  // mpse_block_qr: decompositions that took the Cholesky-QR path / that fell back from it to Householder
  long long qr_chol_calls = 0, qr_chol_fallbacks = 0, qr_calls = 0;
  // optimistic mode of the Cholesky-QR path (mpse_block_qr_optimistic): breakdowns raise this sticky device word
  // instead of being read back per decomposition
  bool qr_optimistic = false;
  int qr_scheme = -1;            // mpse_block_qr_scheme: -1 environment default, 0 Householder only, 1 default rule, 2 every eligible shape
  // four device words of the block QR: [0] sticky breakdown flag of the optimistic mode, [2] blocks factorised by the
  // Cholesky-QR kernels, [3] of them finished after two passes (mpse_block_qr_pass_stats); allocated by qr_words()
  int* qr_words_dev = nullptr;
};
int qr_words(mpse_ctx* ctx);     // allocate + zero ctx->qr_words_dev once (mpse_qr.hip)

int mpse_fail(mpse_ctx* ctx, int code, const char* fmt, ...);
// Makes ctx->device the calling thread's current HIP device (a new thread starts on device 0; allocations and
// launches follow the CURRENT device, not the stream's).  Every entry point that allocates or launches calls it.
inline int mpse_bind(mpse_ctx* ctx) {
  int cur = -1;
  if (hipGetDevice(&cur) == hipSuccess && cur == ctx->device) return MPSE_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) return mpse_fail(ctx, MPSE_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
  return MPSE_OK;
}
// asynchronous upload of a small host array through the pinned ring (the host buffer may be reused at once)
int stage_h2d(mpse_ctx* ctx, void* dst, const void* src_host, size_t bytes);
// zero fill as a plain kernel on the context stream (8-byte aligned ranges; others go through hipMemsetAsync)
int device_zero(mpse_ctx* ctx, void* dst, size_t bytes);
// an entry point is about to write `bytes` at `dst`: a described MPO site (mpse_mpo_site_hint) that overlaps the range no
// longer holds the values that were analysed - its description goes
inline void wsite_written(mpse_ctx* ctx, const void* dst, size_t bytes) {
  if (!dst || !bytes) return;
  std::lock_guard<std::mutex> lock(ctx->pool_mu);   // (before the first look at the map: mpse_free may run on a GC thread)
  if (ctx->wsite_info.empty()) return;
  const char* lo = static_cast<const char*>(dst);
  for (auto it = ctx->wsite_info.begin(); it != ctx->wsite_info.end();) {
    const char* w = static_cast<const char*>(it->first);
    if (lo < w + it->second.bytes && w < lo + bytes)
      it = ctx->wsite_info.erase(it);
    else
      ++it;
  }
}
// fold finished profiling records into the totals; call only when the stream is idle
void prof_drain(mpse_ctx* ctx);
// HIP-event bracket around a group of launches on the context stream; begin returns false when this call is not
// sampled (profiling off or not the N-th call) or no event could be had
bool prof_begin(mpse_ctx* ctx, int variant, double flops, double bytes, mpse_ctx::ProfRec* rec);
void prof_end(mpse_ctx* ctx, const mpse_ctx::ProfRec& rec);
// bracket that cannot leak its event pair: a scope left without end() (error return) hands the events back
struct ProfScope {
  mpse_ctx* ctx;
  mpse_ctx::ProfRec rec;
  bool on;
  ProfScope(mpse_ctx* c, int variant, double flops, double bytes) : ctx(c), on(prof_begin(c, variant, flops, bytes, &rec)) {}
  ProfScope(const ProfScope&) = delete;
  ProfScope& operator=(const ProfScope&) = delete;
  void end() {
    if (on) prof_end(ctx, rec);
    on = false;
  }
  ~ProfScope() {
    if (!on) return;
    ctx->prof_free_events.push_back(rec.e0);
    ctx->prof_free_events.push_back(rec.e1);
  }
};

#define MPSE_HIP(ctx, call)                                                              \
  do {                                                                                   \
    hipError_t _e = (call);                                                              \
    if (_e != hipSuccess)                                                                \
      return mpse_fail((ctx), (_e == hipErrorOutOfMemory) ? MPSE_ERR_OOM : MPSE_ERR_HIP, \
                       "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(_e)); \
  } while (0)

#define MPSE_TRY(call)            \
  do {                            \
    int _s = (call);              \
    if (_s != MPSE_OK) return _s; \
  } while (0)

#define MPSE_BIND(ctx) MPSE_TRY(mpse_bind(ctx))

// Entry points that may be recorded start with this: true -> the call was stored, return MPSE_OK.
#define MPSE_RECORDING(ctx) ((ctx)->defer_recording >= 0)
// runs and empties the armed list (no-op when none is armed); `status` of the solve it follows: a failed solve
// drops the list
int defer_replay(mpse_ctx* ctx, int status);

// RAII temporary from the pool
struct TmpBuf {
  mpse_ctx* ctx;
  void* p = nullptr;
  TmpBuf(mpse_ctx* c) : ctx(c) {}
  int alloc(size_t bytes) { return mpse_malloc(ctx, bytes ? bytes : 16, &p); }
  ~TmpBuf() {
    if (p) mpse_free(ctx, p);
  }
  template <class T>
  T* as() { return reinterpret_cast<T*>(p); }
};

static inline size_t dtype_size(int dt) { return dt == MPSE_C128 ? 16 : 8; }
static inline mpse_index idx1(int64_t ext, int64_t stride) { return mpse_index{ext, ext > 0 ? ext : 1, 0, stride}; }
static inline mpse_index idx2(int64_t hi_ext, int64_t lo_ext, int64_t s_hi, int64_t s_lo) {
  return mpse_index{hi_ext * lo_ext, lo_ext > 0 ? lo_ext : 1, s_hi, s_lo};
}

// One-launch matvec of small 0- / 1-site centres (mpse_small.hip); *taken says whether it ran (else: the plans)
int heff_small_try(mpse_ctx* ctx, int dtype, const mpse_heff* h, const void* C, void* out, bool* taken);
void heff_small_drop_cache(mpse_ctx* ctx);
// Fused 0-site matvec for large complex bond matrices (mpse_heff0.hip): number of parts it would deliver (0 = not
// eligible), the attempt itself (needs mpse_ctx::parts_req.masked_ok), and the release of its per-solve data
int heff0_fused_parts(const mpse_heff* h, int dtype);
int heff0_fused_try(mpse_ctx* ctx, int dtype, const mpse_heff* h, const void* C, const double* w_host, bool* taken);
void heff0_drop_cache(mpse_ctx* ctx);
struct SmallRtScope {   // for the duration of one eigensolve: the right environment does not change
  mpse_ctx* c;
  explicit SmallRtScope(mpse_ctx* ctx) : c(ctx) { c->small_rt_scope = true; }
  ~SmallRtScope() {
    c->small_rt_scope = false;
    heff_small_drop_cache(c);
  }
};

// convenience wrapper over mpse_gemm used by the contraction entry points
int gemm_call(mpse_ctx* ctx, int dta, int dtb, int conja, int conjb, mpse_index ma, mpse_index ka,
              mpse_index kb, mpse_index nb, mpse_index mc, mpse_index nc, int64_t batch, int64_t sba,
              int64_t sbb, int64_t sbc, const void* A, const void* B, void* C, double alpha = 1.0,
              double beta = 0.0, int skip_zero = 0);

// Grouped launch of the contraction kernel (mpse_gemm.hip): up to 8 groups of equal height dividing the tile rows, each
// with its own result C (same index maps) and up to 4 (A, B) operand pairs whose products are summed (+ beta C, beta 0
// or 1).  am / bm: tile-occupancy flags of the operands (occ_mask_get layout, row pitches below), null = dense.
struct GroupedSeg {
  const void* A = nullptr;
  const void* B = nullptr;
  const unsigned char* am = nullptr;
  const unsigned char* bm = nullptr;
};
struct GroupedGrp {
  GroupedSeg seg[4];
  void* C = nullptr;
  int nseg = 0;
  double beta = 0.0;
};
struct GroupedDesc {
  int dta = MPSE_F64, dtb = MPSE_F64;
  mpse_index ma{}, ka{}, kb{}, nb{}, mc{}, nc{};
  int ngrp = 0;
  GroupedGrp grp[8];
  int am_pitch = 0, bm_pitch = 0;
  bool masks_stable = false;   // the flags live as long as the running solve's caches: the launch order is kept with them
  // one group only: two workgroups per output tile, each over half of its occupied K tiles; the first half (+ beta C)
  // goes to C, the second to c2 (laid out like C): the consumer adds them
  bool split2 = false;
  void* c2 = nullptr;
};
int gemm_grouped(mpse_ctx* ctx, const GroupedDesc& d);
int occ_mask_get(mpse_ctx* ctx, const void* ptr, int dtype, mpse_index r, mpse_index k, TmpBuf& tmp,
                 const unsigned char** flags, int* pitch, bool* stable);

// Low-latency read-back of a few device doubles: a one-wave kernel copies them into the mapped pinned buffer
// and then publishes a sequence number; the host spins on that number instead of going through a copy-engine
// transfer plus hipStreamSynchronize (the gap the GPU idles after every convergence check shrinks from ~25 us to
// the launch latency).  count <= 1024; the values land at ctx->pinned + slot.
int publish_and_wait(mpse_ctx* ctx, const double* dsrc, int count, int slot);
// The waiting half alone, for a kernel that publishes by itself (writes `count` doubles at pinned_dev + slot, then the
// sequence number `seq` = double(++ctx->publish_seq) at pinned_dev + 4095, each followed by __threadfence_system()).
int publish_wait_seq(mpse_ctx* ctx, double seq, const double* dsrc, int count, int slot);

// reductions (mpse_vec.hip): results land in ctx->pinned after a stream sync
int dotc_sync(mpse_ctx* ctx, int dtype, const void* x, const void* y, int64_t n, double* re, double* im);

// Householder building blocks on column-major workspaces (mpse_qr.hip), shared with the SVD
struct HhParam {  // per reflector: H = I - tau v v^H, v = (1, scale * tail)
  double tau_re, tau_im;
  double scale_re, scale_im;
  // panel-blocked kernels (mpse_qr2.hip): inner products u_i^H u_l with the earlier reflectors i of the same
  // four-column panel (re, im; i = 0 .. l-1), by-products of the panel factorisation that the compact-WY
  // applications need for T
  double g[6];
};
int hh_factor_colmajor(mpse_ctx* ctx, bool cplx, double* ws, int mm, int nn, int k, HhParam* prm);
// nq >= k columns of Q are formed (columns beyond k span the orthogonal complement)
int hh_formq_colmajor(mpse_ctx* ctx, bool cplx, double* q, const double* ws, int mm, int k, const HhParam* prm,
                      int nq);

// Batched panel-blocked Householder QR (mpse_qr2.hip): blocks live in one column-major workspace
struct QrBlk {
  long long ws_off;  // element offset of the mm x nn block inside the workspace
  long long q_off;   // element offset of its mm x k Q inside the Q buffer
  int mm, nn, k;
  int prm_off;       // offset of its reflector parameters
  int nq = 0;        // columns of Q to form (0 -> k); columns beyond k complete the basis (full_matrices SVD)
  long long row_off = 0, col_off = 0;   // block_qr: where the block's row / column index lists start (device lists)
};
constexpr int HH_BATCH_MAX_ROWS = 4096;
constexpr unsigned long long GEMM_TRACE_CAP = 1ull << 21;   // records
constexpr int GEMM_TRACE_WORDS = 10;                         // 64-bit words per record
// ``blks_dev``: the same descriptors already on the device (else they are uploaded here)
int hh_qr_batched(mpse_ctx* ctx, bool cplx, double* ws, double* q, HhParam* prm, const QrBlk* blks_host, int nblk,
                  bool form_q, const QrBlk* blks_dev = nullptr);
// Shifted Cholesky-QR of tall blocks on MFMA (mpse_cholqr.hip).  The blocks are factorised in place in their column-major
// workspaces and scattered to U / Vt like mpse_block_qr does; *ok = false when a block was rank deficient or too ill
// conditioned for the scheme (device flag, one read-back): the caller then runs the Householder path on fresh copies.
bool cholqr_eligible(const mpse_ctx* ctx, const QrBlk* blks, int nblk);
int cholqr_blocks(mpse_ctx* ctx, bool cplx, double* ws, const QrBlk* blks, int nblk, const long long* drows,
                  const long long* dcols, int herm, void* U, void* Vt, long long K, long long ncol, bool* ok);
// zero fill of two ranges in one launch (8-byte aligned)
int device_zero2(mpse_ctx* ctx, void* a, size_t abytes, void* b, size_t bbytes);
