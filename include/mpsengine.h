/*
 * mpsengine.h - C ABI of the MI355X (gfx950) matrix-product-state sweep engine.
 *
 * This is the drop-in boundary for the per-site hot path of Renormalizer's
 * DMRG (optimize_mps) and TDVP (Mps.evolve) loops.  Each entry point names the
 * reference call site(s) it replaces (paths relative to renormalizer/ in
 * shuaigroup/Renormalizer v0.0.11).  The reference reaches all of this
 * arithmetic through `xp.tensordot`, `opt_einsum` and SciPy LAPACK; here it is
 * hand-written HIP (FP64 MFMA contraction kernel, wavefront reductions,
 * Householder / one-sided Jacobi factorizations), device resident.
 *
 * Conventions
 *   - plain C types only; every function returns an int status (MPSE_OK == 0)
 *     and never throws; mpse_last_error(ctx) gives a message for the last failure;
 *   - `void*` tensor arguments are DEVICE pointers obtained from mpse_malloc
 *     unless the name ends in `_host`;
 *   - tensors are dense, C-order (last index fastest), dtype MPSE_F64 (8 B) or
 *     MPSE_C128 (interleaved re,im; 16 B);
 *   - index roles follow the reference: environment (bra bond, mpo bond, ket bond),
 *     mps site (D_l, d[, d_anc], D_r), mpo site (w_l, d_up, d_down, w_r);
 *   - all work is enqueued on the context's HIP stream; only the functions
 *     documented as synchronous (downloads, scalar results) wait for it;
 *   - one context per GPU per process (mirrors mps/backend.py:129-132).
 */
#ifndef MPSENGINE_H
#define MPSENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mpse_ctx mpse_ctx;

enum {
  MPSE_OK = 0,
  MPSE_ERR_OOM = 1,      /* device allocation failed (reference: MEMORY_ERRORS, mps/backend.py:89-94) */
  MPSE_ERR_SHAPE = 2,    /* inconsistent extents / "Invalid quantum number" (mps/svd_qn.py:219-220) */
  MPSE_ERR_NOCONV = 3,   /* iterative routine hit its iteration limit */
  MPSE_ERR_HIP = 4,      /* a HIP runtime call failed */
  MPSE_ERR_ARG = 5       /* bad argument (null pointer, unknown dtype, ...) */
};

enum { MPSE_F64 = 0, MPSE_C128 = 1 };
enum { MPSE_DOMAIN_L = 0, MPSE_DOMAIN_R = 1 };

/* ---------------------------------------------------------------- context */

/* Replaces backend selection / device binding, mps/backend.py:29-62 (RENO_GPU). */
int mpse_ctx_create(int device, mpse_ctx** out);
int mpse_ctx_destroy(mpse_ctx* ctx);
/* mps/backend.py:129-132 Backend.sync() */
int mpse_sync(mpse_ctx* ctx);
const char* mpse_last_error(const mpse_ctx* ctx);
const char* mpse_version(void);
/* device name, compute-unit count and the HIP stream handle (as void*) for callers that time with HIP events */
int mpse_device_info(mpse_ctx* ctx, char* name, size_t name_len, int* n_cu, void** stream);

/* ------------------------------------------------------------- profiling */
/* Optional per-launch timing of the contraction kernel with HIP events on the context stream
 * (no reference counterpart; used by bench.py for the roofline figures).  variant indexes the
 * operand types of mpse_gemm: 0 = f64 x f64, 1 = c128 x f64, 2 = f64 x c128, 3 = c128 x c128.
 * Totals cover the TIMED launches since the last reset; algorithmic flops use 2/4/4/8 per MAC.
 * on == 1 times every launch, on == N > 1 every N-th launch (sampling: two HIP events per timed launch
 * cost a few microseconds of stream time each). */
/* variants 4 and 5 of mpse_prof_get: 4 = the HBM-bound Lanczos vector kernels inside mpse_expm_lanczos (algorithmic
 * bytes in total_bytes), 5 = whole mpse_block_qr calls (Householder flops in total_flops).  The sampling counter is
 * shared by all variants.  mpse_prof_get_ktiles: 64 x 64 x 16 multiply-add blocks the timed contraction launches of a
 * variant (0-3) actually multiplied - with structural-zero skipping fewer than the dense count; the MFMA work issued
 * is ktiles x 65536 MACs x {2, 4, 4, 6} real flops (complex x complex uses three real products per complex one). */
int mpse_prof_get_ktiles(mpse_ctx* ctx, int variant, int64_t* ktiles);
/* variant 6 of mpse_prof_get = whole batched mpse_block_svd[_full] calls (one-sided Jacobi): total_bytes / total_flops are
 * the traffic and the arithmetic of sweeps x all column pairs (an upper bound: converged pairs are not rotated);
 * mpse_prof_get_svd_sweeps: Jacobi sweeps summed over the timed calls. */
int mpse_prof_get_svd_sweeps(mpse_ctx* ctx, int64_t* sweeps);
/* variant 7 of mpse_prof_get = the fused bond / two-level-site matvec launches (k_heff0_fused): total_flops are the
 * algorithmic flops of the matvec (dense formula of the contraction, no credit for skipped zero blocks), total_bytes the
 * operands read once plus the result written once. */
int mpse_prof_enable(mpse_ctx* ctx, int on);
int mpse_prof_reset(mpse_ctx* ctx);
int mpse_prof_get(mpse_ctx* ctx, int variant, double* total_ms, double* total_flops, double* total_bytes,
                  int64_t* launches);

/* --------------------------------------------------------- device memory */

/* Pooled device allocator (mps/backend.py:116-127 free_all_blocks/log_memory_usage). */
int mpse_malloc(mpse_ctx* ctx, size_t bytes, void** dptr);
int mpse_free(mpse_ctx* ctx, void* dptr);
int mpse_pool_trim(mpse_ctx* ctx);
int mpse_mem_info(mpse_ctx* ctx, size_t* pool_bytes, size_t* in_use_bytes, size_t* device_free, size_t* device_total);
/* mps/matrix.py:298-322 asnumpy/asxp. h2d/d2h are synchronous with respect to the host buffer. */
int mpse_memcpy_h2d(mpse_ctx* ctx, void* dst, const void* src_host, size_t bytes);
int mpse_memcpy_d2h(mpse_ctx* ctx, void* dst_host, const void* src, size_t bytes);
int mpse_memcpy_d2d(mpse_ctx* ctx, void* dst, const void* src, size_t bytes);
int mpse_memset_zero(mpse_ctx* ctx, void* dst, size_t bytes);
/* strided block copy (height rows of width_bytes; pitches in bytes): the sub-block assignments of
 * MatrixProduct.add (mps/mp.py:386-398) and dstack/vstack of the edge sites. */
int mpse_memcpy_2d(mpse_ctx* ctx, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width_bytes,
                   size_t height);

/* ------------------------------------------------------ vector primitives */
/* The Lanczos / Davidson vector algebra of lib/krylov/krylov.py:54-82 and
 * lib/davidson/davidson.py.  n counts elements of the given dtype.          */
int mpse_cast_f64_to_c128(mpse_ctx* ctx, void* dst, const void* src, int64_t n);
int mpse_conj_inplace(mpse_ctx* ctx, void* x, int64_t n);                       /* C128 only */
int mpse_scal(mpse_ctx* ctx, int dtype, void* x, int64_t n, double a_re, double a_im);
int mpse_axpy(mpse_ctx* ctx, int dtype, void* y, const void* x, int64_t n, double a_re, double a_im);
/* x_i *= m_i with real weights m (quantum-number mask of mps/gs.py:236-237, 520-523 kept as a dense 0/1 vector) */
int mpse_mul_real(mpse_ctx* ctx, int dtype, void* x, const void* m_f64, int64_t n);
/* Davidson preconditioner out = r / (hdiag - e + shift), zero where mask == 0 (mps/gs.py:530-531; mask may be NULL) */
int mpse_davidson_precond(mpse_ctx* ctx, int dtype, void* out, const void* r, const void* hdiag_f64,
                          const void* mask_f64, int64_t n, double e, double shift);
/* out_i = Re z_i */
int mpse_real_part(mpse_ctx* ctx, void* out_f64, const void* z_c128, int64_t n);
/* out_host[0..1] = sum conj(x_i) y_i  (xp.vdot); synchronous */
int mpse_dotc(mpse_ctx* ctx, int dtype, const void* x, const void* y, int64_t n, double* out_host);
/* out_host[0] = ||x||_2 (xp.linalg.norm); synchronous */
int mpse_nrm2(mpse_ctx* ctx, int dtype, const void* x, int64_t n, double* out_host);

/* out_host[0] = sqrt(mean_i |x_i|^2 / (atol + rtol max(|y1_i|, |y2_i|))^2): the error norm and the initial-step
 * norms of the embedded Runge-Kutta pairs that propagate single sites (ivp_solver = "RK45", mps/mps.py:1299-1315, and
 * the per-site integrations of TDVP-CMF, :1096-1265, where the reference calls scipy.integrate.solve_ivp).  Synchronous. */
int mpse_scaled_rms(mpse_ctx* ctx, int dtype, const void* x, const void* y1, const void* y2, int64_t n, double rtol,
                    double atol, double* out_host);

/* ------------------------------------------- deferred calls
 * A sweep knows what follows a local solve before the solve has converged: the QR of the new centre and the
 * environment update after a site step (mps/mps.py:1316-1378), the absorption of the bond factor into the next site
 * after a bond step (:1379-1395).  The host can only issue them once the solve has returned - and while the host
 * language gets there the GPU idles.  Between mpse_defer_begin and mpse_defer_end the calls mpse_gemm, mpse_block_qr
 * and mpse_env_update on this context are stored (scalar arguments and host index arrays copied, device pointers as
 * given - the buffers must exist) instead of executed; mpse_defer_arm(list) makes the next mpse_expm_lanczos run the
 * stored calls of that list, in order, right after it has enqueued the end of the solve, before it returns.  Two
 * lists (0, 1) so that the calls following the next-but-one solve can be recorded while one list waits.  Device blocks
 * freed while a list is open or waiting are released only after it has run.  Any other entry point called while a
 * list is being recorded executes at once, as usual (it must not depend on results of stored calls).
 * mpse_defer_run executes a list immediately; mpse_defer_discard drops everything (error paths). */
int mpse_defer_begin(mpse_ctx* ctx, int list);
int mpse_defer_end(mpse_ctx* ctx);
int mpse_defer_arm(mpse_ctx* ctx, int list);
int mpse_defer_run(mpse_ctx* ctx, int list);
int mpse_defer_discard(mpse_ctx* ctx);

/* ------------------------------------------- general tensor contraction */

/* A logical matrix index that addresses memory through up to two levels:
 *   offset(i) = (i / lo_ext) * s_hi + (i % lo_ext) * s_lo      (strides in elements)
 * Single-level indices use lo_ext >= ext (s_hi ignored). */
typedef struct {
  int64_t ext;
  int64_t lo_ext;
  int64_t s_hi;
  int64_t s_lo;
} mpse_index;

/* C[b](i,j) = alpha * sum_k opA(A[b](i,k)) * opB(B[b](k,j)) + beta * C[b](i,j)
 * Replaces mps/matrix.py:210-211 tensordot and :283-295 pair_tensor_contract
 * (transpose-copy + ?gemm in the reference) with one FP64-MFMA kernel that reads
 * the operands through their strides.  C is C128 if either operand is.       */
typedef struct {
  int dtype_a, dtype_b;
  int conj_a, conj_b;
  mpse_index m_a, k_a;      /* A(i,k) */
  mpse_index k_b, n_b;      /* B(k,j) */
  mpse_index m_c, n_c;      /* C(i,j) */
  int64_t batch;
  int64_t sb_a, sb_b, sb_c; /* batch strides, elements */
  double alpha_re, alpha_im;
  double beta_re, beta_im;
  int skip_zero_tiles;      /* hint for block-sparse operands: bit 0 scan A, bit 1 scan B for all-zero 64 x 16
                               tiles and skip them in the K loop (same result; pays from ~1e8 multiply-adds);
                               0 = plain dense GEMM */
} mpse_gemm_desc;

int mpse_gemm(mpse_ctx* ctx, const mpse_gemm_desc* desc, const void* A, const void* B, void* C);

/* out = transpose of `in` viewed as (d0,d1,d2) -> (d0,d2,d1); optional conjugation. */
int mpse_transpose_inner(mpse_ctx* ctx, int dtype, void* out, const void* in,
                         int64_t d0, int64_t d1, int64_t d2, int conj);

/* --------------------------------------------------- hot-path contractions */

/* Extents of a (one- or two-site) centre and its surroundings. */
typedef struct {
  int64_t Dl_bra, Dl_ket;   /* left bond of bra / ket (equal for an effective Hamiltonian used by Lanczos / Davidson;
                               mpse_heff_apply alone accepts bra != ket: H C projected onto another state's bonds) */
  int64_t Dr_bra, Dr_ket;
  int64_t d0, d1;           /* physical dims of the centre site(s); d1 unused for 0/1-site */
  int64_t danc;             /* ancilla dim of an MPDM site, 1 for an MPS */
  int64_t wl, wm, wr;       /* mpo bonds: left, middle (2-site only), right */
  int64_t env_unit;         /* mpse_env_update only: 1-based MPO-bond channel b with env[:, b, :] == identity
                               (see mpse_env_unit_channel); 0 = none / unknown */
  int64_t danc1;            /* ancilla dim of the second site of a two-site MPDM centre; 0 = same as danc */
} mpse_dims;

/* Environment update, replaces mps/lib.py:169-250 contract_one_site
 * (L: abc,adf->bcdf; bcdf,bdeg->cfeg; cfeg,ceh->fgh   R: fda,abc->fdbc; fdbc,gdeb->fcge; fcge,hec->fgh,
 * ancilla variants lib.py:207-211/239-243).
 *   env : (Dl_bra,wl,Dl_ket) for L, (Dr_bra,wr,Dr_ket) for R;  ket : (Dl_ket,d0[,danc],Dr_ket)
 *   bra : same layout with the bra extents, or NULL to use ket; bra_conj!=0 -> conjugate it here
 *         (pass 0 when the buffer already holds the conjugated tensor, like the reference's ms_conj)
 *   W   : (wl,d0,d0,wr), dtype w_dtype
 *   out : (Dr_bra,wr,Dr_ket) for L, (Dl_bra,wl,Dl_ket) for R; dtype `dtype`
 * env_dtype may be MPSE_F64 for the all-ones sentinel (lib.py:25). */
int mpse_env_update(mpse_ctx* ctx, int dtype, int domain, const mpse_dims* dims,
                    const void* env, int env_dtype, const void* ket, const void* bra, int bra_conj,
                    const void* W, int w_dtype, void* out);

/* Environment update through a stack of n_mpo MPO sites, replaces mps/lib.py:121-166 contract_one_site_multi_mpo
 * (used by optimize_mps(omega=...), mps/gs.py:106-112, for the (H - omega)^2 functional: Environ(mps, [mpo, mpo])).
 *   env : (D_bra, w_1, .., w_n, D_ket) - layer 1 touches the bra, layer n the ket;  W[i] : (wl[i], d0, d0, wr[i]), device
 *   out : (D'_bra, w'_1, .., w'_n, D'_ket);  dims->wl / wr / wm / env_unit are ignored.  1 <= n_mpo <= 4. */
int mpse_env_update_multi(mpse_ctx* ctx, int dtype, int domain, const mpse_dims* dims, int n_mpo,
                          const int64_t* wl, const int64_t* wr, const void* env, int env_dtype,
                          const void* ket, const void* bra, int bra_conj, const void* const* W, int w_dtype, void* out);

/* Finds an MPO-bond channel b of a square environment env (D, w, D) with max|env[:, b, :] - 1| <= tol: the
 * channel in which no operator has acted yet is the identity matrix when the sites behind it are canonical
 * (the reference contracts it like any other, mps/lib.py:200-205).  *unit_host = b + 1, or 0 if there is none.
 * The result is passed as mpse_dims.env_unit / mpse_heff.l_unit / r_unit so that the big GEMMs skip that channel.
 * Synchronous (one small read-back). */
int mpse_env_unit_channel(mpse_ctx* ctx, int dtype, const void* env, int64_t D, int64_t w, double tol,
                          int64_t* unit_host);

/* Effective Hamiltonian applied to the centre, replaces the closures built by
 * mps/hop_expr.py:57-115 (0-site abc,lbk,ck->al ; 1-site abc,bdef,lfk,cek->adl ;
 * 2-site abc,bdef,fghj,ljk,cehk->adgl ; ancilla variants), order (L.C).W.R.
 *   L (Dl,wl,Dl)  R (Dr,wr,Dr)  W0 (wl,d0,d0,wm|wr)  W1 (wm,d1,d1,wr)  C (Dl_ket,d0[,danc][,d1[,danc1]],Dr_ket), out the same with the bra bonds
 * nsite in {0,1,2}; for nsite==0 wl==wr is the shared mpo bond. */
typedef struct {
  int nsite;
  mpse_dims dims;
  const void* L; int l_dtype;
  const void* R; int r_dtype;
  const void* W0; const void* W1; int w_dtype;
  int64_t l_unit, r_unit;   /* 1-based channel along which L (resp. R) is the identity matrix, 0 = none:
                               that slice of the contraction is a copy instead of a GEMM */
} mpse_heff;

int mpse_heff_apply(mpse_ctx* ctx, int dtype, const mpse_heff* h, const void* C, void* out);

/* Optional: tells the engine the values of a real MPO site W (wl, d, d, wr) that lives at W_dev, from a host copy
 * (row major, the same numbers).  The engine keeps the block structure W[b, :, :, f] (which channel pairs are non-zero,
 * which are the identity) and, for large one-site centres on that site, absorbs the MPO step of mpse_heff_apply /
 * mpse_expm_lanczos / mpse_davidson into the operands of the two large products instead of running it as a step of
 * its own (same result: the reference's single expression mps/hop_expr.py:75-79).  The contents of W_dev must not
 * change while the hint stands: mpse_free(W_dev), a call with W_host == NULL, and every element-level entry point that
 * writes into the described range (mpse_memcpy_h2d / _d2d / _2d, mpse_memset_zero, mpse_scal, mpse_conj_inplace,
 * mpse_axpy) drop it; a caller that uses W_dev as the OUTPUT of a contraction has to drop it itself.
 * mpse_env_update on a described site with d >= 8 runs its MPO step as an elementwise pass over the site's non-zero
 * blocks as well.  No device work. */
int mpse_mpo_site_hint(mpse_ctx* ctx, const void* W_dev, const double* W_host_f64, int64_t wl, int64_t d, int64_t wr);

/* How many effective-Hamiltonian applications of this context ran as the single fused launch of mpse_heff0.hip (inside
 * mpse_expm_lanczos: bond matrices, mps/hop_expr.py:63-67, and one-site centres with a two-level physical index,
 * :75-79) instead of through the contraction plans.  Diagnostics; tests use it to see that the path they mean to check
 * is the one that ran.  Either pointer may be NULL. */
int mpse_heff_fused_stats(mpse_ctx* ctx, int64_t* bond_launches, int64_t* site_launches);

/* Two-layer effective Hamiltonian of the (H - omega)^2 functional, replaces the twolayer=True closures of
 * mps/hop_expr.py:24-52 (1-site abcd,befg,cfhi,jgik,aej->dhk ; 2-site abcd,befg,cfhi,gjkl,ikmn,olnp,aejo->dhmp).
 *   L (Dl, wl, wl, Dl), R (Dr, wr, wr, Dr); W0 / W1 serve both layers; no ancilla; nsite in {1, 2}; the unit-channel
 *   fields are ignored. */
int mpse_heff_apply2(mpse_ctx* ctx, int dtype, const mpse_heff* h, const void* C, void* out);

/* Lanczos exponential out = expm(dt*Heff) C, replaces lib/krylov/krylov.py:27-82
 * expm_krylov as called at mps/mps.py:1300-1303, 1343-1346, 1377-1380 (same
 * recurrence without re-orthogonalisation, same stopping rule: successive
 * approximations allclose(rtol,atol) on even j > 3, breakdown at beta < 100 n eps).
 * Synchronous; *nvec receives the Krylov dimension. */
/* Optional, for the NEXT mpse_expm_lanczos on this context only: the tile-occupancy pattern of the centre tensor as
 * the sweep knows it from the quantum numbers (mps/mp.py:308-352: entry (a, sigma, b) can be non-zero only where the
 * bond and physical quantum numbers add up to the total) - the same for every Krylov vector of the solve, so the
 * engine need not scan each vector for empty tiles before multiplying it by the left environment.  Layout: for the
 * centre tensor viewed as the matrix C[a, (sigma.., b)] (Dl rows, N columns): byte [tn * nkw * 8 + kt] is 1 if any
 * entry with a in [16 kt, 16 kt + 16) and column in [64 tn, 64 tn + 64) may be non-zero, nkw = ceil(ceil(Dl / 16) / 8),
 * tn < ceil(N / 64); nbytes = ceil(N / 64) * nkw * 8.  A mask of another size is ignored.  The mask must mark every
 * tile that holds a non-zero (a superset is fine, a missing tile drops its contribution). */
int mpse_expm_centre_mask(mpse_ctx* ctx, const void* mask_dev, int64_t nbytes);

int mpse_expm_lanczos(mpse_ctx* ctx, int dtype, const mpse_heff* h, double dt_re, double dt_im,
                      const void* C, void* out, double rtol, double atol, int max_dim, int* nvec);

/* Davidson eigensolver for the lowest nroots eigenpairs of the effective Hamiltonian, replaces
 * lib/davidson/davidson.py:154-441 as called at mps/gs.py:533-538 (diagonal preconditioner r / (hdiag - e + shift),
 * Gram-Schmidt twice, restart from the Ritz vectors when max_space vectors are held, convergence of a root when
 * |de| < tol and |r| < sqrt(tol), new directions dropped when their squared norm falls under lindep; tol < 0: the
 * residual alone decides, |r| < -tol - the convergence test of the reference's algo = "primme", gs.py:552-569).
 *   h         : the projected operator (mpse_heff_apply; twolayer != 0: mpse_heff_apply2, the (H - omega)^2 form)
 *   hdiag_f64 : its diagonal (n doubles, device);  mask_f64: 0/1 weights of the symmetry-allowed entries or NULL
 *               (the reference compresses vectors to those entries on the host, mps/gs.py:260, 520-523)
 *   guess     : nguess start vectors of n elements each, contiguous (device);  x_out: nroots vectors (device)
 *   e_host    : nroots eigenvalues;  max_space <= 0 selects 12 + 3 (nroots - 1), max_cycle <= 0 selects 100
 * The subspace matrix grows by one batched reduction per new vector; subspace eigenproblems run on the host.
 * Synchronous. */
int mpse_davidson(mpse_ctx* ctx, int dtype, const mpse_heff* h, int twolayer, const void* hdiag_f64,
                  const void* mask_f64, int nroots, int nguess, const void* guess, double tol, int max_cycle,
                  int max_space, double lindep, double shift, double* e_host, void* x_out, int* ncycle, int* nmatvec);

/* Which renormalised basis states to keep, replaces select_basis of mps/lib.py:253-322 (the index selection; the
 * column copies are mpse_gather_cols / mpse_gather_rows): an equal quota int(m_max * percent / nblocks) per
 * quantum-number block (ascending block id, descending weight inside a block), the remaining slots by descending
 * weight; stable on ties.  block_id_host[i] = rank of state i's quantum number among the distinct ones (may be NULL
 * when percent == 0).  picked_host receives min(count, m_max) indices.  Host-side integer logic, no device work. */
int mpse_truncate_select(const double* sigma_host, const int64_t* block_id_host, int64_t count, int64_t m_max,
                         double percent, int64_t* picked_host, int64_t* npicked);

/* ------------------------------------------------ block decompositions */

/* Quantum-number blocked QR (system 'L') / RQ (system 'R') of the centre matrix
 * coef (nrow x ncol), replaces mps/svd_qn.py:99-227 with QR=True, full_matrices=False
 * (scipy.linalg.qr / rq per block + blockrecover).  The integer bookkeeping stays
 * with the caller: block b owns rows row_idx_host[row_off_host[b]..row_off_host[b+1])
 * and columns col_idx_host[col_off_host[b]..col_off_host[b+1]); it contributes
 * min(rows,cols) columns.  Outputs (device, zero outside the blocks):
 *   U  (nrow x K)  and  Vt (K x ncol),  K = sum_b min(m_b,n_b),  coef == U @ Vt on the blocks;
 *   system 'L': U has orthonormal columns;  system 'R': Vt has orthonormal rows.
 * system_is_R: bit 0 = system 'R'; bit 1 = this decomposition by the Householder kernels whatever the context's scheme
 * (mpse_block_qr_scheme) says - for a caller that knows the blocks to be rank deficient (a sweep that has seen this site
 * break the Cholesky-QR path before). */
int mpse_block_qr(mpse_ctx* ctx, int dtype, const void* coef, int64_t nrow, int64_t ncol,
                  int nblocks, const int64_t* row_idx_host, const int64_t* row_off_host,
                  const int64_t* col_idx_host, const int64_t* col_off_host,
                  int system_is_R, void* U, void* Vt, int64_t K);

/* How many mpse_block_qr decompositions this context has run, how many of them went through the Cholesky-QR kernels
 * (tall blocks of up to 256 columns: three Gram / Cholesky / triangular-solve passes on MFMA, all compute units) and how
 * many of those were redone by the Householder kernels because a block was rank deficient or too ill conditioned for
 * the scheme (device-side flag).  No reference counterpart (diagnostics; bench.py reports the rates).  Any pointer may
 * be NULL. */
int mpse_block_qr_stats(mpse_ctx* ctx, int64_t* calls, int64_t* chol_calls, int64_t* chol_fallbacks);

/* Optimistic mode of the Cholesky-QR path, for callers that can repeat a whole step (the TDVP-PS sweep, whose input
 * state stays untouched until the step returns): while it is on, mpse_block_qr does not read the breakdown flag back
 * after each decomposition - the read-back stalls the host exactly where it should be enqueueing the next local solve -
 * a breakdown raises a sticky device word instead and the results of that decomposition are NOT an isometry.
 * mpse_block_qr_check (synchronous) says whether any decomposition since the mode was switched on broke down: the caller
 * then discards the step and repeats it with the mode off (every decomposition verified, Householder where needed).
 * Switching the mode on OR off clears the word - ask before switching off - and while the mode is off
 * mpse_block_qr_check reports 0 without touching the device.  No reference counterpart. */
int mpse_block_qr_optimistic(mpse_ctx* ctx, int on);
int mpse_block_qr_check(mpse_ctx* ctx, int* tripped);

/* Pass counts of the Cholesky-QR path since the context was created (synchronous read of two device counters): quantum-
 * number blocks factorised by those kernels, and how many of them ended after TWO passes - the kernels decide per block,
 * on the device, whether the Gram matrix of the second pass is close enough to the identity for its factor to leave an
 * isometry to rounding (n max|G - I| <= 0.1); the third pass of such a block is skipped.  No reference counterpart
 * (diagnostics; bench.py reports the rate).  Either pointer may be NULL. */
int mpse_block_qr_pass_stats(mpse_ctx* ctx, int64_t* blocks, int64_t* two_pass);

/* Which kernels mpse_block_qr uses on this context: 0 Householder only (the column-by-column elimination of LAPACK's
 * geqrf, mps/svd_qn.py:171-185 calls scipy.linalg.qr: the SAME isometry as the reference up to rounding, also in the
 * directions of numerically zero singular values, where a QR factorisation is not unique), 1 Cholesky-QR for tall blocks
 * from 256 rows on (default), 2 Cholesky-QR for every block shape it supports, -1 back to the MPSE_CHOLQR environment
 * setting.  Both schemes return U @ Vt == coef and an isometry to rounding; they may differ by a rotation inside the
 * numerical null space of a block.  No reference counterpart. */
int mpse_block_qr_scheme(mpse_ctx* ctx, int scheme);

/* Quantum-number blocked economic SVD by one-sided Jacobi, replaces mps/svd_qn.py:99-240
 * with QR=False, full_matrices=False (scipy.linalg.svd gesdd per block).  Same block
 * description; outputs U (nrow x K), Vt (K x ncol) (block order, NOT globally sorted) and
 * the singular values S_host[K] (descending inside each block).  Synchronous. */
int mpse_block_svd(mpse_ctx* ctx, int dtype, const void* coef, int64_t nrow, int64_t ncol,
                   int nblocks, const int64_t* row_idx_host, const int64_t* row_off_host,
                   const int64_t* col_idx_host, const int64_t* col_off_host,
                   void* U, void* Vt, double* S_host, int64_t K);

/* full_matrices=True variant (mps/svd_qn.py:65-86, 187-213 as called by MatrixProduct._update_mps, mps/mp.py:693-695):
 * after the K singular triplets, block b contributes extra_host[b] null-space vectors of its taller side (zero
 * singular value; 0 <= extra <= |m_b - n_b|): extra columns of U (nrow x KU) for m_b >= n_b, extra rows of
 * Vt (KV x ncol) otherwise, appended after column/row K in block order.  extra_host == NULL means none. */
int mpse_block_svd_full(mpse_ctx* ctx, int dtype, const void* coef, int64_t nrow, int64_t ncol,
                        int nblocks, const int64_t* row_idx_host, const int64_t* row_off_host,
                        const int64_t* col_idx_host, const int64_t* col_off_host, const int64_t* extra_host,
                        void* U, int64_t KU, void* Vt, int64_t KV, double* S_host, int64_t K);

/* out[:, j] = in[:, cols_host[j]] * scale_host[j]  (column gather of a row-major matrix,
 * replaces the per-column copies of mps/lib.py:303-316 select_basis; scale_host may be NULL). */
int mpse_gather_cols(mpse_ctx* ctx, int dtype, void* out, const void* in, int64_t nrow, int64_t ncol_in,
                     const int64_t* cols_host, const double* scale_host, int64_t ncol_out);
/* out[i, :] = in[rows_host[i], :] * scale_host[i] */
int mpse_gather_rows(mpse_ctx* ctx, int dtype, void* out, const void* in, int64_t ncol,
                     const int64_t* rows_host, const double* scale_host, int64_t nrow_out);

#ifdef __cplusplus
}
#endif
#endif /* MPSENGINE_H */
