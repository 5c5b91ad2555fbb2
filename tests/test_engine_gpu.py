"""GPU parity tests of the C-ABI primitives (libmpsengine.so through ctypes) against the
reference-pinned oracle and the golden vectors.  Run on the MI355X box: pytest -m gpu."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import mps_oracle as orc
from renormalizer_amd import engine as E
from renormalizer_amd.mps.hop_expr import hop_expr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    return E.get_engine()


_FUSED_ON = os.environ.get("MPSE_HEFF0", "1") != "0" and os.environ.get("MPSE_LANCZOS_ASYNC", "1") != "0"


def _rand(rng, shape, cplx):
    a = rng.standard_normal(shape)
    return a + 1j * rng.standard_normal(shape) if cplx else a


def _relerr(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(1e-300, np.abs(b).max()))


# ------------------------------------------------------------------ gemm

@pytest.mark.parametrize("ca,cb", [(False, False), (True, False), (False, True), (True, True)])
def test_gemm_dtypes_and_edges(eng, ca, cb):
    rng = np.random.default_rng(1)
    for (M, N, K) in [(1, 1, 1), (3, 5, 2), (16, 16, 4), (64, 64, 16), (65, 63, 17), (130, 70, 33), (200, 31, 129)]:
        a = _rand(rng, (M, K), ca)
        b = _rand(rng, (K, N), cb)
        A, B = eng.asdevice(a), eng.asdevice(b)
        out = eng.matmul(A, B).to_host()
        assert _relerr(out, a @ b) < 1e-13, (M, N, K)
        # transposed / conjugated operand views through strides
        At, Bt = eng.asdevice(np.ascontiguousarray(a.T)), eng.asdevice(np.ascontiguousarray(b.T))
        out = eng.matmul(At, Bt, trans_a=True, trans_b=True, conj_a=True, conj_b=True).to_host()
        assert _relerr(out, a.conj() @ b.conj()) < 1e-13, (M, N, K)


@pytest.mark.parametrize("ca,cb", [(False, False), (True, False), (True, True)])
def test_gemm_split_k(eng, ca, cb):
    """Skinny outputs with long K take the split-K path (partials + fixed-order reduction)."""
    rng = np.random.default_rng(12)
    for (M, N, K) in [(100, 90, 3000), (256, 256, 1280), (7, 300, 1000), (64, 64, 64)]:
        a, b = _rand(rng, (M, K), ca), _rand(rng, (K, N), cb)
        c0 = _rand(rng, (M, N), ca or cb)
        A, B, Cd = eng.asdevice(a), eng.asdevice(b), eng.asdevice(c0)
        eng.gemm(A, B, Cd, E.idx1(M, K), E.idx1(K, 1), E.idx1(K, N), E.idx1(N, 1), E.idx1(M, N), E.idx1(N, 1),
                 alpha=0.7, beta=-1.25)
        ref = 0.7 * (a @ b) - 1.25 * c0
        assert _relerr(Cd.to_host(), ref) < 1e-13, (M, N, K)
        r1 = eng.matmul(A, B).to_host()
        r2 = eng.matmul(A, B).to_host()
        assert np.array_equal(r1, r2)


def test_gemm_is_asymmetric_safe(eng):
    """A = I against an asymmetric B catches transposed operand / output maps."""
    n = 48
    b = np.arange(n * n, dtype=float).reshape(n, n) + 1j * np.arange(n * n)[::-1].reshape(n, n)
    out = eng.matmul(eng.asdevice(np.eye(n)), eng.asdevice(b)).to_host()
    assert np.array_equal(out, b)
    out = eng.matmul(eng.asdevice(b), eng.asdevice(np.eye(n))).to_host()
    assert np.array_equal(out, b)


def test_gemm_composite_batch_alpha_beta(eng):
    rng = np.random.default_rng(2)
    nb, d0, d1, K, N = 3, 4, 5, 7, 6
    # A[b, i0, k, i1] (composite row index (i0 | i1) around k), B[b, k, n]
    a = _rand(rng, (nb, d0, K, d1), True)
    b = _rand(rng, (nb, K, N), False)
    c0 = _rand(rng, (nb, d0 * d1, N), True)
    A, B, Cd = eng.asdevice(a), eng.asdevice(b), eng.asdevice(c0)
    eng.gemm(A, B, Cd, E.idx2(d0, d1, K * d1, 1), E.idx1(K, d1), E.idx1(K, N), E.idx1(N, 1),
             E.idx1(d0 * d1, N), E.idx1(N, 1), batch=nb, sb_a=d0 * K * d1, sb_b=K * N, sb_c=d0 * d1 * N,
             alpha=0.5 - 2j, beta=1.5 + 0.25j)
    ref = (0.5 - 2j) * np.einsum("bikj,bkn->bijn", a, b).reshape(nb, d0 * d1, N) + (1.5 + 0.25j) * c0
    assert _relerr(Cd.to_host(), ref) < 1e-13


def test_gemm_bitwise_reproducible(eng):
    rng = np.random.default_rng(3)
    a, b = _rand(rng, (150, 300), True), _rand(rng, (300, 90), True)
    A, B = eng.asdevice(a), eng.asdevice(b)
    r1 = eng.matmul(A, B).to_host()
    r2 = eng.matmul(A, B).to_host()
    assert np.array_equal(r1, r2)


def test_transpose_inner(eng):
    rng = np.random.default_rng(4)
    for cplx in (False, True):
        a = _rand(rng, (3, 37, 45), cplx)
        A = eng.asdevice(a)
        out = eng.empty((3, 45, 37), a.dtype)
        eng._check(eng.lib.mpse_transpose_inner(eng.ctx, A.code, out.ptr, A.ptr, 3, 37, 45, 1))
        assert np.array_equal(out.to_host(), a.transpose(0, 2, 1).conj())


# ---------------------------------------------------------- vector algebra

def test_vector_ops(eng):
    rng = np.random.default_rng(5)
    for cplx in (False, True):
        for n in (1, 63, 1000, 300001):
            x, y = _rand(rng, n, cplx), _rand(rng, n, cplx)
            X, Y = eng.asdevice(x), eng.asdevice(y)
            assert abs(X.vdot(Y) - np.vdot(x, y)) < 1e-12 * n
            assert abs(X.norm() - np.linalg.norm(x)) < 1e-13 * np.sqrt(n) * 10
            a = (0.3 - 1.2j) if cplx else -0.7
            a_ = complex(a)
            eng._check(eng.lib.mpse_axpy(eng.ctx, X.code, Y.ptr, X.ptr, n, a_.real, a_.imag))
            assert _relerr(Y.to_host(), y + a * x) < 1e-14
            X.scale_(a)
            assert _relerr(X.to_host(), a * x) < 1e-14
    r = rng.standard_normal(1000)
    assert np.array_equal(eng.asdevice(r).to_complex().to_host(), r.astype(complex))
    z = _rand(rng, 1000, True)
    assert np.array_equal(eng.asdevice(z).conj().to_host(), z.conj())


# ------------------------------------------------- hot-path contractions

def _dims(ket, mo, bra=None):
    d = E.mpse_dims()
    bra = ket if bra is None else bra
    d.Dl_ket, d.Dr_ket = ket.shape[0], ket.shape[-1]
    d.Dl_bra, d.Dr_bra = bra.shape[0], bra.shape[-1]
    d.d0 = ket.shape[1]
    d.d1 = 1
    d.danc = ket.shape[2] if ket.ndim == 4 else 1
    d.wl, d.wr, d.wm = mo.shape[0], mo.shape[3], 1
    return d


def dev_env_update(eng, env, ket, mo, dom, bra=None, bra_conj=True):
    cplx = np.iscomplexobj(ket) or np.iscomplexobj(env) or (bra is not None and np.iscomplexobj(bra))
    wdt = np.complex128 if cplx else np.float64
    K = eng.asdevice(ket, wdt)
    Bt = None if bra is None else eng.asdevice(bra, wdt)
    Ev, W = eng.asdevice(env), eng.asdevice(mo)
    d = _dims(ket, mo, bra)
    oshape = (d.Dr_bra, d.wr, d.Dr_ket) if dom == "L" else (d.Dl_bra, d.wl, d.Dl_ket)
    out = eng.empty(oshape, wdt)
    eng._check(eng.lib.mpse_env_update(eng.ctx, out.code, 0 if dom == "L" else 1, C.byref(d), Ev.ptr, Ev.code, K.ptr,
                                       None if Bt is None else Bt.ptr, int(bra_conj), W.ptr, W.code, out.ptr))
    return out.to_host()


def make_heff(eng, l, r, cmo, cshape):
    """returns (mpse_heff, keepalive)"""
    ns = len(cmo)
    h = E.mpse_heff()
    h.nsite = ns
    d = h.dims
    d.Dl_ket, d.Dr_ket = cshape[0], cshape[-1]
    d.Dl_bra, d.Dr_bra = l.shape[0], r.shape[0]
    d.danc = cshape[2] if (ns >= 1 and len(cshape) == 2 * ns + 2) else 1
    d.danc1 = cshape[4] if (ns == 2 and len(cshape) == 6) else 0
    d.wl, d.wr = l.shape[1], r.shape[1]
    d.d0 = cmo[0].shape[1] if ns >= 1 else 1
    d.d1 = cmo[1].shape[1] if ns == 2 else 1
    d.wm = cmo[0].shape[3] if ns == 2 else 1
    keep = [eng.asdevice(l), eng.asdevice(r)] + [eng.asdevice(w) for w in cmo]
    h.L, h.l_dtype, h.R, h.r_dtype = keep[0].ptr, keep[0].code, keep[1].ptr, keep[1].code
    if ns >= 1:
        h.W0, h.w_dtype = keep[2].ptr, keep[2].code
    if ns == 2:
        h.W1 = keep[3].ptr
    return h, keep


def dev_heff_apply(eng, l, r, cmo, c):
    h, keep = make_heff(eng, l, r, cmo, c.shape)
    Cd = eng.asdevice(c)
    out = eng.empty((l.shape[0],) + c.shape[1:-1] + (r.shape[0],), c.dtype)
    eng._check(eng.lib.mpse_heff_apply(eng.ctx, Cd.code, C.byref(h), Cd.ptr, out.ptr))
    return out.to_host()


def test_env_update_golden(eng, golden_dir):
    z = np.load(os.path.join(golden_dir, "seams.npz"))
    for k in range(int(z["c1s_n"])):
        g = lambda n: z[f"c1s_{k}_{n}"]
        dom = str(g("dom"))
        assert _relerr(dev_env_update(eng, g("env"), g("ms"), g("mo"), dom, bra=g("bra")), g("out")) < 1e-12
        assert _relerr(dev_env_update(eng, g("env"), g("ms"), g("mo"), dom), g("out_self")) < 1e-12


@pytest.mark.parametrize("dom", ["L", "R"])
def test_env_update_described_site_elementwise_mpo_step(eng, dom):
    """Environment update behind a described MPO site (mpse_mpo_site_hint) with a large physical index: the MPO step runs
    as the elementwise pass of the folded matvec (mpse_plans.h push_env_wmix) instead of a batched real x complex product.
    Holstein-like site (identity, diagonal and tridiagonal blocks; a channel that receives nothing; five planes on the
    R side = two passes), block-sparse environment with a unit channel, bra = ket and a separate bra, against the oracle
    and against the same call without the description (the product path)."""
    rng = np.random.default_rng(41)
    D, d = 128, 16
    w = _holstein_like_site(rng, d)                      # (5, d, d, 4)
    wl, wr = w.shape[0], w.shape[3]
    ket = _rand(rng, (D, d, D), True)
    bra = _rand(rng, (D, d, D), True)
    we = wl if dom == "L" else wr
    env = _rand(rng, (D, we, D), True)
    env[:, 0, :] = np.eye(D)
    for other in (None, bra):
        ref = orc.contract_one_site(env, ket, w, dom, ms_conj=None if other is None else other.conj())
        K, Ev, W = eng.asdevice(ket), eng.asdevice(env), eng.asdevice(w)
        Bt = None if other is None else eng.asdevice(other)
        dd = _dims(ket, w, other)
        oshape = (dd.Dr_bra, dd.wr, dd.Dr_ket) if dom == "L" else (dd.Dl_bra, dd.wl, dd.Dl_ket)
        outs = []
        for described in (False, True):
            if described:
                eng.mpo_site_hint(W, w)
            out = eng.empty(oshape, np.complex128)
            eng._check(eng.lib.mpse_env_update(eng.ctx, out.code, 0 if dom == "L" else 1, C.byref(dd), Ev.ptr, Ev.code, K.ptr,
                                               None if Bt is None else Bt.ptr, 1, W.ptr, W.code, out.ptr))
            outs.append(out.to_host())
            assert _relerr(outs[-1], ref) < 1e-12, (dom, other is None, described)
        assert _relerr(outs[1], outs[0]) < 1e-13


def test_heff_apply_golden(eng, golden_dir):
    z = np.load(os.path.join(golden_dir, "seams.npz"))
    for k in range(int(z["hop_n"])):
        g = lambda n: z[f"hop_{k}_{n}"]
        cmo = [g(f"w{j}") for j in range(int(g("nsite")))]
        assert _relerr(dev_heff_apply(eng, g("l"), g("r"), cmo, g("c")), g("out")) < 1e-12


@pytest.mark.parametrize("cplx", [False, True])
def test_heff_rectangular_and_unequal_ancillas(eng, cplx):
    """bra bonds != ket bonds (H C projected onto another state's bond spaces, variational compression) and two-site
    density-operator centres whose sites differ in size (ancilla legs of different dimension), against einsum;
    the Krylov driver refuses a rectangular operator"""
    rng = np.random.default_rng(19)
    for (Dlb, Dlk, Drb, Drk, d0, a0, d1, a1, wl, wm, wr) in ((20, 8, 36, 10, 4, 4, 2, 2, 5, 4, 5), (7, 7, 9, 9, 2, 2, 4, 4, 3, 4, 3),
                                                            (70, 33, 20, 65, 3, 1, 5, 1, 4, 3, 2)):
        l, r = _rand(rng, (Dlb, wl, Dlk), cplx), _rand(rng, (Drb, wr, Drk), cplx)
        w0, w1 = _rand(rng, (wl, d0, d0, wm), False), _rand(rng, (wm, d1, d1, wr), False)
        if a0 == 1:
            c = _rand(rng, (Dlk, d0, d1, Drk), cplx)
            ref = np.einsum("abc,bdef,fghj,ljk,cehk->adgl", l, w0, w1, r, c, optimize=True)
        else:
            c = _rand(rng, (Dlk, d0, a0, d1, a1, Drk), cplx)
            ref = np.einsum("abc,bdef,fghj,ljk,cemhnk->admgnl", l, w0, w1, r, c, optimize=True)
        assert _relerr(dev_heff_apply(eng, l, r, [w0, w1], c), ref) < 1e-12
        w = _rand(rng, (wl, d0, d0, wr), False)
        c = _rand(rng, (Dlk, d0, Drk) if a0 == 1 else (Dlk, d0, a0, Drk), cplx)
        ref = np.einsum("abc,bdef,lfk,cek->adl" if a0 == 1 else "abc,bdef,lfk,cegk->adgl", l, w, r, c, optimize=True)
        assert _relerr(dev_heff_apply(eng, l, r, [w], c), ref) < 1e-12
    h, keep = make_heff(eng, _rand(rng, (5, 2, 4), True), _rand(rng, (4, 2, 4), True), [_rand(rng, (2, 2, 2, 2), False)], (4, 2, 4))
    v = eng.asdevice(_rand(rng, (4, 2, 4), True))
    out = eng.empty((4, 2, 4), np.complex128)
    nv = C.c_int()
    st = eng.lib.mpse_expm_lanczos(eng.ctx, v.code, C.byref(h), 0.0, -0.1, v.ptr, out.ptr, 1e-5, 1e-8, 0, C.byref(nv))
    assert st == E.MPSE_ERR_SHAPE


@pytest.mark.parametrize("cplx", [False, True])
def test_heff_small_centres_one_launch(eng, cplx):
    """Centres of up to 32 768 elements take the one-launch matvec (mpse_small.hip: row of L per workgroup, MPO step
    as a sparse list in LDS, transposed right environment): every thread mapping of the kernel against the oracle -
    columns > / = / < 256 threads (K groups of the first step), bonds that do not divide 256, d = 1 ... 16, sparse and
    dense MPO sites, the 0-site matvec, bitwise repeatability; and through the Krylov solve, where the result arrives
    as slices of the ket bond that the Lanczos update adds (same Krylov dimension and vector as the oracle's)."""
    rng = np.random.default_rng(23)
    shapes = [(32, 8, 32, 5, 5), (64, 8, 64, 3, 3), (32, 4, 32, 5, 4), (7, 3, 5, 2, 3), (48, 2, 40, 4, 4), (1, 2, 6, 1, 3),
              (6, 2, 1, 3, 1), (20, 16, 24, 5, 5), (33, 5, 100, 8, 7), (128, 2, 128, 5, 5), (3, 1, 3, 2, 2)]
    for (Dl, d, Dr, wl, wr) in shapes:
        l, r = _rand(rng, (Dl, wl, Dl), cplx), _rand(rng, (Dr, wr, Dr), cplx)
        w = _rand(rng, (wl, d, d, wr), False)
        c = _rand(rng, (Dl, d, Dr), cplx)
        out = dev_heff_apply(eng, l, r, [w], c)
        assert _relerr(out, orc.hop_apply(l, r, [w], c)) < 1e-12, (Dl, d, Dr, wl, wr)
        assert np.array_equal(out, dev_heff_apply(eng, l, r, [w], c))
        # a sum-of-products site: identity blocks and a few operator blocks, most of W zero
        ws = np.zeros((wl, d, d, wr))
        for b in range(wl):
            ws[b, :, :, min(b, wr - 1)] = np.eye(d)
        ws[0, :, :, wr - 1] += np.diag(np.arange(d, dtype=float))
        if d > 1:
            ws[wl - 1, :, :, 0] = np.diag(np.sqrt(np.arange(1, d)), 1)
        assert _relerr(dev_heff_apply(eng, l, r, [ws], c), orc.hop_apply(l, r, [ws], c)) < 1e-12, (Dl, d, Dr, wl, wr)
        # 0-site
        r0 = _rand(rng, (Dr, wl, Dr), cplx)
        s0 = _rand(rng, (Dl, Dr), cplx)
        assert _relerr(dev_heff_apply(eng, l, r0, [], s0), orc.hop_apply(l, r0, [], s0)) < 1e-12, (Dl, Dr, wl)
    # Krylov solves on Hermitian parts (complex: real-time step; real: imaginary-time step)
    for (Dl, d, Dr, wl) in [(32, 8, 32, 5), (64, 8, 64, 3), (32, 4, 32, 4), (64, 2, 64, 5), (24, 3, 40, 3)]:
        l, r = _rand(rng, (Dl, wl, Dl), cplx), _rand(rng, (Dr, wl, Dr), cplx)
        l = (l + l.transpose(2, 1, 0).conj()) / (4 * Dl)
        r = (r + r.transpose(2, 1, 0).conj()) / (4 * Dr)
        w = _rand(rng, (wl, d, d, wl), False)
        w = (w + w.transpose(0, 2, 1, 3)) / 2
        c = _rand(rng, (Dl, d, Dr), cplx)
        c /= np.linalg.norm(c)
        dt = -0.4j if cplx else -0.4
        ref, nref = orc.expm_krylov(lambda y: orc.hop_apply(l, r, [w], y.reshape(c.shape)).ravel(), dt, c.ravel())
        out, nv = dev_expm(eng, l, r, [w], c, dt)
        assert nv == nref and _relerr(out.ravel(), ref) < 1e-10, (Dl, d, Dr, wl)
        s0 = _rand(rng, (Dl, Dr), cplx)
        s0 /= np.linalg.norm(s0)
        ref, nref = orc.expm_krylov(lambda y: orc.hop_apply(l, r, [], y.reshape(s0.shape)).ravel(), dt, s0.ravel())
        out, nv = dev_expm(eng, l, r, [], s0, dt)
        assert nv == nref and _relerr(out.ravel(), ref) < 1e-10, (Dl, Dr, wl)


@pytest.mark.parametrize("cplx", [False, True])
def test_contractions_midsize_vs_oracle(eng, cplx):
    """Shapes that cross tile boundaries (not multiples of 64/16) against the oracle."""
    rng = np.random.default_rng(7)
    Dl, Dr, d, wl, wr = 70, 45, 6, 5, 4
    l, r = _rand(rng, (Dl, wl, Dl), cplx), _rand(rng, (Dr, wr, Dr), cplx)
    w = _rand(rng, (wl, d, d, wr), False)
    c = _rand(rng, (Dl, d, Dr), cplx)
    assert _relerr(dev_heff_apply(eng, l, r, [w], c), orc.hop_apply(l, r, [w], c)) < 1e-12
    r0 = _rand(rng, (Dr, wl, Dr), cplx)
    s = _rand(rng, (Dl, Dr), cplx)
    assert _relerr(dev_heff_apply(eng, l, r0, [], s), orc.hop_apply(l, r0, [], s)) < 1e-12
    w2 = _rand(rng, (wr, 3, 3, 2), False)
    r2 = _rand(rng, (Dr, 2, Dr), cplx)
    c2 = _rand(rng, (Dl, d, 3, Dr), cplx)
    assert _relerr(dev_heff_apply(eng, l, r2, [w, w2], c2), orc.hop_apply(l, r2, [w, w2], c2)) < 1e-12
    ket = _rand(rng, (Dl, d, Dr), cplx)
    assert _relerr(dev_env_update(eng, l, ket, w, "L"), orc.contract_one_site(l, ket, w, "L")) < 1e-12
    assert _relerr(dev_env_update(eng, r, ket, w, "R"), orc.contract_one_site(r, ket, w, "R")) < 1e-12
    # hermiticity of the effective Hamiltonian built from Hermitian parts: <x|H y> == <H x|y>
    lh = l + l.transpose(2, 1, 0).conj()
    rh = r + r.transpose(2, 1, 0).conj()
    wh = w + w.transpose(0, 2, 1, 3)
    x, y = _rand(rng, (Dl, d, Dr), cplx), _rand(rng, (Dl, d, Dr), cplx)
    hx, hy = dev_heff_apply(eng, lh, rh, [wh], x), dev_heff_apply(eng, lh, rh, [wh], y)
    assert abs(np.vdot(x, hy) - np.vdot(hx, y)) < 1e-9 * abs(np.vdot(x, hy))


# ---------------------------------------------------------------- Lanczos

def dev_expm(eng, l, r, cmo, c, dt, rtol=1e-5, atol=1e-8):
    h, keep = make_heff(eng, l, r, cmo, c.shape)
    Cd = eng.asdevice(c)
    out = eng.empty(c.shape, c.dtype)
    nv = C.c_int()
    dt = complex(dt)
    eng._check(eng.lib.mpse_expm_lanczos(eng.ctx, Cd.code, C.byref(h), dt.real, dt.imag, Cd.ptr, out.ptr, rtol, atol,
                                         0, C.byref(nv)))
    return out.to_host(), nv.value


def test_expm_lanczos_vs_oracle(eng):
    rng = np.random.default_rng(8)
    Dl, Dr, d, wl, wr = 12, 9, 4, 3, 3
    l = _rand(rng, (Dl, wl, Dl), True)
    r = _rand(rng, (Dr, wr, Dr), True)
    l = (l + l.transpose(2, 1, 0).conj()) / 8
    r = (r + r.transpose(2, 1, 0).conj()) / 8
    w = _rand(rng, (wl, d, d, wr), False)
    w = (w + w.transpose(0, 2, 1, 3)) / 2
    c = _rand(rng, (Dl, d, Dr), True)
    for dt in (-0.05j, 0.2j, -0.5j):
        ref, nref = orc.expm_krylov(lambda y: orc.hop_apply(l, r, [w], y.reshape(c.shape)).ravel(), dt, c.ravel())
        out, nv = dev_expm(eng, l, r, [w], c, dt)
        assert nv == nref
        assert _relerr(out.ravel(), ref) < 1e-10
        # against the exact exponential too
        hd = orc.hop_dense(l, r, [w])
        ev, U = np.linalg.eigh(hd)
        exact = U @ (np.exp(dt * ev) * (U.conj().T @ c.ravel()))
        assert _relerr(out.ravel(), exact) < 1e-6
    # 0-site (bond) solve, real dtype with a real step (imaginary-time), tiny full-space case
    r0 = _rand(rng, (Dr, wl, Dr), True)
    r0 = (r0 + r0.transpose(2, 1, 0).conj()) / 8
    s = _rand(rng, (Dl, Dr), True)
    ref, nref = orc.expm_krylov(lambda y: orc.hop_apply(l, r0, [], y.reshape(s.shape)).ravel(), 0.3j, s.ravel())
    out, nv = dev_expm(eng, l, r0, [], s, 0.3j)
    assert nv == nref and _relerr(out.ravel(), ref) < 1e-10
    lr, rr = l.real.copy(), r.real.copy()
    lr = lr + lr.transpose(2, 1, 0)
    rr = rr + rr.transpose(2, 1, 0)
    cr = c.real.copy()
    ref, nref = orc.expm_krylov(lambda y: orc.hop_apply(lr, rr, [w], y.reshape(cr.shape)).ravel(), -0.1, cr.ravel())
    out, nv = dev_expm(eng, lr, rr, [w], cr, -0.1)
    assert nv == nref and _relerr(out.ravel(), ref) < 1e-10
    one = np.ones((1, 1, 1))
    wt = _rand(rng, (1, 2, 2, 1), False)
    wt = wt + wt.transpose(0, 2, 1, 3)
    ct = _rand(rng, (1, 2, 1), True)
    ref, nref = orc.expm_krylov(lambda y: orc.hop_apply(one, one, [wt], y.reshape(ct.shape)).ravel(), -0.4j, ct.ravel())
    out, nv = dev_expm(eng, one, one, [wt], ct, -0.4j)
    assert nv == nref == 2 and _relerr(out.ravel(), ref) < 1e-12


# --------------------------------------------------------------- block QR

def dev_block_qr(eng, c, qnbigl, qnbigr, qntot, system):
    blocks = orc.qn_blocks(qnbigl, qnbigr, qntot)
    nrow = int(np.prod(np.asarray(qnbigl).shape[:-1]))
    ncol = int(np.prod(np.asarray(qnbigr).shape[:-1]))
    rows = np.concatenate([b[2] for b in blocks]).astype(np.int64)
    cols = np.concatenate([b[3] for b in blocks]).astype(np.int64)
    roff = np.cumsum([0] + [len(b[2]) for b in blocks]).astype(np.int64)
    coff = np.cumsum([0] + [len(b[3]) for b in blocks]).astype(np.int64)
    K = int(sum(min(len(b[2]), len(b[3])) for b in blocks))
    Cd = eng.asdevice(np.ascontiguousarray(c).reshape(nrow, ncol))
    U, Vt = eng.empty((nrow, K), c.dtype), eng.empty((K, ncol), c.dtype)
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
    eng._check(eng.lib.mpse_block_qr(eng.ctx, Cd.code, Cd.ptr, nrow, ncol, len(blocks), p(rows), p(roff), p(cols),
                                     p(coff), int(system == "R"), U.ptr, Vt.ptr, K))
    return U.to_host(), Vt.to_host(), blocks


def test_block_qr_golden_inputs(eng, golden_dir):
    z = np.load(os.path.join(golden_dir, "seams.npz"))
    for k in range(int(z["svd_n"])):
        g = lambda n: z[f"svd_{k}_{n}"]
        if not bool(g("QR")):
            continue
        system = str(g("system"))
        c = g("c")
        u, vt, blocks = dev_block_qr(eng, c, g("qnbigl"), g("qnbigr"), g("qntot"), system)
        mat = c.reshape(u.shape[0], -1)
        assert _relerr(u @ vt, mat) < 1e-13
        iso = u if system == "L" else vt.conj().T
        assert np.abs(iso.conj().T @ iso - np.eye(iso.shape[1])).max() < 1e-13
        assert u.shape == g("u").shape and vt.T.shape == g("v").shape


@pytest.mark.parametrize("cplx", [False, True])
def test_block_qr_rank_deficient_and_shapes(eng, cplx):
    rng = np.random.default_rng(9)
    # single block (no symmetry), tall / wide / square, with exactly dependent and zero columns
    for (m, n) in [(300, 40), (40, 300), (64, 64), (1, 7), (7, 1), (513, 130), (1500, 66), (2600, 37), (37, 2600), (6, 1100), (1030, 5),
                   (700, 33), (2100, 20), (3300, 30), (4000, 12)]:   # every rows-per-thread configuration of the panels
        a = _rand(rng, (m, n), cplx)
        if n > 3:
            a[:, 2] = a[:, 0] * (2.0 - 0.5j if cplx else 2.0)   # dependent column
            a[:, 3] = 0                                           # zero column
        if min(m, n) > 10:
            a[:, 5:8] *= 1e-14                                    # numerically negligible columns
        qnl = np.zeros((m, 1), dtype=int)
        qnr = np.zeros((n, 1), dtype=int)
        for system in ("L", "R"):
            u, vt, _ = dev_block_qr(eng, a, qnl, qnr, np.array([0]), system)
            assert _relerr(u @ vt, a) < 1e-13
            iso = u if system == "L" else vt.conj().T
            assert np.abs(iso.conj().T @ iso - np.eye(iso.shape[1])).max() < 1e-13


@pytest.mark.parametrize("cplx", [False, True])
def test_block_qr_many_ragged_blocks_tree_shapes(eng, cplx):
    """Several quantum-number blocks of different heights in ONE decomposition, chosen around the row-slot (256 / 512
    threads x rows per thread) and panel boundaries of the batched Householder kernels: blocks of 255 .. 4096 rows,
    blocks with fewer rows than columns, one-row and one-column blocks; interleaved row / column order.
    Reconstruction, isometry, and the triangular shape of every block's R."""
    rng = np.random.default_rng(21)
    heights = [255, 256, 257, 511, 17, 1, 4096, 1300, 16, 15, 33]
    widths = [40, 16, 130, 17, 33, 5, 256, 48, 16, 31, 1]
    qnl = np.concatenate([np.full(h, b) for b, h in enumerate(heights)])
    qnr = np.concatenate([np.full(w, b) for b, w in enumerate(widths)])
    pl, pr = rng.permutation(len(qnl)), rng.permutation(len(qnr))
    qnl, qnr = qnl[pl], qnr[pr]
    mask = (qnl[:, None] - qnr[None, :]) == 0
    a = _rand(rng, (len(qnl), len(qnr)), cplx) * mask
    a[:, np.where(qnr == 6)[0][3]] = 0                                   # a zero column inside the big block
    a[:, np.where(qnr == 6)[0][7]] = a[:, np.where(qnr == 6)[0][5]]     # an exactly dependent one
    for system in ("L", "R", "L", "R"):
        # blocks: qnl - qnr = 0  <=>  add_outer(qnl, -qnr) == 0
        u, vt, blocks = dev_block_qr(eng, a, qnl[:, None], -qnr[:, None], np.array([0]), system)
        assert _relerr(u @ vt, a) < 1e-13
        iso = u if system == "L" else vt.conj().T
        assert np.abs(iso.conj().T @ iso - np.eye(iso.shape[1])).max() < 1e-13
        k0 = 0
        for _, _, rows, cols in blocks:
            k = min(len(rows), len(cols))
            if system == "L":
                r = vt[k0:k0 + k][:, cols]                                # k x n, upper triangular
                assert np.abs(np.tril(r, -1)).max() == 0
            else:
                l = u[rows][:, k0:k0 + k]                                 # m x k factor of A = L Q: lower triangular
                assert np.abs(np.triu(l, 1)).max() == 0
            k0 += k


def _with_cond(rng, m, n, cond, cplx, rank=None):
    a = _rand(rng, (m, n), cplx)
    u, _, v = np.linalg.svd(a, full_matrices=False)
    sv = np.logspace(0, -np.log10(cond), n) if cond > 1 else np.ones(n)
    if rank is not None:
        sv[rank:] = 0.0
    return (u * sv) @ v


@pytest.mark.parametrize("cplx", [True, False])
def test_block_qr_cholesky_path(eng, cplx):
    """The Cholesky-QR kernels of mpse_block_qr (mpse_cholqr.hip; default for tall blocks of 96 - 256 columns, here also
    forced onto every shape they support, mpse_block_qr_scheme 2) against the same contract as the Householder path:
    reconstruction and isometry to 1e-13, exactly triangular other factor - for two-block layouts like the headline's
    sites, condition numbers up to 1e12 (three real passes), a well-conditioned input (first-order third pass), odd
    sizes and column counts on both sides of the right-looking / left-looking Cholesky kernels (<= 160 / <= 256 columns);
    an exactly zero column has to raise the device flag and come back through the Householder kernels; rank-deficient
    and kappa = 1e18 inputs may do either (the result is held to the same contract).
    Which path ran is read from mpse_block_qr_stats."""
    rng = np.random.default_rng(77)
    cases = [  # rows of block 0 / 1, columns of block 0 / 1, condition, rank of block 0 (None = full), expected path
        (2816, 1280, 145, 111, 1e6, None, "chol"), (2816, 1280, 145, 111, 1e12, None, "chol"),
        (256, 256, 150, 106, 1e7, None, "chol"), (2608, 1488, 182, 74, 1e7, None, "chol"),
        (4096, 0, 256, 0, 1e4, None, "chol"), (1001, 333, 77, 19, 1e8, None, "chol"), (2816, 1280, 145, 111, 3, None, "chol"),
        (2816, 1280, 145, 111, 1e3, 100, "either"), (2816, 1280, 145, 111, 1e18, None, "either"),
        (2816, 1280, 145, 111, 1e3, -7, "fallback"), (700, 300, 33, 17, 1e5, None, "chol")]
    try:
        for scheme in (2, 1):
            eng.block_qr_scheme(scheme)
            for m0, m1, n0, n1, cond, rank, expect in cases:
                m, n = m0 + m1, n0 + n1
                qnl = np.concatenate([np.zeros(m0, int), np.ones(m1, int)])[rng.permutation(m)]
                qnr = np.concatenate([np.zeros(n0, int), np.ones(n1, int)])[rng.permutation(n)]
                a = np.zeros((m, n), dtype=complex if cplx else float)
                blk0 = _with_cond(rng, m0, n0, cond, cplx, rank if rank is None or rank > 0 else None)
                if rank is not None and rank < 0:
                    blk0[:, -rank] = 0                     # an exactly zero column: nothing to normalise in pass 2
                a[np.ix_(qnl == 0, qnr == 0)] = blk0
                if m1 and n1:
                    a[np.ix_(qnl == 1, qnr == 1)] = _with_cond(rng, m1, n1, cond, cplx)
                for system in ("L", "R"):
                    x = a if system == "L" else np.ascontiguousarray(a.conj().T)
                    ql, qr = (qnl, qnr) if system == "L" else (qnr, qnl)
                    s0 = eng.block_qr_stats()
                    u, vt, blocks = dev_block_qr(eng, x, ql[:, None], -qr[:, None], np.array([0]), system)
                    s1 = eng.block_qr_stats()
                    tag = (scheme, m0, m1, n0, n1, cond, rank, system)
                    assert _relerr(u @ vt, x) < 1e-13, tag
                    iso = u if system == "L" else vt.conj().T
                    assert np.abs(iso.conj().T @ iso - np.eye(iso.shape[1])).max() < 1e-13, tag
                    k0 = 0
                    for _, _, rows, cols in blocks:
                        k = min(len(rows), len(cols))
                        if system == "L":
                            assert np.abs(np.tril(vt[k0:k0 + k][:, cols], -1)).max() == 0, tag
                        else:
                            assert np.abs(np.triu(u[rows][:, k0:k0 + k], 1)).max() == 0, tag
                        k0 += k
                    took = s1[1] - s0[1], s1[2] - s0[2]
                    wide_enough = scheme == 2 or (max(m0, m1) >= 256 and max(n0, n1) >= 96)
                    if not wide_enough:
                        assert took == (0, 0), (tag, took)             # default rule: Householder from the start
                    elif expect == "chol":
                        assert took == (1, 0), (tag, took)
                    elif expect == "fallback":
                        assert took == (1, 1), (tag, took)             # tried, flagged on the device, redone
                    else:
                        # rank-deficient / kappa = 1e18 blocks: since round 6 pass 1 shifts the pivots that collapsed
                        # (not the whole diagonal), and the columns it cannot determine may come out as an orthonormal
                        # completion without a breakdown - or the flag goes up and Householder decides.  Either way the
                        # result was checked above
                        assert took in ((1, 0), (1, 1)), (tag, took)
    finally:
        eng.block_qr_scheme(-1)


@pytest.mark.parametrize("cplx", [True, False])
def test_block_qr_cholesky_kappa_window(eng, cplx):
    """The window between "three passes are enough" (kappa <= 1e12) and "certainly flagged" (kappa = 1e18): condition
    numbers 1e13 ... 1e17, several spectra each (geometric decay, a cliff - one tiny singular value -, a plateau of tiny
    ones), tall two-block layouts of the sweep.  Whatever the device decides, the RESULT must be an isometry to 1e-13
    with exact reconstruction: either the Cholesky-QR kernels delivered it (flag down) or the flag went up and the
    Householder kernels did.  "Flag not raised but orthogonality degraded" fails here (round-5 verdict, weak 1 ii)."""
    rng = np.random.default_rng(1234)
    took_tot = np.zeros(2, dtype=int)
    worst = 0.0
    try:
        eng.block_qr_scheme(2)
        for m0, m1, n0, n1 in [(2816, 1280, 145, 111), (512, 300, 165, 91)]:
            m, n = m0 + m1, n0 + n1
            for logk in range(13, 18):
                for spectrum in ("geometric", "cliff", "plateau"):
                    sv = np.ones(n0)
                    if spectrum == "geometric":
                        sv = np.logspace(0, -logk, n0)
                    elif spectrum == "cliff":
                        sv[-1] = 10.0 ** -logk
                    else:
                        sv[n0 // 2:] = 10.0 ** -logk
                    g = _rand(rng, (m0, n0), cplx)
                    uu, _, vv = np.linalg.svd(g, full_matrices=False)
                    blk0 = (uu * sv) @ vv
                    qnl = np.concatenate([np.zeros(m0, int), np.ones(m1, int)])[rng.permutation(m)]
                    qnr = np.concatenate([np.zeros(n0, int), np.ones(n1, int)])[rng.permutation(n)]
                    a = np.zeros((m, n), dtype=complex if cplx else float)
                    a[np.ix_(qnl == 0, qnr == 0)] = blk0
                    a[np.ix_(qnl == 1, qnr == 1)] = _with_cond(rng, m1, n1, 1e5, cplx)
                    for system in ("L", "R"):
                        x = a if system == "L" else np.ascontiguousarray(a.conj().T)
                        ql, qr = (qnl, qnr) if system == "L" else (qnr, qnl)
                        s0 = eng.block_qr_stats()
                        u, vt, _ = dev_block_qr(eng, x, ql[:, None], -qr[:, None], np.array([0]), system)
                        s1 = eng.block_qr_stats()
                        tag = (m0, n0, logk, spectrum, system, s1[1] - s0[1], s1[2] - s0[2])
                        assert s1[1] - s0[1] == 1, tag                     # the Cholesky-QR kernels were tried
                        took_tot += (1, s1[2] - s0[2])
                        iso = u if system == "L" else vt.conj().T
                        orth = np.abs(iso.conj().T @ iso - np.eye(iso.shape[1])).max()
                        worst = max(worst, orth)
                        assert orth < 1e-13, tag
                        assert _relerr(u @ vt, x) < 1e-13, tag
    finally:
        eng.block_qr_scheme(-1)
    # (round 5, shift on the whole diagonal: both outcomes occurred in this window; round 6, shift per pivot: the
    # Cholesky-QR kernels may decide all of it - what is pinned is the contract above, for every case)
    assert took_tot[0] == 2 * 5 * 3 * 2 and 0 <= took_tot[1] <= took_tot[0], took_tot


def test_block_qr_optimistic_flag_edges(eng):
    """The sticky breakdown word of the optimistic mode: up after a tall block with a zero column went through the
    Cholesky-QR kernels unverified, reported only while the mode is on, cleared on BOTH edges of the mode (round-5
    advisor: it used to survive into the verified repeat and into whatever ran next on the context)."""
    rng = np.random.default_rng(5)
    m, n = 2048, 128
    bad = _with_cond(rng, m, n, 1e3, True)
    bad[:, 7] = 0                                  # an exactly zero column: pass 2 has nothing to normalise -> flag
    good = _with_cond(rng, m, n, 1e3, True)
    qnl, qnr = np.zeros((m, 1), int), np.zeros((n, 1), int)
    try:
        eng.block_qr_scheme(2)
        assert eng.block_qr_check() is False
        eng.block_qr_optimistic(True)
        dev_block_qr(eng, good, qnl, qnr, np.array([0]), "L")
        assert eng.block_qr_check() is False
        s0 = eng.block_qr_stats()
        u, vt, _ = dev_block_qr(eng, bad, qnl, qnr, np.array([0]), "L")
        s1 = eng.block_qr_stats()
        assert (s1[1] - s0[1], s1[2] - s0[2]) == (1, 0)              # tried, NOT redone: the caller repeats the step
        assert eng.block_qr_check() is True
        dev_block_qr(eng, good, qnl, qnr, np.array([0]), "L")
        assert eng.block_qr_check() is True                           # sticky while the mode is on
        eng.block_qr_optimistic(False)
        assert eng.block_qr_check() is False                          # off: nothing to report ...
        u, vt, _ = dev_block_qr(eng, bad, qnl, qnr, np.array([0]), "L")
        assert np.abs(u.conj().T @ u - np.eye(n)).max() < 1e-13      # ... every call verified: Householder took it
        eng.block_qr_optimistic(True)
        assert eng.block_qr_check() is False                          # ... and a new optimistic stretch starts clean
    finally:
        eng.block_qr_optimistic(False)
        eng.block_qr_scheme(-1)


@pytest.mark.parametrize("cplx", [False, True])
def test_svd_qn_qr_full_matrices(eng, cplx):
    """``svd_qn(QR=True, full_matrices=True)`` (mps/svd_qn.py:194-197: scipy's ``mode="full"`` per block): the isometry
    side completed to a square unitary per block, the triangular factor padded with zeros, ``u @ v.T`` still the input,
    and the quantum numbers of the extra vectors those of their block.  Tall, wide and square blocks, both systems."""
    from renormalizer_amd.mps import svd_qn as sq
    rng = np.random.default_rng(17)
    heights, widths = [40, 7, 12, 3], [9, 20, 12, 3]
    qnl = np.concatenate([np.full(h, b) for b, h in enumerate(heights)])
    qnr = np.concatenate([np.full(w, b) for b, w in enumerate(widths)])
    qnl, qnr = qnl[rng.permutation(len(qnl))], qnr[rng.permutation(len(qnr))]
    a = _rand(rng, (len(qnl), len(qnr)), cplx) * ((qnl[:, None] - qnr[None, :]) == 0)
    for system in ("L", "R"):
        u, ql, v, qr = sq.svd_qn(eng.asdevice(a), qnl[:, None], -qnr[:, None], np.array([0]), QR=True, system=system,
                                 full_matrices=True)
        ue, qle, ve, qre = sq.svd_qn(eng.asdevice(a), qnl[:, None], -qnr[:, None], np.array([0]), QR=True, system=system,
                                     full_matrices=False)
        uh, vth, K = u.to_host(), v.T.to_host(), ue.shape[1]
        nex = sum(max(h - w, 0) for h, w in zip(heights, widths)) if system == "L" else \
            sum(max(w - h, 0) for h, w in zip(heights, widths))
        assert uh.shape == (len(qnl), K + nex) and vth.shape == (K + nex, len(qnr))
        assert len(ql) == len(qr) == K + nex and ql[:K] == qle and qr[:K] == qre
        assert _relerr(uh @ vth, a) < 1e-13
        assert np.array_equal(uh[:, :K], ue.to_host()) and np.array_equal(vth[:K], ve.T.to_host())
        iso = uh if system == "L" else vth.conj().T
        assert np.abs(iso.conj().T @ iso - np.eye(K + nex)).max() < 1e-13
        tri_extra = vth[K:] if system == "L" else uh[:, K:]
        assert np.abs(tri_extra).max() == 0
        # block by block the completed isometry is SQUARE on its own rows (columns): a unitary of the block
        own = qnl if system == "L" else qnr
        lab = np.array([q[0] for q in (ql if system == "L" else qr)])
        lab = lab if system == "L" else -lab
        for b, (h, w) in enumerate(zip(heights, widths)):
            side = h if system == "L" else w
            blk = iso[np.ix_(own == b, lab == b)]
            if (h >= w) == (system == "L") or h == w:
                assert blk.shape == (side, side)
                assert np.abs(blk @ blk.conj().T - np.eye(side)).max() < 1e-13


# -------------------------------------------------------------- block SVD

def dev_block_svd(eng, c, qnbigl, qnbigr, qntot):
    blocks = orc.qn_blocks(qnbigl, qnbigr, qntot)
    nrow = int(np.prod(np.asarray(qnbigl).shape[:-1]))
    ncol = int(np.prod(np.asarray(qnbigr).shape[:-1]))
    rows = np.concatenate([b[2] for b in blocks]).astype(np.int64)
    cols = np.concatenate([b[3] for b in blocks]).astype(np.int64)
    roff = np.cumsum([0] + [len(b[2]) for b in blocks]).astype(np.int64)
    coff = np.cumsum([0] + [len(b[3]) for b in blocks]).astype(np.int64)
    K = int(sum(min(len(b[2]), len(b[3])) for b in blocks))
    Cd = eng.asdevice(np.ascontiguousarray(c).reshape(nrow, ncol))
    U, Vt = eng.empty((nrow, K), c.dtype), eng.empty((K, ncol), c.dtype)
    S = np.zeros(K)
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
    eng._check(eng.lib.mpse_block_svd(eng.ctx, Cd.code, Cd.ptr, nrow, ncol, len(blocks), p(rows), p(roff), p(cols),
                                      p(coff), U.ptr, Vt.ptr, S.ctypes.data_as(C.POINTER(C.c_double)), K))
    return U.to_host(), S, Vt.to_host(), blocks


def test_block_svd_golden_inputs(eng, golden_dir):
    z = np.load(os.path.join(golden_dir, "seams.npz"))
    for k in range(int(z["svd_n"])):
        g = lambda n: z[f"svd_{k}_{n}"]
        if bool(g("QR")) or bool(g("full")):
            continue
        c = g("c")
        u, s, vt, blocks = dev_block_svd(eng, c, g("qnbigl"), g("qnbigr"), g("qntot"))
        mat = c.reshape(u.shape[0], -1)
        assert _relerr((u * s) @ vt, mat) < 1e-13
        assert np.abs(u.conj().T @ u - np.eye(u.shape[1])).max() < 1e-13
        assert np.abs(vt @ vt.conj().T - np.eye(vt.shape[0])).max() < 1e-13
        # singular values: same multiset as the reference's LAPACK result
        assert np.abs(np.sort(s)[::-1] - g("su")).max() < 1e-13 * g("su").max()


@pytest.mark.parametrize("cplx", [False, True])
def test_block_svd_shapes_and_rank(eng, cplx):
    rng = np.random.default_rng(10)
    for (m, n) in [(120, 40), (40, 120), (64, 64), (1, 7), (7, 1), (33, 1), (257, 90)]:
        a = _rand(rng, (m, n), cplx)
        k = min(m, n)
        if k > 6:
            # rank deficient: kill some singular directions exactly, scale one tiny
            uu, ss, vv = np.linalg.svd(a, full_matrices=False)
            ss[-3:] = 0
            ss[-4] *= 1e-12
            a = (uu * ss) @ vv
        qnl, qnr = np.zeros((m, 1), dtype=int), np.zeros((n, 1), dtype=int)
        u, s, vt, _ = dev_block_svd(eng, a, qnl, qnr, np.array([0]))
        sref = np.linalg.svd(a, compute_uv=False)
        assert np.abs(s - sref).max() < 1e-13 * sref.max()
        assert np.all(np.diff(s) <= 0)
        assert _relerr((u * s) @ vt, a) < 1e-13
        assert np.abs(u.conj().T @ u - np.eye(k)).max() < 1e-12
        assert np.abs(vt @ vt.conj().T - np.eye(k)).max() < 1e-12


@pytest.mark.parametrize("cplx", [False, True])
def test_block_svd_wide_blocks_gram_step(eng, cplx, monkeypatch):
    """Blocks wider than the column kernel's LDS holds (more than 256 complex / 512 real columns) run their sweeps through
    the Gram-matrix block step (``k_jacobi_gram``: MFMA Gram tile, two-sided rotations on the 32 x 32 matrix, MFMA update
    of the rows): values against LAPACK, both factors isometries, reconstruction - for a graded spectrum with a rank
    deficit, two blocks of different width in one call (the narrower one idles through the extra steps), and
    ``MPSE_SVD_GRAM=2`` sending EVERY size of ``test_block_svd_shapes_and_rank`` and the captured extreme-range input
    through the same kernel (column counts that are no multiple of 16, single columns, a block narrower than one tile)."""
    rng = np.random.default_rng(12)

    def check(a, qnl, qnr, tol=1e-13, tol_o=1e-12):
        # (tol: 1e-13 up to ~250 columns as in the other tests of the decomposition; the wide blocks - 300 / 530 columns,
        # eleven sweeps of up to 530 rotations per column - are held to 1e-12: the column kernels give the same figures
        # on these inputs, 3.4e-13 at 530 columns)
        u, s, vt, blocks = dev_block_svd(eng, a, qnl, qnr, np.array([0]))
        sref = np.concatenate([np.linalg.svd(a[np.ix_(ls, rs)], compute_uv=False) for _, _, ls, rs in blocks])
        assert np.abs(np.sort(s)[::-1] - np.sort(sref)[::-1]).max() < tol * sref.max()
        assert _relerr((u * s) @ vt, a) < tol
        assert np.abs(u.conj().T @ u - np.eye(u.shape[1])).max() < tol_o
        assert np.abs(vt @ vt.conj().T - np.eye(vt.shape[0])).max() < tol_o

    n = 300 if cplx else 530
    m = 2 * n + 7
    uu, _ = np.linalg.qr(_rand(rng, (m, n), cplx))
    vv, _ = np.linalg.qr(_rand(rng, (n, n), cplx))
    sv = np.exp(-30.0 * np.arange(n) / n)
    sv[-5:] = 0.0
    a = (uu * sv) @ vv.conj().T
    check(a, np.zeros((m, 1), dtype=int), np.zeros((n, 1), dtype=int), tol=1e-12)
    # two blocks, the second one narrow: one call, shared launches
    n2 = 70
    b = np.zeros((m + 150, n + n2), dtype=a.dtype)
    b[:m, :n] = a
    b[m:, n:] = _rand(rng, (150, n2), cplx)
    qnl = np.concatenate([np.zeros(m, dtype=int), np.ones(150, dtype=int)])[:, None]
    qnr = -np.concatenate([np.zeros(n, dtype=int), np.ones(n2, dtype=int)])[:, None]
    check(b, qnl, qnr, tol=1e-12)
    # every size through the Gram kernel
    monkeypatch.setenv("MPSE_SVD_GRAM", "2")
    for (mm, nn) in [(120, 40), (40, 120), (64, 64), (1, 7), (7, 1), (33, 1), (257, 90), (200, 17), (90, 33)]:
        c = _rand(rng, (mm, nn), cplx)
        k = min(mm, nn)
        if k > 6:
            u0, s0, v0 = np.linalg.svd(c, full_matrices=False)
            s0[-3:] = 0
            s0[-4] *= 1e-12
            c = (u0 * s0) @ v0
        check(c, np.zeros((mm, 1), dtype=int), np.zeros((nn, 1), dtype=int))


def test_block_svd_extreme_dynamic_range_gram_step(eng, golden_dir, monkeypatch):
    monkeypatch.setenv("MPSE_SVD_GRAM", "2")
    test_block_svd_extreme_dynamic_range(eng, golden_dir)


def test_block_svd_extreme_dynamic_range(eng, golden_dir):
    """Regression input captured from expand_bond_dimension: column norms 0.7, 2e-21, 2e-119, 4e-142, 3e-152
    inside one block (products of squared norms underflow).  Jacobi must converge and still return isometries."""
    z = np.load(os.path.join(golden_dir, "svd_dynamic_range.npz"))
    c = z["c"]
    u, s, vt, blocks = dev_block_svd(eng, c, z["qbl"], z["qbr"], z["qntot"])
    mat = c.reshape(u.shape[0], -1)
    assert _relerr((u * s) @ vt, mat) < 1e-13
    assert np.abs(u.conj().T @ u - np.eye(u.shape[1])).max() < 1e-12
    assert np.abs(vt @ vt.conj().T - np.eye(vt.shape[0])).max() < 1e-12
    koff = 0
    for nl, nr, ls, rs in blocks:
        k = min(len(ls), len(rs))
        sref = np.linalg.svd(mat[np.ix_(ls, rs)], compute_uv=False)
        assert abs(s[koff] - sref[0]) < 1e-14
        assert np.all(s[koff + 1: koff + k] < 1e-15)        # the rest is numerically zero either way
        koff += k


@pytest.mark.parametrize("ca,cb", [(True, True), (False, True), (False, False)])
def test_gemm_skip_zero_tiles_equals_dense(eng, ca, cb):
    """Block-sparse operands with the tile-skipping hint: identical results (only all-zero 64 x 16 tiles are skipped),
    for ragged shapes, blocks that straddle tile boundaries, K not a multiple of 16, split-K, beta accumulation, batches
    and an all-zero operand."""
    rng = np.random.default_rng(21)

    def sparse(shape, cplx, row_blocks, col_blocks, fill=0.5):
        a = _rand(rng, shape, cplx)
        mask = np.zeros(shape[-2:], dtype=bool)
        rb = np.linspace(0, shape[-2], row_blocks + 1).astype(int)
        cbk = np.linspace(0, shape[-1], col_blocks + 1).astype(int)
        for i in range(row_blocks):
            for j in range(col_blocks):
                if rng.random() < fill:
                    mask[rb[i]:rb[i + 1], cbk[j]:cbk[j + 1]] = True
        return a * mask
    i1 = E.idx1
    for (M, K, N, batch) in ((300, 520, 200, 1), (64, 1000, 70, 1), (130, 37, 260, 3), (700, 256, 900, 1)):
        A = sparse((batch, M, K), ca, 3, 5)
        B = sparse((batch, K, N), cb, 5, 2)
        C0 = _rand(rng, (batch, M, N), ca or cb)
        dA, dB = eng.asdevice(A), eng.asdevice(B)
        maps = (i1(M, K), i1(K, 1), i1(K, N), i1(N, 1), i1(M, N), i1(N, 1))
        kw = dict(batch=batch, sb_a=M * K, sb_b=K * N, sb_c=M * N, alpha=0.7, beta=-0.3)
        dense = eng.asdevice(C0)
        eng.gemm(dA, dB, dense, *maps, **kw)
        for hint in (1, 2, 3):
            out = eng.asdevice(C0)
            eng.gemm(dA, dB, out, *maps, skip_zero_tiles=hint, **kw)
            assert np.array_equal(out.to_host(), dense.to_host())
        ref = 0.7 * (A @ B) - 0.3 * C0
        assert _relerr(dense.to_host(), ref) < 1e-12
    # one operand entirely zero: the result is beta * C
    A = np.zeros((200, 400))
    B = _rand(rng, (400, 300), cb)
    C0 = _rand(rng, (200, 300), cb)
    out = eng.asdevice(C0)
    eng.gemm(eng.asdevice(A), eng.asdevice(B), out, i1(200, 400), i1(400, 1), i1(400, 300), i1(300, 1), i1(200, 300),
             i1(300, 1), beta=2.0, skip_zero_tiles=3)
    assert np.array_equal(out.to_host(), 2.0 * C0)


def test_device_copies_all_alignments(eng):
    """device-to-device copies: the 16-byte kernel path (complex, even-length real) and the runtime path (odd-length
    real, views that start 8 bytes into an allocation) must both be exact"""
    rng = np.random.default_rng(3)
    for n in (1, 2, 3, 255, 256, 257, 70001):
        a = rng.standard_normal(n)
        t = eng.asdevice(a)
        assert np.array_equal(t.copy().to_host(), a)
        z = a + 1j * rng.standard_normal(n)
        tz = eng.asdevice(z)
        assert np.array_equal(tz.copy().to_host(), z)
        if n > 2:
            shifted = t.shifted(1)                       # starts 8 bytes into the buffer
            dst = eng.empty((n - 1,), np.float64)
            eng._check(eng.lib.mpse_memcpy_d2d(eng.ctx, dst.ptr, shifted.ptr, (n - 1) * 8))
            assert np.array_equal(dst.to_host(), a[1:])


def _holstein_like_site(rng, d):
    """(wl, wr) = (5, 4) MPO site with the block structure of a phonon site of a Holstein chain: channels that pass
    through (identity blocks), one diagonal and one tridiagonal block into the last channel."""
    w = np.zeros((5, d, d, 4))
    eye = np.eye(d)
    w[0, :, :, 0] = eye
    w[0, :, :, 3] = np.diag(rng.standard_normal(d))
    w[1, :, :, 1] = eye
    w[2, :, :, 2] = eye
    w[3, :, :, 3] = np.diag(rng.standard_normal(d - 1), 1) + np.diag(rng.standard_normal(d - 1), -1)
    w[4, :, :, 3] = eye
    return w


@pytest.mark.parametrize("cplx", [True, False])
def test_folded_one_site_matvec_block_sparse_vs_oracle(eng, cplx):
    """The one-site matvec with the MPO step absorbed into the operands of the two large products (mpse_plans.h
    plan_heff1_fold: prepared operands, grouped launches with K segments) at a size that takes that path (D = 256,
    d = 16): block-sparse operands with two bond sectors that do not align with the 64-wide tiles, with and without
    unit channels, against the oracle; then the same product inside a Lanczos solve with the structural mask of the
    centre (masks of the prepared operands derived from it) against the solve without it."""
    from renormalizer_amd.lib.krylov import expm_krylov
    from renormalizer_amd.mps.hop_expr import centre_tile_mask
    rng = np.random.default_rng(31)
    D, d, wl, wr = 256, 16, 5, 4
    sec = np.array([0] * 100 + [1] * 156)
    dl = np.array([0, 1, -1, 0, 0])                       # charge carried by the channels of L
    dr = np.array([0, 1, -1, 0])
    l = _rand(rng, (D, wl, D), cplx) * (sec[:, None, None] - sec[None, None, :] == dl[None, :, None])
    r = _rand(rng, (D, wr, D), cplx) * (sec[:, None, None] - sec[None, None, :] == dr[None, :, None])
    w = _holstein_like_site(rng, d)
    c = _rand(rng, (D, d, D), cplx) * (sec[:, None, None] == sec[None, None, :])
    for lu, ru in ((0, 0), (1, wr), (1, 0), (0, wr)):
        l2, r2 = l.copy(), r.copy()
        if lu:
            l2[:, lu - 1, :] = np.eye(D)
        if ru:
            r2[:, ru - 1, :] = np.eye(D)
        ld, rd = eng.asdevice(l2), eng.asdevice(r2)
        ld.unit, rd.unit = lu, ru
        hop = hop_expr(ld, rd, [w], c.shape)              # a host MPO site: its block structure goes to the engine
        out = hop(eng.asdevice(c)).to_host()
        ref = orc.hop_apply(l2, r2, [w], c)
        assert _relerr(out.ravel(), ref.ravel()) < 1e-12, (lu, ru)
        z = hop(eng.zeros(c.shape, c.dtype)).to_host()
        assert np.abs(z).max() == 0.0
    # (the operator is not Hermitian: with a tiny step both solves stop at their second estimate, and they run the
    # same recurrence on the same numbers - only the tile masks differ)
    ld, rd = eng.asdevice(l2), eng.asdevice(r2)
    ld.unit, rd.unit = 0, wr
    hop = hop_expr(ld, rd, [w], c.shape)
    v0 = eng.asdevice(c.astype(complex))
    plain, n0 = expm_krylov(hop, -1e-7j, v0)
    qn_l = np.repeat(sec[:, None], d, axis=1).reshape(D * d, 1)          # phonon levels carry no charge
    hop.cmask = centre_tile_mask(eng, qn_l, -sec[:, None], np.array([0]), c.shape)
    assert hop.cmask is not None
    masked, n1 = expm_krylov(hop, -1e-7j, v0)
    assert n0 == n1
    assert _relerr(masked.to_host().ravel(), plain.to_host().ravel()) < 1e-13


def test_described_mpo_site_follows_in_place_writes(eng):
    """mpse_mpo_site_hint stores an analysis of the MPO site's values (which channels are zero / the identity) that the
    folded one-site plan relies on: an entry point that writes into the described buffer (scal, copies into it, memset)
    drops the description, so the product follows the new values instead of the stale analysis."""
    rng = np.random.default_rng(33)
    D, d = 256, 16
    w = _holstein_like_site(rng, d)
    l, r = _rand(rng, (D, w.shape[0], D), True), _rand(rng, (D, w.shape[3], D), True)
    c = _rand(rng, (D, d, D), True)
    hop = hop_expr(eng.asdevice(l), eng.asdevice(r), [w], c.shape)
    cd = eng.asdevice(c)
    assert _relerr(hop(cd).to_host().ravel(), orc.hop_apply(l, r, [w], c).ravel()) < 1e-12
    # 1. scale in place: identity channels stop being identities
    hop.cmo[0].scale_(0.5)
    assert _relerr(hop(cd).to_host().ravel(), orc.hop_apply(l, r, [0.5 * w], c).ravel()) < 1e-12
    # 2. overwrite with a dense site: channels the analysis saw as zero are now populated
    w2 = rng.standard_normal(w.shape)
    eng._check(eng.lib.mpse_mpo_site_hint(eng.ctx, hop.cmo[0].ptr, w.ctypes.data, *[int(x) for x in w.shape[:2]], int(w.shape[3])))
    eng._check(eng.lib.mpse_memcpy_h2d(eng.ctx, hop.cmo[0].ptr, w2.ctypes.data, w2.nbytes))
    assert _relerr(hop(cd).to_host().ravel(), orc.hop_apply(l, r, [w2], c).ravel()) < 1e-12
    # 3. a partial write (one row block through a 2-D copy) drops it as well
    eng.mpo_site_hint(hop.cmo[0], w2)
    w3 = w2.copy()
    w3[1] = rng.standard_normal(w3[1].shape)
    flat = hop.cmo[0].reshape(w.shape[0], -1)
    eng.copy_block(flat, 1, 0, eng.asdevice(w3[1].reshape(1, -1)))
    assert _relerr(hop(cd).to_host().ravel(), orc.hop_apply(l, r, [w3], c).ravel()) < 1e-12


@pytest.mark.parametrize("cplx", [False, True])
def test_matrix_module_tensordot_and_paths(eng, cplx):
    """renormalizer_amd.mps.matrix (the module-level names of SURVEY 8(b)): tensordot over device tensors against numpy
    for single, multiple, permuted and interleaved contracted axes (two-level strided indices, host loops beyond),
    multi_tensor_contract on the reference's own contract_one_site paths (mps/lib.py:200-243), asnumpy / asxp."""
    from renormalizer_amd.mps import matrix as mx
    from renormalizer_amd.mps.backend import OE_BACKEND
    assert OE_BACKEND == "mpsengine"
    rng = np.random.default_rng(5)
    cases = [((6, 5, 4), (4, 7), ([2], [0])), ((6, 5, 4), (5, 6, 3), ([0, 1], [1, 0])),
             ((3, 4, 5, 6), (5, 2, 3), ([2, 0], [0, 2])), ((2, 3, 4, 5), (5, 3, 7), ([1, 3], [1, 0])),
             ((4, 3, 2, 5, 6), (6, 2, 3), ([4, 2], [0, 1])), ((5, 4), (5, 4), ([0, 1], [0, 1])), ((7,), (7,), ([0], [0])),
             ((3, 4, 5, 2, 6), (2, 4, 7, 5), ([3, 1, 2], [0, 1, 3]))]
    for sa, sb, axes in cases:
        a, b = _rand(rng, sa, cplx), _rand(rng, sb, cplx and len(sb) > 1)
        got = mx.tensordot(mx.asxp(a), b, axes)
        ref = np.tensordot(a, b, axes)
        assert got.shape == (ref.shape if ref.shape else (1,)) or got.shape == ref.shape
        assert _relerr(mx.asnumpy(got).ravel(), np.asarray(ref).ravel()) < 1e-13, (sa, sb, axes)
    assert mx.asnumpy(None) is None and mx.asxp(None) is None
    # environment update through the reference's path strings, against the engine's own plan
    D, d, w = 12, 3, 4
    ms, mo = _rand(rng, (D, d, D), cplx), rng.standard_normal((w, d, d, w))
    env = _rand(rng, (D, w, D), cplx)
    path = [([0, 1], "fda, abc -> fdbc"), ([2, 0], "fdbc, gdeb -> fcge"), ([1, 0], "fcge, hec -> fgh")]
    out = mx.multi_tensor_contract(path, ms.conj(), env, mo, ms)
    ref = orc.contract_one_site(env, ms, mo, "R")
    assert _relerr(mx.asnumpy(out).ravel(), ref.ravel()) < 1e-12
    m = mx.Matrix(ms)
    assert m.pdim == (d,) and m.bond_dim == (D, D) and m.l_combine().shape == (D * d, D) and m.r_combine().shape == (D, d * D)


def test_truncate_select_vs_oracle_through_cabi(eng):
    """select_basis (mps/lib.py:253-322) through the C ABI - mpse_truncate_select, host-side integer logic inside the
    library - against the oracle: equal quotas per quantum-number block, remaining slots by weight, stable on ties,
    for percent = 0 / 0.2 / 1 and selections smaller and larger than the candidate set (the CPU suite
    holds the same comparison for the Python wrapper)."""
    from renormalizer_amd.mps.basis_select import select_basis_indices
    rng = np.random.default_rng(3)
    for n, nq in ((30, 3), (257, 5), (1, 1)):
        s = rng.random(n)
        if n > 10:
            s[4] = s[9]                                   # a tie
            s[7] = 0.0
        qn = rng.integers(0, nq, (n, 1)).tolist()
        for percent in (0, 0.2, 1.0):
            for mmax in (1, 5, 17, n, n + 50):
                assert select_basis_indices(s, qn, mmax, percent) == orc.select_basis_indices(s, qn, mmax, percent), \
                    (n, percent, mmax)


def test_eigh_qn_density_matrix_blocks(eng):
    """svd_qn.eigh_qn (mps/svd_qn.py:243-302): block eigen-decomposition of a reduced density matrix against LAPACK."""
    from renormalizer_amd.mps import svd_qn
    rng = np.random.default_rng(12)
    qn = rng.integers(0, 3, size=(40, 1))
    comp = np.array([[2], [1], [0], [0]])                     # every sector has a partner for qntot = 2
    a = _rand(rng, (40, 25), True) * 1.0
    same = (qn == qn.T)
    dm = (a @ a.conj().T) * same                              # Hermitian, positive semi-definite, block structured
    u, s, new_qn = svd_qn.eigh_qn(dm, qn, comp, np.array([2]), "L")
    uh = u.to_host()
    assert uh.shape == (40, 40) and len(new_qn) == 40
    assert np.abs(uh.conj().T @ uh - np.eye(40)).max() < 1e-12
    assert np.abs(uh.conj().T @ dm @ uh - np.diag(s ** 2)).max() < 1e-11 * np.abs(dm).max()
    for sector in (0, 1, 2):
        rows = np.nonzero(qn[:, 0] == sector)[0]
        w = np.linalg.eigvalsh(dm[np.ix_(rows, rows)])
        mine = np.sort(np.array([x for x, qq in zip(s, new_qn) if qq == [sector]]) ** 2)
        assert np.abs(mine - np.clip(w, 0, None)).max() < 1e-11 * np.abs(dm).max()
        assert not np.any(np.abs(uh[np.ix_(np.nonzero(qn[:, 0] != sector)[0], [i for i, qq in enumerate(new_qn) if qq == [sector]])]))


@pytest.mark.parametrize("Dl,Dr,w,masked", [(256, 256, 5, True), (256, 256, 5, False), (192, 320, 4, True), (64, 48, 3, True)])
def test_fused_bond_matvec_in_lanczos(eng, Dl, Dr, w, masked, monkeypatch):
    """The 0-site effective Hamiltonian as ONE launch with tile-masked parts (mpse_heff0.hip), through the Lanczos solve
    that consumes it, against the oracle's solve of the same problem (mps/hop_expr.py:63-67, lib/krylov/krylov.py:27-82):
    block-sparse environments with an identity channel each (as canonical sites leave them), a block-diagonal bond
    matrix with and without the structural centre mask, bonds that are / are not multiples of 64, the Krylov dimension
    and the result; then twice more: bitwise the same."""
    import renormalizer_amd.mps.hop_expr as HE
    if os.environ.get("MPSE_HEFF0", "1") == "0":
        pytest.skip("fused bond matvec switched off")
    rng = np.random.default_rng(Dl + 3 * Dr + w)
    # two quantum-number sectors per bond: channel b couples sectors (p, (p + shift_b) % 2)
    sl, sr = (np.arange(Dl) >= Dl // 2 + 16).astype(int), (np.arange(Dr) >= Dr // 2 - 16).astype(int)
    l = np.zeros((Dl, w, Dl), complex)
    r = np.zeros((Dr, w, Dr), complex)

    def block(n, sec, shift, herm):
        x = _rand(rng, (n, n), True) * ((sec[:, None] + shift) % 2 == sec[None, :]) / (4 * np.sqrt(n))
        return x + x.conj().T if herm else x

    # H = sum_b L_b (x) R_b Hermitian: channel 0 = identity (x) Hermitian, channel w - 1 = Hermitian (x) identity, the
    # middle channels change the sector and come in adjoint pairs (an unpaired one keeps the sector and is Hermitian)
    l[:, 0, :], r[:, 0, :] = np.eye(Dl), block(Dr, sr, 0, True)
    l[:, w - 1, :], r[:, w - 1, :] = block(Dl, sl, 0, True), np.eye(Dr)
    b = 1
    while b < w - 1:
        if b + 1 < w - 1:
            l[:, b, :], r[:, b, :] = block(Dl, sl, 1, False), block(Dr, sr, 1, False)
            l[:, b + 1, :], r[:, b + 1, :] = l[:, b, :].conj().T, r[:, b, :].conj().T
            b += 2
        else:
            l[:, b, :], r[:, b, :] = block(Dl, sl, 0, True), block(Dr, sr, 0, True)
            b += 1
    c = _rand(rng, (Dl, Dr), True) * (sl[:, None] == sr[None, :])
    hop = HE.hop_expr(l, r, [], c.shape)
    if masked:
        hop.cmask = HE.centre_tile_mask(eng, sl[:, None], (1 - sr)[:, None], np.array([1]), c.shape)
        assert hop.cmask is not None
    from renormalizer_amd.lib.krylov import expm_krylov
    dt = -0.4j
    ref, nref = orc.expm_krylov(lambda y: orc.hop_apply(l, r, [], y.reshape(c.shape)).ravel(), dt, c.ravel())
    outs = []
    n0 = eng.heff_fused_stats()[0]
    for _ in range(3):
        out, nv = expm_krylov(hop, dt, eng.asdevice(c))
        outs.append(out.to_host())
        assert nv == nref
    # (the fused launch serves the asynchronous solve - the one that takes its result as tile-masked parts; centres of
    # up to 32 768 elements belong to the one-launch kernel of mpse_small.hip whatever MPSE_HEFF0 says)
    if min(Dl, Dr) >= 128 and _FUSED_ON:
        assert eng.heff_fused_stats()[0] - n0 == 3 * nref          # every matvec of the three solves ran fused
    assert _relerr(outs[0].ravel(), ref) < 1e-10
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    # structure is kept: nothing leaks outside the quantum-number blocks
    assert np.abs(outs[0] * (sl[:, None] != sr[None, :])).max() == 0


@pytest.mark.parametrize("Dl,Dr,w,masked", [(256, 256, 4, True), (256, 256, 5, False), (192, 128, 3, True)])
def test_fused_two_level_site_matvec_in_lanczos(eng, Dl, Dr, w, masked):
    """A one-site centre with a two-level physical index (abc,bdef,lfk,cek->adl, mps/hop_expr.py:75-79) through the same
    fused launch: sector-preserving block-sparse Hermitian environments, a real MPO site with diagonal and off-diagonal
    channel blocks, centre with / without its structural mask; Krylov dimension and result against the oracle, three
    runs bitwise equal."""
    import renormalizer_amd.mps.hop_expr as HE
    from renormalizer_amd.lib.krylov import expm_krylov
    if os.environ.get("MPSE_HEFF0", "1") == "0":
        pytest.skip("fused matvec switched off")
    rng = np.random.default_rng(7 * Dl + Dr + w)
    d = 2
    sl, sr = (np.arange(Dl) >= Dl // 2 + 16).astype(int), (np.arange(Dr) >= Dr // 2 - 16).astype(int)

    def herm(n, sec):
        x = _rand(rng, (n, n), True) * (sec[:, None] == sec[None, :]) / (4 * np.sqrt(n))
        return x + x.conj().T

    l = np.stack([np.eye(Dl) if b == 0 else herm(Dl, sl) for b in range(w)], axis=1)
    r = np.stack([np.eye(Dr) if b == w - 1 else herm(Dr, sr) for b in range(w)], axis=1)
    # H = sum_{b, f} L_b (x) W[b, :, :, f] (x) R_f with symmetric 2 x 2 blocks, the channel matrix symmetric: Hermitian
    wm = np.zeros((w, d, d, w))
    ops = [np.eye(2), np.array([[0.0, 1.0], [1.0, 0.0]]), np.diag([0.0, 1.0]), np.array([[0.3, -0.7], [-0.7, 0.1]])]
    for b in range(w):
        wm[b, :, :, b] = ops[b % 4]
    if w >= 3:
        wm[1, :, :, 2] = wm[2, :, :, 1] = 0.5 * ops[3]
        l[:, 2, :] = l[:, 1, :]          # the pair (1, 2) <-> (2, 1) needs L_1 = L_2, R_1 = R_2 for a Hermitian sum
        r[:, 2, :] = r[:, 1, :]
    c = _rand(rng, (Dl, d, Dr), True) * (sl[:, None, None] == sr[None, None, :])
    hop = HE.hop_expr(l, r, [wm], c.shape)
    if masked:
        hop.cmask = HE.centre_tile_mask(eng, np.repeat(sl, d)[:, None], (1 - sr)[:, None], np.array([1]), c.shape)
        assert hop.cmask is not None
    dt = -0.3j
    ref, nref = orc.expm_krylov(lambda y: orc.hop_apply(l, r, [wm], y.reshape(c.shape)).ravel(), dt, c.ravel())
    outs = []
    n0 = eng.heff_fused_stats()[1]
    for _ in range(3):
        out, nv = expm_krylov(hop, dt, eng.asdevice(c))
        outs.append(out.to_host())
        assert nv == nref
    if _FUSED_ON:
        assert eng.heff_fused_stats()[1] - n0 == 3 * nref          # every matvec of the three solves ran fused
    assert _relerr(outs[0].ravel(), ref) < 1e-10
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    assert np.abs(outs[0] * (sl[:, None, None] != sr[None, None, :])).max() == 0
