"""CPU check of the product's contraction plans (renormalizer_amd/csrc/mpse_plans.h):
the plans are executed on host memory by a naive strided-GEMM loop (tests/host_emu) and
compared with the reference-pinned oracle and with golden vectors.  No GPU needed."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import mps_oracle as orc
from renormalizer_amd import engine as E

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("emu") / "libplan_emu.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC",
                           os.path.join(REPO, "tests", "host_emu", "plan_emu.cpp"), "-o", out])
    lib = C.CDLL(out)
    lib.emu_heff_apply.argtypes = [C.c_int, C.POINTER(E.mpse_heff), C.c_void_p, C.c_void_p]
    lib.emu_env_update.argtypes = [C.c_int, C.c_int, C.POINTER(E.mpse_dims), C.c_void_p, C.c_int, C.c_void_p,
                                   C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    lib.emu_set_unit_threshold.argtypes = [C.c_longlong]
    lib.emu_set_unit_threshold(0)        # the product takes the unit-channel copy path only for large slices
    return lib


def _rand(rng, shape, cplx):
    a = rng.standard_normal(shape)
    return a + 1j * rng.standard_normal(shape) if cplx else a


def _c(a):
    return np.ascontiguousarray(a)


def emu_heff(lib, l, r, cmo, c, l_unit=0, r_unit=0):
    cplx = np.iscomplexobj(c)
    dt = E.C128 if cplx else E.F64
    ns = len(cmo)
    h = E.mpse_heff()
    h.nsite = ns
    h.l_unit, h.r_unit = l_unit, r_unit
    anc = 1
    if ns >= 1 and c.ndim == 2 * ns + 2:
        anc = c.shape[2]
    d = h.dims
    d.Dl_ket, d.Dr_ket = c.shape[0], c.shape[-1]
    d.Dl_bra, d.Dr_bra = l.shape[0], r.shape[0]          # rows of the environments: the bonds of the result
    d.danc = anc
    d.danc1 = c.shape[4] if (ns == 2 and c.ndim == 6) else 0
    d.wl, d.wr = l.shape[1], r.shape[1]
    d.d0 = cmo[0].shape[1] if ns >= 1 else 1
    d.d1 = cmo[1].shape[1] if ns == 2 else 1
    d.wm = cmo[0].shape[3] if ns == 2 else 1
    keep = [_c(l), _c(r)] + [_c(w) for w in cmo] + [_c(c)]
    h.L, h.l_dtype = keep[0].ctypes.data, E.dtype_code(keep[0].dtype)
    h.R, h.r_dtype = keep[1].ctypes.data, E.dtype_code(keep[1].dtype)
    if ns >= 1:
        h.W0 = keep[2].ctypes.data
        h.w_dtype = E.dtype_code(keep[2].dtype)
    if ns == 2:
        h.W1 = keep[3].ctypes.data
    out = np.full((l.shape[0],) + c.shape[1:-1] + (r.shape[0],), np.nan, dtype=c.dtype)
    st = lib.emu_heff_apply(dt, C.byref(h), keep[-1].ctypes.data, out.ctypes.data)
    assert st == 0
    return out


def emu_env(lib, env, ket, mo, dom, bra=None, bra_conj=True, env_unit=0):
    cplx = np.iscomplexobj(ket) or np.iscomplexobj(env)
    wdt = complex if cplx else float
    ket = _c(ket.astype(wdt))
    brab = ket if bra is None else _c(bra.astype(wdt))
    env = _c(env)
    mo = _c(mo)
    d = E.mpse_dims()
    d.Dl_ket, d.Dr_ket = ket.shape[0], ket.shape[-1]
    d.Dl_bra, d.Dr_bra = brab.shape[0], brab.shape[-1]
    d.d0 = ket.shape[1]
    d.danc = ket.shape[2] if ket.ndim == 4 else 1
    d.wl, d.wr = mo.shape[0], mo.shape[3]
    d.env_unit = env_unit
    if dom == "L":
        oshape = (d.Dr_bra, d.wr, d.Dr_ket)
    else:
        oshape = (d.Dl_bra, d.wl, d.Dl_ket)
    out = np.full(oshape, np.nan, dtype=wdt)
    st = lib.emu_env_update(E.C128 if cplx else E.F64, 0 if dom == "L" else 1, C.byref(d), env.ctypes.data,
                            E.dtype_code(env.dtype), ket.ctypes.data, brab.ctypes.data, int(bra_conj),
                            mo.ctypes.data, E.dtype_code(mo.dtype), out.ctypes.data)
    assert st == 0
    return out


def test_heff_plans_vs_golden(emu, golden_dir):
    z = np.load(os.path.join(golden_dir, "seams.npz"))
    for k in range(int(z["hop_n"])):
        g = lambda n: z[f"hop_{k}_{n}"]
        ns = int(g("nsite"))
        cmo = [g(f"w{j}") for j in range(ns)]
        out = emu_heff(emu, g("l"), g("r"), cmo, g("c"))
        assert np.abs(out - g("out")).max() < 1e-11 * max(1, np.abs(g("out")).max())


def test_env_plans_vs_golden(emu, golden_dir):
    z = np.load(os.path.join(golden_dir, "seams.npz"))
    for k in range(int(z["c1s_n"])):
        g = lambda n: z[f"c1s_{k}_{n}"]
        dom = str(g("dom"))
        out = emu_env(emu, g("env"), g("ms"), g("mo"), dom, bra=g("bra"), bra_conj=True)
        assert np.abs(out - g("out")).max() < 1e-11 * max(1, np.abs(g("out")).max())
        out = emu_env(emu, g("env"), g("ms"), g("mo"), dom)
        assert np.abs(out - g("out_self")).max() < 1e-11 * max(1, np.abs(g("out_self")).max())


@pytest.mark.parametrize("cplx", [False, True])
def test_env_plans_rectangular_and_sentinel(emu, cplx):
    """bra and ket with different bond dims (transition amplitudes) and the real all-ones sentinel."""
    rng = np.random.default_rng(5)
    for anc in (False, True):
        for dom in ("L", "R"):
            Dlk, Drk, Dlb, Drb, d, da, wl, wr = 4, 6, 3, 5, 3, 2, 2, 4
            ks = (Dlk, d, da, Drk) if anc else (Dlk, d, Drk)
            bs = (Dlb, d, da, Drb) if anc else (Dlb, d, Drb)
            ket, bra = _rand(rng, ks, cplx), _rand(rng, bs, cplx)
            mo = _rand(rng, (wl, d, d, wr), False)
            env = _rand(rng, (Dlb, wl, Dlk) if dom == "L" else (Drb, wr, Drk), cplx)
            ref = orc.contract_one_site(env, ket, mo, dom, ms_conj=bra.conj())
            out = emu_env(emu, env, ket, mo, dom, bra=bra)
            assert np.abs(out - ref).max() < 1e-11 * np.abs(ref).max()
            # pre-conjugated bra buffer with bra_conj=0 (the reference's ms_conj convention)
            out = emu_env(emu, env, ket, mo, dom, bra=bra.conj(), bra_conj=False)
            assert np.abs(out - ref).max() < 1e-11 * np.abs(ref).max()
    # sentinel: edge site, env = ones((1,1,1)) real while the sites are complex
    ket = _rand(rng, (1, 3, 5), cplx)
    mo = _rand(rng, (1, 3, 3, 4), False)
    ref = orc.contract_one_site(np.ones((1, 1, 1)), ket, mo, "L")
    assert np.abs(emu_env(emu, np.ones((1, 1, 1)), ket, mo, "L") - ref).max() < 1e-12


def test_heff_plans_odd_shapes(emu):
    rng = np.random.default_rng(11)
    for cplx in (False, True):
        for (Dl, Dr, d0, d1, wl, wm, wr) in ((1, 3, 2, 3, 1, 2, 3), (7, 1, 4, 2, 3, 1, 1), (5, 5, 2, 2, 4, 5, 4)):
            l = _rand(rng, (Dl, wl, Dl), cplx)
            r = _rand(rng, (Dr, wr, Dr), cplx)
            w0 = _rand(rng, (wl, d0, d0, wr), False)
            c = _rand(rng, (Dl, d0, Dr), cplx)
            ref = orc.hop_apply(l, r, [w0], c)
            assert np.abs(emu_heff(emu, l, r, [w0], c) - ref).max() < 1e-11 * np.abs(ref).max()
            w0 = _rand(rng, (wl, d0, d0, wm), False)
            w1 = _rand(rng, (wm, d1, d1, wr), False)
            c = _rand(rng, (Dl, d0, d1, Dr), cplx)
            ref = orc.hop_apply(l, r, [w0, w1], c)
            assert np.abs(emu_heff(emu, l, r, [w0, w1], c) - ref).max() < 1e-11 * np.abs(ref).max()
            r0 = _rand(rng, (Dr, wl, Dr), cplx)
            c = _rand(rng, (Dl, Dr), cplx)
            ref = orc.hop_apply(l, r0, [], c)
            assert np.abs(emu_heff(emu, l, r0, [], c) - ref).max() < 1e-11 * np.abs(ref).max()


@pytest.mark.parametrize("cplx", [False, True])
def test_heff_plans_rectangular(emu, cplx):
    """bra bonds != ket bonds: H C projected onto another state's bond spaces (variational compression,
    mps/mp.py:513-650), all site counts, with and without the ancilla leg; unit channels must be ignored there"""
    rng = np.random.default_rng(23)
    for (Dlb, Dlk, Drb, Drk, d0, d1, wl, wm, wr, anc) in ((3, 5, 4, 2, 2, 3, 2, 3, 2, 1), (6, 2, 1, 5, 3, 2, 3, 2, 4, 2),
                                                          (4, 4, 2, 6, 2, 2, 2, 2, 3, 1)):
        l = _rand(rng, (Dlb, wl, Dlk), cplx)
        r = _rand(rng, (Drb, wr, Drk), cplx)
        w0 = _rand(rng, (wl, d0, d0, wr), False)
        c = _rand(rng, (Dlk, d0, Drk) if anc == 1 else (Dlk, d0, anc, Drk), cplx)
        ref = np.einsum("abc,bdef,lfk,cek->adl" if anc == 1 else "abc,bdef,lfk,cegk->adgl", l, w0, r, c)
        # a unit-channel hint on a rectangular environment is meaningless and must be ignored
        got = emu_heff(emu, l, r, [w0], c, l_unit=int(Dlb != Dlk), r_unit=int(Drb != Drk))
        assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-11 * np.abs(ref).max()
        w0 = _rand(rng, (wl, d0, d0, wm), False)
        w1 = _rand(rng, (wm, d1, d1, wr), False)
        if anc == 1:
            c = _rand(rng, (Dlk, d0, d1, Drk), cplx)
            ref = np.einsum("abc,bdef,fghj,ljk,cehk->adgl", l, w0, w1, r, c)
        else:
            c = _rand(rng, (Dlk, d0, anc, d1, anc + 1, Drk), cplx)      # the two ancilla legs differ (sites of different size)
            ref = np.einsum("abc,bdef,fghj,ljk,cemhnk->admgnl", l, w0, w1, r, c)
        got = emu_heff(emu, l, r, [w0, w1], c)
        assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-11 * np.abs(ref).max()
        r0 = _rand(rng, (Drb, wl, Drk), cplx)
        c = _rand(rng, (Dlk, Drk), cplx)
        ref = np.einsum("abc,lbk,ck->al", l, r0, c)
        got = emu_heff(emu, l, r0, [], c)
        assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-11 * np.abs(ref).max()


@pytest.mark.parametrize("beta_source", [1, 0])
@pytest.mark.parametrize("cplx", [False, True])
def test_plans_unit_channels(emu, cplx, beta_source):
    """Environments whose channel u is the identity matrix (canonical MPS, no operator applied yet): with
    l_unit / r_unit / env_unit set the plans copy that slice instead of multiplying by it; same result.  On the
    right-hand side the slice is either copied into the result first or read by the first product as its beta term
    straight from the intermediate (``beta_source``); includes w = 1, where no product follows."""
    emu.emu_set_beta_source(beta_source)
    rng = np.random.default_rng(23)

    def with_unit(D, w, u):
        e = _rand(rng, (D, w, D), cplx)
        e[:, u, :] = np.eye(D)
        return e
    for (Dl, Dr, d0, d1, wl, wm, wr, ul, ur) in ((5, 4, 3, 2, 3, 2, 4, 0, 3), (4, 6, 2, 3, 4, 3, 3, 2, 1),
                                                 (3, 3, 2, 2, 1, 2, 1, 0, 0), (6, 5, 4, 2, 5, 2, 2, 4, 0)):
        l, r = with_unit(Dl, wl, ul), with_unit(Dr, wr, ur)
        w0 = _rand(rng, (wl, d0, d0, wr), False)
        for anc in (False, True):
            c = _rand(rng, (Dl, d0, 2, Dr) if anc else (Dl, d0, Dr), cplx)
            ref = orc.hop_apply(l, r, [w0], c)
            for lu, ru in ((ul + 1, ur + 1), (ul + 1, 0), (0, ur + 1)):
                out = emu_heff(emu, l, r, [w0], c, l_unit=lu, r_unit=ru)
                assert np.abs(out - ref).max() < 1e-11 * np.abs(ref).max()
        w0m = _rand(rng, (wl, d0, d0, wm), False)
        w1 = _rand(rng, (wm, d1, d1, wr), False)
        for anc in (False, True):
            c = _rand(rng, (Dl, d0, 2, d1, 2, Dr) if anc else (Dl, d0, d1, Dr), cplx)
            ref = orc.hop_apply(l, r, [w0m, w1], c)
            out = emu_heff(emu, l, r, [w0m, w1], c, l_unit=ul + 1, r_unit=ur + 1)
            assert np.abs(out - ref).max() < 1e-11 * np.abs(ref).max()
        if wl == wr or True:
            r0 = with_unit(Dr, wl, min(ur, wl - 1))
            c = _rand(rng, (Dl, Dr), cplx)
            ref = orc.hop_apply(l, r0, [], c)
            out = emu_heff(emu, l, r0, [], c, l_unit=ul + 1, r_unit=min(ur, wl - 1) + 1)
            assert np.abs(out - ref).max() < 1e-11 * np.abs(ref).max()
        # environment updates (bra = ket so that the bonds agree)
        for anc in (False, True):
            ket = _rand(rng, (Dl, d0, 2, Dr) if anc else (Dl, d0, Dr), cplx)
            ref = orc.contract_one_site(l, ket, w0, "L")
            assert np.abs(emu_env(emu, l, ket, w0, "L", env_unit=ul + 1) - ref).max() < 1e-11 * np.abs(ref).max()
            ref = orc.contract_one_site(r, ket, w0, "R")
            assert np.abs(emu_env(emu, r, ket, w0, "R", env_unit=ur + 1) - ref).max() < 1e-11 * np.abs(ref).max()
    # a unit flag on a rectangular environment (bra and ket bonds differ) must be ignored, not misapplied
    ket, bra = _rand(rng, (4, 3, 6), cplx), _rand(rng, (3, 3, 5), cplx)
    mo = _rand(rng, (2, 3, 3, 4), False)
    env = _rand(rng, (3, 2, 4), cplx)
    ref = orc.contract_one_site(env, ket, mo, "L", ms_conj=bra.conj())
    assert np.abs(emu_env(emu, env, ket, mo, "L", bra=bra, env_unit=1) - ref).max() < 1e-11 * np.abs(ref).max()
    emu.emu_set_beta_source(1)


def test_plans_random_shapes_property(emu):
    """Randomised sweep over extents (ragged, 1-wide bonds, ancilla on/off, unit channels on/off, real / complex):
    every plan the engine can generate must agree with the oracle's three tensordots."""
    from hypothesis import given, settings, strategies as st

    dims = st.integers(min_value=1, max_value=6)

    @settings(max_examples=40, deadline=None, derandomize=True, database=None)
    @given(Dl=dims, Dr=dims, d0=st.integers(1, 4), d1=st.integers(1, 3), wl=st.integers(1, 4), wm=st.integers(1, 3),
           wr=st.integers(1, 4), anc=st.booleans(), cplx=st.booleans(), unit=st.booleans(), seed=st.integers(0, 2 ** 16))
    def check(Dl, Dr, d0, d1, wl, wm, wr, anc, cplx, unit, seed):
        rng = np.random.default_rng(seed)
        l, r = _rand(rng, (Dl, wl, Dl), cplx), _rand(rng, (Dr, wr, Dr), cplx)
        ul = ur = 0
        if unit:
            ul, ur = int(rng.integers(0, wl)) + 1, int(rng.integers(0, wr)) + 1
            l[:, ul - 1, :] = np.eye(Dl)
            r[:, ur - 1, :] = np.eye(Dr)
        w0 = _rand(rng, (wl, d0, d0, wr), False)
        c = _rand(rng, (Dl, d0, 2, Dr) if anc else (Dl, d0, Dr), cplx)
        ref = orc.hop_apply(l, r, [w0], c)
        out = emu_heff(emu, l, r, [w0], c, l_unit=ul, r_unit=ur)
        assert np.abs(out - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())
        w0m, w1 = _rand(rng, (wl, d0, d0, wm), False), _rand(rng, (wm, d1, d1, wr), False)
        c2 = _rand(rng, (Dl, d0, 2, d1, 2, Dr) if anc else (Dl, d0, d1, Dr), cplx)
        ref = orc.hop_apply(l, r, [w0m, w1], c2)
        out = emu_heff(emu, l, r, [w0m, w1], c2, l_unit=ul, r_unit=ur)
        assert np.abs(out - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())
        ket = _rand(rng, (Dl, d0, 2, Dr) if anc else (Dl, d0, Dr), cplx)
        for dom, env, u in (("L", l, ul), ("R", r, ur)):
            ref = orc.contract_one_site(env, ket, w0, dom)
            out = emu_env(emu, env, ket, w0, dom, env_unit=u)
            assert np.abs(out - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())

    check()


def test_stacked_mpo_plans_vs_golden(emu, golden_dir):
    """plan_env_multi / plan_heff2 (mps/lib.py:121-166, mps/hop_expr.py:24-52) against reference-captured seams
    (tests/golden/dmrg_seams.npz), and with three layers against einsum."""
    emu.emu_env_update_multi.argtypes = [C.c_int, C.c_int, C.POINTER(E.mpse_dims), C.c_int, C.POINTER(C.c_int64),
                                         C.POINTER(C.c_int64), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                         C.POINTER(C.c_void_p), C.c_int, C.c_void_p]
    emu.emu_heff_apply2.argtypes = [C.c_int, C.POINTER(E.mpse_heff), C.c_void_p, C.c_void_p]
    z = np.load(os.path.join(golden_dir, "dmrg_seams.npz"))

    def env_multi(env, ms, mos, dom):
        cplx = np.iscomplexobj(ms) or np.iscomplexobj(env)
        wdt = complex if cplx else float
        ket = _c(ms.astype(wdt))
        env = _c(env)
        mos = [_c(m) for m in mos]
        n = len(mos)
        d = E.mpse_dims()
        d.Dl_ket = d.Dl_bra = ket.shape[0]
        d.Dr_ket = d.Dr_bra = ket.shape[-1]
        d.d0 = ket.shape[1]
        d.danc = ket.shape[2] if ket.ndim == 4 else 1
        wl = (C.c_int64 * n)(*[m.shape[0] for m in mos])
        wr = (C.c_int64 * n)(*[m.shape[3] for m in mos])
        ws = (C.c_void_p * n)(*[m.ctypes.data for m in mos])
        if dom == "L":
            oshape = (ket.shape[-1],) + tuple(m.shape[3] for m in mos) + (ket.shape[-1],)
        else:
            oshape = (ket.shape[0],) + tuple(m.shape[0] for m in mos) + (ket.shape[0],)
        out = np.full(oshape, np.nan, dtype=wdt)
        st = emu.emu_env_update_multi(E.C128 if cplx else E.F64, 0 if dom == "L" else 1, C.byref(d), n, wl, wr,
                                      env.ctypes.data, E.dtype_code(env.dtype), ket.ctypes.data, None, 1, ws,
                                      E.dtype_code(mos[0].dtype), out.ctypes.data)
        assert st == 0
        return out

    for k in range(int(z["menv_n"])):
        g = lambda n: z[f"menv{k}_{n}"]
        out = env_multi(g("env"), g("ms"), [g("mo1"), g("mo2")], str(g("dom")))
        assert out.shape == g("out").shape
        assert np.abs(out - g("out")).max() < 1e-11 * max(1, np.abs(g("out")).max()), k
    rng = np.random.default_rng(3)
    ms = _rand(rng, (4, 3, 5), True)
    mos = [_rand(rng, s, False) for s in ((2, 3, 3, 3), (3, 3, 3, 2), (2, 3, 3, 4))]
    envl = _rand(rng, (4, 2, 3, 2, 4), True)
    ref = np.einsum("abcde,axp,bxyf,cyzg,dzuh,euq->pfghq", envl, ms.conj(), *mos, ms)
    assert np.abs(env_multi(envl, ms, mos, "L") - ref).max() < 1e-11 * np.abs(ref).max()
    envr = _rand(rng, (5, 3, 2, 4, 5), True)
    ref = np.einsum("abcde,pxa,fxyb,gyzc,hzud,que->pfghq", envr, ms.conj(), *mos, ms)
    assert np.abs(env_multi(envr, ms, mos, "R") - ref).max() < 1e-11 * np.abs(ref).max()
    one = env_multi(envl[:, :, 0, 0, :], ms, mos[:1], "L")               # a single layer is the ordinary update
    assert np.abs(one - orc.contract_one_site(envl[:, :, 0, 0, :], ms, mos[0], "L")).max() < 1e-11

    for k in range(int(z["hop2_n"])):
        g = lambda n: z[f"hop2_{k}_{n}"]
        ns = int(g("nsite"))
        l, r, c = _c(g("l")), _c(g("r")), _c(g("c"))
        cmo = [_c(g(f"w{j}")) for j in range(ns)]
        h = E.mpse_heff()
        h.nsite = ns
        d = h.dims
        d.Dl_ket = d.Dl_bra = c.shape[0]
        d.Dr_ket = d.Dr_bra = c.shape[-1]
        d.danc = 1
        d.wl, d.wr = l.shape[1], r.shape[1]
        d.d0 = cmo[0].shape[1]
        d.d1 = cmo[1].shape[1] if ns == 2 else 1
        d.wm = cmo[0].shape[3] if ns == 2 else 1
        h.L, h.l_dtype, h.R, h.r_dtype = l.ctypes.data, E.dtype_code(l.dtype), r.ctypes.data, E.dtype_code(r.dtype)
        h.W0, h.w_dtype = cmo[0].ctypes.data, E.dtype_code(cmo[0].dtype)
        if ns == 2:
            h.W1 = cmo[1].ctypes.data
        out = np.full(c.shape, np.nan, dtype=c.dtype)
        assert emu.emu_heff_apply2(E.dtype_code(c.dtype), C.byref(h), c.ctypes.data, out.ctypes.data) == 0
        assert np.abs(out - g("out")).max() < 1e-11 * max(1, np.abs(g("out")).max()), k
        # three centres at once through the batch index next to the right bond
        cz = _c(np.stack([c, 2 * c, -c[::-1]], axis=-2))
        if ns == 1:
            d.danc = 3
        else:
            d.danc1 = 3
        outz = np.full(cz.shape, np.nan, dtype=c.dtype)
        assert emu.emu_heff_apply2(E.dtype_code(c.dtype), C.byref(h), cz.ctypes.data, outz.ctypes.data) == 0
        assert np.abs(outz[..., 0, :] - g("out")).max() < 1e-11 * max(1, np.abs(g("out")).max())
        assert np.abs(outz[..., 1, :] - 2 * g("out")).max() < 1e-10 * max(1, np.abs(g("out")).max())


def _fold_w(rng, wl, d, wr, kind):
    """MPO sites with the block structure the folded plan distinguishes: identity blocks (channels that pass through),
    general blocks, empty channels."""
    w = np.zeros((wl, d, d, wr))
    if kind == "holstein":        # (wl, wr) = (5, 4): I->I, I->done (diag), a->a, b->b, x->done (general), done->done
        assert (wl, wr) == (5, 4)
        eye = np.eye(d)
        w[0, :, :, 0] = eye
        w[0, :, :, 3] = np.diag(rng.standard_normal(d))
        w[1, :, :, 1] = eye
        w[2, :, :, 2] = eye
        w[3, :, :, 3] = np.diag(rng.standard_normal(d - 1), 1) + np.diag(rng.standard_normal(d - 1), -1)
        w[4, :, :, 3] = eye
    elif kind == "mixed":         # identity blocks that are not alone in their channel, a scaled identity, an empty channel
        eye = np.eye(d)
        for b in range(wl):
            w[b, :, :, b % (wr - 1)] = eye if b % 2 == 0 else 0.5 * eye
        w[0, :, :, 0] += rng.standard_normal((d, d))
    else:                         # dense: every block general
        w = rng.standard_normal((wl, d, d, wr))
    return w


def test_folded_one_site_plan(emu):
    """plan_heff1_fold (MPO step absorbed into the operands of the two large products, mpse_plans.h): the same
    numbers as the dense contraction with and without unit channels on either side, for pass-through, mixed and dense
    MPO sites; a site with more blocks per channel than the elementwise pass takes is refused (the caller takes the
    three-step chain)."""
    emu.emu_heff_apply_fold.argtypes = [C.c_int, C.POINTER(E.mpse_heff), C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    emu.emu_set_fold_min.argtypes = [C.c_longlong, C.c_longlong, C.c_longlong]
    emu.emu_set_fold_min(1, 4, 1)       # (the device needs bonds in multiples of 64: tile = channel; not the host loops)
    try:
        rng = np.random.default_rng(23)
        for cplx in (False, True):
            # (D d a multiple of 64 with a complex centre: the products with R as halved tiles adding into a zeroed result)
            for (D, d, wl, wr, kind) in ((12, 4, 5, 4, "holstein"), (16, 4, 5, 4, "holstein"), (8, 3, 4, 4, "mixed"),
                                         (8, 2, 2, 2, "dense")):
                w0 = _fold_w(rng, wl, d, wr, kind)
                c = _rand(rng, (D, d, D), cplx)
                for lu, ru in ((0, 0), (1, wr), (1, 0), (0, wr), (wl, 1)):
                    l = _rand(rng, (D, wl, D), cplx)
                    r = _rand(rng, (D, wr, D), cplx)
                    if lu:
                        l[:, lu - 1, :] = np.eye(D)
                    if ru:
                        r[:, ru - 1, :] = np.eye(D)
                    ref = np.einsum("abc,bdef,lfk,cek->adl", l, w0, r, c)
                    h = E.mpse_heff()
                    h.nsite, h.l_unit, h.r_unit = 1, lu, ru
                    dm = h.dims
                    dm.Dl_ket = dm.Dl_bra = dm.Dr_ket = dm.Dr_bra = D
                    dm.danc, dm.wl, dm.wr, dm.d0, dm.d1, dm.wm = 1, wl, wr, d, 1, 1
                    keep = [_c(l), _c(r), _c(w0), _c(c)]
                    h.L, h.l_dtype = keep[0].ctypes.data, E.dtype_code(keep[0].dtype)
                    h.R, h.r_dtype = keep[1].ctypes.data, E.dtype_code(keep[1].dtype)
                    h.W0, h.w_dtype = keep[2].ctypes.data, E.F64
                    out = np.full(c.shape, np.nan, dtype=c.dtype)
                    nsteps = C.c_int(0)
                    st = emu.emu_heff_apply_fold(E.C128 if cplx else E.F64, C.byref(h), keep[3].ctypes.data,
                                                 out.ctypes.data, C.byref(nsteps))
                    assert st == 0, (kind, lu, ru)
                    assert nsteps.value % 100 <= 3
                    assert (nsteps.value >= 100) == (cplx and (D * d) % 64 == 0)      # the result in two parts
                    assert np.abs(out - ref).max() < 1e-11 * np.abs(ref).max(), (kind, lu, ru)
        # six channels into one: refused
        w0 = rng.standard_normal((6, 2, 2, 1))
        l, r, c = _rand(rng, (8, 6, 8), False), _rand(rng, (8, 1, 8), False), _rand(rng, (8, 2, 8), False)
        h = E.mpse_heff()
        h.nsite = 1
        dm = h.dims
        dm.Dl_ket = dm.Dl_bra = dm.Dr_ket = dm.Dr_bra = 8
        dm.danc, dm.wl, dm.wr, dm.d0, dm.d1, dm.wm = 1, 6, 1, 2, 1, 1
        keep = [_c(l), _c(r), _c(w0), _c(c)]
        h.L, h.l_dtype, h.R, h.r_dtype = keep[0].ctypes.data, E.F64, keep[1].ctypes.data, E.F64
        h.W0, h.w_dtype = keep[2].ctypes.data, E.F64
        out = np.zeros_like(c)
        assert emu.emu_heff_apply_fold(E.F64, C.byref(h), keep[3].ctypes.data, out.ctypes.data, None) != 0
    finally:
        emu.emu_set_fold_min(1 << 28, 64, 8)
