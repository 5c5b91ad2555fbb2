"""Environment tensors of the sweep algorithms, device resident.

Counterpart of renormalizer/mps/lib.py: ``contract_one_site`` (:169-250) is one C-ABI call
(mpse_env_update: two large FP64-MFMA GEMMs around the small MPO contraction) and ``Environ``
(:12-118) is a table of HBM-resident handles - the reference's CuPy path copies every
environment to the host on write and back on read (lib.py:114-118)."""
import ctypes as C
import os

import numpy as np

from ..engine import DOMAIN_L, DOMAIN_R, DeviceTensor, get_engine, mpse_dims


def _as_w(eng, mo):
    if isinstance(mo, DeviceTensor):
        return mo
    return eng.asdevice(np.asarray(mo))


_UNIT_PREDICTIONS = {}


def predict_unit_channel(mo_host, domain, unit_in):
    """Unit channel (1-based, 0 = none) of the environment that results from updating an environment whose channel
    ``unit_in`` is the identity matrix with a site that is an exact isometry in that direction: the outgoing channel g
    is again the identity when the MPO site passes ``unit_in`` through unchanged and nothing else feeds g.  Pure host
    logic on the (KB sized) MPO site; the alternative is to measure it on the device (``find_unit_channel``: one small
    kernel and a read-back per update)."""
    if unit_in <= 0 or mo_host is None:
        return 0
    key = (id(mo_host), domain, int(unit_in))
    hit = _UNIT_PREDICTIONS.get(key)
    if hit is not None and hit[0] is mo_host:
        return hit[1]
    w = np.asarray(mo_host)
    u = int(unit_in) - 1
    eye = np.eye(w.shape[1])
    out = 0
    if domain == "L" and u < w.shape[0]:
        for g in range(w.shape[3]):
            if np.array_equal(w[u, :, :, g], eye) and not np.any(np.delete(w[:, :, :, g], u, axis=0)):
                out = g + 1
                break
    elif domain == "R" and u < w.shape[3]:
        for q in range(w.shape[0]):
            if np.array_equal(w[q, :, :, u], eye) and not np.any(np.delete(w[q], u, axis=2)):
                out = q + 1
                break
    if len(_UNIT_PREDICTIONS) > 4096:
        _UNIT_PREDICTIONS.clear()
    _UNIT_PREDICTIONS[key] = (mo_host, out)
    return out


def contract_one_site(environ, ms, mo, domain, ms_conj=None, canonical_mo_host=None):
    """One environment update.  ``ms_conj`` follows the reference convention: it holds the
    already-conjugated bra tensor; ``None`` means conj(ms) (no copy is made).  ``canonical_mo_host``: the host copy of
    ``mo`` when the caller guarantees that ``ms`` is an isometry in the direction of the update (it just came out of a
    QR): the unit channel of the result is then predicted on the host instead of measured on the device."""
    assert domain in ["L", "R"]
    eng = ms.eng
    mo = _as_w(eng, mo)
    bra = ms if ms_conj is None else ms_conj
    if ms.ndim not in (3, 4):
        raise ValueError(f"MPS ndim is not 3 or 4, got {ms.ndim}")
    cplx = ms.is_complex or environ.is_complex or bra.is_complex or mo.is_complex
    dt = np.complex128 if cplx else np.float64
    ket = ms.to_complex() if cplx else ms
    bra = bra.to_complex() if cplx else bra
    d = mpse_dims()
    d.Dl_ket, d.Dr_ket = ket.shape[0], ket.shape[-1]
    d.Dl_bra, d.Dr_bra = bra.shape[0], bra.shape[-1]
    d.d0, d.d1 = ket.shape[1], 1
    d.danc = ket.shape[2] if ket.ndim == 4 else 1
    d.wl, d.wr, d.wm = mo.shape[0], mo.shape[3], 1
    d.env_unit = environ.unit
    if domain == "L":
        assert environ.shape == (d.Dl_bra, d.wl, d.Dl_ket), (environ.shape, bra.shape, mo.shape, ket.shape)
        oshape = (d.Dr_bra, d.wr, d.Dr_ket)
    else:
        assert environ.shape == (d.Dr_bra, d.wr, d.Dr_ket), (environ.shape, bra.shape, mo.shape, ket.shape)
        oshape = (d.Dl_bra, d.wl, d.Dl_ket)
    out = eng.empty(oshape, dt)
    eng._check(eng.lib.mpse_env_update(
        eng.ctx, out.code, DOMAIN_L if domain == "L" else DOMAIN_R, C.byref(d), environ.ptr, environ.code,
        ket.ptr, bra.ptr, 1 if ms_conj is None else 0, mo.ptr, mo.code, out.ptr))
    if ms_conj is None and oshape[0] == oshape[2]:
        deferred = eng.recording_list >= 0      # the update has only been recorded: nothing to measure yet
        if canonical_mo_host is not None and ms.ndim == 3 and environ.unit > 0:
            # the chain of predictions starts at the sentinel and is carried through small bonds as well
            out.unit = predict_unit_channel(canonical_mo_host, domain, environ.unit)
            if VERIFY_UNIT and oshape[0] >= UNIT_MIN_BOND and not deferred:
                measured = find_unit_channel(out)
                if measured != out.unit and eng.block_qr_check():
                    # optimistic block QR (Mps._evolve_tdvp_ps): this site came out of a decomposition that broke down
                    # and is no isometry - the step is going to be discarded and repeated, tell the driver now
                    raise ArithmeticError("unit channel of an environment behind a block QR that broke down")
                assert measured == out.unit, f"unit channel predicted {out.unit}, measured {measured}"
        elif oshape[0] >= UNIT_MIN_BOND and not deferred:
            # (a recorded update behind an environment without unit channel keeps unit = 0: an isometric site cannot
            # create one, and a missed unit channel costs a GEMM, never correctness)
            out.unit = find_unit_channel(out)
    return out


def contract_one_site_multi_mpo(environ, ms, mos, domain, ms_conj=None):
    """Environment update through a stack of MPO sites (mps/lib.py:121-166): ``environ`` has one leg per MPO between
    its bra and ket legs, layer 0 touches the bra.  One C-ABI call (mpse_env_update_multi)."""
    assert domain in ["L", "R"]
    eng = ms.eng
    mos = [_as_w(eng, mo) for mo in mos]
    n = len(mos)
    bra = ms if ms_conj is None else ms_conj
    if ms.ndim not in (3, 4):
        raise ValueError(f"MPS ndim is not 3 or 4, got {ms.ndim}")
    if environ.ndim != n + 2:
        raise ValueError(f"environment with {environ.ndim - 2} MPO legs for {n} MPO sites")
    wcplx = any(mo.is_complex for mo in mos)
    cplx = ms.is_complex or environ.is_complex or bra.is_complex or wcplx
    dt = np.complex128 if cplx else np.float64
    ket = ms.to_complex() if cplx else ms
    bra = bra.to_complex() if cplx else bra
    if wcplx:
        mos = [mo.to_complex() for mo in mos]
    d = mpse_dims()
    d.Dl_ket, d.Dr_ket = ket.shape[0], ket.shape[-1]
    d.Dl_bra, d.Dr_bra = bra.shape[0], bra.shape[-1]
    d.d0, d.d1 = ket.shape[1], 1
    d.danc = ket.shape[2] if ket.ndim == 4 else 1
    wl = (C.c_int64 * n)(*[mo.shape[0] for mo in mos])
    wr = (C.c_int64 * n)(*[mo.shape[3] for mo in mos])
    ptrs = (C.c_void_p * n)(*[mo.ptr for mo in mos])
    if domain == "L":
        assert environ.shape == (d.Dl_bra,) + tuple(wl) + (d.Dl_ket,), (environ.shape, tuple(wl))
        oshape = (d.Dr_bra,) + tuple(wr) + (d.Dr_ket,)
    else:
        assert environ.shape == (d.Dr_bra,) + tuple(wr) + (d.Dr_ket,), (environ.shape, tuple(wr))
        oshape = (d.Dl_bra,) + tuple(wl) + (d.Dl_ket,)
    out = eng.empty(oshape, dt)
    eng._check(eng.lib.mpse_env_update_multi(
        eng.ctx, out.code, DOMAIN_L if domain == "L" else DOMAIN_R, C.byref(d), n, wl, wr, environ.ptr, environ.code,
        ket.ptr, bra.ptr, 1 if ms_conj is None else 0, ptrs, mos[0].code, out.ptr))
    return out


UNIT_TOL = 1e-12
# The contraction plans use a unit channel only when the product it saves has at least 2^27 multiply-adds
# (mpse_plans.h unit_pays: D^2 x d D for a one-site matvec): below D = 128 no physical dimension reaches that, and the
# detection - a small kernel plus a host round trip per environment update - would be pure latency.
UNIT_MIN_BOND = 128
# debug (tests): every predicted unit channel is also measured on the device and compared; the sweep then runs the
# plain loop (a recorded update cannot be measured before it has run)
VERIFY_UNIT = False


def find_unit_channel(env):
    """1-based MPO-bond channel along which the environment is the identity matrix to UNIT_TOL (0 if none).  With
    a canonical MPS the channel in which no operator has acted yet is exactly that; the contraction plans then
    replace its share of the two big GEMMs of every matvec / environment update by a copy."""
    eng = env.eng
    unit = C.c_int64(0)
    eng._check(eng.lib.mpse_env_unit_channel(eng.ctx, env.code, env.ptr, env.shape[0], env.shape[1], UNIT_TOL,
                                             C.byref(unit)))
    return int(unit.value)


class Environ:
    """Cache {("L"|"R", idx): environment}; idx is the site the tensor reaches up to
    (L(idx-1) - mpo(idx) - R(idx+1)).  ``domain=None`` builds both directions like the
    reference; pass "L" or "R" to build only what a sweep needs."""

    def __init__(self, mps, mpo, domain=None, mps_conj=None):
        self.eng = get_engine()
        self._virtual_disk = {}
        # a list of MPOs stacks their sites between bra and ket (lib.py:24-28, used for (H - omega)^2)
        self.multi = isinstance(mpo, (list, tuple))
        if self.multi:
            self.sentinel = self.eng.ones((1,) * (len(mpo) + 2), np.float64)
        else:
            self.sentinel = self.eng.ones((1, 1, 1), np.float64)
            self.sentinel.unit = 1
        self._construct(mps, mpo, domain, mps_conj)

    def _mo(self, mpo, idx):
        if isinstance(mpo, (list, tuple)):
            return [self._mo(m, idx) for m in mpo]
        return mpo.device(idx, self.eng) if hasattr(mpo, "device") else _as_w(self.eng, mpo[idx])

    def _step(self, tensor, ms, mo, domain, ms_conj):
        if isinstance(mo, list):
            return contract_one_site_multi_mpo(tensor, ms, mo, domain, ms_conj=ms_conj)
        return contract_one_site(tensor, ms, mo, domain, ms_conj=ms_conj)

    def _construct(self, mps, mpo, domain=None, mps_conj=None):
        assert domain in ["L", "R", None]
        if domain is None:
            self._construct(mps, mpo, "L", mps_conj)
            self._construct(mps, mpo, "R", mps_conj)
            return
        n = len(mps)
        rng = range(0, n - 1) if domain == "L" else range(n - 1, 0, -1)
        self.write("L", -1, self.sentinel)
        self.write("R", n, self.sentinel)
        tensor = self.sentinel
        for idx in rng:
            cj = None if mps_conj is None else mps_conj[idx]
            tensor = self._step(tensor, mps[idx], self._mo(mpo, idx), domain, cj)
            self.write(domain, idx, tensor)

    def GetLR(self, domain, siteidx, mps, mpo, itensor=None, method="Scratch", mps_conj=None, canonical=False):
        """``canonical``: (method "System" only) the site at ``siteidx`` is an exact isometry in the direction of the
        update - the sweeps pass True right after a QR - so the unit channel of the new environment is predicted from
        the MPO site instead of being measured."""
        assert domain in ["L", "R"]
        assert method in ["Enviro", "System", "Scratch"]
        if mps_conj is None:
            mps_conj = [None] * len(mps)
        if siteidx not in range(len(mps)):
            return self.sentinel
        if method == "Scratch":
            itensor = self.sentinel
            sites = range(siteidx + 1) if domain == "L" else range(len(mps) - 1, siteidx - 1, -1)
            for i in sites:
                itensor = self._step(itensor, mps[i], self._mo(mpo, i), domain, mps_conj[i])
        elif method == "Enviro":
            itensor = self.read(domain, siteidx)
        else:
            if itensor is None:
                itensor = self.read(domain, siteidx + (-1 if domain == "L" else 1))
            mo = self._mo(mpo, siteidx)
            if canonical and not isinstance(mo, list) and mps_conj[siteidx] is None and hasattr(mpo, "device"):
                itensor = contract_one_site(itensor, mps[siteidx], mo, domain, canonical_mo_host=mpo[siteidx])
            else:
                itensor = self._step(itensor, mps[siteidx], mo, domain, mps_conj[siteidx])
            self.write(domain, siteidx, itensor)
        return itensor

    def write(self, domain, siteidx, tensor):
        self._virtual_disk[(domain, siteidx)] = tensor

    def drop(self, domain):
        """forget the environments of one direction (the sentinels stay)"""
        for key in [k for k, v in self._virtual_disk.items() if k[0] == domain and v is not self.sentinel]:
            del self._virtual_disk[key]

    def read(self, domain, siteidx):
        return self._virtual_disk[(domain, siteidx)]
