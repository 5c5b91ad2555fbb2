"""Effective-Hamiltonian matvec closures (counterpart of renormalizer/mps/hop_expr.py:7-117).

``hop_expr(l, r, cmo, cshape)`` returns a callable ``hop(C) -> H C`` on device tensors;
the callable also carries the C-ABI descriptor (``mpse_heff``) so that the Lanczos / Davidson
drivers can hand the whole solve to the engine without coming back to Python per matvec."""
import ctypes as C

import numpy as np

from ..engine import DeviceTensor, get_engine, mpse_heff


class Hop:
    def __init__(self, ltensor, rtensor, cmo, cshape, twolayer=False):
        self.eng = get_engine()
        self.twolayer = bool(twolayer)
        eng = self.eng
        nsite = len(cmo)
        cshape = tuple(int(s) for s in cshape)
        ancilla = (nsite > 0) and (2 * nsite + 2 == len(cshape))
        if not ancilla:
            assert nsite + 2 == len(cshape)
        self.nsite = nsite
        self.cshape = cshape
        self.cmask = None          # optional structural tile mask of the centre (centre_tile_mask), used by expm_krylov
        self.l = eng.asdevice(ltensor)
        self.r = eng.asdevice(rtensor)
        if self.twolayer:
            # (H - omega)^2: L (Dl, wl, wl, Dl), R (Dr, wr, wr, Dr), the same MPO sites in both layers, no ancilla
            # (hop_expr.py:24-52); the engine's batch index next to the right bond serves the dense direct solver
            assert nsite in (1, 2) and self.l.ndim == 4 and self.r.ndim == 4
            assert self.l.shape[1] == self.l.shape[2] and self.r.shape[1] == self.r.shape[2]
        self.cmo = [eng.asdevice(w) for w in cmo]
        if nsite == 1 and isinstance(cmo[0], np.ndarray):
            eng.mpo_site_hint(self.cmo[0], cmo[0])
        if nsite == 2 and self.cmo[0].is_complex != self.cmo[1].is_complex:
            self.cmo = [w.to_complex() for w in self.cmo]
        h = mpse_heff()
        h.nsite = nsite
        d = h.dims
        d.Dl_ket, d.Dr_ket = cshape[0], cshape[-1]
        # rows of the environments: equal to the ket bonds for H itself, the bonds of another state when H C is
        # projected onto it (variational compression); the result then carries those bonds
        d.Dl_bra, d.Dr_bra = self.l.shape[0], self.r.shape[0]
        assert self.l.shape[-1] == cshape[0] and self.r.shape[-1] == cshape[-1], (self.l.shape, self.r.shape, cshape)
        self.oshape = (self.l.shape[0],) + cshape[1:-1] + (self.r.shape[0],)
        self.square = self.oshape == cshape
        d.danc = cshape[2] if ancilla else 1
        d.danc1 = cshape[4] if (ancilla and nsite == 2) else 0          # neighbouring sites may differ in size
        d.wl, d.wr = self.l.shape[1], self.r.shape[1]
        d.d0 = self.cmo[0].shape[1] if nsite >= 1 else 1
        d.d1 = self.cmo[1].shape[1] if nsite == 2 else 1
        d.wm = self.cmo[0].shape[3] if nsite == 2 else 1
        h.L, h.l_dtype, h.R, h.r_dtype = self.l.ptr, self.l.code, self.r.ptr, self.r.code
        h.l_unit, h.r_unit = self.l.unit, self.r.unit
        if nsite >= 1:
            h.W0, h.w_dtype = self.cmo[0].ptr, self.cmo[0].code
        if nsite == 2:
            h.W1 = self.cmo[1].ptr
        self.heff = h
        self.operator_is_complex = self.l.is_complex or self.r.is_complex or any(w.is_complex for w in self.cmo)

    def __call__(self, c: DeviceTensor) -> DeviceTensor:
        eng = self.eng
        c = eng.asdevice(c)
        if self.operator_is_complex and not c.is_complex:
            c = c.to_complex()
        c = c.reshape(self.cshape)
        out = eng.empty(self.oshape, c.dtype)
        apply = eng.lib.mpse_heff_apply2 if self.twolayer else eng.lib.mpse_heff_apply
        eng._check(apply(eng.ctx, c.code, C.byref(self.heff), c.ptr, out.ptr))
        return out

    def dense(self):
        """The projected operator as a matrix H[(out indices), (in indices)] on the host (get_ham_direct,
        mps/gs.py:307-369): applied to all columns of the identity in ONE call - the unit vectors ride on the batch
        index next to the right bond (the ancilla slot of the contraction plans)."""
        assert self.square and (self.heff.dims.danc == 1 or self.nsite == 0)
        eng = self.eng
        n = int(np.prod(self.cshape))
        dt = np.complex128 if self.operator_is_complex else np.float64
        eye = np.eye(n, dtype=dt).reshape(self.cshape + self.cshape)          # [in indices..., column]
        # columns -> batch index placed before the right bond: (Dl, d.., z, Dr)
        nd = len(self.cshape)
        x = np.moveaxis(eye.reshape(self.cshape + (n,)), -1, nd - 1)
        h = mpse_heff()
        C.memmove(C.byref(h), C.byref(self.heff), C.sizeof(mpse_heff))
        if self.nsite == 1:
            h.dims.danc = n
        else:
            h.dims.danc, h.dims.danc1 = 1, n
        xin = eng.asdevice(np.ascontiguousarray(x))
        out = eng.empty(xin.shape, dt)
        apply = eng.lib.mpse_heff_apply2 if self.twolayer else eng.lib.mpse_heff_apply
        eng._check(apply(eng.ctx, xin.code, C.byref(h), xin.ptr, out.ptr))
        res = np.moveaxis(out.to_host(), nd - 1, -1)
        return res.reshape(n, n)


def hop_expr(ltensor, rtensor, cmo, cshape, twolayer: bool = False):
    return Hop(ltensor, rtensor, list(cmo), cshape, twolayer)


_CMASK_CACHE = {}


def centre_tile_mask(eng, qnbigl, qnbigr, qntot, cshape):
    """Tile-occupancy pattern of a one-site centre tensor from its quantum numbers, for ``mpse_expm_centre_mask``:
    entry (a, sigma, b) can be non-zero only where ``qnbigl[a, sigma] + qnbigr[b] == qntot`` (the rule the block
    decompositions use, mps/mp.py:308-352).  Returns a device byte array in the layout of include/mpsengine.h - flag
    [tn][kt] for rows a in [16 kt, 16 kt + 16) and columns (sigma, b) in [64 tn, 64 tn + 64) - or None when the
    quantum numbers do not describe this shape.  Cached per quantum-number pattern (sites keep theirs from step to
    step)."""
    Dl, Dr = int(cshape[0]), int(cshape[-1])
    inner = int(np.prod(cshape[1:-1]))
    ql = np.ascontiguousarray(np.asarray(qnbigl).reshape(-1, np.asarray(qnbigl).shape[-1]))
    qr = np.ascontiguousarray(np.asarray(qnbigr).reshape(-1, np.asarray(qnbigr).shape[-1]))
    qt = np.asarray(qntot).reshape(-1)
    if ql.shape[0] != Dl * inner or qr.shape[0] != Dr:
        return None
    key = (id(eng), Dl, inner, Dr, ql.tobytes(), qr.tobytes(), qt.tobytes())
    hit = _CMASK_CACHE.get(key)
    if hit is not None:
        return hit
    # allowed[(a, sigma), b]: compare through the distinct quantum numbers (a handful) instead of row by row
    ul, il = np.unique(ql, axis=0, return_inverse=True)
    ur, ir = np.unique(qr, axis=0, return_inverse=True)
    ok = (ul[:, None, :] + ur[None, :, :] == qt).all(-1)                    # (distinct left, distinct right)
    allowed = ok[np.asarray(il).reshape(-1)][:, np.asarray(ir).reshape(-1)]  # (Dl * inner, Dr)
    n = inner * Dr
    view = allowed.reshape(Dl, n)
    nkt, ntn = (Dl + 15) // 16, (n + 63) // 64
    pad = np.zeros((nkt * 16, ntn * 64), dtype=bool)
    pad[:Dl, :n] = view
    flags = pad.reshape(nkt, 16, ntn, 64).any(axis=(1, 3))                  # (kt, tn)
    nkw = (nkt + 7) // 8
    out = np.zeros((ntn, nkw * 8), dtype=np.uint8)
    out[:, :nkt] = flags.T
    dev = eng.asdevice(out.view(np.float64).reshape(-1))
    if len(_CMASK_CACHE) > 4096:
        _CMASK_CACHE.clear()
    _CMASK_CACHE[key] = dev
    return dev
