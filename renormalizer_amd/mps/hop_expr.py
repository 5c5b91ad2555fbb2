"""Effective-Hamiltonian matvec closures (counterpart of renormalizer/mps/hop_expr.py:7-117).

``hop_expr(l, r, cmo, cshape)`` returns a callable ``hop(C) -> H C`` on device tensors;
the callable also carries the C-ABI descriptor (``mpse_heff``) so that the Lanczos / Davidson
drivers can hand the whole solve to the engine without coming back to Python per matvec."""
import ctypes as C

from ..engine import DeviceTensor, get_engine, mpse_heff


class Hop:
    def __init__(self, ltensor, rtensor, cmo, cshape):
        self.eng = get_engine()
        eng = self.eng
        nsite = len(cmo)
        cshape = tuple(int(s) for s in cshape)
        ancilla = (nsite > 0) and (2 * nsite + 2 == len(cshape))
        if not ancilla:
            assert nsite + 2 == len(cshape)
        self.nsite = nsite
        self.cshape = cshape
        self.l = eng.asdevice(ltensor)
        self.r = eng.asdevice(rtensor)
        self.cmo = [eng.asdevice(w) for w in cmo]
        if nsite == 2 and self.cmo[0].is_complex != self.cmo[1].is_complex:
            self.cmo = [w.to_complex() for w in self.cmo]
        h = mpse_heff()
        h.nsite = nsite
        d = h.dims
        d.Dl_ket, d.Dr_ket = cshape[0], cshape[-1]
        # rows of the environments: equal to the ket bonds for H itself, the bonds of another state when H C is
        # projected onto it (variational compression); the result then carries those bonds
        d.Dl_bra, d.Dr_bra = self.l.shape[0], self.r.shape[0]
        assert self.l.shape[2] == cshape[0] and self.r.shape[2] == cshape[-1], (self.l.shape, self.r.shape, cshape)
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
        eng._check(eng.lib.mpse_heff_apply(eng.ctx, c.code, C.byref(self.heff), c.ptr, out.ptr))
        return out


def hop_expr(ltensor, rtensor, cmo, cshape, twolayer: bool = False):
    if twolayer:
        raise NotImplementedError("two-layer (H - omega)^2 effective Hamiltonians are not implemented")
    return Hop(ltensor, rtensor, list(cmo), cshape)
