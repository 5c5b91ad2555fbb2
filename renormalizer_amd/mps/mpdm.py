"""Matrix product density operators: purified finite-temperature states with an ancilla leg per site.

Counterpart of renormalizer/mps/mpdm.py.  A site is (D_l, d_up, d_down, D_r); the lower leg is the ancilla that
every environment update / effective-Hamiltonian matvec traces over as a batch index (mps/lib.py:207-211, 239-243;
mps/hop_expr.py:87-91, 111-115 - the ``danc`` variants of the engine's contraction plans).  Everything else
(canonical forms, compression, TDVP-PS / P&C drivers, observables) is inherited from ``Mps``: the sweep code only
ever sees "left bond x physical legs x right bond"."""
import numpy as np

from ..engine import get_engine
from .mpo import Mpo
from .mps import Mps
from .svd_qn import add_outer


class MpDm(Mps):
    is_mps, is_mpo, is_mpdm = False, False, True

    @classmethod
    def random(cls, *args, **kwargs):
        raise ValueError("MpDm don't have to produce random state")            # mpdm.py:17-20

    @classmethod
    def ground_state(cls, *args, **kwargs):
        raise ValueError("Use max_entangled_ex or max_entangled_gs for matrix product density matrix")

    @classmethod
    def from_mps(cls, mps: Mps) -> "MpDm":
        """|psi> -> operator with the amplitudes on the diagonal of the two physical legs (mpdm.py:28-47)"""
        eng = get_engine()
        new = cls()
        new.__dict__.update(mps.metacopy().__dict__)
        sites = []
        for ms in mps:
            a = ms.to_host()
            mo = np.zeros((a.shape[0], a.shape[1], a.shape[1], a.shape[2]), dtype=a.dtype)
            for k in range(a.shape[1]):
                mo[:, k, k, :] = a[:, k, :]
            sites.append(eng.asdevice(mo))
        new._mp = sites
        new.compress_config = mps.compress_config.copy()
        return new

    @classmethod
    def max_entangled_ex(cls, model, normalize=True) -> "MpDm":
        """T = infinity state of the one-exciton space: sum_i a_i^+ applied to the maximally entangled
        vibrational state (mpdm.py:54-66)"""
        mps = Mps.ground_state(model, max_entangled=True)
        ex_mps = Mpo.onsite(model, r"a^\dagger").apply(mps)
        if normalize:
            ex_mps.normalize("mps_and_coeff")
        return cls.from_mps(ex_mps)

    @classmethod
    def max_entangled_gs(cls, model) -> "MpDm":
        return cls.from_mps(Mps.ground_state(model, max_entangled=True))

    def _get_sigmaqn(self, idx):
        """quantum numbers of the (up, down) leg pair: only the physical (upper) leg carries charge (mpdm.py:71-74)"""
        up = np.asarray(self.model.basis[idx].sigmaqn)
        return add_outer(up, np.zeros_like(up))

    def todense(self) -> np.ndarray:
        """the operator as a (prod d) x (prod d) matrix, upper legs = rows (mpdm.py:85-87); small systems only"""
        t = np.ones((1, 1, 1), dtype=complex if self.is_complex else float)
        for ms in self:
            a = ms.to_host()
            t = np.tensordot(t, a, axes=([2], [0]))              # rows, cols, d_up, d_down, D_r
            r, c, du, dd, dr = t.shape
            t = t.transpose(0, 2, 1, 3, 4).reshape(r * du, c * dd, dr)
        return t[:, :, 0] * self.coeff

    def apply(self, mp, canonicalise: bool = False) -> "MpDm":
        """rho @ O: the operator acts on the lower (ancilla-side) legs, site = einsum("apqb,cqrd->acprbd")
        (mpdm.py:130-165).  One strided GEMM per (operator channel, upper index), batched over the left bond."""
        from ..engine import idx1, idx2
        eng = get_engine()
        assert not getattr(mp, "is_mps", False) and len(mp) == len(self)
        new = self.metacopy()
        cplx = self.is_complex or mp.is_complex
        if cplx:
            new.dtype = np.dtype(np.complex128)     # mpdm.py:140-141 to_complex(inplace=True)
        for i, ms in enumerate(self):
            w = mp.device(i, eng)
            if cplx:
                ms = ms.to_complex()
            dl, dp, dq, dr = ms.shape
            wl, wq, wd, wr = w.shape
            assert dq == wq
            out = eng.empty((dl * wl, dp, wd, dr * wr), np.complex128 if cplx else np.float64)
            for c in range(wl):
                for p in range(dp):
                    eng.gemm(w.row_block(c, c + 1), ms.shifted(p * dq * dr), out.shifted((c * dp + p) * wd * dr * wr),
                             idx1(wd * wr, 1), idx1(wq, wd * wr), idx1(dq, dr), idx1(dr, 1),
                             idx2(wd, wr, dr * wr, 1), idx1(dr, wr), batch=dl, sb_a=0, sb_b=dp * dq * dr,
                             sb_c=wl * dp * wd * dr * wr)
            new._mp[i] = out
        q = len(np.atleast_1d(self.qntot))
        # the operator's bonds carry no labels here (the reference uses dummy quantum numbers as well)
        new.qn = [np.repeat(np.asarray(qs).reshape(-1, q), nb, axis=0) for qs, nb in zip(self.qn, mp.bond_dims)]
        if canonicalise:
            new.canonicalise()
        return new

    def evolve_exact(self, h_mpo, evolve_dt, space):
        """rho exp(-i H dt) for the local vibrational Hamiltonian: the density operator is applied ON the
        propagator, unlike the pure-state method (mpdm.py:76-83)"""
        offset = getattr(h_mpo, "offset", 0.0)
        offset = offset.as_au() if hasattr(offset, "as_au") else float(offset)
        prop = Mpo.exact_propagator(self.model, -1.0j * evolve_dt, space=space, shift=-offset)
        new = self.apply(prop, canonicalise=True)
        new.coeff = new.coeff * np.exp(-1.0j * offset * evolve_dt)
        return new
