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
