"""One-particle retarded Green's function of a translationally invariant chain at zero temperature.

Counterpart of renormalizer/transport/spectral_function.py (``SpectralFunctionZT``): i G_ij(t) = <0| c_i(t) c+_j |0>
with j the first electronic degree of freedom; the ket c+_j |0> is propagated, the bra stays the (phonon) vacuum, and
every step evaluates the transition matrix elements <0| c_i |ket(t)> for all i with shared environments.  Finite
temperature goes through thermofield-doubled models (model/thermofield.py; transport/tests/test_spectral_function.py).
The dump holds G_ij(t) ("G array"), its lattice Fourier transform G_k(t) ("Gk array") and the populations."""
import numpy as np

from ..mps import Mpo, Mps
from ..utils import Quantity, CompressConfig, EvolveConfig
from ..utils.tdmps import TdMpsJob


class SpectralFunctionZT(TdMpsJob):
    def __init__(self, model, compress_config: CompressConfig = None, evolve_config: EvolveConfig = None,
                 dump_dir: str = None, job_name: str = None):
        self.model = model
        self.compress_config = CompressConfig() if compress_config is None else compress_config
        self._G_array = []
        self.e_occupations_array = []
        self.temperature = Quantity(0)
        super().__init__(evolve_config=evolve_config, dump_dir=dump_dir, job_name=job_name)

    @property
    def G_array(self):
        """G_ij(t): first index time, second |i - j|"""
        return np.array(self._G_array)

    def init_mps(self):
        creation = Mpo.onsite(self.model, r"a^\dagger", dof_set={self.model.e_dofs[0]})
        gs = Mps.ground_state(self.model, False)
        self.h_mpo = Mpo(self.model, offset=Quantity(gs.expectation(Mpo(self.model))))
        a_ket = creation.apply(gs, canonicalise=True)
        a_ket.compress_config = self.compress_config
        a_ket.evolve_config = self.evolve_config
        a_ket.normalize("mps_norm_to_coeff")
        if self.evolve_config.is_tdvp:
            a_ket = a_ket.expand_bond_dimension(self.h_mpo)
        return gs, a_ket

    def process_mps(self, mps):
        key = "a"
        if key not in self.model.mpos:
            self.model.mpos[key] = [Mpo.onsite(self.model, "a", dof_set={dof}) for dof in self.model.e_dofs]
        bra, ket = mps
        self._G_array.append(np.asarray(ket.expectations(self.model.mpos[key], bra.conj())) / 1j)
        self.e_occupations_array.append(ket.e_occupations)

    def evolve_single_step(self, evolve_dt):
        bra, ket = self.latest_mps
        return bra, ket.evolve(self.h_mpo, evolve_dt)

    def get_dump_dict(self):
        ne = self.model.n_edofs
        ka = (np.arange(ne // 2 + 1) * (2 * np.pi / ne)).reshape(1, 1, -1)
        ijdiff = np.arange(ne).reshape(1, -1, 1)
        return {"temperature": self.temperature.as_au(), "time series": self.evolve_times, "G array": self.G_array,
                "Gk array": np.sum(self.G_array.reshape(-1, ne, 1) * np.exp(1j * ka * ijdiff), axis=1),
                "electron occupations array": self.e_occupations_array}
