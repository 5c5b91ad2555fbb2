"""Carrier mobility from the Green-Kubo current autocorrelation function at finite temperature.

Counterpart of renormalizer/transport/kubo.py (``TransportKubo``):
C(t) = Tr{ rho(T) j(t) j(0) } with j = -i [P, H], P = sum_m R_m a+_m a_m.  rho(T)^(1/2) is reached from the
maximally entangled one-exciton density operator by imaginary-time propagation (``ThermalProp``); then
rho^(1/2) and j rho^(1/2) are propagated in real time side by side and C(t) = -<bra(t)| j |ket(t)> (j is kept
anti-Hermitian "real", hence the sign).  Hamiltonian terms a+_m a_n (m != n) enter the current with the factor
R_m - R_n; terms that also carry one vibrational coordinate (Peierls coupling) form a second, phonon-assisted
current operator and the four cross correlations are reported separately."""
import logging
import os

import numpy as np

from ..mps import Mpo, MpDm, ThermalProp
from ..mps.mps import BraKetPair
from ..mps.thermalprop import load_thermal_state
from ..utils import Quantity, CompressConfig, EvolveConfig
from ..utils.constant import mobility2au
from ..utils.tdmps import TdMpsJob

logger = logging.getLogger("renormalizer_amd")


class TransportKubo(TdMpsJob):
    def __init__(self, model, temperature: Quantity, distance_matrix: np.ndarray = None, insteps: int = 1,
                 ievolve_config=None, compress_config=None, evolve_config=None, dump_dir: str = None,
                 job_name: str = None, thermal_dump_path: str = None, properties=None):
        self.model = model
        self.distance_matrix = distance_matrix
        self.h_mpo = Mpo(model)
        self._construct_current_operator()
        if temperature == 0:
            raise ValueError("Can't set temperature to 0.")
        self.temperature = temperature
        if ievolve_config is None:
            self.ievolve_config = EvolveConfig()
            if insteps is None:
                self.ievolve_config.adaptive = True
                self.ievolve_config.guess_dt = temperature.to_beta() / 1e5j
                insteps = 1
        else:
            self.ievolve_config = ievolve_config
        self.insteps = insteps
        self.compress_config = CompressConfig() if compress_config is None else compress_config
        if thermal_dump_path is not None:
            self.thermal_dump_path = thermal_dump_path
        elif dump_dir is not None and job_name is not None:
            self.thermal_dump_path = os.path.join(dump_dir, job_name + "_impdm.npz")
        else:
            self.thermal_dump_path = None
        self.properties = properties
        self._auto_corr = []
        self._auto_corr_decomposition = []
        super().__init__(evolve_config=evolve_config, dump_dir=dump_dir, job_name=job_name)

    def _construct_current_operator(self):
        """split the inter-site electronic terms of H into the bare (a+_m a_n) and the phonon-assisted
        (a+_m a_n x) current, each weighted with R_m - R_n (kubo.py:140-216)"""
        model = self.model
        n = model.n_edofs
        if self.distance_matrix is None:
            # periodic chain with unit spacing
            self.distance_matrix = np.arange(n).reshape(-1, 1) - np.arange(n).reshape(1, -1)
            self.distance_matrix[0][-1] = 1
            self.distance_matrix[-1][0] = -1
        bare, assisted = [], []
        for op in model.ham_terms:
            e_pos = [k for k, dof in enumerate(op.dofs) if model.basis[model.dof_to_siteidx[dof]].is_electron]
            if len(e_pos) > 2:
                raise ValueError(f"The model contains three-electron (or more complex) operator {op}")
            if len(e_pos) < 2:
                continue
            k1, k2 = e_pos
            e1, e2 = model.e_dofs.index(op.dofs[k1]), model.e_dofs.index(op.dofs[k2])
            if e1 == e2:
                continue
            if len(op.dofs) not in (2, 3):
                raise NotImplementedError("Complex vibration potential not implemented")
            if len(op.dofs) == 3:
                assert op.split_symbol[3 - k1 - k2].replace(" ", "") in (r"b^\dagger+b", "x")
            s1, s2 = op.split_symbol[k1], op.split_symbol[k2]
            if {s1, s2} != {r"a^\dagger", "a"}:
                raise ValueError(f"Unknown symbol: {s1}, {s2}")
            factor = self.distance_matrix[e1][e2] if s1 == r"a^\dagger" else self.distance_matrix[e2][e1]
            (bare if len(op.dofs) == 2 else assisted).append(op * factor)
        self.j_oper = Mpo(model, bare)
        self.j_oper2 = Mpo(model, assisted) if assisted else None

    def init_mps(self):
        mpdm = None
        if self.thermal_dump_path is not None:
            mpdm = load_thermal_state(self.model, self.thermal_dump_path)
        if mpdm is None:
            i_mpdm = MpDm.max_entangled_ex(self.model)
            i_mpdm.compress_config = self.compress_config
            tp = ThermalProp(i_mpdm, evolve_config=self.ievolve_config, dump_dir=self.dump_dir,
                             job_name=None if self.job_name is None else self.job_name + "_thermal_prop")
            tp.evolve(None, self.insteps, self.temperature.to_beta() / 2j)
            mpdm = tp.latest_mps
            if self.thermal_dump_path is not None:
                mpdm.dump(self.thermal_dump_path)
        mpdm.compress_config = self.compress_config
        self.h_mpo = Mpo(self.model, offset=Quantity(mpdm.expectation(self.h_mpo)))
        mpdm.evolve_config = self.evolve_config
        ket = self.j_oper.contract(mpdm).normalize("mps_norm_to_coeff")
        bra = mpdm.copy()
        if self.j_oper2 is None:
            return BraKetPair(bra, ket, self.j_oper)
        ket2 = self.j_oper2.contract(mpdm).normalize("mps_norm_to_coeff")
        return BraKetPair(bra, ket, self.j_oper), BraKetPair(bra, ket2, self.j_oper2)

    def process_mps(self, mps):
        if self.j_oper2 is None:
            self._auto_corr.append(-mps.ft)
            if self.properties is not None:
                self.properties.calc_properties_braketpair(mps)
            return
        (bra, ket), (_, ket2) = mps
        parts = [-BraKetPair(bra, k, j).ft for j in (self.j_oper, self.j_oper2) for k in (ket, ket2)]
        self._auto_corr.append(sum(parts))
        self._auto_corr_decomposition.append(parts)

    def evolve_single_step(self, evolve_dt):
        if self.j_oper2 is None:
            prev_bra, prev_ket = self.latest_mps
            prev_ket2 = None
        else:
            (prev_bra, prev_ket), (_, prev_ket2) = self.latest_mps
        ket = prev_ket.evolve(self.h_mpo, evolve_dt)
        bra = prev_bra.evolve(self.h_mpo, evolve_dt)
        if self.j_oper2 is None:
            return BraKetPair(bra, ket, self.j_oper)
        ket2 = prev_ket2.evolve(self.h_mpo, evolve_dt)
        return BraKetPair(bra, ket, self.j_oper), BraKetPair(bra, ket2, self.j_oper2)

    def stop_evolve_criteria(self):
        """the last ten values of C(t) have decayed to 1e-5 of C(0) (kubo.py:288-294)"""
        corr = self.auto_corr
        if len(corr) < 10:
            return False
        last, first = corr[-10:], corr[0]
        return np.abs(last.mean()) < 1e-5 * np.abs(first) and last.std() < 1e-5 * np.abs(first)

    @property
    def auto_corr(self) -> np.ndarray:
        return np.array(self._auto_corr)

    @property
    def auto_corr_decomposition(self) -> np.ndarray:
        """columns <j1 j1>, <j1 j2>, <j2 j1>, <j2 j2> (first index: operator at time t)"""
        return np.array(self._auto_corr_decomposition)

    def get_dump_dict(self):
        return {"mol list": self.model.to_dict(), "temperature": self.temperature.as_au(),
                "time series": self.evolve_times, "auto correlation": self.auto_corr,
                "auto correlation decomposition": self.auto_corr_decomposition, "mobility": self.calc_mobility()[1],
                **({} if self.properties is None else dict(self.properties.prop_res))}

    def calc_mobility(self):
        """(mobility in a.u., in cm^2 / V s): trapezoid integral of Re C(t) over kT"""
        t = np.asarray(self.evolve_times, dtype=float)
        c = self.auto_corr.real
        integral = float(np.sum((c[1:] + c[:-1]) * np.diff(t)) / 2) if len(t) > 1 else 0.0
        mobility_in_au = integral / self.temperature.as_au()
        return mobility_in_au, mobility_in_au / mobility2au
