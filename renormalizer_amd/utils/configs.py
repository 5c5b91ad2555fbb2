"""Configuration objects read by the sweep drivers.

Kept API surface of renormalizer/utils/configs.py (CompressConfig :41-264, OptimizeConfig
:267-300, EvolveConfig :342-416): same attribute names and defaults for everything the
hot path consumes (compute_m_trunc, procedure, method, ivp_solver...)."""
from enum import Enum

import numpy as np

from .rk import RungeKutta, TaylorExpansion


class CompressCriteria(Enum):
    threshold = "threshold"
    fixed = "fixed"
    both = "both"


class OFS(Enum):
    """criterion for swapping the two centre sites of a two-site step on the fly (utils/configs.py:27-38)"""
    ofs_s = "OFS-S"            # smaller entanglement entropy
    ofs_ds = "OFS-D/S"         # discarded weight, entropy when nothing is discarded
    ofs_d = "OFS-D"            # smaller discarded weight
    ofs_debug = "OFS-Debug"    # evaluate both orders, never swap


class CompressConfig:
    def __init__(self, criteria=CompressCriteria.threshold, threshold: float = 1e-3, max_bonddim: int = 32,
                 vmethod: str = "2site", vprocedure=None, vrtol=1e-5, vguess_m=(5, 5), ofs: OFS = None,
                 ofs_swap_jw: bool = False):
        if isinstance(criteria, str):
            criteria = getattr(CompressCriteria, criteria)
        if not isinstance(criteria, CompressCriteria):
            raise ValueError(f"Unknown compress criteria {criteria}")
        self.criteria = criteria
        self._threshold = None
        self.threshold = threshold
        self.bond_dim_max_value = max_bonddim
        # length nsite + 1 (terminal bonds included), filled by set_bonddim
        self.max_dims = None
        self.vmethod = vmethod
        if vprocedure is None:
            head = [1.0, 0.7, 0.5, 0.3, 0.1] if vmethod == "1site" else [0.5, 0.3, 0.1]
            vprocedure = [[max_bonddim, p] for p in head] + [[max_bonddim, 0]] * 10
        self.vprocedure = vprocedure
        self.vrtol = vrtol
        self.vguess_m = vguess_m
        self.ofs = ofs
        self.ofs_swap_jw = ofs_swap_jw

    @property
    def threshold(self):
        return self._threshold

    @threshold.setter
    def threshold(self, v):
        if v <= 0:
            raise ValueError("non-positive threshold")
        if v == 1:
            raise ValueError("1 is an ambiguous threshold")
        if v > 1:
            raise ValueError("Can't set threshold to be larger than 1")
        self._threshold = v

    @property
    def bonddim_should_set(self):
        return self.criteria is not CompressCriteria.threshold and self.max_dims is None

    def set_bonddim(self, length):
        if self.max_dims is None:
            self.max_dims = np.full(length, self.bond_dim_max_value, dtype=int)

    def _threshold_m_trunc(self, sigma):
        sigma = np.asarray(sigma)
        return int(np.sum(sigma / np.linalg.norm(sigma) > self.threshold))

    def _fixed_m_trunc(self, sigma, idx, left):
        bond_idx = idx + 1 if left else idx
        return int(min(self.max_dims[bond_idx], len(sigma)))

    def compute_m_trunc(self, sigma, idx, left):
        """configs.py:207-219"""
        if self.criteria is CompressCriteria.threshold:
            return self._threshold_m_trunc(sigma)
        if self.criteria is CompressCriteria.fixed:
            return self._fixed_m_trunc(sigma, idx, left)
        return min(self._threshold_m_trunc(sigma), self._fixed_m_trunc(sigma, idx, left))

    def copy(self):
        new = self.__class__.__new__(self.__class__)
        new.__dict__ = self.__dict__.copy()
        if self.max_dims is not None:
            new.max_dims = self.max_dims.copy()
        return new


class OptimizeConfig:
    def __init__(self, procedure=None):
        if procedure is None:
            procedure = [[10, 0.4], [20, 0.2], [30, 0.1], [40, 0], [40, 0]]
        self.procedure = procedure
        self.method = "2site"
        self.algo = "davidson"
        self.nroots = 1
        self.e_rtol = 1e-6
        self.e_atol = 1e-8
        self.inverse = 1.0

    def copy(self):
        new = self.__class__.__new__(self.__class__)
        new.__dict__ = self.__dict__.copy()
        new.procedure = [list(p) for p in self.procedure]
        return new


class EvolveMethod(Enum):
    prop_and_compress = "P&C"
    prop_and_compress_tdrk4 = "P&C TDRK4"
    prop_and_compress_tdrk = "P&C TDRK"
    tdvp_mu_vmf = "TDVP Matrix Unfolding VMF"
    tdvp_vmf = "TDVP VMF"
    tdvp_mu_cmf = "TDVP Matrix Unfolding CMF"
    tdvp_ps = "TDVP PS"
    tdvp_ps2 = "TDVP PS2"


class EvolveConfig:
    def __init__(self, method: EvolveMethod = EvolveMethod.prop_and_compress, adaptive=False, guess_dt=1e-1,
                 adaptive_rtol=5e-4, taylor_order: int = None, rk_solver="C_RK4", reg_epsilon=1e-10, ivp_rtol=1e-5,
                 ivp_atol=1e-8, ivp_solver="krylov", force_ovlp=True):
        if isinstance(method, str):
            method = EvolveMethod[method]
        self.method = method
        self.rk_config = RungeKutta(rk_solver)
        self.adaptive = adaptive
        self.guess_dt = guess_dt
        self.adaptive_rtol = adaptive_rtol
        # utils/configs.py:364-368: one order more when the last Taylor term serves as the error estimate
        self.taylor_order = (5 if adaptive else 4) if taylor_order is None else taylor_order
        self.taylor_config = TaylorExpansion(self.taylor_order)
        self.reg_epsilon = reg_epsilon
        self.ivp_rtol = ivp_rtol
        self.ivp_atol = ivp_atol
        self.ivp_solver = ivp_solver
        self.force_ovlp = force_ovlp
        self.vmf_auto_switch = True          # tdvp_mu_vmf <-> tdvp_vmf by the smallest singular value (mps.py:1078-1090)
        self.tdvp_cmf_midpoint = True
        self.tdvp_cmf_c_trapz = False
        self.stat = None

    @property
    def is_tdvp(self):
        return self.method not in (EvolveMethod.prop_and_compress, EvolveMethod.prop_and_compress_tdrk4,
                                   EvolveMethod.prop_and_compress_tdrk)

    def check_valid_dt(self, evolve_dt):
        """utils/configs.py:394-402: the guessed and the requested step must be both real or both imaginary-time
        and point the same way"""
        info = f"in config: {self.guess_dt}, in arg: {evolve_dt}"
        if bool(np.iscomplex(evolve_dt)) ^ bool(np.iscomplex(self.guess_dt)):
            raise ValueError("real and imag not compatible. " + info)
        if (np.iscomplex(evolve_dt) and np.imag(evolve_dt) * np.imag(self.guess_dt) < 0) or \
                (not np.iscomplex(evolve_dt) and evolve_dt * self.guess_dt < 0):
            raise ValueError("evolve into wrong direction. " + info)

    def copy(self):
        new = self.__class__.__new__(self.__class__)
        new.__dict__ = self.__dict__.copy()
        return new
