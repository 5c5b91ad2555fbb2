"""renormalizer_amd - MI355X-native MPS sweep engine behind Renormalizer's API surface.

The per-site hot path of TDVP (``Mps.evolve``) and DMRG (``optimize_mps``) runs as
hand-written HIP for gfx950 in ``csrc/libmpsengine.so`` (C ABI: include/mpsengine.h),
driven from Python through ctypes.  There is no CPU fallback in this package.
"""
__version__ = "0.1.0"

from .model import (Op, OpSum, Model, HolsteinModel, SpinBosonModel, Phonon, Mol, BasisSHO, BasisHalfSpin,
                    BasisSimpleElectron, BasisMultiElectron, BasisMultiElectronVac)
from .utils import Quantity, CompressConfig, CompressCriteria, EvolveConfig, EvolveMethod, OptimizeConfig
from .mps.mpo import Mpo


def __getattr__(name):
    # Mps / backend touch the GPU engine on import of their module; load them lazily so that the
    # host-side model / MPO layer stays importable on machines without the HIP library.
    if name == "Mps":
        from .mps.mps import Mps
        return Mps
    if name == "optimize_mps":
        from .mps.gs import optimize_mps
        return optimize_mps
    if name in ("MpDm", "thermal_state"):
        from . import mps as _mps
        return getattr(_mps, name)
    if name == "backend":
        from .mps.backend import backend
        return backend
    raise AttributeError(name)
