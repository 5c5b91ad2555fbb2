"""renormalizer_amd - MI355X-native MPS sweep engine behind Renormalizer's API surface.

The per-site hot path of TDVP (``Mps.evolve``) and DMRG (``optimize_mps``) runs as
hand-written HIP for gfx950 in ``csrc/libmpsengine.so`` (C ABI: include/mpsengine.h),
driven from Python through ctypes.  There is no CPU fallback in this package.
"""
__version__ = "0.1.0"
