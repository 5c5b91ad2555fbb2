"""Charge-transport drivers built on the sweep engine (counterpart of renormalizer/transport)."""
from .dynamics import ChargeDiffusionDynamics, InitElectron, EDGE_THRESHOLD, calc_r_square
from .spectral_function import SpectralFunctionZT
from .kubo import TransportKubo
