"""Unit conversion factors (CODATA via scipy.constants); mirrors the names of
renormalizer/utils/constant.py so user scripts keep working."""
from scipy.constants import physical_constants as _c

au2ev = _c["Hartree energy in eV"][0]
ev2au = 1.0 / au2ev
cm2au = 1.0e2 * _c["inverse meter-hertz relationship"][0] / _c["hartree-hertz relationship"][0]
au2cm = 1.0 / cm2au
cm2ev = cm2au * au2ev
ev2cm = 1.0 / cm2ev
fs2au = 1.0e-15 / _c["atomic unit of time"][0]
au2fs = 1.0 / fs2au
K2au = _c["kelvin-hartree relationship"][0]
au2K = _c["hartree-kelvin relationship"][0]
# 1 cm^2 / (V s) in atomic units of mobility (e a0^2 / hbar)
mobility2au = au2ev * _c["atomic unit of time"][0] / (_c["atomic unit of length"][0] * 100) ** 2
