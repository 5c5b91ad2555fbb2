"""Spin-boson model construction: Ohmic spectral density, adiabatic renormalisation, bath discretisation.

Host-side counterpart of renormalizer/sbm/lib.py (SpectralDensityFunction :38-137, param2mollist :205-217)
- the producer of the model for BASELINE config 2.  No arithmetic of the sweep lives here."""
import numpy as np
import scipy.integrate

from .model import Phonon, SpinBosonModel
from .utils import Quantity


class SpectralDensityFunction:
    r"""J(\omega) = \pi/2 \alpha \omega^s \omega_c^{1-s} e^{-\omega/\omega_c}"""

    def __init__(self, alpha, omega_c, s=1):
        self.alpha = alpha
        self.omega_c = omega_c.as_au() if isinstance(omega_c, Quantity) else omega_c
        self.s = s

    def func(self, w):
        return np.pi / 2.0 * self.alpha * w ** self.s * self.omega_c ** (1 - self.s) * np.exp(-w / self.omega_c)

    def adiabatic_renormalization(self, delta, p):
        """Self-consistent tunnelling renormalisation with cut-off omega_l = p*delta (lib.py:61-84)."""
        delta = delta.as_au() if isinstance(delta, Quantity) else delta
        re = 1.0
        for _ in range(50):
            re_old = re
            omega_l = delta * re * p
            val = scipy.integrate.quad(lambda x: self.func(x) / x ** 2, a=omega_l, b=self.omega_c * 30)[0]
            re = np.exp(-val * 2 / np.pi)
            if np.allclose(re, re_old):
                break
        return delta * re, delta * re * p

    def trapz(self, nb, x0, x1):
        """Equidistant discretisation, couplings from the trapezoid rule (lib.py:127-137)."""
        dw = (x1 - x0) / float(nb)
        x = [x0 + i * dw for i in range(nb + 1)]
        omega = np.array([(x[i] + x[i + 1]) / 2.0 for i in range(nb)])
        c2 = np.array([(self.func(x[i]) + self.func(x[i + 1])) / 2 for i in range(nb)]) * 2.0 / np.pi * omega * dw
        return omega, c2

    @staticmethod
    def post_process(omega, c2, ifsort=True):
        dis = np.sqrt(c2) / omega ** 2
        idx = np.argsort(c2 / omega)[::-1] if ifsort else np.arange(len(omega))
        return [Quantity(omega[i]) for i in idx], [Quantity(dis[i]) for i in idx]


def param2model(alpha, raw_delta, omega_c, renormalization_p, n_phonons, n_phys_dim):
    """lib.py:205-217 with a fixed number of phonon levels per mode (``Phonon.simple_phonon``)."""
    sdf = SpectralDensityFunction(alpha, omega_c, s=1)
    delta, max_omega = sdf.adiabatic_renormalization(raw_delta, renormalization_p)
    omega, c2 = sdf.trapz(n_phonons, 0.0, max_omega)
    omega_list, dis_list = sdf.post_process(omega, c2)
    ph_list = [Phonon.simple_phonon(o, d, n_phys_dim) for o, d in zip(omega_list, dis_list)]
    return SpinBosonModel(Quantity(0), Quantity(delta), ph_list), delta
