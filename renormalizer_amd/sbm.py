"""Spin-boson model construction: spectral densities, adiabatic renormalisation, bath discretisation.

Counterpart of renormalizer/sbm/lib.py (``SpectralDensityFunction`` / ``OhmicSDF`` :38-139, ``DebyeSDF`` :18-35,
``ColeDavidsonSDF`` :142-202, ``param2mollist`` :205-217) - the producer of BASELINE config 2's model.  Host-side
only; the dynamics are plain ``Mps.evolve`` calls (examples/sbm.py)."""
import logging

import numpy as np
import scipy.integrate

from .model import Phonon, SpinBosonModel
from .utils import Quantity

logger = logging.getLogger("renormalizer_amd")


class DebyeSpectralDensityFunction:
    r"""J(\omega) = 2 \lambda \omega \omega_c / (\omega^2 + \omega_c^2)"""

    def __init__(self, lamb, omega_c):
        self.lamb = lamb
        self.omega_c = omega_c

    def func(self, w):
        return 2.0 * self.lamb * w * self.omega_c / (w ** 2 + self.omega_c ** 2)


DebyeSDF = DebyeSpectralDensityFunction


class SpectralDensityFunction:
    r"""J(\omega) = \pi/2 \alpha \omega^s \omega_c^{1-s} e^{-\omega/\omega_c}"""

    def __init__(self, alpha, omega_c, s=1):
        self.alpha = alpha
        self.omega_c = omega_c.as_au() if isinstance(omega_c, Quantity) else omega_c
        self.s = s

    def func(self, w):
        return np.pi / 2.0 * self.alpha * w ** self.s * self.omega_c ** (1 - self.s) * np.exp(-w / self.omega_c)

    def reno(self, omega_l) -> float:
        """tunnelling renormalisation factor exp(-2/pi int_{omega_l}^{30 omega_c} J(w)/w^2 dw) (lib.py:51-59)"""
        val = scipy.integrate.quad(lambda x: self.func(x) / x ** 2, a=omega_l, b=self.omega_c * 30)[0]
        return float(np.exp(-val * 2 / np.pi))

    def _dos_wang1(self, nb, w):
        return (nb + 1) / self.omega_c * np.exp(-w / self.omega_c)

    def Wang1(self, nb):
        """Wang's first discretisation: modes placed at equal weight of the density (nb + 1) / omega_c e^{-w/omega_c}
        (lib.py:110-125)"""
        omega = np.array([-np.log(1.0 - float(j) / (nb + 1)) * self.omega_c for j in range(1, nb + 1)])
        return omega, 2.0 / np.pi * omega * self.func(omega) / self._dos_wang1(nb, omega)

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


OhmicSDF = SpectralDensityFunction


class ColeDavidsonSDF:
    r"""J(\omega) = \eta sin(\beta \theta) / (1 + \omega^2 / \omega_c^2)^{\beta / 2}, \theta = atan(\omega / \omega_c)
    (lib.py:142-202)"""

    def __init__(self, ita, omega_c, beta, omega_limit):
        self.ita, self.omega_c, self.beta, self.omega_limit = ita, omega_c, beta, omega_limit

    def func(self, w):
        theta = np.arctan(w / self.omega_c)
        return self.ita * np.sin(self.beta * theta) / (1 + w ** 2 / self.omega_c ** 2) ** (self.beta / 2)

    def reno(self, omega_l):
        val = scipy.integrate.quad(lambda x: self.func(x) / x ** 2, a=omega_l, b=omega_l * 1000)[0]
        return float(np.exp(-val * 2 / np.pi))

    def Wang1(self, nb, nsamples=int(1e7)):
        """modes where the cumulated density A J(w)/w passes the integers, A normalising it to nb + 1 over
        (0, omega_limit); located on a grid of ``nsamples`` points"""
        a = (nb + 1) / scipy.integrate.quad(lambda x: self.func(x) / x, a=0, b=self.omega_limit)[0]
        step = self.omega_limit / nsamples
        grid = np.linspace(step, self.omega_limit, nsamples)
        frac = (np.cumsum(a * self.func(grid) / grid) * step) % 1
        omega = grid[np.where(frac[1:] - frac[:-1] < 0)[0]]
        assert len(omega) == nb
        return omega, 2.0 / np.pi * omega * self.func(omega) / (a * self.func(omega) / omega)


def param2mollist(alpha: float, raw_delta: Quantity, omega_c: Quantity, renormalization_p: float, n_phonons: int):
    """lib.py:205-217: Ohmic bath, adiabatic renormalisation, ``n_phonons`` equidistant modes up to the cut-off, level
    counts picked per mode by ``Phonon.simplest_phonon``"""
    sdf = SpectralDensityFunction(alpha, omega_c, s=1)
    delta, max_omega = sdf.adiabatic_renormalization(raw_delta, renormalization_p)
    omega_list, dis_list = sdf.post_process(*sdf.trapz(n_phonons, 0.0, max_omega))
    ph_list = [Phonon.simplest_phonon(o, d) for o, d in zip(omega_list, dis_list)]
    return SpinBosonModel(Quantity(0), Quantity(delta), ph_list)


def param2model(alpha, raw_delta, omega_c, renormalization_p, n_phonons, n_phys_dim):
    """lib.py:205-217 with a fixed number of phonon levels per mode (``Phonon.simple_phonon``)."""
    sdf = SpectralDensityFunction(alpha, omega_c, s=1)
    delta, max_omega = sdf.adiabatic_renormalization(raw_delta, renormalization_p)
    omega, c2 = sdf.trapz(n_phonons, 0.0, max_omega)
    omega_list, dis_list = sdf.post_process(omega, c2)
    ph_list = [Phonon.simple_phonon(o, d, n_phys_dim) for o, d in zip(omega_list, dis_list)]
    return SpinBosonModel(Quantity(0), Quantity(delta), ph_list), delta
