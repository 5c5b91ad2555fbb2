"""Vibrational mode / molecule parameter holders (renormalizer/model/phonon.py, mol.py)."""
import numpy as np

from ..utils import Quantity


class Phonon:
    """omega = [ground, excited] frequencies, displacement = [ground, excited] equilibrium positions."""

    def __init__(self, omega, displacement, n_phys_dim=None):
        self.omega = [o.as_au() for o in omega]
        self.dis = [d.as_au() for d in displacement]
        self.n_phys_dim = int(n_phys_dim)

    @classmethod
    def simple_phonon(cls, omega, displacement, n_phys_dim):
        return cls([omega, omega], [Quantity(0), displacement], n_phys_dim)

    @classmethod
    def simplest_phonon(cls, omega, displacement, temperature: Quantity = Quantity(0), lam: bool = False, max_pdim=128):
        """Number of levels chosen automatically (model/phonon.py:31-60): halve / double a trial dimension until the
        ground state of the displaced oscillator is resolved (weight of the upper half of the levels < 1e-4, last
        coefficient < 1e-3), plus 10 kT / omega thermal levels.  ``lam=True``: the second argument is the
        reorganisation energy instead of the displacement."""
        if lam:
            displacement = Quantity(np.sqrt(2 * displacement.as_au()) / omega.as_au())
        pdim = 256
        while True:
            gs = cls.simple_phonon(omega, displacement, pdim).get_displacement_evecs()[:, 0]
            if not (np.all(gs <= 1e-8) or np.all(gs >= -1e-8)):
                raise ValueError("displaced-oscillator ground state changes sign")
            if 0.9999 < gs[:len(gs) // 2].sum() / gs.sum():
                pdim //= 2                      # too many levels
            elif 0.001 < abs(gs[-1]):
                if pdim == 256:
                    raise ValueError(f"Too many phonon level required. omega: {omega}. displacement: {displacement}")
                pdim *= 2                       # one halving too far
                break
            else:
                break
        pdim = min(pdim + int(temperature.as_au() * 10 / omega.as_au()), max_pdim)
        return cls.simple_phonon(omega, displacement, pdim)

    def get_displacement_evecs(self) -> np.ndarray:
        """eigenvectors of b^+ b - g (b^+ + b) in the first n_phys_dim number states (model/phonon.py:83-94)"""
        n = self.n_phys_dim
        off = -self.coupling_constant * np.sqrt(np.arange(1, n))
        h = np.diag(np.arange(n, dtype=float)) + np.diag(off, -1) + np.diag(off, 1)
        return np.linalg.eigh(h)[1]

    @property
    def is_simple(self):
        return self.omega[0] == self.omega[1]

    @property
    def reorganization_energy(self):
        return Quantity(0.5 * (self.dis[1] - self.dis[0]) ** 2 * self.omega[1] ** 2)

    @property
    def e0(self):
        return self.reorganization_energy

    @property
    def coupling_constant(self):
        return float(np.sqrt(self.reorganization_energy.as_au() / self.omega[0]))

    @property
    def term10(self):
        """coefficient of (b^+ + b) in the excited-state potential (model/phonon.py:144-146)"""
        return self.omega[1] ** 2 / np.sqrt(2.0 * self.omega[0]) * (-self.dis[1])

    @property
    def pbond(self):
        return self.n_phys_dim

    def to_dict(self):
        return {"omega": self.omega, "displacement": self.dis, "num physical dimension": self.n_phys_dim}


class Mol:
    def __init__(self, elocalex, ph_list, dipole=None):
        self.elocalex = elocalex.as_au()
        self.dipole = dipole
        if len(ph_list) == 0:
            raise ValueError("No phonon mode in phonon list")
        self.ph_list = list(ph_list)
        self.e0 = sum(ph.reorganization_energy.as_au() for ph in ph_list)

    @property
    def gs_zpe(self):
        return sum(ph.omega[0] for ph in self.ph_list) / 2

    @property
    def reorganization_energy(self):
        return self.e0

    def to_dict(self):
        return {"elocalex": self.elocalex, "dipole": self.dipole, "reorganization energy in a.u.": self.e0,
                "phonon list": [ph.to_dict() for ph in self.ph_list]}
