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
