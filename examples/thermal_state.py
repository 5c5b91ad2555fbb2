#!/usr/bin/env python3
"""Finite-temperature state of the one-exciton Holstein trimer by imaginary-time TDVP on a purified density
operator (mps/tests/test_mpdm.py of the reference): populations at 298 K."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from renormalizer_amd import HolsteinModel, Mol, Phonon, Quantity  # noqa: E402
from renormalizer_amd import Mpo  # noqa: E402
from renormalizer_amd.mps import MpDm, thermal_state  # noqa: E402
from renormalizer_amd.utils import CompressConfig, CompressCriteria, EvolveConfig, EvolveMethod, constant  # noqa: E402

ph_list = [Phonon.simple_phonon(Quantity(106.51, "cm^{-1}"), Quantity(30.1370), 4),
           Phonon.simple_phonon(Quantity(1555.55, "cm^{-1}"), Quantity(8.7729), 4)]
j = np.array([[0.0, -0.1, -0.2], [-0.1, 0.0, -0.3], [-0.2, -0.3, 0.0]]) / constant.au2ev
model = HolsteinModel([Mol(Quantity(2.67, "eV"), ph_list, 15.45)] * 3, j, 3)
rho = MpDm.max_entangled_ex(model)
rho.compress_config = CompressConfig(CompressCriteria.fixed, max_bonddim=16)
beta = Quantity(298, "K").to_beta()
rho.evolve_config = EvolveConfig(EvolveMethod.tdvp_ps, guess_dt=0.1 / 1j)
h = Mpo(model)
rho, energies = thermal_state(rho, h, beta / 2j / 20, 20)
print("populations :", rho.e_occupations, " exact 0.20896541 0.35240030 0.43863429")
print("energy      :", energies[-1], " exact", 0.0853388 + model.gs_zpe)
