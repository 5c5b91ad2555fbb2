#!/usr/bin/env python3
"""Charge transport in a Holstein chain (the headline workload at a size of your choice):
electron created on the centre molecule of the phonon vacuum, bonds expanded, TDVP-PS.

    python examples/holstein_dynamics.py [nmol=9] [pdim=8] [D=32] [nsteps=10]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from renormalizer_amd import (CompressConfig, CompressCriteria, EvolveConfig, EvolveMethod, HolsteinModel, Mol, Mpo,  # noqa: E402
                              Mps, Phonon, Quantity)

nmol, pdim, D, nsteps = [int(a) for a in sys.argv[1:5]] + [9, 8, 32, 10][len(sys.argv[1:5]):]
ph = Phonon.simple_phonon(Quantity(6.128e-3), Quantity(16.274571056529368), pdim)      # example/std.yaml
model = HolsteinModel([Mol(Quantity(0), [ph])] * nmol, Quantity(3.0e-2), 3)
psi = Mpo.onsite(model, r"a^\dagger", dof_set={nmol // 2}).apply(Mps.ground_state(model, False))
mpo = Mpo(model, offset=Quantity(psi.expectation(Mpo(model))))
psi.compress_config = CompressConfig(CompressCriteria.fixed, max_bonddim=D)
psi.evolve_config = EvolveConfig(EvolveMethod.tdvp_ps)
psi = psi.expand_bond_dimension(mpo).canonicalise()
sites = np.arange(nmol) - nmol // 2
for step in range(nsteps + 1):
    occ = np.asarray(psi.e_occupations)
    print(f"t = {10.0 * step:6.1f} a.u.  <r^2> = {float(np.sum(occ * sites ** 2)):9.5f}  norm = {psi.mp_norm:.12f}  "
          f"E = {psi.expectation(mpo):+.3e}")
    if step < nsteps:
        psi = psi.evolve(mpo, 10.0)
