#!/usr/bin/env python3
"""Spin-boson dynamics at zero temperature - the model of the reference's example/sbm.py on the MI355X engine
(Ohmic bath, adiabatically renormalised tunnelling): spin up, bath in its vacuum, <sigma_z(t)> per step.

    python examples/sbm.py [n_phonons=300] [evolve_time=20] [tdvp]

With the third argument the run uses TDVP-PS at a fixed bond dimension of 64 instead of adaptive P&C."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from renormalizer_amd import CompressConfig, CompressCriteria, EvolveConfig, EvolveMethod, Mpo, Quantity  # noqa: E402
from renormalizer_amd.mps import Mps  # noqa: E402
from renormalizer_amd.sbm import param2mollist  # noqa: E402


def sigma_z(mps, idx):
    rho = mps.calc_1site_rdm(idx=idx)[idx]
    return float((rho[0, 0] - rho[1, 1]).real)


if __name__ == "__main__":
    n_phonons = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    evolve_time = float(sys.argv[2]) if len(sys.argv) > 2 else 20
    model = param2mollist(alpha=0.05, raw_delta=Quantity(1), omega_c=Quantity(20), renormalization_p=1, n_phonons=n_phonons)
    mpo = Mpo(model)
    mps = Mps.ground_state(model, False)
    if len(sys.argv) > 3:
        mps.compress_config = CompressConfig(CompressCriteria.fixed, max_bonddim=64)
        mps.evolve_config = EvolveConfig(EvolveMethod.tdvp_ps)
        mps = mps.expand_bond_dimension(mpo, coef=1e-16, include_ex=False)
    else:
        mps.compress_config = CompressConfig(threshold=1e-4)
        mps.evolve_config = EvolveConfig(adaptive=True, guess_dt=0.1)
    spin = next(i for i, b in enumerate(model.basis) if b.is_spin)
    t = 0.0
    print(f"t = {t:6.2f}  <sigma_z> = {sigma_z(mps, spin):+.8f}")
    while t < evolve_time - 1e-12:
        mps = mps.evolve(mpo, 0.1)
        t += 0.1
        print(f"t = {t:6.2f}  <sigma_z> = {sigma_z(mps, spin):+.8f}")
