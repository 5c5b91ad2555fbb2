"""Purified thermal state by imaginary-time propagation (the preparer behind the reference's finite-temperature
runs, mps/thermalprop.py:95-133): rho(beta/2) = exp(-beta H / 2) rho(0), ``nsteps`` calls of ``MpDm.evolve`` with an
imaginary step, the Hamiltonian re-referenced to the latest energy before every step so the norm stays O(1)."""
import numpy as np

from ..utils import Quantity
from .mpo import Mpo


def thermal_state(init_mpdm, h_mpo, evolve_dt, nsteps, auto_expand=True, on_step=None):
    """Returns (rho(beta / 2), energies).  ``evolve_dt`` = beta / 2j / nsteps (negative imaginary); ``on_step(rho)`` is
    called on the initial state and after every step."""
    assert np.iscomplex(evolve_dt) and np.imag(evolve_dt) < 0
    rho = init_mpdm.canonicalise()
    if rho.evolve_config.is_tdvp and auto_expand:
        rho = rho.expand_bond_dimension(h_mpo)
    energies = [rho.expectation(h_mpo)]
    if on_step is not None:
        on_step(rho)
    for _ in range(nsteps):
        rho = rho.evolve(Mpo(h_mpo.model, offset=Quantity(energies[-1])), evolve_dt)
        energies.append(rho.expectation(h_mpo))
        if on_step is not None:
            on_step(rho)
    return rho, energies
