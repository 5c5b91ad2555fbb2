#!/usr/bin/env python3
"""Exciton dynamics in the FMO complex (example/fmo.py of the reference): 7 sites, Holstein modes sampled from
the tabulated spectral density, T = 0, TDVP-PS at fixed bond dimension.

    python examples/fmo.py [modes per site = 35] [D = 32] [steps = 5] [trajectories = 1] [temperature / K = 0]

A temperature > 0 switches to the thermofield form (BASELINE config 4): every mode is doubled, 7 + 2 x 7 x modes sites.

With more than one trajectory, static disorder (Gaussian, 50 cm^-1) is added to the site energies with a seed per
trajectory; with one process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT in the environment, e.g. from
``python -m torch.distributed.run --nproc-per-node N examples/fmo.py ...``) the trajectories are dealt to the ranks and
only the population tables are gathered at the end - one RCCL all-gather bound through ctypes, no PyTorch."""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from renormalizer_amd import (CompressConfig, CompressCriteria, EvolveConfig, EvolveMethod, HolsteinModel, Mol, Mpo, Mps,  # noqa: E402
                              Phonon, Quantity)
from renormalizer_amd.parallel import gather_observables, make_collective, trajectory_seed, units_of_rank  # noqa: E402
from renormalizer_amd.utils.constant import cm2au  # noqa: E402

J_CM = np.array([[310, -98, 6, -6, 7, -12, -10, 38], [-98, 230, 30, 7, 2, 12, 5, 8], [6, 30, 0, -59, -2, -10, 5, 2],
                 [-6, 7, -59, 180, -65, -17, -65, -2], [7, 2, -2, -65, 405, 89, -6, 5], [-12, 11, -10, -17, 89, 320, 32, -10],
                 [-10, 5, 5, -64, -6, 32, 270, -11], [38, 8, 2, -2, 5, -10, -11, 505]], dtype=float)


def fmo_model(n_phonons=35, total_hr=0.42, disorder_cm=0.0, rng=None, temperature_k=0.0):
    sdf = np.array(json.load(open(os.path.join(REPO, "tests", "golden", "fmo_sdf.json"))))
    om_cm = np.linspace(2, 300, n_phonons)
    om = om_cm * cm2au
    hr = np.interp(om_cm, sdf[:, 0], sdf[:, 1])
    hr *= total_hr / hr.sum()
    phonons = [Phonon.simplest_phonon(Quantity(o), Quantity(l), lam=True) for o, l in zip(om, hr * om)]
    j = J_CM * cm2au
    eps = np.diag(j).copy()
    if disorder_cm:
        eps = eps + rng.normal(0.0, disorder_cm * cm2au, size=len(eps))
    mols = [Mol(Quantity(e), phonons) for e in eps]
    arr = np.array([7, 5, 3, 1, 2, 4, 6]) - 1
    if temperature_k > 0:
        from renormalizer_amd.model import thermofield_holstein
        return thermofield_holstein([mols[i] for i in arr], j[arr][:, arr], Quantity(temperature_k, "K"))
    return HolsteinModel([mols[i] for i in arr], j[arr][:, arr])


def run(model, D, nsteps, dt=160.0):
    psi = Mpo.onsite(model, r"a^\dagger", dof_set={model.mol_num // 2}).apply(Mps.ground_state(model, False))
    mpo = Mpo(model, offset=Quantity(psi.expectation(Mpo(model))))
    psi.compress_config = CompressConfig(CompressCriteria.fixed, max_bonddim=D)
    psi.evolve_config = EvolveConfig(EvolveMethod.tdvp_ps)
    psi = psi.expand_bond_dimension(mpo).canonicalise()
    occ = [np.asarray(psi.e_occupations)]
    for _ in range(nsteps):
        psi = psi.evolve(mpo, dt)
        occ.append(np.asarray(psi.e_occupations))
    return np.array(occ)


if __name__ == "__main__":
    nph, D, nsteps, ntraj = [int(a) for a in sys.argv[1:5]] + [35, 32, 5, 1][len(sys.argv[1:5]):]
    temperature_k = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    os.environ.setdefault("RENO_GPU", os.environ.get("LOCAL_RANK", "0"))
    coll = make_collective()
    mine = units_of_rank(ntraj, rank, world)
    rows = []
    for u in mine:
        rng = np.random.default_rng(trajectory_seed(2024, u))
        model = fmo_model(nph, disorder_cm=50.0 if ntraj > 1 else 0.0, rng=rng, temperature_k=temperature_k)
        rows.append(run(model, D, nsteps).ravel())
    table = gather_observables(coll, np.array(rows), mine, ntraj)
    coll.close()
    if rank == 0:
        pops = table.reshape(ntraj, nsteps + 1, -1).mean(axis=0)
        for i, p in enumerate(pops):
            print(f"t = {160.0 * i:7.1f} a.u.  populations " + " ".join(f"{x:.4f}" for x in p))
