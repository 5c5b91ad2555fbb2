#!/usr/bin/env python3
"""Ground state of the optical Su-Schrieffer-Heeger model (one electron, hopping modulated by the difference of the
neighbouring oscillators' coordinates) by two-site DMRG, with the observables of the reference's example/ssh.py:
electronic reduced density matrix, phonon numbers and displacements, density-density correlations.

    python examples/ssh.py [nsites=2] [bond_dim=16] [nboson_max=4]

H = t sum_i (a+_i a_{i+1} + h.c.) + w0 sum_i b+_i b_i + g sum_i (a+_{i+1} a_i + h.c.) (X_{i+1} - X_i),  X = b+ + b"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from renormalizer_amd import BasisSHO, BasisSimpleElectron, Model, Mpo, Mps, Op, Quantity, optimize_mps  # noqa: E402
from renormalizer_amd.model import construct_j_matrix  # noqa: E402


def ssh_model(nsites, t=-1.0, g=0.7, w0=0.5, nboson_max=4, periodic=True):
    j = construct_j_matrix(nsites, Quantity(t), periodic)
    basis, ham = [], []
    for i in range(nsites):
        basis += [BasisSimpleElectron(i), BasisSHO((i, 0), w0, nboson_max)]
        ham.append(Op(r"b^\dagger b", (i, 0), w0))
        for k in range(nsites):
            if j[i, k] != 0:
                ham.append(Op(r"a^\dagger a", [i, k], j[i, k]))
    bonds = [(i, i + 1) for i in range(nsites - 1)] + ([(nsites - 1, 0)] if periodic else [])
    for lo, hi in bonds:
        for a, b in ((lo, hi), (hi, lo)):
            ham.append(Op(r"a^\dagger a", [a, b], g) * Op(r"b^\dagger+b", (hi, 0)))
            ham.append(Op(r"a^\dagger a", [a, b], -g) * Op(r"b^\dagger+b", (lo, 0)))
    return Model(basis, ham)


def ground_state(model, bond_dim, nsweeps=10, seed=0):
    mps = Mps.random(model, 1, bond_dim, percent=1.0, rng=np.random.default_rng(seed))
    mps.optimize_config.procedure = [[max(bond_dim // 4, 2), 0.4], [max(bond_dim // 2, 2), 0.2],
                                     [max(3 * bond_dim // 4, 2), 0.1]] + [[bond_dim, 0]] * (nsweeps - 3)
    mps.optimize_config.method = "2site"
    energies, mps = optimize_mps(mps, Mpo(model))
    return min(energies), mps


def observables(model, mps):
    n = model.n_edofs
    number = [Mpo(model, Op(r"a^\dagger a", [i, i])) for i in range(n)]
    return {"edof_rdm": mps.calc_edof_rdm(), "phonon_occupations": np.asarray(mps.ph_occupations),
            "phonon_displacement": np.array([mps.expectation(Mpo(model, Op(r"b^\dagger+b", (i, 0)))) for i in range(n)]),
            "ni_nj": np.array([[mps.expectation(number[i] @ number[k]) for k in range(n)] for i in range(n)])}


if __name__ == "__main__":
    nsites, bond_dim, nboson = [int(a) for a in sys.argv[1:4]] + [2, 16, 4][len(sys.argv[1:4]):]
    model = ssh_model(nsites, nboson_max=nboson)
    energy, mps = ground_state(model, bond_dim)
    print(f"ground-state energy {energy:.10f}")
    for key, val in observables(model, mps).items():
        print(key, np.round(np.real(val), 8).tolist())
