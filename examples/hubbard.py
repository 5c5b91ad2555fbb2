#!/usr/bin/env python3
"""Ground state of the one-dimensional Hubbard model (open chain) two ways: two-site DMRG and imaginary-time
TDVP-PS with adaptive steps - the workflow of the reference's example/hubbard.py on the MI355X engine.

    python examples/hubbard.py [nsites=10] [M=100]

H = t sum_{i,s} (c+_{i,s} c_{i+1,s} + h.c.) + U sum_i n_{i,up} n_{i,down}, Jordan-Wigner mapped onto 2 nsites spins
ordered (0 up, 0 down, 1 up, ...); every operator carries the change of (N_up, N_down) so the sweeps stay in the
requested particle-number sector."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from renormalizer_amd import BasisHalfSpin, EvolveConfig, EvolveMethod, Model, Mpo, Mps, Op, optimize_mps  # noqa: E402


def hubbard_model(nsites, t=-1.0, u=4.0):
    up = {"+": [-1, 0], "-": [1, 0], "Z": [0, 0]}
    down = {"+": [0, -1], "-": [0, 1], "Z": [0, 0]}
    terms = []
    for i in range(2 * (nsites - 1)):
        # hop between spin orbital i and i + 2 (same spin, next site); the string crosses orbital i + 1
        same, other = (up, down) if i % 2 == 0 else (down, up)
        terms.append(Op("Z + Z -", [i, i, i + 1, i + 2], factor=t, qn=[same["Z"], same["+"], other["Z"], same["-"]]))
        terms.append(Op("Z - Z +", [i, i, i + 1, i + 2], factor=-t, qn=[same["Z"], same["-"], other["Z"], same["+"]]))
    for i in range(0, 2 * nsites, 2):
        terms.append(Op("- + - +", [i, i, i + 1, i + 1], factor=u, qn=[up["-"], up["+"], down["-"], down["+"]]))
    basis = [BasisHalfSpin(i, sigmaqn=[[0, 0], [1, 0]] if i % 2 == 0 else [[0, 0], [0, 1]]) for i in range(2 * nsites)]
    return Model(basis, terms)


def dmrg(model, mpo, nelec, m, seed=0):
    mps = Mps.random(model, nelec, m, percent=1.0, rng=np.random.default_rng(seed))
    mps.optimize_config.procedure = [[m, 0.4], [m, 0.2], [m, 0.1], [m, 0], [m, 0], [m, 0], [m, 0]]
    mps.optimize_config.method = "2site"
    energies, mps = optimize_mps(mps, mpo)
    return min(energies), mps


def imaginary_time(mps, mpo, tol=1e-5, max_steps=200):
    mps.evolve_config = EvolveConfig(EvolveMethod.tdvp_ps, adaptive=True, guess_dt=1e-3 / 1j, adaptive_rtol=5e-4)
    e_old, trace = 0.0, []
    for step in range(max_steps):
        mps = mps.evolve(mpo, 0.5 / 1j)
        e = mps.expectation(mpo)
        trace.append(e)
        if abs(e - e_old) < tol:
            break
        e_old = e
    return trace, mps


if __name__ == "__main__":
    nsites = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    model = hubbard_model(nsites)
    mpo = Mpo(model)
    print("MPO bond dimensions:", mpo.bond_dims)
    nelec = [nsites // 2, nsites // 2]
    e_dmrg, _ = dmrg(model, mpo, nelec, m)
    print(f"two-site DMRG           E = {e_dmrg:.10f}")
    start = Mps.random(model, nelec, m, percent=1.0, rng=np.random.default_rng(1))
    trace, _ = imaginary_time(start, mpo)
    print(f"imaginary-time TDVP-PS  E = {trace[-1]:.10f}  ({len(trace)} steps of 0.5)")
