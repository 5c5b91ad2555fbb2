#!/usr/bin/env python3
"""Small drivers for rocprofv3 of the paths that are not the headline: `h2o` (config 5: two-site DMRG, M = 512),
`ps2` (two-site TDVP of a Holstein chain, D = 64), `expand` (bond expansion of the headline state, D = 256).
Usage: rocprofv3 --kernel-trace --stats -d out -- python tools/profile_cases.py <case>"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from renormalizer_amd import (CompressConfig, CompressCriteria, EvolveConfig, EvolveMethod, HolsteinModel, Model, Mol, Mpo,  # noqa: E402
                              Mps, Phonon, Quantity, optimize_mps)
from renormalizer_amd.engine import get_engine  # noqa: E402
from renormalizer_amd.model import h_qc  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "h2o"
eng = get_engine()


def holstein(nmol, D, method):
    ph = Phonon.simple_phonon(Quantity(6.128e-3), Quantity(16.274571056529368), 16)
    model = HolsteinModel([Mol(Quantity(0), [ph])] * nmol, Quantity(3.0e-2), 3)
    psi = Mpo.onsite(model, r"a^\dagger", dof_set={nmol // 2}).apply(Mps.ground_state(model, False))
    mpo = Mpo(model, offset=Quantity(psi.expectation(Mpo(model))))
    psi.compress_config = CompressConfig(CompressCriteria.fixed, max_bonddim=D)
    psi.evolve_config = EvolveConfig(method)
    return model, mpo, psi


t0 = time.perf_counter()
if case == "h2o":
    sh, aseri, nuc = h_qc.read_fcidump(os.path.join(REPO, "tests", "golden", "h2o_fcidump.txt"), 7)
    model = Model(*h_qc.qc_model(sh, aseri))
    mpo = Mpo(model)
    M = 512
    mps = Mps.random(model, [5, 5], M, percent=1.0, rng=np.random.default_rng(1))
    mps.optimize_config.procedure = [[M, 0.4], [M, 0.2], [M, 0.1], [M, 0]]
    mps.optimize_config.method = "2site"
    eng.sync()
    t0 = time.perf_counter()
    energies, gs = optimize_mps(mps, mpo)
    eng.sync()
    print("h2o", time.perf_counter() - t0, "s", [e + nuc for e in energies], gs.bond_dims)
elif case == "ps2":
    model, mpo, psi = holstein(10, 64, EvolveMethod.tdvp_ps2)
    psi = psi.evolve(mpo, 10.0)
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        psi = psi.evolve(mpo, 10.0)
    eng.sync()
    print("ps2", (time.perf_counter() - t0) / 3, "s per evolve", psi.bond_dims)
elif case == "expand":
    model, mpo, psi = holstein(25, 256, EvolveMethod.tdvp_ps)
    eng.sync()
    t0 = time.perf_counter()
    psi = psi.expand_bond_dimension(mpo).canonicalise()
    eng.sync()
    print("expand", time.perf_counter() - t0, "s", psi.bond_dims)
