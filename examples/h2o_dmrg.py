#!/usr/bin/env python3
"""Water / STO-3G ground state by 2-site DMRG from an FCIDUMP file (example/h2o_qc.py of the reference)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from renormalizer_amd import Model, Mpo, Mps, optimize_mps  # noqa: E402
from renormalizer_amd.model import h_qc  # noqa: E402

sh, aseri, nuc = h_qc.read_fcidump(os.path.join(REPO, "tests", "golden", "h2o_fcidump.txt"), 7)
basis, terms = h_qc.qc_model(sh, aseri)
model = Model(basis, terms)
mpo = Mpo(model)
print("MPO bond dimensions:", mpo.bond_dims)
M = 50
mps = Mps.random(model, [5, 5], M, percent=1.0, rng=np.random.default_rng(1))
mps.optimize_config.procedure = [[M, 0.4], [M, 0.2], [M, 0.1], [M, 0], [M, 0], [M, 0], [M, 0]]
mps.optimize_config.method = "2site"
energies, gs = optimize_mps(mps, mpo)
print("sweep energies + E_nuc:", [e + nuc for e in energies])
print("FCI reference           -75.008697516450")
