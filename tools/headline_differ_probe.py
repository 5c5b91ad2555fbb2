#!/usr/bin/env python3
"""Which solves of a headline evolve differ in Krylov dimension between the device and the oracle, and what do the bonds
next to them look like?  (Diagnostic behind tests/test_headline_gpu.py::_compare_evolve.)  GPU box:
python tools/headline_differ_probe.py [evolves]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import bench  # noqa: E402
from oracle import mps_oracle as orc  # noqa: E402
from test_headline_gpu import _solve_sites, _oracle_state  # noqa: E402

model, mpo, mps = bench.build_workload(25, 16, 256, 0, "physical")
mps = mps.evolve(mpo, 10.0)
w_host = [mpo[i] for i in range(len(mpo))]
dev, ost = mps, _oracle_state(model, mps)
for ev in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
    sing = dev.calc_bond_singular_values()
    ratios = []
    for b in range(1, len(dev)):
        sv = np.asarray(sing[b - 1], dtype=float)[: dev.bond_dims[b]]
        ratios.append(sv.min() / sv.max())
    print("evolve", ev, "bond sigma_min/sigma_max:", " ".join("%d:%.0e" % (b + 1, r) for b, r in enumerate(ratios)), flush=True)
    where = _solve_sites(len(dev), bool(ost.to_right))
    dev2 = dev.evolve(mpo, 10.0)
    ost2 = orc.tdvp_ps_step(ost, w_host, 10.0)
    dd, od = list(dev2.evolve_config.stat["steps"]), list(ost2.krylov_dims)
    for i, (a, b) in enumerate(zip(dd, od)):
        if a != b:
            site, nbr = where[i]
            bonds = sorted({site, site + 1} if nbr is None else {site, site + 1, nbr, nbr + 1})
            print("  solve", i, where[i], "dev", a, "oracle", b, "margins", ["%.3g" % m for m in ost2.krylov_margins[i]],
                  "bonds", [(bb, "%.1e" % ratios[bb - 1]) for bb in bonds if 0 < bb < len(dev)], flush=True)
    dev, ost = dev2, ost2
