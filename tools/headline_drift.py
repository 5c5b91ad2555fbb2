#!/usr/bin/env python3
"""Physics sanity of the headline run: energy / norm drift, total population and mean square displacement over a few
TDVP-PS steps of the bench workload (GPU box only).  Usage: tools/headline_drift.py [nsteps=10]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from renormalizer_amd.engine import get_engine  # noqa: E402

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
eng = get_engine()
model, mpo, mps = bench.build_workload(25, 16, 256, 1234, "physical")
sites = np.arange(25) - 12
print("| step | t (a.u.) | <H - E0> | norm - 1 | sum n_e - 1 | <r^2> | s per evolve |")
print("|---|---|---|---|---|---|---|")
for k in range(nsteps + 1):
    occ = np.asarray(mps.e_occupations)
    e = mps.expectation(mpo)
    print(f"| {k} | {10.0 * k:.0f} | {e:+.2e} | {mps.mp_norm - 1:+.1e} | {occ.sum() - 1:+.1e} | {float((occ * sites ** 2).sum()):.6f} | "
          f"{'' if k == 0 else f'{dt:.3f}'} |")
    if k < nsteps:
        eng.sync()
        t0 = time.perf_counter()
        mps = mps.evolve(mpo, 10.0)
        eng.sync()
        dt = time.perf_counter() - t0
