#!/usr/bin/env python3
"""Are two prepared start states of the headline benchmark the same physics?  Both are evolved for N steps with the
current engine; the electronic populations, <H> and the mean Krylov dimension are compared step by step.
Usage: tools/state_compare.py stateA.npz stateB.npz [steps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
runs = []
for path in sys.argv[1:3]:
    model, mpo, mps = bench.build_workload(25, 16, 256, 0, "physical", state_file=path)
    occ, en, kry = [np.asarray(mps.e_occupations)], [mps.expectation(mpo)], []
    for _ in range(steps):
        mps = mps.evolve(mpo, 10.0)
        occ.append(np.asarray(mps.e_occupations))
        en.append(mps.expectation(mpo))
        kry.append(mps.evolve_config.stat["mean"])
    runs.append((np.array(occ), np.array(en), np.array(kry)))
(oa, ea, ka), (ob, eb, kb) = runs
print("| step | max |occ_A - occ_B| | <H>_A - <H>_A(0) | <H>_B - <H>_B(0) | mean Krylov dim A | B |")
print("|---|---|---|---|---|---|")
for s in range(0, steps + 1, max(1, steps // 10)):
    print(f"| {s} | {np.abs(oa[s] - ob[s]).max():.3e} | {ea[s] - ea[0]:+.3e} | {eb[s] - eb[0]:+.3e} | "
          f"{ka[s - 1] if s else float('nan'):.3f} | {kb[s - 1] if s else float('nan'):.3f} |")
print(f"\ninitial <H>: A {ea[0]:.12f}  B {eb[0]:.12f}; largest population difference over the run {np.abs(oa - ob).max():.3e}")
