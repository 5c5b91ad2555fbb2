#!/usr/bin/env python3
"""Writes tests/golden/headline_state_pin.npz FROM THIS ENGINE (GPU box): what the prepared benchmark state does in its
first evolves - the mean Krylov dimension of the first evolve and the electronic populations after evolves 1 .. 6.
It pins DRIFT of the prepared state (a change in expand_bond_dimension or in the block SVD that moves the Krylov
dimension moves the headline figure: VERDICT round 4, item 6), not parity - parity of the same evolve is
tests/test_headline_gpu.py::test_headline_one_evolve_vs_oracle.
    python tools/make_headline_state_pin.py gpurun_out/headline_state_pin.npz"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/headline_state_pin.npz"
    import bench
    model, mpo, mps = bench.build_workload(25, 16, 256, 0, "physical")
    occ0 = np.asarray(mps.e_occupations, dtype=np.float64)
    occ, kry = [], []
    for _ in range(6):
        mps = mps.evolve(mpo, 10.0)
        occ.append(np.asarray(mps.e_occupations, dtype=np.float64))
        kry.append(float(mps.evolve_config.stat["mean"]))
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    np.savez(out, mean_krylov=np.array(kry), occ=np.array(occ), occ_start=occ0, bond_dims=np.array(mps.bond_dims))
    print("mean Krylov dimensions", kry)
    print("populations after evolve 1", occ[0][10:15])


if __name__ == "__main__":
    main()
