#!/usr/bin/env python3
"""The reference's README quick start on the MI355X engine: two spins, H = s+_0 s-_1 + s+_1 s-_0, <Z_0>(t)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from renormalizer_amd import BasisHalfSpin, Model, Mpo, Mps, Op  # noqa: E402

ham_terms = Op("sigma_+ sigma_-", [0, 1]) + Op("sigma_+ sigma_-", [1, 0])
model = Model([BasisHalfSpin(0), BasisHalfSpin(1)], ham_terms)
mpo = Mpo(model)
mps = Mps.hartree_product_state(model, condition={0: [0, 1]})
z_op = Mpo(model, Op("Z", 0))
for _ in range(10):
    mps = mps.evolve(mpo, evolve_dt=0.05)          # default method: propagate & compress
    print(mps.expectation(z_op))                    # -0.9950041657975273 ... -0.5403023496556285
