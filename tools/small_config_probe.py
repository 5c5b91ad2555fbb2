#!/usr/bin/env python3
"""Config 2 (spin-boson, D = 64) evolve time and a host profile of it (GPU box): python tools/small_config_probe.py [prof]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renormalizer_amd import CompressConfig, CompressCriteria, EvolveConfig, EvolveMethod, Mpo, Mps, Quantity  # noqa: E402
from renormalizer_amd.engine import get_engine  # noqa: E402
from renormalizer_amd.mps import mps as _m  # noqa: E402
from renormalizer_amd.sbm import param2model  # noqa: E402

eng = get_engine()
model, _ = param2model(0.05, Quantity(1), Quantity(20), 1, 20, 8)
mpo = Mpo(model)
mps = Mps.ground_state(model, False)
mps.compress_config = CompressConfig(CompressCriteria.fixed, max_bonddim=64)
mps.evolve_config = EvolveConfig(EvolveMethod.tdvp_ps)
mps = mps.expand_bond_dimension(mpo, coef=1e-16, include_ex=False)
for s in range(8):
    r0, q0 = _m._OPTIMISTIC_REDONE[0], eng.block_qr_stats()
    eng.sync()
    t0 = time.perf_counter()
    mps = mps.evolve(mpo, 0.1)
    eng.sync()
    q1 = eng.block_qr_stats()
    print(s, "ms %.2f" % ((time.perf_counter() - t0) * 1e3), "redone", _m._OPTIMISTIC_REDONE[0] - r0,
          "qr", tuple(b - a for a, b in zip(q0, q1)), "fused", eng.heff_fused_stats(), flush=True)
if len(sys.argv) > 1:
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        mps = mps.evolve(mpo, 0.1)
    eng.sync()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
