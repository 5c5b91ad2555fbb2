#!/usr/bin/env python3
"""Does the cyclic garbage collector cost the host-bound configurations anything?  (The cProfile of the H2O sweeps shows
70 - 100 ms chunks that move from one allocation-heavy function to another between runs.)  Times config 5 (H2O DMRG) and
config 4 (FMO thermofield, 497 sites) with the collector as it is, after gc.freeze(), and disabled."""
import gc
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
from renormalizer_amd import Model, Mpo, Mps, optimize_mps  # noqa: E402
from renormalizer_amd.engine import get_engine  # noqa: E402
from renormalizer_amd.model import h_qc  # noqa: E402
import traj_scaling as ts  # noqa: E402

eng = get_engine()
sh, aseri, nuc = h_qc.read_fcidump(os.path.join(REPO, "tests", "golden", "h2o_fcidump.txt"), 7)
basis, terms = h_qc.qc_model(sh, aseri)
model = Model(basis, terms)
mpo = Mpo(model)


def h2o(M=50):
    mps = Mps.random(model, [5, 5], M, percent=1.0, rng=np.random.default_rng(1))
    mps.optimize_config.procedure = [[M, 0.4], [M, 0.2], [M, 0.1], [M, 0]]
    mps.optimize_config.method = "2site"
    eng.sync()
    t0 = time.perf_counter()
    optimize_mps(mps, mpo)
    eng.sync()
    return time.perf_counter() - t0


fmo_mpo, fmo_psi = ts.prepare(0)
fmo_psi = fmo_psi.evolve(fmo_mpo, 160.0)


def fmo():
    global fmo_psi
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(2):
        fmo_psi = fmo_psi.evolve(fmo_mpo, 160.0)
    eng.sync()
    return (time.perf_counter() - t0) / 2


def report(tag):
    c0 = [s["collections"] for s in gc.get_stats()]
    a = [h2o() for _ in range(3)]
    b = [fmo() for _ in range(3)]
    c1 = [s["collections"] for s in gc.get_stats()]
    print(f"{tag}: H2O sweeps {min(a):.3f} s (3 runs: {' '.join('%.3f' % x for x in a)});  FMO 497 sites {min(b):.3f} s per evolve "
          f"= {2 * 497 / min(b):.0f} site-updates/s;  gc collections gen0/1/2 in this block: {[y - x for x, y in zip(c0, c1)]}; "
          f"tracked objects {len(gc.get_objects())}", flush=True)


h2o()
report("gc as it is   ")
gc.collect()
gc.freeze()
report("gc.freeze()   ")
gc.unfreeze()
gc.disable()
report("gc.disable()  ")
gc.enable()
report("gc as it is   ")
