#!/usr/bin/env python3
"""cProfile of the 2-site DMRG of water / STO-3G (BASELINE config 5) - the run is host bound (41 ms of kernels in 0.3 - 0.5 s)."""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from renormalizer_amd import Model, Mpo, Mps, optimize_mps  # noqa: E402
from renormalizer_amd.engine import get_engine  # noqa: E402
from renormalizer_amd.model import h_qc  # noqa: E402

sh, aseri, nuc = h_qc.read_fcidump(os.path.join(REPO, "tests", "golden", "h2o_fcidump.txt"), 7)
basis, terms = h_qc.qc_model(sh, aseri)
model = Model(basis, terms)
mpo = Mpo(model)
eng = get_engine()


def run(M=50):
    mps = Mps.random(model, [5, 5], M, percent=1.0, rng=np.random.default_rng(1))
    mps.optimize_config.procedure = [[M, 0.4], [M, 0.2], [M, 0.1], [M, 0]]
    mps.optimize_config.method = "2site"
    eng.sync()
    t0 = time.perf_counter()
    energies, gs = optimize_mps(mps, mpo)
    eng.sync()
    return time.perf_counter() - t0, min(energies) + nuc


print("cold", run())
print("warm", run())
pr = cProfile.Profile()
pr.enable()
t = run()
pr.disable()
print("profiled", t)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30)
print(s.getvalue()[:7000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(30)
print(s.getvalue()[:6000])
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
st.print_callers("nonzero")
st.print_callers("eigh")
st.print_callers("asdevice")
print(s.getvalue()[:6000])
# what Hop.__init__ waits for: the device (the environments are still being computed when it is called)?
import renormalizer_amd.mps.hop_expr as HE
orig = HE.Hop.__init__
acc = [0.0, 0.0]
def timed(self, *a, **k):
    t0 = time.perf_counter(); eng.sync(); acc[0] += time.perf_counter() - t0
    t0 = time.perf_counter(); orig(self, *a, **k); acc[1] += time.perf_counter() - t0
HE.Hop.__init__ = timed
t = run()
print("run with a sync before every Hop.__init__:", t, "sync wait", acc[0], "init itself", acc[1])
