#!/usr/bin/env python3
"""cProfile of one un-instrumented TDVP-PS evolve on the headline config (host side: where the Python thread
spends its time, including the waits on the device inside ctypes calls).  GPU box only."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

from renormalizer_amd.engine import get_engine  # noqa: E402

eng = get_engine()
D = int(sys.argv[1]) if len(sys.argv) > 1 else 256
init = sys.argv[2] if len(sys.argv) > 2 else "physical"
model, mpo, mps = bench.build_workload(25, 16, D, 1234, init)
mps = mps.evolve(mpo, 10.0)
eng.sync()
t0 = time.perf_counter()
mps = mps.evolve(mpo, 10.0)
eng.sync()
print(f"plain evolve: {(time.perf_counter()-t0)*1e3:.1f} ms")
pr = cProfile.Profile()
pr.enable()
mps = mps.evolve(mpo, 10.0)
eng.sync()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
