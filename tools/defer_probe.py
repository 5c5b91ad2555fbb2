#!/usr/bin/env python3
"""Step times and pool statistics of the headline evolve with / without recorded post-solve calls (GPU box)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MPSE_POOL_STATS"] = "1"
import bench  # noqa: E402

from renormalizer_amd.engine import get_engine  # noqa: E402

eng = get_engine()
model, mpo, mps = bench.build_workload(25, 16, 256, 1234, "physical")
for i in range(7):
    eng.sync()
    t0 = time.perf_counter()
    mps = mps.evolve(mpo, 10.0)
    eng.sync()
    dt = time.perf_counter() - t0
    print(f"evolve {i}: {dt*1e3:.1f} ms", eng.mem_info(), flush=True)
