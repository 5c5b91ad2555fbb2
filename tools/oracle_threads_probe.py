#!/usr/bin/env python3
"""How many BLAS threads make the oracle's headline evolve fastest on this host?  (The GPU parity suite spends most of
its time in three oracle evolves.)  Times the first 12 site updates of one oracle evolve per thread limit."""
import os
import sys
import time

import numpy as np
from threadpoolctl import threadpool_limits

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import bench  # noqa: E402
from oracle import mps_oracle as orc  # noqa: E402
from test_headline_gpu import _oracle_state  # noqa: E402

model, mpo, mps = bench.build_workload(25, 16, 256, 0, "physical")
mps = mps.evolve(mpo, 10.0)
w_host = [mpo[i] for i in range(len(mpo))]
ost = _oracle_state(model, mps)
for lim in (None, 64, 16, 8, 4):
    t0 = time.perf_counter()
    if lim is None:
        orc.tdvp_ps_step(ost, w_host, 10.0, max_updates=12, env_domain="R")
    else:
        with threadpool_limits(limits=lim):
            orc.tdvp_ps_step(ost, w_host, 10.0, max_updates=12, env_domain="R")
    print("threads", lim, "12 updates incl. environments: %.1f s" % (time.perf_counter() - t0), flush=True)
