#!/usr/bin/env python3
"""Wall-time breakdown of one TDVP-PS evolve on the headline config by phase (synchronising around each
phase, so the sum slightly exceeds the un-instrumented step time).  GPU box only."""
import os
import sys
import time
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import numpy as np  # noqa: E402

from renormalizer_amd.engine import get_engine  # noqa: E402
import renormalizer_amd.mps.mps as M  # noqa: E402

eng = get_engine()
acc = defaultdict(float)
cnt = defaultdict(int)


def wrap(mod, name, label=None):
    f = getattr(mod, name)

    def g(*a, **k):
        eng.sync()
        t0 = time.perf_counter()
        r = f(*a, **k)
        eng.sync()
        key = label(a, k) if callable(label) else (label or name)
        acc[key] += time.perf_counter() - t0
        cnt[key] += 1
        return r
    setattr(mod, name, g)


wrap(M, "expm_krylov", lambda a, k: "expm 1-site" if a[0].nsite == 1 else "expm 0-site")
wrap(M.svd_qn, "svd_qn", "block QR")
wrap(M, "contract_one_site", "env update (direct)")
import renormalizer_amd.mps.lib as L  # noqa: E402
wrap(L, "contract_one_site", "env update")
wrap(M, "hop_expr", "hop_expr setup")

D = int(sys.argv[1]) if len(sys.argv) > 1 else 256
model, mpo, mps = bench.build_workload(25, 16, D, 1234, sys.argv[2] if len(sys.argv) > 2 else "random")
mps = mps.evolve(mpo, 10.0)
acc.clear(); cnt.clear()
eng.sync()
t0 = time.perf_counter()
mps = mps.evolve(mpo, 10.0)
eng.sync()
tot = time.perf_counter() - t0
print(f"instrumented evolve: {tot*1e3:.1f} ms")
s = 0
for k, v in sorted(acc.items(), key=lambda x: -x[1]):
    print(f"  {k:24s} {v*1e3:9.1f} ms  {cnt[k]:5d} calls  {v/cnt[k]*1e3:8.3f} ms/call")
    s += v
print(f"  {'other (python, absorb, norm)':24s} {(tot-s)*1e3:9.1f} ms")
