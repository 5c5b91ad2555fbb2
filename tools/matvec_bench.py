#!/usr/bin/env python3
"""One-site effective-Hamiltonian matvec on dense random operands of the headline shapes, repeated: for kernel traces
of its three launches without the rest of a sweep (GPU box).
Usage: tools/matvec_bench.py [d = 16] [wl = 5] [wr = 4] [reps = 50]
MPSE_BENCH_W=holstein: the MPO site is taken from the headline Hamiltonian (first site with that d, wl, wr) instead of a
dense random one."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renormalizer_amd.engine import get_engine  # noqa: E402
from renormalizer_amd.mps.hop_expr import hop_expr  # noqa: E402

d, wl, wr, reps = [int(a) for a in sys.argv[1:5]] + [16, 5, 4, 50][len(sys.argv[1:5]):]
D = 256
eng = get_engine()
rng = np.random.default_rng(0)
c = lambda *s: rng.normal(size=s) + 1j * rng.normal(size=s)
w_host = rng.normal(size=(wl, d, d, wr))
if os.environ.get("MPSE_BENCH_W") == "holstein":
    import bench
    model, mpo, _ = bench.build_workload(25, 16, 4, 1, "random")
    w_host = next(np.asarray(mpo[i]) for i in range(5, len(mpo) - 5) if np.asarray(mpo[i]).shape[1] == d)
    wl, wr = w_host.shape[0], w_host.shape[3]
    print(f"MPO site of the headline chain: {np.count_nonzero(w_host)} of {w_host.size} entries non-zero")
W = eng.asdevice(np.ascontiguousarray(w_host.real if np.iscomplexobj(w_host) else w_host, dtype=np.float64))
L, R = eng.asdevice(c(D, wl, D)), eng.asdevice(c(D, wr, D))
C = eng.asdevice(c(D, d, D))
hop = hop_expr(L, R, [W], (D, d, D))
for _ in range(5):
    out = hop(C)
eng.sync()
t0 = time.perf_counter()
for _ in range(reps):
    out = hop(C)
eng.sync()
print(f"matvec (D={D}, d={d}, wl={wl}, wr={wr}), dense operands: {(time.perf_counter() - t0) / reps * 1e6:.1f} us")
