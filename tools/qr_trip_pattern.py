import sys, time
sys.path.insert(0, "/root/repo")
import bench
from renormalizer_amd.mps import mps as _m
from renormalizer_amd.engine import get_engine
eng = get_engine()
model, mpo, mps = bench.build_workload(25, 16, 256, 0, "physical")
for s in range(30):
    r0 = _m._OPTIMISTIC_REDONE[0]; q0 = eng.block_qr_stats()
    t0 = time.perf_counter()
    mps = mps.evolve(mpo, 10.0)
    eng.sync()
    q1 = eng.block_qr_stats()
    print(s, "redone", _m._OPTIMISTIC_REDONE[0] - r0, "qr", tuple(b - a for a, b in zip(q0, q1)), "ms %.1f" % ((time.perf_counter() - t0) * 1e3), flush=True)
