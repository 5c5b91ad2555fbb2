import os, sys, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
model, mpo, mps = bench.build_workload(25, 16, 256, 1, "random")
mps = mps.to_complex()
mps = mps.evolve(mpo, 10.0)
from renormalizer_amd.engine import get_engine
eng = get_engine()
eng.sync()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(2):
    mps = mps.evolve(mpo, 10.0)
eng.sync()
pr.disable()
print("wall per evolve", (time.perf_counter() - t0) / 2)
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
