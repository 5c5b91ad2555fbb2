import cProfile, pstats, sys, os, io
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tools"))
import traj_scaling as ts
from renormalizer_amd.engine import get_engine
mpo, psi = ts.prepare(0)
psi = psi.evolve(mpo, 160.0); get_engine().sync()
pr = cProfile.Profile(); pr.enable()
for _ in range(2): psi = psi.evolve(mpo, 160.0)
get_engine().sync(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
