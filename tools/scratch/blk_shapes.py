import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from renormalizer_amd.mps import svd_qn
from collections import Counter
for init in ("physical", "random"):
    model, mpo, mps = bench.build_workload(25, 16, 256, 1, init)
    mps = mps.to_complex().evolve(mpo, 10.0)
    c = Counter()
    for key, plan in svd_qn._PLAN_CACHE.items():
        c[tuple((len(b[2]), len(b[3])) for b in plan["blocks"])] += 1
    print(init)
    for k, v in sorted(c.items(), key=lambda kv: -kv[1])[:14]:
        print("  ", v, k)
    svd_qn._PLAN_CACHE.clear()
