import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renormalizer_amd.engine import get_engine
from renormalizer_amd.mps import svd_qn
eng = get_engine(); rng = np.random.default_rng(0)
Dl, d, Dr = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
nq = int(sys.argv[4]) if len(sys.argv) > 4 else 1
qnl = np.sort(rng.integers(0, nq, size=(Dl, 1)), axis=0)
qnr = (nq - 1) - np.sort(rng.integers(0, nq, size=(Dr, 1)), axis=0)[::-1] if nq > 1 else np.zeros((Dr, 1), int)
sig = np.zeros((d, 1), int)
qbl, qbr = svd_qn.add_outer(qnl, sig), qnr
mask = svd_qn.get_qn_mask(svd_qn.add_outer(qbl, qbr), np.array([nq - 1]))
c = (rng.standard_normal((Dl,d,Dr)) + 1j*rng.standard_normal((Dl,d,Dr))) * mask
C = eng.asdevice(c)
for _ in range(5):
    svd_qn.svd_qn(C, qbl, qbr, np.array([nq - 1]), QR=True, system="L", full_matrices=False)
eng.sync()
