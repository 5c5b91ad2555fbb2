import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renormalizer_amd.engine import get_engine
from renormalizer_amd.mps import svd_qn
eng = get_engine(); rng = np.random.default_rng(0)
Dl, d, Dr = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
qnl = np.zeros((Dl,1),int); qnr=np.zeros((Dr,1),int); sig=np.zeros((d,1),int)
qbl, qbr = svd_qn.add_outer(qnl, sig), qnr
c = rng.standard_normal((Dl,d,Dr)) + 1j*rng.standard_normal((Dl,d,Dr))
C = eng.asdevice(c)
for _ in range(5):
    svd_qn.svd_qn(C, qbl, qbr, np.array([0]), QR=True, system="L", full_matrices=False)
eng.sync()
