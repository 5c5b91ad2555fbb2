#!/usr/bin/env python3
"""Timing of the blocked QR / SVD entry points in isolation (GPU box)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renormalizer_amd.engine import get_engine  # noqa: E402
from renormalizer_amd.mps import svd_qn  # noqa: E402

eng = get_engine()
rng = np.random.default_rng(0)
for (Dl, d, Dr, nq) in ((256, 16, 256, 2), (256, 2, 256, 2), (256, 16, 256, 1), (64, 8, 64, 1)):
    qnl = np.sort(rng.integers(0, nq, size=(Dl, 1)), axis=0)
    qnr = (nq - 1) - np.sort(rng.integers(0, nq, size=(Dr, 1)), axis=0)[::-1] if nq > 1 else np.zeros((Dr, 1), int)
    sig = np.zeros((d, 1), dtype=int)
    for system in ("L", "R"):
        if system == "L":
            qbl, qbr = svd_qn.add_outer(qnl, sig), qnr
        else:
            qbl, qbr = qnl, svd_qn.add_outer(sig, qnr)
        mask = svd_qn.get_qn_mask(svd_qn.add_outer(qbl, qbr), np.array([nq - 1]))
        c = (rng.standard_normal((Dl, d, Dr)) + 1j * rng.standard_normal((Dl, d, Dr))) * mask
        C = eng.asdevice(c)
        for _ in range(3):
            svd_qn.svd_qn(C, qbl, qbr, np.array([nq - 1]), QR=True, system=system, full_matrices=False)
        eng.sync()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            svd_qn.svd_qn(C, qbl, qbr, np.array([nq - 1]), QR=True, system=system, full_matrices=False)
        eng.sync()
        dt = (time.perf_counter() - t0) / n
        nb = len(svd_qn.qn_blocks(qbl, qbr, np.array([nq - 1])))
        print(f"QR {system} ({Dl},{d},{Dr}) blocks={nb}: {dt*1e3:.3f} ms")
