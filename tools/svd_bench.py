#!/usr/bin/env python3
"""Timing and accuracy of the blocked SVD entry point in isolation (GPU box): two-site centres of the headline chain
(256 x 2 x 16 x 256 -> 512 x 4096 in two quantum-number blocks) with a decaying spectrum, real and complex.

    python tools/svd_bench.py [out.md [case substring]]      MPSE_SVD_GRAM=0 selects the column kernel of rounds 4 - 5

Per case: ms per decomposition (20 calls), sweeps per call (the engine's profile counter), largest deviation of the
singular values from LAPACK's relative to the largest one, orthogonality of both factors, reconstruction."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renormalizer_amd.engine import get_engine  # noqa: E402
from renormalizer_amd.mps import svd_qn  # noqa: E402

eng = get_engine()
rng = np.random.default_rng(5)
lines = ["| case | blocks (rows x cols) | ms per decomposition | max dev of sigma / sigma_max | U^H U - 1 | V V^H - 1 | reconstruction |",
         "|---|---|---|---|---|---|---|"]


def spectrum_matrix(m, n, cplx, decay):
    k = min(m, n)
    a = rng.standard_normal((m, k)) + (1j * rng.standard_normal((m, k)) if cplx else 0)
    b = rng.standard_normal((k, n)) + (1j * rng.standard_normal((k, n)) if cplx else 0)
    u, _ = np.linalg.qr(a)
    v, _ = np.linalg.qr(b.conj().T)
    s = np.exp(-decay * np.arange(k) / k)
    return (u * s) @ v.conj().T


for name, (Dl, dl, dr, Dr, nq, cplx, decay) in {
        "two-site centre 512 x 4096, complex, 2 blocks": (256, 2, 16, 256, 2, True, 25.0),
        "two-site centre 512 x 4096, complex, 1 block": (256, 2, 16, 256, 1, True, 25.0),
        "one-site centre 4096 x 256, complex, 2 blocks": (256, 16, 1, 256, 2, True, 25.0),
        "two-site centre 512 x 4096, real, 2 blocks": (256, 2, 16, 256, 2, False, 25.0),
        "flat spectrum 4096 x 256, complex, 2 blocks": (256, 16, 1, 256, 2, True, 0.0),
        "D = 64, 512 x 64, complex": (64, 8, 1, 64, 1, True, 20.0)}.items():
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    qnl = np.sort(rng.integers(0, nq, size=(Dl, 1)), axis=0)
    qnr = (nq - 1) - np.sort(rng.integers(0, nq, size=(Dr, 1)), axis=0)[::-1] if nq > 1 else np.zeros((Dr, 1), int)
    qbl = svd_qn.add_outer(qnl, np.zeros((dl, 1), dtype=int))
    qbr = svd_qn.add_outer(np.zeros((dr, 1), dtype=int), qnr) if dr > 1 else qnr
    qntot = np.array([nq - 1])
    blocks = svd_qn.qn_blocks(qbl, qbr, qntot)
    c = np.zeros((Dl * dl, dr * Dr), dtype=complex if cplx else float)
    for blk in blocks:
        ls, rs = np.asarray(blk[2]), np.asarray(blk[3])
        c[np.ix_(ls, rs)] = spectrum_matrix(len(ls), len(rs), cplx, decay)
    C = eng.asdevice(c)
    run = lambda: svd_qn.svd_qn(C, qbl, qbr, qntot, QR=False, system="L", full_matrices=False)
    for _ in range(2):
        out = run()
    eng.sync()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        out = run()
    eng.sync()
    dt = (time.perf_counter() - t0) / n
    u, s, vt = out[0].to_host(), np.asarray(out[1]), out[3].T.to_host()
    sref = np.concatenate([np.linalg.svd(c[np.ix_(np.asarray(b[2]), np.asarray(b[3]))], compute_uv=False) for b in blocks])
    dev = np.abs(np.sort(s)[::-1] - np.sort(sref)[::-1]).max() / sref.max()
    ou = np.abs(u.conj().T @ u - np.eye(u.shape[1])).max()
    ov = np.abs(vt @ vt.conj().T - np.eye(vt.shape[0])).max()
    rec = np.abs((u * s) @ vt - c).max() / np.abs(c).max()
    shapes = ", ".join(f"{len(b[2])} x {len(b[3])}" for b in blocks)
    lines.append(f"| {name} | {shapes} | {dt * 1e3:.2f} | {dev:.1e} | {ou:.1e} | {ov:.1e} | {rec:.1e} |")
    print(lines[-1], flush=True)
txt = "\n".join(lines) + f"\n\nMPSE_SVD_GRAM={os.environ.get('MPSE_SVD_GRAM', '(default: 1)')}  MPSE_SVD_GRAM_R={os.environ.get('MPSE_SVD_GRAM_R', '(default)')}\n"
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(txt)
