#!/usr/bin/env python3
"""Cholesky-QR path of mpse_block_qr against NumPy on shapes and conditionings of the sweeps (GPU box).

For every case: |U Vt - A| / |A|, |iso^H iso - I|_max, exact triangularity of the other factor, which path ran
(mpse_block_qr_stats), median time per call (each call synchronised).  Exit status 1 when a case misses 1e-13.
    python tools/cholqr_check.py [out.md]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renormalizer_amd.engine import get_engine  # noqa: E402
from renormalizer_amd.mps import svd_qn  # noqa: E402

eng = get_engine()
rng = np.random.default_rng(7)
p64 = svd_qn._p64


def run(a, blocks, system, reps=10):
    """blocks: list of (row index array, column index array)"""
    nrow, ncol = a.shape
    rows = np.concatenate([b[0] for b in blocks]).astype(np.int64)
    cols = np.concatenate([b[1] for b in blocks]).astype(np.int64)
    roff = np.cumsum([0] + [len(b[0]) for b in blocks]).astype(np.int64)
    coff = np.cumsum([0] + [len(b[1]) for b in blocks]).astype(np.int64)
    K = int(sum(min(len(b[0]), len(b[1])) for b in blocks))
    A = eng.asdevice(a)
    u = eng.empty((nrow, K), a.dtype)
    vt = eng.empty((K, ncol), a.dtype)
    s0 = eng.block_qr_stats()

    def call():
        eng._check(eng.lib.mpse_block_qr(eng.ctx, A.code, A.ptr, nrow, ncol, len(blocks), p64(rows), p64(roff), p64(cols),
                                         p64(coff), int(system == "R"), u.ptr, vt.ptr, K))
    call()
    s1 = eng.block_qr_stats()
    call()
    eng.sync()
    times = []
    for _ in range(reps):       # median of individually timed calls: a pool growth (one device allocation of tens of ms)
        t0 = time.perf_counter()    # now and then lands in one of them
        call()
        eng.sync()
        times.append(time.perf_counter() - t0)
    dt = float(np.median(times))
    uh, vh = u.to_host(), vt.to_host()
    mask = np.zeros(a.shape, bool)
    for r, c in blocks:
        mask[np.ix_(r, c)] = True
    ref = a * mask
    res = np.linalg.norm(uh @ vh - ref) / max(np.linalg.norm(ref), 1e-300)
    iso = uh if system == "L" else vh.conj().T
    orth = np.abs(iso.conj().T @ iso - np.eye(iso.shape[1])).max()
    # triangularity of the other factor inside every block
    tri = 0.0
    k0 = 0
    for r, c in blocks:
        k = min(len(r), len(c))
        if system == "L":
            rf = vh[k0:k0 + k][:, c]
            tri = max(tri, np.abs(np.tril(rf, -1)).max() if k > 1 else 0.0)
        else:
            lf = uh[r][:, k0:k0 + k]
            tri = max(tri, np.abs(np.triu(lf[:k], 1)).max() if k > 1 else 0.0)   # (RQ of the adjoint: lower triangular)
        k0 += k
    path = "cholesky" if s1[1] > s0[1] and s1[2] == s0[2] else ("cholesky->householder" if s1[2] > s0[2] else "householder")
    return res, orth, tri, path, dt


def with_cond(m, n, cond, cplx=True, rank=None):
    a = rng.standard_normal((m, n)) + (1j * rng.standard_normal((m, n)) if cplx else 0)
    u, _, v = np.linalg.svd(a, full_matrices=False)
    s = np.logspace(0, -np.log10(cond), n) if cond > 1 else np.ones(n)
    if rank is not None:
        s[rank:] = 0.0
    return (u * s) @ v


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else None
    cases = []
    # two-block layouts like the headline's sites: (rows of block 0, rows of block 1, cols 0, cols 1)
    for name, m0, m1, n0, n1, cond, cplx, system in (
            ("d16 site", 2816, 1280, 145, 111, 1e6, True, "L"),
            ("d16 site RQ", 2816, 1280, 145, 111, 1e6, True, "R"),
            ("d16 site kappa 1e12", 2816, 1280, 145, 111, 1e12, True, "L"),
            ("d2 site", 256, 256, 150, 106, 1e7, True, "L"),
            ("d2 site RQ", 256, 256, 150, 106, 1e7, True, "R"),
            ("real d16", 2560, 1536, 128, 128, 1e5, False, "L"),
            ("real d2", 256, 256, 160, 96, 1e3, False, "R"),
            ("one block 4096x256", 4096, 0, 256, 0, 1e4, True, "L"),
            ("odd sizes", 1001, 333, 77, 19, 1e8, True, "L"),
            ("d16 site, 182 + 74 columns", 2608, 1488, 182, 74, 1e7, True, "L"),
            ("d2 site, 182 + 74 columns", 256, 256, 182, 74, 1e7, True, "L"),
            ("tiny n", 300, 200, 5, 1, 10, True, "L"),
            ("well conditioned (first-order pass 3)", 2816, 1280, 145, 111, 3, True, "L"),
            ("rank deficient", 2816, 1280, 145, 111, 1e3, True, "L"),
            ("kappa 1e18", 2816, 1280, 145, 111, 1e18, True, "L")):
        m, n = m0 + m1, n0 + n1
        a = np.zeros((m, n), dtype=complex if cplx else float)
        perm_r, perm_c = rng.permutation(m), rng.permutation(n)
        r0, r1 = np.sort(perm_r[:m0]), np.sort(perm_r[m0:])
        c0, c1 = np.sort(perm_c[:n0]), np.sort(perm_c[n0:])
        rank = 100 if name == "rank deficient" else None
        a[np.ix_(r0, c0)] = with_cond(m0, n0, cond, cplx, rank)
        blocks = [(r0, c0)]
        if m1 and n1:
            a[np.ix_(r1, c1)] = with_cond(m1, n1, cond, cplx)
            blocks.append((r1, c1))
        if system == "R":
            a = np.ascontiguousarray(a.conj().T)
            blocks = [(c, r) for r, c in blocks]
        res, orth, tri, path, dt = run(a, blocks, system)
        cases.append((name, a.shape, cond, system, path, res, orth, tri, dt))
    lines = ["| case | shape | cond | sys | path | residual | orthogonality | off-triangle | ms / call |", "|---|---|---|---|---|---|---|---|---|"]
    bad = 0
    for name, shape, cond, system, path, res, orth, tri, dt in cases:
        okc = res < 1e-13 and orth < 1e-13 and tri == 0.0
        bad += not okc
        lines.append(f"| {name} | {shape[0]}x{shape[1]} | {cond:.0e} | {system} | {path} | {res:.1e} | {orth:.1e} | {tri:.1e} | {dt * 1e3:.3f} |"
                     + ("" if okc else " FAIL"))
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
