#!/usr/bin/env python3
"""One headline evolve with the Householder and with the Cholesky-QR block QR from the same state (GPU box): per-solve
Krylov dimensions, overlap of the two evolved states, electronic occupations, Schmidt spectra at the bonds where the
dimensions differ.  Both schemes return an isometry and U @ Vt == coef to rounding; where a block has numerically zero
singular values the isometry is not unique there, and the element-wise stopping test of the Krylov solver
(np.allclose on the local tensor, mps/tdh... evolve_utils) sees a rotated tensor.
    python tools/qr_gauge_study.py [out.md]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from renormalizer_amd.engine import get_engine  # noqa: E402


def overlap(a, b):
    e = np.ones((1, 1), complex)
    for x, y in zip(a, b):
        e = np.einsum("ab,aic,bid->cd", e, x.conj(), y, optimize=True)
    return e[0, 0]


def schmidt(arrs, bond):
    """singular values across ``bond`` (between site bond-1 and bond) by plain numpy canonicalisation"""
    arrs = [a.copy() for a in arrs]
    for i in range(bond):
        a = arrs[i]
        q, r = np.linalg.qr(a.reshape(-1, a.shape[2]))
        arrs[i] = q.reshape(a.shape[0], a.shape[1], -1)
        arrs[i + 1] = np.einsum("ab,bic->aic", r, arrs[i + 1])
    for i in range(len(arrs) - 1, bond, -1):
        a = arrs[i]
        q, r = np.linalg.qr(a.reshape(a.shape[0], -1).T)
        arrs[i] = q.T.reshape(-1, a.shape[1], a.shape[2])
        arrs[i - 1] = np.einsum("aib,cb->aic", arrs[i - 1], r)
    a = arrs[bond]
    return np.linalg.svd(a.reshape(a.shape[0], -1), compute_uv=False)


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else None
    eng = get_engine()
    model, mpo, mps = bench.build_workload(25, 16, 256, 0, "physical")
    eng.block_qr_scheme(int(os.environ.get("FIRST_EVOLVE_SCHEME", "1")))
    mps = mps.evolve(mpo, 10.0)
    res = {}
    for name, scheme in (("householder", 0), ("cholesky", 1)):
        eng.block_qr_scheme(scheme)
        s0 = eng.block_qr_stats()
        ev = mps.evolve(mpo, 10.0)
        s1 = eng.block_qr_stats()
        res[name] = dict(dims=list(ev.evolve_config.stat["steps"]), arrs=ev.to_arrays(), occ=np.asarray(ev.e_occupations),
                         e=ev.expectation(mpo), qr=tuple(b - a for a, b in zip(s0, s1)))
    eng.block_qr_scheme(-1)
    h, c = res["householder"], res["cholesky"]
    differ = [(i, a, b) for i, (a, b) in enumerate(zip(h["dims"], c["dims"])) if a != b]
    ov = overlap(h["arrs"], c["arrs"])
    lines = ["# Householder and Cholesky-QR block QR on one headline evolve (same start state, prepared with scheme "
             + os.environ.get("FIRST_EVOLVE_SCHEME", "1") + ")", "",
             f"block QR calls (all, Cholesky-QR, redone): householder {h['qr']}, cholesky {c['qr']}",
             f"|<psi_householder|psi_cholesky>| - 1 = {abs(ov) - 1.0:.2e}",
             f"max |occupation difference| = {np.abs(h['occ'] - c['occ']).max():.2e}",
             f"<H> difference = {abs(h['e'] - c['e']):.2e}",
             f"solves whose Krylov dimension differs (index, householder, cholesky): {differ}",
             f"mean Krylov dimension: {np.mean(h['dims']):.4f} / {np.mean(c['dims']):.4f}", ""]
    n = len(h["arrs"])
    bonds = set()
    for i, _, _ in differ:
        k = i - (2 * n - 1) if i >= 2 * n - 1 else i          # position inside its sweep
        site = (n - 1 - (k + 1) // 2) if i >= 2 * n - 1 else (k + 1) // 2
        bonds.update({max(site, 1), min(site + 1, n - 1)})
    lines += ["| bond | Schmidt values >= 1e-8 smax | >= 1e-12 | >= 1e-15 | smallest / largest |", "|---|---|---|---|---|"]
    for b in sorted(bonds | {n // 2}):
        s = schmidt(h["arrs"], b)
        lines.append(f"| {b} | {(s >= 1e-8 * s[0]).sum()} | {(s >= 1e-12 * s[0]).sum()} | {(s >= 1e-15 * s[0]).sum()} | {s[-1] / s[0]:.1e} |"
                     f" of {len(s)}")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
