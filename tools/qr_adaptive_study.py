#!/usr/bin/env python3
"""How many passes does Cholesky-QR need on the blocks of a headline evolve when the shift of pass 1 is applied per
pivot instead of to the whole diagonal?  (round 6, verdict item 1a: adaptive pass count.)

Host emulation (float64 NumPy) of the scheme the device runs (mpse_cholqr.hip, round 6) on every block the block QR
factorises during one headline TDVP-PS evolve (captured like tools/qr_cond_study.py):

  pass 1: G = B^H B; right-looking Cholesky; a pivot d_k that has lost all but `theta` of its diagonal entry
          (d_k < theta G_kk) gets the Fukaya shift s = 11 (m n + n (n + 1)) u trace(G) added - D <= s I, so the bound
          on kappa(Q1) of the fully shifted scheme holds, and a well-conditioned block is not shifted at all;
  pass 2: G2 = Q1^H Q1; dev2 = n max|G2 - I|; plain Cholesky;  if dev2 <= tau the factorisation ends here;
  pass 3: plain Cholesky-QR of Q2.

For theta in a list (0 = shift everywhere = the round-5 scheme): blocks and calls (all blocks of a call) that finish in
two passes for tau in {0.1, 0.5}, the orthogonality reached after the pass taken as last, breakdowns.
GPU box:  python tools/qr_adaptive_study.py gpurun_out/qr_adaptive [evolves_before]"""
import json
import os
import sys

os.environ["MPSE_DEFER"] = "0"
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")
import numpy as np  # noqa: E402
import scipy.linalg as sla  # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

U_ROUND = np.finfo(float).eps / 2
THETAS = [0.0, 1e-10, 1e-11, 1e-12, 1e-13, 1e-14]


def chol_dynamic(G, theta, shift):
    """Upper factor R with R^H R = G + D, D_kk in {0, shift} chosen per pivot; (R, shifted pivots) or (None, k) on a
    non-positive pivot.  theta = 0: the shift goes onto every diagonal entry up front (round-5 scheme)."""
    n = G.shape[0]
    S = G.copy()
    diag0 = np.real(np.diag(G)).copy()
    if theta == 0.0:
        S[np.diag_indices(n)] += shift
    R = np.zeros_like(G)
    nshift = 0
    for k in range(n):
        d = S[k, k].real
        if theta > 0.0 and d < theta * diag0[k]:
            d += shift
            nshift += 1
        if not (d > 0.0):
            return None, k
        r = S[k, k:] / np.sqrt(d)
        r[0] = np.sqrt(d)
        R[k, k:] = r
        S[k + 1:, k + 1:] -= np.outer(r[1:].conj(), r[1:])
    return R, nshift


def solve(B, R):
    return sla.solve_triangular(R, B.conj().T, trans="C", lower=False).conj().T     # B R^-1


def study(B):
    m, n = B.shape
    rec = dict(m=m, n=n)
    if m < n:
        rec["kind"] = "wide"
        return rec
    s = np.linalg.svd(B, compute_uv=False)
    rec["cond"] = float(s[0] / s[-1]) if s[-1] > 0 else float("inf")
    cn = np.linalg.norm(B, axis=0)
    cn[cn == 0] = 1.0
    ss = np.linalg.svd(B / cn, compute_uv=False)
    rec["cond_scaled"] = float(ss[0] / ss[-1]) if ss[-1] > 0 else float("inf")
    eye = np.eye(n)
    G = B.conj().T @ B
    shift = 11.0 * (m * n + n * (n + 1)) * U_ROUND * np.trace(G).real
    for th in THETAS:
        tag = "t%g" % th
        R1, ns = chol_dynamic(G, th, shift)
        if R1 is None:
            rec[tag] = dict(fail=1)
            continue
        Q1 = solve(B, R1)
        G2 = Q1.conj().T @ Q1
        dev2 = n * float(np.abs(G2 - eye).max())
        R2, _ = chol_dynamic(G2, 1.0, 0.0) if False else (None, None)
        try:
            R2 = np.linalg.cholesky(G2).conj().T
        except np.linalg.LinAlgError:
            rec[tag] = dict(fail=2, nshift=ns, dev2=dev2)
            continue
        Q2 = solve(Q1, R2)
        G3 = Q2.conj().T @ Q2
        orth2 = float(np.abs(G3 - eye).max())
        out = dict(fail=0, nshift=ns, dev2=dev2, orth2=orth2)
        try:
            R3 = np.linalg.cholesky(G3).conj().T
            Q3 = solve(Q2, R3)
            out["orth3"] = float(np.abs(Q3.conj().T @ Q3 - eye).max())
        except np.linalg.LinAlgError:
            out["fail"] = 3
        rec[tag] = out
    return rec


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/qr_adaptive"
    nbefore = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    import bench
    from renormalizer_amd.mps import svd_qn as sq
    model, mpo, mps = bench.build_workload(25, 16, 256, 1234, "physical")
    for _ in range(nbefore):
        mps = mps.evolve(mpo, 10.0)
    recs = []
    orig = sq.svd_qn

    def hook(coef_array, qnbigl, qnbigr, qntot, QR=False, system=None, plan=None, **kw):
        if QR:
            p = plan if plan is not None else sq.block_plan(qnbigl, qnbigr, qntot)
            nrow = int(np.prod(np.asarray(qnbigl).shape[:-1]))
            ncol = int(np.prod(np.asarray(qnbigr).shape[:-1]))
            a = np.asarray(coef_array.to_host()).reshape(nrow, ncol)
            shapes = []
            for b in p["blocks"]:
                B = a[np.ix_(b[2], b[3])]
                if system == "R":
                    B = B.conj().T
                shapes.append(B.shape)
            # the device takes the Cholesky-QR path for a call whose blocks are all tall, <= 256 columns, tallest >= 256
            # rows, widest >= 96 columns (cholqr_eligible)
            elig = (all(mm >= nn and nn <= 256 for mm, nn in shapes) and max(mm for mm, _ in shapes) >= 256
                    and max(nn for _, nn in shapes) >= 96)
            for b in p["blocks"]:
                B = a[np.ix_(b[2], b[3])]
                if system == "R":
                    B = B.conj().T
                r = study(np.ascontiguousarray(B)) if elig else dict(m=B.shape[0], n=B.shape[1], kind="householder")
                r["call"] = hook.ncall
                r["system"] = system
                recs.append(r)
            hook.ncall += 1
        return orig(coef_array, qnbigl, qnbigr, qntot, QR=QR, system=system, plan=plan, **kw)

    hook.ncall = 0
    sq.svd_qn = hook
    mps = mps.evolve(mpo, 10.0)
    sq.svd_qn = orig
    el = [r for r in recs if "kind" not in r]
    calls = sorted({r["call"] for r in el})
    summ = dict(calls=hook.ncall, eligible_calls=len(calls), eligible_blocks=len(el), evolves_before=nbefore, by_theta={})
    for th in THETAS:
        tag = "t%g" % th
        row = {}
        for tau in (0.1, 0.5):
            two = lambda r: r[tag]["fail"] == 0 and r[tag]["dev2"] <= tau                # noqa: E731
            row["blocks_2pass_tau%g" % tau] = sum(two(r) for r in el)
            row["calls_2pass_tau%g" % tau] = sum(all(two(r) for r in el if r["call"] == c) for c in calls)
            o = [r[tag]["orth2"] for r in el if two(r)]
            row["worst_orth_2pass_tau%g" % tau] = max(o) if o else None
        row["fail1"] = sum(r[tag]["fail"] == 1 for r in el)
        row["fail2"] = sum(r[tag]["fail"] == 2 for r in el)
        row["fail3"] = sum(r[tag]["fail"] == 3 for r in el)
        o3 = [r[tag]["orth3"] for r in el if r[tag].get("orth3") is not None]
        row["worst_orth_3pass"] = max(o3) if o3 else None
        row["blocks_shifted"] = sum(r[tag].get("nshift", 0) > 0 for r in el)
        summ["by_theta"][tag] = row
    with open(out + ".json", "w") as fh:
        json.dump(dict(summary=summ, blocks=recs), fh)
    lines = ["# Adaptive pass count of Cholesky-QR on the blocks of one headline evolve (host emulation, float64)", "",
             "```", json.dumps(summ, indent=1), "```", "",
             "| call | sys | m | n | cond | cond (unit columns) | " + " | ".join("th=%g: shifted pivots, dev2, orth2" % t for t in THETAS) + " |",
             "|---|---|---|---|---|---|" + "---|" * len(THETAS)]
    for r in el:
        cells = []
        for th in THETAS:
            x = r["t%g" % th]
            cells.append("fail %d" % x["fail"] if x["fail"] in (1, 2) else "%d, %.1e, %.1e" % (x["nshift"], x["dev2"], x["orth2"]))
        lines.append("| %d | %s | %d | %d | %.1e | %.1e | %s |" % (r["call"], r["system"], r["m"], r["n"], r["cond"],
                                                                    r["cond_scaled"], " | ".join(cells)))
    with open(out + ".md", "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print(json.dumps(summ))


if __name__ == "__main__":
    main()
