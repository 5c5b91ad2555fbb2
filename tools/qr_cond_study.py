#!/usr/bin/env python3
"""Conditioning of the quantum-number blocks the block QR factorises during a headline TDVP-PS evolve, and what a
Gram / Cholesky orthogonalisation would make of them (round 5).

Every `svd_qn(QR=True)` call of one evolve (MPSE_DEFER=0 so that the centre holds its values when the Python call
happens) is intercepted: the centre is downloaded, every block gathered as the engine gathers it (adjoint for RQ) and
on the host, in float64 NumPy:
  * singular values -> condition number, numerical rank (sigma > m eps sigma_max);
  * plain CholeskyQR2  (G = B^H B, R = chol(G), Q = B R^-1, twice);
  * shifted CholeskyQR3 (first pass with s = 11 (m n + n (n + 1)) u |B|_2^2, Fukaya et al. 2020, then CholeskyQR2);
  for each: did a Cholesky factorisation break down, |Q^H Q - I|_max, |Q R - B|_F / |B|_F.
Writes a markdown table + JSON summary.  GPU box:  python tools/qr_cond_study.py gpurun_out/qr_cond [nsteps_before]"""
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


def cholqr_pass(B, shift=0.0):
    G = B.conj().T @ B
    if shift:
        G = G + shift * np.eye(G.shape[0])
    try:
        R = np.linalg.cholesky(G).conj().T
    except np.linalg.LinAlgError:
        return None, None
    Q = sla.solve_triangular(R, B.conj().T, trans="C", lower=False).conj().T   # Q = B R^-1
    return Q, R


def study(B):
    m, n = B.shape
    s = np.linalg.svd(B, compute_uv=False)
    smax = s[0] if len(s) else 0.0
    rank = int(np.sum(s > max(m, n) * np.finfo(float).eps * smax)) if smax > 0 else 0
    rec = dict(m=m, n=n, smax=float(smax), smin=float(s[-1]) if len(s) else 0.0, rank=rank,
               cond=float(smax / s[-1]) if len(s) and s[-1] > 0 else float("inf"))
    if m < n:
        rec["kind"] = "wide"
        return rec
    eye = np.eye(n)
    nb = np.linalg.norm(B)

    def finish(tag, Q, R):
        if Q is None:
            rec[tag + "_fail"] = True
            return
        rec[tag + "_fail"] = False
        rec[tag + "_orth"] = float(np.abs(Q.conj().T @ Q - eye).max())
        rec[tag + "_res"] = float(np.linalg.norm(Q @ R - B) / nb) if nb > 0 else 0.0

    # plain CholeskyQR2
    Q1, R1 = cholqr_pass(B)
    if Q1 is not None:
        Q2, R2 = cholqr_pass(Q1)
        finish("cqr2", Q2, None if Q2 is None else R2 @ R1)
    else:
        finish("cqr2", None, None)
    # shifted CholeskyQR3
    u = np.finfo(float).eps / 2
    shift = 11.0 * (m * n + n * (n + 1)) * u * smax * smax
    Q1, R1 = cholqr_pass(B, shift)
    if Q1 is not None:
        Q2, R2 = cholqr_pass(Q1)
        if Q2 is not None:
            Q3, R3 = cholqr_pass(Q2)
            finish("scqr3", Q3, None if Q3 is None else R3 @ R2 @ R1)
        else:
            finish("scqr3", None, None)
    else:
        finish("scqr3", None, None)
    return rec


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/qr_cond"
    nbefore = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    import bench
    from renormalizer_amd.mps import svd_qn as sq
    model, mpo, mps = bench.build_workload(25, 16, 256, 1234, "physical")
    for _ in range(nbefore):
        mps = mps.evolve(mpo, 10.0)
    recs = []
    orig = sq.svd_qn

    def hook(coef_array, qnbigl, qnbigr, qntot, QR=False, system=None, full_matrices=True, opt_full_matrices=True,
             plan=None):
        if QR:
            p = plan if plan is not None else sq.block_plan(qnbigl, qnbigr, qntot)
            nrow = int(np.prod(np.asarray(qnbigl).shape[:-1]))
            ncol = int(np.prod(np.asarray(qnbigr).shape[:-1]))
            a = np.asarray(coef_array.to_host()).reshape(nrow, ncol)
            for b in p["blocks"]:
                B = a[np.ix_(b[2], b[3])]
                if system == "R":
                    B = B.conj().T
                r = study(B)
                r["call"] = hook.ncall
                r["system"] = system
                recs.append(r)
            hook.ncall += 1
        return orig(coef_array, qnbigl, qnbigr, qntot, QR=QR, system=system, full_matrices=full_matrices,
                    opt_full_matrices=opt_full_matrices, plan=plan)

    hook.ncall = 0
    sq.svd_qn = hook
    mps = mps.evolve(mpo, 10.0)
    sq.svd_qn = orig
    tall = [r for r in recs if r.get("kind") != "wide"]
    summ = dict(calls=hook.ncall, blocks=len(recs), wide_blocks=len(recs) - len(tall),
                rank_deficient_blocks=sum(r["rank"] < min(r["m"], r["n"]) for r in recs),
                cond_quantiles={q: float(np.quantile([min(r["cond"], 1e300) for r in recs], q))
                                for q in (0.0, 0.1, 0.5, 0.9, 1.0)},
                cqr2_fail=sum(bool(r.get("cqr2_fail")) for r in tall),
                cqr2_bad_orth=sum((not r.get("cqr2_fail")) and r["cqr2_orth"] > 1e-12 for r in tall),
                scqr3_fail=sum(bool(r.get("scqr3_fail")) for r in tall),
                scqr3_bad_orth=sum((not r.get("scqr3_fail")) and r["scqr3_orth"] > 1e-12 for r in tall),
                calls_with_any_cqr2_problem=len({r["call"] for r in tall if r.get("cqr2_fail") or r.get("cqr2_orth", 0) > 1e-12}),
                calls_with_any_scqr3_problem=len({r["call"] for r in tall if r.get("scqr3_fail") or r.get("scqr3_orth", 0) > 1e-12}),
                evolves_before=nbefore)
    with open(out + ".json", "w") as fh:
        json.dump(dict(summary=summ, blocks=recs), fh)
    lines = ["# Block-QR inputs of one headline evolve: conditioning and Cholesky-QR emulation (float64 NumPy on the host)",
             "", "```", json.dumps(summ, indent=1), "```", "",
             "| call | sys | m | n | rank | cond | cqr2 fail | cqr2 orth | cqr2 res | scqr3 fail | scqr3 orth | scqr3 res |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in recs:
        lines.append("| %d | %s | %d | %d | %d | %.2e | %s | %s | %s | %s | %s | %s |" % (
            r["call"], r["system"], r["m"], r["n"], r["rank"], r["cond"],
            r.get("cqr2_fail", "-"), "%.1e" % r["cqr2_orth"] if "cqr2_orth" in r else "-",
            "%.1e" % r["cqr2_res"] if "cqr2_res" in r else "-",
            r.get("scqr3_fail", "-"), "%.1e" % r["scqr3_orth"] if "scqr3_orth" in r else "-",
            "%.1e" % r["scqr3_res"] if "scqr3_res" in r else "-"))
    with open(out + ".md", "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print(json.dumps(summ))


if __name__ == "__main__":
    main()
