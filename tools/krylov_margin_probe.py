#!/usr/bin/env python3
"""Is a Krylov dimension that differs between the device and the oracle a matter of gauge, or of the solver?

One headline evolve in the oracle, with every local solve recorded: (L, W, R, start vector, dt) -> (Krylov dimension,
margins of its stopping tests).  Every solve that the oracle's own test decided within a factor 2.2 of its threshold is
then run AGAIN by the device's Lanczos exponential on the oracle's very inputs (same gauge, same numbers): if the device
still stops elsewhere the cause is the solver's arithmetic, not the completion of an isometry by a QR.
For those solves the oracle's solver is also repeated with the stopping tolerance scaled by 0.5 and 2 (does the
dimension move?) and the device's result is compared element by element.
GPU box:  python tools/krylov_margin_probe.py"""
import ctypes as C
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import bench  # noqa: E402
from oracle import mps_oracle as orc  # noqa: E402
from test_engine_gpu import dev_expm  # noqa: E402
from test_headline_gpu import _oracle_state, _solve_sites  # noqa: E402
from renormalizer_amd.engine import get_engine  # noqa: E402

model, mpo, mps = bench.build_workload(25, 16, 256, 0, "physical")
mps = mps.evolve(mpo, 10.0)
nev = int(sys.argv[1]) if len(sys.argv) > 1 else 1
w_host = [mpo[i] for i in range(len(mpo))]
ost = _oracle_state(model, mps)
eng = get_engine()

rec = []
last = {}
orig_hop, orig_expm = orc.hop_apply, orc.expm_krylov


def hop_spy(l, r, cmo, c):
    last["l"], last["r"], last["cmo"], last["shape"] = l, r, cmo, c.shape
    return orig_hop(l, r, cmo, c)


def expm_spy(Afunc, dt, vstart, margins=None, **kw):
    m = [] if margins is None else margins
    out, k = orig_expm(Afunc, dt, vstart, margins=m, **kw)
    entry = dict(k=k, margins=list(m))
    if any(1 / 2.2 <= x <= 2.2 for x in m):
        entry.update(l=last["l"].copy(), r=last["r"].copy(), cmo=[w.copy() for w in last["cmo"]], shape=last["shape"],
                     v=np.array(vstart).copy(), dt=dt, out=out.copy())
    rec.append(entry)
    return out, k


orc.hop_apply, orc.expm_krylov = hop_spy, expm_spy
for ev in range(nev):
    rec.clear()
    where = _solve_sites(len(mps), bool(ost.to_right))
    ost = orc.tdvp_ps_step(ost, w_host, 10.0)
    print(f"evolve {ev}: {len(rec)} solves, {sum('l' in e for e in rec)} marginal", flush=True)
    for i, e in enumerate(rec):
        if "l" not in e:
            continue
        c = e["v"].reshape(e["shape"])
        out_d, k_d = dev_expm(eng, e["l"], e["r"], e["cmo"], c, e["dt"])
        f = lambda y: orig_hop(e["l"], e["r"], e["cmo"], y.reshape(e["shape"])).ravel()   # noqa: E731
        k_half = orig_expm(f, e["dt"], e["v"], rtol=0.5e-5, atol=0.5e-8)[1]
        k_dbl = orig_expm(f, e["dt"], e["v"], rtol=2e-5, atol=2e-8)[1]
        err = np.abs(out_d.ravel() - e["out"])
        print(f"  solve {i} {where[i]} shape {e['shape']}: oracle k {e['k']} margins {['%.3g' % x for x in e['margins']]} | "
              f"device on the SAME inputs k {k_d} | oracle with tol x0.5: {k_half}, x2: {k_dbl} | "
              f"max|dev - oracle| {err.max():.2e} (|x|max {np.abs(e['out']).max():.2e})", flush=True)
