#!/usr/bin/env python3
"""Launch-bound configurations (BASELINE configs 2 and 4, and the 20-site Holstein chain at D = 64) timed in THIS process
with the environment as given: run once with MPSE_SMALL=0 (three-step plans everywhere) and once with the default
(one-launch matvec of small centres, mpse_small.hip) to compare.  Prints one JSON line per configuration with the
rate, the energy and an observable (the two runs must agree to ~1e-10).
    python tools/small_ab.py [sbm] [fmo] [fmo77] [holstein]"""
import importlib.util
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from renormalizer_amd import (CompressConfig, CompressCriteria, EvolveConfig, EvolveMethod, HolsteinModel, Mol, Mpo,  # noqa: E402
                              Mps, Phonon, Quantity)
from renormalizer_amd.engine import get_engine  # noqa: E402
from renormalizer_amd.sbm import param2model  # noqa: E402

eng = get_engine()
which = sys.argv[1:] or ["sbm", "fmo", "holstein"]


def timed(mps, mpo, dt, n, name, nupd):
    mps = mps.evolve(mpo, dt)
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        mps = mps.evolve(mpo, dt)
    eng.sync()
    t = (time.perf_counter() - t0) / n
    print(json.dumps({"config": name, "MPSE_SMALL": os.environ.get("MPSE_SMALL", "default"), "s_per_evolve": t,
                      "site_updates_per_s": nupd / t, "energy": float(np.real(mps.expectation(mpo))),
                      "norm": float(np.real(mps.norm))}), flush=True)


if "sbm" in which:
    model, _ = param2model(0.05, Quantity(1), Quantity(20), 1, 20, 8)
    mpo = Mpo(model)
    mps = Mps.ground_state(model, False)
    mps.compress_config = CompressConfig(CompressCriteria.fixed, max_bonddim=64)
    mps.evolve_config = EvolveConfig(EvolveMethod.tdvp_ps)
    mps = mps.expand_bond_dimension(mpo, coef=1e-16, include_ex=False)
    timed(mps, mpo, 0.1, 10, "#2 spin-boson 21 sites d=2/8 D=64", 2 * 21)

for key, D in (("holstein", 64), ("holstein128", 128)):
    if key not in which:
        continue
    nmol = 10
    ph = Phonon.simple_phonon(Quantity(6.128e-3), Quantity(16.274571056529368), 16)
    model = HolsteinModel([Mol(Quantity(0), [ph])] * nmol, Quantity(3.0e-2), 3)
    psi = Mpo.onsite(model, r"a^\dagger", dof_set={nmol // 2}).apply(Mps.ground_state(model, False))
    mpo = Mpo(model, offset=Quantity(psi.expectation(Mpo(model))))
    psi.compress_config = CompressConfig(CompressCriteria.fixed, max_bonddim=D)
    psi.evolve_config = EvolveConfig(EvolveMethod.tdvp_ps)
    psi = psi.expand_bond_dimension(mpo).canonicalise()
    timed(psi, mpo, 10.0, 5, f"#3 Holstein 20 sites d=2/16 D={D}", 4 * nmol)

for key, temp in (("fmo", 0.0), ("fmo77", 77.0)):
    if key not in which:
        continue
    spec = importlib.util.spec_from_file_location("fmo_example", os.path.join(REPO, "examples", "fmo.py"))
    fmo = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fmo)
    model = fmo.fmo_model(35, temperature_k=temp)
    psi = Mpo.onsite(model, r"a^\dagger", dof_set={model.mol_num // 2}).apply(Mps.ground_state(model, False))
    mpo = Mpo(model, offset=Quantity(psi.expectation(Mpo(model))))
    psi.compress_config = CompressConfig(CompressCriteria.fixed, max_bonddim=32)
    psi.evolve_config = EvolveConfig(EvolveMethod.tdvp_ps)
    psi = psi.expand_bond_dimension(mpo).canonicalise()
    timed(psi, mpo, 160.0, 3, f"#4 FMO {'T=0' if temp == 0 else '77 K thermofield'} {len(mpo)} sites D=32", 2 * len(mpo))
