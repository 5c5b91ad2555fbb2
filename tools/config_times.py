#!/usr/bin/env python3
"""Wall time per Mps.evolve / DMRG run of the BASELINE.json configs on one MI355X, next to BASELINE.md's reference
(CPU, 4 threads) figures.  Usage: python tools/config_times.py [out.md]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from renormalizer_amd import (CompressConfig, CompressCriteria, EvolveConfig, EvolveMethod, HolsteinModel, Model, Mol, Mpo,  # noqa: E402
                              Mps, Op, Phonon, Quantity, optimize_mps)
from renormalizer_amd.engine import get_engine  # noqa: E402
from renormalizer_amd.model import h_qc  # noqa: E402
from renormalizer_amd.sbm import param2model  # noqa: E402

eng = get_engine()
rows = []


def timed_evolves(mps, mpo, dt, n):
    mps = mps.evolve(mpo, dt)                 # warm-up (allocator pool, qn plan cache)
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        mps = mps.evolve(mpo, dt)
    eng.sync()
    return (time.perf_counter() - t0) / n, mps


# config 2: spin-boson, 20 modes, d = 8, D = 64
model, _ = param2model(0.05, Quantity(1), Quantity(20), 1, 20, 8)
mpo = Mpo(model)
mps = Mps.ground_state(model, False)
mps.compress_config = CompressConfig(CompressCriteria.fixed, max_bonddim=64)
mps.evolve_config = EvolveConfig(EvolveMethod.tdvp_ps)
mps = mps.expand_bond_dimension(mpo, coef=1e-16, include_ex=False)
t, _ = timed_evolves(mps, mpo, 0.1, 5)
rows.append(("#2 spin-boson 21 sites, d = 2/8, D = 64, TDVP-PS", t, 2 * 21 / t, "1.88 s, 22.4/s"))

# config 3 at three sizes
for nmol, D, ref in ((10, 64, "3.73 s, 10.7/s"), (10, 128, "31.6 s, 1.27/s"), (25, 256, "214.7 s, 0.47/s")):
    ph = Phonon.simple_phonon(Quantity(6.128e-3), Quantity(16.274571056529368), 16)
    model = HolsteinModel([Mol(Quantity(0), [ph])] * nmol, Quantity(3.0e-2), 3)
    psi = Mpo.onsite(model, r"a^\dagger", dof_set={nmol // 2}).apply(Mps.ground_state(model, False))
    mpo = Mpo(model, offset=Quantity(psi.expectation(Mpo(model))))
    psi.compress_config = CompressConfig(CompressCriteria.fixed, max_bonddim=D)
    psi.evolve_config = EvolveConfig(EvolveMethod.tdvp_ps)
    eng.sync()
    t0 = time.perf_counter()
    psi = psi.expand_bond_dimension(mpo).canonicalise()
    eng.sync()
    texp = time.perf_counter() - t0
    t, _ = timed_evolves(psi, mpo, 10.0, 3)
    rows.append((f"#3 Holstein {2 * nmol} sites, d = 2/16, D = {D}, TDVP-PS (bond expansion {texp:.1f} s)", t, 4 * nmol / t, ref))

# config 4: FMO, 7 sites x 35 modes, D = 32 (T = 0: 252 sites; thermofield at 77 K: 497 sites)
import importlib.util  # noqa: E402
spec = importlib.util.spec_from_file_location("fmo_example", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "fmo.py"))
fmo = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fmo)
for temp in (0.0, 77.0):
    model = fmo.fmo_model(35, temperature_k=temp)
    psi = Mpo.onsite(model, r"a^\dagger", dof_set={model.mol_num // 2}).apply(Mps.ground_state(model, False))
    mpo = Mpo(model, offset=Quantity(psi.expectation(Mpo(model))))
    psi.compress_config = CompressConfig(CompressCriteria.fixed, max_bonddim=32)
    psi.evolve_config = EvolveConfig(EvolveMethod.tdvp_ps)
    psi = psi.expand_bond_dimension(mpo).canonicalise()
    t, _ = timed_evolves(psi, mpo, 160.0, 3)
    rows.append((f"#4 FMO {'T = 0' if temp == 0 else 'thermofield 77 K'}: {len(mpo)} sites, d <= {max(model.pbond_list)}, D = 32, "
                 f"TDVP-PS, one trajectory", t, 2 * len(mpo) / t, "n/a"))

# config 5: H2O STO-3G DMRG
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sh, aseri, nuc = h_qc.read_fcidump(os.path.join(repo, "tests", "golden", "h2o_fcidump.txt"), 7)
basis, terms = h_qc.qc_model(sh, aseri)
model = Model(basis, terms)
mpo = Mpo(model)
for M in (50, 512):
    mps = Mps.random(model, [5, 5], M, percent=1.0, rng=np.random.default_rng(1))
    mps.optimize_config.procedure = [[M, 0.4], [M, 0.2], [M, 0.1], [M, 0]]
    mps.optimize_config.method = "2site"
    eng.sync()
    t0 = time.perf_counter()
    energies, gs = optimize_mps(mps, mpo)
    eng.sync()
    t = time.perf_counter() - t0
    rows.append((f"#5 H2O STO-3G 2-site DMRG, M = {M} (effective bonds <= {max(gs.bond_dims)}), 4 sweeps = 52 solves, "
                 f"E = {min(energies) + nuc:.9f}", t, 52 / t, "13.5 s, 3.9/s (M = 50)"))

lines = ["| config | this engine, 1 MI355X: s per evolve (DMRG: per run) | site-updates/s | reference, 4 CPU threads (BASELINE.md) |",
         "|---|---|---|---|"]
for name, t, rate, ref in rows:
    lines.append(f"| {name} | {t:.3f} | {rate:.1f} | {ref} |")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(out + "\n")
