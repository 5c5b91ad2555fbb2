import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from renormalizer_amd import Model, Mpo, Mps, optimize_mps
from renormalizer_amd.model import h_qc
from renormalizer_amd.engine import get_engine
sh, aseri, nuc = h_qc.read_fcidump("tests/golden/h2o_fcidump.txt", 7)
basis, terms = h_qc.qc_model(sh, aseri)
model = Model(basis, terms)
t0 = time.perf_counter(); mpo = Mpo(model); print("mpo build", time.perf_counter() - t0)
for M in (50, 512):
    mps = Mps.random(model, [5, 5], M, percent=1.0, rng=np.random.default_rng(1))
    mps.optimize_config.procedure = [[M, 0.4], [M, 0.2], [M, 0.1], [M, 0]]
    mps.optimize_config.method = "2site"
    get_engine().sync(); t0 = time.perf_counter()
    energies, gs = optimize_mps(mps, mpo)
    get_engine().sync(); dt = time.perf_counter() - t0
    print(f"M={M}: {dt:.2f} s for {len(energies)} sweeps ({13 * len(energies)} two-site solves) -> {13 * len(energies) / dt:.2f} solves/s; E = {min(energies) + nuc:.12f}; bond dims {gs.bond_dims}")
