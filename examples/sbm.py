#!/usr/bin/env python3
"""Spin-boson dynamics at zero temperature - the workflow of the reference's example/sbm.py on the MI355X engine
(Ohmic bath, adiabatically renormalised tunnelling, adaptive P&C by default; results in sbm.npz).

    python examples/sbm.py [n_phonons=300] [evolve_time=20] [tdvp]

With the third argument the job runs TDVP-PS at a fixed bond dimension of 64 instead of adaptive P&C."""
import logging
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from renormalizer_amd import CompressConfig, CompressCriteria, EvolveConfig, EvolveMethod, Quantity  # noqa: E402
from renormalizer_amd.sbm import SpinBosonDynamics, param2mollist  # noqa: E402

if __name__ == "__main__":
    n_phonons = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    evolve_time = float(sys.argv[2]) if len(sys.argv) > 2 else 20
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s")
    model = param2mollist(alpha=0.05, raw_delta=Quantity(1), omega_c=Quantity(20), renormalization_p=1, n_phonons=n_phonons)
    if len(sys.argv) > 3:
        compress_config = CompressConfig(CompressCriteria.fixed, max_bonddim=64)
        evolve_config = EvolveConfig(EvolveMethod.tdvp_ps)
    else:
        compress_config = CompressConfig(threshold=1e-4)
        evolve_config = EvolveConfig(adaptive=True, guess_dt=0.1)
    job = SpinBosonDynamics(model, compress_config=compress_config, evolve_config=evolve_config, dump_dir="./", job_name="sbm")
    job.evolve(evolve_dt=0.1, evolve_time=evolve_time)
    for t, sz in zip(job.evolve_times, job.sigma_z):
        print(f"t = {t:6.2f}  <sigma_z> = {sz:+.8f}")
