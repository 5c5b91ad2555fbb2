#!/usr/bin/env python3
"""Charge diffusion in a Holstein chain from a YAML parameter file - the workflow of the reference's
example/dynamics.py on the MI355X engine (ChargeDiffusionDynamics, adaptive TDVP-PS, results in <fname>.npz).

    python examples/dynamics.py examples/holstein_chain.yaml [max_bonddim=16] [max_steps]

holstein_chain.yaml (the parameters of the reference's example/std.yaml) describes 21 molecules at 298 K: the thermal vibrational state is prepared as a purified density operator
(every site carries an ancilla leg) and cached in <fname>_impdm.npz next to the results."""
import logging
import os
import sys

import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from renormalizer_amd import CompressConfig, CompressCriteria, EvolveConfig, EvolveMethod  # noqa: E402
from renormalizer_amd.model import load_from_dict  # noqa: E402
from renormalizer_amd.transport import ChargeDiffusionDynamics  # noqa: E402

if __name__ == "__main__":
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    with open(sys.argv[1]) as fin:
        param = yaml.safe_load(fin)
    max_bonddim = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s")
    model, temperature = load_from_dict(param, 3, False)
    cdd = ChargeDiffusionDynamics(
        model, temperature=temperature,
        compress_config=CompressConfig(CompressCriteria.fixed, max_bonddim=max_bonddim),
        evolve_config=EvolveConfig(EvolveMethod.tdvp_ps, adaptive=True, guess_dt=2),
        dump_dir=param["output dir"], job_name=param["fname"])
    cdd.custom_dump_info["comment"] = param["comment"]
    nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else param.get("nsteps")
    cdd.evolve(param.get("evolve dt"), nsteps, None if len(sys.argv) > 3 else param.get("evolve time"))
    print("time (a.u.)  <r^2>")
    for t, r2 in zip(cdd.evolve_times, cdd.r_square_array):
        print(f"{t:10.1f}  {r2:9.5f}")
