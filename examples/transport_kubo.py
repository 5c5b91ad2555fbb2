#!/usr/bin/env python3
"""Mobility of a Holstein chain from the Green-Kubo current autocorrelation function - the workflow of the
reference's example/transport_kubo.py on the MI355X engine (TransportKubo; results in <fname>_autocorr.npz).

    python examples/transport_kubo.py examples/holstein_chain.yaml [max_steps]"""
import logging
import os
import sys

import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from renormalizer_amd import CompressConfig, EvolveConfig  # noqa: E402
from renormalizer_amd.model import load_from_dict  # noqa: E402
from renormalizer_amd.transport import TransportKubo  # noqa: E402

if __name__ == "__main__":
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    with open(sys.argv[1]) as fin:
        param = yaml.safe_load(fin)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s")
    model, temperature = load_from_dict(param, 3, False)
    ct = TransportKubo(model, temperature=temperature, compress_config=CompressConfig(threshold=1e-4),
                       ievolve_config=EvolveConfig(adaptive=True, guess_dt=temperature.to_beta() / 1000j),
                       evolve_config=EvolveConfig(adaptive=True, guess_dt=2),
                       dump_dir=param["output dir"], job_name=param["fname"] + "_autocorr")
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else param.get("nsteps")
    ct.evolve(param.get("evolve dt"), nsteps, None if len(sys.argv) > 2 else param.get("evolve time"))
    print("time (a.u.)  C(t)")
    for t, c in zip(ct.evolve_times, ct.auto_corr):
        print(f"{t:10.1f}  {c.real:+.6e} {c.imag:+.6e}j")
    print("mobility (cm^2/Vs):", ct.calc_mobility()[1])
