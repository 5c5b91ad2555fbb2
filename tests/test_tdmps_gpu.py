"""``TdMpsJob`` on the device: the example's charge-diffusion job writes the reference's dump keys
(transport/dynamics.py:250-267) after every step and a state checkpoint that ``Mps.load`` reads back.  pytest -m gpu."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_charge_diffusion_job_dumps_reference_keys(tmp_path):
    from renormalizer_amd import Mps
    spec = importlib.util.spec_from_file_location(
        "holstein_dynamics", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "examples", "holstein_dynamics.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    job = mod.ChargeDiffusion(5, 4, 8, dump_dir=str(tmp_path), job_name="cd", dump_mps="one")
    job.evolve(evolve_dt=10.0, nsteps=3)
    z = np.load(tmp_path / "cd.npz")
    for key in ("tempearture", "total time", "r square array", "electron occupations array",
                "phonon occupations array", "bond entropy", "time series"):
        assert key in z.files, key
    assert z["time series"].tolist() == [0, 10.0, 20.0, 30.0] and float(z["total time"]) == 30.0
    occ = z["electron occupations array"]
    assert occ.shape == (4, 5) and np.abs(occ.sum(axis=1) - 1).max() < 1e-9 and occ[0, 2] > 1 - 1e-9
    assert abs(z["r square array"][0]) < 1e-12 and np.all(np.diff(z["r square array"]) > 0)          # the carrier spreads
    assert np.abs(z["energies"] - z["energies"][0]).max() < 1e-6                           # TDVP conserves <H>
    again = Mps.load(job.model, str(tmp_path / "cd_mps.npz"))
    assert np.abs(np.asarray(again.e_occupations) - occ[-1]).max() < 1e-12
