"""The JSON line of bench.py as the driver reads it: the latest committed line (profiles/rNN_bench_20steps.json, printed
by `python bench.py --gpus 1 --steps 20 --warmup 5` on an MI355X) carries every field of the contract, is consistent with
itself, and names BASELINE.json's metric; bench.py accepts the driver's flags.  No GPU needed."""
import glob
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest_line():
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r[0-9][0-9]_bench_20steps.json")))
    assert files, "no committed bench line"
    with open(files[-1]) as fh:
        text = fh.read().strip()
    assert "\n" not in text, "the bench prints ONE line"
    return json.loads(text)


def test_bench_line_has_the_contract_fields():
    d = _latest_line()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "c128" and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5
    assert "workload" in d["config"] and "model" not in d["config"]
    nsite = d["config"]["nsite"]
    # one step = one evolve = 2 N site updates
    assert abs(d["value"] - 2 * nsite / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    assert r["traffic"] is None or r["traffic"] >= r["compulsory_bytes_per_launch"]
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("reference", "port") and c["unit"] == d["unit"] and c["value"] > 0
    dims = d["config"]["bond_dims"]
    assert len(dims) == nsite + 1 and all(x == d["config"]["bond_dim"] for x in dims[4:-4])


def test_bench_line_names_the_baseline_metric():
    """BASELINE.json: "DMRG/TDVP sweep sites/sec at (Nsite, Dbond, dphys); % MFMA roofline" - the line quotes sweep
    site updates per second at the named (Nsite, Dbond, dphys) and carries the MFMA roofline fraction."""
    d = _latest_line()
    with open(os.path.join(REPO, "BASELINE.json")) as fh:
        base = json.load(fh)
    assert "sweep" in base["metric"] and "Nsite" in base["metric"] and "MFMA" in base["metric"]
    assert "TDVP" in d["metric"] and "sweep" in d["metric"] and "/sec" in d["metric"]
    assert all(k in d["metric"] for k in ("Nsite=", "Dbond=", "dphys="))
    assert d["roofline"]["bound"] == "mfma"


def test_bench_accepts_the_driver_flags():
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--help"], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout
