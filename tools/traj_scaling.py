#!/usr/bin/env python3
"""Independent trajectories sharing ONE GPU at a launch-latency-bound size: BASELINE config 4 (FMO complex, thermofield
at 77 K: 497 sites, D = 32, TDVP-PS), every trajectory a static-disorder realisation with its own seed.

    python tools/traj_scaling.py threads T [steps]      T host threads, one engine context + stream each (one process)
    python tools/traj_scaling.py procs T [steps]        T processes, one context each

Prints one JSON line: aggregate site-updates/s over all trajectories (the slowest trajectory's wall time)."""
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "examples"))


def prepare(unit, D=32):
    import fmo
    from renormalizer_amd import CompressConfig, CompressCriteria, EvolveConfig, EvolveMethod, Mpo, Mps, Quantity
    from renormalizer_amd.parallel import trajectory_seed
    rng = np.random.default_rng(trajectory_seed(7, unit))
    model = fmo.fmo_model(35, disorder_cm=50.0 if unit else 0.0, rng=rng, temperature_k=77.0)
    psi = Mpo.onsite(model, r"a^\dagger", dof_set={model.mol_num // 2}).apply(Mps.ground_state(model, False))
    mpo = Mpo(model, offset=Quantity(psi.expectation(Mpo(model))))
    psi.compress_config = CompressConfig(CompressCriteria.fixed, max_bonddim=D)
    psi.evolve_config = EvolveConfig(EvolveMethod.tdvp_ps)
    psi = psi.expand_bond_dimension(mpo).canonicalise()
    return mpo, psi


def run_threads(T, steps):
    from renormalizer_amd.engine import Engine, get_engine, use_engine
    engines = [get_engine()] + [Engine(0) for _ in range(T - 1)]
    sync = threading.Barrier(T + 1)
    nsite = [0] * T

    def traj(t):
        use_engine(engines[t])
        mpo, psi = prepare(t)
        psi = psi.evolve(mpo, 160.0)          # warm-up
        engines[t].sync()
        nsite[t] = len(psi)
        sync.wait()
        for _ in range(steps):
            psi = psi.evolve(mpo, 160.0)
        engines[t].sync()
        sync.wait()

    th = [threading.Thread(target=traj, args=(t,)) for t in range(T)]
    for x in th:
        x.start()
    sync.wait()
    t0 = time.perf_counter()
    sync.wait()
    dt = time.perf_counter() - t0
    for x in th:
        x.join()
    return sum(2 * n * steps for n in nsite) / dt, dt


def main():
    mode, T = sys.argv[1], int(sys.argv[2])
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    if mode == "threads":
        rate, dt = run_threads(T, steps)
    elif mode == "one":                         # worker of the process mode: waits for the start file
        unit, start = int(sys.argv[4]), sys.argv[5]
        mpo, psi = prepare(unit)
        psi = psi.evolve(mpo, 160.0)
        from renormalizer_amd.engine import get_engine
        get_engine().sync()
        open(f"{start}.ready{unit}", "w").close()
        while not os.path.exists(start):
            time.sleep(0.001)
        t0 = time.perf_counter()
        for _ in range(steps):
            psi = psi.evolve(mpo, 160.0)
        get_engine().sync()
        print(json.dumps({"sec": time.perf_counter() - t0, "nsite": len(psi)}))
        return
    else:
        start = f"/tmp/traj_start_{os.getpid()}"
        ps = [subprocess.Popen([sys.executable, __file__, "one", str(T), str(steps), str(u), start],
                               stdout=subprocess.PIPE, text=True) for u in range(T)]
        while not all(os.path.exists(f"{start}.ready{u}") for u in range(T)):
            time.sleep(0.01)
        open(start, "w").close()
        outs = [json.loads(p.communicate()[0].strip().splitlines()[-1]) for p in ps]
        dt = max(o["sec"] for o in outs)
        rate = sum(2 * o["nsite"] * steps for o in outs) / dt
    print(json.dumps({"workload": "configs[3]: FMO thermofield 77 K, 497 sites, D = 32, TDVP-PS", "mode": mode,
                      "trajectories_on_one_gpu": T, "steps": steps, "site_updates_per_s": rate, "wall_s": dt}))


if __name__ == "__main__":
    main()
