#!/usr/bin/env python3
"""The environment-tensor contraction on its own (north_star: ">= 40 % MFMA utilisation on the env-tensor contraction"):
left and right environments of the headline state (50 sites, d = 2 / 16, D = 256, complex128) built site by site
exactly as the sweeps build them (mps/lib.py:200-243 in the reference; renormalizer_amd.mps.lib.Environ here), timed
with the engine idle before and after, next to the algorithmic flops of SURVEY.md section 8(d).
    python tools/env_bench.py [reps] [out.json]
Under `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` every launch of the pass is an
environment-update launch (tools/pmc_mfma_util.py then reads the MFMA-busy of exactly these)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from renormalizer_amd.engine import get_engine  # noqa: E402
from renormalizer_amd.mps.lib import Environ  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    out = sys.argv[2] if len(sys.argv) > 2 else None
    eng = get_engine()
    state = os.environ.get("ENV_BENCH_STATE", "/tmp/state.npz")
    evolved = state + ".evolved.npz"             # a generic complex state with filled bonds: one evolve, kept on disk so
    if os.path.exists(evolved):                   # that a profiled run holds environment updates only
        model, mpo, mps = bench.build_workload(25, 16, 256, 0, "physical", state_file=evolved)
    else:
        model, mpo, mps = bench.build_workload(25, 16, 256, 0, "physical", state_file=state)
        mps = mps.evolve(mpo, 10.0)
        mps.dump(evolved)
    dims, pd, wd = list(mps.bond_dims), [int(x) for x in mps.pbond_dims], list(mpo.bond_dims)
    n = len(pd)

    def flops(i):
        dl, dr, d, wl, wr = dims[i], dims[i + 1], pd[i], wd[i], wd[i + 1]
        return 8.0 * dl * dl * wl * d * dr + 4.0 * dl * dr * wl * wr * d * d + 8.0 * dl * dr * dr * wr * d

    res = {}
    tracing = bool(os.environ.get("MPSE_GEMM_TRACE"))     # per-workgroup timeline of these launches (tools/gemm_trace.py)
    if tracing:
        eng.prof_reset()
        eng.prof_enable(1)
    for dom in ("L", "R"):
        sites = range(0, n - 1) if dom == "L" else range(n - 1, 0, -1)
        fl = sum(flops(i) for i in sites)
        Environ(mps, mpo, dom)
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            Environ(mps, mpo, dom)
        eng.sync()
        dt = (time.perf_counter() - t0) / reps
        res[dom] = dict(ms=dt * 1e3, updates=len(list(sites)), gflop=fl / 1e9, tflops=fl / dt / 1e12,
                        frac_of_78_6=fl / dt / 78.6e12)
    if tracing:
        eng.prof_get()                                     # (writes the trace file)
        eng.prof_enable(False)
    res["bond_dims"] = dims
    print(json.dumps(res))
    if out:
        json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
