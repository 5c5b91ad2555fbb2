#!/usr/bin/env python3
"""Headline benchmark: TDVP-PS sweep throughput on the 50-site Holstein chain (BASELINE.json
configs[2]: 25 molecules x 1 mode, dphys = 2/16 alternating, Dbond = 256, complex128, dt = 10 a.u.).

    python bench.py --gpus N --steps K --warmup W          (N > 1: one process per GPU, e.g. launched by
                                                            python -m torch.distributed.run; RANK / LOCAL_RANK /
                                                            WORLD_SIZE / MASTER_PORT are read from the environment)

A "step" is one ``Mps.evolve(mpo, dt)`` = two half sweeps = 2*Nsite site updates (forward solves)
plus 2*(Nsite-1) bond back-steps, on one independent trajectory per GPU.  The MPS, the MPO and all
environments are resident in HBM before the timed region starts.  Rank 0 prints ONE JSON line:
    value      = site updates per second, all ranks together (weak scaling: one trajectory per GPU)
    roofline   = FP64-MFMA roofline of the dominant kernel (complex x complex contraction), measured
                 with HIP events on the engine's stream over the timed region: `achieved` counts the MFMA
                 work actually issued (visited K tiles x 6 real flops per complex multiply-add of the 3M
                 scheme), `dense_equiv_tflops` the algorithmic 8 M N K of SURVEY.md 8(d) (no fraction: it exceeds the
                 peak by kernel time - empty K tiles are skipped); `classes` adds the
                 HBM-bound Lanczos vector kernels and the block QR
No PyTorch: ranks talk through librccl.so bound with ctypes (renormalizer_amd/parallel.py), only for the barrier,
the max of the elapsed time and ONE all-gather of the observables.
    cpu_baseline = the NumPy/SciPy restatement of the reference path (oracle/) timed on the host on a
                 bounded sample of the same workload (4 BLAS threads, like RENO_NUM_THREADS=4)
"""
import argparse
import json
import os
import sys
import time

# BLAS threads for the CPU baseline must be fixed before numpy is imported
# (mirrors renormalizer/__init__.py:9-22 with RENO_NUM_THREADS=4)
_CPU_THREADS = int(os.environ.get("RENO_NUM_THREADS", "4"))
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
    os.environ[_v] = str(_CPU_THREADS)

import numpy as np  # noqa: E402

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PROF_STRIDE = int(os.environ.get("MPSE_BENCH_PROF_STRIDE", "23"))   # HIP-event timing of every 23rd launch of each class inside the timed region (every 7th costs 1.7 % of the step)
FP64_MFMA_PEAK_TFLOPS = 78.6      # MI355X FP64 matrix peak (AMD datasheet)
HBM_PEAK_GBS = 8000.0             # MI355X HBM3E (MI355X_MICROARCH.md)


def _roctx():
    """ROCTx pause / resume hooks: under `rocprofv3 --selected-regions` only the timed region is profiled, so the
    committed kernel-trace summary (profiles/) covers exactly the launches the roofline numbers are taken from.
    No-ops when the library is absent or no profiler is attached."""
    import ctypes
    try:
        lib = ctypes.CDLL("librocprofiler-sdk-roctx.so")
        lib.roctxProfilerResume.argtypes = [ctypes.c_uint64]
        lib.roctxProfilerPause.argtypes = [ctypes.c_uint64]
        return lib
    except (OSError, AttributeError):
        return None


DISORDER_AU = 50.0 * 4.556335e-6   # static diagonal disorder of the trajectories beyond the first: sigma = 50 cm^-1


def _pci_bus_id(device):
    """PCI bus id of a HIP device as one integer (domain << 24 | bus << 16 | device << 8 | function), -1 if unknown:
    the JSON line shows that N ranks ran on N distinct devices."""
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(device)) != 0:
            return -1
        dom, bus, rest = buf.value.decode().split(":")
        dev, fn = rest.split(".")
        return (int(dom, 16) << 24) | (int(bus, 16) << 16) | (int(dev, 16) << 8) | int(fn, 16)
    except (OSError, ValueError, AttributeError):
        return -1


def _kernel_source_sha():
    """Fingerprint of the contraction kernel's sources: a committed PMC traffic figure is quoted as `roofline.traffic`
    only while it was measured on these very sources (tools/pmc_traffic.py stores the same fingerprint)."""
    import hashlib
    h = hashlib.sha256()
    for name in ("mpse_gemm.hip", "mpse_plans.h", "mpse_contract.hip"):
        try:
            with open(os.path.join(REPO, "renormalizer_amd", "csrc", name), "rb") as fh:
                h.update(fh.read())
        except OSError:
            return None
    return h.hexdigest()[:16]


def _source_sha(*names):
    """Fingerprint of the named csrc files (a committed counter figure is quoted only for the sources it was measured on)."""
    import hashlib
    h = hashlib.sha256()
    for name in names:
        try:
            with open(os.path.join(REPO, "renormalizer_amd", "csrc", name), "rb") as fh:
                h.update(fh.read())
        except OSError:
            return None
    return h.hexdigest()[:16]


def _committed_mfma_busy(kernel, *sources):
    """(busy fraction by duration, by GUI-active cycles, source note) of ``kernel`` from the newest committed
    profiles/rNN_pmc_mfma_busy.json (tools/pmc_mfma_util.py --json on a rocprofv3 --pmc pass of this command), or
    (None, None, note) when the file is missing or was measured on other sources of that kernel."""
    import glob
    import hashlib
    try:
        f = sorted(glob.glob(os.path.join(REPO, "profiles", "r[0-9][0-9]_pmc_mfma_busy.json")))[-1]
        with open(f, "rb") as fh:
            raw = fh.read()
        d = json.loads(raw)
        k = d["kernels"][kernel]
        note = f"profiles/{os.path.basename(f)} sha256:{hashlib.sha256(raw).hexdigest()[:16]} (SQ_VALU_MFMA_BUSY_CYCLES, launch-time weighted)"
        if k.get("source_sha") != _source_sha(*sources):
            return None, None, note + "; kernel sources CHANGED since that pass: figure withheld"
        return k["busy_by_duration"], k["busy_by_gui_active"], note
    except (OSError, KeyError, ValueError, IndexError):
        return None, None, None


def build_workload(nmol, pdim, bond_dim, seed, init, unit=0, state_file=None, scheme="tdvp_ps"):
    from renormalizer_amd import (HolsteinModel, Phonon, Mol, Quantity, Mpo, CompressConfig, CompressCriteria,
                                  EvolveConfig, EvolveMethod)
    from renormalizer_amd.mps.mps import Mps
    from renormalizer_amd.parallel import trajectory_seed
    # example/std.yaml parameters, T = 0, fixed phonon levels.  Trajectory 0 is the
    # clean chain; every further independent trajectory (other ranks, other streams of a GPU) is a static-disorder
    # realisation of it with its own seed - same sizes, different numbers
    ph = Phonon.simple_phonon(Quantity(6.128e-3), Quantity(16.274571056529368), pdim)
    eps = np.zeros(nmol)
    if unit > 0:
        eps = np.random.default_rng(trajectory_seed(2024, unit)).normal(0.0, DISORDER_AU, size=nmol)
    model = HolsteinModel([Mol(Quantity(float(e)), [ph]) for e in eps], Quantity(3.0e-2), 3)
    fc = Mps.hartree_product_state(model, {nmol // 2: 1})          # electron created on the centre molecule
    e0 = fc.expectation(Mpo(model))
    mpo = Mpo(model, offset=Quantity(e0))
    if state_file and os.path.exists(state_file):
        mps = Mps.load(model, state_file)
    elif init == "random":
        mps = Mps.random(model, 1, bond_dim, percent=1.0, rng=np.random.default_rng(seed))
    elif init == "physical":
        # transport/dynamics.py:173-199: vacuum, electron created on the centre molecule, bonds expanded
        # with states reachable through H (device-resident apply / add / QR / SVD sweeps)
        mps = Mpo.onsite(model, r"a^\dagger", dof_set={nmol // 2}).apply(Mps.ground_state(model, False))
        mps.compress_config = CompressConfig(CompressCriteria.fixed, max_bonddim=bond_dim)
        mps = mps.expand_bond_dimension(mpo)
        mps.canonicalise()
    else:
        raise SystemExit(f"unknown --init {init}")
    if state_file and not os.path.exists(state_file):
        mps.dump(state_file)
    mps.compress_config = CompressConfig(CompressCriteria.fixed, max_bonddim=bond_dim)
    mps.evolve_config = EvolveConfig(EvolveMethod.tdvp_ps2 if scheme == "tdvp_ps2" else EvolveMethod.tdvp_ps)
    return model, mpo, mps


def cpu_baseline(model, mpo, mps, dt, n_updates):
    """Oracle (NumPy/SciPy restatement of the reference CPU path) on the first ``n_updates`` site
    updates of the same evolve; per-shape times are extrapolated to the full 2*Nsite updates."""
    from oracle import mps_oracle as orc
    sites = mps.to_arrays()
    st = orc.MpsState(sites, [q.copy() for q in mps.qn], mps.qnidx, mps.qntot.copy(), mps.to_right,
                      [np.array(b.sigmaqn) for b in model.basis])
    w = [mpo[i] for i in range(len(mpo))]
    timings = []
    t0 = time.perf_counter()
    orc.tdvp_ps_step(st, w, dt, normalize_after=False, max_updates=n_updates, timings=timings,
                     env_domain="R" if st.to_right else "L")
    wall = time.perf_counter() - t0
    by_shape = {}
    for _, shape, sec in timings:
        by_shape.setdefault(shape, []).append(sec)
    by_shape = {k: float(np.mean(v)) for k, v in by_shape.items()}
    dims = mps.bond_dims
    shapes = [(dims[i], model.pbond_list[i], dims[i + 1]) for i in range(len(mps))]

    def est(shape):
        if shape in by_shape:
            return by_shape[shape]
        # scale the nearest measured class with the same physical dimension by the dominant D^3 d cost
        cands = [k for k in by_shape if k[1] == shape[1]] or list(by_shape)
        ref = max(cands, key=lambda k: k[0] * k[2])
        return by_shape[ref] * (shape[0] * shape[2] * (shape[0] + shape[2])) / (ref[0] * ref[2] * (ref[0] + ref[2]))

    step_time = 2.0 * sum(est(s) for s in shapes)
    measured = sum(sec for _, _, sec in timings)
    return dict(value=2 * len(mps) / step_time, unit="site-updates/s", cores=_CPU_THREADS, kind="port",
                sample=f"first {len(timings)} site updates of one evolve on the same MPS/MPO "
                       f"({measured:.1f} s incl. QR, env update, bond back-step; +{wall - measured:.1f} s environment "
                       f"build not counted), per-shape times extrapolated to all {2 * len(mps)} updates",
                host_cpus=os.cpu_count(), est_step_s=step_time)


def step_work(steps, to_right_first, bond_dims, pdims, wdims):
    """Algorithmic work of one TDVP-PS evolve from what it actually did (complex128: a complex
    multiply-add = 8 real flops, complex x real = 4): ``steps`` = Krylov dimensions of its 2 (2N - 1) local solves in the
    order the sweep performed them (site, bond, site, ..), bond / physical / MPO-bond dimensions of the chain.
    Returns (effective-Hamiltonian applications, flops): k F_hop1 per site solve + F_qr + F_env (+ F_absorb) per split
    site, k F_hop0 per bond solve."""
    n = len(pdims)

    def hop1(i):
        dl, dr, d, wl, wr = bond_dims[i], bond_dims[i + 1], pdims[i], wdims[i], wdims[i + 1]
        return 8.0 * dl * dl * wl * d * dr + 4.0 * dl * dr * wl * wr * d * d + 8.0 * dl * dr * dr * wr * d

    def hop0(b):            # bond b sits between sites b - 1 and b
        dl = dr = bond_dims[b]
        return 8.0 * wdims[b] * dl * dr * (dl + dr)

    def qr(i, right):
        m, k = (bond_dims[i] * pdims[i], bond_dims[i + 1]) if right else (bond_dims[i + 1] * pdims[i], bond_dims[i])
        k = min(m, k)
        return 4.0 * (4.0 * m * k * k - 4.0 * k ** 3 / 3.0)

    matvecs, flops, pos = 0, 0.0, 0
    for half in range(2):
        right = to_right_first if half == 0 else not to_right_first
        order = list(range(n)) if right else list(range(n - 1, -1, -1))
        for idx, i in enumerate(order):
            k = steps[pos]
            pos += 1
            matvecs += k
            flops += k * hop1(i)
            if idx == n - 1:
                continue                     # the last site of a half sweep is not split
            flops += qr(i, right) + hop1(i)  # block QR + one environment update
            b = i + 1 if right else i
            kb = steps[pos]
            pos += 1
            matvecs += kb
            flops += kb * hop0(b)
            nb = i + 1 if right else i - 1   # absorption of the bond factor into the neighbour
            flops += 8.0 * bond_dims[b] ** 2 * pdims[nb] * (bond_dims[nb + 1] if right else bond_dims[nb])
    assert pos == len(steps)
    return matvecs, flops


def _redone():
    """evolve steps (whole run, warm-up included) the optimistic block QR discarded and repeated on verified
    decompositions (mps/mps.py::_evolve_tdvp_ps)"""
    from renormalizer_amd.mps import mps as _m
    return _m._OPTIMISTIC_REDONE[0]


def spawn_ranks(n):
    """``python bench.py --gpus N`` without a launcher: start the N ranks here (one process per GPU, LOCAL_RANK = i, a
    free MASTER_PORT, a launch id that keys their rendezvous), relay rank 0's JSON line, and return non-zero if any
    rank does - the others are stopped then (they would wait in a collective for ever)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    launch_id = os.urandom(8).hex()
    procs = []
    for i in range(n):
        env = dict(os.environ, RANK=str(i), LOCAL_RANK=str(i), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MPSE_LAUNCH_ID=launch_id)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if i == 0 else subprocess.DEVNULL))
    print(f"bench.py: started {n} ranks (pids {[p.pid for p in procs]}, MASTER_PORT {port})", file=sys.stderr, flush=True)
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    code = 0
    live = set(range(n))
    while live:
        for i in list(live):
            rc = procs[i].poll()
            if rc is not None:
                live.discard(i)
                if rc != 0 and code == 0:
                    code = rc
                    print(f"bench.py: rank {i} exited with status {rc}; stopping the other ranks", file=sys.stderr, flush=True)
                    for j in live:
                        procs[j].terminate()
        time.sleep(0.05)
    reader.join(5.0)
    if code == 0 and out0 and out0[0]:
        sys.stdout.write(out0[0].decode())
        sys.stdout.flush()
    return code if code >= 0 else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nmol", type=int, default=25)
    ap.add_argument("--pdim", type=int, default=16)
    ap.add_argument("--bond-dim", type=int, default=256)
    ap.add_argument("--dt", type=float, default=10.0)
    ap.add_argument("--init", default="physical", choices=["physical", "random"])
    ap.add_argument("--scheme", default="tdvp_ps", choices=["tdvp_ps", "tdvp_ps2"],
                    help="tdvp_ps = the headline (one-site projector splitting, block QR per site); tdvp_ps2 = two-site "
                         "TDVP on the same chain: every bond is truncated through the blocked one-sided Jacobi SVD "
                         "(mps/mps.py:1406-1517) - the profile of that kernel at the headline bond size")
    ap.add_argument("--traj-per-gpu", type=int, default=1,
                    help="independent trajectories sharing each GPU (threads with their own stream); 1 = headline")
    ap.add_argument("--cpu-updates", type=int, default=9, help="site updates in the CPU baseline sample (0 = skip)")
    ap.add_argument("--state-file", default=None,
                    help="single trajectory only: load the prepared start state from this npz if it exists, else build "
                         "it and write it there (lets a profiler pass skip the preparation sweeps)")
    ap.add_argument("--dist-backend", default="rccl", choices=["rccl", "gloo"],
                    help="gloo + --share-gpu exercises the multi-rank flow on a single GPU (testing only)")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks use GPU 0 (testing only)")
    args = ap.parse_args()

    # --gpus N is the number of ranks.  Under a launcher (RANK / WORLD_SIZE set) the two must agree; without one this
    # process starts the N ranks itself
    if os.environ.get("WORLD_SIZE") is None and os.environ.get("RANK") is None:
        if args.gpus > 1:
            sys.exit(spawn_ranks(args.gpus))
    elif int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher's WORLD_SIZE is {os.environ.get('WORLD_SIZE', '1')}: "
              "refusing to report a figure for another number of ranks than asked for", file=sys.stderr, flush=True)
        sys.exit(2)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.share_gpu:
        local_rank = 0
    os.environ["RENO_GPU"] = str(local_rank)

    from renormalizer_amd.engine import Engine, get_engine, use_engine
    from renormalizer_amd.parallel import gather_observables, make_collective, max_over_ranks
    T = max(1, args.traj_per_gpu)
    engines = [get_engine()] + [Engine(local_rank) for _ in range(T - 1)]
    eng = engines[0]
    nsite = 2 * args.nmol
    # one communicator per process (ctypes RCCL on the engine's device and stream); serial for a single process
    if world > 1 and args.dist_backend == "gloo":      # testing only: the CPU stand-in lives with the tests
        sys.path.insert(0, os.path.join(REPO, "tests"))
        from gloo_collective import GlooCollective
        coll = GlooCollective()
    else:
        # A scaling run must not print a number obtained without RCCL: unless the file collective was asked for
        # (MPSE_COLLECTIVE=file), a communicator that cannot be created on every rank ends the job on every rank
        from renormalizer_amd.parallel import CollectiveUnavailable
        try:
            coll = make_collective(eng, strict=world > 1 and os.environ.get("MPSE_COLLECTIVE", "") != "file")
        except CollectiveUnavailable as exc:
            # no result line; leave at once (a thread stuck inside ncclCommInitRank would keep the process from ending)
            print(f"bench.py: CollectiveUnavailable: {exc}", file=sys.stderr, flush=True)
            os._exit(3)
    if world > 1:
        print(f"[rank {rank}] device {local_rank}: {eng.device_name}; {coll.kind} communicator of {coll.world} ranks",
              file=sys.stderr, flush=True)

    def barrier():
        for e in engines:
            e.sync()
        coll.barrier()
        for e in engines:
            e.sync()

    import threading
    sync = threading.Barrier(T + 1)
    results = [None] * T
    errors = []

    def trajectory(t):
        try:
            use_engine(engines[t])
            model, mpo, mps = build_workload(args.nmol, args.pdim, args.bond_dim, seed=1234 + rank * T + t,
                                             init=args.init, unit=rank * T + t,
                                             state_file=args.state_file if (T == 1 and world == 1) else None,
                                             scheme=args.scheme)
            for _ in range(args.warmup):
                mps = mps.evolve(mpo, args.dt)
            engines[t].prof_reset()
            engines[t].prof_enable(PROF_STRIDE)
            qr0 = engines[t].block_qr_stats()
            qp0 = engines[t].block_qr_pass_stats()
            redone0 = _redone()
            engines[t].sync()
            sync.wait()                    # all trajectories ready -> main thread takes t0
            rtx = _roctx()
            if rtx is not None:
                rtx.roctxProfilerResume(0)
            kry, work = [], []
            for _ in range(args.steps):
                first_right = bool(mps.to_right)
                mps = mps.evolve(mpo, args.dt)
                kry.append(mps.evolve_config.stat["mean"])
                if args.scheme == "tdvp_ps":
                    work.append((list(mps.evolve_config.stat["steps"]), first_right))
            engines[t].sync()
            if rtx is not None:
                rtx.roctxProfilerPause(0)
            redone_timed = _redone() - redone0
            qp1 = engines[t].block_qr_pass_stats()
            sync.wait()                    # all done -> main thread takes t1
            engines[t].prof_enable(False)
            results[t] = dict(model=model, mpo=mpo, mps=mps, kry=kry, work=work, prof=engines[t].prof_get(),
                              redone_timed=redone_timed, qr_pass=(qp1[0] - qp0[0], qp1[1] - qp0[1]),
                              qr=tuple(b - a for a, b in zip(qr0, engines[t].block_qr_stats())))
        except BaseException as exc:       # noqa: BLE001 - report and unblock the barrier
            errors.append(exc)
            sync.abort()

    threads = [threading.Thread(target=trajectory, args=(t,)) for t in range(T)]
    for th in threads:
        th.start()
    try:
        sync.wait()
        barrier()
        t0 = time.perf_counter()
        sync.wait()
        barrier()
        elapsed = time.perf_counter() - t0
    except threading.BrokenBarrierError:
        for th in threads:
            th.join()
        raise errors[0]
    for th in threads:
        th.join()
    if errors:
        raise errors[0]
    model, mpo, mps = results[0]["model"], results[0]["mpo"], results[0]["mps"]
    kry = [k for r in results for k in r["kry"]]
    work_tot = []
    for r in results:
        dims = [int(d) for d in r["mps"].bond_dims]
        pd, wd = list(r["model"].pbond_list), [int(w) for w in r["mpo"].bond_dims]
        work_tot += [step_work(st, fr, dims, pd, wd) for st, fr in r["work"]]
    prof = {}
    for r in results:
        for k, v in r["prof"].items():
            acc = prof.setdefault(k, {f: 0 * x for f, x in v.items()})
            for f in v:
                acc[f] += v[f]
    use_engine(None)

    elapsed = max_over_ranks(coll, elapsed)
    # the only collective of the whole job besides the barriers: all-gather of the per-trajectory observables (KBs)
    # (each row: the electronic populations of the rank's trajectory, then the device ordinal the rank ran on)
    # (each row: the populations, then what identifies the rank's device - ordinal, PCI bus id - and what the RCCL
    # communicator itself reports for the rank: ranks in the communicator, the rank's number there, its device)
    comm_info = coll.describe() if hasattr(coll, "describe") else (-1, -1, -1)
    row = np.concatenate([np.asarray(mps.e_occupations, dtype=np.float64),
                          [float(local_rank), float(_pci_bus_id(local_rank))], [float(x) for x in comm_info]])
    table = gather_observables(coll, row[None, :], [rank], world)
    assert table.shape[0] == world
    occ_table, rank_devices = table[:, :-5], [int(x) for x in table[:, -5]]
    rank_pci = ["%04x:%02x:%02x.%x" % ((int(x) >> 24) & 0xFFFF, (int(x) >> 16) & 0xFF, (int(x) >> 8) & 0xFF, int(x) & 0xFF)
                if x >= 0 else None for x in table[:, -4]]
    comm_count, comm_rank, comm_dev = ([int(x) for x in table[:, k]] for k in (-3, -2, -1))
    distinct = len({tuple(np.round(r, 10)) for r in occ_table})
    if world > 1 and rank == 0:
        print(f"[rank 0] gathered populations of {world} trajectories, {distinct} distinct", file=sys.stderr, flush=True)

    if rank == 0:
        # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE;
        # tools/pmc_traffic.py, gfx950 correction 2 x fetch + write) committed under profiles/: counters cannot be
        # collected from inside this process
        traffic, traffic_src = None, None
        try:
            import glob
            import hashlib
            pmc_file = sorted(glob.glob(os.path.join(REPO, "profiles", "r[0-9][0-9]_pmc_traffic.json")))[-1]   # latest round
            with open(pmc_file, "rb") as fh:
                raw = fh.read()
            pmc = json.loads(raw)
            kern = pmc["kernels"]
            # every instantiation of the complex x complex contraction kernel (tile / wave-layout arguments differ),
            # weighted by its launches
            hits = [v for k, v in kern.items() if k.startswith("void k_gemm<true, true, true")]
            traffic_ref = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in hits) / sum(v["launches"] for v in hits)
            fresh = pmc.get("kernel_source_sha") is not None and pmc.get("kernel_source_sha") == _kernel_source_sha()
            traffic = traffic_ref if fresh else None       # a figure measured on other kernel sources is not this kernel's
            traffic_src = (f"profiles/{os.path.basename(pmc_file)} sha256:{hashlib.sha256(raw).hexdigest()[:16]} "
                           "(separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command with --init "
                           "physical; a committed measurement, not one of this run: counters cannot be read from "
                           "inside the process); kernel sources "
                           + ("unchanged since that pass" if fresh else
                              f"CHANGED since that pass: traffic withheld, the stale figure was {traffic_ref:.4g} bytes/launch"))
        except (OSError, KeyError, ValueError, StopIteration, IndexError, ZeroDivisionError):
            pass
        # the same for the kernel's duration by the profiler's clock: profiles/rNN_kernel_time.json (tools/rocpd_kernel_time.py
        # on a rocprofv3 kernel trace of this command's timed region)
        ktime, ktime_src = None, None
        try:
            kt_file = sorted(glob.glob(os.path.join(REPO, "profiles", "r[0-9][0-9]_kernel_time.json")))[-1]
            with open(kt_file, "rb") as fh:
                raw = fh.read()
            kt = json.loads(raw)
            fresh_kt = kt.get("kernel_source_sha") is not None and kt.get("kernel_source_sha") == _kernel_source_sha()
            ktime = kt if fresh_kt else None
            ktime_src = (f"profiles/{os.path.basename(kt_file)} sha256:{hashlib.sha256(raw).hexdigest()[:16]} (rocprofv3 "
                         "--kernel-trace of this command's timed region, a committed measurement); kernel sources "
                         + ("unchanged since that trace" if fresh_kt else
                            f"CHANGED since that trace: figure withheld, the stale average was {kt.get('avg_us', 0):.1f} us"))
        except (OSError, KeyError, ValueError, IndexError, NameError):
            pass
        zz = prof["c128xc128"]
        sec = zz["ms"] * 1e-3
        dense = zz["flops"] / sec / 1e12 if sec > 0 else 0.0
        issued = zz["issued_flops"] / sec / 1e12 if sec > 0 else 0.0
        total_ms = sum(v["ms"] for v in prof.values())
        classes = []
        vv, qq = prof["lanczos_vec"], prof["block_qr"]
        if vv["ms"] > 0:
            gbs = vv["bytes"] / (vv["ms"] * 1e-3) / 1e9
            classes.append({"kernel": "Lanczos vector kernels (dot / three-term update + norm / normalise)", "bound": "hbm",
                            "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                            "timed_launches": vv["launches"], "avg_launch_ms": vv["ms"] / max(1, vv["launches"]),
                            "alg_bytes_per_launch": vv["bytes"] / max(1, vv["launches"])})
        if qq["ms"] > 0:
            tf = qq["flops"] / (qq["ms"] * 1e-3) / 1e12
            classes.append({"kernel": "block QR / RQ (whole mpse_block_qr calls: Cholesky-QR on MFMA for tall blocks, Householder "
                                      "panels otherwise)", "bound": "latency (chain of dependent launches, one workgroup per block "
                                                                    "in the Cholesky factor)",
                            "achieved": tf, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_MFMA_PEAK_TFLOPS,
                            "timed_calls": qq["launches"], "avg_call_ms": qq["ms"] / max(1, qq["launches"]),
                            "alg_flops_per_call": qq["flops"] / max(1, qq["launches"])})
        ff = prof.get("heff_fused", {"ms": 0})
        if ff["ms"] > 0:
            tf = ff["flops"] / (ff["ms"] * 1e-3) / 1e12
            busy_d, busy_g, busy_src = _committed_mfma_busy("k_heff0_fused", "mpse_heff0.hip")
            classes.append({"kernel": "k_heff0_fused (bond and two-level-site matvecs in one MFMA launch each)",
                            "bound": "latency (one 4-wave workgroup per compute unit, dependent loads)",
                            "achieved": tf, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                            # NOT a utilisation: dense-formula flops over time (the kernel skips the zero blocks)
                            "dense_equiv_frac": tf / FP64_MFMA_PEAK_TFLOPS,
                            "mfma_busy": busy_d, "mfma_busy_gui_active": busy_g, "mfma_busy_source": busy_src,
                            "achieved_note": "algorithmic (dense-formula) flops of the matvec over its launch time; the kernel "
                                             "skips the quantum-number zero blocks, so the MFMA work issued is smaller",
                            "timed_launches": ff["launches"], "avg_launch_ms": ff["ms"] / max(1, ff["launches"]),
                            "alg_flops_per_launch": ff["flops"] / max(1, ff["launches"])})
        ss = prof.get("block_svd", {"ms": 0})
        if ss["ms"] > 0:
            gbs = ss["bytes"] / (ss["ms"] * 1e-3) / 1e9
            classes.append({"kernel": "block SVD (batched one-sided Jacobi, whole mpse_block_svd calls: gather, sweeps, "
                                      "Householder completion, scatter)", "bound": "hbm (L2-resident column pairs)",
                            "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                            "achieved_tflops": ss["flops"] / (ss["ms"] * 1e-3) / 1e12,
                            "timed_calls": ss["launches"], "avg_call_ms": ss["ms"] / max(1, ss["launches"]),
                            "sweeps_per_call": ss["sweeps"] / max(1, ss["launches"]),
                            "note": "bytes / flops count every column pair of every sweep (upper bound: converged pairs are not rotated)"})
        for nm, what in (("c128xf64", "small products with a real second operand"), ("f64xc128", "MPO step")):
            v = prof[nm]
            if v["ms"] > 0:
                classes.append({"kernel": f"k_gemm<{nm}> ({what})", "bound": "hbm",
                                "achieved": v["bytes"] / (v["ms"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": v["bytes"] / (v["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, "timed_launches": v["launches"],
                                "avg_launch_ms": v["ms"] / max(1, v["launches"])})
        out = {
            "metric": ("TDVP-PS sweep site-updates/sec at (Nsite=%d, Dbond=%d, dphys=2/%d)" if args.scheme == "tdvp_ps" else
                       "TDVP-PS2 (two-site) sweep bond-updates/sec at (Nsite=%d, Dbond=%d, dphys=2/%d)")
                      % (nsite, args.bond_dim, args.pdim),
            # tdvp_ps: 2 N one-site forward solves per evolve; tdvp_ps2: 2 (N - 1) two-site solves, each followed by the block SVD
            "value": world * T * args.steps * (2 * nsite if args.scheme == "tdvp_ps" else 2 * (nsite - 1)) / elapsed,
            "unit": "site-updates/s" if args.scheme == "tdvp_ps" else "bond-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,   # wall time per evolve of each trajectory
            # state-independent rates: the effective-Hamiltonian applications the solves of the timed steps needed
            # (their Krylov dimensions) and the algorithmic flops of SURVEY.md 8(d) at the printed bond dimensions
            "heff_matvecs_per_s": (world * sum(w[0] for w in work_tot) / elapsed) if work_tot else None,
            "alg_tflops_per_step": (sum(w[1] for w in work_tot) / len(work_tot) / 1e12) if work_tot else None,
            "alg_tflops_per_s": (world * sum(w[1] for w in work_tot) / elapsed / 1e12) if work_tot else None,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "c128",
            "data": ("synthetic (std.yaml Holstein parameters; electron created on the centre molecule of the phonon "
                     "vacuum, bonds expanded to Dbond with expand_bond_dimension, as transport/dynamics.py:173-199; "
                     "trajectories beyond the first carry static diagonal disorder, sigma = 50 cm^-1, own seed)"
                     if args.init == "physical" else
                     "synthetic (random quantum-number-conserving MPS, 1 exciton; std.yaml Holstein parameters)"),
            "config": {"workload": "configs[2]: %d-site Holstein chain TDVP-PS, %d independent trajector%s per GPU"
                                   % (nsite, T, "y" if T == 1 else "ies"),
                       "nsite": nsite, "bond_dim": args.bond_dim, "dphys": [2, args.pdim], "mpo_bond": max(mpo.bond_dims),
                       "dt": args.dt, "init": args.init, "mean_krylov_dim": float(np.mean(kry)),
                       "device": eng.device_name, "collective": coll.kind, "ranks": coll.world,
                       "rank_devices": rank_devices, "rank_pci_bus_ids": rank_pci,
                       "rccl_comm_count": comm_count, "rccl_comm_user_rank": comm_rank, "rccl_comm_device": comm_dev,
                       "distinct_trajectories": distinct,
                       # block QR / RQ decompositions of the timed steps: how many went through the Cholesky-QR kernels
                       # (tall blocks) and how many of those a device flag sent back to the Householder kernels
                       "block_qr": dict(zip(("calls", "cholesky_qr", "redone_by_householder"),
                                            [int(sum(r["qr"][k] for r in results)) for k in range(3)]),
                                        # evolves the optimistic block QR discarded and ran again verified: inside
                                        # the timed region (their time IS in `value`; with several trajectories per
                                        # GPU the counter is shared, so this is an upper bound per trajectory) and
                                        # over the whole run, warm-up included
                                        # quantum-number blocks the Cholesky-QR kernels factorised in the timed steps and
                                        # how many of them the device ended after two passes (adaptive pass count)
                                        cholesky_qr_blocks=int(sum(r["qr_pass"][0] for r in results)),
                                        cholesky_qr_blocks_two_passes=int(sum(r["qr_pass"][1] for r in results)),
                                        steps_repeated_in_timed_region=int(max(r["redone_timed"] for r in results)),
                                        steps_repeated_after_breakdown=int(_redone())),
                       "bond_dims": [int(d) for d in mps.bond_dims],
                       "environments": ("rebuilt at every step (MPSE_ENV_CARRY=0)" if os.environ.get("MPSE_ENV_CARRY") == "0"
                                        else "those ahead of the first half sweep are taken over from the previous step "
                                             "(identical tensors; every step performs all 2 N site updates)")},
            "roofline": {"bound": "mfma", "kernel": "k_gemm<c128,c128> (FP64 MFMA strided contraction, 3M complex products)",
                         "achieved": issued, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": issued / FP64_MFMA_PEAK_TFLOPS,
                         "achieved_note": "issued MFMA work: K tiles visited (device counters) x 65536 MACs x 6 real flops",
                         # the same issued work over the kernel's own duration in a committed rocprofv3 trace (the HIP-event
                         # bracket above also spans the split-K reduction launches and the events' own gaps)
                         "frac_kernel_time": (zz["issued_flops"] / max(1, zz["launches"]) / (ktime["avg_us"] * 1e-6) / 1e12
                                              / FP64_MFMA_PEAK_TFLOPS) if ktime else None,
                         "kernel_time_avg_us": ktime["avg_us"] if ktime else None,
                         "kernel_time_source": ktime_src,
                         # algorithmic 8 M N K of SURVEY.md 8(d) over the same time: NOT a roofline fraction (the kernel
                         # skips structurally empty K tiles and uses three real products per complex one)
                         "dense_equiv_tflops": dense,
                         "visited_ktile_share": (zz["ktiles"] * 65536.0 * 8.0 / zz["flops"]) if zz["flops"] else None,
                         "traffic": traffic, "traffic_unit": "bytes/launch",
                         "traffic_source": traffic_src,
                         "compulsory_bytes_per_launch": zz["bytes"] / max(1, zz["launches"]),
                         "timed_launches": zz["launches"], "sampling_stride": PROF_STRIDE, "avg_launch_ms": zz["ms"] / max(1, zz["launches"]),
                         "alg_flops_per_launch": zz["flops"] / max(1, zz["launches"]),
                         "issued_flops_per_launch": zz["issued_flops"] / max(1, zz["launches"]),
                         "sampled_time_share_of_wall_est": PROF_STRIDE * 1e-3 * total_ms / elapsed / T,
                         "classes": classes},
        }
        if world == 1 and args.cpu_updates > 0:
            out["cpu_baseline"] = cpu_baseline(model, mpo, mps, args.dt, args.cpu_updates)
        print(json.dumps(out))
    coll.close()


if __name__ == "__main__":
    main()
