"""Trajectory sharding across ranks: one independent trajectory per rank, no data-path
collective, one all_gather of the observables at the end.  Exercised here with world_size 2 on the gloo
backend (CPU): ``GlooCollective`` is the stand-in of the ctypes ``RcclCollective`` that bench.py and
examples/fmo.py use on the GPUs (same three operations: barrier, max, all-gather)."""
import os
import socket
import subprocess
import sys
import textwrap

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {REPO!r})
        import numpy as np
        from renormalizer_amd.parallel import (trajectory_seed, gather_observables, max_over_ranks, make_collective,
                                               units_of_rank)
        coll = make_collective(backend="gloo")          # CPU stand-in of the RCCL collective (same interface)
        rank, world = coll.rank, coll.world
        # unit u -> rank u mod world; every rank works on its own trajectories only
        units = units_of_rank(5, rank, world)
        obs = np.array([[trajectory_seed(1234, u), u * 0.5] for u in units], dtype=np.float64)
        allobs = gather_observables(coll, obs, units, 5)
        coll.barrier()
        t = max_over_ranks(coll, float(rank + 1))
        if rank == 0:
            assert allobs.shape == (5, 2), allobs.shape
            assert np.array_equal(allobs[:, 1], np.arange(5) * 0.5)
            assert len(set(allobs[:, 0].tolist())) == 5           # distinct seeds per trajectory
            assert t == float(world)
            print("OK")
        coll.close()
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", env["MASTER_PORT"], str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "OK" in out.stdout


def test_rendezvous_file_and_serial_collective(tmp_path, monkeypatch):
    """The unique-id rendezvous is keyed by the launcher's pid and port; one process needs no communicator."""
    import numpy as np
    from renormalizer_amd import parallel
    monkeypatch.setenv("MPSE_RENDEZVOUS_DIR", str(tmp_path))
    monkeypatch.setenv("MASTER_PORT", "29517")
    p = parallel._rendezvous_path()
    assert p.startswith(str(tmp_path)) and str(os.getppid()) in p and p.endswith("_29517.id")
    monkeypatch.setenv("WORLD_SIZE", "1")
    coll = parallel.make_collective()
    assert coll.kind == "serial" and coll.allreduce_max(3.5) == 3.5
    tab = parallel.gather_observables(coll, np.array([[1.0, 2.0], [3.0, 4.0]]), [0, 1], 2)
    assert tab.tolist() == [[1.0, 2.0], [3.0, 4.0]]
