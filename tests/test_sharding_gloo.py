"""Trajectory sharding across ranks: one independent trajectory per rank, no data-path
collective, one all_gather of the observables at the end.  Exercised here with world_size 2 on the gloo
backend (CPU) through the same helper bench.py uses on RCCL."""
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
        import torch
        import torch.distributed as dist
        from renormalizer_amd.parallel import trajectory_seed, gather_observables, max_over_ranks
        dist.init_process_group(backend="gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        # unit u -> rank u mod world; every rank works on its own trajectories only
        units = [u for u in range(5) if u % world == rank]
        obs = np.array([[trajectory_seed(1234, u), u * 0.5] for u in units], dtype=np.float64)
        allobs = gather_observables(obs, units, 5, device="cpu")
        t = max_over_ranks(float(rank + 1), device="cpu")
        if rank == 0:
            assert allobs.shape == (5, 2), allobs.shape
            assert np.array_equal(allobs[:, 1], np.arange(5) * 0.5)
            assert len(set(allobs[:, 0].tolist())) == 5           # distinct seeds per trajectory
            assert t == float(world)
            print("OK")
        dist.destroy_process_group()
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", env["MASTER_PORT"], str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "OK" in out.stdout
