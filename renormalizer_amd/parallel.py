"""Multi-GPU execution model: independent trajectories, one process per GPU.

A TDVP/DMRG sweep is strictly sequential in the site index, so nothing is sharded inside a sweep.  Independent units
(trajectories, disorder realisations, parameter points) are dealt round-robin to ranks; there is no collective on the
data path.  At the end every rank contributes its observable rows to one all-gather - kilobytes, latency bound on
xGMI.

The product path binds ``librccl.so`` directly through ctypes (``RcclCollective``: ncclGetUniqueId / ncclCommInitRank /
ncclAllGather / ncclAllReduce on the engine's HIP stream, buffers from the engine's allocator; the 128-byte unique id
travels through a file named after the launcher's process id).  No PyTorch.  ``GlooCollective`` is the CPU stand-in of
the same three operations for the world_size-2 tests (torch.distributed's gloo backend, imported only there)."""
import ctypes as C
import os
import time

import numpy as np


def trajectory_seed(base_seed: int, unit: int) -> int:
    """Deterministic, distinct RNG seed per independent unit."""
    return int(np.random.SeedSequence([int(base_seed), int(unit)]).generate_state(1)[0])


def units_of_rank(n_units: int, rank: int, world: int):
    return [u for u in range(n_units) if u % world == rank]


class SerialCollective:
    """One process: every collective is the identity."""
    rank, world, kind = 0, 1, "serial"

    def barrier(self):
        pass

    def allreduce_max(self, value: float) -> float:
        return float(value)

    def allgather(self, row: np.ndarray) -> np.ndarray:
        return np.asarray(row, dtype=np.float64).reshape(1, -1)

    def close(self):
        pass


class _ncclUniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


_NCCL_FLOAT64, _NCCL_SUM, _NCCL_MAX = 8, 0, 2


def _rendezvous_path():
    """All local ranks of one launch share their parent (the torch.distributed.run / torchrun agent or the shell that
    started them): its pid plus the advertised port name the id file."""
    tag = os.environ.get("MPSE_RENDEZVOUS_TAG") or f"{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}"
    d = os.environ.get("MPSE_RENDEZVOUS_DIR", "/tmp")
    return os.path.join(d, f"mpse_rccl_{tag}.id")


class RcclCollective:
    """RCCL over xGMI through ctypes; one communicator per process, collectives on the engine's stream."""
    kind = "rccl"

    def __init__(self, eng, rank: int, world: int, timeout_s: float = 300.0):
        self.eng, self.rank, self.world = eng, int(rank), int(world)
        self.lib = C.CDLL(os.environ.get("MPSE_RCCL_LIB", "librccl.so"))
        L = self.lib
        L.ncclGetErrorString.restype = C.c_char_p
        L.ncclGetUniqueId.argtypes = [C.POINTER(_ncclUniqueId)]
        L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _ncclUniqueId, C.c_int]
        L.ncclCommDestroy.argtypes = [C.c_void_p]
        L.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        L.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        uid = _ncclUniqueId()
        path = _rendezvous_path()
        self._path = path if self.rank == 0 else None
        if self.rank == 0:
            self._ok(L.ncclGetUniqueId(C.byref(uid)))
            tmp = f"{path}.{os.getpid()}.tmp"
            with open(tmp, "wb") as fh:
                fh.write(bytes(uid.internal))
            os.replace(tmp, path)                      # atomic: readers never see a partial id
        else:
            t0 = time.time()
            while True:
                try:
                    with open(path, "rb") as fh:
                        raw = fh.read()
                    if len(raw) == 128:
                        break
                except OSError:
                    pass
                if time.time() - t0 > timeout_s:
                    raise TimeoutError(f"rank {self.rank}: no RCCL unique id at {path} after {timeout_s} s")
                time.sleep(0.01)
            C.memmove(C.byref(uid), raw, 128)
        eng.sync()                                     # binds this thread to the engine's device (the comm's device)
        self.comm = C.c_void_p()
        self._ok(L.ncclCommInitRank(C.byref(self.comm), self.world, uid, self.rank))

    def _ok(self, st):
        if st != 0:
            msg = self.lib.ncclGetErrorString(st)
            raise RuntimeError(f"RCCL error {st}: {msg.decode() if msg else ''}")

    def barrier(self):
        self.allreduce_max(0.0)

    def allreduce_max(self, value: float) -> float:
        buf = self.eng.asdevice(np.array([value], dtype=np.float64))
        self._ok(self.lib.ncclAllReduce(buf.ptr, buf.ptr, 1, _NCCL_FLOAT64, _NCCL_MAX, self.comm, self.eng.stream))
        return float(buf.to_host()[0])

    def allgather(self, row: np.ndarray) -> np.ndarray:
        row = np.ascontiguousarray(row, dtype=np.float64).ravel()
        send = self.eng.asdevice(row)
        recv = self.eng.empty((self.world, row.size), np.float64)
        self._ok(self.lib.ncclAllGather(send.ptr, recv.ptr, row.size, _NCCL_FLOAT64, self.comm, self.eng.stream))
        return recv.to_host()

    def close(self):
        if getattr(self, "comm", None):
            self.barrier()
            self.eng.sync()
            self.lib.ncclCommDestroy(self.comm)
            self.comm = None
            if self._path:
                try:
                    os.remove(self._path)
                except OSError:
                    pass


class GlooCollective:
    """CPU stand-in (tests): the same three operations on torch.distributed's gloo backend."""
    kind = "gloo"

    def __init__(self):
        import torch.distributed as dist
        if not dist.is_initialized():
            dist.init_process_group(backend="gloo")
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def barrier(self):
        self.dist.barrier()

    def allreduce_max(self, value: float) -> float:
        import torch
        t = torch.tensor([value], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def allgather(self, row: np.ndarray) -> np.ndarray:
        import torch
        t = torch.as_tensor(np.ascontiguousarray(row, dtype=np.float64).ravel())
        parts = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(parts, t)
        return np.stack([p.numpy() for p in parts])

    def close(self):
        self.dist.destroy_process_group()


def make_collective(eng=None, backend=None):
    """Collective of this process from the launcher's environment (RANK / WORLD_SIZE as set by
    torch.distributed.run): serial for one process, RCCL (ctypes) otherwise; ``backend="gloo"`` selects the CPU
    stand-in."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if backend == "gloo":
        return GlooCollective()
    if world == 1 and backend != "rccl":
        return SerialCollective()
    if eng is None:
        from .engine import get_engine
        eng = get_engine()
    return RcclCollective(eng, rank, world)


def max_over_ranks(coll, value: float) -> float:
    return coll.allreduce_max(value)


def gather_observables(coll, local_rows: np.ndarray, local_units, n_units: int) -> np.ndarray:
    """One all-gather of per-unit observable rows; returns the (n_units, nobs) table on every rank.  Every rank sends
    a fixed-size packet [(unit id, row)...] padded with unit id -1."""
    local_rows = np.atleast_2d(np.asarray(local_rows, dtype=np.float64))
    nobs = int(coll.allreduce_max(float(local_rows.shape[1] if local_rows.size else 0)))
    per_rank = (n_units + coll.world - 1) // coll.world
    packet = np.full((per_rank, nobs + 1), -1.0)
    for k, (u, row) in enumerate(zip(local_units, local_rows)):
        packet[k, 0] = float(u)
        packet[k, 1:] = row
    table = coll.allgather(packet.ravel()).reshape(coll.world * per_rank, nobs + 1)
    out = np.zeros((n_units, nobs))
    for row in table:
        if row[0] >= 0:
            out[int(row[0])] = row[1:]
    return out
