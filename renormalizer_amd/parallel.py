"""Multi-GPU execution model: independent trajectories, one process per GPU.

A TDVP/DMRG sweep is strictly sequential in the site index, so nothing is sharded inside a sweep.  Independent units
(trajectories, disorder realisations, parameter points) are dealt round-robin to ranks; there is no collective on the
data path.  At the end every rank contributes its observable rows to one all-gather - kilobytes, latency bound on
xGMI.

The product path binds ``librccl.so`` directly through ctypes (``RcclCollective``: ncclGetUniqueId / ncclCommInitRank /
ncclAllGather / ncclAllReduce on the engine's HIP stream, buffers from the engine's allocator).  The 128-byte unique id
and a per-launch nonce travel over a plain TCP exchange with rank 0 at MASTER_ADDR (``SocketRendezvous``: rank 0 listens
on the first free port of MASTER_PORT .. MASTER_PORT + 16 - under torch.distributed.run the agent's own store holds
MASTER_PORT itself - and the other ranks find it by its handshake), so ranks need not share a parent process or a
start time; ``MPSE_RENDEZVOUS_TAG`` selects the older exchange through a private file instead.  No PyTorch: the CPU
stand-in of the same three operations that the world_size-2 tests use lives in ``tests/gloo_collective.py``."""
import ctypes as C
import os
import sys
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
    # (unsigned bytes: a c_char array would read back as a Python bytes object cut at the first zero byte)
    _fields_ = [("internal", C.c_ubyte * 128)]


_NCCL_FLOAT64, _NCCL_SUM, _NCCL_MAX = 8, 0, 2


def _parent_instance():
    """pid and start time (clock ticks since boot, /proc/<pid>/stat field 22) of the parent: all local ranks of one
    launch share their parent (the torch.distributed.run agent or the shell that started them), and the start time
    tells a re-used pid from the same process."""
    ppid = os.getppid()
    try:
        with open(f"/proc/{ppid}/stat", "rb") as fh:
            start = fh.read().rsplit(b")", 1)[1].split()[19].decode()
    except (OSError, IndexError):
        start = "0"
    return ppid, start


_RDZV_MAGIC = b"MPSE-RDZV-1\0"      # 12 bytes
_RDZV_PORTS = 17                      # MASTER_PORT .. MASTER_PORT + 16


def _launch_key() -> bytes:
    """What all ranks of one launch know without talking to each other: the number of ranks, the advertised port, the
    launcher's run id when it has a real one and - when bench.py started the ranks itself - its launch id.  It keeps a
    rank from joining the rendezvous of ANOTHER launch that happens to listen in the same port range."""
    import hashlib
    run_id = os.environ.get("TORCHELASTIC_RUN_ID", "")
    txt = "|".join([os.environ.get("WORLD_SIZE", "1"), os.environ.get("MASTER_PORT", "0"),
                    run_id if run_id not in ("", "none") else "", os.environ.get("MPSE_LAUNCH_ID", "")])
    return hashlib.sha256(txt.encode()).digest()[:16]


def _recv_exact(conn, n):
    buf = b""
    while len(buf) < n:
        chunk = conn.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("peer closed the connection")
        buf += chunk
    return buf


class SocketRendezvous:
    """Small key -> bytes exchange of one launch over MASTER_ADDR.  Meant for the ranks of ONE node on a trusted
    network (what this engine shards over: the GPUs of a node): the launch key is derivable from the environment, the
    exchange is neither authenticated beyond it nor encrypted, and it carries nothing but the RCCL unique id and a nonce.
    Rank 0 listens on the first free port of MASTER_PORT .. MASTER_PORT + 16 - skipping MASTER_PORT itself when a
    launcher's own store is known to sit there (torch.distributed.run's agent: the other ranks skip it too, so a rank 0
    that found it free would listen where nobody looks) - on MASTER_ADDR when that is a local address and on every
    interface otherwise (global rank 0 need not run on the MASTER_ADDR host), and draws the launch nonce; rank r > 0
    walks the same ports and takes the first listener that answers with the right magic AND the right launch key.
    Every connection is served by a short-lived thread with a 2 s budget: a silent peer cannot stall the others.
    Request: magic | launch key (16) | rank (int32) | key (16).
    Reply: magic | nonce (8) | status (int32: 0 data follows, 1 not published yet, 2 another launch) | length | data.
    Independent of process ancestry and of the file system; a crashed earlier launch leaves nothing behind."""

    def __init__(self, rank: int, world: int, timeout_s: float = 120.0):
        import socket
        import threading
        self.rank, self.world, self.timeout_s = int(rank), int(world), float(timeout_s)
        self.addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        self.base_port = int(os.environ.get("MASTER_PORT", "29500"))
        self.key = _launch_key()
        self.port = None
        self._store, self._lock, self._stop, self._thread, self._srv = {}, threading.Lock(), False, None, None
        if self.rank == 0:
            self.nonce = os.urandom(8)
            last = None
            agent_store = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "").lower() in ("1", "true")
            for off in range(_RDZV_PORTS):
                if off == 0 and agent_store:
                    continue                          # the launcher's store: the other ranks never ask there
                for host in (self.addr, ""):          # MASTER_ADDR if it is one of this host's addresses, else any
                    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    try:
                        srv.bind((host, self.base_port + off))
                        srv.listen(64)
                        self._srv, self.port = srv, self.base_port + off
                        break
                    except OSError as exc:
                        last = exc
                        srv.close()
                        import errno
                        if exc.errno != errno.EADDRNOTAVAIL:
                            break                     # port taken: next port; address not local: same port, any interface
                if self._srv is not None:
                    break
            if self._srv is None:
                raise OSError(f"rank 0: no free port in {self.base_port} .. {self.base_port + _RDZV_PORTS - 1} "
                              f"on {self.addr} for the rendezvous ({last})")
            self._srv.settimeout(0.2)
            self._thread = threading.Thread(target=self._serve, daemon=True)
            self._thread.start()
        else:
            self.nonce = None
            self.nonce = self.fetch("nonce")[:8]

    # ---- rank 0
    def _serve(self):
        import socket
        import threading
        while not self._stop:
            try:
                conn, _ = self._srv.accept()
            except socket.timeout:
                continue
            except OSError:
                return
            threading.Thread(target=self._answer, args=(conn,), daemon=True).start()

    def _answer(self, conn):
        import struct
        try:
            conn.settimeout(2.0)
            req = _recv_exact(conn, 12 + 16 + 4 + 16)
            if req[:12] != _RDZV_MAGIC:
                return
            key = req[32:48].rstrip(b"\0").decode(errors="replace")
            if req[12:28] != self.key:
                status, data = 2, b""
            else:
                with self._lock:
                    data = self.nonce if key == "nonce" else self._store.get(key)
                status, data = (0, data) if data is not None else (1, b"")
            conn.sendall(_RDZV_MAGIC + self.nonce + struct.pack("<ii", status, len(data)) + data)
        except (OSError, ConnectionError):
            pass
        finally:
            conn.close()

    def publish(self, key: str, data: bytes):
        assert self.rank == 0 and len(key.encode()) <= 16
        with self._lock:
            self._store[key] = bytes(data)

    # ---- ranks > 0
    def _ask(self, port, key):
        import socket
        import struct
        with socket.create_connection((self.addr, port), timeout=1.0) as conn:
            conn.settimeout(5.0)
            conn.sendall(_RDZV_MAGIC + self.key + struct.pack("<i", self.rank) + key.encode().ljust(16, b"\0"))
            head = _recv_exact(conn, 12 + 8 + 8)
            if head[:12] != _RDZV_MAGIC:
                return 2, b"", b""
            status, n = struct.unpack("<ii", head[20:28])
            return status, head[12:20], (_recv_exact(conn, n) if n > 0 else b"")

    def fetch(self, key: str, timeout_s: float = None) -> bytes:
        """The bytes rank 0 published under ``key`` (waits until they are there)."""
        timeout_s = self.timeout_s if timeout_s is None else timeout_s
        t0 = time.time()
        agent_store = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "").lower() in ("1", "true")
        while True:
            ports = [self.port] if self.port else [self.base_port + off for off in range(_RDZV_PORTS)
                                                   if not (off == 0 and agent_store)]
            for port in ports:
                try:
                    status, nonce, data = self._ask(port, key)
                except (OSError, ConnectionError):
                    continue
                if status == 2:
                    continue
                if self.nonce is not None and nonce != self.nonce:
                    raise RuntimeError(f"rank {self.rank}: the rendezvous at {self.addr}:{port} changed its launch nonce")
                self.port = port
                if status == 0:
                    return nonce + data if key == "nonce" else data
            if time.time() - t0 > timeout_s:
                raise TimeoutError(f"rank {self.rank}: no answer for '{key}' from rank 0's rendezvous at {self.addr}:"
                                   f"{self.base_port}..{self.base_port + _RDZV_PORTS - 1} within {timeout_s} s "
                                   "(is rank 0 of this launch running?)")
            time.sleep(0.02)

    def close(self):
        self._stop = True
        if self._srv is not None:
            try:
                self._srv.close()
            except OSError:
                pass


_RDZV = {}


def launch_rendezvous(rank: int, world: int, timeout_s: float = 120.0):
    """The socket rendezvous of this process's launch (created once), or None when there is only one rank or when
    ``MPSE_RENDEZVOUS_TAG`` asks for the exchange through a private file."""
    if world <= 1 or os.environ.get("MPSE_RENDEZVOUS_TAG"):
        return None
    if "r" not in _RDZV:
        _RDZV["r"] = SocketRendezvous(rank, world, timeout_s)
    return _RDZV["r"]


def _rendezvous_dir():
    """Private directory (mode 0700, owned by this user) for the id files: nobody else can plant or link one."""
    d = os.environ.get("MPSE_RENDEZVOUS_DIR") or os.path.join(
        os.environ.get("XDG_RUNTIME_DIR") or "/tmp", f"mpse_rccl_{os.getuid()}")
    os.makedirs(d, mode=0o700, exist_ok=True)
    st = os.lstat(d)
    import stat as _stat
    if not _stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid():
        raise RuntimeError(f"rendezvous directory {d} is not a directory owned by this user")
    return d


def _rendezvous_path():
    """One file per launch: MPSE_RENDEZVOUS_TAG / TORCHELASTIC_RUN_ID when the launcher gives a real one, else the
    parent's pid and start time, plus the advertised port.  Single node only (shared file system path)."""
    tag = os.environ.get("MPSE_RENDEZVOUS_TAG")
    if not tag and "r" in _RDZV:
        tag = "n" + _RDZV["r"].nonce.hex()          # socket rendezvous: a name no other launch can have
    if not tag:
        run_id = os.environ.get("TORCHELASTIC_RUN_ID", "")
        ppid, start = _parent_instance()
        tag = f"{run_id if run_id not in ('', 'none') else 'p'}_{ppid}_{start}_{os.environ.get('MASTER_PORT', '0')}"
    return os.path.join(_rendezvous_dir(), f"mpse_rccl_{tag}.id")


def _process_start_time():
    """Wall-clock start of this process (seconds since the epoch)."""
    try:
        with open("/proc/self/stat", "rb") as fh:
            ticks = float(fh.read().rsplit(b")", 1)[1].split()[19])
        with open("/proc/uptime") as fh:
            up = float(fh.read().split()[0])
        return time.time() - up + ticks / os.sysconf("SC_CLK_TCK")
    except (OSError, IndexError, ValueError):
        return time.time()


_STALE_SLACK_S = 120.0     # an id file older than this before the reader started belongs to an earlier launch


def publish_id(path: str, raw: bytes):
    """Rank 0: replace whatever an earlier launch with the same tag left behind by this launch's id (created
    exclusively with mode 0600, moved into place atomically: readers never see a partial id)."""
    try:
        os.unlink(path)
    except FileNotFoundError:
        pass
    tmp = f"{path}.{os.getpid()}.tmp"
    fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
    with os.fdopen(fd, "wb") as fh:
        fh.write(raw)
    os.replace(tmp, path)


def await_id(path: str, timeout_s: float, rank: int = -1) -> bytes:
    """Ranks > 0: wait for a FRESH id file of this user.  A file written long before this process existed is a
    crashed launch's (rank 0 of this launch replaces it): it is ignored, never trusted."""
    t0 = time.time()
    born = _process_start_time()
    while True:
        try:
            st = os.stat(path)
            if st.st_uid == os.getuid() and st.st_mtime >= born - _STALE_SLACK_S:
                with open(path, "rb") as fh:
                    raw = fh.read()
                if len(raw) == 128:
                    return raw
        except OSError:
            pass
        if time.time() - t0 > timeout_s:
            raise TimeoutError(f"rank {rank}: no fresh RCCL unique id at {path} after {timeout_s} s "
                               f"(is rank 0 of this launch running on this node?)")
        time.sleep(0.01)


class _StdoutToStderr:
    """RCCL prints a version banner on the C stdout of the process; a job's stdout carries one JSON line.  While the
    communicator is created, file descriptor 1 points at stderr, and the C buffers are flushed before it is restored."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        try:
            C.CDLL(None).fflush(None)
        except (OSError, AttributeError):
            pass
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


_HARD_EXIT = []


def _exit_hard_at_end():
    """A thread of this process is stuck inside ncclCommInitRank (two ranks on one device, a rank that never came):
    the interpreter would finish its work and then hang in the runtime's teardown, waiting for that thread.  From
    now on the process leaves through os._exit once Python's own exit handlers have run - with the status it would
    have had (1 after an uncaught exception)."""
    if _HARD_EXIT:
        return
    _HARD_EXIT.append(True)
    import atexit
    code = {"v": 0}
    real_exit, real_hook = sys.exit, sys.excepthook

    def exit_recording(status=0):          # sys.exit(n) -> SystemExit(n): remember n for the hard exit below
        code["v"] = status if isinstance(status, int) else (0 if status is None else 1)
        real_exit(status)

    def hook_recording(*exc):
        code["v"] = 1
        real_hook(*exc)

    sys.exit, sys.excepthook = exit_recording, hook_recording

    def bye():
        # atexit runs handlers last-registered-first: this one would run BEFORE everything registered earlier (logging
        # shutdown, result writers) and os._exit would skip them - so it runs them itself, then leaves.  CPython keeps a
        # callback registered while it runs: without the unregister the call below would enter bye() again and recurse
        # until RecursionError, with the earlier handlers never reached
        atexit.unregister(bye)
        try:
            atexit._run_exitfuncs()
        except Exception:                      # noqa: BLE001 - a failing handler must not keep the process alive
            code["v"] = code["v"] or 1
        for fh in (sys.stdout, sys.stderr):
            try:
                fh.flush()
            except (OSError, ValueError):
                pass
        os._exit(code["v"] if code["v"] else (1 if getattr(sys, "last_value", None) is not None else 0))

    atexit.register(bye)


class RcclCollective:
    """RCCL over xGMI through ctypes; one communicator per process, collectives on the engine's stream."""
    kind = "rccl"

    def __init__(self, eng, rank: int, world: int, timeout_s: float = 300.0):
        self.eng, self.rank, self.world = eng, int(rank), int(world)
        self.lib = C.CDLL("librccl.so")
        L = self.lib
        L.ncclGetErrorString.restype = C.c_char_p
        L.ncclGetUniqueId.argtypes = [C.POINTER(_ncclUniqueId)]
        L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _ncclUniqueId, C.c_int]
        L.ncclCommDestroy.argtypes = [C.c_void_p]
        L.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        L.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        for name in ("ncclCommCount", "ncclCommUserRank", "ncclCommCuDevice"):
            getattr(L, name).argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        uid = _ncclUniqueId()
        rdz = launch_rendezvous(self.rank, self.world, timeout_s)
        path = _rendezvous_path()
        self._path = path if (self.rank == 0 and rdz is None) else None
        with _StdoutToStderr():
            if self.rank == 0:
                self._ok(L.ncclGetUniqueId(C.byref(uid)))
                raw = C.string_at(C.addressof(uid), 128)                    # all 128 bytes, zeros included
                if rdz is not None:
                    rdz.publish("rccl_id", raw)
                else:
                    publish_id(path, raw)
            else:
                raw = rdz.fetch("rccl_id", timeout_s) if rdz is not None else await_id(path, timeout_s, self.rank)
                if len(raw) != 128:
                    raise RuntimeError(f"rank {self.rank}: the rendezvous returned {len(raw)} bytes for the RCCL id")
                C.memmove(C.byref(uid), raw, 128)
            eng.sync()                                 # binds this thread to the engine's device (the comm's device)
            self.comm = C.c_void_p()
            # ncclCommInitRank blocks until every rank has called it with the same id: a mismatched id (or a rank
            # that died) must end in an error, not in a hang
            import threading
            res = {}

            def init():
                try:
                    eng.sync()                         # the HIP device is per thread: bind this one as well
                    res["st"] = L.ncclCommInitRank(C.byref(self.comm), self.world, uid, self.rank)
                except BaseException as exc:           # noqa: BLE001 - re-raised on the calling thread below
                    res["exc"] = exc

            if self.world > 1:
                th = threading.Thread(target=init, daemon=True)
                th.start()
                th.join(timeout_s)
                if th.is_alive():
                    _exit_hard_at_end()
                    raise TimeoutError(f"rank {self.rank}: ncclCommInitRank did not return within {timeout_s} s "
                                       f"({self.world} ranks expected; id through "
                                       f"{'the socket rendezvous' if rdz is not None else 'the file ' + path})")
            else:
                init()
        if "exc" in res:
            raise RuntimeError(f"rank {self.rank}: creating the RCCL communicator failed: {res['exc']!r}") from res["exc"]
        self._ok(res.get("st", -1))

    def describe(self):
        """What the communicator itself says about this rank: (ranks in the communicator, this rank's number there, its
        device ordinal) - whether RCCL saw N ranks on N devices is answerable from these."""
        out = []
        for name in ("ncclCommCount", "ncclCommUserRank", "ncclCommCuDevice"):
            v = C.c_int(-1)
            self._ok(getattr(self.lib, name)(self.comm, C.byref(v)))
            out.append(int(v.value))
        return tuple(out)

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


class FileCollective:
    """Single-node stand-in for the three operations of a run (barrier, max, one all-gather of a few KB) through
    files in the private rendezvous directory: rank r publishes ``<tag>.fc.<seq>.<r>`` (written aside, renamed into
    place) and polls for the other ranks' files of the same sequence number.  Not a data-path collective - there is
    none - and slow (milliseconds per operation); it exists so that a job whose RCCL communicator cannot be created
    still gathers its observables and times its steps instead of dying at start-up (``make_collective``)."""
    kind = "file"

    def __init__(self, rank: int, world: int, timeout_s: float = 600.0, tag: str = "fc"):
        self.rank, self.world, self.timeout_s = int(rank), int(world), float(timeout_s)
        self._base = _rendezvous_path() + "." + tag
        self._seq = 0
        self._t_start = _process_start_time()
        # Every message carries the launch's nonce: a file an earlier launch with the same tag left behind (same name,
        # same sequence number, possibly only seconds old) can never be taken for one of this launch.  Rank 0 draws it
        # and publishes it like the RCCL id (exclusive create + rename; readers insist on a file no older than they are).
        # With the socket rendezvous the nonce is the launch's own (and the file names carry it).
        rdz = launch_rendezvous(self.rank, self.world, self.timeout_s)
        if rdz is not None:
            self._base = _rendezvous_path() + "." + tag
            raw = rdz.nonce
        else:
            npath = self._base + ".nonce"
            if self.rank == 0:
                raw = os.urandom(8) + b"\0" * 120
                publish_id(npath, raw)
            else:
                raw = await_id(npath, self.timeout_s, self.rank)
        self._nonce = raw[:8]
        self._mine = []

    def _name(self, seq, rank):
        return f"{self._base}.{seq}.{rank}"

    def _exchange(self, row: np.ndarray) -> np.ndarray:
        row = np.ascontiguousarray(row, dtype=np.float64).ravel()
        seq = self._seq
        self._seq += 1
        tmp = self._name(seq, self.rank) + ".tmp"
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600)
        with os.fdopen(fd, "wb") as fh:
            fh.write(self._nonce + np.int64(row.size).tobytes() + row.tobytes())
        os.replace(tmp, self._name(seq, self.rank))
        self._mine.append(self._name(seq, self.rank))
        # every rank that writes sequence number s has read all files of s - 1, hence every rank has written s - 1,
        # hence finished reading s - 2: this rank's file of s - 2 can go
        try:
            os.remove(self._name(seq - 2, self.rank))
            self._mine.remove(self._name(seq - 2, self.rank))
        except (OSError, ValueError):
            pass
        out = np.empty((self.world, row.size))
        t0 = time.time()
        for r in range(self.world):
            while True:
                try:
                    with open(self._name(seq, r), "rb") as fh:
                        raw = fh.read()
                    if raw[:8] != self._nonce:
                        raise FileNotFoundError              # left by another launch with the same tag
                    raw = raw[8:]
                    n = int(np.frombuffer(raw[:8], dtype=np.int64)[0]) if len(raw) >= 8 else -1
                    if n == row.size and len(raw) == 8 + 8 * n:
                        out[r] = np.frombuffer(raw[8:], dtype=np.float64)
                        break
                    if n >= 0 and n != row.size:
                        raise RuntimeError(f"rank {self.rank}: rank {r} sent {n} values where {row.size} were expected")
                except FileNotFoundError:
                    pass
                if time.time() - t0 > self.timeout_s:
                    raise TimeoutError(f"rank {self.rank}: no message {seq} from rank {r} within {self.timeout_s} s "
                                       f"({self._name(seq, r)})")
                time.sleep(0.0005)
        return out

    def barrier(self):
        self._exchange(np.zeros(1))

    def allreduce_max(self, value: float) -> float:
        return float(self._exchange(np.array([value])).max())

    def allgather(self, row: np.ndarray) -> np.ndarray:
        return self._exchange(row)

    def close(self):
        # two barriers: whoever leaves the second one knows that every rank has WRITTEN its file of the second, hence
        # finished reading the first - so everything up to the first can go now; this rank's file of the second may
        # still be read by a slower rank and stays (a few dozen bytes; it carries the nonce, so no later launch with the
        # same tag mistakes it for one of its own)
        self.barrier()
        self.barrier()
        last = self._name(self._seq - 1, self.rank)
        for name in list(self._mine):
            if name != last:
                try:
                    os.remove(name)
                except OSError:
                    pass
                self._mine.remove(name)


class CollectiveUnavailable(RuntimeError):
    """The RCCL communicator could not be created on every rank and the caller did not allow the file fallback."""


def make_collective(eng=None, backend=None, strict=False):
    """Collective of this process from the launcher's environment (RANK / WORLD_SIZE as set by
    torch.distributed.run): serial for one process, RCCL (ctypes) otherwise.  Whether RCCL is used is decided by
    ALL ranks together (a vote through ``FileCollective``): if the communicator cannot be created on any rank -
    library missing, rendezvous or ``ncclCommInitRank`` timing out (``MPSE_RCCL_TIMEOUT``, default 120 s) - every rank
    falls back to the file collective and says so; ``backend="rccl"`` (or MPSE_COLLECTIVE=rccl) forbids the fallback,
    MPSE_COLLECTIVE=file skips RCCL.  ``strict=True`` (a scaling run: ``bench.py --gpus N``): the vote still takes place
    - so that every rank learns the outcome and none is left waiting - but a failed vote raises
    ``CollectiveUnavailable`` on EVERY rank instead of degrading: a run that was to measure RCCL over xGMI must not
    print a number obtained without it."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    want = backend or os.environ.get("MPSE_COLLECTIVE", "")
    if world == 1 and want != "rccl":
        return SerialCollective()
    if want == "file":
        launch_rendezvous(rank, world, 600.0)
        return FileCollective(rank, world)
    if eng is None:
        from .engine import get_engine
        eng = get_engine()
    timeout = float(os.environ.get("MPSE_RCCL_TIMEOUT", "120"))
    launch_rendezvous(rank, world, timeout + 300.0)        # rank 0 listens before anything else can wait for it
    if want == "rccl" or world == 1:
        return RcclCollective(eng, rank, world, timeout_s=timeout)
    vote = FileCollective(rank, world, timeout_s=timeout + 300.0, tag="vote")
    rc, err = None, None
    try:
        rc = RcclCollective(eng, rank, world, timeout_s=timeout)
    except (TimeoutError, RuntimeError, OSError) as e:       # (OSError: librccl.so not loadable)
        err = e
    failed = vote.allreduce_max(0.0 if rc is not None else 1.0) > 0.0
    if not failed:
        vote.close()
        return rc
    if strict:
        vote.close()
        raise CollectiveUnavailable(f"rank {rank}: the RCCL communicator is not available on every rank"
                                    f"{' (' + str(err) + ')' if err else ''}; set MPSE_COLLECTIVE=file to run with "
                                    "the file-based barrier / gather instead")
    sys.stderr.write(f"[rank {rank}] RCCL communicator not available on every rank"
                     f"{' (' + str(err) + ')' if err else ''}: barrier / gather through files instead\n")
    sys.stderr.flush()
    vote.close()
    return FileCollective(rank, world)       # (a communicator this rank did create is abandoned, not used)


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
