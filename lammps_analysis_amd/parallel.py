"""Multi-GPU sharding: one process per GPU, one exchange step per calculator run.

The units of work are independent: atom-pair tiles or frames for the RDF, frames for
the atom reductions, time origins (windows) for the correlations.  Every rank processes
its share and the results are combined once at the end: integer histograms and
correlation sums by one all-reduce, the per-window integral statistics by the
parallel-variance merge.  The reference itself is single-process
(mdsuite/utils/meta_functions.py:49-51 is its only device notion).

Two transports, no third-party package:

* ``rccl`` -- the all-reduce of libmdx (``mdx_comm_allreduce`` / ``mdx_rdf_reduce``:
  RCCL over xGMI on device memory).  Used when every rank drives a GPU of its own.
* ``host`` -- a star over a local socket: rank 0 collects, adds in rank order and sends
  the sum back.  It carries the rendezvous (the 128-byte RCCL id), Python objects
  (window moments) and barriers in every case, and the array sums when ranks share a
  GPU or have none (tests).  Payloads are 10^4..10^5 bytes once per run.

Ranks are described by the environment a launcher sets (``RANK``, ``LOCAL_RANK``,
``WORLD_SIZE`` as ``python -m torch.distributed.run`` and most MPI-style launchers
export them, or ``MDX_RANK`` / ``MDX_LOCAL_RANK`` / ``MDX_WORLD_SIZE``).  ``world()``
initialises from it on first use; ``init()`` does so explicitly.  ``launch()`` starts
the ranks of one node itself.

File writes: every rank runs the same script against the same project directory, so
only rank 0 (``is_writer()``) writes; ``run_on_writer`` makes the others wait.

Trust: the host channel carries pickled Python objects, so both ends of a connection
prove who they are before the first object is read.  On a ``unix:`` address the kernel
reports the peer's user id (``SO_PEERCRED``) and it must be this process's own; on a
``tcp:`` address both ends answer an HMAC-SHA256 challenge keyed with ``MDX_COMM_TOKEN``
(``launch()`` exports a random one; without a token a ``tcp:`` rendezvous is refused).
The first message of a connection is a fixed-size binary record, never a pickle.

No silent downgrade: when every rank drives a GPU of its own the sums go through RCCL, and
a communicator that cannot be created is an error (``RuntimeError`` on every rank) unless
the host transport was asked for by name (``MDX_COMM_TRANSPORT=host``).  Ranks that share
a GPU or have none (tests) use the host transport; RCCL cannot serve them.
"""
from __future__ import annotations

import atexit
import ctypes
import hashlib
import hmac
import logging
import os
import pickle
import secrets
import socket
import struct
import subprocess
import sys
import time
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

log = logging.getLogger(__name__)

_CONNECT_TIMEOUT_S = float(os.environ.get("MDX_COMM_TIMEOUT", "180"))


# ------------------------------------------------------------------ host channel


def _send_msg(sock: socket.socket, payload: bytes) -> None:
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv_exact(sock: socket.socket, n: int) -> bytes:
    chunks = []
    while n:
        part = sock.recv(min(n, 1 << 20))
        if not part:
            raise ConnectionError("peer closed the connection")
        chunks.append(part)
        n -= len(part)
    return b"".join(chunks)


def _recv_msg(sock: socket.socket) -> bytes:
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    return _recv_exact(sock, n)


def _parse_address(address: str):
    """``unix:<name>`` (abstract socket, this node only) or ``tcp:<host>:<port>``."""
    kind, _, rest = address.partition(":")
    if kind == "unix":
        return socket.AF_UNIX, "\0" + rest
    if kind == "tcp":
        host, _, port = rest.rpartition(":")
        return socket.AF_INET, (host, int(port))
    raise ValueError(f"bad rendezvous address '{address}' (unix:<name> or tcp:<host>:<port>)")


_HELLO = struct.Struct("<4sII")   # magic, rank, world size
_MAGIC = b"MDX1"
_NONCE_BYTES = 32


def _peer_uid(conn: socket.socket) -> int:
    """User id of the process at the other end of a unix-domain connection."""
    creds = conn.getsockopt(socket.SOL_SOCKET, socket.SO_PEERCRED, struct.calcsize("3i"))
    _pid, uid, _gid = struct.unpack("3i", creds)
    return uid


def _token() -> Optional[bytes]:
    value = os.environ.get("MDX_COMM_TOKEN")
    return value.encode() if value else None


def _answer(token: bytes, nonce: bytes, role: bytes) -> bytes:
    return hmac.new(token, role + nonce, hashlib.sha256).digest()


def _authenticate(conn: socket.socket, family: int, serving: bool) -> None:
    """Both ends prove their identity before any pickled object crosses the connection
    (see the module docstring).  Raises PermissionError."""
    if family == socket.AF_UNIX:
        uid = _peer_uid(conn)
        if uid != os.getuid():
            raise PermissionError(f"rendezvous peer runs as uid {uid}, not {os.getuid()}")
        return
    token = _token()
    if token is None:
        raise PermissionError("a tcp: rendezvous needs MDX_COMM_TOKEN (the same secret on "
                              "every rank); parallel.launch() sets one")
    mine = secrets.token_bytes(_NONCE_BYTES)
    conn.sendall(mine)
    theirs = _recv_exact(conn, _NONCE_BYTES)
    # each side answers the other's nonce, tagged with its own role, so an answer cannot
    # be reflected back to its author
    conn.sendall(_answer(token, theirs, b"S" if serving else b"C"))
    proof = _recv_exact(conn, hashlib.sha256().digest_size)
    if not hmac.compare_digest(proof, _answer(token, mine, b"C" if serving else b"S")):
        raise PermissionError("rendezvous peer does not hold MDX_COMM_TOKEN")


class HostChannel:
    """Star over stream sockets: rank 0 serves, ranks 1..n-1 each hold one connection."""

    def __init__(self, rank: int, size: int, address: str):
        self.rank, self.size, self.address = rank, size, address
        family, target = _parse_address(address)
        self._peers: List[Optional[socket.socket]] = [None] * size
        self._server = None
        if size == 1:
            return
        try:
            if rank == 0:
                self._serve(family, target)
            else:
                self._join(family, target)
        except BaseException:
            self.close()
            raise

    def _serve(self, family, target) -> None:
        size, address = self.size, self.address
        server = socket.socket(family, socket.SOCK_STREAM)
        self._server = server
        if family == socket.AF_INET:
            server.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        server.bind(target)
        server.listen(size)
        deadline = time.monotonic() + _CONNECT_TIMEOUT_S
        joined = 0
        while joined < size - 1:
            left = deadline - time.monotonic()
            if left <= 0:
                raise TimeoutError(f"rank 0 waited {_CONNECT_TIMEOUT_S:.0f} s for "
                                   f"{size - 1} rank(s) at {address}")
            server.settimeout(left)
            try:
                conn, _addr = server.accept()
            except socket.timeout:
                continue
            # the hello is read under the rendezvous deadline: a connection that says
            # nothing, or a stranger, is dropped and does not hold up the others
            try:
                conn.settimeout(max(1.0, min(30.0, deadline - time.monotonic())))
                _authenticate(conn, family, serving=True)
                magic, peer, peer_size = _HELLO.unpack(_recv_exact(conn, _HELLO.size))
                if magic != _MAGIC or peer_size != size or not 0 < peer < size \
                        or self._peers[peer] is not None:
                    raise PermissionError(f"unexpected hello (rank {peer} of {peer_size})")
            except (OSError, PermissionError, struct.error) as exc:
                log.warning("rendezvous at %s: dropped a connection (%s)", address, exc)
                conn.close()
                continue
            conn.settimeout(None)
            if family == socket.AF_INET:
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            self._peers[peer] = conn
            joined += 1

    def _join(self, family, target) -> None:
        deadline = time.monotonic() + _CONNECT_TIMEOUT_S
        while True:
            conn = socket.socket(family, socket.SOCK_STREAM)
            try:
                conn.connect(target)
                break
            except (ConnectionRefusedError, FileNotFoundError, OSError):
                conn.close()
                if time.monotonic() > deadline:
                    raise TimeoutError(f"rank {self.rank} could not reach rank 0 at "
                                       f"{self.address}") from None
                time.sleep(0.05)
        self._peers[0] = conn
        conn.settimeout(max(1.0, deadline - time.monotonic()))
        _authenticate(conn, family, serving=False)
        conn.sendall(_HELLO.pack(_MAGIC, self.rank, self.size))
        conn.settimeout(None)
        if family == socket.AF_INET:
            conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)

    def gather(self, obj) -> Optional[list]:
        """Every rank's object in rank order on rank 0, None elsewhere."""
        if self.size == 1:
            return [obj]
        if self.rank == 0:
            out = [obj]
            for peer in range(1, self.size):
                out.append(pickle.loads(_recv_msg(self._peers[peer])))
            return out
        _send_msg(self._peers[0], pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL))
        return None

    def bcast(self, obj):
        """Rank 0's object on every rank."""
        if self.size == 1:
            return obj
        if self.rank == 0:
            payload = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
            for peer in range(1, self.size):
                _send_msg(self._peers[peer], payload)
            return obj
        return pickle.loads(_recv_msg(self._peers[0]))

    def allgather(self, obj) -> list:
        return self.bcast(self.gather(obj))

    def barrier(self) -> None:
        self.allgather(None)

    def close(self) -> None:
        for conn in self._peers:
            if conn is not None:
                try:
                    conn.close()
                except OSError:
                    pass
        if self._server is not None:
            self._server.close()
        self._peers = [None] * self.size
        self._server = None


# ------------------------------------------------------------------------ state


class _State:
    def __init__(self):
        self.initialised = False
        self.rank = 0
        self.size = 1
        self.local_rank = 0
        self.channel: Optional[HostChannel] = None
        self.comm = None          # mdx_comm_t (c_void_p value) when the transport is rccl
        self.transport = "none"   # "none" (one rank), "host", "rccl"
        self.device_index: Optional[int] = None


_state = _State()


def _env_int(*names: str) -> Optional[int]:
    for name in names:
        value = os.environ.get(name)
        if value not in (None, ""):
            return int(value)
    return None


def default_address() -> str:
    """Rendezvous address derived from the launcher's environment."""
    address = os.environ.get("MDX_COMM_ADDR")
    if address:
        return address
    # every rank of one launch has the launcher as its parent; the port tells
    # concurrent launches of one user apart
    port = os.environ.get("MASTER_PORT", "0")
    return f"unix:mdx-{os.getuid()}-{port}-{os.getppid()}"


def init(rank: Optional[int] = None, world_size: Optional[int] = None,
         local_rank: Optional[int] = None, address: Optional[str] = None,
         transport: Optional[str] = None) -> Tuple[int, int]:
    """Join the ranks of this job; returns (rank, world size).

    Selects GPU ``local_rank % device_count`` for this process (``config.device_index``),
    connects the host channel and, when every rank has a GPU of its own, creates the RCCL
    communicator.  ``transport`` (or ``MDX_COMM_TRANSPORT``): ``auto`` (default: RCCL when
    every rank owns a GPU, and then a failure to set it up raises; the host channel when
    ranks share a GPU or have none), ``host`` (the host channel, by request), ``rccl``
    (RCCL or an error).  Nothing is left half-initialised when this raises.
    """
    if _state.initialised:
        return _state.rank, _state.size
    rank = _env_int("MDX_RANK", "RANK") if rank is None else rank
    world_size = _env_int("MDX_WORLD_SIZE", "WORLD_SIZE") if world_size is None else world_size
    local_rank = _env_int("MDX_LOCAL_RANK", "LOCAL_RANK") if local_rank is None else local_rank
    rank = 0 if rank is None else int(rank)
    world_size = 1 if world_size is None else int(world_size)
    local_rank = rank if local_rank is None else int(local_rank)
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside a world of {world_size}")
    transport = (transport or os.environ.get("MDX_COMM_TRANSPORT", "auto")).lower()
    if transport not in ("auto", "host", "rccl"):
        raise ValueError("transport must be auto, host or rccl")
    if world_size == 1 and transport != "rccl":
        _state.rank, _state.size, _state.local_rank = rank, world_size, local_rank
        _state.transport = "none"
        _state.initialised = True
        return rank, world_size

    pending = _State()
    pending.rank, pending.size, pending.local_rank = rank, world_size, local_rank
    try:
        _join_job(pending, transport, address)
    except BaseException:
        # a communicator a peer may still be initialising is left alone (destroying it
        # could pull the ground from under that peer); the sockets are ours to close
        if pending.channel is not None:
            pending.channel.close()
        raise
    _state.__dict__.update(pending.__dict__)
    _state.initialised = True
    atexit.register(shutdown)
    return rank, world_size


def _join_job(state: "_State", transport: str, address: Optional[str]) -> None:
    """The part of init() that can fail: fills ``state`` (channel, device, communicator)."""
    from . import _lib
    from .utils.config import config

    rank, world_size, local_rank = state.rank, state.size, state.local_rank
    n_devices = _lib.device_count()
    if n_devices > 0:
        if config.device_ids:
            state.device_index = int(config.device_ids[local_rank % len(config.device_ids)])
        else:
            state.device_index = local_rank % n_devices
        config.device_index = state.device_index
        _lib.check(_lib.load().mdx_set_device(state.device_index))
    state.channel = HostChannel(rank, world_size, address or default_address())
    state.transport = "host"

    # one GPU per rank and librccl present on every rank -> device collectives
    lib = _lib.load()
    mine = (socket.gethostname(), state.device_index,
            bool(n_devices > 0 and lib.mdx_comm_available()))
    everyone = state.channel.allgather(mine)
    own_gpu = len({(h, d) for h, d, _ in everyone}) == world_size and \
        all(d is not None for _, d, _ in everyone)
    loadable = all(ok for _, _, ok in everyone)
    if transport == "host":
        return
    if not own_gpu:
        if transport == "rccl":
            raise RuntimeError("MDX_COMM_TRANSPORT=rccl, but the ranks do not each own a "
                               f"GPU: {everyone}")
        return  # ranks sharing a GPU (or without one): only the host channel can serve them
    if not loadable:
        raise RuntimeError("every rank owns a GPU but librccl.so cannot be loaded on all of "
                           f"them ({everyone}); set MDX_COMM_TRANSPORT=host to run the sums "
                           "through the host channel instead")
    uid = ctypes.create_string_buffer(_lib.MDX_COMM_ID_BYTES)
    if rank == 0:
        _lib.check(lib.mdx_comm_unique_id(uid, _lib.MDX_COMM_ID_BYTES))
    uid_bytes = state.channel.bcast(uid.raw if rank == 0 else None)
    # The communicator is created on a helper thread: if a peer never shows up (or the
    # node's fabric is misconfigured) ncclCommInitRank blocks for good; the job should
    # then end with an error on every rank rather than hang.  A thread that is still
    # inside the call when the time is up is abandoned (it may finish later; the process
    # is about to fail anyway).
    handle = ctypes.c_void_p()
    outcome = {}

    def create():
        lib.mdx_set_device(state.device_index)  # the current device is a per-thread setting
        rc_ = lib.mdx_comm_create(ctypes.byref(handle), rank, world_size, uid_bytes,
                                  _lib.MDX_COMM_ID_BYTES)
        outcome["error"] = None if rc_ == 0 else lib.mdx_last_error().decode("utf-8", "replace")

    import threading
    worker = threading.Thread(target=create, name="mdx-comm-create", daemon=True)
    worker.start()
    worker.join(float(os.environ.get("MDX_COMM_INIT_TIMEOUT", "120")))
    if worker.is_alive():
        error = "ncclCommInitRank did not return in time"
    else:
        error = outcome.get("error", "mdx_comm_create did not report")
    errors = state.channel.allgather(error)
    if any(e is not None for e in errors):
        raise RuntimeError(
            "the RCCL communicator could not be created: "
            f"{ {r: e for r, e in enumerate(errors) if e} }; the ranks each own a GPU, so "
            "this is not downgraded to host sums (ask for them with MDX_COMM_TRANSPORT=host)")
    state.comm = handle.value
    state.transport = "rccl"


def comm_ranks() -> Optional[int]:
    """Number of ranks the RCCL communicator reports for itself (``mdx_comm_rank``), None
    when the sums do not go through RCCL."""
    _ensure()
    if _state.comm is None:
        return None
    from . import _lib
    r, n = ctypes.c_int(-1), ctypes.c_int(-1)
    _lib.check(_lib.load().mdx_comm_rank(_state.comm, ctypes.byref(r), ctypes.byref(n)))
    if r.value != _state.rank:
        raise RuntimeError(f"RCCL reports rank {r.value}, the job says {_state.rank}")
    return int(n.value)


def shutdown() -> None:
    """Leave the job: destroys the communicator and closes the sockets."""
    if not _state.initialised:
        return
    if _state.comm is not None:
        from . import _lib
        try:
            _lib.load().mdx_comm_destroy(_state.comm)
        except Exception:  # interpreter shutdown
            pass
        _state.comm = None
    if _state.channel is not None:
        _state.channel.close()
        _state.channel = None
    _state.__init__()


def _ensure() -> None:
    if _state.initialised:
        return
    size = _env_int("MDX_WORLD_SIZE", "WORLD_SIZE")
    if size is not None and size > 1:
        init()
    else:
        _state.initialised = True


def world() -> Tuple[int, int]:
    """(rank, world size); (0, 1) for a process started without a launcher."""
    _ensure()
    return _state.rank, _state.size


def transport() -> str:
    _ensure()
    return _state.transport


def communicator():
    """The ``mdx_comm_t`` handle of this rank, or None (one rank, or host transport)."""
    _ensure()
    return _state.comm


def is_writer() -> bool:
    """True on the one rank that writes project files and datasets."""
    return world()[0] == 0


def barrier() -> None:
    _ensure()
    if _state.channel is not None:
        _state.channel.barrier()


def run_on_writer(fn: Callable[[], object]):
    """Run ``fn`` on rank 0 while the other ranks wait; its result (or its failure) is
    delivered to every rank."""
    _ensure()
    if _state.size == 1:
        return fn()
    # nobody is still reading what the writer is about to change
    _state.channel.barrier()
    outcome = None
    if _state.rank == 0:
        try:
            outcome = ("ok", fn())
        except BaseException as exc:  # delivered to all ranks below
            outcome = ("error", f"{type(exc).__name__}: {exc}")
            _state.channel.bcast(outcome)
            raise
    outcome = _state.channel.bcast(outcome)
    if outcome[0] == "error":
        raise RuntimeError(f"rank 0 failed: {outcome[1]}")
    return outcome[1]


# ------------------------------------------------------------------- partitions


def split_range(n: int, rank: int, size: int) -> Tuple[int, int]:
    """Contiguous share ``[begin, end)`` of ``n`` units; sizes differ by at most one."""
    base, extra = divmod(int(n), int(size))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def split_frames(frames: np.ndarray, rank: int, size: int) -> np.ndarray:
    """This rank's share of a list of frame indices."""
    begin, end = split_range(len(frames), rank, size)
    return frames[begin:end]


# ------------------------------------------------------------------ collectives

_REDUCIBLE = {np.dtype(np.uint64): 0, np.dtype(np.int64): 1, np.dtype(np.float64): 2}


def _all_reduce(array: np.ndarray, op: str) -> np.ndarray:
    _ensure()
    if _state.size == 1:
        return array
    arr = np.ascontiguousarray(array)
    if _state.comm is not None:
        from . import _lib
        work = arr if arr.dtype in _REDUCIBLE else arr.astype(
            np.float64 if arr.dtype.kind == "f" else np.int64)
        work = work.copy()
        _lib.check(_lib.load().mdx_comm_allreduce(
            _state.comm, work.ctypes.data, _lib.MDX_HOST, work.size, _REDUCIBLE[work.dtype],
            _lib.MDX_COMM_SUM if op == "sum" else _lib.MDX_COMM_MAX))
        return work.astype(arr.dtype, copy=False)
    parts = _state.channel.gather(arr)
    total = None
    if _state.rank == 0:
        total = parts[0].copy()
        for part in parts[1:]:  # rank order: the same sum on every run
            if op == "sum":
                total += part
            else:
                np.maximum(total, part, out=total)
    return _state.channel.bcast(total)


def all_reduce_sum(array: np.ndarray) -> np.ndarray:
    """Element-wise sum over ranks (the array itself for one rank)."""
    return _all_reduce(array, "sum")


def all_reduce_max(array) -> np.ndarray:
    """Element-wise maximum over ranks."""
    return _all_reduce(np.asarray(array, dtype=np.float64), "max")


def reduce_histograms(rdf_engine) -> np.ndarray:
    """Total pair histograms of an ``engine.RdfEngine`` over all ranks, on every rank:
    in place on the devices (``mdx_rdf_reduce``) with the rccl transport, else through
    the host channel."""
    _ensure()
    if _state.size > 1 and _state.comm is not None:
        rdf_engine.reduce(_state.comm)
        return rdf_engine.read()
    return all_reduce_sum(rdf_engine.read())


def all_gather_objects(obj) -> List:
    """Every rank's object, in rank order."""
    _ensure()
    if _state.size == 1:
        return [obj]
    return _state.channel.allgather(obj)


def broadcast_object(obj):
    """Rank 0's object on every rank."""
    _ensure()
    if _state.size == 1:
        return obj
    return _state.channel.bcast(obj)


def merge_moments(moments):
    """Combine every rank's window statistics (engine.Moments) in rank order."""
    merged = None
    for part in all_gather_objects(moments):
        if part is None:
            continue
        merged = part if merged is None else merged.merge(part)
    return merged


# --------------------------------------------------------------------- launcher


def launch(argv: Sequence[str], n_ranks: int, env: Optional[dict] = None,
           timeout: Optional[float] = None) -> int:
    """Start ``n_ranks`` copies of ``python argv...`` on this node, one per GPU, and wait.

    The parent makes no GPU call.  Rank r gets ``MDX_RANK`` / ``MDX_LOCAL_RANK`` = r,
    ``MDX_WORLD_SIZE`` and a rendezvous address of its own; stdout and stderr are
    inherited.  Returns the first non-zero exit status (0 when all ranks succeeded); the
    remaining ranks are terminated when one fails.
    """
    base = dict(os.environ if env is None else env)
    for name in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        base.pop(name, None)
    base["MDX_WORLD_SIZE"] = str(int(n_ranks))
    base["MDX_COMM_ADDR"] = f"unix:mdx-launch-{os.getpid()}-{time.monotonic_ns()}"
    base.setdefault("MDX_COMM_TOKEN", secrets.token_hex(32))
    procs = []
    for r in range(int(n_ranks)):
        child = dict(base, MDX_RANK=str(r), MDX_LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, *argv], env=child))
    deadline = None if timeout is None else time.monotonic() + timeout
    status = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            rc = p.poll()
            if rc is None:
                continue
            pending.remove(p)
            if rc != 0 and status == 0:
                status = rc
        if status != 0 or (deadline is not None and time.monotonic() > deadline):
            for p in pending:
                p.terminate()
            for p in pending:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            if status == 0:
                status = 124
            break
        time.sleep(0.02)
    return status
