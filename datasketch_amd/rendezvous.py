"""Process-group plumbing for the sharded path, without PyTorch: a TCP star over which ranks
exchange small host payloads (the 128-byte RCCL id, shard sizes, timings, barriers).

Data never travels here on a GPU job -- signature shards move over RCCL/xGMI through libmhx's own
binding (``mhx_comm_*``).  On a host without GPUs the same collectives carry the shards themselves,
which is the CPU stand-in the tests use.

Every rank calls the collectives in the same order.  Rank 0 owns the listening socket; a collective
is one frame from every rank to rank 0 and one frame back, so a barrier costs one round trip
(~50 us on loopback).

Address of rank 0, in order of preference:
  * ``MHX_RDZV_ADDR=host:port`` (set by ``bench.py`` when it spawns its own ranks);
  * ``MASTER_ADDR`` / ``MASTER_PORT`` as exported by ``python -m torch.distributed.run``: that port
    itself belongs to the launcher's store, so rank 0 binds a free port and publishes it in
    ``$TMPDIR/mhx_rdzv_<uid>_<MASTER_PORT>_<parent pid>`` (all ranks of one node share the launcher
    as parent); with ranks on several nodes (``LOCAL_WORLD_SIZE`` < ``WORLD_SIZE``) rank 0 listens
    on ``MASTER_PORT + 1`` instead.
"""
from __future__ import annotations

import json
import os
import socket
import struct
import tempfile
import time
from typing import List, Optional, Sequence

_MAGIC = b"MHXR"
_HDR = struct.Struct("<4sIQ")  # magic, rank, payload length


def _recv_exact(sock: socket.socket, n: int) -> bytes:
    buf = bytearray(n)
    view = memoryview(buf)
    got = 0
    while got < n:
        r = sock.recv_into(view[got:], n - got)
        if r == 0:
            raise ConnectionError("rendezvous peer closed the connection")
        got += r
    return bytes(buf)


def _send_frame(sock: socket.socket, rank: int, payload: bytes) -> None:
    sock.sendall(_HDR.pack(_MAGIC, rank, len(payload)) + payload)


def _recv_frame(sock: socket.socket):
    magic, rank, n = _HDR.unpack(_recv_exact(sock, _HDR.size))
    if magic != _MAGIC:
        raise ConnectionError("not a libmhx rendezvous peer")
    return rank, _recv_exact(sock, n)


def free_port(host: str = "127.0.0.1") -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind((host, 0))
        return s.getsockname()[1]


class Group:
    """``world`` processes, this one being ``rank``.  Collectives: :meth:`allgather`,
    :meth:`broadcast`, :meth:`barrier`, :meth:`allreduce_max`."""

    def __init__(self, rank: int, world: int, host: str = "127.0.0.1", port: int = 0, timeout: float = 120.0,
                 publish: Optional[str] = None):
        if not (0 <= rank < world):
            raise ValueError("rank out of range")
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        self._peers: List[Optional[socket.socket]] = [None] * world  # rank 0 only
        self._sock: Optional[socket.socket] = None                   # ranks > 0: connection to rank 0
        self._listener: Optional[socket.socket] = None
        self._publish = publish
        if world == 1:
            return
        if rank == 0:
            lst = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            lst.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            # a loopback / literal address is bound as given; a host NAME (MASTER_ADDR on a cluster) may not
            # resolve to a local interface, so rank 0 then listens on all of them
            lst.bind((host if host[:1].isdigit() else "", port))
            lst.listen(world)
            lst.settimeout(timeout)
            self._listener = lst
            self.port = lst.getsockname()[1]
            if publish:
                tmp = publish + ".%d.tmp" % os.getpid()
                with open(tmp, "w") as f:
                    json.dump({"host": host if host[:1].isdigit() else "127.0.0.1", "port": self.port, "pid": os.getpid(),
                               "time": time.time()}, f)
                os.replace(tmp, publish)
            for _ in range(world - 1):
                conn, _addr = lst.accept()
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.settimeout(timeout)
                peer, _ = _recv_frame(conn)
                if not (0 < peer < world) or self._peers[peer] is not None:
                    raise ConnectionError(f"unexpected rendezvous peer rank {peer}")
                self._peers[peer] = conn
        else:
            deadline = time.time() + timeout
            last: Optional[Exception] = None
            while True:
                target = (host, port)
                if publish:  # the port is whatever rank 0 published (re-read: a stale file may still be there)
                    try:
                        with open(publish) as f:
                            info = json.load(f)
                        target = (info["host"], int(info["port"]))
                    except (OSError, ValueError, KeyError) as e:
                        last, target = e, None
                if target is not None:
                    try:
                        s = socket.create_connection(target, timeout=5.0)
                        s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        s.settimeout(timeout)
                        _send_frame(s, rank, b"")
                        self._sock = s
                        break
                    except OSError as e:
                        last = e
                if time.time() > deadline:
                    raise TimeoutError(f"rank {rank}: no rendezvous with rank 0 within {timeout:.0f} s ({last!r})")
                time.sleep(0.05)
            self.port = target[1]

    # -- collectives ------------------------------------------------------------------------
    def allgather(self, payload: bytes) -> List[bytes]:
        """Every rank contributes ``payload``; every rank receives the list ordered by rank."""
        payload = bytes(payload)
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            parts: List[bytes] = [payload] + [b""] * (self.world - 1)
            for r in range(1, self.world):
                peer, data = _recv_frame(self._peers[r])
                parts[peer] = data
            blob = b"".join(struct.pack("<Q", len(p)) + p for p in parts)
            for r in range(1, self.world):
                _send_frame(self._peers[r], 0, blob)
            return parts
        _send_frame(self._sock, self.rank, payload)
        _, blob = _recv_frame(self._sock)
        parts, pos = [], 0
        for _ in range(self.world):
            (n,) = struct.unpack_from("<Q", blob, pos)
            parts.append(blob[pos + 8: pos + 8 + n])
            pos += 8 + n
        return parts

    def broadcast(self, payload: Optional[bytes], src: int = 0) -> bytes:
        return self.allgather(payload if self.rank == src and payload is not None else b"")[src]

    def barrier(self) -> None:
        self.allgather(b"")

    def allreduce_max(self, x: float) -> float:
        return max(struct.unpack("<d", p)[0] for p in self.allgather(struct.pack("<d", float(x))))

    def allgather_ints(self, values: Sequence[int]) -> List[List[int]]:
        fmt = "<%dq" % len(values)
        return [list(struct.unpack(fmt, p)) for p in self.allgather(struct.pack(fmt, *[int(v) for v in values]))]

    def close(self) -> None:
        for s in [self._sock, self._listener] + [p for p in self._peers if p is not None]:
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self._sock = self._listener = None
        self._peers = [None] * self.world
        if self.rank == 0 and self._publish:
            try:
                os.unlink(self._publish)
            except OSError:
                pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def from_env(timeout: float = 120.0) -> Group:
    """The group this process belongs to, from the launcher's environment (``RANK``, ``WORLD_SIZE`` and
    ``MHX_RDZV_ADDR`` or ``MASTER_ADDR``/``MASTER_PORT``); a lone process gets a group of one."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1:
        return Group(0, 1)
    addr = os.environ.get("MHX_RDZV_ADDR")
    if addr:
        host, _, port = addr.rpartition(":")
        return Group(rank, world, host or "127.0.0.1", int(port), timeout)
    host = os.environ.get("MASTER_ADDR", "127.0.0.1")
    mport = int(os.environ.get("MASTER_PORT", "29500"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if local_world == world:  # one node: rank 0 publishes a free port under the launcher's pid
        uid = os.getuid() if hasattr(os, "getuid") else 0
        path = os.path.join(tempfile.gettempdir(), f"mhx_rdzv_{uid}_{mport}_{os.getppid()}")
        return Group(rank, world, host, 0, timeout, publish=path)
    return Group(rank, world, host, mport + 1, timeout)
