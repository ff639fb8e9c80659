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
    ``$TMPDIR/mhx_rdzv_<uid>/<MASTER_PORT>_<parent pid>`` (a 0700 directory of this user; the file is
    created with O_EXCL, mode 0600, and holds a random nonce every joining rank must present; all
    ranks of one node share the launcher as parent) and listens on the loopback interface only; with
    ranks on several nodes (``LOCAL_WORLD_SIZE`` < ``WORLD_SIZE``) rank 0 listens on
    ``MASTER_PORT + 1`` of the address ``MASTER_ADDR`` resolves to (``MHX_RDZV_NONCE``, if the
    launcher exports one to every rank, is the shared secret there).
Connections that are not ranks of the group (wrong magic, wrong nonce, a rank twice, an oversized
frame) are dropped; they do not abort the group.
"""
from __future__ import annotations

import hmac
import json
import os
import secrets
import socket
import struct
import tempfile
import time
from typing import List, Optional, Sequence

_MAGIC = b"MHXR"
_HDR = struct.Struct("<4sIQ")  # magic, rank, payload length
_HELLO_MAX = 256               # a hello frame carries the group's nonce and nothing else
_ACK = b"mhx-joined"           # rank 0's answer to a hello it accepted
MAX_FRAME = 1 << 32            # no frame is larger (the CPU stand-in carries signature shards; a GPU job a few hundred bytes)


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


def _recv_frame(sock: socket.socket, limit: int = MAX_FRAME):
    magic, rank, n = _HDR.unpack(_recv_exact(sock, _HDR.size))
    if magic != _MAGIC:
        raise ConnectionError("not a libmhx rendezvous peer")
    if n > limit:  # the length comes from the peer: never allocate on its say-so alone
        raise ConnectionError(f"rendezvous frame of {n} bytes exceeds the limit of {limit}")
    return rank, _recv_exact(sock, n)


def free_port(host: str = "127.0.0.1") -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind((host, 0))
        return s.getsockname()[1]


def _listen_address(host: str, port: int, single_node: bool):
    """(family, sockaddr) rank 0 binds: the loopback interface when every rank is on this node; otherwise the address
    MASTER_ADDR resolves to (IPv4 or IPv6) when that is an address of this host, else every interface of its family."""
    if single_node:
        return (socket.AF_INET6, ("::1", port, 0, 0)) if host == "::1" else (socket.AF_INET, ("127.0.0.1", port))
    try:
        infos = socket.getaddrinfo(host or None, port, type=socket.SOCK_STREAM, flags=socket.AI_PASSIVE if not host else 0)
    except socket.gaierror:
        infos = []
    for family, _t, _p, _c, sockaddr in infos:
        try:  # bindable = one of this host's own addresses
            with socket.socket(family, socket.SOCK_STREAM) as probe:
                probe.bind(sockaddr[:1] + (0,) + sockaddr[2:])
            return family, sockaddr
        except OSError:
            continue
    family = infos[0][0] if infos else socket.AF_INET
    return family, (("::", port, 0, 0) if family == socket.AF_INET6 else ("", port))


def _publish_dir() -> str:
    """A directory only this user can write: $TMPDIR/mhx_rdzv_<uid> (0700, owned by us, not a symlink)."""
    uid = os.getuid() if hasattr(os, "getuid") else 0
    path = os.path.join(tempfile.gettempdir(), f"mhx_rdzv_{uid}")
    try:
        os.mkdir(path, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(path)
    import stat

    if not stat.S_ISDIR(st.st_mode) or (hasattr(os, "getuid") and st.st_uid != uid) or (st.st_mode & 0o077):
        raise PermissionError(f"{path} is not a private directory of this user")
    return path


def _socket_is_dead(sock) -> bool:
    """True when the peer has closed this connection (EOF or an error on a non-blocking peek); pending data or silence = alive."""
    try:
        return sock.recv(1, socket.MSG_PEEK | socket.MSG_DONTWAIT) == b""
    except (BlockingIOError, InterruptedError):
        return False
    except OSError:
        return True


class Group:
    """``world`` processes, this one being ``rank``.  Collectives: :meth:`allgather`,
    :meth:`broadcast`, :meth:`barrier`, :meth:`allreduce_max`.

    ``nonce``: a secret every rank of the group knows (``MHX_RDZV_NONCE`` from the spawning parent, or the one
    rank 0 writes next to its port in the published file); a connection whose hello does not carry it is dropped.
    ``single_node``: rank 0 listens on the loopback interface only."""

    def __init__(self, rank: int, world: int, host: str = "127.0.0.1", port: int = 0, timeout: float = 120.0,
                 publish: Optional[str] = None, nonce: Optional[str] = None, single_node: Optional[bool] = None):
        if not (0 <= rank < world):
            raise ValueError("rank out of range")
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        self._peers: List[Optional[socket.socket]] = [None] * world  # rank 0 only
        self._sock: Optional[socket.socket] = None                   # ranks > 0: connection to rank 0
        self._listener: Optional[socket.socket] = None
        self._publish = publish
        self._published = False
        if world == 1:
            return
        if single_node is None:
            single_node = publish is not None or host in ("127.0.0.1", "localhost", "::1")
        try:
            if rank == 0:
                self._serve(host, port, timeout, publish, nonce, single_node)
            else:
                self._join(host, port, timeout, publish, nonce)
        except BaseException:
            self.close()
            raise

    def _serve(self, host, port, timeout, publish, nonce, single_node):
        family, sockaddr = _listen_address(host, port, single_node)
        lst = socket.socket(family, socket.SOCK_STREAM)
        self._listener = lst
        lst.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        lst.bind(sockaddr)
        lst.listen(self.world)
        self.port = lst.getsockname()[1]
        if publish:
            if nonce is None:
                nonce = secrets.token_hex(16)
            info = {"host": "::1" if family == socket.AF_INET6 and single_node else ("127.0.0.1" if single_node else host),
                    "port": self.port, "pid": os.getpid(), "time": time.time(), "nonce": nonce}
            try:
                os.unlink(publish)  # a stale file of an earlier job of this user (the directory is ours alone)
            except FileNotFoundError:
                pass
            fd = os.open(publish, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
            with os.fdopen(fd, "w") as f:
                json.dump(info, f)
            self._published = True
        want = (nonce or "").encode()
        deadline = time.time() + timeout
        missing = self.world - 1
        while missing:
            left = deadline - time.time()
            if left <= 0:
                raise TimeoutError(f"rank 0: {missing} of {self.world - 1} ranks did not join within {timeout:.0f} s")
            lst.settimeout(left)
            try:
                conn, _addr = lst.accept()
            except socket.timeout:
                continue
            try:  # anything that is not a rank of this group is dropped, not fatal: a port scanner, a stale job
                conn.settimeout(min(1.0, max(0.1, left)))  # (hellos are read one after the other: a silent connection holds the others up this long)
                peer, hello = _recv_frame(conn, _HELLO_MAX)
                if not (0 < peer < self.world) or not hmac.compare_digest(hello, want):
                    raise ConnectionError("unexpected rendezvous peer")
                old = self._peers[peer]
                if old is not None and not (want and _socket_is_dead(old)):
                    # a hello for a rank that is registered already.  Its first ACK may have reached it too late (it waits
                    # min(timeout, 10 s) while rank 0 reads silent connections at 1 s each): it closed that socket and came
                    # again, and the new connection replaces the dead one -- but ONLY when this group has a nonce (so that the
                    # hello proves membership; without one any local process could evict a live peer by claiming its rank) and
                    # the registered socket really is dead (the peer closed it: EOF or an error on a peek).  ADVICE r5.
                    raise ConnectionError("rank already registered")
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.settimeout(timeout)
                _send_frame(conn, 0, _ACK)  # the joiner waits for this: a rank that was dropped must not believe it has joined
                if old is not None:
                    try:
                        old.close()
                    except OSError:
                        pass
                else:
                    missing -= 1
                self._peers[peer] = conn
            except (OSError, struct.error):
                conn.close()

    def _join(self, host, port, timeout, publish, nonce):  # noqa: C901
        deadline = time.time() + timeout
        last: Optional[Exception] = None
        while True:
            target, hello = (host, port), (nonce or "")
            if publish:  # port and nonce are whatever rank 0 published (re-read: a stale file may still be there)
                try:
                    with open(publish) as f:
                        info = json.load(f)
                    target, hello = (info["host"], int(info["port"])), str(info.get("nonce", ""))
                except (OSError, ValueError, KeyError) as e:
                    last, target = e, None
            if target is not None:
                try:
                    s = socket.create_connection(target, timeout=5.0)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    s.settimeout(min(timeout, 10.0))
                    _send_frame(s, self.rank, hello.encode())
                    src, ack = _recv_frame(s, _HELLO_MAX)  # rank 0 answers an accepted hello; EOF (dropped) -> try again
                    if src != 0 or ack != _ACK:
                        raise ConnectionError("rank 0 did not acknowledge the hello")
                    s.settimeout(timeout)
                    self._sock = s
                    break
                except (OSError, struct.error) as e:
                    last = e
                    if "s" in locals() and s is not None and s is not self._sock:
                        try:
                            s.close()
                        except OSError:
                            pass
                        s = None
            if time.time() > deadline:
                raise TimeoutError(f"rank {self.rank}: no rendezvous with rank 0 within {timeout:.0f} s ({last!r})")
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
        for comm in list(getattr(self, "_mhx_comms", {}).values()):  # RCCL communicators made for this group (dist.communicator)
            try:
                comm.close()
            except Exception:  # noqa: BLE001
                pass
        if hasattr(self, "_mhx_comms"):
            self._mhx_comms.clear()
        for s in [self._sock, self._listener] + [p for p in self._peers if p is not None]:
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self._sock = self._listener = None
        self._peers = [None] * self.world
        if self.rank == 0 and self._publish and self._published:
            self._published = False
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
    nonce = os.environ.get("MHX_RDZV_NONCE")
    addr = os.environ.get("MHX_RDZV_ADDR")
    if addr:
        host, _, port = addr.rpartition(":")
        return Group(rank, world, host.strip("[]") or "127.0.0.1", int(port), timeout, nonce=nonce)
    host = os.environ.get("MASTER_ADDR", "127.0.0.1")
    mport = int(os.environ.get("MASTER_PORT", "29500"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if local_world == world:  # one node: rank 0 publishes a free port (and a nonce) under the launcher's pid
        path = os.path.join(_publish_dir(), f"{mport}_{os.getppid()}")
        return Group(rank, world, host, 0, timeout, publish=path, nonce=nonce, single_node=True)
    return Group(rank, world, host, mport + 1, timeout, nonce=nonce, single_node=False)
