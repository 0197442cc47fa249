"""Host-side rendezvous for one-process-per-GPU runs: stdlib sockets only.

The library's RCCL communicator needs ONE thing from the host: the 128-byte unique id of rank 0
handed to every other rank (include/synth_mi355x.h: smx_comm_unique_id / smx_bank_comm_init).
bench.py additionally wants a barrier and a max over ranks for its timing.  Both are a few bytes
between processes of one node, so this is a plain TCP star on 127.0.0.1 -- no torch, no MPI, no
GPU call: rank 0 listens on an ephemeral port and publishes it in a file of the rendezvous
directory; the others connect.  Every operation is an all-gather of a small byte string.

Directory: $SMX_RDZV_DIR (bench.py's own launcher creates one), else derived from what
`python -m torch.distributed.run` exports to its workers (MASTER_PORT + the agent's pid, which all
workers share as their parent) -- torchrun's own store occupies MASTER_PORT, so it is not reused.
"""
import os
import socket
import struct
import tempfile
import time


class RendezvousError(RuntimeError):
    pass


def _send(sock, payload):
    sock.sendall(struct.pack("<I", len(payload)) + payload)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise RendezvousError("rendezvous peer closed the connection")
        buf += chunk
    return bytes(buf)


def _recv(sock):
    (n,) = struct.unpack("<I", _recv_exact(sock, 4))
    return _recv_exact(sock, n)


def default_directory(env=os.environ):
    d = env.get("SMX_RDZV_DIR")
    if d:
        return d
    return os.path.join(tempfile.gettempdir(), "smx_rdzv_%s_%d" % (env.get("MASTER_PORT", "0"), os.getppid()))


class Rendezvous:
    """world == 1: every operation is local and no socket is opened."""

    def __init__(self, rank, world, directory=None, addr=None, timeout=120.0):
        self.rank, self.world, self.timeout = rank, world, timeout
        self.dir = directory or default_directory()
        self.addr = addr or "127.0.0.1"
        self._peers = {}           # rank 0: rank -> socket
        self._sock = None          # others: socket to rank 0
        self._listener = None
        if world <= 1:
            return
        os.makedirs(self.dir, exist_ok=True)
        port_file = os.path.join(self.dir, "port")
        if rank == 0:
            try:
                os.unlink(port_file)                   # a file left by a run that died (same directory name)
            except OSError:
                pass
            ls = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            ls.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            ls.bind((self.addr, 0))
            ls.listen(world)
            ls.settimeout(timeout)
            self._listener = ls
            tmp = port_file + ".tmp.%d" % os.getpid()
            with open(tmp, "w") as f:
                f.write("%d %d\n" % (ls.getsockname()[1], os.getpid()))
            os.replace(tmp, port_file)                 # atomic: readers see nothing or all of it
            try:
                while len(self._peers) < world - 1:
                    c, _ = ls.accept()
                    c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    c.settimeout(timeout)
                    try:
                        magic, r = struct.unpack("<4sI", _recv_exact(c, 8))
                    except (RendezvousError, OSError):
                        magic, r = b"", 0
                    if magic != b"SMX1" or r <= 0 or r >= world or r in self._peers:
                        c.close()                      # not one of ours (or a duplicate): keep listening
                        continue
                    c.sendall(b"SMX1")
                    self._peers[r] = c
            except socket.timeout:
                raise RendezvousError("rendezvous: %d of %d ranks arrived within %.0f s"
                                      % (len(self._peers) + 1, world, timeout))
        else:
            deadline = time.monotonic() + timeout
            last_err = None
            while True:
                if time.monotonic() > deadline:
                    raise RendezvousError("rendezvous: rank %d could not reach rank 0 via %s (%s)"
                                          % (rank, port_file, last_err))
                try:
                    with open(port_file) as f:
                        port = int(f.read().split()[0])
                    s = socket.create_connection((self.addr, port), timeout=2.0)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    s.settimeout(timeout)
                    s.sendall(struct.pack("<4sI", b"SMX1", rank))
                    if _recv_exact(s, 4) != b"SMX1":               # whoever listens there is not rank 0
                        raise OSError("not a rendezvous server")
                    self._sock = s
                    break
                except (OSError, ValueError, IndexError, RendezvousError) as e:   # not published yet / stale file
                    last_err = e
                    time.sleep(0.02)

    # ---- the one primitive ------------------------------------------------------------
    def allgather(self, payload=b""):
        """-> [payload of rank 0, ..., payload of rank world-1], the same list on every rank."""
        if self.world <= 1:
            return [payload]
        try:
            if self.rank == 0:
                parts = [payload] + [_recv(self._peers[r]) for r in range(1, self.world)]
                blob = b"".join(struct.pack("<I", len(p)) + p for p in parts)
                for r in range(1, self.world):
                    _send(self._peers[r], blob)
                return parts
            _send(self._sock, payload)
            blob = _recv(self._sock)
        except socket.timeout:
            raise RendezvousError("rendezvous: rank %d timed out waiting for the other ranks" % self.rank)
        parts, off = [], 0
        while off < len(blob):
            (n,) = struct.unpack_from("<I", blob, off)
            parts.append(blob[off + 4:off + 4 + n])
            off += 4 + n
        return parts

    def barrier(self):
        self.allgather(b"")

    def broadcast(self, payload, root=0):
        return self.allgather(payload if self.rank == root else b"")[root]

    def max_floats(self, values):
        """Element-wise max over ranks of a short list of floats."""
        rows = [struct.unpack("<%dd" % len(values), p)
                for p in self.allgather(struct.pack("<%dd" % len(values), *values))]
        return [max(col) for col in zip(*rows)]

    def all_ok(self, ok):
        """True iff every rank passed True (a failing rank is reported everywhere)."""
        return all(p == b"\x01" for p in self.allgather(b"\x01" if ok else b"\x00"))

    def close(self):
        for s in list(self._peers.values()) + [self._sock, self._listener]:
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self._peers, self._sock, self._listener = {}, None, None
        if self.world > 1 and self.rank == 0:
            try:
                os.unlink(os.path.join(self.dir, "port"))
            except OSError:
                pass
