"""
Multi-GPU exchange without torch: one process per GPU, RCCL through the C-ABI (csrc/comm.hip, include/ldsim.h
``ldsim_comm_*``).  The launcher (``python -m torch.distributed.run`` or anything else that sets RANK / WORLD_SIZE /
LOCAL_RANK / MASTER_ADDR / MASTER_PORT) only starts the processes; the communicator is bootstrapped here:
rank 0 draws the ncclUniqueId and serves its 128 bytes over a TCP socket on MASTER_ADDR : MASTER_PORT + 1 + offset.

The path shards by batch (event x TPC group) with no data-path collective; the one exchange reassembles the compact
hit rows on every rank: ``allgather_hits``.
"""
import ctypes as C
import os
import socket
import time

import numpy as np

from . import lib

ID_BYTES = 128
HIT_ROW = np.dtype([("batch", "<i4"), ("pixel", "<i4"), ("adc", "<i4"), ("slot", "<i4"), ("tick", "<f8")])   # 24 B


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


MAGIC = b"LDSIMID2"          # frames every message: a foreign service that happens to own a candidate port is not mistaken for rank 0
PORT_TRIES = 16
TOKEN_BYTES = 16


def job_token(world, addr=None, port=None):
    """What tells this job's ranks from another job's on the same node whose port range overlaps: a hash of the launcher's
    rendezvous (MASTER_ADDR : MASTER_PORT, the elastic run id if there is one) and the world size."""
    import hashlib
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = port or os.environ.get("MASTER_PORT", "29511")
    run = os.environ.get("TORCHELASTIC_RUN_ID", "") + "|" + os.environ.get("LDSIM_JOB_ID", "")
    return hashlib.sha256(f"{addr}:{port}:{run}:{world}".encode()).digest()[:TOKEN_BYTES]


def _recv_exact(s, n):
    buf = b""
    while len(buf) < n:
        chunk = s.recv(n - len(buf))
        if not chunk:
            break
        buf += chunk
    return buf


def exchange_id(payload, rank, world, addr=None, port=None, timeout=300.0, token=None):
    """Rank 0 sends ``payload`` (bytes) to every other rank; returns the payload on every rank.

    Rank 0 listens on the first free port of ``base .. base + PORT_TRIES - 1`` (base = MASTER_PORT + 1 + LDSIM_PORT_OFFSET: the
    launcher's own store sits on MASTER_PORT, and whatever else runs on the node may own the next one); the other ranks go
    round the same candidates.  A worker introduces itself with MAGIC | job token | rank; rank 0 answers -- and counts the rank
    as served -- only for a matching token and a rank of 1 .. world - 1 it has not served yet, so neither a port probe nor a
    rank of another job whose port range overlaps uses up a slot or walks away with this job's id; a worker accepts only an
    answer that carries its own token."""
    if world == 1:
        return payload
    token = token or job_token(world, addr, port)
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    base = int(port or (int(os.environ.get("MASTER_PORT", "29511")) + 1 + int(os.environ.get("LDSIM_PORT_OFFSET", "0"))))
    ports = [base + k for k in range(PORT_TRIES)]
    deadline = time.time() + timeout
    hello = len(MAGIC) + TOKEN_BYTES + 4
    if rank == 0:
        srv = None
        while srv is None:
            for p in ports:
                s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                try:
                    s.bind((addr, p))
                    srv = s
                    break
                except OSError:
                    s.close()
            if srv is None:
                if time.time() > deadline:
                    raise lib.LdsimError(f"rank 0: none of the ports {ports[0]}..{ports[-1]} on {addr} could be bound")
                time.sleep(0.2)
        srv.listen(world + 8)
        served = set()
        try:
            while len(served) < world - 1:
                left = deadline - time.time()
                if left <= 0:
                    raise lib.LdsimError(f"rank 0: ranks {sorted(set(range(1, world)) - served)} did not ask for the ncclUniqueId "
                                         f"within {timeout:.0f} s")
                srv.settimeout(min(left, 5.0))
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    continue
                with conn:
                    try:
                        conn.settimeout(5.0)
                        msg = _recv_exact(conn, hello)
                        r = int.from_bytes(msg[-4:], "little") if len(msg) == hello else -1
                        if (len(msg) != hello or not msg.startswith(MAGIC) or msg[len(MAGIC):len(MAGIC) + TOKEN_BYTES] != token
                                or not 1 <= r < world or r in served):
                            continue                      # a probe, another job's rank, a repeat: not served, not counted
                        conn.sendall(MAGIC + token + payload)
                        served.add(r)
                    except OSError:
                        continue
        finally:
            srv.close()
        return payload
    need = len(MAGIC) + TOKEN_BYTES + ID_BYTES
    while True:
        for p in ports:
            try:
                with socket.create_connection((addr, p), timeout=2.0) as s:
                    s.settimeout(5.0)
                    s.sendall(MAGIC + token + int(rank).to_bytes(4, "little"))
                    buf = _recv_exact(s, need)
                if len(buf) == need and buf.startswith(MAGIC) and buf[len(MAGIC):len(MAGIC) + TOKEN_BYTES] == token:
                    return buf[len(MAGIC) + TOKEN_BYTES:]
            except OSError:
                pass
        if time.time() > deadline:
            raise lib.LdsimError(f"rank {rank}: no ncclUniqueId from rank 0 at {addr}:{ports[0]}..{ports[-1]} within {timeout:.0f} s")
        time.sleep(0.1)


class Communicator:
    """RCCL communicator bound to the process-wide ldsim ctx (one GPU per process)."""

    def __init__(self, ctx, rank=None, world=None):
        r, w, _ = env_world()
        self.rank = r if rank is None else rank
        self.world = w if world is None else world
        self.ctx = ctx
        ident = C.create_string_buffer(ID_BYTES)
        if self.rank == 0:
            lib.check(lib.load().ldsim_comm_unique_id(ident))
        payload = exchange_id(ident.raw if self.rank == 0 else b"", self.rank, self.world)
        lib.check(lib.load().ldsim_comm_init(ctx, C.c_char_p(payload), C.c_int32(self.rank), C.c_int32(self.world)))

    def count(self):
        """(ranks in the communicator, this rank) as RCCL reports them (ncclCommCount / ncclCommUserRank)"""
        n, r = C.c_int32(), C.c_int32()
        lib.check(lib.load().ldsim_comm_count(self.ctx, C.byref(n), C.byref(r)))
        return n.value, r.value

    def allreduce(self, value, op="sum"):
        v = C.c_double(float(value))
        lib.check(lib.load().ldsim_comm_allreduce_f64(self.ctx, C.byref(v), C.c_int32(1 if op == "max" else 0)))
        return v.value

    def barrier(self):
        self.allreduce(0.0)

    def accumulate_hits(self, reset=False):
        lib.check(lib.load().ldsim_hits_accumulate(self.ctx, C.c_int32(int(reset))))

    def allgather_hits(self, download=False):
        """All-gather-v of the rows accumulated since the last reset.  Returns (total rows, per-rank counts[, rows])."""
        p, n = C.c_void_p(), C.c_int64()
        counts = (C.c_int64 * self.world)()
        lib.check(lib.load().ldsim_comm_allgather_hits(self.ctx, C.byref(p), C.byref(n), counts))
        cnt = [int(c) for c in counts]
        if not download:
            return n.value, cnt
        rows = np.zeros(n.value, dtype=HIT_ROW)
        lib.check(lib.load().ldsim_comm_gathered_download(self.ctx, lib.ptr(rows), C.c_int64(n.value)))
        return n.value, cnt, rows

    def destroy(self):
        lib.check(lib.load().ldsim_comm_destroy(self.ctx))
