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


MAGIC = b"LDSIMID1"          # frames the payload: a foreign service that happens to own a candidate port is not mistaken for rank 0
PORT_TRIES = 16


def exchange_id(payload, rank, world, addr=None, port=None, timeout=300.0):
    """Rank 0 sends ``payload`` (bytes) to every other rank; returns the payload on every rank.

    Rank 0 listens on the first free port of ``base .. base + PORT_TRIES - 1`` (base = MASTER_PORT + 1 + LDSIM_PORT_OFFSET: the
    launcher's own store sits on MASTER_PORT, and whatever else runs on the node may own the next one); the other ranks go
    round the same candidates until one of them answers with a framed payload."""
    if world == 1:
        return payload
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    base = int(port or (int(os.environ.get("MASTER_PORT", "29511")) + 1 + int(os.environ.get("LDSIM_PORT_OFFSET", "0"))))
    ports = [base + k for k in range(PORT_TRIES)]
    deadline = time.time() + timeout
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
        srv.listen(world)
        srv.settimeout(timeout)
        served = 0
        while served < world - 1:
            conn, _ = srv.accept()
            with conn:
                conn.sendall(MAGIC + payload)
            served += 1
        srv.close()
        return payload
    need = len(MAGIC) + ID_BYTES
    while True:
        for p in ports:
            try:
                with socket.create_connection((addr, p), timeout=2.0) as s:
                    s.settimeout(5.0)
                    buf = b""
                    while len(buf) < need:
                        chunk = s.recv(need - len(buf))
                        if not chunk:
                            break
                        buf += chunk
                if len(buf) == need and buf.startswith(MAGIC):
                    return buf[len(MAGIC):]
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
