"""CPU: the multi-rank plumbing with gloo, world_size 2 (sharding is disjoint + complete, gather reassembles)."""
import os
import socket
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _allgather_rows(rows):
    """CPU rehearsal of csrc/comm.hip's all-gather-v: the row counts first, then the payload of every rank; returns the
    rows of rank 0, 1, .. back to back and the counts."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    n = torch.tensor([rows.shape[0]], dtype=torch.int64)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    parts = []
    for r in range(world):                                      # one broadcast per rank, as the grouped ncclBroadcasts do
        buf = rows.clone() if r == dist.get_rank() else torch.zeros((counts[r], rows.shape[1]), dtype=rows.dtype)
        dist.broadcast(buf, src=r)
        parts.append(buf)
    return torch.cat(parts, dim=0), counts


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(REPO, "larnd-sim_amd"))
    import torch
    import torch.distributed as dist
    from larndsim_amd import batching, consts, dist as ldist, synth
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    consts.load_snapshot("2x2_no_modvar")
    seg = synth.make_segments(9000, seed=5, segs_per_event=1500, spill=True)
    batching.swap_coordinates(seg)
    bid, order, table = batching.assign_batches(seg)
    idx, my_bid = ldist.shard_segments(bid, order, table, rank, world)
    assert (np.diff(my_bid) >= 0).all() and (my_bid >= 0).all()
    # stand-in for the chain's compact hit rows: one row per owned segment {batch, segment index, ...}
    rows = torch.zeros((len(idx), 6), dtype=torch.int32)
    rows[:, 0] = torch.from_numpy(my_bid.astype(np.int32))
    rows[:, 1] = torch.from_numpy(idx.astype(np.int32))
    gathered, counts = _allgather_rows(rows)
    t = torch.tensor([float(len(idx))], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    q.put((rank, idx, gathered.numpy(), counts, float(t.item()), int((bid >= 0).sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_allgather_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda r: r[0])
    (_, idx0, g0, c0, tot0, nsim), (_, idx1, g1, c1, tot1, _) = res
    assert len(np.intersect1d(idx0, idx1)) == 0                 # disjoint shards
    assert len(idx0) + len(idx1) == nsim == int(tot0) == int(tot1)
    assert abs(len(idx0) - len(idx1)) <= 1500                   # balanced to within one batch
    assert np.array_equal(g0, g1) and c0 == c1 == [len(idx0), len(idx1)]
    assert np.array_equal(np.sort(g0[:, 1]), np.sort(np.r_[idx0, idx1]))
    # rank order == batch order: the concatenation is sorted by batch id, like a single-GPU run
    assert (np.diff(g0[:, 0]) >= 0).all()


def _id_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(REPO, "larnd-sim_amd"))
    from larndsim_amd import comm
    payload = bytes(range(128)) if rank == 0 else b""
    got = comm.exchange_id(payload, rank, world, addr="127.0.0.1", port=port, timeout=60.0)
    q.put((rank, got))


def test_unique_id_rendezvous_world3():
    """The ncclUniqueId hand-out of larndsim_amd/comm.py (rank 0 serves the 128 bytes over TCP): every rank ends up with
    rank 0's bytes, whichever process starts first.  The collectives themselves need GPUs (bench.py --force-dist)."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_id_worker, args=(r, 3, port, q)) for r in (2, 1, 0)]   # rank 0 last: the others retry
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == res[1] == res[2] == bytes(range(128))


def test_unique_id_rendezvous_skips_a_port_someone_else_owns():
    """MASTER_PORT + 1 may belong to another service on the node: rank 0 then listens on the next candidate, and the other
    ranks recognise the foreign listener by the missing frame and move on (no retry on a different port)."""
    import multiprocessing as mp
    import socket
    import threading
    port = _free_port()
    foreign = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    foreign.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    foreign.bind(("127.0.0.1", port))
    foreign.listen(8)
    stop = threading.Event()

    def chatter():                      # answers every connection with 136 bytes that are not a framed id
        foreign.settimeout(0.2)
        while not stop.is_set():
            try:
                c, _ = foreign.accept()
            except OSError:
                continue
            with c:
                c.sendall(b"x" * 136)

    th = threading.Thread(target=chatter, daemon=True)
    th.start()
    try:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_id_worker, args=(r, 2, port, q)) for r in (1, 0)]
        for p in procs:
            p.start()
        res = dict(q.get(timeout=120) for _ in procs)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert res[0] == res[1] == bytes(range(128))
    finally:
        stop.set()
        th.join(timeout=5)
        foreign.close()


def _id_worker_job(rank, world, base_port, job_port, fill, q):
    """a rank of the job whose launcher sits on `job_port` (its token), looking for its id in the candidates from `base_port`"""
    sys.path.insert(0, os.path.join(REPO, "larnd-sim_amd"))
    from larndsim_amd import comm
    payload = bytes([fill]) * 128 if rank == 0 else b""
    got = comm.exchange_id(payload, rank, world, addr="127.0.0.1", port=base_port, timeout=60.0,
                           token=comm.job_token(world, "127.0.0.1", job_port))
    q.put((fill, rank, got))


def test_unique_id_rendezvous_of_two_jobs_on_overlapping_ports():
    """Two jobs on one node whose candidate port ranges overlap (MASTER_PORTs a few apart), and a port probe that connects and
    says nothing: every rank ends with ITS job's id, no serve slot is used up by a stranger (the frame carried only
    a constant, and rank 0 counted every accepted connection)."""
    import multiprocessing as mp
    import threading
    import time
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    base = _free_port()
    # job A owns candidates base .. base + 15, job B base + 2 .. base + 17; B's workers start first and meet A's listener
    jobs = [(r, 3, base, 1000, 0xA1) for r in (0,)] + [(r, 3, base + 2, 2000, 0xB2) for r in (2, 1)]
    procs = [ctx.Process(target=_id_worker_job, args=j + (q,)) for j in jobs]
    for p in procs:
        p.start()
    time.sleep(1.0)

    def probe():                          # a scanner: connects to the first candidates and hangs up
        for k in range(4):
            try:
                socket.create_connection(("127.0.0.1", base + k), timeout=1.0).close()
            except OSError:
                pass
    th = threading.Thread(target=probe)
    th.start()
    rest = [(r, 3, base, 1000, 0xA1) for r in (1, 2)] + [(0, 3, base + 2, 2000, 0xB2)]
    procs2 = [ctx.Process(target=_id_worker_job, args=j + (q,)) for j in rest]
    for p in procs2:
        p.start()
    res = [q.get(timeout=120) for _ in procs + procs2]
    th.join(timeout=10)
    for p in procs + procs2:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(res) == 6
    for fill, rank, got in res:
        assert got == bytes([fill]) * 128, f"job {fill:#x} rank {rank} got another job's id"


def test_bench_gpus_2_starts_two_ranks_itself():
    """`python bench.py --gpus 2` without a launcher: the parent spawns two fresh ranks (RANK / WORLD_SIZE / MASTER_* set), they meet
    over the id hand-out, rank 0's line is relayed and the exit code is 0; a WORLD_SIZE that contradicts --gpus is refused and a
    failing rank makes the whole command fail without a result line.  LDSIM_BENCH_REHEARSAL stops the ranks
    before anything touches a GPU."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["LDSIM_BENCH_REHEARSAL"] = "1"
    bench = os.path.join(REPO, "bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "3", "--warmup", "1"], env=env, capture_output=True, timeout=180)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec == {"rehearsal": True, "n_gpus": 2, "world": 2, "steps": 3, "warmup": 1}
    # a launcher's WORLD_SIZE that contradicts --gpus: refused (the line would describe another job)
    r = subprocess.run([sys.executable, bench, "--gpus", "8"], env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, timeout=60)
    assert r.returncode != 0 and not r.stdout.strip() and b"WORLD_SIZE=1" in r.stderr
    # one rank dies (here: all of them, an unknown option value inside the children only): non-zero, no line
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--config", "module0"], env=dict(env, LDSIM_BENCH_REHEARSAL_FAIL_RANK="1"),
                       capture_output=True, timeout=180)
    assert r.returncode != 0 and not r.stdout.strip()
