#!/usr/bin/env python3
"""
bench.py -- throughput of the charge hot path (quench -> drift -> pixels -> induced current -> ADC), plus the light leg
(light incidence + photon sum) for the ndlar configuration (BASELINE.json configs[4]).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`; the launcher only
   starts the processes -- the collective layer is RCCL through the C-ABI, larndsim_amd/comm.py, no torch in this file.
   Started WITHOUT a launcher (no WORLD_SIZE in the environment) `--gpus N` starts its N ranks itself: the parent, which never
   touches the GPU, spawns N fresh children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relays rank 0's JSON line and
   exits non-zero if any child does.  A line is never printed for a run whose RCCL communicator holds another number of ranks
   than --gpus.)

A step = one pass of the whole path over this rank's resident synthetic segment set (BASELINE.json configs[1]: module0,
100k segments; the example edep-sim file is absent so the SURVEY 8d synthetic straight tracks are used).  Segments are
uploaded (H2D) before the timed region; a step re-unpacks the resident records, runs quench+drift and the fused chain
chunk by chunk; per-pixel ADC results stay in HBM.  N > 1 is weak scaling: every rank owns its own 100k-segment set of
events (batches are sharded by (event, TPC group), no data-path collective until the final all-gather-v of the compact
hit rows, which is inside the timed region).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(REPO, "larnd-sim_amd"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

SEGS_PER_GPU = 100_000
# segments per chain launch (whole batches).  Measured on module0 (gpurun_out sweep): 25 k 3.36, 50 k 3.51,
# 100 k 3.58, 200 k 3.67 x 10^6 segments/s -- fixed costs per launch (5 host round trips, sorts, tails of the big kernels)
CHUNK_SEGMENTS = int(os.environ.get("LDSIM_CHUNK_SEGMENTS", "100000"))
FP64_VALU_PEAK_TFLOPS = 78.6      # MI355X vector FP64 (spec)
HBM_PEAK_GBS = 8000.0             # MI355X HBM3E (spec), /opt/skills/guides/MI355X_MICROARCH.md
TRAFFIC_FILE = os.path.join(REPO, "profiles", "r04_traffic.json")   # written by tools/pmc_traffic.py from rocprofv3 --pmc passes


def host_cpu():
    """(physical cores, logical CPUs this process may use, model name) of the box, from /proc/cpuinfo and the affinity mask"""
    model, phys = "unknown", set()
    try:
        pid = cid = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                k, _, v = line.partition(":")
                k, v = k.strip(), v.strip()
                if k == "model name":
                    model = v
                elif k == "physical id":
                    pid = v
                elif k == "core id":
                    cid = v
                elif not k and pid is not None:
                    phys.add((pid, cid)); pid = cid = None
        if pid is not None:
            phys.add((pid, cid))
    except OSError:
        pass
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return (len(phys) or usable), usable, model


def cpu_baseline(response, n_seg=400):
    """The oracle (C port of the reference algorithm) on a bounded sample of the same workload, OpenMP over the (segment,
    pixel) pairs of tracks_current on every core this process may use (threads = physical cores, stated)."""
    from larndsim_amd import batching, consts, synth
    from oracle import oracle as O
    O.build()
    phys, usable, model = host_cpu()
    cores = max(1, min(phys, usable))
    seg = synth.make_segments(n_seg, seed=synth.SEED_BASE + 2, segs_per_event=n_seg)
    batching.swap_coordinates(seg)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    t0 = time.perf_counter()
    O.quench(seg, consts.physics.BIRKS)
    O.drift(seg)
    nmax = O.max_pixels(seg)
    r = int(np.ceil(seg["tran_diff"].max() * 5 / consts.detector.PIXEL_PITCH))
    P = (2 * r + 1) * nmax + (1 + 2 * r) * r * 2
    _, neigh, nrad, _ = O.get_pixels(seg, nmax, P, r)
    upix = O.unique_pixels(neigh)
    starts, T = O.time_intervals(seg)
    sig = O.tracks_current(seg, neigh, T, response)
    pim = O.pixel_index_map(neigh, upix)
    tpm = O.track_pixel_map(upix, neigh, nrad, int(nrad.max()) + 1, consts.sim.MAX_TRACKS_PER_PIXEL)
    ps, pts, _ = O.sum_pixel_signals(sig, starts, pim, tpm, len(upix))
    tt = np.linspace(0, consts.detector.TIME_INTERVAL[1], ps.shape[1] + 1)
    adc, _, _ = O.get_adc_values(ps, pts, tt, np.full(len(upix), consts.detector.DISCRIMINATION_THRESHOLD))
    O.digitize(adc)
    dt = time.perf_counter() - t0
    # quench + drift alone on the whole workload size ("quench+drift for all sizes"): the two HBM-bound stages of
    # the path on one host thread, median of five passes after a warm-up
    full = synth.make_segments(SEGS_PER_GPU, seed=synth.SEED_BASE + 2, segs_per_event=5000)
    batching.swap_coordinates(full)
    qd = []
    for i in range(6):
        work = full.copy()
        t1 = time.perf_counter()
        O.quench(work, consts.physics.BIRKS)
        O.drift(work)
        qd.append(time.perf_counter() - t1)
    qd_t = sorted(qd[1:])[2]
    return {"value": n_seg / dt, "unit": "segments/s", "cores": cores, "kind": "port",
            "physical_cores": phys, "usable_cpus": usable, "cpu_model": model,
            "sample": f"{n_seg} segments of the same synthetic set (charge chain), {dt:.1f} s; OpenMP ({cores} threads) over "
                      f"(segment,pixel) pairs in tracks_current, the other stages single-threaded",
            "quench_drift_only": {"value": SEGS_PER_GPU / qd_t, "unit": "segments/s", "cores": 1, "segments": SEGS_PER_GPU,
                                  "ms": 1e3 * qd_t, "note": "oracle quench + drift on all 100000 segments of the workload, one "
                                                            "thread, median of 5"}}


def profiled_traffic(config, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE,
    MI355X_MICROARCH.md, HBM), or (None, why) when this workload was not profiled."""
    try:
        with open(TRAFFIC_FILE) as f:
            tab = json.load(f)
        if kernel in tab[config]:
            e = tab[config][kernel]
            return float(e["bytes_per_launch"]), e.get("source", os.path.basename(TRAFFIC_FILE))
        # a kernel with further template arguments ("gcorr_kernel<1>" is launched as gcorr_kernel<1, false> for all pairs and as
        # gcorr_kernel<1, true> for the listed LDS classes): the chain launch's traffic is the sum over its variants
        parts = [v for k, v in tab[config].items() if kernel.endswith(">") and k.startswith(kernel[:-1] + ",")]
        if not parts:
            raise KeyError(kernel)
        return (float(sum(v["bytes_per_launch"] for v in parts)),
                parts[0].get("source", os.path.basename(TRAFFIC_FILE)) + f"; sum over {len(parts)} variants of {kernel}")
    except Exception:
        return None, f"no PMC pass for {config}/{kernel} in profiles/{os.path.basename(TRAFFIC_FILE)} (tools/pmc_traffic.py writes it)"


def self_launch(n):
    """--gpus N without a launcher: N fresh child processes, one rank each (never an exec of a process that has touched the
    GPU: this parent imports nothing that does).  Rank 0's stdout is relayed; the exit code is the worst child's."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:          # a free port for the ranks' rendezvous
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        for p in procs:
            if p.poll() is None:
                p.kill()
        sys.stderr.write(f"bench.py --gpus {n}: ranks failed (rank, exit code): {bad}; no result line\n")
        raise SystemExit(max(1, max(abs(c) for _, c in bad) & 0xFF))
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    raise SystemExit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--segments", type=int, default=None, help="segments per GPU (default 100000; 1000000 with --baseline-config 3..5)")
    ap.add_argument("--baseline-config", type=int, default=None, choices=[2, 3, 4, 5],
                    help="BASELINE.json configs[n-1] as SURVEY 8d spells it: 2 = module0 100k (the default contract line), "
                         "3 = 2x2 1M, 4 = 2x2 1M per GPU (8M on 8 GPUs; seed +4, 10000 segments per event), 5 = ndlar 1M + light")
    ap.add_argument("--response", default="survey", choices=["survey", "dense", "golden"])
    ap.add_argument("--config", default="module0", choices=["module0", "2x2_no_modvar", "ndlar"],
                    help="detector configuration of the synthetic workload (SURVEY 8d seeds); the contract line is the "
                         "default, module0 = BASELINE configs[1]")
    ap.add_argument("--light-stream", default="own", choices=["own", "main"],
                    help="photon sums without truth slots on a stream of their own beside the charge chain (own, default) or "
                         "in the ctx's stream ahead of it (main)")
    ap.add_argument("--light", default="auto", choices=["auto", "on", "off"],
                    help="time the light leg (incidence + per-batch photon sum) with the charge chain; auto = ndlar only "
                         "(BASELINE configs[4], synthetic light set-up of SURVEY 8d)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the dense-response and PCIe-inclusive side measurements")
    ap.add_argument("--fractions", type=int, default=1, help="compute backtracking fractions (reference always does)")
    ap.add_argument("--force-dist", action="store_true", help="run the RCCL all-gather path even with one rank")
    ap.add_argument("--weights-mode", type=int, default=2, choices=[0, 1, 2],
                    help="tracks_current kernels: 2 = node-separable form (gtables_kernel + gcorr_kernel, default), 1 = qweights_kernel "
                         "+ mac_shift kernels (round 2), 0 = weights_kernel + mac_shift")
    ap.add_argument("--trim-response-log", type=float, default=None,
                    help="response ticks below exp(-v) of the table's largest entry are not read (library default 23; 0 = exact zeros only)")
    a = ap.parse_args()

    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a.gpus)                      # (does not return)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: the line would not describe the job that ran")

    from larndsim_amd import batching, comm as lcomm, consts, dist as ldist, lib, synth
    if os.environ.get("LDSIM_BENCH_REHEARSAL"):
        # CPU rehearsal of the launch path (tests/test_cpu_dist.py): the ranks meet over the ncclUniqueId hand-out and stop
        # before anything touches a GPU
        got = lcomm.exchange_id(bytes(range(128)) if rank == 0 else b"", rank, world, timeout=60.0)
        if os.environ.get("LDSIM_BENCH_REHEARSAL_FAIL_RANK") == str(rank):
            raise SystemExit(f"rank {rank}: asked to fail (test of the launcher's error path)")
        if got != bytes(range(128)):
            raise SystemExit(f"rank {rank}: rendezvous payload differs")
        if rank == 0:
            print(json.dumps({"rehearsal": True, "n_gpus": a.gpus, "world": world, "steps": a.steps, "warmup": a.warmup}))
        return
    from larndsim_amd.chain import ChargeChain

    use_dist = world > 1 or a.force_dist
    result_fd = None
    if use_dist:
        # RCCL may print to stdout when the communicator is created.  The result stream must carry exactly one JSON
        # line, so fd 1 points at stderr for the whole process and the line is written to a duplicate of the original.
        sys.stdout.flush()
        result_fd = os.dup(1)
        os.dup2(2, 1)

    if a.baseline_config is not None:
        a.config = {2: "module0", 3: "2x2_no_modvar", 4: "2x2_no_modvar", 5: "ndlar"}[a.baseline_config]
        if a.segments is None:
            a.segments = SEGS_PER_GPU if a.baseline_config == 2 else 1_000_000
    if a.segments is None:
        a.segments = SEGS_PER_GPU
    consts.load_snapshot(a.config)
    for k in ("RESET_NOISE_CHARGE", "UNCORRELATED_NOISE_CHARGE", "DISCRIMINATOR_NOISE"):
        setattr(consts.detector, k, 0)     # noise off: the contract workload is the deterministic chain
    # BASELINE config number -> seed and event size.  The 2x2 geometry on more than one GPU is config 4 (seed +4,
    # 10000 segments per event so that a (event, TPC group) batch is a useful unit of sharding), on one GPU config 3.
    seed_index = a.baseline_config or {"module0": 2, "2x2_no_modvar": 4 if world > 1 else 3, "ndlar": 5}[a.config]
    segs_per_event = 10000 if seed_index == 4 else 5000
    light_on = a.light == "on" or (a.light == "auto" and a.config == "ndlar")
    lut = None
    if light_on:
        if not consts.light.LIGHT_SIMULATED or consts.light.N_OP_CHANNEL == 0:
            synth.set_synthetic_light(48)                       # ndlar ships no light configuration
        lut = synth.make_lut((14, 26, 8), 48, 100, synth.SEED_BASE + seed_index)

    # ---- workload: the global set of world x `segments` segments, this rank's shard of its (event, TPC group) batches.  Up to
    # 2M segments every rank builds the global set and takes its shard (dist.shard_segments: balanced by segment count,
    # tests/test_cpu_dist.py).  Beyond that (config 4: 8M) building it eight times over would cost minutes before the timed
    # region: rank r then builds its own `segments` segments as events [r E, (r+1) E) with its own seed -- the same kind of
    # shard (whole events, hence whole batches), decided before instead of after the generation.
    n_total = a.segments * world
    own_events = world > 1 and n_total > 2_000_000

    def undo_spill_offset(s):              # the driver subtracts the spill offset again (cli/simulate_pixels.py:574-582)
        if consts.sim.IS_SPILL_SIM:
            loc = s["event_id"] % consts.sim.MAX_EVENTS_PER_FILE
            for f in ("t0", "t0_start", "t0_end"):
                s[f] = s[f] - loc * consts.sim.SPILL_PERIOD
    if own_events:
        ev_per_rank = -(-a.segments // segs_per_event)
        seg_all = synth.make_segments(a.segments, seed=synth.SEED_BASE + seed_index + 7919 * rank, segs_per_event=segs_per_event,
                                      spill=bool(consts.sim.IS_SPILL_SIM), event_id0=rank * ev_per_rank)
        undo_spill_offset(seg_all)
        batching.swap_coordinates(seg_all)
        bid_all, order, table = batching.assign_batches(seg_all)
        idx, bid = ldist.shard_segments(bid_all, order, table, 0, 1)
    else:
        seg_all = synth.make_segments(n_total, seed=synth.SEED_BASE + seed_index, segs_per_event=segs_per_event,
                                      spill=bool(consts.sim.IS_SPILL_SIM))
        undo_spill_offset(seg_all)
        batching.swap_coordinates(seg_all)
        bid_all, order, table = batching.assign_batches(seg_all)
        idx, bid = ldist.shard_segments(bid_all, order, table, rank, world)
    seg = np.ascontiguousarray(seg_all[idx])
    del seg_all
    response = synth.make_response(a.response)

    ch = ChargeChain(response, device=local_rank)
    lib.set_option("weights_mode", a.weights_mode, ch.ctx)
    if a.trim_response_log is not None:
        lib.set_option("trim_response_log", a.trim_response_log, ch.ctx)
    light_own_stream = bool(light_on and consts.sim.MAX_MC_TRUTH_IDS == 0 and a.light_stream != "main")
    if light_own_stream:
        lib.set_option("light_sum_async", 1, ch.ctx)
    cm = lcomm.Communicator(ch.ctx, rank, world) if use_dist else None
    t_up = time.perf_counter()
    ch.upload(seg, bid)                      # H2D happens here, outside the timed region
    ch.synchronize()
    t_up = time.perf_counter() - t_up
    ranges = batching.chunk_ranges(bid, CHUNK_SEGMENTS)
    n_sim = int((bid >= 0).sum())
    bedges = np.flatnonzero(np.r_[True, bid[1:n_sim] != bid[:n_sim - 1], True]) if n_sim else np.array([0])
    op_all = consts.light.TPC_TO_OP_CHANNEL[:].ravel().astype(np.int32) if light_on else None

    def new_acc():
        return {"cur_ms": 0.0, "w_ms": 0.0, "m_ms": 0.0, "f_ms": 0.0, "samples": 0, "adc_ms": 0.0, "bytes": 0.0, "dfma": 0, "dfma_useful": 0,
                "S": 0, "U": 0, "pairs": 0, "launches": 0, "hits": 0, "ambig": 0, "ovf": 0, "inc_ms": 0.0, "inc_n": 0,
                "sum_ms": 0.0, "sum_n": 0, "photons": 0.0}
    acc = new_acc()
    gathered = {"rows": 0, "per_rank": []}

    def step(record, download=False):
        ch.reset()
        ch.quench_drift()
        if light_on:
            ch.light_incidence(lut)
            if record:
                acc["inc_ms"] += ch.light_kernel_ms()["incidence_ms"]; acc["inc_n"] += 1
            # one photon sum per (event, TPC group) batch, like the driver.  Without truth slots a sum returns with its kernels
            # in flight, and on their own stream (option light_sum_async) they run beside the charge chain of the same segments:
            # what is timed here is the host issuing them; their duration on the GPU is tools/light_sum_loop.py's figure.
            # With truth slots (2x2) every sum ends in a host read-back: the loop's wall time is the sums' time.
            if record:
                if not light_own_stream:
                    ch.synchronize()
                t_sum = time.perf_counter()
            for b, e in zip(bedges[:-1], bedges[1:]):
                ch.sum_light(int(b), int(e), op_all)
            if record:
                if not light_own_stream:
                    ch.synchronize()
                acc["sum_ms"] += 1e3 * (time.perf_counter() - t_sum); acc["sum_n"] += len(bedges) - 1
                # (the photon sums stay in HBM in the PCIe-inclusive passes too: their consumers -- scintillation, SiPM response,
                # triggers -- run on the device, in the reference as here; what reaches the host is the digitised trigger windows)
        for i, (b, e) in enumerate(ranges):
            st = ch.run(b, e, want_fractions=bool(a.fractions))
            if record:
                ms = ch.kernel_ms()
                acc["cur_ms"] += ms["current_ms"]; acc["adc_ms"] += ms["adc_ms"]
                acc["w_ms"] += ms["weights_ms"]; acc["m_ms"] += ms["mac_ms"]; acc["f_ms"] += ms["fallback_ms"]
                acc["samples"] += st.n_samples
                acc["bytes"] += 184.0 * st.n_segments + 484.0 * st.n_unique     # SURVEY 8d B_alg
                acc["dfma"] += st.n_dfma; acc["dfma_useful"] += st.n_dfma_useful; acc["S"] += st.n_segments; acc["U"] += st.n_unique
                acc["pairs"] += st.n_pairs; acc["launches"] += 1; acc["ambig"] += st.n_ambiguous
                acc["ovf"] += st.n_overflow
                acc["hits"] += ch.compact_hits()[1]
            if cm is not None:
                cm.accumulate_hits(reset=(i == 0))
            if download == "compact":            # hit pixels, hits and per-hit fractions only, gathered on the device first
                ch.download_compact()
            elif download == "overlapped":       # this chunk's rows cross PCIe while the next chunk computes
                ch.download_async()
            elif download:
                ch.download(pinned=True)
        if download == "overlapped":
            ch.wait_download()
        if cm is not None:
            total, counts = cm.allgather_hits()
            gathered["rows"], gathered["per_rank"] = total, counts
            return total
        return 0

    def barrier():
        ch.synchronize()
        if cm is not None:
            cm.barrier()

    for _ in range(a.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    n_job = float(len(seg))
    n_ranks_rccl = None
    if cm is not None:
        elapsed = cm.allreduce(elapsed, "max")
        n_job = cm.allreduce(n_job, "sum")
        n_ranks_rccl = cm.count()[0]                 # ncclCommCount: the ranks that really joined
        if n_ranks_rccl != a.gpus:
            raise SystemExit(f"--gpus {a.gpus} but the RCCL communicator holds {n_ranks_rccl} ranks: no result line")

    # ---- side measurements (one GPU only; never part of `value`) ------------------------------------------------------------
    extras = {}
    if rank == 0 and world == 1 and not a.no_extras:
        main_acc, acc = acc, new_acc()
        # (a) host buffers in, host results out: H2D of the records + the step + D2H of every per-pixel array
        def timed_pass(download):            # median of three passes: upload + step + downloads (one pass alone varies by 2x)
            ts = []
            for _ in range(3):
                ch.synchronize()
                t1 = time.perf_counter()
                ch.upload(seg, bid)
                step(False, download=download)
                ch.synchronize()
                ts.append(time.perf_counter() - t1)
            return sorted(ts)[1]
        step(False, download=True)           # first pass sizes the chain's page-locked download buffers
        t_incl = timed_pass(True)
        extras["pcie_inclusive"] = {"value": len(seg) / t_incl, "unit": "segments/s", "ms_per_step": 1e3 * t_incl,
                                    "h2d_ms": 1e3 * t_up,
                                    "note": "median of three passes incl. H2D of the 152-byte records and D2H of unique_pix / adc_list / "
                                            "adc_ticks / adc_digit / track_pixel_map / current_fractions"
                                            + " per chunk (records from pageable memory, results into page-locked buffers)"}
        # (a') the same with every chunk's D2H on the copy stream beside the next chunk's kernels (ldsim_chain_download_async)
        for _ in range(2 if len(ranges) == 1 else 1):      # size both alternating sets of output / host buffers
            step(False, download="overlapped")
        t_ovl = timed_pass("overlapped")
        extras["pcie_inclusive"]["overlapped_value"] = len(seg) / t_ovl
        extras["pcie_inclusive"]["overlapped_ms_per_step"] = 1e3 * t_ovl
        # (a'') the results in compact form: what the exporter reads (hit pixels, hits, fractions of the track slots a pixel has)
        step(False, download="compact")
        t_cpt = timed_pass("compact")
        extras["pcie_inclusive"]["compact_value"] = len(seg) / t_cpt
        extras["pcie_inclusive"]["compact_ms_per_step"] = 1e3 * t_cpt
        extras["pcie_inclusive"]["compact_note"] = ("H2D of the records + the step + ldsim_chain_compact_build / _download per chunk: "
                                                    "hit pixels, 24-byte hit rows, charges, track slots and per-hit backtracking "
                                                    "fractions (a few MB instead of 13 KB per unique pixel)")
        # (c) the same workload with only exactly-zero response ticks skipped (the library default also skips ticks below 1e-10
        # of the table's largest entry: DESIGN.md section 4, "trim_response_log")
        if a.trim_response_log is None:
            lib.set_option("trim_response_log", 0.0, ch.ctx)
            step(False)
            ch.synchronize()
            t2 = time.perf_counter()
            nd = max(1, min(3, a.steps))
            for _ in range(nd):
                step(False)
            ch.synchronize()
            t_exact = (time.perf_counter() - t2) / nd
            extras["exact_zero_trim"] = {"value": len(seg) / t_exact, "unit": "segments/s", "ms_per_step": 1e3 * t_exact, "steps": nd,
                                         "note": "trim_response_log 0: every response tick that is not exactly 0.0 is read "
                                                 "(the survey table's Gaussian tails down to 1e-308)"}
            lib.set_option("trim_response_log", 23.0, ch.ctx)
        # (b) the same workload on a response table without exact zeros (real response files are dense)
        if a.response != "dense":
            lib.set_response(synth.make_response("dense"), ch.ctx)
            step(False)
            ch.synchronize()
            t2 = time.perf_counter()
            nd = max(1, min(3, a.steps))
            for _ in range(nd):
                step(True)
            ch.synchronize()
            t_dense = (time.perf_counter() - t2) / nd
            extras["dense_response_value"] = len(seg) / t_dense
            nl_d = max(acc["launches"], 1)
            useful_d = acc["dfma_useful"] if acc["dfma_useful"] > 0 else acc["dfma"]
            # the path a full-support response table takes: since round 4 the default (weights_mode 2) keeps it in the matrix form
            # (gform_max_support unlimited: profiles/r04_dense_handover.log); weights_mode 1 / 0 run the shifted-window kernels
            Md = int(round(consts.detector.TIME_SAMPLING / consts.detector.RESPONSE_SAMPLING))
            dense_gform = a.weights_mode == 2
            dk = f"gcorr_kernel<{Md}>" if dense_gform else ("mac_shift_kernel" if Md == 1 else "mac_shift2_kernel")
            dwk = f"gtables_wave_kernel<{Md}, 55>" if dense_gform else "qweights_kernel"
            tr_d, tr_src = profiled_traffic(a.config + "_dense", dk)
            extras["roofline_dense"] = {
                "bound": "mfma" if dense_gform else "valu_f64", "kernel": dk,
                "launch_ms_avg": acc["m_ms"] / nl_d, "weights_kernel": dwk, "weights_kernel_ms_avg": acc["w_ms"] / nl_d,
                "achieved": (2.0 * useful_d / (acc["m_ms"] * 1e-3) / 1e12) if acc["m_ms"] > 0 else None, "peak": FP64_VALU_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": (2.0 * useful_d / (acc["m_ms"] * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS) if acc["m_ms"] > 0 else None,
                "issued_frac": (2.0 * acc["dfma"] / (acc["m_ms"] * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS) if acc["m_ms"] > 0 else None,
                "traffic": tr_d, "traffic_source": tr_src,
                "note": ("algorithmic flops = 2 x (nodes x cells x response ticks + the Toeplitz sum) over the correlation kernel's time against "
                         "the f64 matrix peak (the same 78.6 TFLOP/s as the DFMA peak); issued = 2 x 1024 per v_mfma_f64_16x16x4"
                         if dense_gform else
                         "algorithmic flops = 2 x (weights kept) x (window ticks) over the correlation kernel's time against the f64 DFMA peak; "
                         "issued counts every FMA lane (8-shift block and 512-tick tile padding)")}
            extras["dense_response"] = {"value": len(seg) / t_dense, "unit": "segments/s", "ms_per_step": 1e3 * t_dense,
                                        "steps": nd, "mac_kernel_ms_avg": acc["m_ms"] / max(acc["launches"], 1),
                                        "dfma_per_segment": acc["dfma"] / max(acc["S"], 1),
                                        "kernel": dk,
                                        "valu_f64_frac": (2.0 * acc["dfma"] / (acc["m_ms"] * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS)
                                        if acc["m_ms"] > 0 else None,
                                        "useful_share_of_issued": (acc["dfma_useful"] / acc["dfma"]) if acc["dfma"] else None,
                                        "note": "synthetic 'dense' response (no exact zeros, so trim_response skips nothing)"}
            lib.set_response(response, ch.ctx)
        acc = main_acc

    if rank == 0:
        ms_step = 1e3 * elapsed / a.steps
        value = n_job * a.steps / elapsed
        split = acc["w_ms"] > 0
        M = int(round(consts.detector.TIME_SAMPLING / consts.detector.RESPONSE_SAMPLING))
        gform = a.weights_mode == 2
        # csrc/kernels_gcorr.hip (node-separable form) or csrc/kernels_macshift.hip
        mac_name = f"gcorr_kernel<{M}>" if gform else ("mac_shift_kernel" if M == 1 else "mac_shift2_kernel")
        w_name = f"gtables_wave_kernel<{M}, 55>" if gform else (f"qweights_kernel<{M}>" if a.weights_mode == 1 else f"weights_kernel<{M}>")
        # dominant kernel of the path: the longer of the split path's two kernels, or the monolithic current_kernel
        if not split:
            dom_name, dom_ms = f"current_kernel<{M}>", acc["cur_ms"]
        elif acc["w_ms"] >= acc["m_ms"]:
            dom_name, dom_ms = w_name, acc["w_ms"]
        else:
            dom_name, dom_ms = mac_name, acc["m_ms"]
        dom_s = dom_ms * 1e-3
        achieved = acc["bytes"] / dom_s / 1e9 if dom_s > 0 else 0.0
        mac_s = (acc["m_ms"] if split else acc["cur_ms"]) * 1e-3
        tflops = 2.0 * acc["dfma"] / mac_s / 1e12 if mac_s > 0 else 0.0
        nl = max(acc["launches"], 1)
        default_workload = (a.segments == SEGS_PER_GPU and a.response == "survey" and a.fractions and split and
                            seed_index == {"module0": 2, "2x2_no_modvar": 3, "ndlar": 5}.get(a.config))
        traffic, traffic_src = profiled_traffic(a.config, dom_name) if default_workload else (None, "not the profiled workload")
        what = "charge chain quench->drift->pixels->tracks_current->pixel sum->ADC+digitize"
        if light_on:
            what += " + light incidence (all segments) + photon sum per (event, TPC group) batch"
            if light_own_stream:
                what += " (sums on their own stream beside the charge chain)"
        out = {
            "metric": f"edep segments/s end-to-end (quench->ADC), {a.config} config",
            "value": value, "unit": "segments/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{a.config}, {a.segments} synthetic straight-track segments per GPU "
                                   f"(seed {synth.SEED_BASE + seed_index}, {segs_per_event}/event), {what}, backtracking fractions "
                                   f"{'on' if a.fractions else 'off'}",
                       "response": f"synthetic '{a.response}' (45,45,1950) f64", "noise": "off",
                       "trim_response_log": 23.0 if a.trim_response_log is None else a.trim_response_log,
                       "quad_accuracy_log10": 7,
                       "approximations": "response ticks below exp(-trim_response_log) = 1e-10 of the table's largest entry are not read "
                                         "(a waveform moves by < 1e-10 of its peak; `exact_zero_trim` is the figure without it); quadrature "
                                         "along the segment for 1e-7 of the peak weight (0.012 of the parity tolerance on the reference's "
                                         "goldens: profiles/r04_acc_sweep_module0.log)",
                       "light": (f"synthetic LUT (14,26,8) x 48 ch/TPC x 100 bins, {int(consts.light.N_OP_CHANNEL)} channels, "
                                 f"{len(bedges) - 1} photon sums per step") if light_on else "off",
                       "segments_per_step": int(n_job), "pairs_per_segment": acc["pairs"] / max(acc["S"], 1),
                       "unique_pixels_per_segment": acc["U"] / max(acc["S"], 1),
                       "hits_per_step": acc["hits"] // max(a.steps, 1),
                       "chunk_segments": CHUNK_SEGMENTS, "parallelism": f"batch-sharded x{world}",
                       "n_ranks": n_ranks_rccl if n_ranks_rccl is not None else 1,     # as RCCL reports (ncclCommCount)
                       "hit_rows_per_rank": gathered["per_rank"] if cm is not None else [acc["hits"] // max(a.steps, 1)],
                       "hit_rows_gathered": gathered["rows"] if cm is not None else acc["hits"] // max(a.steps, 1),
                       "baseline_config": seed_index, "segments_per_event": segs_per_event,
                       "sharding": ("every rank builds its own events" if own_events else "global set, dist.shard_segments"),
                       "collective": "RCCL all-gather (counts) + all-gather-v (24-byte hit rows), C-ABI ldsim_comm_*"
                                     if cm is not None else "none (one rank)"},
            "roofline": {"bound": "hbm", "kernel": dom_name,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "launch_ms_avg": dom_ms / nl,
                         "launches": acc["launches"],
                         "algorithmic_bytes_per_launch": acc["bytes"] / nl,
                         "note": "f64-VALU-bound path: HBM fraction is tiny by construction (SURVEY 8d: ~2 KB of "
                                 "compulsory traffic per segment); see valu_f64 and stage_kernels",
                         "stage_kernels": {"weights_kernel": w_name, "mac_kernel": mac_name,
                                           "weights_kernel_ms_avg": acc["w_ms"] / nl, "mac_kernel_ms_avg": acc["m_ms"] / nl,
                                           "current_kernel_ms_avg": acc["f_ms"] / nl,
                                           "pixel_adc_kernel_ms_avg": acc["adc_ms"] / nl,
                                           "quadrature_nodes_per_pair": acc["samples"] / max(acc["pairs"], 1)},
                         "valu_f64": {"kernel": mac_name if split else f"current_kernel<{M}>",
                                      "achieved": tflops, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                      "frac": tflops / FP64_VALU_PEAK_TFLOPS,
                                      "dfma_per_segment": acc["dfma"] / max(acc["S"], 1),
                                      # the FMAs that are neither zero padding of an 8-shift weight block nor ticks outside the
                                      # pair's window (LdsimChainStats.n_dfma_useful)
                                      "useful_share_of_issued": (acc["dfma_useful"] / acc["dfma"]) if acc["dfma"] else None,
                                      "frac_useful": (tflops / FP64_VALU_PEAK_TFLOPS * acc["dfma_useful"] / acc["dfma"])
                                      if acc["dfma"] else None}},
        }
        # The dominant kernel is an f64 correlation that never comes near the HBM roof ("bound by FP64 VALU ...
        # report both"): when it dominates, the top-level roofline is the compute one -- ALGORITHMIC flops (2 x the FMAs that
        # are neither block nor tile padding, LdsimChainStats.n_dfma_useful) over the kernel's time against the dense f64 peak
        # (78.6 TFLOP/s, the same for v_fma_f64 and v_mfma_f64 on this part) -- with the issued-FMA figure and the HBM view beside it.
        if split and dom_name == mac_name and acc["dfma"] > 0 and mac_s > 0:
            hbm_view = {k: out["roofline"][k] for k in ("achieved", "peak", "unit", "frac", "algorithmic_bytes_per_launch")}
            hbm_view["note"] = "SURVEY 8d algorithmic bytes (184 B per segment + 484 B per unique pixel) over the kernel's time: tiny by construction"
            useful = acc["dfma_useful"] if acc["dfma_useful"] > 0 else acc["dfma"]
            tf_alg = 2.0 * useful / mac_s / 1e12
            out["roofline"].update({
                "bound": "mfma" if gform else "valu_f64", "achieved": tf_alg, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": tf_alg / FP64_VALU_PEAK_TFLOPS,
                "algorithmic_flops_per_launch": 2.0 * useful / nl,
                "issued": {"achieved": tflops, "frac": tflops / FP64_VALU_PEAK_TFLOPS,
                           "note": "every FMA lane the kernel issues (padding included)"},
                "hbm": hbm_view,
                "note": ("f64 matrix-pipe kernel (v_mfma_f64_16x16x4, dense peak 78.6 TFLOP/s): algorithmic flops = 2 x nodes x cells x "
                         "response ticks + the Toeplitz sum, issued = 2 x 1024 per MFMA (16-row / 4-cell / 16-tick padding included)"
                         if gform else
                         "f64-VALU-bound kernel (no MFMA in it: the f64 matrix and vector pipes have the same peak here and are not "
                         "additive)") + "; `traffic` = HBM bytes of this kernel per launch from the PMC passes"})
        if light_on and acc["inc_n"]:
            n_op = int(consts.light.N_OP_CHANNEL)
            inc_ms = acc["inc_ms"] / acc["inc_n"]
            # bytes the stage has to move in this library's layout: 44 B of SoA columns read per segment, n_photons_det
            # (+ t0_det in trigger mode 0) f4 per (segment, channel) and the voxel written.  SURVEY 8d counts the reference's
            # record (152 B) and {segment_id, n_photons_det, t0_det} per channel; that figure is kept beside it.
            per_ch = 8.0 if consts.light.LIGHT_TRIG_MODE == 0 else 4.0
            b_alg = len(seg) * (44.0 + 12.0 + per_ch * n_op)
            b_survey = len(seg) * (152.0 + 12.0 * n_op)
            out["roofline_light_incidence"] = {
                "bound": "hbm", "kernel": "light_incidence4_kernel", "achieved": b_alg / (inc_ms * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": b_alg / (inc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "launch_ms_avg": inc_ms, "algorithmic_bytes_per_launch": b_alg,
                "survey_8d_bytes_per_launch": b_survey,
                "traffic": None, "traffic_source": "no PMC pass",
                "note": "streaming write of the dense [segment][channel] f4 array(s); torch.zero_() of the same size runs at "
                        "5.0-6.9 TB/s on this GPU (tools/fill_bw.py)",
                "photon_sum_ms_avg": acc["sum_ms"] / max(acc["sum_n"], 1), "photon_sums": acc["sum_n"] // max(a.steps, 1),
                "photon_sum_note": ("host time to issue a step's photon sums / batches: they run on their own stream beside the "
                                    "charge chain (option light_sum_async) and end inside the step's timed region"
                                    if light_own_stream else
                                    "wall time of a step's photon-sum loop over its batches (host calls included) / batches")}
        out.update(extras)
        if world > 1:
            # timed on rank 0 of the single-GPU run only: at N > 1 it would keep the other ranks waiting at the barrier
            out["cpu_baseline"] = {"value": None, "unit": "segments/s", "cores": 0, "kind": "port",
                                   "sample": "timed in the 1-GPU run only"}
        elif not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(response)
            except Exception as e:      # the baseline is reported, never required for the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "segments/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {e}"}
        out["ambiguous_slices"] = acc["ambig"]
        out["overflow_pixels"] = acc["ovf"]
        line = json.dumps(out) + "\n"
        if result_fd is None:
            sys.stdout.write(line)
            sys.stdout.flush()
        else:
            os.write(result_fd, line.encode())
    if cm is not None:
        cm.barrier()
        cm.destroy()


if __name__ == "__main__":
    main()
