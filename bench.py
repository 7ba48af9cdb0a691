#!/usr/bin/env python3
"""
bench.py -- throughput of the charge hot path (quench -> drift -> pixels -> induced current -> ADC).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

A step = one pass of the whole path over this rank's resident synthetic segment set
(BASELINE.json configs[1]: module0, 100k segments; the example edep-sim file is absent so the
SURVEY §8d synthetic straight tracks are used).  Segments are uploaded (H2D) before the timed
region; a step re-unpacks the resident records, runs quench+drift and the fused chain chunk by
chunk; per-pixel ADC results stay in HBM.  N > 1 is weak scaling: every rank owns its own
100k-segment set of events (batches are sharded by (event, TPC group), no data-path collective
until the final all-gather of the compact hit rows, which is inside the timed region).
Prints ONE JSON line on rank 0.
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
CHUNK_SEGMENTS = 50_000
FP64_VALU_PEAK_TFLOPS = 78.6      # MI355X vector FP64 (spec)
HBM_PEAK_GBS = 8000.0             # MI355X HBM3E (spec), /opt/skills/guides/MI355X_MICROARCH.md
# weights_kernel<1>, one launch = 50k segments (profiles/r01_split_pmc_*.csv): FETCH_SIZE 196.5 MB (x2 on gfx950 = 393 MB)
# + WRITE_SIZE 21.80 GB; the writes are the per-pair weight blocks and item lists handed to mac_kernel (intermediates the
# algorithmic count excludes) plus ~5 GB of register-spill scratch written back (the kernel is built for 4 workgroups
# per CU: 128 VGPRs, 52 B/lane of scratch).  mac_kernel reads the blocks back: FETCH 8.99 GB x2, WRITE 5.06 GB.
PROFILED_TRAFFIC_BYTES = 22.19e9


def cpu_baseline(response, n_seg=400):
    """The oracle (C port of the reference algorithm) on a bounded sample of the same workload: 400 segments are
    about 12 s on the GPU box's 16 host threads (the rate does not depend on the sample size: 48 segments give the same)."""
    from larndsim_amd import batching, consts, synth
    from oracle import oracle as O
    O.build()
    seg = synth.make_segments(n_seg, seed=synth.SEED_BASE + 2, segs_per_event=n_seg)
    batching.swap_coordinates(seg)
    cores = min(os.cpu_count() or 1, 16)
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
    return {"value": n_seg / dt, "unit": "segments/s", "cores": cores, "kind": "port",
            "sample": f"{n_seg} segments of the same synthetic module0 set, full chain, {dt:.1f} s; OpenMP over "
                      f"(segment,pixel) pairs in tracks_current only"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--segments", type=int, default=SEGS_PER_GPU)
    ap.add_argument("--response", default="survey", choices=["survey", "dense", "golden"])
    ap.add_argument("--config", default="module0", choices=["module0", "2x2_no_modvar", "ndlar"],
                    help="detector configuration of the synthetic workload (SURVEY 8d seeds); the contract line is the "
                         "default, module0 = BASELINE configs[1]")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fractions", type=int, default=1, help="compute backtracking fractions (reference always does)")
    ap.add_argument("--force-dist", action="store_true", help="run the torch.distributed / all-gather path even with one rank")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")

    from larndsim_amd import batching, consts, dist as ldist, lib, synth
    from larndsim_amd.chain import ChargeChain

    tdist = None
    torch = None
    use_dist = world > 1 or a.force_dist
    result_fd = None
    if use_dist:
        # RCCL prints a version banner to stdout when the communicator is created (and may print again on teardown).
        # The result stream must carry exactly one JSON line, so for the whole process fd 1 points at stderr and the line
        # is written to a saved duplicate of the original stdout.
        sys.stdout.flush()
        result_fd = os.dup(1)
        os.dup2(2, 1)
        import datetime
        import torch
        import torch.distributed as tdist
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        tdist.init_process_group(backend="nccl", timeout=datetime.timedelta(seconds=300), rank=rank, world_size=world,
                                 device_id=torch.device("cuda", local_rank))

    consts.load_snapshot(a.config)
    for k in ("RESET_NOISE_CHARGE", "UNCORRELATED_NOISE_CHARGE", "DISCRIMINATOR_NOISE"):
        setattr(consts.detector, k, 0)     # the reference's RNG stream is third-party/unpinned: noise off
    seed_index = {"module0": 2, "2x2_no_modvar": 3, "ndlar": 5}[a.config]      # BASELINE config number

    # ---- workload: global set of world x 100k segments, this rank's shard of its batches ----------------------------
    n_total = a.segments * world
    seg_all = synth.make_segments(n_total, seed=synth.SEED_BASE + seed_index, segs_per_event=5000,
                                  spill=bool(consts.sim.IS_SPILL_SIM))
    if consts.sim.IS_SPILL_SIM:            # the driver subtracts the spill offset again (cli/simulate_pixels.py:574-582)
        loc = seg_all["event_id"] % consts.sim.MAX_EVENTS_PER_FILE
        for f in ("t0", "t0_start", "t0_end"):
            seg_all[f] = seg_all[f] - loc * consts.sim.SPILL_PERIOD
    batching.swap_coordinates(seg_all)
    bid_all, order, table = batching.assign_batches(seg_all)
    idx, bid = ldist.shard_segments(bid_all, order, table, rank, world)
    seg = np.ascontiguousarray(seg_all[idx])
    del seg_all
    response = synth.make_response(a.response)

    ch = ChargeChain(response, device=local_rank)
    ch.upload(seg, bid)                      # H2D happens here, outside the timed region
    ranges = batching.chunk_ranges(bid, CHUNK_SEGMENTS)

    acc = {"cur_ms": 0.0, "w_ms": 0.0, "m_ms": 0.0, "f_ms": 0.0, "samples": 0, "adc_ms": 0.0, "bytes": 0.0, "dfma": 0, "S": 0, "U": 0, "pairs": 0, "launches": 0,
           "hits": 0, "ambig": 0, "ovf": 0}

    def step(record):
        ch.reset()
        ch.quench_drift()
        rows_all = []
        for (b, e) in ranges:
            st = ch.run(b, e, want_fractions=bool(a.fractions))
            if record:
                ms = ch.kernel_ms()
                acc["cur_ms"] += ms["current_ms"]; acc["adc_ms"] += ms["adc_ms"]
                acc["w_ms"] += ms["weights_ms"]; acc["m_ms"] += ms["mac_ms"]; acc["f_ms"] += ms["fallback_ms"]
                acc["samples"] += st.n_samples
                acc["bytes"] += 184.0 * st.n_segments + 484.0 * st.n_unique     # SURVEY §8d B_alg
                acc["dfma"] += st.n_dfma; acc["S"] += st.n_segments; acc["U"] += st.n_unique
                acc["pairs"] += st.n_pairs; acc["launches"] += 1; acc["ambig"] += st.n_ambiguous
                acc["ovf"] += st.n_overflow
            if use_dist:
                p, n, rb = ch.compact_hits()
                rows_all.append(ldist.device_rows_as_tensor(p, n, rb, torch.device("cuda", local_rank)).clone())
            if record:
                acc["hits"] += ch.compact_hits()[1]
        if use_dist:
            rows = torch.cat(rows_all, dim=0)
            gathered, _ = ldist.allgather_rows(rows)
            torch.cuda.synchronize()
            return gathered.shape[0]
        return 0

    def barrier():
        ch.synchronize()
        if use_dist:
            torch.cuda.synchronize()
            tdist.barrier()

    for _ in range(a.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=torch.device("cuda", local_rank))
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([float(len(seg))], dtype=torch.float64, device=torch.device("cuda", local_rank))
        tdist.all_reduce(tot, op=tdist.ReduceOp.SUM)
        n_job = float(tot.item())
    else:
        n_job = float(len(seg))

    if rank == 0:
        ms_step = 1e3 * elapsed / a.steps
        value = n_job * a.steps / elapsed
        split = acc["w_ms"] > 0
        M = int(round(consts.detector.TIME_SAMPLING / consts.detector.RESPONSE_SAMPLING))
        # dominant kernel of the path: the longer of the split path's two kernels (weights_kernel on module0 / 2x2,
        # mac_kernel on ndlar), or the monolithic current_kernel when the split path is off
        if not split:
            dom_name, dom_ms = f"current_kernel<{M}>", acc["cur_ms"]
        elif acc["w_ms"] >= acc["m_ms"]:
            dom_name, dom_ms = f"weights_kernel<{M}>", acc["w_ms"]
        else:
            dom_name, dom_ms = f"mac_kernel<{M}>", acc["m_ms"]
        dom_s = dom_ms * 1e-3
        achieved = acc["bytes"] / dom_s / 1e9 if dom_s > 0 else 0.0
        mac_s = (acc["m_ms"] if split else acc["cur_ms"]) * 1e-3
        tflops = 2.0 * acc["dfma"] / mac_s / 1e12 if mac_s > 0 else 0.0
        nl = max(acc["launches"], 1)
        out = {
            "metric": f"edep segments/s end-to-end (quench->ADC), {a.config} config",
            "value": value, "unit": "segments/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{a.config}, {a.segments} synthetic straight-track segments per GPU "
                                   f"(seed {synth.SEED_BASE + seed_index}, 5000/event), full charge chain quench->drift->pixels->"
                                   f"tracks_current->pixel sum->ADC+digitize, backtracking fractions "
                                   f"{'on' if a.fractions else 'off'}",
                       "response": f"synthetic '{a.response}' (45,45,1950) f64", "noise": "off",
                       "segments_per_step": int(n_job), "pairs_per_segment": acc["pairs"] / max(acc["S"], 1),
                       "unique_pixels_per_segment": acc["U"] / max(acc["S"], 1),
                       "hits_per_step": acc["hits"] // max(a.steps, 1),
                       "chunk_segments": CHUNK_SEGMENTS, "parallelism": f"batch-sharded x{world}"},
            "roofline": {"bound": "hbm", "kernel": dom_name,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         # HBM bytes per launch from the committed PMC passes (profiles/r01_pmc_*: FETCH_SIZE x2 per the
                         # gfx950 correction + WRITE_SIZE) -- only valid for the default workload/chunking
                         "traffic": PROFILED_TRAFFIC_BYTES if (a.config == "module0" and a.segments == SEGS_PER_GPU
                                                               and a.response == "survey" and a.fractions and split
                                                               and dom_name == "weights_kernel<1>") else None,
                         "launch_ms_avg": dom_ms / nl,
                         "launches": acc["launches"],
                         "algorithmic_bytes_per_launch": acc["bytes"] / nl,
                         "note": "f64-VALU-bound path: HBM fraction is tiny by construction (SURVEY 8d: ~2 KB of "
                                 "compulsory traffic per segment); see valu_f64 and stage_kernels",
                         "stage_kernels": {"weights_kernel_ms_avg": acc["w_ms"] / nl, "mac_kernel_ms_avg": acc["m_ms"] / nl,
                                           "current_kernel_ms_avg": acc["f_ms"] / nl,
                                           "pixel_adc_kernel_ms_avg": acc["adc_ms"] / nl,
                                           "charge_samples_per_launch": acc["samples"] / nl},
                         "valu_f64": {"kernel": f"mac_kernel<{M}>" if split else f"current_kernel<{M}>",
                                      "achieved": tflops, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                      "frac": tflops / FP64_VALU_PEAK_TFLOPS,
                                      "dfma_per_segment": acc["dfma"] / max(acc["S"], 1)}},
        }
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
    if use_dist:
        tdist.barrier()
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
