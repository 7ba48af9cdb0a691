"""Photon-sum loop of the ndlar bench step (one ldsim_dev_sum_light per (event, TPC group) batch, no truth slots): how long the host
takes to issue the calls, and how long until the GPU has finished them.
    python tools/light_sum_loop.py [n_segments]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "larnd-sim_amd"))
from larndsim_amd import batching, consts, synth
from larndsim_amd.chain import ChargeChain

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
consts.load_snapshot("ndlar")
synth.set_synthetic_light(48)
lut = synth.make_lut((14, 26, 8), 48, 100, synth.SEED_BASE + 5)
seg = synth.make_segments(n, seed=synth.SEED_BASE + 5, spill=bool(consts.sim.IS_SPILL_SIM))
if consts.sim.IS_SPILL_SIM:
    loc = seg["event_id"] % consts.sim.MAX_EVENTS_PER_FILE
    for f in ("t0", "t0_start", "t0_end"):
        seg[f] = seg[f] - loc * consts.sim.SPILL_PERIOD
batching.swap_coordinates(seg)
bid, order, table = batching.assign_batches(seg)
seg, bid = np.ascontiguousarray(seg[order]), bid[order]
n_sim = int((bid >= 0).sum())
edges = np.flatnonzero(np.r_[True, bid[1:n_sim] != bid[:n_sim - 1], True])
opc = consts.light.TPC_TO_OP_CHANNEL[:].ravel().astype(np.int32)
ch = ChargeChain()
ch.upload(seg, bid)
ch.quench_drift()
ch.light_incidence(lut)
from larndsim_amd import lib
for rep in range(4):
    if rep == 2:
        lib.set_option("light_sum_async", 1, ch.ctx)
        print("on the light stream (option light_sum_async):")
    ch.synchronize()
    t0 = time.perf_counter()
    for b, e in zip(edges[:-1], edges[1:]):
        ch.sum_light(int(b), int(e), opc)
    t1 = time.perf_counter()
    ch.synchronize()
    t2 = time.perf_counter()
    nb = len(edges) - 1
    print(f"{nb} photon sums: host issued them in {1e3 * (t1 - t0):.2f} ms ({1e6 * (t1 - t0) / nb:.1f} us each), "
          f"GPU done after {1e3 * (t2 - t0):.2f} ms ({1e6 * (t2 - t0) / nb:.1f} us each)")
