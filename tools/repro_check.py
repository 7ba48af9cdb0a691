"""Is a batch's result bit-reproducible?  Mirrors tests/test_gpu_parity.py::test_chain_properties_baseline_sizes: a
50 k-segment chunking of the whole set, then a 20 k chunking of the first / last 100 k segments and single-batch launches;
every batch that differs is reported with the size of the difference and both launches' fallback counts."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'larnd-sim_amd')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np
import helpers as H
from larndsim_amd import batching, consts, lib, synth
from larndsim_amd.chain import ChargeChain

cfg = sys.argv[1] if len(sys.argv) > 1 else "2x2_no_modvar"
seed = {"module0": 2, "2x2_no_modvar": 3, "ndlar": 5}[cfg]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
H.load_cfg(cfg)
seg = synth.make_segments(n, seed=20241016 + seed, spill=bool(consts.sim.IS_SPILL_SIM))
if consts.sim.IS_SPILL_SIM:
    loc = seg["event_id"] % consts.sim.MAX_EVENTS_PER_FILE
    for f in ("t0", "t0_start", "t0_end"): seg[f] = seg[f] - loc * consts.sim.SPILL_PERIOD
batching.swap_coordinates(seg)
bid, order, table = batching.assign_batches(seg); seg, bid = seg[order], bid[order]
ch = ChargeChain(H.response_for("survey")); ch.upload(seg, bid); ch.quench_drift()
KEYS = ("unique_pix", "adc_list", "adc_ticks_list", "adc_digit")


def per_batch(b0, e0):
    st = ch.run(b0, e0); r = ch.download()
    out = {}
    edges = np.flatnonzero(np.r_[True, r["batch"][1:] != r["batch"][:-1], True])
    for lo, hi in zip(edges[:-1], edges[1:]):
        out[int(r["batch"][lo])] = {k: r[k][lo:hi].copy() for k in KEYS}
    return out, (st.n_pairs, st.n_fallback, st.n_wbuf)


second = [r for r in batching.chunk_ranges(bid, 20_000) if r[1] <= 100_000 or r[0] >= n - 100_000]
ub = np.unique(bid)
for b in (ub[0], ub[len(ub) // 2], ub[-1]):
    w = np.flatnonzero(bid == b); second.append((int(w[0]), int(w[-1]) + 1))
want = set()
for b0, e0 in second:
    want |= set(np.unique(bid[b0:e0]).tolist())
first, first_launch = {}, {}
for b0, e0 in batching.chunk_ranges(bid, 50_000):
    rows, info = per_batch(b0, e0)
    for b, v in rows.items():
        if b in want:
            first[b] = v; first_launch[b] = (b0, e0, info)
print(cfg, n, "batches kept", len(first), "second-pass launches", len(second))
bad = 0
for b0, e0 in second:
    rows, info = per_batch(b0, e0)
    for b, v in rows.items():
        a = first[b]
        if all(np.array_equal(a[k], v[k]) for k in KEYS):
            continue
        bad += 1
        same_pix = np.array_equal(a["unique_pix"], v["unique_pix"])
        msg = "batch %d: launch %s (pairs,fallback,wbuf %s) vs launch %s %s; pixels same %s rows %d/%d" % (
            b, (b0, e0), info, first_launch[b][:2], first_launch[b][2], same_pix, len(a["unique_pix"]), len(v["unique_pix"]))
        if same_pix:
            d = a["adc_list"] != v["adc_list"]
            rel = np.abs(a["adc_list"] - v["adc_list"])[d] / np.maximum(np.abs(a["adc_list"][d]), 1e-300)
            msg += "; adc differing %d max rel %.3e; ticks equal %s; differing pixel rows %s" % (
                int(d.sum()), rel.max() if rel.size else 0, np.array_equal(a["adc_ticks_list"], v["adc_ticks_list"]),
                np.flatnonzero(d.any(axis=1))[:5].tolist())
        print(msg)
print("batches differing:", bad)
