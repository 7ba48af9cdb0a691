"""qweights_kernel (Gauss-Legendre weights, weights_mode 1) against weights_kernel (per-sample closed form, mode 0) and the
monolithic current_kernel: hit pattern / ticks / ADC counts must be identical, charges equal to ~1e-11; then the kernel times
of both weight stages on 20k segments.  Run on the GPU box: python tools/qweights_check.py"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "larnd-sim_amd"), REPO, os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
from larndsim_amd import batching, consts, lib, synth   # noqa: E402
from larndsim_amd.chain import ChargeChain              # noqa: E402
import helpers as H                                      # noqa: E402


def prepared(cfg, n, seed, per_event):
    H.load_cfg(cfg)
    seg = synth.make_segments(n, seed=seed, segs_per_event=per_event, spill=bool(consts.sim.IS_SPILL_SIM))
    if consts.sim.IS_SPILL_SIM:
        loc = seg["event_id"] % consts.sim.MAX_EVENTS_PER_FILE
        for f in ("t0", "t0_start", "t0_end"):
            seg[f] = seg[f] - loc * consts.sim.SPILL_PERIOD
    batching.swap_coordinates(seg)
    bid, order, table = batching.assign_batches(seg)
    return seg[order], bid[order]


def compare(cfg, kind, n=1200):
    seg, bid = prepared(cfg, n, 33, 600)
    ch = ChargeChain(H.response_for(kind))
    ch.upload(seg, bid)
    ch.quench_drift()
    res = {}
    for name, split, mode in (("mono", 0, 0), ("sample", 1, 0), ("quad", 1, 1)):
        lib.set_option("split_kernels", split)
        lib.set_option("weights_mode", mode)
        st = ch.run(0, len(seg), want_fractions=True)
        res[name] = ch.download()
        print(f"  {cfg}/{kind} {name}: pairs {st.n_pairs} fallback {st.n_fallback} wbuf/pair {st.n_wbuf / max(st.n_pairs, 1):.0f} "
              f"samples-or-nodes/pair {st.n_samples / max(st.n_pairs, 1):.1f} dfma/seg {st.n_dfma / n:.3g}", flush=True)
    lib.set_option("split_kernels", 1)
    lib.set_option("weights_mode", 1)
    a = res["mono"]
    for name in ("sample", "quad"):
        b = res[name]
        ok = (np.array_equal(a["unique_pix"], b["unique_pix"]) and np.array_equal(a["track_pixel_map"], b["track_pixel_map"]))
        hits = np.array_equal(a["adc_list"] != 0, b["adc_list"] != 0)
        ticks = np.array_equal(a["adc_ticks_list"], b["adc_ticks_list"])
        digit = np.array_equal(a["adc_digit"], b["adc_digit"])
        m = a["adc_list"] != 0
        rel = np.abs(b["adc_list"][m] - a["adc_list"][m]) / np.abs(a["adc_list"][m])
        fr = np.abs(b["current_fractions"] - a["current_fractions"]).max()
        print(f"  {cfg}/{kind} {name} vs mono: pix/map {ok} hits {hits} ({m.sum()}) ticks {ticks} digit {digit} "
              f"max rel charge {rel.max():.2e} median {np.median(rel):.2e} max |dfrac| {fr:.2e}", flush=True)


def timing(cfg, kind, n=20000):
    seg, bid = prepared(cfg, n, synth.SEED_BASE + 2, 5000)
    ch = ChargeChain(H.response_for(kind))
    ch.upload(seg, bid)
    ch.quench_drift()
    for mode in (0, 1, 0, 1):
        lib.set_option("weights_mode", mode)
        ch.run(0, len(seg), want_fractions=True)     # warm-up (pool sizing)
        t0 = time.perf_counter()
        st = ch.run(0, len(seg), want_fractions=True)
        wall = time.perf_counter() - t0
        ms = ch.kernel_ms()
        print(f"  {cfg}/{kind} weights_mode {mode}: weights {ms['weights_ms']:.2f} ms mac {ms['mac_ms']:.2f} fallback "
              f"{ms['fallback_ms']:.2f} adc {ms['adc_ms']:.2f} total {ms['total_ms']:.2f} wall {1e3 * wall:.1f} ms; "
              f"fallback pairs {st.n_fallback}/{st.n_pairs}", flush=True)
    lib.set_option("weights_mode", 1)


if __name__ == "__main__":
    for cfg, kind in (("module0", "dense"), ("ndlar", "golden"), ("2x2_no_modvar", "survey")):
        compare(cfg, kind)
    timing("module0", "survey")
    timing("ndlar", "survey")
