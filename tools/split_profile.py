"""Per-kernel time of the tracks_current stage on 20 k synthetic segments of one configuration:
python tools/split_profile.py [module0|2x2_no_modvar|ndlar] [survey|dense]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'larnd-sim_amd')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np
import helpers as H
from larndsim_amd import batching, consts, lib, synth
from larndsim_amd.chain import ChargeChain
cfg = sys.argv[1] if len(sys.argv) > 1 else "module0"
kind = sys.argv[2] if len(sys.argv) > 2 else "survey"
seed = {"module0": 2, "2x2_no_modvar": 3, "ndlar": 5}[cfg]
H.load_cfg(cfg)
n = 20000
seg = synth.make_segments(n, seed=20241016 + seed, spill=bool(consts.sim.IS_SPILL_SIM))
if consts.sim.IS_SPILL_SIM:
    loc = seg["event_id"] % consts.sim.MAX_EVENTS_PER_FILE
    for f in ("t0", "t0_start", "t0_end"): seg[f] = seg[f] - loc * consts.sim.SPILL_PERIOD
batching.swap_coordinates(seg)
bid, order, table = batching.assign_batches(seg); seg, bid = seg[order], bid[order]
ch = ChargeChain(H.response_for(kind)); ch.upload(seg, bid)
for _ in range(3):      # the first launches size the weight pool
    ch.reset(); ch.quench_drift(); st = ch.run(0, n)
ms = ch.kernel_ms()
print(cfg, kind, "M", consts.detector.TIME_SAMPLING / consts.detector.RESPONSE_SAMPLING,
      "| weights %.1f mac %.1f fallback-kernel %.1f adc %.1f total %.1f ms" % (ms["weights_ms"], ms["mac_ms"], ms["fallback_ms"], ms["adc_ms"], ms["total_ms"]),
      "| pairs", st.n_pairs, "fallback pairs", st.n_fallback, "(%.2f %%)" % (100.0 * st.n_fallback / max(st.n_pairs, 1)),
      "| weight doubles/pair %.0f" % (st.n_wbuf / max(st.n_pairs, 1)), "samples/pair %.0f" % (st.n_samples / max(st.n_pairs, 1)),
      "dfma/seg %.3g" % (st.n_dfma / n))
