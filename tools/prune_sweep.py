"""Speed and accuracy of the split path versus prune_log (weights below exp(-prune_log) of the pair's peak weight are dropped;
0 = nothing dropped).  python tools/prune_sweep.py [cfg] [response]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'larnd-sim_amd')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np
from larndsim_amd import batching, consts, lib, synth
from larndsim_amd.chain import ChargeChain
cfg = sys.argv[1] if len(sys.argv) > 1 else "module0"
kind = sys.argv[2] if len(sys.argv) > 2 else "survey"
consts.load_snapshot(cfg)
for k in ("RESET_NOISE_CHARGE", "UNCORRELATED_NOISE_CHARGE", "DISCRIMINATOR_NOISE"): setattr(consts.detector, k, 0)
n = 20000
seg = synth.make_segments(n, seed=synth.SEED_BASE + 2, spill=bool(consts.sim.IS_SPILL_SIM)); batching.swap_coordinates(seg)
bid, order, table = batching.assign_batches(seg); seg, bid = seg[order], bid[order]
ch = ChargeChain(synth.make_response(kind, response_sampling=consts.detector.RESPONSE_SAMPLING)); ch.upload(seg, bid)
ref = None
for pl in (0, 30, 27, 25, 23, 21, 19):
    lib.set_option("prune_log", pl)
    for _ in range(2):
        ch.reset(); ch.quench_drift(); st = ch.run(0, n)
    ms = ch.kernel_ms()
    out = ch.download()
    if ref is None:
        ref = out
        print("%s %s prune_log 0: weights %.2f mac %.2f ms" % (cfg, kind, ms["weights_ms"], ms["mac_ms"]), flush=True)
        continue
    a, b = out["adc_list"], ref["adc_list"]
    m = b != 0
    rel = np.abs(a[m] - b[m]) / np.abs(b[m])
    print("prune_log %2d: weights %.2f mac %.2f ms  adc rel err max %.2e median %.2e  hits same %s ticks equal %s digit equal %s"
          % (pl, ms["weights_ms"], ms["mac_ms"], rel.max(), np.median(rel), np.array_equal(a != 0, b != 0),
             np.array_equal(out["adc_ticks_list"], ref["adc_ticks_list"]), np.array_equal(out["adc_digit"], ref["adc_digit"])), flush=True)
lib.set_option("prune_log", 30)
