"""Accuracy and speed of the f32 tail class of weights_kernel versus tail_log (0 = all f64)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'larnd-sim_amd')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np
from larndsim_amd import batching, consts, lib, synth
from larndsim_amd.chain import ChargeChain
consts.load_snapshot("module0")
for k in ("RESET_NOISE_CHARGE", "UNCORRELATED_NOISE_CHARGE", "DISCRIMINATOR_NOISE"): setattr(consts.detector, k, 0)
n = 20000
seg = synth.make_segments(n, seed=synth.SEED_BASE + 2); batching.swap_coordinates(seg)
bid, order, table = batching.assign_batches(seg); seg, bid = seg[order], bid[order]
ch = ChargeChain(synth.make_response("survey")); ch.upload(seg, bid)
ref = None
for tl in (0, 20, 17, 14, 11, 8, 5):
    lib.set_option("tail_log", tl)
    for _ in range(2):
        ch.reset(); ch.quench_drift(); st = ch.run(0, n)
    ms = ch.kernel_ms()
    out = ch.download()
    if ref is None:
        ref = out
        print("tail_log 0: weights %.2f mac %.2f ms" % (ms["weights_ms"], ms["mac_ms"]))
        continue
    a, b = out["adc_list"], ref["adc_list"]
    m = b != 0
    rel = np.abs(a[m] - b[m]) / np.abs(b[m])
    same_hits = np.array_equal(a != 0, b != 0)
    ticks_equal = np.array_equal(out["adc_ticks_list"], ref["adc_ticks_list"])
    digit_equal = np.array_equal(out["adc_digit"], ref["adc_digit"])
    print("tail_log %2d: weights %.2f ms  adc rel err max %.2e median %.2e  hits same %s ticks equal %s digit equal %s"
          % (tl, ms["weights_ms"], rel.max(), np.median(rel), same_hits, ticks_equal, digit_equal))
