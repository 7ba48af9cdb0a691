"""Where weights_kernel's time goes: debug_phases bit0 = evaluate queued samples, bit2 = real transcendentals,
bit3 = deposit into the LDS bins.  (Results are wrong unless all bits are set.)"""
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
for tl in (0, 14):
    lib.set_option("tail_log", tl)
    for ph, what in ((15, "all"), (7, "no deposit"), (11, "no transcendentals"), (3, "no transcendentals, no deposit"), (2, "pass A + emit only"), (0, "fixed per-pair work only")):
        lib.set_option("debug_phases", ph)
        for _ in range(2):
            ch.reset(); ch.quench_drift(); st = ch.run(0, n)
        ms = ch.kernel_ms()
        print("tail_log %2d phases %2d (%s): weights %.2f ms  samples/pair %.0f" % (tl, ph, what, ms["weights_ms"], st.n_samples / max(st.n_pairs, 1)))
lib.set_option("debug_phases", 15)
