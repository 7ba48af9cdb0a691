import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'larnd-sim_amd')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np
from larndsim_amd import batching, consts, lib, synth
from larndsim_amd.chain import ChargeChain
consts.load_snapshot("module0")
for k in ("RESET_NOISE_CHARGE", "UNCORRELATED_NOISE_CHARGE", "DISCRIMINATOR_NOISE"): setattr(consts.detector, k, 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
seg = synth.make_segments(n, seed=synth.SEED_BASE + 2); batching.swap_coordinates(seg)
bid, order, table = batching.assign_batches(seg); seg, bid = seg[order], bid[order]
for kind in (sys.argv[2:] or ["survey", "dense"]):
    ch = ChargeChain(synth.make_response(kind))
    ch.upload(seg, bid)
    for phases in (15, 13, 14, 12):
        lib.set_option("debug_phases", phases)
        for prune in ((30.0,) if phases != 15 else (30.0, 20.0, 0.0)):
            lib.set_option("prune_log", prune)
            ch.reset(); ch.quench_drift(); st = ch.run(0, n)
            ch.reset(); ch.quench_drift(); st = ch.run(0, n)
            ms = ch.kernel_ms()
            print(f"{kind:7s} phases={phases} prune={prune:4.0f} current {ms['current_ms']:8.2f} ms adc {ms['adc_ms']:6.2f} total {ms['total_ms']:8.2f} "
                  f"dfma/seg {st.n_dfma/n:.3g} TF {2*st.n_dfma/ms['current_ms']/1e9:.2f} pairs/seg {st.n_pairs/n:.2f}", flush=True)
    lib.set_option("debug_phases", 15); lib.set_option("prune_log", 30.0)
