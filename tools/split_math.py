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
kind = sys.argv[1] if len(sys.argv) > 1 else "survey"
ch = ChargeChain(synth.make_response(kind)); ch.upload(seg, bid)
lib.set_option("split_kernels", 1)
for ph in (15,):
    lib.set_option("debug_phases", ph)
    for _ in range(2):
        ch.reset(); ch.quench_drift(); st = ch.run(0, n)
    print("phases", ph, ch.kernel_ms(), "wbuf per pair", st.n_wbuf / st.n_pairs, "samples per pair", st.n_samples / st.n_pairs, "fallback", st.n_fallback)
