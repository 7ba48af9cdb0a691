"""Where should weights_mode 2 hand a response table to the shifted-window kernels?  The bench workload on the synthetic `dense` table
(no zeros over 1950 ticks) with gform_max_support at its default (768 ticks: qweights_kernel + mac_shift_kernel run) and forced to the
matrix form.  python tools/dense_handover.py [config [response [segments]]]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "larnd-sim_amd"), REPO, os.path.join(REPO, "tests"), os.path.join(REPO, "tools")):
    sys.path.insert(0, p)
from larndsim_amd import batching, synth, lib
from larndsim_amd.chain import ChargeChain
import helpers as H
from qweights_check import prepared
CFG = sys.argv[1] if len(sys.argv) > 1 else "module0"
RESP = sys.argv[2] if len(sys.argv) > 2 else "dense"
NSEG = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
seg, bid = prepared(CFG, NSEG, synth.SEED_BASE + 2, 5000)
ch = ChargeChain(H.response_for(RESP))
ch.upload(seg, bid)
ranges = batching.chunk_ranges(bid, 50000)
def run(steps):
    for _ in range(steps):
        ch.reset(); ch.quench_drift()
        for b, e in ranges:
            st = ch.run(b, e, want_fractions=True)
    ch.synchronize()
    return st
for sup in (768, 1e9):
    lib.set_option("gform_max_support", sup)
    run(1)
    t=time.time(); st=run(2); dt=(time.time()-t)/2
    ms=ch.kernel_ms()
    print(f"{CFG} {RESP} gform_max_support {sup:g}: {NSEG/dt:.4g} segments/s, weights {ms['weights_ms']:.1f} mac {ms['mac_ms']:.1f} ms per 50k, fallback pairs {st.n_fallback}", flush=True)
