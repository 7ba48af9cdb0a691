"""Tables and correlation of the node-separable form in K pair ranges on two streams (option gform_chunks): the chain's time per
launch against K.  python tools/gform_chunks.py [config [response [segments]]]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "larnd-sim_amd"), REPO, os.path.join(REPO, "tests"), os.path.join(REPO, "tools")):
    sys.path.insert(0, p)
from larndsim_amd import synth, lib
from larndsim_amd.chain import ChargeChain
import helpers as H
from qweights_check import prepared
CFG = sys.argv[1] if len(sys.argv) > 1 else "module0"
RESP = sys.argv[2] if len(sys.argv) > 2 else "survey"
NSEG = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
seg, bid = prepared(CFG, NSEG, synth.SEED_BASE + 2, 5000)
ch = ChargeChain(H.response_for(RESP))
ch.upload(seg, bid)
ch.quench_drift()
ref = None
for K in (1, 2, 4, 8, 16, 32, 1):
    lib.set_option("gform_chunks", K)
    ch.run(0, len(seg), want_fractions=True)
    best = 1e9
    for _ in range(3):
        st = ch.run(0, len(seg), want_fractions=True)
        ms = ch.kernel_ms()
        best = min(best, ms["total_ms"])
    c = ch.download_compact()
    sig = (int(c["hit_pixels"][:, 3].sum()), c["hit_charge"].tobytes(), c["fractions"].tobytes())
    ref = ref or sig
    print(f"{CFG} {RESP} {NSEG}: gform_chunks {K:2d}: chain {best:.2f} ms (current stage {ms['current_ms']:.2f}: first tables {ms['weights_ms']:.2f} + rest {ms['mac_ms']:.2f}), "
          f"hits {sig[0]} charges and fractions {'same bits' if sig == ref else 'DIFFERENT'}", flush=True)
lib.set_option("gform_chunks", 1)
