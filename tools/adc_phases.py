"""pixel_adc_kernel with parts switched off (debug_phases bits 0x10000 no waveform sum, 0x20000 no trigger scan, 0x40000 no
backtracking fractions; results are wrong in those runs by construction).  python tools/adc_phases.py [cfg]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "larnd-sim_amd"), REPO, os.path.join(REPO, "tests"), os.path.join(REPO, "tools")):
    sys.path.insert(0, p)
from larndsim_amd import lib, synth          # noqa: E402
from larndsim_amd.chain import ChargeChain   # noqa: E402
import helpers as H                          # noqa: E402
from qweights_check import prepared          # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "module0"
seg, bid = prepared(cfg, 20000, synth.SEED_BASE + 2, 5000)
ch = ChargeChain(H.response_for("survey"))
ch.upload(seg, bid)
ch.quench_drift()
ch.run(0, len(seg), want_fractions=True)
for mask, what in ((0, "all"), (0x10000, "no waveform sum"), (0x20000, "no scan"), (0x40000, "no fractions"),
                   (0x30000, "no sum, no scan"), (0x70000, "prologue + stores only"), (0, "all")):
    lib.set_option("debug_phases", 15 | mask)
    ch.run(0, len(seg), want_fractions=True)
    ms = ch.kernel_ms()
    print(f"{cfg} {what:24s}: adc {ms['adc_ms']:.3f} ms", flush=True)
for wf in (True, False):
    lib.set_option("debug_phases", 15)
    ch.run(0, len(seg), want_fractions=wf)
    print(f"{cfg} want_fractions={wf}: adc {ch.kernel_ms()['adc_ms']:.3f} ms", flush=True)
