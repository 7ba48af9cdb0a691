"""Runs the weights stage in modes 1 (qweights_kernel, workgroup per pair) and 2 (qwave_kernel, wave per pair) on 20k module0
segments, three launches each -- the target of rocprofv3 --pmc passes (profiles/README.md)."""
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
for mode in (1, 2):
    lib.set_option("weights_mode", mode)
    for _ in range(3):
        ch.run(0, len(seg), want_fractions=True)
    print(mode, ch.kernel_ms())
