"""Phase timing of qweights_kernel via the debug_phases switches (results are wrong unless 15): 1 tables, 2 accumulate,
4 window-edge corrections, 8 emission."""
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
lib.set_option("weights_mode", 1)
ch.run(0, len(seg), want_fractions=True)
for mask in (15, 0x100, 0x200, 0x400, 0, 1, 2, 4, 8, 3, 7, 11, 15):
    lib.set_option("debug_phases", mask)
    ch.run(0, len(seg), want_fractions=True)
    ms = ch.kernel_ms()
    print(f"{cfg} debug_phases {mask:4d}: weights {ms['weights_ms']:.2f} ms  mac {ms['mac_ms']:.2f}", flush=True)
lib.set_option("debug_phases", 15)
