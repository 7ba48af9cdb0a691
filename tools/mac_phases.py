"""Where mac_shift_kernel's time goes, by switching parts off (results are wrong in those runs): debug_phases bit 0x1000 skips
the correlation blocks (item walk, loads and the first shift stay), 0x2000 skips the item walk (prologue + combine + store),
0x4000 makes every item read the row of cell 0 (row traffic served by the nearest cache)."""
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
resp = sys.argv[2] if len(sys.argv) > 2 else "survey"
seg, bid = prepared(cfg, 20000, synth.SEED_BASE + 2, 5000)
ch = ChargeChain(H.response_for(resp))
ch.upload(seg, bid)
ch.quench_drift()
ch.run(0, len(seg), want_fractions=True)
for mode in (1, 0):
    lib.set_option("mac_mode", mode)
    for mask in ((15, 15 | 0x1000, 15 | 0x2000, 15 | 0x4000, 15 | 0x5000, 15) if mode else (15,)):
        lib.set_option("debug_phases", mask)
        ch.run(0, len(seg), want_fractions=True)
        ms = ch.kernel_ms()
        print(f"{cfg} {resp} mac_mode {mode} debug_phases {mask:#6x}: weights {ms['weights_ms']:.2f} ms  mac {ms['mac_ms']:.2f}", flush=True)
lib.set_option("debug_phases", 15)
lib.set_option("mac_mode", 1)
