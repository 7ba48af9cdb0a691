"""gcorr_kernel's second LDS class (pairs whose tables do not fit the first budget): time of the correlation stage against the
budget of that class (option debug_lds_b1_kb; 32 KB = five pairs per CU is the default).
    python tools/lds_b1_sweep.py [cfg] [n_segments]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "larnd-sim_amd"), REPO, os.path.join(REPO, "tests"), os.path.join(REPO, "tools")):
    sys.path.insert(0, p)
from larndsim_amd import lib, synth          # noqa: E402
from larndsim_amd.chain import ChargeChain   # noqa: E402
import helpers as H                          # noqa: E402
from qweights_check import prepared          # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "ndlar"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
seg, bid = prepared(cfg, n, synth.SEED_BASE + 2, 5000)
ch = ChargeChain(H.response_for("survey"))
ch.upload(seg, bid)
ch.quench_drift()
ch.run(0, len(seg), want_fractions=True)
os.environ["LDSIM_DEBUG_GFORM"] = "1"
for kb in (32, 28, 26, 24, 22, 20, 18, 32):
    lib.set_option("debug_lds_b1_kb", kb)
    ch.run(0, len(seg), want_fractions=True)
    ms = ch.kernel_ms()
    print(f"{cfg} second class at {kb} KB: tables {ms['weights_ms']:.2f} ms  corr {ms['mac_ms']:.2f}", flush=True)
lib.set_option("debug_lds_b1_kb", 0)
