"""gcorr_kernel on a full-support response table against the LDS budgets of its launch classes (debug_lds_pad_kb: first class,
debug_lds_b1_kb: second).  python tools/lds_b1_sweep_dense.py [config]"""
import os, sys, time
REPO="/root/repo"
for p in (os.path.join(REPO, "larnd-sim_amd"), REPO, os.path.join(REPO, "tests"), os.path.join(REPO, "tools")):
    sys.path.insert(0, p)
from larndsim_amd import batching, synth, lib
from larndsim_amd.chain import ChargeChain
import helpers as H
from qweights_check import prepared
CFG = sys.argv[1] if len(sys.argv) > 1 else "module0"
seg, bid = prepared(CFG, 50000, synth.SEED_BASE + 2, 5000)
ch = ChargeChain(H.response_for("dense"))
ch.upload(seg, bid); ch.quench_drift()
os.environ["LDSIM_DEBUG_GFORM"]="1"
ch.run(0, len(seg), want_fractions=True)
del os.environ["LDSIM_DEBUG_GFORM"]
for pad, b1, dbg in ((0, 0, 0), (0, 18, 0), (0, 19, 0), (0, 20, 0), (0, 21, 0), (0, 22, 0), (0, 0, 0)):
    lib.set_option("debug_gform", dbg)
    lib.set_option("debug_lds_pad_kb", pad)
    lib.set_option("debug_lds_b1_kb", b1)
    ch.run(0, len(seg), want_fractions=True)
    ch.run(0, len(seg), want_fractions=True)
    ms = ch.kernel_ms()
    print(f"dbg {dbg} b0 {12 - pad} KB, b1 {b1} KB: tables {ms['weights_ms']:.2f} corr {ms['mac_ms']:.2f}", flush=True)
