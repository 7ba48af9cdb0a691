"""[The switches and cycle stamps INSIDE gcorr_kernel (modes corr, stamps, share, zlds, psum, brow) need a library built with
`make -C larnd-sim_amd/csrc DEBUG_GCORR=1`: the production build compiles them out -- they cost vector instructions.]
Where the time of gtables_kernel / gcorr_kernel goes, by switching parts off (results are wrong in those runs).
gcorr: 0x100000 no tile loop (prologue + combine + store), 0x200000 no P step, 0x400000 every cell reads response row 0,
0x800000 no G products.  gtables: 0x1000000 stop after the sample maps, 0x2000000 no X / Y tables, 0x4000000 no Z tables,
0x8000000 no cell list."""
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
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
seg, bid = prepared(cfg, n, synth.SEED_BASE + 2, 5000)
ch = ChargeChain(H.response_for(resp))
ch.upload(seg, bid)
ch.quench_drift()
ch.run(0, len(seg), want_fractions=True)
# pad: KB off the LDS budget of gcorr's small class (19.5 KB); dbg: option debug_gform (1 no counter atomics, 2 no edge rows,
# 4 no X / Y / Z staging in gcorr, 8 no stores of the result, 16 no record stores in gtables, 32 no edge rows in gtables)
WHICH = sys.argv[4] if len(sys.argv) > 4 else "tables"
RUNS = {"tables": ((0, 0, 0), (0x1000000, 0, 0), (0x2000000, 0, 0), (0x4000000, 0, 0), (0x8000000, 0, 0), (0x6000000, 0, 0),
                   (0xe000000, 0, 0), (0xe000000, 0, 16), (0, 0, 16), (0, 0, 32), (0, 0, 0)),
        "adc": ((0, 0, 0), (0x10000, 0, 0), (0x20000, 0, 0), (0x40000, 0, 0), (0x60000, 0, 0), (0x70000, 0, 0), (0, 0, 0)),
        "occ": ((0, 0, 0), (0, -1, 0), (0, -2, 0), (0, -3, 0), (0, 1, 0), (0, 2, 0), (0, 0, 0)),
        # in-kernel cycle stamps of gcorr_kernel's waves (debug_gform 128; printed to stderr by the library)
        "stamps": ((0, 0, 128), (0, 0, 0)),
        # the same for gtables_wave_kernel (debug_gform 2048)
        "tstamps": ((0, 0, 2048), (0, 0, 0)),
        # the 4-node-block kernel (104 VGPRs, four waves per SIMD) also for the launch over all pairs (debug_gform 65536)
        "qball": ((0, 0, 0), (0, 0, 65536), (0, 0, 0), (0, 0, 65536)),
        # every cell reads response row 0 (debug_gform 32768; results are wrong): what the G loop costs when its B operands come from L1
        "brow": ((0, 0, 0), (0, 0, 32768), (0, 0, 128), (0, 0, 128 + 32768)),
        # G by 4-node blocks in the launches of the larger LDS classes (default) against the 16-node product everywhere (debug_gform 4096)
        "q4": ((0, 0, 0), (0, 0, 4096), (0, 0, 0), (0, 0, 4096)),
        # the P step without its LDS atomics (debug_gform 16384; results are wrong), with cycle stamps
        "psum": ((0, 0, 128), (0, 0, 128 + 16384), (0, 0, 0), (0, 0, 16384)),
        # an odd tile pair shared by the pair's two waves (default) against dealt whole (debug_gform 256)
        # Z staged in LDS where it fits (default) against read from the record in the P step for every pair (debug_gform 1024), at
        # several LDS budgets of the first class (13 KB - pad)
        "zlds": ((0, 0, 0), (0, 0, 1024), (0, 2, 1024), (0, 4, 1024), (0, 1, 0), (0, 1, 1024), (0, 0, 0)),
        "share": ((0, 0, 0), (0, 0, 256), (0, 0, 128), (0, 0, 384), (0, 0, 0), (0, 0, 256)),
        "corr": ((0, 0, 0), (0x100000, 0, 0), (0x200000, 0, 0), (0, 0, 2), (0, 0, 4), (0, 0, 8), (0x100000, 0, 4), (0x100000, 0, 12),
                 (0x100000, 0, 14), (0, -12, 0), (0, 6, 0), (0, 0, 0))}
for mask, pad, dbg in RUNS[WHICH]:
    lib.set_option("debug_phases", 15 | mask)
    lib.set_option("debug_lds_pad_kb", pad)
    lib.set_option("debug_gform", dbg)
    st = ch.run(0, len(seg), want_fractions=True)
    ms = ch.kernel_ms()
    print(f"{cfg} {resp} debug_phases {mask:#10x} lds pad {pad:2d} KB dbg {dbg:2d}: tables {ms['weights_ms']:.2f} ms  corr {ms['mac_ms']:.2f}  fallback {ms['fallback_ms']:.2f}  adc {ms['adc_ms']:.2f}  pairs {st.n_pairs} pool {st.n_wbuf} mfma {st.n_dfma // 1024}", flush=True)
lib.set_option("debug_phases", 15)
lib.set_option("debug_gform", 0)
