"""Node rule of the quadrature along the segment against "quad_accuracy_log10" (7, 8, 9, 10, 12): per setting
* speed on the bench set (module0, n segments): tables / correlation ms, average nodes per pair, share of pairs in one node batch;
* accuracy in units of the contract tolerance  tol = 1e-5 |ref| + 1e-7 peak(waveform)  per tick:
    - against every reference sampled golden (tests/golden/sampled_*.npz: the reference's own tracks_current),
    - against the tightest rule (12) on every tick of a slice of the bench set and of fuzz-shaped sets (tools/fuzz_chain.py flavours),
  reported as the largest  |got - ref| / tol  (must stay < 1 for a parity test to pass) and as a fraction of the waveform's peak.
VERDICT r03 item 1(a).  usage: quad_sweep.py [cfg] [n]"""
import glob
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "larnd-sim_amd"), REPO, os.path.join(REPO, "tests"), os.path.join(REPO, "tools")):
    sys.path.insert(0, p)
from larndsim_amd import batching, consts, detsim, lib, synth   # noqa: E402
from larndsim_amd.chain import ChargeChain                       # noqa: E402
from oracle import oracle as O                                   # noqa: E402
import helpers as H                                              # noqa: E402
from qweights_check import prepared                              # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "module0"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
# (quad_accuracy_log10, prune_log, trim_response_log); the first is the reference of the "vs tightest" comparisons
ACCS = ((12, 0.0, 0.0), (10, 23.0, 23.0), (9, 23.0, 23.0), (8, 23.0, 23.0), (7, 23.0, 23.0), (8, 21.0, 21.0), (8, 19.0, 19.0), (7, 19.0, 19.0),
        (7, 17.0, 17.0), (7, 16.0, 16.0))
if os.environ.get("SWEEP"):
    ACCS = tuple(tuple(float(v) for v in a.split(",")) for a in os.environ["SWEEP"].split(";"))


def set_acc(acc):
    lib.set_option("quad_accuracy_log10", acc[0])
    lib.set_option("prune_log", acc[1])
    lib.set_option("trim_response_log", acc[2])


def stage(neigh, r, resp, T, acc):
    sig = np.zeros(neigh.shape + (T,), dtype=np.float32)
    set_acc(acc)
    lib.set_option("gform_max_support", 1e9)
    detsim.tracks_current[(1, 1, 1), (1, 1, 64)](sig, np.ascontiguousarray(neigh), r, resp)
    return sig


def excess(got, ref, peak):
    tol = 1e-5 * np.abs(ref) + 1e-7 * peak
    with np.errstate(invalid="ignore", divide="ignore"):
        e = np.where(tol > 0, np.abs(got - ref) / tol, 0.0)
        rel = np.where(peak > 0, np.abs(got - ref) / peak, 0.0)
    return float(e.max()), float(rel.max())


# ---- speed ---------------------------------------------------------------------------------------------------------------
H.load_cfg(cfg)
resp_kind = "survey" if cfg != "ndlar" else "golden"
seg, bid = prepared(cfg, n, synth.SEED_BASE + 2, 5000)
ch = ChargeChain(H.response_for(resp_kind))
ch.upload(seg, bid)
ch.quench_drift()
for acc in ACCS:
    set_acc(acc)
    ch.run(0, len(seg), want_fractions=True)
    st = ch.run(0, len(seg), want_fractions=True)
    ms = ch.kernel_ms()
    print(f"speed {cfg} {n} acc {acc}: tables {ms['weights_ms']:.2f} ms corr {ms['mac_ms']:.2f} adc {ms['adc_ms']:.2f} total {ms['total_ms']:.2f}"
          f"  nodes/pair {st.n_samples / max(st.n_pairs, 1):.2f} pool {st.n_wbuf * 8e-9:.2f} GB", flush=True)

# ---- accuracy against the reference's sampled goldens --------------------------------------------------------------------
for path in sorted(glob.glob(os.path.join(REPO, "tests", "golden", "sampled_*.npz"))):
    name = os.path.basename(path)[len("sampled_"):-4]
    c = name.replace("corners_", "")
    H.load_cfg(c)
    g = np.load(path, allow_pickle=True)
    r = H.quench_drift(O, g["segments_in"])
    neigh = np.ascontiguousarray(g["neigh"])
    T = int(g["max_length"])
    resp = H.response_for(g["response_kind"])
    for acc in ACCS:
        sig = stage(neigh, r, resp, T, acc)
        peak = np.abs(sig).max(axis=-1, keepdims=True).astype(np.float64)
        e, rel = excess(sig[:, :, g["ticks"]].astype(np.float64), g["signals"].astype(np.float64), peak)
        print(f"golden {name} acc {acc}: worst |err| / tol {e:.3f}   worst |err| / peak {rel:.2e}", flush=True)

# ---- accuracy against the tightest rule, every tick ------------------------------------------------------------------------
sys.argv = [sys.argv[0]]
import fuzz_chain as F     # noqa: E402  (flavoured segment sets)


def sets_for(c):
    H.load_cfg(c)
    out = []
    seg, bid = prepared(c, 400, synth.SEED_BASE + 2, 200)
    out.append(("bench", seg))
    for fl in range(F.N_FLAVOURS):           # 3 cases of 20 segments per flavour
        out.append((f"fuzz flavour {fl}", np.concatenate([F.make_case(7000 + fl + F.N_FLAVOURS * k, c)[0] for k in range(3)])))
    return out


for c in (cfg,):
    for label, s in sets_for(c):
        r = H.quench_drift(O, s[:400])
        nmax = O.max_pixels(r)
        rad = int(np.ceil(r["tran_diff"].max() * 5 / consts.detector.PIXEL_PITCH))
        P = (2 * rad + 1) * nmax + (1 + 2 * rad) * rad * 2
        _, neigh, _, _ = O.get_pixels(r, nmax, P, rad)
        _, T = O.time_intervals(r)
        resp = H.response_for("survey" if c != "ndlar" else "golden")
        ref = stage(neigh, r, resp, T, ACCS[0]).astype(np.float64)
        peak = np.abs(ref).max(axis=-1, keepdims=True)
        for acc in ACCS[1:]:
            sig = stage(neigh, r, resp, T, acc).astype(np.float64)
            e, rel = excess(sig, ref, peak)
            print(f"vs acc 12, {c} {label} ({neigh.size} pairs x {T} ticks) acc {acc}: worst |diff| / tol {e:.3f}   worst |diff| / peak {rel:.2e}",
                  flush=True)
set_acc((10, 23.0, 23.0))
