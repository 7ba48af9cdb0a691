"""debug: per-pixel waveform peaks of one sweep segment on every kernel path vs the oracle"""
import sys, os
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "larnd-sim_amd"), REPO, os.path.join(REPO, "tests")]
import helpers as H
from larndsim_amd import batching, consts, detsim, lib, synth
from oracle import oracle as O
import test_gpu_parity as T

cfg = sys.argv[1] if len(sys.argv) > 1 else "module0"
which = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [14]
H.load_cfg(cfg)
det = consts.detector
B = det.TPC_BORDERS[0]
sgn = np.sign(B[2][1] - B[2][0])
rs = [0.01, 0.8, 4.0, 12.0, 30.0, 36.0, 38.5, 45.0, 70.0, 110.0, 150.0, 156.0, 159.0, 165.0]
dists = [0.03, 0.12, 0.6]
seg = synth.make_segments(len(rs) * len(dists), seed=23, segs_per_event=len(rs) * len(dists))
batching.swap_coordinates(seg)
x0, y0 = B[0][0] + 0.37 * (B[0][1] - B[0][0]), B[1][0] + 0.41 * (B[1][1] - B[1][0])
k = 0
for d in dists:
    sT = np.sqrt(2 * det.TRAN_DIFF * d / det.V_DRIFT)
    for rr in rs:
        L = rr * sT
        ang = 0.7 + 0.37 * k
        a = np.array([x0 + 0.9 * (k % 5), y0 + 0.7 * (k // 5), B[2][0] + sgn * d])
        tilt = 0.0998 if rr >= 1 else 0.6
        b = a + L * np.array([np.cos(ang) * np.sqrt(1 - tilt * tilt), np.sin(ang) * np.sqrt(1 - tilt * tilt), sgn * tilt])
        for i, ax in enumerate("xyz"):
            seg[ax + "_start"][k] = a[i]; seg[ax + "_end"][k] = b[i]
            seg[ax][k] = 0.5 * (np.float32(a[i]).astype(np.float64) + np.float32(b[i]))
        seg["dx"][k] = max(L, 1e-4); seg["dEdx"][k] = 2.1; seg["dE"][k] = 2.1 * seg["dx"][k]
        k += 1
r = H.quench_drift(O, seg)
nmax = O.max_pixels(r)
P = 3 * nmax + 6
_, neigh, nrad, _ = O.get_pixels(r, nmax, P, 1)
_, Tn = O.time_intervals(r)
resp = H.response_for("golden")
sel = np.array(which)
r2, neigh2 = r[sel].copy(), np.ascontiguousarray(neigh[sel])
ref = O.tracks_current(r2, neigh2, Tn, resp)
out = {}
for path in ("mono", "closed", "quad"):
    out[path], st = T._tracks_current_on(path, neigh2, r2, resp, Tn)
    print(path, "fallback", st.n_fallback, "pool", st.n_wbuf)
for i, s in enumerate(which):
    print("segment", s, "n_e", r2["n_electrons"][i], "tran", r2["tran_diff"][i], "len", np.linalg.norm([r2["x_end"][i]-r2["x_start"][i], r2["y_end"][i]-r2["y_start"][i], r2["z_end"][i]-r2["z_start"][i]]))
    for p in range(neigh2.shape[1]):
        pk = np.abs(ref[i, p]).max()
        if pk > 0 or any(np.abs(out[q][i, p]).max() > 0 for q in out):
            print("  pix", neigh2[i, p], "ref %.4e" % pk, " ".join("%s %.4e (maxdiff %.2e)" % (q, np.abs(out[q][i, p]).max(), np.abs(out[q][i, p] - ref[i, p]).max()) for q in out))
