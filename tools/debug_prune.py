import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'larnd-sim_amd')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np
import helpers as H
from larndsim_amd import consts, detsim, lib, synth
from oracle import oracle as O
cfg = sys.argv[1] if len(sys.argv) > 1 else 'module0'
H.load_cfg(cfg)
g = H.gold(f"sampled_{cfg}.npz")
r = H.quench_drift(O, g["segments_in"])
neigh = np.ascontiguousarray(g["neigh"]); T = int(g["max_length"])
resp = H.response_for(g["response_kind"])
lib.context()
res = {}
for prune in (30.0, 0.0):
    lib.set_option("prune_log", prune)
    sig = np.zeros(neigh.shape + (T,), dtype=np.float32)
    detsim.tracks_current[(1,1,1),(1,1,64)](sig, neigh, r, resp)
    res[prune] = sig.astype(np.float64)
d = np.abs(res[30.0] - res[0.0])
pk = np.abs(res[30.0]).max(-1)
bad = np.argwhere(d.max(-1) > 1e-6 * (pk + 1e-30))
print("pairs differing:", len(bad), "of", (neigh >= 0).sum())
for (s, p) in bad[:6]:
    w = np.flatnonzero(d[s, p] > 1e-6 * pk[s, p])
    seg = r[s]
    dz = abs(seg['z_end'] - seg['z_start']); L = np.sqrt((seg['x_end']-seg['x_start'])**2 + (seg['y_end']-seg['y_start'])**2 + dz**2)
    print(f"seg {s} pix {p} id {neigh[s,p]} peak {pk[s,p]:.4g} ticks {w.min()}..{w.max()} n {len(w)} dz {dz:.3f} L {L:.3f} sL {seg['long_diff']:.4f} "
          f"maxdiff {d[s,p].max():.4g} at {d[s,p].argmax()} on {res[30.0][s,p,d[s,p].argmax()]:.5g} off {res[0.0][s,p,d[s,p].argmax()]:.5g}")
