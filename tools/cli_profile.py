"""Where the bundled simulate_pixels CLI spends its time on a synthetic spill: python tools/cli_profile.py [cfg] [n_segments] [light]
(`light`: also the light leg on a synthetic LUT -- incidence, photon sums, waveform chain, light datasets)
Prints wall time, segments/s and the cProfile top of cumulative time (host side: batching, packet building, writers)."""
import cProfile
import importlib.util
import os
import pstats
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "larnd-sim_amd"), REPO, os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import numpy as np                                      # noqa: E402
from larndsim_amd import synth                          # noqa: E402
import helpers as H                                     # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "module0"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
with_light = len(sys.argv) > 3 and sys.argv[3] == "light"
spec = importlib.util.spec_from_file_location("ldsim_cli", os.path.join(REPO, "larnd-sim_amd", "cli", "simulate_pixels.py"))
cli = importlib.util.module_from_spec(spec)
spec.loader.exec_module(cli)
write_batch = int(os.environ.get("LDSIM_WRITE_BATCH_SIZE", "0"))      # simulation property write_batch_size (0: the configuration's)
if write_batch:
    from larndsim_amd import consts
    _load = consts.load_snapshot

    def _load_with_write_batch(name):
        r = _load(name)
        consts.sim.WRITE_BATCH_SIZE = write_batch
        return r
    consts.load_snapshot = _load_with_write_batch
H.load_cfg(cfg)
seg = synth.make_segments(n, seed=synth.SEED_BASE + 2, segs_per_event=5000)
with tempfile.TemporaryDirectory() as d:
    np.save(os.path.join(d, "in.npy"), seg)
    np.save(os.path.join(d, "resp.npy"), synth.make_response("survey"))
    kw = dict(config=cfg, response_file=os.path.join(d, "resp.npy"), rand_seed=5)
    if with_light:
        np.savez(os.path.join(d, "lut.npz"), arr=synth.make_lut((14, 26, 8), 48, 100, 3))
        kw["light_lut_filename"] = os.path.join(d, "lut.npz")
    cli.run_simulation(os.path.join(d, "in.npy"), os.path.join(d, "warm.npz"), **kw)       # library, constants, caches
    pr = cProfile.Profile()
    t0 = time.time()
    pr.enable()
    res = cli.run_simulation(os.path.join(d, "in.npy"), os.path.join(d, "out.npz"), **kw)
    pr.disable()
    dt = time.time() - t0
print(f"{cfg}{' + light' if with_light else ''}{f', write_batch_size {write_batch}' if write_batch else ''}: {n} segments in {dt:.2f} s = {n / dt:.3g} segments/s; packets {res.get('n_packets')}")
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
