"""Throughput of repeated passes of the charge chain over n segments; run one and two copies at the same time on one GPU to
see whether the kernels of two processes (tables stage: VALU-bound, correlation: L1 / latency-bound) overlap:
python tools/two_contexts.py module0 100000 6 [start_at_unix_time]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "larnd-sim_amd"), REPO, os.path.join(REPO, "tests"), os.path.join(REPO, "tools")):
    sys.path.insert(0, p)
from larndsim_amd import batching, synth            # noqa: E402
from larndsim_amd.chain import ChargeChain          # noqa: E402
import helpers as H                                 # noqa: E402
from qweights_check import prepared                 # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "module0"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 6
start_at = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
seg, bid = prepared(cfg, n, synth.SEED_BASE + 2, 5000)
ch = ChargeChain(H.response_for("survey"))
ch.upload(seg, bid)
ranges = batching.chunk_ranges(bid, 50000)


def run(steps):
    for _ in range(steps):
        ch.reset()
        ch.quench_drift()
        for b, e in ranges:
            ch.run(b, e, want_fractions=True)
    ch.synchronize()


run(1)
while time.time() < start_at:
    time.sleep(0.001)
t0 = time.time()
run(passes)
t1 = time.time()
print(f"pid {os.getpid()}: {n * passes / (t1 - t0):.4g} segments/s, {1e3 * (t1 - t0) / passes:.1f} ms per pass, from {t0:.3f} to {t1:.3f}", flush=True)
