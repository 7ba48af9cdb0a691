"""
Multi-GPU plumbing: one process per GPU, batches (event x TPC-group) sharded across ranks, one
all-gather of the compact hit rows to reassemble the per-pixel ADC output.

The exchange itself is RCCL behind the C-ABI (csrc/comm.hip, larndsim_amd/comm.py); this module only decides which batches
a rank owns.  No torch here: tests/test_cpu_dist.py rehearses the gather's two-step algorithm over gloo on its own.
"""
import os

import numpy as np

from . import batching


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_segments(batch_id, order, table, rank, world):
    """Segment indices (into the original array) and re-based batch ids of this rank's shard."""
    brank = batching.shard_batches(table, world)
    sorted_bid = batch_id[order]
    nsim = int((batch_id >= 0).sum())
    sel = np.zeros(len(order), dtype=bool)
    sel[:nsim] = brank[sorted_bid[:nsim]] == rank
    idx = order[sel]
    return idx, batch_id[idx]
