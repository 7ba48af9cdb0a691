"""
Multi-GPU plumbing: one process per GPU, batches (event x TPC-group) sharded across ranks, one
all-gather of the compact hit rows to reassemble the per-pixel ADC output.

torch.distributed is used only as the process-group / collective layer (backend "nccl" == RCCL over xGMI
on ROCm, "gloo" on CPU for the tests); the data path itself never touches torch.
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


class _DevRows:
    """Zero-copy view of a device buffer for torch via the CUDA array interface."""

    def __init__(self, ptr, n_words):
        self.__cuda_array_interface__ = {"shape": (n_words,), "typestr": "<i4", "data": (ptr, False), "version": 2}


def allgather_rows(rows, group=None):
    """All-gather variable-length int32 row blocks (torch tensors on the group's device).

    Two collectives: the row counts (tiny), then the padded payload.  Returns the concatenated rows of all
    ranks (on the same device) and the per-rank counts."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    width = rows.shape[1]
    pad = max(max(counts), 1)
    buf = torch.zeros((pad, width), dtype=rows.dtype, device=rows.device)
    buf[:rows.shape[0]] = rows
    out = torch.empty((world * pad, width), dtype=rows.dtype, device=rows.device)
    dist.all_gather_into_tensor(out, buf, group=group)
    parts = [out[r * pad:r * pad + counts[r]] for r in range(world)]
    return torch.cat(parts, dim=0), counts


def device_rows_as_tensor(ptr, n_rows, row_bytes, device):
    """Wrap the chain's compact hit rows (device memory owned by the ctx) as an int32 [n_rows, row_bytes/4] tensor."""
    import torch
    words = row_bytes // 4
    if n_rows == 0:
        return torch.zeros((0, words), dtype=torch.int32, device=device)
    t = torch.as_tensor(_DevRows(ptr, n_rows * words), device=device)
    return t.view(n_rows, words)
