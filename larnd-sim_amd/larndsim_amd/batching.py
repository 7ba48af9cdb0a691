"""
Host-side grouping of segments into the reference's simulation batches.

Restates (numpy only) the pieces of the reference driver that decide WHICH segments are simulated
together: ``swap_coordinates`` (cli/simulate_pixels.py:66-90), ``select_active_volume``
(larndsim/active_volume.py:4-46), ``TPCBatcher`` (larndsim/util/batching.py:17-67) and the
BATCH_SIZE sub-batch split (cli/simulate_pixels.py:902-905).  The result is one int32 batch id per
segment -- the sort key prefix of the device chain and the shard key for multi-GPU runs.
"""
import numpy as np

from . import consts


def swap_coordinates(tracks):
    """x <-> z swap between the edep-sim frame and the TPC_BORDERS frame (in place)."""
    for a, b in (("x_start", "z_start"), ("x_end", "z_end"), ("x", "z")):
        tmp = np.copy(tracks[a])
        tracks[a] = tracks[b]
        tracks[b] = tmp
    return tracks


def _inside(tracks, bound, which):
    return ((tracks['x_' + which] > bound[0, 0]) & (tracks['x_' + which] < bound[0, 1])
            & (tracks['y_' + which] > bound[1, 0]) & (tracks['y_' + which] < bound[1, 1])
            & (tracks['z_' + which] > bound[2, 0]) & (tracks['z_' + which] < bound[2, 1]))


def select_active_volume(track_seg, tpc_borders, i_module=-1):
    """Indices of segments whose start or end point lies strictly inside any TPC box."""
    tpc_borders = np.sort(np.asarray(tpc_borders), axis=-1)
    mask = np.zeros(track_seg.shape, dtype=bool)
    tpcs = range(tpc_borders.shape[0]) if i_module < 0 else range((i_module - 1) * 2, i_module * 2)
    for i in tpcs:
        mask |= _inside(track_seg, tpc_borders[i], 'end') | _inside(track_seg, tpc_borders[i], 'start')
    return np.nonzero(mask)[0]


def assign_batches(tracks, tpc_borders=None, event_separator=None, tpc_batch_size=None, batch_size=None):
    """Vectorised equivalent of iterating the reference's TPCBatcher (larndsim/util/batching.py:17-67) + the BATCH_SIZE
    sub-batch loop (tests/helpers.py holds the loop form it is checked against).

    Returns (batch_id int32[n], order int64[n], table) where ``batch_id[i]`` is the index of the
    non-empty (event, TPC-group, sub-batch) batch segment i is simulated in (-1 if in no batch),
    ``order`` is a stable permutation that makes batch ids non-decreasing (segments keep their
    original relative order inside a batch, like ``tracks[batch_mask]``), and ``table`` lists
    (event_id, tpc_group, sub_batch, n_segments) per batch id.
    """
    det, sim = consts.detector, consts.sim
    borders = np.sort(np.asarray(det.TPC_BORDERS if tpc_borders is None else tpc_borders), axis=-1)
    sep = sim.EVENT_SEPARATOR if event_separator is None else event_separator
    tbs = sim.EVENT_BATCH_SIZE if tpc_batch_size is None else tpc_batch_size
    bs = sim.BATCH_SIZE if batch_size is None else batch_size
    n = tracks.shape[0]
    n_tpc = borders.shape[0]
    n_groups = int(np.ceil(n_tpc / tbs))
    # first TPC group (in index order) that contains an endpoint of the segment
    group = np.full(n, -1, dtype=np.int64)
    for g in range(n_groups - 1, -1, -1):
        m = np.zeros(n, dtype=bool)
        for i in range(g * tbs, min((g + 1) * tbs, n_tpc)):
            m |= _inside(tracks, borders[i], 'end') | _inside(tracks, borders[i], 'start')
        group[m] = g
    events, ev_idx = np.unique(tracks[sep], return_inverse=True)
    key = np.where(group >= 0, ev_idx * n_groups + group, -1)
    order = np.argsort(key, kind='stable')
    skey = key[order]
    batch_sorted = np.full(n, -1, dtype=np.int32)
    table = []
    valid = skey >= 0
    if valid.any():
        first = int(np.argmax(valid))
        ks = skey[first:]
        starts = np.flatnonzero(np.r_[True, ks[1:] != ks[:-1]])
        ends = np.r_[starts[1:], len(ks)]
        bid = 0
        for s, e in zip(starts, ends):
            k = int(ks[s])
            for sub, o in enumerate(range(s, e, bs)):
                oe = min(o + bs, e)
                batch_sorted[first + o:first + oe] = bid
                table.append((events[k // n_groups], k % n_groups, sub, oe - o))
                bid += 1
    batch_id = np.empty(n, dtype=np.int32)
    batch_id[order] = batch_sorted
    # move unsimulated segments (-1) to the end so ids are non-decreasing over the simulated prefix
    order = np.argsort(np.where(batch_id < 0, np.iinfo(np.int32).max, batch_id), kind='stable')
    return batch_id, order, table


def shard_batches(table, world_size):
    """Greedy balance of batches over ranks by segment count, keeping batch order inside a rank.

    Returns rank int32[n_batches].  Batches are contiguous runs per rank (events are independent, so any
    assignment is valid; contiguity keeps each rank's segment range one slice)."""
    sizes = np.array([t[3] for t in table], dtype=np.int64)
    total = sizes.sum()
    rank = np.zeros(len(table), dtype=np.int32)
    if world_size <= 1 or len(table) == 0:
        return rank
    cum = np.cumsum(sizes) - sizes / 2.0
    rank[:] = np.minimum((cum * world_size / max(total, 1)).astype(np.int64), world_size - 1)
    return rank


def chunk_ranges(bid, max_segments):
    """[begin, end) ranges over batch-sorted segments, cut only at batch boundaries, each the smallest run of whole
    batches holding at least ``max_segments`` segments (the last one may be shorter).  One range = one chain launch: a
    batch (event x TPC group, cli/simulate_pixels.py:864-905) is never split, so a launch sees every segment of the
    pixels it sums."""
    n = len(bid)
    edges = np.flatnonzero(np.r_[True, bid[1:] != bid[:-1], True])
    out, b = [], 0
    for e in edges[1:]:
        if e - b >= max_segments or e == n:
            if e > b:
                out.append((int(b), int(e)))
            b = e
    return out
