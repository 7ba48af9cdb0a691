"""
LArPix packet stream and ``mc_packets_assn`` from the chain's per-pixel ADC arrays -- mirrors ``fee.export_to_hdf5``
(larndsim/fee.py:84-356) without larpix-control: the packets are produced directly as the structured array larpix-control's
HDF5 format stores in its ``packets`` dataset.

What is the reference's (golden-tested against ``fee.export_to_hdf5`` run under stand-in packet classes,
tests/golden/packets_*.npz, and against a hit-by-hit restatement of its loop kept with the tests, tests/packets_loop.py): which hits become packets (a pixel's slots stop at the first ADC <= digitize(0),
fee.py:159-162,281-282), ``time_tick = floor(t / CLOCK_CYCLE + event_t0)`` with the rollover rule (:164-183), the pixel ->
(io_group, io_channel, chip, channel) mapping through tile map / tile orientation / pixel layout (:150-157,238-255), the
timestamp / sync / trigger packets inserted around the data packets (:187-230,267-277), and the association rows
(:284-344: fractions sorted descending, the ASSOCIATION_COUNT_TO_STORE largest kept; trajectories summed per id).

What is third-party (module ``larpix``, larpix-control, not under the reference tree; restated from its published
format, unpinned): the dataset dtype (format version 2.4), the packet type codes (0 data, 4 timestamp, 6 sync,
7 trigger) and Packet_v2's odd parity over its 64 bits.
"""
import numpy as np

from . import consts

# larpix.format.hdf5format, version 2.4, dataset 'packets'
packets_dtype = np.dtype([
    ('io_group', 'u1'), ('io_channel', 'u1'), ('chip_id', 'u1'), ('packet_type', 'u1'), ('downstream_marker', 'u1'),
    ('parity', 'u1'), ('valid_parity', 'u1'), ('channel_id', 'u1'), ('timestamp', 'u8'), ('dataword', 'u1'),
    ('trigger_type', 'u1'), ('local_fifo', 'u1'), ('shared_fifo', 'u1'), ('register_address', 'u1'),
    ('register_data', 'u1'), ('direction', 'u1'), ('local_fifo_events', 'u1'), ('shared_fifo_events', 'u2'),
    ('counter', 'u4'), ('fifo_diagnostics_enabled', 'u1'), ('first_packet', 'u1'), ('receipt_timestamp', 'u4')])
DATA, TIMESTAMP, SYNC, TRIGGER = 0, 4, 6, 7


def assn_dtype():
    n = consts.sim.ASSOCIATION_COUNT_TO_STORE
    return np.dtype([('event_ids', '(1,)i8'), ('segment_ids', f'({n},)i8'), ('fraction', f'({n},)f8'),
                     ('file_traj_ids', f'({n},)i8'), ('fraction_traj', f'({n},)f8')])


def _digitize0():
    """fee.digitize(0): the pedestal code (fee.py:499-515)."""
    d, u = consts.detector, consts.units
    v = max(0 * d.GAIN * u.mV / u.e + d.V_PEDESTAL * u.mV - d.V_CM * u.mV, 0) * d.ADC_COUNTS / (d.V_REF * u.mV - d.V_CM * u.mV)
    return min(np.around(v), d.ADC_COUNTS - 1)


def _other_row(ptype, io_group, timestamp, trigger_type=0):
    return (io_group, 0, 0, ptype, 0, 0, 1, 0, int(timestamp), 0, trigger_type, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0)


def _empty_assn(n):
    """association rows of packets that carry no charge (fee.py:378-418, 441-470)"""
    ds = np.empty(n, dtype=assn_dtype())
    ds['event_ids'] = -1
    ds['segment_ids'] = -1
    ds['fraction'] = 0
    ds['file_traj_ids'] = -1
    ds['fraction_traj'] = 0
    return ds


def get_trig_io():
    """io_group of the PACMAN the light / beam trigger is forwarded to (fee.py:30-38)"""
    return 2 if consts.light.LIGHT_TRIG_MODE == 0 else 1


_READOUT_LUT_CACHE = {}


def _readout_luts():
    """Array forms of the readout maps (pixel in tile -> chip / channel, (tile, chip) -> io, tile orientation, tile map),
    rebuilt when consts.detector carries other objects."""
    d = consts.detector
    # Per batch (the driver calls build_packets once per batch): the same four map objects as last time -- the cache holds
    # references to them, so another configuration's maps cannot turn up under their ids -- and the same tile size.  Maps that were
    # replaced are compared by content (0.9 ms for a 4900-pixel tile: a tenth of a batch's packets when it was done every call);
    # editing one of the dicts in place is not noticed, load another configuration or assign a new dict instead.
    objs = (d.PIXEL_CONNECTION_DICT, d.TILE_CHIP_TO_IO, d.TILE_ORIENTATIONS, d.TILE_MAP)
    tile_size = tuple(int(v) for v in d.N_PIXELS_PER_TILE)
    hit = _READOUT_LUT_CACHE.get("luts")
    if hit is not None and hit[2] == tile_size and all(a is b for a, b in zip(hit[3], objs)):
        return hit[1]
    key = (hash(frozenset(d.PIXEL_CONNECTION_DICT.items())), repr(d.TILE_CHIP_TO_IO), repr(d.TILE_ORIENTATIONS), repr(d.TILE_MAP), tile_size)
    if hit is not None and hit[0] == key:
        _READOUT_LUT_CACHE["luts"] = (key, hit[1], tile_size, objs)
        return hit[1]
    ntx, nty = int(d.N_PIXELS_PER_TILE[0]), int(d.N_PIXELS_PER_TILE[1])
    chip_lut = np.full((ntx, nty), -1, dtype=np.int64)
    chan_lut = np.full((ntx, nty), -1, dtype=np.int64)
    for (px, py), (chip, channel) in d.PIXEL_CONNECTION_DICT.items():
        if 0 <= px < ntx and 0 <= py < nty:
            chip_lut[px, py], chan_lut[px, py] = chip, channel
    tile_map = np.asarray(d.TILE_MAP, dtype=np.int64)                   # [anode][tile_x][tile_y]
    n_tile = int(max(max(d.TILE_ORIENTATIONS), max(d.TILE_CHIP_TO_IO) if d.TILE_CHIP_TO_IO else 0, tile_map.max())) + 1
    n_chip = int(max([max(v) for v in d.TILE_CHIP_TO_IO.values() if v] + [int(chip_lut.max())])) + 1
    io_lut = np.full((n_tile, n_chip), -1, dtype=np.int64)
    for tile, chips in d.TILE_CHIP_TO_IO.items():
        for chip, io in chips.items():
            io_lut[tile, chip] = io
    flip_x = np.zeros(n_tile, dtype=bool)
    flip_y = np.zeros(n_tile, dtype=bool)
    has_orient = np.zeros(n_tile, dtype=bool)
    for tile, axes in d.TILE_ORIENTATIONS.items():
        flip_x[tile], flip_y[tile], has_orient[tile] = axes[2] < 0, axes[1] < 0, True
    luts = dict(chip=chip_lut, chan=chan_lut, io=io_lut, flip_x=flip_x, flip_y=flip_y, has_orient=has_orient, tile_map=tile_map)
    _READOUT_LUT_CACHE["luts"] = (key, luts, tile_size, objs)
    return luts


def _parity_array(chip_id, channel_id, timestamp, first_packet, dataword):
    """_data_parity over arrays."""
    word = ((chip_id.astype(np.uint64) & np.uint64(0xFF)) << np.uint64(2)) | ((channel_id.astype(np.uint64) & np.uint64(0x3F)) << np.uint64(10)) \
        | ((timestamp.astype(np.uint64) & np.uint64(0x7FFFFFFF)) << np.uint64(16)) | (np.uint64(first_packet & 1) << np.uint64(47)) \
        | ((dataword.astype(np.uint64) & np.uint64(0xFF)) << np.uint64(48))
    bits = np.unpackbits(word.view(np.uint8).reshape(-1, 8), axis=1).sum(axis=1)
    return (1 - (bits % 2)).astype(np.uint8)


def build_packets(event_id_list, adc_list, adc_ticks_list, unique_pix, current_fractions, track_ids, traj_ids,
                  event_start_times, light_trigger_times=None, light_trigger_event_id=None, light_trigger_modules=None,
                  bad_channels=None, i_mod=-1):
    """``fee.export_to_hdf5`` up to the file write (larndsim/fee.py:84-356): returns (packets, mc_packets_assn).

    Same arguments and the same bytes as the hit loop of the reference (``tests/packets_loop.py``; tested on the goldens and on
    random inputs), computed with
    array operations: the per-hit state of the reference's loop is (a) the clock rollover, a running count of
    CLOCK_RESET_PERIODs subtracted from every later row (:164-183) -- found by re-evaluating all hits once per rollover --,
    (b) "event changed" and "timestamp changed" flags (:187,267), comparisons with the previous hit, and (c) positions in
    the output stream, a cumulative sum of the packets each hit emits.  Only the per-event packets (timestamp / sync /
    light triggers) and rows whose trajectory sums have 8 or more terms (numpy's pairwise order) are handled in Python."""
    d, light, sim, units = consts.detector, consts.light, consts.sim, consts.units
    if not d.PIXEL_CONNECTION_DICT:
        raise RuntimeError("the pixel layout (chip / channel map) is not loaded: packets need consts.load_properties(...) on "
                           "the detector / pixel-layout YAML files or a snapshot that carries the readout map")
    L = _readout_luts()
    io_groups = np.unique(np.array(list(d.MODULE_TO_IO_GROUPS.values())))
    io_groups = io_groups if i_mod < 0 else io_groups[(i_mod - 1) * 2: i_mod * 2]
    event_id_list = np.asarray(event_id_list)
    adc_list = np.asarray(adc_list)
    adc_ticks_list = np.asarray(adc_ticks_list)
    unique_pix = np.asarray(unique_pix)
    track_ids, traj_ids, current_fractions = np.asarray(track_ids), np.asarray(traj_ids), np.asarray(current_fractions)
    n_trk, n_trj, n_frac = track_ids.shape[1], traj_ids.shape[1], current_fractions.shape[2]
    if not (n_trk == n_trj == n_frac == sim.MAX_TRACKS_PER_PIXEL):
        # the reference stacks the per-packet id / fraction lists with np.array (fee.py:297-300): rows of the inserted
        # timestamp packets are MAX_TRACKS_PER_PIXEL wide (:271-276), so any other width is a ragged list there -- ValueError
        raise ValueError(f"track_ids [U][{n_trk}], traj_ids [U][{n_trj}] and current_fractions [U][A][{n_frac}] must all hold "
                         f"MAX_TRACKS_PER_PIXEL = {sim.MAX_TRACKS_PER_PIXEL} slots per pixel")
    empty = (np.zeros(0, dtype=packets_dtype), np.zeros(0, dtype=assn_dtype()))
    if len(adc_list) == 0:
        return empty
    unique_events, unique_events_inv = np.unique(event_id_list[..., 0], return_inverse=True)
    event_start_times = np.asarray(event_start_times)
    base = (event_start_times[unique_events_inv] / d.CLOCK_CYCLE).astype(int)            # event_start_time_list
    light_trigger_times = np.empty((0,)) if light_trigger_times is None else np.asarray(light_trigger_times)
    light_trigger_event_id = np.empty((0,), dtype=int) if light_trigger_event_id is None else np.asarray(light_trigger_event_id)
    light_trigger_modules = np.empty((0,)) if light_trigger_modules is None else np.asarray(light_trigger_modules)
    ped = _digitize0()
    CRP = int(d.CLOCK_RESET_PERIOD)
    A = adc_list.shape[1]

    # ---- rows that enter the hit loop, and their slots up to the first ADC <= pedestal --------------------------------------
    nx, ny = int(d.N_PIXELS[0]), int(d.N_PIXELS[1])
    pid = unique_pix.astype(np.int64)
    pix_x, pix_y, plane_id = pid % nx, (pid // nx) % ny, pid // (nx * ny)
    module_id = plane_id // 2 + 1
    known_module = np.isin(module_id, np.array(list(d.MODULE_TO_IO_GROUPS.keys()), dtype=np.int64))
    above = adc_list > ped
    n_valid = np.where(above.all(axis=1), A, np.argmin(above, axis=1))
    rows_in = np.flatnonzero(above[:, 0] & known_module)
    counts = n_valid[rows_in]
    n_hit = int(counts.sum())
    if n_hit == 0:
        return empty
    hit_row = np.repeat(rows_in, counts)
    hit_slot = np.arange(n_hit) - np.repeat(np.cumsum(counts) - counts, counts)

    # ---- clock rollover: `off` reset periods have been subtracted from this and every later row when a hit is reached ---------
    t_cc = adc_ticks_list[hit_row, hit_slot] / d.CLOCK_CYCLE
    base_h = base[hit_row].astype(np.int64)
    off = np.zeros(n_hit, dtype=np.int64)
    while True:
        event_t0 = base_h - off * CRP
        time_tick = np.floor(t_cc + event_t0).astype(np.int64)
        viol = (event_t0 > CRP - 1) | (time_tick > CRP - 1)
        if not viol.any():
            break
        off[int(np.argmax(viol)):] += 1
    # row 0 of event_start_time_list as the timestamp packets see it: rollovers hit it only while row 0 is processed
    in_row0 = hit_row == 0
    off_row0_final = int(off[in_row0][-1]) if in_row0.any() else 0
    row0_value = int(base[0]) - np.where(in_row0, off, off_row0_final) * CRP
    event_t0 = event_t0 % CRP
    time_tick = time_tick % CRP
    event = event_id_list[hit_row, hit_slot]

    # ---- readout address of every row; hits of rows without one are dropped after the event bookkeeping -----------------------
    ntx, nty = L["chip"].shape
    tile_x, tile_y = pix_x[rows_in] // ntx, pix_y[rows_in] // nty
    anode = plane_id[rows_in] % 2
    tm = L["tile_map"]
    in_map = (anode < tm.shape[0]) & (tile_x < tm.shape[1]) & (tile_y < tm.shape[2])
    if not in_map.all():
        raise IndexError("pixel outside the tile map")                   # the loop form raises on the same input
    tile_id = tm[anode, tile_x, tile_y]
    px, py = pix_x[rows_in] % ntx, pix_y[rows_in] % nty
    px = np.where(L["flip_x"][tile_id], ntx - px - 1, px)
    py = np.where(L["flip_y"][tile_id], nty - py - 1, py)
    chip, channel = L["chip"][px, py], L["chan"][px, py]
    ok = (chip >= 0) & L["has_orient"][tile_id]       # (a tile without an orientation raises KeyError inside the loop form's try)
    io = np.where(ok & (chip < L["io"].shape[1]), L["io"][np.minimum(tile_id, L["io"].shape[0] - 1),
                                                        np.clip(chip, 0, L["io"].shape[1] - 1)], -1)
    ok &= io >= 0
    io_group_idx, io_channel = io // 1000, io % 1000
    mod_groups = {int(m): list(g) for m, g in d.MODULE_TO_IO_GROUPS.items()}
    io_group = np.zeros(len(rows_in), dtype=np.int64)
    for m, g in mod_groups.items():
        sel = ok & (module_id[rows_in] == m)
        if sel.any():
            io_group[sel] = np.asarray(g, dtype=np.int64)[io_group_idx[sel] - 1]
    if bad_channels:
        bad = set()
        for chip_key, chans in bad_channels.items():
            for ch in chans:
                bad.add("%s:%i" % (chip_key, int(ch)))
        if bad:
            for i in np.flatnonzero(ok):
                if "%i-%i-%i:%i" % (io_group[i], io_channel[i], chip[i], channel[i]) in bad:
                    ok[i] = False
    row_pos = np.repeat(np.arange(len(rows_in)), counts)                 # index into the per-row arrays above
    passes = ok[row_pos]

    # ---- what every hit emits ------------------------------------------------------------------------------------------------
    ev_change = np.zeros(n_hit, dtype=bool)
    if light.LIGHT_TRIG_MODE != 1:
        ev_change[0] = event[0] != -1
        ev_change[1:] = event[1:] != event[:-1]
    ev_rows = {}            # hit index -> list of (ptype, io_group, timestamp, trigger_type)
    for h in np.flatnonzero(ev_change):
        lst = []
        for g in io_groups:
            lst.append((TIMESTAMP, g, int(event_start_times[unique_events_inv[hit_row[h]]] * units.mus / units.s), 0))
            lst.append((SYNC, g, int(time_tick[h]), ord('S')))
        trig_mask = light_trigger_event_id == event[h]
        if trig_mask.any():
            for t_trig, module_trig in zip(light_trigger_times[trig_mask], light_trigger_modules[trig_mask]):
                t_trig = int(np.floor(t_trig / d.CLOCK_CYCLE + event_t0[h])) % CRP
                if light.LIGHT_TRIG_MODE == 0:
                    for g in d.MODULE_TO_IO_GROUPS[int(module_trig)]:
                        lst.append((TRIGGER, g, t_trig, 2))
        ev_rows[int(h)] = lst
    n_ev = np.zeros(n_hit, dtype=np.int64)
    for h, lst in ev_rows.items():
        n_ev[h] = len(lst)
    tick_change = np.zeros(n_hit, dtype=bool)
    ph = np.flatnonzero(passes)
    if len(ph):
        tt = time_tick[ph]
        tick_change[ph[0]] = tt[0] != -1
        tick_change[ph[1:]] = tt[1:] != tt[:-1]
    n_out = n_ev + tick_change + passes
    start = np.cumsum(n_out) - n_out
    n_rows = int(n_out.sum())
    if n_rows == 0:
        return empty

    packets = np.zeros(n_rows, dtype=packets_dtype)
    packets['valid_parity'] = 1
    is_data = np.zeros(n_rows, dtype=bool)
    for h, lst in ev_rows.items():
        for k, (ptype, g, ts, trig) in enumerate(lst):
            r = packets[start[h] + k: start[h] + k + 1]
            r['io_group'], r['packet_type'], r['timestamp'], r['trigger_type'] = g, ptype, ts, trig
    th = np.flatnonzero(tick_change)
    pos = start[th] + n_ev[th]
    packets['io_group'][pos] = io_group[row_pos[th]]
    packets['packet_type'][pos] = TIMESTAMP
    packets['timestamp'][pos] = np.floor(row0_value[th] * d.CLOCK_CYCLE * units.mus / units.s).astype(np.int64)
    pos = start[ph] + n_ev[ph] + tick_change[ph]
    is_data[pos] = True
    rp = row_pos[ph]
    dataword = adc_list[hit_row[ph], hit_slot[ph]].astype(np.int64)
    packets['io_group'][pos] = io_group[rp]
    packets['io_channel'][pos] = io_channel[rp]
    packets['chip_id'][pos] = chip[rp]
    packets['packet_type'][pos] = DATA
    packets['parity'][pos] = _parity_array(chip[rp], channel[rp], time_tick[ph], 1, dataword)
    packets['channel_id'][pos] = channel[rp]
    packets['timestamp'][pos] = time_tick[ph]
    packets['dataword'][pos] = dataword
    packets['first_packet'][pos] = 1
    packets['receipt_timestamp'][pos] = time_tick[ph]

    # ---- mc_packets_assn (fee.py:284-344) -----------------------------------------------------------------------------------
    # Rows of the inserted packets hold the fill values whatever the sort does with them; the work is on the data rows only.
    n_keep = sim.ASSOCIATION_COUNT_TO_STORE
    ds = np.zeros(n_rows, dtype=assn_dtype())
    ds['event_ids'] = -1
    ds['segment_ids'] = -1
    ds['file_traj_ids'] = -1
    n_data = len(ph)
    if n_data == 0:
        return packets, ds
    F = current_fractions[hit_row[ph], hit_slot[ph]].astype(np.float64, copy=False)
    # (stable: equal fractions -- the exact zeros of track slots whose current missed the hit, beside the unused slots' zeros --
    # keep descending slot order like the native hit loop, csrc/packets.hip; the reference's default argsort leaves their order to
    # numpy's build)
    frac_order = np.flip(np.argsort(F, axis=1, kind="stable"), axis=1)
    head = frac_order[:, :n_keep]
    w = head.shape[1]
    hr = hit_row[ph][:, None]
    seg_head = track_ids[hr, head]
    ass_fractions = np.take_along_axis(F, frac_order, axis=1)
    ass_trajectory_ids = traj_ids[hr, frac_order]
    out_seg = np.full((n_data, n_keep), -1, dtype=np.int64)
    out_frac = np.zeros((n_data, n_keep))
    out_seg[:, :w] = seg_head
    out_frac[:, :w] = ass_fractions[:, :w]
    # trajectories: per row the ids in ascending order with the sum of their fractions.  The reference sums every id's terms
    # with np.sum in fraction order and stores f4; np.sum's association of a handful of f8 terms depends on numpy's SIMD
    # build, so f8 agreement to the last bit is not defined -- sums here run left to right (reduceat) and are rounded to f4
    # like the reference's (a differing f8 last bit moves the f4 value once in ~1e9 sums).
    out_tid = np.full((n_data, n_keep), -1, dtype=np.int64)
    out_tfr = np.zeros((n_data, n_keep), dtype=np.float32)
    mask = ass_trajectory_ids > -1
    r, _ = np.nonzero(mask)                                      # row-major: inside a row the fraction order is kept
    if len(r):
        ids = ass_trajectory_ids[mask].astype(np.int64)
        fr = ass_fractions[mask]
        idx = np.lexsort((ids, r))                               # stable: by row, then id, then fraction order
        r2, ids2, fr2 = r[idx], ids[idx], fr[idx]
        newg = np.ones(len(r2), dtype=bool)
        newg[1:] = (r2[1:] != r2[:-1]) | (ids2[1:] != ids2[:-1])
        starts = np.flatnonzero(newg)
        sums = np.add.reduceat(fr2, starts)
        g_row = r2[starts]
        first_of_row = np.ones(len(starts), dtype=bool)
        first_of_row[1:] = g_row[1:] != g_row[:-1]
        g_idx = np.arange(len(starts))
        local = g_idx - np.maximum.accumulate(np.where(first_of_row, g_idx, 0))
        sel = local < n_keep
        out_tid[g_row[sel], local[sel]] = ids2[starts][sel].astype(np.int32)      # the reference's id array is int32
        out_tfr[g_row[sel], local[sel]] = sums[sel]
    ds['segment_ids'][pos] = out_seg
    ds['fraction'][pos] = out_frac
    ds['file_traj_ids'][pos] = out_tid
    ds['fraction_traj'][pos] = out_tfr
    ds['event_ids'][pos, 0] = event[ph]
    return packets, ds


def _row_addresses(pid, L, bad_channels):
    """pixel id -> (ok, io_group, io_channel, chip, channel, module id) as the hit loop finds them (fee.py:150-157,238-255):
    the same expressions as build_packets, on any array of pixel ids."""
    d = consts.detector
    nx, ny = int(d.N_PIXELS[0]), int(d.N_PIXELS[1])
    pix_x, pix_y, plane_id = pid % nx, (pid // nx) % ny, pid // (nx * ny)
    module_id = plane_id // 2 + 1
    ntx, nty = L["chip"].shape
    tile_x, tile_y = pix_x // ntx, pix_y // nty
    anode = plane_id % 2
    tm = L["tile_map"]
    in_map = (anode < tm.shape[0]) & (tile_x < tm.shape[1]) & (tile_y < tm.shape[2])
    if not in_map.all():
        raise IndexError("pixel outside the tile map")
    tile_id = tm[anode, tile_x, tile_y]
    px, py = pix_x % ntx, pix_y % nty
    px = np.where(L["flip_x"][tile_id], ntx - px - 1, px)
    py = np.where(L["flip_y"][tile_id], nty - py - 1, py)
    chip, channel = L["chip"][px, py], L["chan"][px, py]
    ok = (chip >= 0) & L["has_orient"][tile_id]
    io = np.where(ok & (chip < L["io"].shape[1]), L["io"][np.minimum(tile_id, L["io"].shape[0] - 1),
                                                        np.clip(chip, 0, L["io"].shape[1] - 1)], -1)
    ok &= io >= 0
    io_group_idx, io_channel = io // 1000, io % 1000
    io_group = np.zeros(len(pid), dtype=np.int64)
    for m, g in d.MODULE_TO_IO_GROUPS.items():
        sel = ok & (module_id == int(m))
        if sel.any():
            io_group[sel] = np.asarray(list(g), dtype=np.int64)[io_group_idx[sel] - 1]
    if bad_channels:
        bad = set()
        for chip_key, chans in bad_channels.items():
            for ch in chans:
                bad.add("%s:%i" % (chip_key, int(ch)))
        if bad:
            for i in np.flatnonzero(ok):
                if "%i-%i-%i:%i" % (io_group[i], io_channel[i], chip[i], channel[i]) in bad:
                    ok[i] = False
    return ok, io_group, io_channel, chip, channel, module_id


def build_packets_compact(row_event, row_pixel, row_nh, row_nt, hit_adc, hit_tick, hit_frac, trk_segment, trk_traj,
                          event_start_times, light_trigger_times=None, light_trigger_event_id=None, light_trigger_modules=None,
                          bad_channels=None, i_mod=-1):
    """``build_packets`` on the compact rows of the chain (``ChargeChain.download_compact``) through the native hit loop
    (``ldsim_packets_build``, csrc/packets.hip) -- no dense [pixel][30][50] fraction array is ever formed.

    Rows in export order: ``row_event`` (event id), ``row_pixel`` (pixel id), ``row_nh`` hits, ``row_nt`` filled track slots per
    row -- row 0 of the export included even when it holds no hit (the exporter's clock state starts from it).  Hits pixel after
    pixel, slot 0 up: ``hit_adc`` (ADC code), ``hit_tick`` (adc_ticks_list), ``hit_frac`` (per hit one fraction per track slot of
    its row).  Track slots pixel after pixel: ``trk_segment`` / ``trk_traj`` (the exporter's track_ids / traj_ids values).
    ``event_start_times``: one per unique event id of the rows, ascending, like build_packets.  Same bytes as build_packets on the
    dense form of the same rows (tests/test_cpu_host.py; equal fractions of real track slots fall in descending slot order)."""
    import ctypes as C
    from . import lib
    from .abi import LdsimPacketsIn
    d, light, sim, units = consts.detector, consts.light, consts.sim, consts.units
    if not d.PIXEL_CONNECTION_DICT:
        raise RuntimeError("the pixel layout (chip / channel map) is not loaded: packets need consts.load_properties(...) on "
                           "the detector / pixel-layout YAML files or a snapshot that carries the readout map")
    L = _readout_luts()
    row_event = np.ascontiguousarray(row_event, dtype=np.int64)
    row_pixel = np.asarray(row_pixel, dtype=np.int64)
    row_nh = np.asarray(row_nh, dtype=np.int64)
    row_nt = np.asarray(row_nt, dtype=np.int64)
    hit_adc = np.ascontiguousarray(hit_adc, dtype=np.int32)
    hit_tick = np.ascontiguousarray(hit_tick, dtype=np.float64)
    hit_frac = np.ascontiguousarray(hit_frac, dtype=np.float64)
    trk_segment = np.ascontiguousarray(trk_segment, dtype=np.int64)
    trk_traj = np.ascontiguousarray(trk_traj, dtype=np.int64)
    n_keep = int(sim.ASSOCIATION_COUNT_TO_STORE)
    empty = (np.zeros(0, dtype=packets_dtype), np.zeros(0, dtype=assn_dtype()))
    R = len(row_event)
    if R == 0:
        return empty
    io_groups = np.unique(np.array(list(d.MODULE_TO_IO_GROUPS.values())))
    io_groups = io_groups if i_mod < 0 else io_groups[(i_mod - 1) * 2: i_mod * 2]
    unique_events, inv = np.unique(row_event, return_inverse=True)
    event_start_times = np.asarray(event_start_times, dtype=np.float64)
    base = (event_start_times[inv] / d.CLOCK_CYCLE).astype(int).astype(np.int64)
    ts_s = (event_start_times[inv] * units.mus / units.s).astype(np.int64)
    ped = _digitize0()
    hit0 = np.r_[0, np.cumsum(row_nh)]
    trk0 = np.r_[0, np.cumsum(row_nt)]
    frac0 = np.r_[0, np.cumsum(row_nh * row_nt)]
    # slots up to the first ADC <= pedestal (fee.py:159-162,281-282)
    n_valid = np.zeros(R, dtype=np.int64)
    if len(hit_adc):
        row_of_hit = np.repeat(np.arange(R), row_nh)
        slot = np.arange(len(hit_adc)) - hit0[row_of_hit]
        stop = np.where(hit_adc <= ped, slot, np.int64(1 << 40))
        first_stop = np.full(R, np.int64(1 << 40))
        np.minimum.at(first_stop, row_of_hit, stop)
        n_valid = np.minimum(row_nh, first_stop)
    module_known = np.isin(row_pixel // (int(d.N_PIXELS[0]) * int(d.N_PIXELS[1])) // 2 + 1,
                           np.array(list(d.MODULE_TO_IO_GROUPS.keys()), dtype=np.int64))
    rows_in = np.flatnonzero((n_valid > 0) & module_known)
    if len(rows_in) == 0:
        return empty
    ok, io_group, io_channel, chip, channel, _ = _row_addresses(row_pixel[rows_in], L, bad_channels)
    nv = n_valid[rows_in]
    # the C side walks [row_hit0[i], row_hit0[i + 1]) of the hit arrays: the rows' leading slots, gathered
    take = np.repeat(hit0[rows_in], nv) + (np.arange(int(nv.sum())) - np.repeat(np.cumsum(nv) - nv, nv))
    a_adc = np.ascontiguousarray(hit_adc[take])
    a_tick = np.ascontiguousarray(hit_tick[take])
    r_hit0 = np.ascontiguousarray(np.r_[0, np.cumsum(nv)], dtype=np.int64)
    r_trk0_lo = trk0[rows_in]
    nt_in = row_nt[rows_in]
    # track slots: contiguous per row already; pass offsets into the full arrays through a gathered copy of the offsets
    # (row_trk0 must be one array of n + 1 ascending offsets: gather the slots of the rows that enter)
    ttake = np.repeat(r_trk0_lo, nt_in) + (np.arange(int(nt_in.sum())) - np.repeat(np.cumsum(nt_in) - nt_in, nt_in))
    a_seg = np.ascontiguousarray(trk_segment[ttake])
    a_trj = np.ascontiguousarray(trk_traj[ttake])
    r_trk0 = np.ascontiguousarray(np.r_[0, np.cumsum(nt_in)], dtype=np.int64)
    r_frac0 = np.ascontiguousarray(frac0[rows_in], dtype=np.int64)
    lt_t = np.ascontiguousarray(np.empty(0) if light_trigger_times is None else light_trigger_times, dtype=np.float64).ravel()
    lt_e = np.ascontiguousarray(np.empty(0, dtype=np.int64) if light_trigger_event_id is None else light_trigger_event_id, dtype=np.int64).ravel()
    lt_m = np.ascontiguousarray(np.empty(0) if light_trigger_modules is None else light_trigger_modules).ravel().astype(np.int64)
    n_trig_rows = 0
    if light.LIGHT_TRIG_MODE == 0 and len(lt_e):
        per_mod = {int(m): len(g) for m, g in d.MODULE_TO_IO_GROUPS.items()}
        n_trig_rows = int(sum(per_mod.get(int(m), 0) for m in lt_m))
    mods = sorted(int(m) for m in d.MODULE_TO_IO_GROUPS)
    mod_ids = np.array(mods, dtype=np.int64)
    mg = [np.asarray(list(d.MODULE_TO_IO_GROUPS[m]), dtype=np.int32) for m in mods]
    mod_g0 = np.ascontiguousarray(np.r_[0, np.cumsum([len(g) for g in mg])], dtype=np.int32)
    mod_g = np.ascontiguousarray(np.concatenate(mg) if mg else np.zeros(0, dtype=np.int32), dtype=np.int32)
    iog = np.ascontiguousarray(io_groups, dtype=np.int32)
    cap = 2 * int(nv.sum()) + len(unique_events) * 2 * len(iog) + n_trig_rows + 16
    pk = np.zeros(cap, dtype=packets_dtype)
    ds = np.zeros(cap, dtype=assn_dtype())
    Lb = lib.load()
    assert pk.dtype.itemsize == Lb.ldsim_packets_row_bytes() and ds.dtype.itemsize == Lb.ldsim_packets_assn_row_bytes(C.c_int32(n_keep))
    keep = dict(ev=np.ascontiguousarray(row_event[rows_in]), base=np.ascontiguousarray(base[rows_in]), ts=np.ascontiguousarray(ts_s[rows_in]),
                ok=np.ascontiguousarray(ok, dtype=np.uint8), iog=np.ascontiguousarray(io_group, dtype=np.int32),
                ioc=np.ascontiguousarray(io_channel, dtype=np.int32), chip=np.ascontiguousarray(chip, dtype=np.int32),
                chan=np.ascontiguousarray(channel, dtype=np.int32))
    P = LdsimPacketsIn()
    P.n_rows = len(rows_in)
    P.row_event, P.row_base, P.row_ts_s = (a.ctypes.data for a in (keep["ev"], keep["base"], keep["ts"]))
    P.row_ok, P.row_io_group, P.row_io_channel = keep["ok"].ctypes.data, keep["iog"].ctypes.data, keep["ioc"].ctypes.data
    P.row_chip, P.row_channel = keep["chip"].ctypes.data, keep["chan"].ctypes.data
    P.row_hit0, P.row_trk0, P.row_frac0 = r_hit0.ctypes.data, r_trk0.ctypes.data, r_frac0.ctypes.data
    P.hit_adc, P.hit_tick, P.hit_frac = a_adc.ctypes.data, a_tick.ctypes.data, hit_frac.ctypes.data
    P.trk_segment, P.trk_traj = a_seg.ctypes.data, a_trj.ctypes.data
    P.base0 = int(base[0])
    P.first_row_is_row0 = int(rows_in[0] == 0)
    P.light_trig_mode = int(light.LIGHT_TRIG_MODE)
    P.n_trig = len(lt_e)
    P.trig_time, P.trig_event, P.trig_module = lt_t.ctypes.data, lt_e.ctypes.data, lt_m.ctypes.data
    P.n_io_groups, P.io_groups = len(iog), iog.ctypes.data
    P.n_modules, P.module_ids, P.module_group0, P.module_groups = len(mods), mod_ids.ctypes.data, mod_g0.ctypes.data, mod_g.ctypes.data
    P.clock_reset_period = int(d.CLOCK_RESET_PERIOD)
    P.clock_cycle, P.mus, P.s = float(d.CLOCK_CYCLE), float(units.mus), float(units.s)
    P.n_keep, P.max_tracks = n_keep, int(sim.MAX_TRACKS_PER_PIXEL)
    n = int(Lb.ldsim_packets_build(C.byref(P), pk.ctypes.data_as(C.c_void_p), ds.ctypes.data_as(C.c_void_p), C.c_int64(cap)))
    if n < 0:
        raise lib.LdsimError(f"ldsim error {n}: {Lb.ldsim_last_error().decode()}")
    return pk[:n], ds[:n]


def compact_to_rows(c, event_of_batch, first_segment_of_batch, segment_ids, traj_ids, rows=None):
    """The arguments of ``build_packets_compact`` from ``ChargeChain.download_compact()``'s arrays (rows ``rows`` = a slice of its hit
    pixels, default all): events from the rows' batch ids, track slots mapped to segment / trajectory ids (a batch's track slots
    count its segments from its first one, cli/simulate_pixels.py:1019-1026)."""
    hp = c["hit_pixels"]
    nh_all, nt_all = hp[:, 3].astype(np.int64), (hp[:, 4] & 255).astype(np.int64)
    h0 = np.r_[0, np.cumsum(nh_all)]
    t0 = np.r_[0, np.cumsum(nt_all)]
    f0 = np.r_[0, np.cumsum(nh_all * nt_all)]
    a, b = (0, len(hp)) if rows is None else rows
    batch = hp[a:b, 2].astype(np.int64)
    seg_idx = np.repeat(np.asarray(first_segment_of_batch, dtype=np.int64)[batch], nt_all[a:b]) + c["track_segments"][t0[a]:t0[b]]
    return dict(row_event=np.asarray(event_of_batch, dtype=np.int64)[batch], row_pixel=hp[a:b, 1], row_nh=nh_all[a:b], row_nt=nt_all[a:b],
                hit_adc=c["hit_rows"]["adc"][h0[a]:h0[b]], hit_tick=c["hit_rows"]["tick"][h0[a]:h0[b]],
                hit_frac=c["fractions"][f0[a]:f0[b]], trk_segment=np.asarray(segment_ids, dtype=np.int64)[seg_idx],
                trk_traj=np.asarray(traj_ids, dtype=np.int64)[seg_idx])


def build_sync_packets(sync_times, i_mod=-1):
    """``fee.export_sync_to_hdf5`` up to the file write (fee.py:361-424): one sync packet per io_group and sync time [us]."""
    d = consts.detector
    io_groups = np.unique(np.array(list(d.MODULE_TO_IO_GROUPS.values())))
    io_groups = d.MODULE_TO_IO_GROUPS[i_mod] if i_mod > 0 else io_groups
    rows = []
    for sync_tick in np.asarray(sync_times) / d.CLOCK_CYCLE:
        if sync_tick % d.CLOCK_RESET_PERIOD != 0:
            sync_tick = sync_tick // d.CLOCK_RESET_PERIOD * d.CLOCK_RESET_PERIOD
        for io_group in io_groups:
            rows.append(_other_row(SYNC, io_group, sync_tick, trigger_type=ord('S')))
    packets = np.array(rows, dtype=packets_dtype) if rows else np.zeros(0, dtype=packets_dtype)
    return packets, _empty_assn(len(rows))


def build_timestamp_trigger_packets(event_start_times, i_mod=-1):
    """``fee.export_timestamp_trigger_to_hdf5`` up to the file write (fee.py:426-497): a timestamp packet [s] and a trigger
    packet [ticks] on the trigger PACMAN for every event start time [us]."""
    d, units = consts.detector, consts.units
    rows = []
    io_group = get_trig_io()
    for evt_time in np.asarray(event_start_times):
        t_trig = int(np.floor(evt_time / d.CLOCK_CYCLE)) % d.CLOCK_RESET_PERIOD
        rows.append(_other_row(TIMESTAMP, io_group, evt_time * units.mus / units.s))
        rows.append(_other_row(TRIGGER, io_group, t_trig, trigger_type=2))
    packets = np.array(rows, dtype=packets_dtype) if rows else np.zeros(0, dtype=packets_dtype)
    return packets, _empty_assn(len(rows))


def write_hdf5(filename, packets, assn):
    """Append ``packets`` / ``mc_packets_assn`` to an HDF5 file laid out like larpix-control's (``_header`` version 2.4,
    resizable datasets) plus the ``configs`` attributes fee.export_to_hdf5 writes (fee.py:350-354).  Needs h5py."""
    import h5py
    d = consts.detector
    with h5py.File(filename, 'a') as f:
        if '_header' not in f:
            h = f.create_group('_header')
            h.attrs['version'] = '2.4'
            h.attrs['created'] = 0.0
            h.attrs['modified'] = 0.0
        for name, arr in (('packets', packets), ('mc_packets_assn', assn)):
            if not len(arr):
                continue
            if name not in f:
                f.create_dataset(name, data=arr, maxshape=(None,))
            else:
                f[name].resize((f[name].shape[0] + arr.shape[0]), axis=0)
                f[name][-arr.shape[0]:] = arr
        cfg = f.require_group('configs')
        cfg.attrs['vdrift'] = d.V_DRIFT
        cfg.attrs['long_diff'] = d.LONG_DIFF
        cfg.attrs['tran_diff'] = d.TRAN_DIFF
        cfg.attrs['lifetime'] = d.ELECTRON_LIFETIME
        cfg.attrs['drift_length'] = d.DRIFT_LENGTH
