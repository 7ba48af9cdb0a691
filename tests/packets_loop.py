"""
Checker for ``larndsim_amd.packets.build_packets``: the hit loop of ``fee.export_to_hdf5`` (larndsim/fee.py:143-282)
restated hit by hit, with the association rows of :284-344.  Test infrastructure only -- the package never imports it; the
goldens (tests/golden/packets_*.npz, produced by the reference's own exporter) pin this loop, and the array form the driver
uses is held to it on random inputs (rollovers, disabled channels, light triggers, module selection).
"""
import numpy as np

from larndsim_amd import consts
from larndsim_amd.packets import (DATA, SYNC, TIMESTAMP, TRIGGER, _digitize0, _other_row, assn_dtype, packets_dtype)


def _rotate_tile(pixel_id, tile_id):
    d = consts.detector
    axes = d.TILE_ORIENTATIONS[tile_id]
    x_axis, y_axis = axes[2], axes[1]
    px = d.N_PIXELS_PER_TILE[0] - pixel_id[0] - 1 if x_axis < 0 else pixel_id[0]
    py = d.N_PIXELS_PER_TILE[1] - pixel_id[1] - 1 if y_axis < 0 else pixel_id[1]
    return px, py


def _data_parity(chip_id, channel_id, timestamp, first_packet, dataword):
    """Packet_v2 odd parity: bit 63 makes the number of set bits of the 64-bit word odd (type 0, flags 0)."""
    word = (0 & 0x3) | ((int(chip_id) & 0xFF) << 2) | ((int(channel_id) & 0x3F) << 10) | ((int(timestamp) & 0x7FFFFFFF) << 16) \
        | ((int(first_packet) & 1) << 47) | ((int(dataword) & 0xFF) << 48)
    return 1 - (bin(word).count("1") % 2)


def build_packets_loop(event_id_list, adc_list, adc_ticks_list, unique_pix, current_fractions, track_ids, traj_ids,
                       event_start_times, light_trigger_times=None, light_trigger_event_id=None, light_trigger_modules=None,
                       bad_channels=None, i_mod=-1):
    """``fee.export_to_hdf5`` up to the file write, hit by hit like the reference's own loop: returns (packets,
    mc_packets_assn) structured arrays.  Kept as the statement of the logic that the goldens pin and as the checker of
    ``build_packets`` (the array form the driver calls; this one costs ~27 us per hit).

    Same arguments as the reference: ``event_id_list`` [U][A] event of every ADC slot, ``adc_list`` [U][A] digitised ADC,
    ``adc_ticks_list`` [U][A], ``unique_pix`` [U], ``current_fractions`` [U][A][M], ``track_ids`` / ``traj_ids`` [U][M]
    (segment ids / trajectory ids per slot), ``event_start_times`` per unique event [us]; ``bad_channels``: dict
    chip_key -> channels (the parsed YAML) or None."""
    d, light, sim, units = consts.detector, consts.light, consts.sim, consts.units
    if not d.PIXEL_CONNECTION_DICT:
        raise RuntimeError("the pixel layout (chip / channel map) is not loaded: packets need consts.load_properties(...) on "
                           "the detector / pixel-layout YAML files or a snapshot that carries the readout map")
    io_groups = np.unique(np.array(list(d.MODULE_TO_IO_GROUPS.values())))
    io_groups = io_groups if i_mod < 0 else io_groups[(i_mod - 1) * 2: i_mod * 2]
    M_pix = sim.MAX_TRACKS_PER_PIXEL
    rows = []          # packet tuples
    mc_evt, mc_trk, mc_trj, mc_frac = [], [], [], []
    last_event = -1
    event_id_list = np.asarray(event_id_list)
    unique_events, unique_events_inv = np.unique(event_id_list[..., 0], return_inverse=True)
    event_start_times = np.asarray(event_start_times)
    event_start_time_list = (event_start_times[unique_events_inv] / d.CLOCK_CYCLE).astype(int)
    light_trigger_times = np.empty((0,)) if light_trigger_times is None else np.asarray(light_trigger_times)
    light_trigger_event_id = np.empty((0,), dtype=int) if light_trigger_event_id is None else np.asarray(light_trigger_event_id)
    light_trigger_modules = np.empty((0,)) if light_trigger_modules is None else np.asarray(light_trigger_modules)
    ped = _digitize0()
    n_trk, n_trj, n_frac = track_ids.shape[1], traj_ids.shape[1], current_fractions.shape[2]
    CRP = d.CLOCK_RESET_PERIOD

    def meta(evt, trk, trj, frac):
        mc_evt.append(evt); mc_trk.append(trk); mc_trj.append(trj); mc_frac.append(frac)

    def other(ptype, io_group, timestamp, trigger_type=0):
        rows.append((io_group, 0, 0, ptype, 0, 0, 1, 0, int(timestamp), 0, trigger_type, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0))

    last_time_tick = -1
    has_hits = np.flatnonzero(np.asarray(adc_list)[:, 0] > ped) if len(adc_list) else []
    for itick in has_hits:
        adcs = adc_list[itick]
        ts = adc_ticks_list[itick]
        pixel_id = int(unique_pix[itick])
        nx, ny = d.N_PIXELS
        pix_x, pix_y, plane_id = pixel_id % nx, (pixel_id // nx) % ny, pixel_id // (nx * ny)
        module_id = plane_id // 2 + 1
        if module_id not in d.MODULE_TO_IO_GROUPS:
            continue
        tile_x = int(pix_x // d.N_PIXELS_PER_TILE[0])
        tile_y = int(pix_y // d.N_PIXELS_PER_TILE[1])
        anode_id = 0 if plane_id % 2 == 0 else 1
        tile_id = d.TILE_MAP[anode_id][tile_x][tile_y]
        for iadc, adc in enumerate(adcs):
            t = ts[iadc]
            if not adc > ped:
                break
            while True:
                event = event_id_list[itick, iadc]
                event_t0 = event_start_time_list[itick]
                time_tick = int(np.floor(t / d.CLOCK_CYCLE + event_t0))
                if event_t0 > CRP - 1 or time_tick > CRP - 1:       # rollover at PPS / the 31-bit clock limit
                    event_start_time_list[itick:] -= CRP
                else:
                    break
            event_t0 = event_t0 % CRP
            time_tick = time_tick % CRP
            if light.LIGHT_TRIG_MODE != 1:
                if event != last_event:                              # new event: timestamp + sync per io_group, light triggers
                    for io_group in io_groups:
                        other(TIMESTAMP, io_group, event_start_times[unique_events_inv[itick]] * units.mus / units.s)
                        meta([-1], [-1] * n_trk, [-1] * n_trj, [0] * n_frac)
                        other(SYNC, io_group, time_tick, trigger_type=ord('S'))
                        meta([-1], [-1] * n_trk, [-1] * n_trj, [0] * n_frac)
                    trig_mask = light_trigger_event_id == event
                    if trig_mask.any():
                        for t_trig, module_trig in zip(light_trigger_times[trig_mask], light_trigger_modules[trig_mask]):
                            t_trig = int(np.floor(t_trig / d.CLOCK_CYCLE + event_t0)) % CRP
                            if light.LIGHT_TRIG_MODE == 0:
                                for io_group in d.MODULE_TO_IO_GROUPS[int(module_trig)]:
                                    other(TRIGGER, io_group, t_trig, trigger_type=2)
                                    meta([-1], [-1] * n_trk, [-1] * n_trj, [0] * n_frac)
                    last_event = event
            try:
                chip, channel = d.PIXEL_CONNECTION_DICT[_rotate_tile((pix_x % d.N_PIXELS_PER_TILE[0],
                                                                      pix_y % d.N_PIXELS_PER_TILE[1]), tile_id)]
            except KeyError:
                continue
            try:
                io_group_io_channel = d.TILE_CHIP_TO_IO[tile_id][chip]
            except KeyError:
                continue
            io_group, io_channel = io_group_io_channel // 1000, io_group_io_channel % 1000
            io_group = d.MODULE_TO_IO_GROUPS[module_id][io_group - 1]
            if bad_channels:
                chip_key = "%i-%i-%i" % (io_group, io_channel, chip)
                if chip_key in bad_channels and channel in bad_channels[chip_key]:
                    continue
            if not time_tick == last_time_tick:
                # one timestamp packet per group of packets with the same timestamp (fee.py:267-277)
                last_time_tick = time_tick
                other(TIMESTAMP, io_group, np.floor(event_start_time_list[0] * d.CLOCK_CYCLE * units.mus / units.s))
                meta([-1], [-1] * M_pix, [-1] * M_pix, [0] * M_pix)
            dataword = int(adc)
            rows.append((io_group, io_channel, chip, DATA, 0, _data_parity(chip, channel, time_tick, 1, dataword), 1, channel,
                         time_tick, dataword, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, time_tick))
            meta([event], track_ids[itick], traj_ids[itick], current_fractions[itick][iadc])
    packets = np.array(rows, dtype=packets_dtype) if rows else np.zeros(0, dtype=packets_dtype)
    if not rows:
        return packets, np.zeros(0, dtype=assn_dtype())

    # ---- mc_packets_assn (fee.py:284-344) -----------------------------------------------------------------------------------
    n_keep = sim.ASSOCIATION_COUNT_TO_STORE
    ds = np.empty(len(rows), dtype=assn_dtype())
    packets_frac = np.array(mc_frac)
    packets_mc_trk = np.array(mc_trk)
    packets_mc_trj = np.array(mc_trj)
    packets_mc_evt = np.array(mc_evt)
    # (the reference: np.flip(np.argsort(...)); its order of EQUAL fractions -- exact zeros of real track slots beside the unused
    # slots' zeros -- is left to numpy's build (its SIMD sort kernels are not stable): unpinned.  kind="stable" fixes it to descending
    # slot order, the rule of larndsim_amd.packets and csrc/packets.hip; rows without equal real fractions are unaffected)
    frac_order = np.flip(np.argsort(packets_frac, axis=1, kind="stable"), axis=1)
    ass_segment_ids = np.take_along_axis(packets_mc_trk, frac_order, axis=1)
    ass_trajectory_ids = np.take_along_axis(packets_mc_trj, frac_order, axis=1)
    ass_fractions = np.take_along_axis(packets_frac, frac_order, axis=1)

    def keep(a, fill):
        if a.shape[1] >= n_keep:
            return a[:, :n_keep]
        return np.pad(a, pad_width=((0, 0), (0, n_keep - a.shape[1])), mode='constant', constant_values=fill)

    ds['segment_ids'] = keep(ass_segment_ids, -1)
    ds['fraction'] = keep(ass_fractions, 0.)
    ass_track_ids = np.full(ass_trajectory_ids.shape, fill_value=-1, dtype=np.int32)
    ass_fractions_track = np.full(ass_fractions.shape, fill_value=0., dtype=np.float32)
    for pidx, tids in enumerate(ass_trajectory_ids):
        mask = tids > -1
        if not mask.any():
            continue
        for tidx, unique_tid in enumerate(np.unique(tids[mask])):
            ass_track_ids[pidx][tidx] = unique_tid
            ass_fractions_track[pidx][tidx] = np.sum(ass_fractions[pidx][mask][tids[mask] == unique_tid])
    ds['file_traj_ids'] = keep(ass_track_ids, -1)
    ds['fraction_traj'] = keep(ass_fractions_track, 0.)
    ds['event_ids'] = packets_mc_evt
    return packets, ds


