"""Light waveforms per (optical channel, tick) -- mirrors larndsim/light_sim.py: get_nticks (:24-41), sum_light_signals
(:58-129), calc_scintillation_effect (:148-184), calc_stat_fluctuations (:186-238), calc_light_detector_response (:303-337),
get_triggers (:339-443), gen_light_detector_noise (:445-478), sim_triggers incl. digitize_signal (:480-619) and the
light_trig / light_wvfm / light_wvfm_mc_assn writers (:621-757).  Random numbers: the numba-style state table for the Poisson
stage, a counter hash for the noise phases (the reference uses cupy's global generator there); both unpinned."""
import ctypes as C
from math import ceil

import numpy as np

from . import consts, lib
from ._kernel import kernel
from .layout import make_layout
from .lightLUT import _ensure_lut


def get_nticks(light_incidence):
    light = consts.light
    mask = light_incidence['n_photons_det'] > 0
    if np.any(mask) and light.LIGHT_TRIG_MODE == 0:
        start_time = np.min(light_incidence['t0_det'][mask]) - light.LIGHT_WINDOW[0]
        end_time = np.max(light_incidence['t0_det'][mask]) + light.LIGHT_WINDOW[1]
        return int(np.ceil((end_time - start_time) / light.LIGHT_TICK_SIZE)), start_time
    return int((light.LIGHT_WINDOW[1] + light.LIGHT_WINDOW[0]) / light.LIGHT_TICK_SIZE), 0


@kernel
def sum_light_signals(segments, segment_voxel, segment_track_id, light_inc, op_channel, lut, start_time,
                      light_sample_inc, light_sample_inc_true_track_id, light_sample_inc_true_photons,
                      sorted_indices, t0_profile_length):
    """``sum_light_signals[bpg, tpb](...)`` with the reference's argument order."""
    lay = make_layout(segments.dtype)
    n = segments.shape[0]
    lib.context()
    _ensure_lut(lut)
    nph = np.ascontiguousarray(light_inc['n_photons_det'], dtype=np.float32)
    vox = np.ascontiguousarray(segment_voxel, dtype=np.int32)
    tid = np.ascontiguousarray(segment_track_id, dtype=np.int64)
    opc = np.ascontiguousarray(op_channel, dtype=np.int32)
    srt = np.ascontiguousarray(sorted_indices, dtype=np.int32)
    n_det, n_ticks = light_sample_inc.shape
    out = np.ascontiguousarray(light_sample_inc, dtype=np.float32)
    mt = light_sample_inc_true_track_id.shape[-1] if light_sample_inc_true_track_id is not None else 0
    tids = np.ascontiguousarray(light_sample_inc_true_track_id, dtype=np.int64) if mt else None
    tph = np.ascontiguousarray(light_sample_inc_true_photons, dtype=np.float64) if mt else None
    lib.check(lib.load().ldsim_sum_light_signals(
        lib.context(refresh_consts=False), lib.ptr(segments), C.c_int64(n), C.byref(lay), lib.ptr(vox), lib.ptr(tid),
        lib.ptr(nph), C.c_int32(nph.shape[1]), lib.ptr(opc), C.c_int32(n_det), lib.ptr(srt),
        C.c_double(float(start_time)), C.c_int32(n_ticks), lib.ptr(out), lib.ptr(tids), lib.ptr(tph), C.c_int32(mt)))
    light_sample_inc[:] = out
    if mt:
        light_sample_inc_true_track_id[:] = tids
        light_sample_inc_true_photons[:] = tph


def _truth(ids, photons):
    """(max_truth, i8 ids, f8 photons) of a pair of truth arrays; (0, None, None) when there are no slots."""
    if ids is None or ids.shape[-1] == 0:
        return 0, None, None
    return ids.shape[-1], np.ascontiguousarray(ids, dtype=np.int64), np.ascontiguousarray(photons, dtype=np.float64)


@kernel
def calc_scintillation_effect(light_sample_inc, light_sample_inc_true_track_id, light_sample_inc_true_photons,
                              light_sample_inc_scint, light_sample_inc_scint_true_track_id,
                              light_sample_inc_scint_true_photons):
    """``calc_scintillation_effect[bpg, tpb](...)`` with the reference's argument order: the two-component scintillation
    time profile convolved into ``light_sample_inc_scint`` (added to what the caller put there, like the reference)."""
    inc = np.ascontiguousarray(light_sample_inc, dtype=np.float32)
    n_det, n_ticks = inc.shape
    mt, tid, tph = _truth(light_sample_inc_true_track_id, light_sample_inc_true_photons)
    out = np.ascontiguousarray(light_sample_inc_scint, dtype=np.float32)
    otid = np.ascontiguousarray(light_sample_inc_scint_true_track_id, dtype=np.int64) if mt else None
    otph = np.ascontiguousarray(light_sample_inc_scint_true_photons, dtype=np.float64) if mt else None
    lib.check(lib.load().ldsim_scintillation_effect(
        lib.context(), lib.ptr(inc), lib.ptr(tid), lib.ptr(tph), C.c_int32(n_det), C.c_int32(n_ticks), C.c_int32(mt),
        lib.ptr(out), lib.ptr(otid), lib.ptr(otph)))
    light_sample_inc_scint[:] = out
    if mt:
        light_sample_inc_scint_true_track_id[:] = otid
        light_sample_inc_scint_true_photons[:] = otph


@kernel
def calc_light_detector_response(light_sample_inc, light_sample_inc_true_track_id, light_sample_inc_true_photons,
                                 light_response, light_response_true_track_id, light_response_true_photons):
    """``calc_light_detector_response[bpg, tpb](...)`` with the reference's argument order: SiPM response model
    (``consts.light.SIPM_RESPONSE_MODEL``: 0 RLC, 1 ``IMPULSE_MODEL``) times ``LIGHT_GAIN[row]`` convolved into
    ``light_response``."""
    light = consts.light
    inc = np.ascontiguousarray(light_sample_inc, dtype=np.float32)
    n_det, n_ticks = inc.shape
    gain = np.ascontiguousarray(light.LIGHT_GAIN, dtype=np.float64)
    if gain.shape[0] < n_det:
        raise IndexError(f"LIGHT_GAIN has {gain.shape[0]} entries, the arrays have {n_det} rows")
    imp = np.ascontiguousarray(light.IMPULSE_MODEL, dtype=np.float64)
    mt, tid, tph = _truth(light_sample_inc_true_track_id, light_sample_inc_true_photons)
    out = np.ascontiguousarray(light_response, dtype=np.float32)
    otid = np.ascontiguousarray(light_response_true_track_id, dtype=np.int64) if mt else None
    otph = np.ascontiguousarray(light_response_true_photons, dtype=np.float64) if mt else None
    lib.check(lib.load().ldsim_light_detector_response(
        lib.context(), lib.ptr(inc), lib.ptr(tid), lib.ptr(tph), C.c_int32(n_det), C.c_int32(n_ticks), C.c_int32(mt),
        lib.ptr(gain), lib.ptr(imp), C.c_int32(imp.shape[0]), lib.ptr(out), lib.ptr(otid), lib.ptr(otph)))
    light_response[:] = out
    if mt:
        light_response_true_track_id[:] = otid
        light_response_true_photons[:] = otph



@kernel
def calc_stat_fluctuations(light_sample_inc, light_sample_inc_disc, rng_states):
    """``calc_stat_fluctuations[bpg, tpb](light_sample_inc, light_sample_inc_disc, rng_states)``: Poisson fluctuations of the
    PE count per tick.  ``rng_states`` is the handle of ``rng.create_xoroshiro128p_states`` (the table lives on the device);
    element ``(idet, itick)`` uses state ``idet*ntick + itick``."""
    inc = np.ascontiguousarray(light_sample_inc, dtype=np.float32)
    n_det, n_ticks = inc.shape
    if rng_states is None or len(rng_states) < inc.size:
        raise IndexError(f"rng_states holds {0 if rng_states is None else len(rng_states)} states, {inc.size} are indexed")
    out = np.zeros(inc.shape, dtype=np.float32)
    lib.check(lib.load().ldsim_stat_fluctuations(lib.context(), lib.ptr(inc), C.c_int32(n_det), C.c_int32(n_ticks),
                                                 lib.ptr(out)))
    light_sample_inc_disc[:] = out


def _module_rows(op_channel_idx):
    """(sorted module ids, row -> module index, module -> its optical channels) like light_sim.py:374-384."""
    light, detector = consts.light, consts.detector
    op_channel_idx = np.asarray(op_channel_idx)
    tpc_ids = np.unique(light.OP_CHANNEL_TO_TPC[op_channel_idx])
    mod_ids = np.unique([detector.TPC_TO_MODULE[int(t)] for t in tpc_ids])
    row_module = np.full(op_channel_idx.shape[0], -1, dtype=np.int32)
    channels = []
    for im, mod in enumerate(mod_ids):
        ch = light.TPC_TO_OP_CHANNEL[detector.MODULE_TO_TPCS[mod]].ravel()
        channels.append(ch)
        row_module[np.isin(op_channel_idx, ch)] = im
    return mod_ids, row_module, channels


def get_triggers(signal, group_threshold, op_channel_idx, i_subbatch):
    """Ticks that start a trigger: ``(trigger_idx, op_channel_idx per trigger, trigger_type)``.  ``signal`` is the detector
    response ``[ndet, nticks]``, or ``None`` for the device-resident one of ``chain.light_response``."""
    light = consts.light
    op_channel_idx = np.asarray(op_channel_idx)
    trig, chans, types = [], [], []
    if light.LIGHT_TRIG_MODE == 0:
        _, row_module, channels = _module_rows(op_channel_idx)
        thr = np.ascontiguousarray(group_threshold, dtype=np.float64)
        n_det = op_channel_idx.shape[0]
        if signal is not None:
            signal = np.ascontiguousarray(signal, dtype=np.float32)
            if signal.shape[0] != n_det:
                raise ValueError("signal rows and op_channel_idx differ")
            n_ticks = signal.shape[1]
        else:
            n_ticks = lib.context_light_shape()[1]
        cap = max(16, n_ticks // max(1, _digit_ticks()) * len(channels) + len(channels))
        while True:
            idx = np.zeros(cap, dtype=np.int64); mod = np.zeros(cap, dtype=np.int32); n = C.c_int64(0)
            rc = lib.load().ldsim_light_triggers(
                lib.context(), lib.ptr(signal), C.c_int32(n_det), C.c_int32(n_ticks), lib.ptr(thr), C.c_int32(thr.shape[0]),
                lib.ptr(row_module), C.c_int32(len(channels)), lib.ptr(idx), lib.ptr(mod), C.c_int64(cap), C.byref(n))
            if rc == lib.LDSIM_ENOSPC:
                cap = int(n.value)
                continue
            lib.check(rc)
            break
        for i in range(int(n.value)):
            trig.append(int(idx[i])); types.append(0); chans.append(channels[int(mod[i])])
    elif light.LIGHT_TRIG_MODE == 1 and i_subbatch == 0:
        trig.append(0); chans.append(op_channel_idx); types.append(1)
    if trig:
        return np.array(trig), np.array(chans), np.array(types)
    return np.empty((0,), dtype=int), np.empty((0, len(op_channel_idx)), dtype=int), np.empty((0,), dtype=int)


def _digit_ticks():
    light = consts.light
    return ceil((light.LIGHT_TRIG_WINDOW[1] + light.LIGHT_TRIG_WINDOW[0]) / light.LIGHT_TICK_SIZE)


def gen_light_detector_noise(shape, light_det_noise, phases=None):
    """Uncorrelated noise with the frequency spectrum ``light_det_noise`` (one row per output row): ``[shape[0], shape[1]]``
    f8 in digitiser LSBs.  ``phases`` (``[shape[0], shape[1]//2 + 1]`` uniform numbers) replaces the internal draw."""
    shape = (int(shape[0]), int(shape[1]))
    if not shape[0]:
        return np.empty(shape)
    spec = np.ascontiguousarray(light_det_noise, dtype=np.float64)
    if spec.ndim != 2 or spec.shape[0] != shape[0]:
        raise ValueError("shape[0] must equal light_det_noise.shape[0]")
    ph = None if phases is None else np.ascontiguousarray(phases, dtype=np.float64)
    if ph is not None and ph.shape != (shape[0], shape[1] // 2 + 1):
        raise ValueError("phases must be [shape[0], shape[1]//2 + 1]")
    out = np.zeros(shape)
    lib.check(lib.load().ldsim_light_detector_noise(lib.context(), C.c_int32(shape[0]), C.c_int32(shape[1]), lib.ptr(spec),
                                                    C.c_int32(spec.shape[1]), lib.ptr(ph), lib.ptr(out)))
    return out


def sim_triggers(bpg, tpb, signal, signal_op_channel_idx, signal_true_track_id, signal_true_photons, trigger_idx,
                 op_channel_idx, digit_samples, light_det_noise, phases_signal=None, phases_missing=None):
    """Digitised waveforms at the trigger ticks, reference argument order (``bpg``/``tpb`` are accepted and unused):
    ``(digit_signal [ntrigs, ndet_module, digit_samples] f8, true track ids, true photons)``.  ``signal=None`` digitises the
    device-resident response of ``chain.light_response`` (its truth arrays stay on the device too)."""
    trigger_idx = np.ascontiguousarray(trigger_idx, dtype=np.int64)
    top = np.ascontiguousarray(op_channel_idx, dtype=np.int32)
    ntrig = trigger_idx.shape[0]
    ndm = top.shape[-1] if top.ndim else 0
    if top.ndim == 1:
        top = np.ascontiguousarray(np.broadcast_to(top, (ntrig, ndm)))
    sop = np.ascontiguousarray(signal_op_channel_idx, dtype=np.int32)
    if signal is not None:
        sig = np.ascontiguousarray(signal, dtype=np.float32)
        n_det, n_ticks = sig.shape
        mt, tid, tph = _truth(signal_true_track_id, signal_true_photons)
        id_dtype = signal_true_track_id.dtype if mt else np.int64
        ph_dtype = signal_true_photons.dtype if mt else np.float64
    else:
        sig, tid, tph = None, None, None
        n_det, n_ticks, mt = lib.context_light_shape()
        id_dtype, ph_dtype = np.int64, np.float64
    digit = np.zeros((ntrig, ndm, int(digit_samples)), dtype='f8')
    dtid = np.full((ntrig, ndm, int(digit_samples), mt), -1, dtype=np.int64)
    dtph = np.zeros((ntrig, ndm, int(digit_samples), mt), dtype=np.float64)
    if ntrig == 0:
        return digit, dtid.astype(id_dtype), dtph.astype(ph_dtype)
    noise = None if light_det_noise is None else np.ascontiguousarray(light_det_noise, dtype=np.float64)
    ps = None if phases_signal is None else np.ascontiguousarray(phases_signal, dtype=np.float64)
    pm = None if phases_missing is None else np.ascontiguousarray(phases_missing, dtype=np.float64)
    lib.check(lib.load().ldsim_sim_triggers(
        lib.context(), lib.ptr(sig), lib.ptr(sop), C.c_int32(n_det), C.c_int32(n_ticks), lib.ptr(tid), lib.ptr(tph),
        C.c_int32(mt), lib.ptr(trigger_idx), C.c_int32(ntrig), lib.ptr(top), C.c_int32(ndm), C.c_int32(int(digit_samples)),
        lib.ptr(noise), C.c_int32(noise.shape[0] if noise is not None else 0),
        C.c_int32(noise.shape[1] if noise is not None else 0), lib.ptr(ps), lib.ptr(pm), lib.ptr(digit),
        lib.ptr(dtid) if mt else None, lib.ptr(dtph) if mt else None))
    return digit, dtid.astype(id_dtype, copy=False), dtph.astype(ph_dtype, copy=False)


# ---- output datasets (light_sim.py:621-757) -------------------------------------------------------------------------------
light_wvfm_truth_dtype = np.dtype([('trigger_id', 'i4'), ('op_channel_id', 'i4'), ('tick', 'i4'), ('event_id', 'i4'),
                                   ('segment_id', 'i8'), ('pe_current', 'f8')])


def zero_suppress_waveform_truth(waveforms_true_track_id, waveforms_true_photons, i_evt, i_trig, i_mod=-1):
    """Rows of ``light_wvfm_mc_assn``: one per truth slot that is not -1, in array order.  ``trigger_id`` follows the
    reference (:634-645): the running ``i_trig`` is advanced by the row's own trigger index for every row, so with several
    triggers in one call the ids drift exactly as they do there."""
    light = consts.light
    op_channel = (light.TPC_TO_OP_CHANNEL[(i_mod - 1) * 2:i_mod * 2].ravel() if i_mod > 0
                  else light.TPC_TO_OP_CHANNEL[:].ravel())
    ids = np.asarray(waveforms_true_track_id)
    t, c, s, k = np.nonzero(ids != -1)
    out = np.empty(t.shape[0], dtype=light_wvfm_truth_dtype)
    out['trigger_id'] = i_trig + np.cumsum(t)
    out['op_channel_id'] = op_channel[c]
    out['tick'] = s
    out['event_id'] = i_evt
    out['segment_id'] = ids[t, c, s, k]
    out['pe_current'] = np.asarray(waveforms_true_photons)[t, c, s, k]
    return out


def build_light_trig(event_id, start_times, trigger_idx, op_channel_idx, event_times):
    """Rows of the ``light_trig`` dataset (:695-723): ``op_channel`` i4[ndet_module], ``ts_s`` f8 [s], ``ts_sync`` u8 [ticks]."""
    light, detector = consts.light, consts.detector
    event_id = np.asarray(event_id)
    trigger_idx = np.asarray(trigger_idx)
    op_channel_idx = np.asarray(op_channel_idx)
    _, inv = np.unique(event_id, return_inverse=True)
    event_times = np.asarray(event_times)
    event_start_times = event_times[inv]
    event_sync_times = (event_times[inv] / detector.CLOCK_CYCLE).astype(int) % detector.CLOCK_RESET_PERIOD
    trig = np.empty(trigger_idx.shape[0], dtype=np.dtype([('op_channel', 'i4', (op_channel_idx.shape[-1])), ('ts_s', 'f8'),
                                                           ('ts_sync', 'u8')]))
    trig['op_channel'] = op_channel_idx
    trig['ts_s'] = (start_times + trigger_idx * light.LIGHT_TICK_SIZE + event_start_times) * consts.units.mus / consts.units.s
    trig['ts_sync'] = (((start_times + trigger_idx * light.LIGHT_TICK_SIZE) / detector.CLOCK_CYCLE
                        + event_sync_times).astype(int) % detector.CLOCK_RESET_PERIOD)
    return trig


def _append(f, name, data, maxshape):
    if name not in f:
        f.create_dataset(name, data=data, maxshape=maxshape)
    else:
        f[name].resize(f[name].shape[0] + data.shape[0], axis=0)
        f[name][-data.shape[0]:] = data


def export_light_trig_to_hdf5(event_id, start_times, trigger_idx, op_channel_idx, output_filename, event_times):
    """Append to ``light_trig`` (needs h5py; the CLI's .npz fallback uses ``build_light_trig`` directly)."""
    if np.asarray(event_id).shape[0] == 0:
        return
    import h5py
    trig = build_light_trig(event_id, start_times, trigger_idx, op_channel_idx, event_times)
    with h5py.File(output_filename, 'a') as f:
        _append(f, 'light_trig', trig, (None,))


def wvfm_dataset_name(i_mod=-1):
    """Where a call's waveforms go (light_sim.py:668-685): with module variation and the beam trigger every module fills its
    own ``light_wvfm/light_wvfm_mod<i>`` (merged at the end of the file), otherwise ``light_wvfm``."""
    if getattr(consts.sim, 'MOD2MOD_VARIATION', False) and consts.light.LIGHT_TRIG_MODE == 1:
        if not i_mod > 0:
            raise ValueError("Mod2mod variation is activated, but the module id is not provided correctly.")
        return f'light_wvfm/light_wvfm_mod{i_mod - 1}'
    return 'light_wvfm'


def export_light_wvfm_to_hdf5(event_id, waveforms, output_filename, waveforms_true_track_id, waveforms_true_photons, i_trig,
                              i_mod=-1):
    """Append to ``light_wvfm`` (per module: ``light_wvfm/light_wvfm_mod<i>``) and, with ``MAX_MC_TRUTH_IDS > 0``, to
    ``light_wvfm_mc_assn`` (:647-693).  Needs h5py."""
    if np.asarray(event_id).shape[0] == 0:
        return
    import h5py
    with h5py.File(output_filename, 'a') as f:
        _append(f, wvfm_dataset_name(i_mod), np.asarray(waveforms), (None, None, None))
        if consts.sim.MAX_MC_TRUTH_IDS > 0:
            truth = zero_suppress_waveform_truth(waveforms_true_track_id, waveforms_true_photons, event_id[0], i_trig, i_mod)
            if truth.shape[0] > 0:
                _append(f, 'light_wvfm_mc_assn', truth, (None,))


def merge_module_light_wvfm_same_trigger(output_filename, module_ids=None):
    """The per-module waveform datasets side by side along the channel axis as ``light_wvfm`` (:759-775); the group of
    per-module datasets is replaced by the merged dataset.  Needs h5py."""
    import h5py
    module_ids = consts.detector.MOD_IDS if module_ids is None else module_ids
    with h5py.File(output_filename, 'a') as f:
        parts = [np.array(f[f'light_wvfm/light_wvfm_mod{i_mod - 1}']) for i_mod in module_ids]
        if len({p.shape[0] for p in parts}) != 1:
            raise ValueError("The number of triggers should be the same in each module with light trigger mode 1 "
                             "(light waveform).")
        del f['light_wvfm']
        f.create_dataset('light_wvfm', data=np.concatenate(parts, axis=1), maxshape=(None, None, None))


def export_to_hdf5(event_id, start_times, trigger_idx, op_channel_idx, waveforms, output_filename, event_times,
                   waveforms_true_track_id, waveforms_true_photons, i_trig, i_mod=-1):
    export_light_trig_to_hdf5(event_id, start_times, trigger_idx, op_channel_idx, output_filename, event_times)
    export_light_wvfm_to_hdf5(event_id, waveforms, output_filename, waveforms_true_track_id, waveforms_true_photons, i_trig,
                              i_mod)
