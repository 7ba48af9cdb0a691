"""Light waveforms per (optical channel, tick) -- mirrors larndsim/light_sim.py: get_nticks (:24-41), sum_light_signals
(:58-129), calc_scintillation_effect (:148-184) and calc_light_detector_response (:303-337).  The stages that draw random
numbers (calc_stat_fluctuations, detector noise) and the trigger / digitisation stages are out of scope."""
import ctypes as C

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

