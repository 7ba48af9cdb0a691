"""Photon sum per (optical channel, tick) -- mirrors larndsim/light_sim.py sum_light_signals (:58-129) and
get_nticks (:24-41).  The rest of the light chain (scintillation, SiPM response, triggers) is out of scope."""
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
