"""Induced current on the pixels -- mirrors larndsim/detsim.py: time_intervals (:18-40), tracks_current
(:351-453), get_track_pixel_map2 (:564-607), sum_pixel_signals (:468-527)."""
import ctypes as C

import numpy as np

from . import consts, lib
from ._kernel import kernel
from .layout import make_layout


@kernel
def time_intervals(track_starts, time_max, tracks):
    """``time_intervals[bpg, tpb](track_starts, time_max, tracks)``."""
    lay = make_layout(tracks.dtype)
    n = tracks.shape[0]
    ts = track_starts if track_starts.dtype == np.float64 and track_starts.flags.c_contiguous else np.empty(n)
    v = C.c_int64(int(time_max[0]))
    lib.check(lib.load().ldsim_time_intervals(lib.context(), lib.ptr(tracks), C.c_int64(n), C.byref(lay), lib.ptr(ts),
                                              C.byref(v)))
    if ts is not track_starts:
        track_starts[:] = ts
    time_max[0] = v.value


def _ensure_response(response):
    lib.set_response(response)        # no-op when this table is already the resident one


@kernel
def tracks_current(signals, pixels, tracks, response):
    """``tracks_current[(S,P,T/64),(1,1,64)](signals, pixels, tracks, response)``: f32 signals[S,P,T]."""
    lay = make_layout(tracks.dtype)
    S, P, T = signals.shape
    if signals.dtype != np.float32 or not signals.flags.c_contiguous:
        raise TypeError("signals must be C-contiguous float32 [S, P, T]")
    pix = np.ascontiguousarray(pixels, dtype=np.int32)
    lib.context()
    _ensure_response(response)
    lib.check(lib.load().ldsim_tracks_current(lib.context(refresh_consts=False), lib.ptr(tracks), C.c_int64(S),
                                              C.byref(lay), lib.ptr(pix), C.c_int32(P), lib.ptr(signals),
                                              C.c_int32(T)))


def tracks_current_stats():
    """Counters of the last ``tracks_current`` call (``LdsimChainStats``): ``n_wbuf`` > 0 = the split path's weights stage
    ran, ``n_fallback`` = pairs the monolithic kernel recomputed."""
    st = lib.LdsimChainStats()
    lib.check(lib.load().ldsim_tracks_current_stats(lib.context(refresh_consts=False), C.byref(st)))
    return st


@kernel
def tracks_current_mc(signals, pixels, tracks, response, rng_states):
    """``tracks_current_mc[bpg, tpb](signals, pixels, tracks, response, rng_states)`` -- the reference driver's call site
    (cli/simulate_pixels.py:1016; detsim.py:258-348).  ``rng_states``: handle of ``rng.create_xoroshiro128p_states`` (or None
    to use the table as it stands).  The reference's tick threads race on one state per (segment, pixel); here every
    (segment, pixel, tick) has its own stream derived from that state, so runs are reproducible and statistically, not
    bitwise, comparable with the reference's."""
    S, P = pixels.shape
    T = signals.shape[2]
    lay = make_layout(tracks.dtype)
    lib.context()
    lib.set_response(response)
    pix = np.ascontiguousarray(pixels, dtype=np.int32)
    out = np.zeros((S, P, T), dtype=np.float32)
    lib.check(lib.load().ldsim_tracks_current_mc(lib.context(refresh_consts=False), lib.ptr(tracks), C.c_int64(S),
                                                 C.byref(lay), lib.ptr(pix), C.c_int32(P), lib.ptr(out), C.c_int32(T)))
    signals[:] = out


@kernel
def get_track_pixel_map2(track_pixel_map, unique_pix, pixels, distances, max_distance):
    """``get_track_pixel_map2[bpg, tpb](track_pixel_map, unique_pix, pixels, distances, max_distance)``."""
    tpm = track_pixel_map if track_pixel_map.dtype == np.int64 and track_pixel_map.flags.c_contiguous else \
        np.ascontiguousarray(track_pixel_map, dtype=np.int64)
    up = np.ascontiguousarray(unique_pix, dtype=np.int32)
    px = np.ascontiguousarray(pixels, dtype=np.int32)
    ds = np.ascontiguousarray(distances, dtype=np.int32)
    lib.check(lib.load().ldsim_track_pixel_map(lib.context(), lib.ptr(up), C.c_int64(len(up)), lib.ptr(px), lib.ptr(ds),
                                               C.c_int64(px.shape[0]), C.c_int32(px.shape[1]),
                                               C.c_int32(int(max_distance)), lib.ptr(tpm), C.c_int32(tpm.shape[1])))
    if tpm is not track_pixel_map:
        track_pixel_map[:] = tpm


@kernel
def sum_pixel_signals(pixels_signals, signals, track_starts, pixel_index_map, track_pixel_map, pixels_tracks_signals,
                      overflow_flag):
    """``sum_pixel_signals[...]``: accumulates into ``pixels_signals`` / ``pixels_tracks_signals`` (expected zero-filled)."""
    S, P, T = signals.shape
    U = pixels_signals.shape[0]
    sig = np.ascontiguousarray(signals, dtype=np.float32)
    ts = np.ascontiguousarray(track_starts, dtype=np.float64)
    pim = np.ascontiguousarray(pixel_index_map, dtype=np.int64)
    tpm = np.ascontiguousarray(track_pixel_map, dtype=np.int64)
    ps = np.zeros(pixels_signals.shape)
    pts = np.zeros(pixels_tracks_signals.shape) if pixels_tracks_signals is not None else None
    ov = np.zeros(U)
    lib.check(lib.load().ldsim_sum_pixel_signals(lib.context(), lib.ptr(sig), C.c_int64(S), C.c_int32(P), C.c_int32(T),
                                                 lib.ptr(ts), lib.ptr(pim), lib.ptr(tpm), C.c_int32(tpm.shape[1]),
                                                 C.c_int64(U), lib.ptr(ps), lib.ptr(pts), lib.ptr(ov)))
    pixels_signals += ps
    if pts is not None:
        pixels_tracks_signals += pts
    overflow_flag[ov != 0] = 1
