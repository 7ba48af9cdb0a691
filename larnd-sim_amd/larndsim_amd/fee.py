"""Front-end electronics: self-trigger scan and digitisation -- mirrors larndsim/fee.py get_adc_values
(:517-655) and digitize (:499-515).  Packet export (fee.py:30-497) is out of scope."""
import ctypes as C

import numpy as np

from . import consts, lib
from ._kernel import kernel


def digitize(integral_list, gain=None):
    """``fee.digitize(integral_list, gain=GAIN*mV/e)``: ADC counts (f64), half-even rounding, clipped to 0..255."""
    q = np.ascontiguousarray(integral_list, dtype=np.float64)
    out = np.empty_like(q)
    g = None if gain is None else np.ascontiguousarray(np.broadcast_to(gain, q.shape), dtype=np.float64)
    lib.check(lib.load().ldsim_digitize(lib.context(), lib.ptr(q), C.c_int64(q.size), lib.ptr(g), lib.ptr(out)))
    return out


def load_pixel_table(filename):
    """(keys i32, values f64, default) of a pixel thresholds / gains file -- what the reference's ``CudaDict.load``
    reads (util/cuda_dict.py:82-88): an .npz with ``keys`` (pixel ids), ``values`` and ``default``."""
    with np.load(filename) as data:
        keys = np.ascontiguousarray(data["keys"], dtype=np.int32).ravel()
        values = np.ascontiguousarray(data["values"], dtype=np.float64).ravel()
        default = float(np.asarray(data["default"]).ravel()[0])
    if keys.shape != values.shape:
        raise ValueError(f"{filename}: keys and values differ in length")
    if np.unique(keys).size != keys.size:
        raise ValueError(f"{filename}: keys must be unique")
    return keys, values, default


@kernel
def get_adc_values(pixels_signals, pixels_signals_tracks, time_ticks, adc_list, adc_ticks_list, time_padding,
                   rng_states, current_fractions, pixel_thresholds):
    """``get_adc_values[bpg, tpb](...)`` with the reference's argument order.  With non-zero noise charges
    (RESET_NOISE_CHARGE, UNCORRELATED_NOISE_CHARGE, DISCRIMINATOR_NOISE) the normals come from the state table of
    ``rng.create_xoroshiro128p_states`` (``rng_states``: its handle, or None to use the table as it stands): state ``ip``
    for pixel ``ip``, advanced in place like the reference's ``rng_states[ip]`` (fee.py:557,583-584,616-617,621,649)."""
    ps = np.ascontiguousarray(pixels_signals, dtype=np.float64)
    U, NT = ps.shape
    pts = None if pixels_signals_tracks is None else np.ascontiguousarray(pixels_signals_tracks, dtype=np.float64)
    M = pts.shape[2] if pts is not None else (current_fractions.shape[2] if current_fractions is not None else 0)
    tt = np.ascontiguousarray(time_ticks, dtype=np.float64)
    thr = np.ascontiguousarray(pixel_thresholds, dtype=np.float64)
    adc = np.zeros(adc_list.shape); tk = np.zeros(adc_ticks_list.shape)
    fr = np.zeros(current_fractions.shape) if current_fractions is not None else None
    lib.check(lib.load().ldsim_get_adc_values(lib.context(), lib.ptr(ps), lib.ptr(pts), C.c_int64(U), C.c_int32(NT),
                                              C.c_int32(M), lib.ptr(tt), C.c_int32(len(tt)),
                                              C.c_double(float(time_padding)), lib.ptr(thr), lib.ptr(adc), lib.ptr(tk),
                                              lib.ptr(fr)))
    adc_list[:] = adc
    adc_ticks_list[:] = tk
    if fr is not None:
        current_fractions[:] = fr
