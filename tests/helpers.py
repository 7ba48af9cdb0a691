"""Shared test helpers (record builders, golden loading, tolerances)."""
import os

import numpy as np

from larndsim_amd import consts, synth
from larndsim_amd.layout import segments_dtype

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

F4_FIELDS = ["x_start", "y_start", "z_start", "x_end", "y_end", "z_end", "x", "y", "z", "dx", "dEdx", "dE",
             "t", "t_start", "t_end", "n_photons", "long_diff", "tran_diff"]
# the layout the golden generator ran the reference on: f8 floats holding f4-representable values,
# real integer dtypes (u4 n_electrons truncates on store)
REF_DTYPE = np.dtype([("event_id", "u4"), ("segment_id", "u4"), ("traj_id", "u4"), ("n_electrons", "u4"),
                      ("pixel_plane", "i4")] + [(f, "f8") for f in F4_FIELDS] +
                     [("t0", "f8"), ("t0_start", "f8"), ("t0_end", "f8")])

SNAP = {"module0": "module0", "2x2_no_modvar": "2x2_no_modvar", "ndlar": "ndlar"}


def load_cfg(cfg, noise_zero=True):
    consts.load_snapshot(SNAP[cfg])
    if noise_zero:
        consts.detector.RESET_NOISE_CHARGE = 0
        consts.detector.UNCORRELATED_NOISE_CHARGE = 0
        consts.detector.DISCRIMINATOR_NOISE = 0


def gold(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def to_ref(seg):
    r = np.zeros(seg.shape[0], dtype=REF_DTYPE)
    for n in REF_DTYPE.names:
        if n in seg.dtype.names:
            r[n] = seg[n]
    return r


def f4(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def round_f4(r, fields):
    for n in fields:
        r[n] = f4(r[n])


def quench_drift(mod, seg, mode=2):
    """quench + drift with the f4 round trip between stages the golden generator applied."""
    r = to_ref(seg)
    mod.quench(r, mode)
    round_f4(r, ["n_photons"])
    mod.drift(r)
    round_f4(r, ["long_diff", "tran_diff", "t", "t_start", "t_end"])
    return r


def response_for(kind):
    return synth.make_response(str(kind), response_sampling=consts.detector.RESPONSE_SAMPLING)


def assert_wave_close(got, ref, rtol=1e-5, atol_peak=1e-7, what=""):
    """|got-ref| <= rtol*|ref| + atol_peak*max|ref| (per waveform along the last axis)."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    peak = np.max(np.abs(ref), axis=-1, keepdims=True)
    err = np.abs(got - ref)
    tol = rtol * np.abs(ref) + atol_peak * peak
    bad = err > tol
    if bad.any():
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError(f"{what}: {bad.sum()} values out of tolerance; worst at {i}: got {got[i]!r} "
                             f"ref {ref[i]!r} peak {peak[i[:-1]]}")


def load_light_response_case(cfg):
    """Constants of a tests/golden/light_response_<cfg>.npz case (oracle/gen_golden.py gen_light_response): the
    configuration's own constants with the window / SiPM model / gains the golden run used."""
    load_cfg(cfg)
    g = gold(f"light_response_{cfg}.npz")
    l = consts.light
    l.LIGHT_WINDOW = tuple(float(x) for x in g["light_window"])
    l.SIPM_RESPONSE_MODEL = int(g["sipm_response_model"])
    l.IMPULSE_TICK_SIZE = float(g["impulse_tick_size"])
    l.IMPULSE_MODEL = np.asarray(g["impulse_model"], dtype=float)
    l.LIGHT_GAIN = np.asarray(g["light_gain"], dtype=float)
    consts.sim.MC_TRUTH_THRESHOLD = float(g["mc_truth_threshold"])
    return g



LIGHT_WVFM_CASES = ("light_wvfm_module0_0", "light_wvfm_2x2_no_modvar_1", "light_wvfm_2x2_no_modvar_2")


def load_light_wvfm_case(name):
    """Constants of a tests/golden/light_wvfm_*.npz case (oracle/gen_golden.py gen_light_wvfm)."""
    cfg = "module0" if "module0" in name else "2x2_no_modvar"
    load_cfg(cfg)
    g = gold(name + ".npz")
    consts.light.LIGHT_TRIG_MODE = int(g["light_trig_mode"])
    consts.light.LIGHT_TRIG_WINDOW = tuple(float(x) for x in g["light_trig_window"])
    consts.sim.MC_TRUTH_THRESHOLD = float(g["mc_truth_threshold"])
    return g


def det_phases(shape, seed):
    """The golden generator's stand-in for cp.random.uniform(size=shape) (oracle/gen_golden.py det_phases): a
    multiplicative hash of (row, column, seed) in [0, 1), so the fixtures need not store the phases."""
    shape = tuple(int(v) for v in np.atleast_1d(shape))
    i, k = np.meshgrid(np.arange(shape[0], dtype=np.uint64), np.arange(shape[1], dtype=np.uint64), indexing='ij')
    h = (i * np.uint64(7919) + k * np.uint64(104729) + np.uint64(seed)) * np.uint64(2654435761)
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF)
    return h.astype(np.float64) / 4294967296.0


def batch_sequence(all_seg, seg, separator, tpc_batch_size, tpc_borders):
    """(event id, bool mask) in the order the reference's batch loop visits them (larndsim/util/batching.py:40-67): events
    ascending, TPC groups in index order, a segment belongs to the first group that holds one of its end points; empty
    masks included.  The checker of batching.assign_batches."""
    from larndsim_amd import batching
    borders = np.sort(np.asarray(tpc_borders), axis=-1)
    taken = np.zeros(seg.shape[0], dtype=bool)
    for ev in np.unique(all_seg[separator]):
        in_event = seg[separator] == ev
        for first in range(0, borders.shape[0], tpc_batch_size):
            inside = np.zeros(seg.shape[0], dtype=bool)
            inside[batching.select_active_volume(seg, borders[first:first + tpc_batch_size])] = True
            mask = in_event & inside & ~taken
            taken |= mask
            yield ev, mask
