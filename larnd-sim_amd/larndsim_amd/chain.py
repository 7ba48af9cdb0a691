"""
Device-resident charge chain: segments stay in HBM from quench to ADC.

Replaces the body of the reference's batch loop (cli/simulate_pixels.py:917-1105: max_pixels ->
get_pixels -> unique -> time_intervals -> tracks_current -> pixel_index_map -> get_track_pixel_map2 ->
sum_pixel_signals -> get_adc_values -> digitize) for any number of (event, TPC-group) batches per call.
"""
import ctypes as C

import numpy as np

from . import consts, lib
from .abi import LdsimChainStats
from .layout import make_layout


class ChargeChain:
    def __init__(self, response=None, device=None):
        self.ctx = lib.context(device=device, noise_zero=False)
        self._generation = lib.consts_generation()
        self.n = 0
        if response is not None:
            lib.set_response(response, self.ctx)

    def _check_constants(self):
        """The ctx is process-wide and this object froze its constants at construction: refuse to compute once anything
        (another ChargeChain, a stage call after ``consts`` was reloaded) has frozen different ones."""
        if lib.consts_generation() != self._generation:
            raise lib.LdsimError("the detector constants frozen in the GPU context changed since this ChargeChain was "
                                 "created: create a new ChargeChain (and upload again) after loading another configuration")

    def refresh_constants(self):
        """Adopt the constants currently in ``consts`` for this chain (e.g. after changing DISCRIMINATION_THRESHOLD),
        keeping the resident segments.  What was derived under the old constants is the caller's business: drifted
        segments are not recomputed, and pixel tables of another geometry make the next run fail (LDSIM_ESTATE)."""
        self.ctx = lib.context(noise_zero=False)
        self._generation = lib.consts_generation()

    def upload(self, tracks, batch_id=None):
        """H2D + unpack into the SoA segment store. ``batch_id``: int32 per segment, non-decreasing, <0 = skip."""
        self._check_constants()
        lay = make_layout(tracks.dtype)
        self._layout = lay
        b = None if batch_id is None else np.ascontiguousarray(batch_id, dtype=np.int32)
        tr = np.ascontiguousarray(tracks)
        lib.check(lib.load().ldsim_segments_upload(self.ctx, lib.ptr(tr), C.c_int64(tr.shape[0]), C.byref(lay),
                                                   lib.ptr(b)))
        self.n = tr.shape[0]

    def reset(self):
        """Re-unpack the resident records (undo quench/drift) without another H2D copy."""
        lib.check(lib.load().ldsim_segments_reset(self.ctx))

    def synchronize(self):
        lib.check(lib.load().ldsim_synchronize(self.ctx))

    def quench_drift(self, mode=None):
        self._check_constants()
        mode = consts.physics.BIRKS if mode is None else mode
        lib.check(lib.load().ldsim_dev_quench_drift(self.ctx, C.c_int32(int(mode))))

    def download_segments(self, tracks):
        lay = make_layout(tracks.dtype)
        lib.check(lib.load().ldsim_segments_download(self.ctx, lib.ptr(tracks), C.c_int64(tracks.shape[0]),
                                                     C.byref(lay)))
        return tracks

    def run(self, seg_begin=0, seg_end=None, want_fractions=False):
        self._check_constants()
        seg_end = self.n if seg_end is None else seg_end
        st = LdsimChainStats()
        lib.check(lib.load().ldsim_charge_chain(self.ctx, C.c_int64(seg_begin), C.c_int64(seg_end),
                                                C.c_int32(int(want_fractions)), C.byref(st)))
        self.stats = st
        self._want_fractions = bool(want_fractions)
        return st

    def set_pixel_thresholds(self, keys, values, default):
        """Per-pixel discrimination thresholds of the fused chain: ``pixel_thresholds_lut[unique_pix]`` of the reference
        driver (cli/simulate_pixels.py:1079-1084).  Call after the constants are loaded."""
        self._set_table("ldsim_set_pixel_thresholds", keys, values, default)

    def set_pixel_gains(self, keys, values, default):
        """Per-pixel gains for the digitisation: ``pixel_gains_lut[unique_pix]`` (cli/simulate_pixels.py:1097-1100)."""
        self._set_table("ldsim_set_pixel_gains", keys, values, default)

    def clear_pixel_tables(self):
        """Back to DISCRIMINATION_THRESHOLD * e and GAIN * mV / e for every pixel."""
        lib.check(lib.load().ldsim_clear_pixel_tables(self.ctx))

    def _set_table(self, fn, keys, values, default):
        k = np.ascontiguousarray(keys, dtype=np.int32).ravel()
        v = np.ascontiguousarray(values, dtype=np.float64).ravel()
        if k.shape != v.shape:
            raise ValueError("keys and values differ in length")
        self._check_constants()   # the table is dense over the pixel geometry this chain was created under
        lib.check(getattr(lib.load(), fn)(self.ctx, lib.ptr(k), lib.ptr(v), C.c_int64(k.size), C.c_double(float(default))))

    def kernel_ms(self):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        lib.check(lib.load().ldsim_chain_kernel_ms(self.ctx, C.byref(a), C.byref(b), C.byref(c)))
        w, m, f = C.c_double(), C.c_double(), C.c_double()
        lib.check(lib.load().ldsim_chain_kernel_ms_detail(self.ctx, C.byref(w), C.byref(m), C.byref(f)))
        return dict(current_ms=a.value, adc_ms=b.value, total_ms=c.value, weights_ms=w.value, mac_ms=m.value,
                    fallback_ms=f.value)

    def download(self, fractions=None):
        """Per-unique-(batch, pixel) results of the last run(), reference dtypes."""
        U = int(self.stats.n_unique)
        A, M = consts.sim.MAX_ADC_VALUES, consts.sim.MAX_TRACKS_PER_PIXEL
        out = dict(unique_pix=np.zeros(U, dtype=np.int32), batch=np.zeros(U, dtype=np.int32),
                   adc_list=np.zeros((U, A)), adc_ticks_list=np.zeros((U, A)), adc_digit=np.zeros((U, A)),
                   track_pixel_map=np.full((U, M), -1, dtype=np.int64))
        fr = None
        if fractions if fractions is not None else self._want_fractions:
            fr = np.zeros((U, A, M))
            out['current_fractions'] = fr
        lib.check(lib.load().ldsim_chain_download(self.ctx, C.c_int64(U), lib.ptr(out['unique_pix']),
                                                  lib.ptr(out['batch']), lib.ptr(out['adc_list']),
                                                  lib.ptr(out['adc_ticks_list']), lib.ptr(out['adc_digit']),
                                                  lib.ptr(out['track_pixel_map']), lib.ptr(fr)))
        return out

    def compact_hits(self):
        """(device pointer, n_rows, row_bytes) of the compact hit list of the last run()."""
        p, n, rb = C.c_void_p(), C.c_int64(), C.c_int32()
        lib.check(lib.load().ldsim_chain_compact_hits(self.ctx, C.byref(p), C.byref(n), C.byref(rb)))
        return p.value, n.value, rb.value
