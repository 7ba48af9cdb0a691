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
        lib.claim_chain(self)
        self.ctx = lib.context(device=device, noise_zero=False)
        self._generation = lib.consts_generation()
        self.n = 0
        if response is not None:
            lib.set_response(response, self.ctx)

    def seed_rng(self, seed, n_states=1024 * 256):
        """``create_xoroshiro128p_states(1024*256, seed=rand_seed)`` of the driver (cli/simulate_pixels.py:396): needed before
        ``run`` when a FEE noise charge is non-zero.  Row u of a launch draws from state u of the table."""
        from . import rng
        self.rng_states = rng.create_xoroshiro128p_states(n_states, seed, self.ctx)
        return self.rng_states

    def extend_rng(self, n_states, seed):
        """``maybe_create_rng_states(n, seed, rng_states)`` (cli/simulate_pixels.py:92-104) on the ctx's table."""
        lib.check(lib.load().ldsim_rng_extend(self.ctx, C.c_int64(int(n_states)), C.c_uint64(int(seed) & (2 ** 64 - 1))))

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

    def run_async(self, seg_begin=0, seg_end=None, want_fractions=False):
        """``run`` on a worker thread: the C call blocks (a chain launch reads sizes back from the device five times) but releases
        the interpreter, so the caller can do host work that does not enter the ctx meanwhile -- the bundled driver builds the
        previous launch's packets (``ldsim_packets_build`` takes no ctx).  ``wait()`` joins and returns the launch's statistics.
        Any other call into the ctx before ``wait()`` is refused by the library (LDSIM_ESTATE: a ctx serves one thread at a time)."""
        import threading
        if getattr(self, "_worker", None) is not None:
            raise lib.LdsimError("a chain launch is already in flight: wait() first")
        box = {}

        def work():
            try:
                box["st"] = self.run(seg_begin, seg_end, want_fractions)
            except BaseException as e:          # handed to wait()
                box["err"] = e
        self._worker = (threading.Thread(target=work, name="ldsim-chain-launch"), box)
        self._worker[0].start()

    def wait(self):
        """Join the launch started by ``run_async`` and return its statistics (or raise what it raised)."""
        w = getattr(self, "_worker", None)
        if w is None:
            raise lib.LdsimError("no chain launch in flight")
        w[0].join()
        self._worker = None
        if "err" in w[1]:
            raise w[1]["err"]
        return w[1]["st"]

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

    def _pinned(self, name, shape, dtype):
        """view of a page-locked buffer kept per output (grown when too small); valid until the next pinned download"""
        n = int(np.prod(shape))
        pool = self.__dict__.setdefault("_pinned_pool", {})
        buf = pool.get(name)
        if buf is None or buf.size < n or buf.dtype != np.dtype(dtype):
            buf = pool[name] = lib.pinned_array((max(n + n // 4, 1),), dtype)
        return buf[:n].reshape(shape)

    def download(self, fractions=None, pinned=False):
        """Per-unique-(batch, pixel) results of the last run(), reference dtypes.  ``pinned``: the arrays are views of
        page-locked buffers owned by the chain (PCIe-rate copies, no 1 GB allocation per call); they are overwritten by the
        next pinned download."""
        U = int(self.stats.n_unique)
        A, M = consts.sim.MAX_ADC_VALUES, consts.sim.MAX_TRACKS_PER_PIXEL
        want_fr = fractions if fractions is not None else self._want_fractions
        if pinned:
            out = dict(unique_pix=self._pinned("unique_pix", (U,), np.int32), batch=self._pinned("batch", (U,), np.int32),
                       adc_list=self._pinned("adc_list", (U, A), np.float64),
                       adc_ticks_list=self._pinned("adc_ticks_list", (U, A), np.float64),
                       adc_digit=self._pinned("adc_digit", (U, A), np.float64),
                       track_pixel_map=self._pinned("track_pixel_map", (U, M), np.int64))
            fr = self._pinned("current_fractions", (U, A, M), np.float64) if want_fr else None
        else:
            out = dict(unique_pix=np.zeros(U, dtype=np.int32), batch=np.zeros(U, dtype=np.int32),
                       adc_list=np.zeros((U, A)), adc_ticks_list=np.zeros((U, A)), adc_digit=np.zeros((U, A)),
                       track_pixel_map=np.full((U, M), -1, dtype=np.int64))
            fr = np.zeros((U, A, M)) if want_fr else None
        if fr is not None:
            out['current_fractions'] = fr
        lib.check(lib.load().ldsim_chain_download(self.ctx, C.c_int64(U), lib.ptr(out['unique_pix']),
                                                  lib.ptr(out['batch']), lib.ptr(out['adc_list']),
                                                  lib.ptr(out['adc_ticks_list']), lib.ptr(out['adc_digit']),
                                                  lib.ptr(out['track_pixel_map']), lib.ptr(fr)))
        return out

    def download_compact(self):
        """The last run()'s results in compact form (``ldsim_chain_compact_build`` / ``_download``): what the exporter reads --
        hit pixels, their hits and the fractions of the track slots they have -- gathered on the device first: a few MB over
        PCIe instead of 13 KB per unique pixel.  ``expand_compact`` turns it into the dense rows of the hit pixels."""
        from .comm import HIT_ROW
        sizes = (C.c_int64 * 4)()
        lib.check(lib.load().ldsim_chain_compact_build(self.ctx, sizes))
        n_hp, n_hits, n_trk, n_frac = (int(v) for v in sizes)
        out = dict(hit_pixels=np.zeros((n_hp, 5), dtype=np.int32), track_segments=np.zeros(n_trk, dtype=np.int64),
                   hit_rows=np.zeros(n_hits, dtype=HIT_ROW), hit_charge=np.zeros(n_hits), fractions=np.zeros(n_frac),
                   has_fractions=bool(self._want_fractions))
        lib.check(lib.load().ldsim_chain_compact_download(self.ctx, lib.ptr(out["hit_pixels"]), lib.ptr(out["track_segments"]),
                                                          lib.ptr(out["hit_rows"]), lib.ptr(out["hit_charge"]),
                                                          lib.ptr(out["fractions"])))
        return out

    def download_async(self, fractions=None):
        """Start copying the results of the last run() to page-locked host arrays on the library's copy stream and return them
        at once (``ldsim_chain_download_async``): the next ``run()`` overlaps with the transfer.  The arrays hold the rows
        after ``wait_download()``; two sets of host buffers alternate, so the arrays of one call stay valid until the call
        after the next one."""
        U = int(self.stats.n_unique)
        A, M = consts.sim.MAX_ADC_VALUES, consts.sim.MAX_TRACKS_PER_PIXEL
        want_fr = fractions if fractions is not None else self._want_fractions
        slot = self.__dict__["_async_slot"] = 1 - self.__dict__.get("_async_slot", 1)
        tag = f"@{slot}"
        out = dict(unique_pix=self._pinned("unique_pix" + tag, (U,), np.int32), batch=self._pinned("batch" + tag, (U,), np.int32),
                   adc_list=self._pinned("adc_list" + tag, (U, A), np.float64),
                   adc_ticks_list=self._pinned("adc_ticks_list" + tag, (U, A), np.float64),
                   adc_digit=self._pinned("adc_digit" + tag, (U, A), np.float64),
                   track_pixel_map=self._pinned("track_pixel_map" + tag, (U, M), np.int64))
        fr = self._pinned("current_fractions" + tag, (U, A, M), np.float64) if want_fr else None
        if fr is not None:
            out['current_fractions'] = fr
        lib.check(lib.load().ldsim_chain_download_async(self.ctx, C.c_int64(U), lib.ptr(out['unique_pix']),
                                                        lib.ptr(out['batch']), lib.ptr(out['adc_list']),
                                                        lib.ptr(out['adc_ticks_list']), lib.ptr(out['adc_digit']),
                                                        lib.ptr(out['track_pixel_map']), lib.ptr(fr)))
        return out

    def wait_download(self):
        """Block until the transfer started by ``download_async`` has landed."""
        lib.check(lib.load().ldsim_chain_download_wait(self.ctx))

    # ---- device-resident light leg (cli/simulate_pixels.py:749-797, 1120-1153) ------------------------------------------------
    def light_incidence(self, lut=None, n_out=None):
        """``lightLUT.calculate_light_incidence`` over all resident segments (after ``quench_drift``); the arrays stay in
        HBM.  ``lut``: the structured LUT (uploaded unless already resident); ``n_out``: output channels, default all."""
        self._check_constants()
        lib.set_light(lut, self.ctx)
        n_out = consts.light.N_OP_CHANNEL if n_out is None else n_out
        lib.check(lib.load().ldsim_dev_light_incidence(self.ctx, C.c_int32(int(n_out))))
        self._light_n_out = int(n_out)

    def download_light_incidence(self, seg_begin=0, seg_end=None):
        """(light_sim_dat rows, track_light_voxel rows) of resident segments [seg_begin, seg_end) in the reference's dtypes
        (cli/simulate_pixels.py:760-763); ``segment_id`` is left 0 -- the driver fills it from its own ids."""
        seg_end = self.n if seg_end is None else seg_end
        n = seg_end - seg_begin
        nph = np.zeros((n, self._light_n_out), dtype=np.float32)
        t0 = np.zeros((n, self._light_n_out), dtype=np.float32) if consts.light.LIGHT_TRIG_MODE == 0 else None
        vox = np.zeros((n, 3), dtype=np.int32)
        lib.check(lib.load().ldsim_dev_light_incidence_download(self.ctx, C.c_int64(seg_begin), C.c_int64(seg_end),
                                                                lib.ptr(nph), lib.ptr(t0), lib.ptr(vox)))
        inc = np.zeros((n, self._light_n_out), dtype=[('segment_id', 'u4'), ('n_photons_det', 'f4'), ('t0_det', 'f4')])
        inc['n_photons_det'] = nph
        if t0 is not None:
            inc['t0_det'] = t0
        return inc, vox

    def light_nticks(self, seg_begin, seg_end):
        """``light_sim.get_nticks`` (larndsim/light_sim.py:24-41) for the resident rows of one batch: same expressions, with
        the min / max of ``t0_det`` reduced on the device."""
        light = consts.light
        if light.LIGHT_TRIG_MODE == 0:
            lo, hi, any_ = C.c_float(), C.c_float(), C.c_int32()
            lib.check(lib.load().ldsim_dev_light_t0_range(self.ctx, C.c_int64(seg_begin), C.c_int64(seg_end), C.byref(lo),
                                                          C.byref(hi), C.byref(any_)))
            if any_.value:
                start_time = np.float32(lo.value) - light.LIGHT_WINDOW[0]
                end_time = np.float32(hi.value) + light.LIGHT_WINDOW[1]
                return int(np.ceil((end_time - start_time) / light.LIGHT_TICK_SIZE)), start_time
        return int((light.LIGHT_WINDOW[1] + light.LIGHT_WINDOW[0]) / light.LIGHT_TICK_SIZE), 0

    def sum_light(self, seg_begin, seg_end, op_channel, segment_track_id=None, max_truth=None, max_ticks=int(5e4)):
        """``light_sim.sum_light_signals`` for the batch [seg_begin, seg_end) on the resident incidence arrays
        (cli/simulate_pixels.py:1120-1153).  Returns (n_ticks, start_time); fetch the arrays with ``download_light``."""
        self._check_constants()
        mt = consts.sim.MAX_MC_TRUTH_IDS if max_truth is None else max_truth
        if not mt and consts.light.LIGHT_TRIG_MODE != 0:
            # The driver's loop over hundreds of small batches without truth slots (ndlar): the library call returns with its two
            # kernels in flight, so what the loop costs is this wrapper -- the channel array's pointer and the call's fixed
            # arguments are kept from the last call with the same array object (7.5 -> 2.5 us per call)
            fast = self.__dict__.get("_sum_light_fast")
            lw = consts.light.LIGHT_WINDOW
            key = (max_ticks, lw[0], lw[1], consts.light.LIGHT_TICK_SIZE)
            if fast is None or fast[0] is not op_channel or fast[1] != key:
                n_ticks, t_start = self.light_nticks(seg_begin, seg_end)       # (trigger mode 1: the same for every batch)
                n_ticks = min(n_ticks, max_ticks)
                opc = np.ascontiguousarray(op_channel, dtype=np.int32)
                f = lib.load().ldsim_dev_sum_light
                fast = (op_channel, key, opc, f, lib.ptr(opc), C.c_int32(opc.shape[0]), C.c_int32(0),
                        C.c_double(float(t_start)), C.c_int32(int(n_ticks)), int(n_ticks), t_start,
                        (opc.shape[0], int(n_ticks), 0))
                if opc is op_channel:          # (a converted copy would not follow in-place changes of the caller's array)
                    self._sum_light_fast = fast
            rc = fast[3](self.ctx, C.c_int64(seg_begin), C.c_int64(seg_end), fast[4], fast[5], None, fast[6], fast[7], fast[8])
            if rc:
                lib.check(rc)
            self._light_shape = fast[11]
            lib.set_light_shape_tuple(fast[11])
            return fast[9], fast[10]
        n_ticks, t_start = self.light_nticks(seg_begin, seg_end)
        n_ticks = min(n_ticks, max_ticks)
        opc = np.ascontiguousarray(op_channel, dtype=np.int32)
        tid = None
        if mt:
            tid = (np.arange(seg_begin, seg_end, dtype=np.int64) if segment_track_id is None
                   else np.ascontiguousarray(segment_track_id, dtype=np.int64))
            if tid.shape[0] != seg_end - seg_begin:
                raise ValueError("segment_track_id must have one entry per segment of the range")
        lib.check(lib.load().ldsim_dev_sum_light(self.ctx, C.c_int64(seg_begin), C.c_int64(seg_end), lib.ptr(opc),
                                                 C.c_int32(opc.shape[0]), lib.ptr(tid), C.c_int32(int(mt)),
                                                 C.c_double(float(t_start)), C.c_int32(int(n_ticks))))
        self._light_shape = (opc.shape[0], int(n_ticks), int(mt))
        lib.set_light_shape(self._light_shape)
        return n_ticks, t_start

    def light_response(self, fluctuate=True):
        """``calc_scintillation_effect`` -> ``calc_stat_fluctuations`` (if ``fluctuate``; needs ``seed_rng``) ->
        ``calc_light_detector_response`` on the photon sum of the last ``sum_light``, in HBM
        (cli/simulate_pixels.py:1159-1180).  ``light_sim.get_triggers(None, ...)`` / ``light_sim.sim_triggers(.., None, ..)``
        then work on the resident result; ``download_light_response`` fetches it."""
        self._check_constants()
        light = consts.light
        nd = self._light_shape[0]
        gain = np.ascontiguousarray(light.LIGHT_GAIN, dtype=np.float64)
        if gain.shape[0] < nd:
            raise IndexError(f"LIGHT_GAIN has {gain.shape[0]} entries, the photon sum has {nd} rows")
        imp = np.ascontiguousarray(light.IMPULSE_MODEL, dtype=np.float64)
        lib.check(lib.load().ldsim_dev_light_response(self.ctx, lib.ptr(gain), lib.ptr(imp), C.c_int32(imp.shape[0]),
                                                      C.c_int32(int(bool(fluctuate)))))

    def download_light_response(self, stages=False, truth=True):
        """response f4 [n_det][n_ticks] (+ true ids i8, true photons f8); with ``stages`` also the scintillation and the
        fluctuated arrays: (scint, disc, response, ids, photons)."""
        nd, nt, mt = self._light_shape
        resp = np.zeros((nd, nt), dtype=np.float32)
        sc = np.zeros((nd, nt), dtype=np.float32) if stages else None
        di = np.zeros((nd, nt), dtype=np.float32) if stages else None
        tid = np.full((nd, nt, mt if truth else 0), -1, dtype=np.int64)
        tph = np.zeros((nd, nt, mt if truth else 0))
        lib.check(lib.load().ldsim_dev_light_response_download(
            self.ctx, lib.ptr(sc), lib.ptr(di), lib.ptr(resp), lib.ptr(tid) if (mt and truth) else None,
            lib.ptr(tph) if (mt and truth) else None))
        return (sc, di, resp, tid, tph) if stages else (resp, tid, tph)

    def light_response_ms(self):
        a, b, c = C.c_double(0), C.c_double(0), C.c_double(0)
        lib.check(lib.load().ldsim_light_response_ms(self.ctx, C.byref(a), C.byref(b), C.byref(c)))
        return {"scintillation": a.value, "fluctuations": b.value, "detector_response": c.value}

    def download_light(self, truth=True):
        """(light_sample_inc f4 [n_det][n_ticks], true_track_id i8 [..][max_truth], true_photons f8 [..][max_truth])."""
        nd, nt, mt = self._light_shape
        out = np.zeros((nd, nt), dtype=np.float32)
        want = bool(mt and truth)          # (the truth arrays are GBs at 50 slots: only made when asked for)
        tid = np.full((nd, nt, mt if want else 0), -1, dtype=np.int64)
        tph = np.zeros((nd, nt, mt if want else 0))
        lib.check(lib.load().ldsim_dev_light_download(self.ctx, lib.ptr(out), lib.ptr(tid) if want else None,
                                                      lib.ptr(tph) if want else None))
        return out, tid, tph

    def light_kernel_ms(self):
        a, b = C.c_double(), C.c_double()
        lib.check(lib.load().ldsim_light_kernel_ms(self.ctx, C.byref(a), C.byref(b)))
        return dict(incidence_ms=a.value, sum_ms=b.value)

    def compact_hits(self):
        """(device pointer, n_rows, row_bytes) of the compact hit list of the last run()."""
        p, n, rb = C.c_void_p(), C.c_int64(), C.c_int32()
        lib.check(lib.load().ldsim_chain_compact_hits(self.ctx, C.byref(p), C.byref(n), C.byref(rb)))
        return p.value, n.value, rb.value


def expand_compact(c, lead_rows=False):
    """Dense rows of the hit pixels from ``ChargeChain.download_compact()``: the arrays ``download()`` returns, restricted to
    the unique pixels that hold a hit (``row`` = their index in the full arrays): ``unique_pix``, ``batch``, ``adc_list``,
    ``first_of_batch`` (the pixel is row 0 of its batch in the full arrays: the exporter's clock-rollover bookkeeping looks at the
    first row it is handed), ``adc_ticks_list``, ``adc_digit`` [n][A] (slots past a pixel's last hit: charge 0, tick 0, the pedestal code -- what the dense
    arrays hold there), ``track_pixel_map`` [n][M] (-1 pad) and ``current_fractions`` [n][A][M] (written slots only: the dense
    array's un-normalised residue in the slot after the last hit, fee.py:572-573, is not part of the result).

    ``lead_rows``: a batch whose first unique pixel holds no hit keeps that hit-less row in front of its hit pixels (no charge, no
    track) -- the reference's exporter keeps its clock-rollover state in row 0 of what it is handed (fee.py:164-183, 267-277), and
    the driver counts the batches of an export from what it is handed, so a batch without any hit still shows up as its first row.
    The compact form carries the first row of every batch (0 hits) since round 4; for input without it the row is inserted here
    (the pixel id of the row after it, ``row`` -1)."""
    from . import packets
    A, M = consts.sim.MAX_ADC_VALUES, consts.sim.MAX_TRACKS_PER_PIXEL
    hp = c["hit_pixels"]
    if not lead_rows and len(hp) and (hp[:, 3] == 0).any():          # (hit-less first rows carry no hits, slots or fractions)
        hp = hp[hp[:, 3] > 0]
    n0 = hp.shape[0]
    nh, nt = hp[:, 3].astype(np.int64), (hp[:, 4] & 255).astype(np.int64)
    first = (hp[:, 4] & 256) != 0
    ped = float(packets._digitize0())                   # fee.digitize(0): what an unwritten slot digitises to
    pos = np.arange(n0)                                 # output row of hit pixel i
    n = n0
    if lead_rows and n0:
        need = np.r_[True, hp[1:, 2] != hp[:-1, 2]] & ~first          # batches that start with a hit-less pixel
        pos = pos + np.cumsum(need)
        n = n0 + int(need.sum())
    out = dict(row=np.full(n, -1, dtype=hp.dtype), first_of_batch=np.ones(n, dtype=bool), unique_pix=np.zeros(n, dtype=hp.dtype),
               batch=np.zeros(n, dtype=hp.dtype), adc_list=np.zeros((n, A)),
               adc_ticks_list=np.zeros((n, A)), adc_digit=np.full((n, A), ped), track_pixel_map=np.full((n, M), -1, dtype=np.int64))
    out["row"][pos] = hp[:, 0]
    out["first_of_batch"][pos] = first
    out["unique_pix"][pos] = hp[:, 1]
    out["batch"][pos] = hp[:, 2]
    if n != n0:
        lead = pos[need] - 1
        out["unique_pix"][lead] = hp[need, 1]
        out["batch"][lead] = hp[need, 2]
    # hits: pixel after pixel, slot 0 up
    pix_of_hit = np.repeat(pos, nh)
    slot = c["hit_rows"]["slot"].astype(np.int64)
    out["adc_list"][pix_of_hit, slot] = c["hit_charge"]
    out["adc_ticks_list"][pix_of_hit, slot] = c["hit_rows"]["tick"]
    out["adc_digit"][pix_of_hit, slot] = c["hit_rows"]["adc"]
    # track slots: pixel after pixel
    pix_of_trk = np.repeat(pos, nt)
    m_of_trk = np.arange(int(nt.sum())) - np.repeat(np.cumsum(nt) - nt, nt)
    out["track_pixel_map"][pix_of_trk, m_of_trk] = c["track_segments"]
    if c.get("has_fractions"):
        fr = np.zeros((n, A, M))
        per_hit = np.repeat(nt, nh)                                 # fraction entries of every hit
        h_of_f = np.repeat(np.arange(len(pix_of_hit)), per_hit)
        m_of_f = np.arange(int(per_hit.sum())) - np.repeat(np.cumsum(per_hit) - per_hit, per_hit)
        fr[pix_of_hit[h_of_f], slot[h_of_f], m_of_f] = c["fractions"]
        out["current_fractions"] = fr
    return out
