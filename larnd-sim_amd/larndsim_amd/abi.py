"""
ctypes mirror of include/ldsim.h (struct layouts + ``pack_consts``).

Only declarations live here; loading the shared library is in ``lib.py``.
"""
import ctypes as C

import numpy as np

from . import consts
from .layout import LdsimTrackLayout, NFIELDS  # noqa: F401  (re-exported)

LDSIM_MAX_TPC = 128

LDSIM_OK, LDSIM_EINVAL, LDSIM_EHIP, LDSIM_ENOSPC, LDSIM_ESTATE, LDSIM_ENODEV = 0, -1, -2, -3, -4, -5


class LdsimConsts(C.Structure):
    _fields_ = [
        ("box_alpha", C.c_double), ("box_beta", C.c_double), ("birks_ab", C.c_double), ("birks_kb", C.c_double),
        ("w_ion", C.c_double),
        ("w_ph", C.c_double), ("scint_prescale", C.c_double),
        ("e_field", C.c_double), ("lar_density", C.c_double), ("v_drift", C.c_double),
        ("electron_lifetime", C.c_double), ("long_diff", C.c_double), ("tran_diff", C.c_double),
        ("n_tpc", C.c_int32), ("default_plane_index", C.c_int32),
        ("tpc_borders", C.c_double * 2 * 3 * LDSIM_MAX_TPC),
        ("n_pixels", C.c_int32 * 2), ("sampled_points", C.c_int32), ("n_time_ticks", C.c_int32),
        ("pixel_pitch", C.c_double),
        ("time_sampling", C.c_double), ("time_padding", C.c_double), ("time_window", C.c_double),
        ("time_interval", C.c_double * 2),
        ("response_sampling", C.c_double), ("response_bin_size", C.c_double),
        ("discrimination_threshold", C.c_double), ("clock_cycle", C.c_double), ("buffer_risetime", C.c_double),
        ("gain", C.c_double), ("v_cm", C.c_double), ("v_ref", C.c_double), ("v_pedestal", C.c_double),
        ("adc_hold_delay", C.c_int32), ("adc_busy_delay", C.c_int32), ("reset_cycles", C.c_int32),
        ("adc_counts", C.c_int32),
        ("reset_noise_charge", C.c_double), ("uncorrelated_noise_charge", C.c_double),
        ("discriminator_noise", C.c_double),
        ("max_tracks_per_pixel", C.c_int32), ("max_adc_values", C.c_int32),
        ("light_trig_mode", C.c_int32), ("enable_lut_smearing", C.c_int32),
        ("n_op_channel", C.c_int32), ("max_mc_truth_ids", C.c_int32),
        ("light_tick_size", C.c_double), ("mc_truth_threshold", C.c_double),
        ("light_window", C.c_double * 2), ("singlet_fraction", C.c_double), ("tau_s", C.c_double), ("tau_t", C.c_double),
        ("light_response_time", C.c_double), ("light_oscillation_period", C.c_double),
        ("impulse_tick_size", C.c_double), ("sipm_response_model", C.c_int32), ("mc_sample_multiplier", C.c_int32),
        ("min_step_size", C.c_double),
        ("light_trig_window", C.c_double * 2), ("light_digit_sample_spacing", C.c_double),
        ("light_det_noise_sample_spacing", C.c_double), ("light_nbit", C.c_int32), ("op_channel_per_trig", C.c_int32),
    ]


class LdsimChainStats(C.Structure):
    _fields_ = [("n_segments", C.c_int64), ("n_pairs", C.c_int64), ("n_unique", C.c_int64),
                ("n_batches", C.c_int64), ("n_overflow", C.c_int64),
                ("max_active", C.c_int32), ("max_neigh", C.c_int32), ("max_length", C.c_int32),
                ("n_ambiguous", C.c_int32), ("n_dfma", C.c_int64), ("n_fallback", C.c_int64), ("n_samples", C.c_int64), ("n_wbuf", C.c_int64),
                ("n_dfma_useful", C.c_int64)]


ABI_VERSION = 8      # include/ldsim.h LDSIM_ABI_VERSION: the struct layouts of this file


def pack_consts(noise_zero=False):
    """Freeze the current ``consts`` namespaces into the plain struct the C-ABI takes."""
    d, l, s, p = consts.detector, consts.light, consts.sim, consts.physics
    c = LdsimConsts()
    c.box_alpha, c.box_beta, c.birks_ab, c.birks_kb, c.w_ion = p.BOX_ALPHA, p.BOX_BETA, p.BIRKS_Ab, p.BIRKS_kb, p.W_ION
    c.w_ph, c.scint_prescale = l.W_PH, l.SCINT_PRESCALE
    c.e_field, c.lar_density, c.v_drift = d.E_FIELD, d.LAR_DENSITY, d.V_DRIFT
    c.electron_lifetime, c.long_diff, c.tran_diff = d.ELECTRON_LIFETIME, d.LONG_DIFF, d.TRAN_DIFF
    B = np.asarray(d.TPC_BORDERS, dtype=np.float64)
    if B.shape[0] > LDSIM_MAX_TPC:
        raise ValueError(f"{B.shape[0]} TPCs > LDSIM_MAX_TPC")
    c.n_tpc = B.shape[0]
    c.default_plane_index = d.DEFAULT_PLANE_INDEX
    flat = np.zeros((LDSIM_MAX_TPC, 3, 2))
    flat[:B.shape[0]] = B
    C.memmove(C.addressof(c.tpc_borders), flat.ctypes.data, flat.nbytes)
    c.n_pixels[0], c.n_pixels[1] = int(d.N_PIXELS[0]), int(d.N_PIXELS[1])
    c.sampled_points = d.SAMPLED_POINTS
    c.n_time_ticks = len(d.TIME_TICKS)
    c.pixel_pitch = d.PIXEL_PITCH
    c.time_sampling, c.time_padding, c.time_window = d.TIME_SAMPLING, d.TIME_PADDING, d.TIME_WINDOW
    c.time_interval[0], c.time_interval[1] = float(d.TIME_INTERVAL[0]), float(d.TIME_INTERVAL[1])
    c.response_sampling, c.response_bin_size = d.RESPONSE_SAMPLING, d.RESPONSE_BIN_SIZE
    c.discrimination_threshold = d.DISCRIMINATION_THRESHOLD
    c.clock_cycle, c.buffer_risetime, c.gain = d.CLOCK_CYCLE, d.BUFFER_RISETIME, d.GAIN
    c.v_cm, c.v_ref, c.v_pedestal = d.V_CM, d.V_REF, d.V_PEDESTAL
    c.adc_hold_delay, c.adc_busy_delay, c.reset_cycles = d.ADC_HOLD_DELAY, d.ADC_BUSY_DELAY, d.RESET_CYCLES
    c.adc_counts = d.ADC_COUNTS
    z = 0.0 if noise_zero else 1.0
    c.reset_noise_charge = d.RESET_NOISE_CHARGE * z
    c.uncorrelated_noise_charge = d.UNCORRELATED_NOISE_CHARGE * z
    c.discriminator_noise = d.DISCRIMINATOR_NOISE * z
    c.max_tracks_per_pixel, c.max_adc_values = s.MAX_TRACKS_PER_PIXEL, s.MAX_ADC_VALUES
    c.light_trig_mode, c.enable_lut_smearing = l.LIGHT_TRIG_MODE, int(bool(l.ENABLE_LUT_SMEARING))
    c.n_op_channel, c.max_mc_truth_ids = int(l.N_OP_CHANNEL), int(s.MAX_MC_TRUTH_IDS)
    c.light_tick_size, c.mc_truth_threshold = l.LIGHT_TICK_SIZE, s.MC_TRUTH_THRESHOLD
    c.light_window[0], c.light_window[1] = float(l.LIGHT_WINDOW[0]), float(l.LIGHT_WINDOW[1])
    c.singlet_fraction, c.tau_s, c.tau_t = l.SINGLET_FRACTION, l.TAU_S, l.TAU_T
    c.light_response_time, c.light_oscillation_period = l.LIGHT_RESPONSE_TIME, l.LIGHT_OSCILLATION_PERIOD
    c.impulse_tick_size, c.sipm_response_model = l.IMPULSE_TICK_SIZE, int(l.SIPM_RESPONSE_MODEL)
    c.mc_sample_multiplier, c.min_step_size = int(s.MC_SAMPLE_MULTIPLIER), float(s.MIN_STEP_SIZE)
    c.light_trig_window[0], c.light_trig_window[1] = float(l.LIGHT_TRIG_WINDOW[0]), float(l.LIGHT_TRIG_WINDOW[1])
    c.light_digit_sample_spacing = float(l.LIGHT_DIGIT_SAMPLE_SPACING)
    c.light_det_noise_sample_spacing = float(l.LIGHT_DET_NOISE_SAMPLE_SPACING)
    c.light_nbit, c.op_channel_per_trig = int(l.LIGHT_NBIT), int(l.OP_CHANNEL_PER_TRIG)
    return c


class LdsimPacketsIn(C.Structure):
    """include/ldsim.h LdsimPacketsIn (ldsim_packets_build)"""
    _fields_ = [("n_rows", C.c_int64), ("row_event", C.c_void_p), ("row_base", C.c_void_p), ("row_ts_s", C.c_void_p),
                ("row_ok", C.c_void_p), ("row_io_group", C.c_void_p), ("row_io_channel", C.c_void_p), ("row_chip", C.c_void_p),
                ("row_channel", C.c_void_p), ("row_hit0", C.c_void_p), ("row_trk0", C.c_void_p), ("row_frac0", C.c_void_p),
                ("hit_adc", C.c_void_p), ("hit_tick", C.c_void_p), ("hit_frac", C.c_void_p), ("trk_segment", C.c_void_p),
                ("trk_traj", C.c_void_p), ("base0", C.c_int64), ("first_row_is_row0", C.c_int32), ("light_trig_mode", C.c_int32),
                ("n_trig", C.c_int64), ("trig_time", C.c_void_p), ("trig_event", C.c_void_p), ("trig_module", C.c_void_p),
                ("n_io_groups", C.c_int32), ("io_groups", C.c_void_p), ("n_modules", C.c_int32), ("module_ids", C.c_void_p),
                ("module_group0", C.c_void_p), ("module_groups", C.c_void_p), ("clock_reset_period", C.c_int64),
                ("clock_cycle", C.c_double), ("mus", C.c_double), ("s", C.c_double), ("n_keep", C.c_int32), ("max_tracks", C.c_int32)]
