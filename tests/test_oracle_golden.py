"""
CPU: pin the oracle (oracle/ldsim_oracle.c) against golden vectors produced by the reference's own
source (oracle/gen_golden.py).  Integer outputs bit-exact; f64 outputs to 1e-13 relative (same libm
functions, same operation order).
"""
import os

import numpy as np
import pytest

import helpers as H
from larndsim_amd import consts
from oracle import oracle as O

CFGS = ["module0", "2x2_no_modvar", "ndlar"]


@pytest.mark.parametrize("cfg", CFGS)
def test_quench_drift(cfg):
    H.load_cfg(cfg)
    g = H.gold(f"qd_{cfg}.npz")
    seg = g["segments_in"]
    for mode, name in ((2, "birks"), (1, "box")):
        r = H.to_ref(seg)
        O.quench(r, mode)
        assert np.array_equal(r["n_electrons"], g[f"{name}_n_electrons"])
        np.testing.assert_allclose(H.f4(r["n_photons"]), g[f"{name}_n_photons"], rtol=0, atol=0)
        if name == "birks":
            H.round_f4(r, ["n_photons"])
            O.drift(r)
            assert np.array_equal(r["pixel_plane"], g["drift_pixel_plane"])
            assert np.array_equal(r["n_electrons"], g["drift_n_electrons"])
            for f in ("long_diff", "tran_diff", "t", "t_start", "t_end"):
                np.testing.assert_allclose(r[f], g["drift_" + f], rtol=1e-15, atol=0, err_msg=f)
    assert (g["drift_pixel_plane"] == consts.detector.DEFAULT_PLANE_INDEX).any() or cfg != "module0"


@pytest.mark.parametrize("cfg", CFGS)
def test_pixels_time_intervals(cfg):
    H.load_cfg(cfg)
    g = H.gold(f"pixels_{cfg}.npz")
    r = H.quench_drift(O, g["segments_in"])
    assert O.max_pixels(r) == int(g["max_pixels"])
    radius = int(g["max_radius"])
    active, neigh, nrad, nlist = O.get_pixels(r, int(g["max_pixels"]), g["neigh"].shape[1], radius)
    assert np.array_equal(active, g["active"])
    assert np.array_equal(neigh, g["neigh"])
    assert np.array_equal(nrad, g["nrad"])
    assert np.array_equal(nlist, g["n_pixels_list"])
    starts, tmax = O.time_intervals(r)
    assert np.array_equal(starts, g["track_starts"])
    assert tmax == int(g["max_length"])
    assert (g["active"] == -1).any()        # the -1 gap quirk is exercised


SAMPLED_SETS = [(c, "") for c in CFGS] + [(c, "corners_") for c in ("module0", "ndlar")
                                          if os.path.exists(os.path.join(os.path.dirname(__file__), "golden", f"sampled_corners_{c}.npz"))]


@pytest.mark.parametrize("cfg,tag", SAMPLED_SETS)
def test_tracks_current_sampled(cfg, tag):
    """Induced current at sampled ticks for diverse (segment, pixel) pairs incl. the pID == -1 quirk slots.  The `corners_`
    sets hold the degenerate geometries: segments hugging a TPC face, 0.5-50 um long, along / perpendicular to the drift axis,
    heavily ionising long ones (oracle/gen_golden.py:corner_segments)."""
    H.load_cfg(cfg)
    g = H.gold(f"sampled_{tag}{cfg}.npz")
    r = H.quench_drift(O, g["segments_in"])
    neigh = g["neigh"]
    T = int(g["max_length"])
    resp = H.response_for(g["response_kind"])
    sig = O.tracks_current(r, neigh, T, resp)
    got = sig[:, :, g["ticks"]]
    ref = g["signals"]
    assert (ref != 0).sum() > (100 if tag else 500)        # the corner sets hold a dozen ticks per pair
    # same operation order as the reference -> f32 outputs agree to the last bit or two
    np.testing.assert_allclose(got, ref, rtol=3e-7, atol=0)
    nz = ref != 0
    assert np.array_equal(got != 0, nz)


def test_chain_downstream_stages():
    """a13-a16 of the oracle vs the reference's own get_track_pixel_map2 / sum_pixel_signals / get_adc_values /
    digitize run on the golden chain (whose full-tick `signals` are oracle-made, spot-checked against the reference)."""
    H.load_cfg("module0")
    g = H.gold("chain_module0.npz")
    r = H.quench_drift(O, g["segments_in"])
    neigh, nrad = g["neigh"], g["nrad"]
    T = int(g["max_length"])
    sig = O.tracks_current(r, neigh, T, H.response_for(g["response_kind"]))
    assert np.array_equal(sig, g["signals"])
    upix = O.unique_pixels(neigh)
    assert np.array_equal(upix, g["unique_pix"])
    pim = O.pixel_index_map(neigh, upix)
    assert np.array_equal(pim, g["pixel_index_map"])
    M = consts.sim.MAX_TRACKS_PER_PIXEL
    tpm = O.track_pixel_map(upix, neigh, nrad, int(nrad.max()) + 1, M)
    assert np.array_equal(tpm, g["track_pixel_map"])
    ps, pts, ovf = O.sum_pixel_signals(g["signals"], g["track_starts"], pim, tpm, len(upix))
    np.testing.assert_allclose(ps, g["pixels_signals"], rtol=1e-13, atol=1e-13 * np.abs(g["pixels_signals"]).max())
    assert np.array_equal(ovf, g["overflow"])
    tt = np.linspace(0, consts.detector.TIME_INTERVAL[1], ps.shape[1] + 1)
    for name in ("default", "low"):
        thr = np.full(len(upix), float(g[f"threshold_{name}"]))
        adc, ticks, frac = O.get_adc_values(ps, pts, tt, thr)
        ref = g[f"adc_integral_{name}"]
        assert np.array_equal(adc != 0, ref != 0)
        np.testing.assert_allclose(adc, ref, rtol=1e-12)
        assert np.array_equal(ticks, g[f"adc_ticks_{name}"])
        hit = ref != 0
        np.testing.assert_allclose(frac[hit], g[f"adc_fractions_{name}"][hit], rtol=1e-10, atol=1e-14)
        assert np.array_equal(O.digitize(adc), g[f"adc_digit_{name}"])
    assert (g["adc_integral_low"] != 0).sum() >= 10


def test_numba_f32_typing_effect_is_recorded():
    """DESIGN.md §2 typing caveat: Numba keeps f32 (op) f32 in single precision for f4 record fields.  The oracle can
    emulate those spots; the induced current then moves by ~2e-7 of the waveform peak (median), <1e-3 worst case."""
    H.load_cfg("module0")
    from larndsim_amd import batching, synth
    seg = synth.make_segments(6, seed=77, segs_per_event=6)
    batching.swap_coordinates(seg)
    O.quench(seg, 2)
    O.drift(seg)
    nmax = O.max_pixels(seg)
    _, neigh, _, _ = O.get_pixels(seg, nmax, 3 * nmax + 6, 1)
    _, T = O.time_intervals(seg)
    resp = synth.make_response("dense")
    a = O.tracks_current(seg, neigh, T, resp).astype(np.float64)
    O.lib().o_set_numba_f32(1)
    try:
        b = O.tracks_current(seg, neigh, T, resp).astype(np.float64)
    finally:
        O.lib().o_set_numba_f32(0)
    pk = np.abs(a).max(-1)
    ok = (neigh >= 0) & (pk > 0)
    rel = np.abs(a - b).max(-1)[ok] / pk[ok]
    assert 0 < np.median(rel) < 5e-6 and rel.max() < 1e-3


@pytest.mark.parametrize("cfg", ["module0", "2x2_no_modvar"])
def test_light_golden(cfg):
    """a17/a18: light incidence (voxel lookup) and photon sum vs the reference on a synthetic LUT."""
    from larndsim_amd import synth
    H.load_cfg(cfg)
    g = H.gold(f"light_{cfg}.npz")
    r = H.quench_drift(O, g["segments_in"])
    lut = synth.make_lut((14, 26, 8), 48, int(g["n_prof"]), int(g["lut_seed"]))
    nph, t0d, vox = O.light_incidence(r, lut)
    assert np.array_equal(vox, g["voxel"])
    assert np.array_equal(nph, g["n_photons_det"])
    assert np.array_equal(t0d, g["t0_det"])
    n_ticks = int(g["n_ticks"])
    Mt = g["true_id"].shape[-1]
    out, tid, tph = O.sum_light_signals(r, vox, np.arange(len(r)), g["n_photons_det"], g["op_channel"], lut,
                                        float(g["t_start"]), n_ticks, g["sorted_indices"], max_truth=Mt)
    # the generator ran with f8 LUT mirrors (f64 product); the real f4 fields make Numba's product f32: <=1 ulp of f32
    np.testing.assert_allclose(out, g["light_sample_inc"], rtol=3e-7, atol=0)
    assert np.array_equal(tid, g["true_id"])
    assert g["light_sample_inc"].sum() > 0


@pytest.mark.parametrize("cfg", ["module0", "2x2_no_modvar"])
def test_light_response_golden(cfg):
    """light_sim.calc_scintillation_effect and calc_light_detector_response: the oracle is bit-identical
    to the reference's own functions -- f4 waveforms (every term's f4 store included), truth ids and truth photons -- for
    the RLC SiPM model (module0 case) and the measured-impulse model with interpolation (2x2 case)."""
    g = H.load_light_response_case(cfg)
    tid, tph = g["true_id"].astype(np.int64), g["true_photons"]
    scint, s_id, s_ph = O.scintillation_effect(g["light_sample_inc"], tid, tph)
    assert np.array_equal(scint, g["scint"]) and scint.dtype == np.float32
    assert np.array_equal(s_id, g["scint_true_id"]) and np.array_equal(s_ph, g["scint_true_photons"])
    assert (scint[-1] == 0).all() and scint.sum() > 0                      # the empty channel stays empty
    resp, r_id, r_ph = O.light_detector_response(g["disc"], consts.light.LIGHT_GAIN, consts.light.IMPULSE_MODEL,
                                                 g["scint_true_id"].astype(np.int64), g["scint_true_photons"])
    assert np.array_equal(resp, g["response"])
    assert np.array_equal(r_id, g["response_true_id"]) and np.array_equal(r_ph, g["response_true_photons"])
    # without truth slots the waveforms are the same
    assert np.array_equal(O.scintillation_effect(g["light_sample_inc"])[0], g["scint"])
    assert np.array_equal(O.light_detector_response(g["disc"], consts.light.LIGHT_GAIN, consts.light.IMPULSE_MODEL)[0],
                          g["response"])



@pytest.mark.parametrize("name", H.LIGHT_WVFM_CASES)
def test_light_waveform_chain_golden(name):
    """calc_stat_fluctuations, get_triggers, gen_light_detector_noise and sim_triggers/digitize_signal of the oracle against
    the reference's own functions (light_sim.py:186-238, 339-619; oracle/gen_golden.py gen_light_wvfm).  The random
    generator under calc_stat_fluctuations is the restated one on both sides (third-party, unpinned): what this pins is
    the Poisson logic, the state indexing and the number of draws."""
    g = H.load_light_wvfm_case(name)
    # Poisson fluctuations
    states = np.ascontiguousarray(g["fluct_states_before"]).view(O.RNG_DTYPE).reshape(-1)
    disc = O.stat_fluctuations(g["fluct_inc"], states)
    assert np.array_equal(disc, g["fluct_disc"])
    assert np.array_equal(states.view('u8').reshape(-1, 2), g["fluct_states_after"])
    assert (disc > 0).sum() > 500 and disc[0, 0] == 0 and disc[0, 1] == 0
    # triggers
    trig, trig_op, trig_type = O.get_triggers(g["response"], g["group_threshold"], g["op_channel"], 0)
    assert np.array_equal(trig, g["trigger_idx"]) and np.array_equal(trig_op, g["trigger_op_channel_idx"])
    assert np.array_equal(trig_type, g["trigger_type"])
    assert len(O.get_triggers(g["response"], g["group_threshold"], g["op_channel"], 1)[0]) == int(g["n_trig_subbatch1"])
    # noise with the recorded phases: the values are integers x 2**(16-NBIT); the FFT sizes are the reference's
    for n in (1500, 1501):
        ref = g[f"noise_{n}"]
        ph = H.det_phases((ref.shape[0], n // 2 + 1), int(g[f"noise_{n}_phase_seed"]))
        got = O.gen_light_detector_noise(ref.shape, g["noise_spectrum"][:ref.shape[0]], ph)
        assert np.array_equal(got, ref)
    # digitised waveforms
    keep = g["wvfm_keep_rows"]
    args = (g["response"][keep], g["op_channel"][keep], g["response_true_id"][keep].astype('i8'),
            g["response_true_photons"][keep], g["trigger_idx"], g["trigger_op_channel_idx"], int(g["digit_samples"]))
    d, dt, dp = O.sim_triggers(*args, np.zeros_like(g["noise_spectrum"]))
    assert np.array_equal(d, g["wvfm_quiet"]) and np.array_equal(dt, g["wvfm_true_id"])
    np.testing.assert_allclose(dp, g["wvfm_true_photons"], rtol=1e-15, atol=0)
    assert (dt >= 0).sum() > 100 and (d != 0).sum() > 100
    R = int(keep.sum())
    Tp = None
    # phases of the two noise calls: (rows, Tp//2+1) with Tp the padded length -- recompute it the way sim_triggers does
    l = consts.light
    pre = int(np.ceil(l.LIGHT_TRIG_WINDOW[0] / l.LIGHT_TICK_SIZE)); post = int(np.ceil(l.LIGHT_TRIG_WINDOW[1] / l.LIGHT_TICK_SIZE))
    tmin, tmax = int(g["trigger_idx"].min()), int(g["trigger_idx"].max())
    n0 = max(pre - tmin, 0)
    Tp = g["response"].shape[1] + n0
    Tp += max(post + tmax + n0 - Tp, 0)
    seeds = g["wvfm_noisy_phase_seeds"]
    n_missing = len(np.setdiff1d(np.unique(g["trigger_op_channel_idx"]), g["op_channel"][keep]))
    ph_sig = H.det_phases((R, Tp // 2 + 1), int(seeds[0]))
    ph_mis = H.det_phases((n_missing, Tp // 2 + 1), int(seeds[1])) if len(seeds) > 1 else None
    d2, _, _ = O.sim_triggers(*args, g["noise_spectrum"], phases_signal=ph_sig, phases_missing=ph_mis)
    assert np.array_equal(d2, g["wvfm_noisy"])
