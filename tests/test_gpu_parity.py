"""
GPU parity: the HIP path (through the C-ABI, via the reference-named Python wrappers) against
  (a) the golden vectors produced by the reference's own source, and
  (b) the CPU oracle on seeded inputs.
Integer / index outputs bit-exact; f64 per-segment outputs to 1e-13; induced current and ADC
values within 1e-5 relative (tolerance of BASELINE.json's north_star) + 1e-7 of the waveform peak.
"""
import os

import numpy as np
import pytest

import helpers as H
from larndsim_amd import (batching, consts, detsim, drifting, fee, lib, light_sim, lightLUT, pixels_from_track,
                          quenching, rng as lrng, synth)
from larndsim_amd.chain import ChargeChain
from larndsim_amd.layout import segments_dtype
from oracle import oracle as O

pytestmark = pytest.mark.gpu
CFGS = ["module0", "2x2_no_modvar", "ndlar"]


class _HipQD:
    quench = staticmethod(lambda r, mode: quenching.quench[1, 256](r, mode))
    drift = staticmethod(lambda r: drifting.drift[1, 256](r))


@pytest.mark.parametrize("cfg", CFGS)
def test_quench_drift_golden(cfg):
    H.load_cfg(cfg)
    g = H.gold(f"qd_{cfg}.npz")
    seg = g["segments_in"]
    for mode, name in ((2, "birks"), (1, "box")):
        r = H.to_ref(seg)
        quenching.quench[4, 256](r, mode)
        assert np.array_equal(r["n_electrons"], g[f"{name}_n_electrons"])
        np.testing.assert_allclose(H.f4(r["n_photons"]), g[f"{name}_n_photons"], rtol=1e-6)
        if name == "birks":
            H.round_f4(r, ["n_photons"])
            drifting.drift[4, 256](r)
            assert np.array_equal(r["pixel_plane"], g["drift_pixel_plane"])
            # exact: the lifetime factor exp() is ocml here and libm in the reference, but an ulp of it moves
            # n_e * exp(..) (~1e4..1e5) by ~1e-11, so the u4 truncation flips for about one segment in 1e11
            assert np.array_equal(r["n_electrons"], g["drift_n_electrons"])
            for f in ("long_diff", "tran_diff", "t", "t_start", "t_end"):
                np.testing.assert_allclose(r[f], g["drift_" + f], rtol=1e-13, atol=0, err_msg=f)


@pytest.mark.parametrize("cfg", CFGS)
def test_quench_drift_152B_schema_vs_oracle(cfg):
    """The edep-sim HDF5 schema (f4 fields, u4 n_electrons): stores narrow exactly like the oracle's."""
    H.load_cfg(cfg)
    seg = synth.make_segments(5000, seed=3, spill=consts.sim.IS_SPILL_SIM)
    batching.swap_coordinates(seg)
    a, b = seg.copy(), seg.copy()
    quenching.quench[20, 256](a, consts.physics.BIRKS)
    drifting.drift[20, 256](a)
    O.quench(b, consts.physics.BIRKS)
    O.drift(b)
    assert np.array_equal(a["pixel_plane"], b["pixel_plane"])
    assert np.array_equal(a["n_electrons"], b["n_electrons"])          # u4 truncation included (see the golden test)
    for f in ("n_photons", "long_diff", "tran_diff", "t", "t_start", "t_end"):
        np.testing.assert_allclose(a[f], b[f], rtol=2e-7, atol=0, err_msg=f)   # f4 fields: <= 1 ulp
    for f in seg.dtype.names:
        if f not in ("n_electrons", "n_photons", "pixel_plane", "long_diff", "tran_diff", "t", "t_start", "t_end"):
            assert np.array_equal(a[f], seg[f]), f"untouched field {f} changed"


def test_reference_unit_tests_restated():
    """tests/testQuenching.py:39-124 and tests/testDrifting.py:31-49 of the reference, on f8 records."""
    H.load_cfg("module0")
    det, phys = consts.detector, consts.physics
    rng = np.random.default_rng(5)
    names = "eventID, dEdx, x_start, dE, t_start, z_end, trackID, x_end, y_end, n_electrons, n_photons, t, dx, " \
            "pdgId, y, x, long_diff, z, z_start, y_start, tran_diff, t_end, pixel_plane".split(", ")
    formats = ["i8"] + ["f8"] * 5 + ["i8"] + ["f8"] * 15 + ["i8"]
    dt = np.dtype(list(zip(names, formats)))
    tracks = np.zeros(100, dtype=dt)
    tracks["dE"] = rng.uniform(0.1, 100, 100)
    tracks["dEdx"] = rng.uniform(1, 100, 100)
    tb = tracks.copy()
    quenching.quench[1, 128](tb, phys.BIRKS)
    recomb = phys.BIRKS_Ab / (1 + phys.BIRKS_kb * tracks["dEdx"] / (det.E_FIELD * det.LAR_DENSITY))
    assert tb["n_electrons"] == pytest.approx(recomb * tracks["dE"] / phys.W_ION)
    tx = tracks.copy()
    quenching.quench[1, 128](tx, phys.BOX)
    csi = phys.BOX_BETA * tracks["dEdx"] / (det.E_FIELD * det.LAR_DENSITY)
    assert tx["n_electrons"] == pytest.approx(np.log(phys.BOX_ALPHA + csi) / csi * tracks["dE"] / phys.W_ION)
    z = np.zeros(1, dtype=dt); z["dE"] = 1
    zb, zx = z.copy(), z.copy()
    quenching.quench[1, 128](zb, phys.BIRKS)
    quenching.quench[1, 128](zx, phys.BOX)
    assert zb["n_electrons"] == pytest.approx(phys.BIRKS_Ab * 1 / phys.W_ION)
    assert zx["n_electrons"][0] == 0
    inf = np.zeros(1, dtype=dt); inf["dE"] = 1e10; inf["dEdx"] = 1e10
    for mode in (phys.BIRKS, phys.BOX):
        t = inf.copy()
        quenching.quench[1, 128](t, mode)
        rec = t["n_electrons"][0] * phys.W_ION / 1e10
        assert 0 < rec < 1e-6
    with pytest.raises(ValueError):
        quenching.quench[1, 128](tracks.copy(), 7)
    # drifting
    dn = "eventID, z_end, trackID, tran_diff, z_start, x_end, y_end, n_electrons, pdgId, x_start, y_start, t_start, dx, " \
         "long_diff, pixel_plane, t_end, dEdx, dE, t, y, x, z, t0_start, t0_end, t0".split(", ")
    df = "i8, f8, i8, f8, f8, f8, f8, i8, i8, f8, f8, f8, f8, f8, i8, f8, f8, f8, f8, f8, f8, f8, f8, f8, f8".split(", ")
    d = np.zeros(1, dtype=np.dtype(list(zip(dn, df))))
    B = det.TPC_BORDERS[0]
    d["z"] = rng.uniform(min(B[2]), max(B[2])); d["x"] = rng.uniform(*B[0]); d["y"] = rng.uniform(*B[1])
    d["n_electrons"] = rng.uniform(1e6, 1e7)
    expect = d["n_electrons"] * np.exp(-np.abs(d["z"] - B[2][0]) / det.V_DRIFT / det.ELECTRON_LIFETIME)
    drifting.drift[1, 128](d)
    assert d["n_electrons"] == pytest.approx(expect)


@pytest.mark.parametrize("cfg", CFGS)
def test_pixels_time_intervals_golden(cfg):
    H.load_cfg(cfg)
    g = H.gold(f"pixels_{cfg}.npz")
    r = H.quench_drift(O, g["segments_in"])      # oracle upstream: isolates the stage under test
    n = r.shape[0]
    mp = np.array([0])
    pixels_from_track.max_pixels[1, 128](r, mp)
    assert mp[0] == int(g["max_pixels"])
    active = np.full((n, mp[0]), -1, dtype=np.int32)
    neigh = np.full(g["neigh"].shape, -1, dtype=np.int32)
    nrad = np.full(g["neigh"].shape, -1, dtype=np.int32)
    nlist = np.zeros(n)
    pixels_from_track.get_pixels[1, 128](r, active, neigh, nrad, nlist, int(g["max_radius"]))
    assert np.array_equal(active, g["active"])
    assert np.array_equal(neigh, g["neigh"])
    assert np.array_equal(nrad, g["nrad"])
    assert np.array_equal(nlist, g["n_pixels_list"])
    starts = np.empty(n); tmax = np.array([0])
    detsim.time_intervals[1, 128](starts, tmax, r)
    assert np.array_equal(starts, g["track_starts"])
    assert tmax[0] == int(g["max_length"])


# The kernels that can carry tracks_current, by option set: the monolithic closed-form kernel (overflow fallback of the others),
# the round-1 split path (closed form per sample + LDS-staged correlation) and the default (quadrature weights + shifted-window
# correlation).  Every per-tick test runs through all of them; ldsim_tracks_current_stats tells which kernels really ran.
CURRENT_PATHS = {"mono": dict(split_kernels=0),
                 "closed": dict(split_kernels=1, weights_mode=0, mac_mode=0),
                 "quad": dict(split_kernels=1, weights_mode=1, mac_mode=1),
                 "gform": dict(split_kernels=1, weights_mode=2, gform_max_support=1e9),     # (forced for tables of any support)
                 # the tables stage's workgroup kernel for every pair (by default it gets what the wave kernel cannot take)
                 "gform_wg": dict(split_kernels=1, weights_mode=2, gform_max_support=1e9, gform_wave_tables=0)}


def _tracks_current_on(path, neigh, r, resp, T, **extra):
    """signals[S][P][T] of the HIP tracks_current on the named kernels + the call's counters."""
    lib.context()
    sig = np.zeros(neigh.shape + (T,), dtype=np.float32)
    try:
        for k, v in {**CURRENT_PATHS[path], **extra}.items():
            lib.set_option(k, v)
        detsim.tracks_current[(1, 1, 1), (1, 1, 64)](sig, np.ascontiguousarray(neigh), r, resp)
        st = detsim.tracks_current_stats()
    finally:
        _reset_current_options()
    assert st.n_pairs == neigh.size
    if path == "mono":
        assert st.n_wbuf == 0 and st.n_fallback == 0
    else:
        assert st.n_wbuf > 0, "the split path's tables / weights stage did not run"
    return sig, st


@pytest.mark.parametrize("cfg,tag", [(c, "") for c in CFGS] + [
    (c, "corners_") for c in ("module0", "ndlar")
    if os.path.exists(os.path.join(os.path.dirname(__file__), "golden", f"sampled_corners_{c}.npz"))])
@pytest.mark.parametrize("prune", [23.0, 0.0])
@pytest.mark.parametrize("path", list(CURRENT_PATHS))
def test_tracks_current_sampled_golden(cfg, tag, prune, path):
    """The HIP tracks_current against the reference's own output at sampled ticks, on each kernel set (module0 / 2x2:
    TIME_SAMPLING = RESPONSE_SAMPLING, ndlar: twice); `corners_` = degenerate geometries (face-hugging, micrometre-short,
    along / perpendicular to the drift axis, heavily ionising)."""
    H.load_cfg(cfg)
    g = H.gold(f"sampled_{tag}{cfg}.npz")
    r = H.quench_drift(O, g["segments_in"])
    neigh = np.ascontiguousarray(g["neigh"])
    T = int(g["max_length"])
    resp = H.response_for(g["response_kind"])
    sig, st = _tracks_current_on(path, neigh, r, resp, T, prune_log=prune)
    # tolerance is relative to the peak of the FULL waveform, not of the sampled ticks
    peak = np.abs(sig).max(axis=-1, keepdims=True)
    got, ref = sig[:, :, g["ticks"]].astype(np.float64), g["signals"].astype(np.float64)
    err = np.abs(got - ref)
    tol = 1e-5 * np.abs(ref) + 1e-7 * peak
    assert (err <= tol).all(), f"max excess {np.max(err - tol)} at {np.unravel_index(np.argmax(err - tol), err.shape)}"
    assert (ref != 0).sum() > (100 if tag else 500)        # the corner sets hold a dozen ticks per pair
    n_live = int((np.abs(sig).max(axis=-1) > 0).sum())
    print(f"{cfg} {tag}{path} prune {prune}: {n_live} live pairs, {st.n_fallback} through the fallback, pool {st.n_wbuf}")
    if path != "mono":
        # the named kernels, not their fallback, carried the pairs (the corner sets hold the geometries that overflow them,
        # and with every weight kept -- prune 0 -- a good share of the pairs exceeds the item capacity)
        assert st.n_fallback <= (0.5 if (tag or prune == 0) else 0.05) * n_live


@pytest.mark.parametrize("path", list(CURRENT_PATHS))
def test_tracks_current_vs_oracle_full_ticks(path):
    """All ticks of a few pairs, incl. long / steep segments, vs the oracle (which is pinned to the reference)."""
    H.load_cfg("module0")
    seg = synth.make_segments(6, seed=11, segs_per_event=6)
    batching.swap_coordinates(seg)
    # one long, steep segment and one crossing several pixels
    seg["z_end"][0] = seg["z_start"][0] + 1.9; seg["x_end"][0] = seg["x_start"][0] + 0.05
    seg["x_end"][1] = seg["x_start"][1] + 1.4; seg["y_end"][1] = seg["y_start"][1] - 0.9
    for ax in "xyz":
        seg[ax] = 0.5 * (seg[ax + "_start"].astype(np.float64) + seg[ax + "_end"])
    r = H.quench_drift(O, seg)
    nmax = O.max_pixels(r)
    P = 3 * nmax + 6
    _, neigh, nrad, _ = O.get_pixels(r, nmax, P, 1)
    _, T = O.time_intervals(r)
    resp = synth.make_response("golden")
    ref = O.tracks_current(r, neigh, T, resp)
    sig, st = _tracks_current_on(path, neigh, r, resp, T)
    H.assert_wave_close(sig, ref, rtol=1e-5, atol_peak=1e-7, what=f"tracks_current[{path}]")
    assert (ref != 0).sum() > 10000
    if path != "mono":        # (the 1.9 cm steep segment's pairs exceed the item capacity: 4 of 54 on the quadrature path)
        assert st.n_fallback <= 0.1 * (np.abs(ref).max(axis=-1) > 0).sum()


@pytest.mark.gpu
@pytest.mark.parametrize("ns", [24, 40, 48])
def test_tracks_current_other_sample_counts_vs_oracle(ns):
    """detector.SAMPLED_POINTS other than the reference's 40: the wave tables kernel reads its sample maps from the 40-sample map
    records pair_setup_kernel writes (gform.h G_MAP_NS) -- fewer samples leave the tail of a record unused, more send every pair to
    the workgroup tables kernel (wave_ok 0), which derives its maps itself.  All ticks against the oracle at the same setting."""
    H.load_cfg("module0")
    consts.detector.SAMPLED_POINTS = ns
    seg = synth.make_segments(8, seed=13, segs_per_event=8)
    batching.swap_coordinates(seg)
    seg["x_end"][1] = seg["x_start"][1] + 1.1; seg["y_end"][1] = seg["y_start"][1] - 0.7
    for ax in "xyz":
        seg[ax] = 0.5 * (seg[ax + "_start"].astype(np.float64) + seg[ax + "_end"])
    r = H.quench_drift(O, seg)
    nmax = O.max_pixels(r)
    P = 3 * nmax + 6
    _, neigh, nrad, _ = O.get_pixels(r, nmax, P, 1)
    _, T = O.time_intervals(r)
    resp = synth.make_response("golden")
    ref = O.tracks_current(r, neigh, T, resp)
    sig, st = _tracks_current_on("gform", neigh, r, resp, T)
    H.assert_wave_close(sig, ref, rtol=1e-5, atol_peak=1e-7, what=f"tracks_current, {ns} samples per axis")
    assert (ref != 0).sum() > 10000 and st.n_fallback <= 0.1 * (np.abs(ref).max(axis=-1) > 0).sum()


@pytest.mark.parametrize("cfg", ["module0", "ndlar"])
@pytest.mark.parametrize("path", ["gform", "quad"])
def test_tracks_current_length_sweep_vs_oracle(cfg, path):
    """Segment length from 0 to 195 Gaussian widths along the segment (r), a few hundred micrometres to a centimetre from
    the anode where sigma_T -> 0: the quadrature's node rule N = ceil(3.4 + 1.38 r) (round 4; it was 4.8 + 1.6 r) on both sides
    of the 64-node LDS copy (r = 44; 37 under the old rule) and of the 256-node cap (r = 183; 157: beyond it the monolithic kernel
    takes the pair), every tick against the oracle's closed form (detsim.py:114-159), sub-threshold waveforms included."""
    H.load_cfg(cfg)
    det = consts.detector
    B = det.TPC_BORDERS[0]
    sgn = np.sign(B[2][1] - B[2][0])
    rs = [0.01, 0.8, 4.0, 12.0, 30.0, 36.0, 38.5, 43.0, 45.0, 70.0, 110.0, 150.0, 156.0, 159.0, 165.0, 181.0, 185.0, 195.0]
    dists = [0.03, 0.12, 0.6]
    seg = synth.make_segments(len(rs) * len(dists), seed=23, segs_per_event=len(rs) * len(dists))
    batching.swap_coordinates(seg)
    x0, y0 = B[0][0] + 0.37 * (B[0][1] - B[0][0]), B[1][0] + 0.41 * (B[1][1] - B[1][0])
    k = 0
    for d in dists:
        sT = np.sqrt(2 * det.TRAN_DIFF * d / det.V_DRIFT)        # drifting.py:47-52 at drift distance d
        for rr in rs:
            ang = 0.7 + 0.37 * k                                 # direction in the pixel plane, a shallow tilt along the drift
            a = np.array([x0 + 0.9 * (k % 5), y0 + 0.7 * (k // 5), B[2][0] + sgn * d])
            # (the shortest ones steeper, so that z_end - z_start survives the f4 fields: the reference divides 0/0 otherwise)
            tilt = 0.0998 if rr >= 1 else 0.6
            L = rr * sT
            for _ in range(4):                                   # rr widths of the cloud at the segment's MIDPOINT (drifting.py:47-52)
                L = rr * np.sqrt(2 * det.TRAN_DIFF * (d + 0.5 * tilt * L) / det.V_DRIFT)
            b = a + L * np.array([np.cos(ang) * np.sqrt(1 - tilt * tilt), np.sin(ang) * np.sqrt(1 - tilt * tilt), sgn * tilt])
            for i, ax in enumerate("xyz"):
                seg[ax + "_start"][k] = a[i]; seg[ax + "_end"][k] = b[i]
                seg[ax][k] = 0.5 * (np.float32(a[i]).astype(np.float64) + np.float32(b[i]))
            seg["dx"][k] = max(L, 1e-4); seg["dEdx"][k] = 2.1; seg["dE"][k] = 2.1 * seg["dx"][k]
            k += 1
    r = H.quench_drift(O, seg)
    assert (r["pixel_plane"] == 0).all()
    nmax = O.max_pixels(r)
    P = 3 * nmax + 6
    _, neigh, nrad, _ = O.get_pixels(r, nmax, P, 1)
    _, T = O.time_intervals(r)
    resp = H.response_for("golden")
    ref = O.tracks_current(r, neigh, T, resp)
    assert np.isfinite(ref).all()
    live = np.abs(ref).max(axis=-1) > 0
    assert live.sum() > 3 * len(rs) * len(dists)
    # Weights below exp(-prune_log) = 1e-10 of the largest weight the segment can put on a sample are dropped.  A pair whose
    # every sample sits in the Gaussian's far tail -- the reference's sample grid spans the chord of the segment's LINE inside
    # the pixel's impact circle (detsim.py:404-411), so for a micrometre-short segment next to the anode the 40 x 40 points
    # are 7 sigma apart and can all miss the cloud -- is then made of dropped weights only and comes out as zeros, where the
    # reference holds a waveform of 1e-7 of what the segment's charge would induce.  With the default pruning the absolute
    # floor of the tolerance is therefore 1e-7 of (segment charge x largest response entry): the current 1e-7 of the charge
    # could induce at most (at most ~1e3 dropped bins of < 1e-10 each).  Every pair above 1e-5 of that scale holds the
    # per-pair bar, and with every weight kept (prune_log 0) all pairs do.
    def close(got, floor_peak, what):
        err = np.abs(got.astype(np.float64) - ref)
        tol = 1e-5 * np.abs(ref) + 1e-7 * floor_peak
        assert (err <= tol).all(), f"{what}: max excess {np.max(err - tol)} at {np.unravel_index(np.argmax(err - tol), err.shape)}"
    pair_peak = np.abs(ref).max(axis=-1, keepdims=True)
    charge_scale = r["n_electrons"].astype(np.float64)[:, None, None] * np.abs(resp).max()
    sig, st = _tracks_current_on(path, neigh, r, resp, T)
    close(sig, charge_scale, f"length sweep {cfg}")
    main = pair_peak >= 1e-5 * charge_scale
    assert main.sum() > 2 * len(rs) * len(dists)
    close(np.where(main, sig, ref), pair_peak, f"length sweep {cfg}, pairs above 1e-5 of the charge scale")
    print(f"length sweep {cfg}: {int(live.sum())} live pairs, {st.n_fallback} beyond the node cap / capacities")
    # both sides of the cap were met: some pairs went to the monolithic kernel, most did not
    assert 0 < st.n_fallback < 0.5 * live.sum()
    # ... with the cap lowered to 40 nodes more pairs take the monolithic kernel: the same waveforms
    sig2, st2 = _tracks_current_on(path, neigh, r, resp, T, quad_max_nodes=40)
    close(sig2, charge_scale, f"length sweep {cfg}, 40-node cap")
    assert st2.n_fallback > st.n_fallback
    # ... and with every weight kept, the per-pair tolerance for every pair down to 1e-12 of the charge scale (below that
    # the pairs that overflow to the monolithic kernel show the cancellation noise of its erf differences: 1e-16 absolute)
    sig3, st3 = _tracks_current_on(path, neigh, r, resp, T, prune_log=0.0)
    deep = pair_peak >= 1e-12 * charge_scale
    assert deep.sum() > main.sum()
    close(np.where(deep, sig3, ref), pair_peak, f"length sweep {cfg}, prune_log 0")


def test_stage_api_chain_golden():
    """Stage-by-stage (materialising) API on the golden chain: every intermediate array vs the reference."""
    H.load_cfg("module0")
    g = H.gold("chain_module0.npz")
    r = H.quench_drift(_HipQD, g["segments_in"])
    neigh, nrad = np.ascontiguousarray(g["neigh"]), np.ascontiguousarray(g["nrad"])
    T = int(g["max_length"])
    resp = H.response_for(g["response_kind"])
    sig = np.zeros(neigh.shape + (T,), dtype=np.float32)
    detsim.tracks_current[(1, 1, 1), (1, 1, 64)](sig, neigh, r, resp)
    H.assert_wave_close(sig, g["signals"], what="signals")
    upix = g["unique_pix"]
    M = consts.sim.MAX_TRACKS_PER_PIXEL
    tpm = np.full((len(upix), M), -1, dtype=np.int64)
    detsim.get_track_pixel_map2[1, 32](tpm, upix, neigh, nrad, int(nrad.max()) + 1)
    assert np.array_equal(tpm, g["track_pixel_map"])
    NT = len(consts.detector.TIME_TICKS)
    ps = np.zeros((len(upix), NT)); pts = np.zeros((len(upix), NT, M)); ovf = np.zeros(len(upix))
    detsim.sum_pixel_signals[1, 1](ps, g["signals"], g["track_starts"], g["pixel_index_map"], tpm, pts, ovf)
    np.testing.assert_allclose(ps, g["pixels_signals"], rtol=1e-12, atol=1e-12 * np.abs(g["pixels_signals"]).max())
    assert np.array_equal(ovf, g["overflow"])
    tt = np.linspace(0, consts.detector.TIME_INTERVAL[1], NT + 1)
    A = consts.sim.MAX_ADC_VALUES
    for name in ("default", "low"):
        adc = np.zeros((len(upix), A)); tk = np.zeros((len(upix), A)); fr = np.zeros((len(upix), A, M))
        thr = np.full(len(upix), float(g[f"threshold_{name}"]))
        fee.get_adc_values[1, 128](ps, pts, tt, adc, tk, 0, None, fr, thr)
        ref = g[f"adc_integral_{name}"]
        assert np.array_equal(adc != 0, ref != 0)
        np.testing.assert_allclose(adc, ref, rtol=1e-9)
        assert np.array_equal(tk, g[f"adc_ticks_{name}"])
        hit = ref != 0
        np.testing.assert_allclose(fr[hit], g[f"adc_fractions_{name}"][hit], rtol=1e-9, atol=1e-12)
        assert np.array_equal(fee.digitize(adc), g[f"adc_digit_{name}"])
    assert (g["adc_integral_low"] != 0).sum() > 0


def test_fused_chain_golden():
    """Device-resident chain (no materialised signals / slabs) against the golden chain, 152-B-like values."""
    H.load_cfg("module0")
    g = H.gold("chain_module0.npz")
    resp = H.response_for(g["response_kind"])
    ch = ChargeChain(resp)
    # the golden run narrowed quench/drift outputs through f4 between stages; do the same here
    r = H.quench_drift(_HipQD, g["segments_in"])
    ch.upload(r, np.zeros(len(r), dtype=np.int32))
    for name in ("default", "low"):
        consts.detector.DISCRIMINATION_THRESHOLD = float(g[f"threshold_{name}"])
        ch.refresh_constants()
        st = ch.run(0, len(r), want_fractions=True)
        out = ch.download()
        assert np.array_equal(out["unique_pix"], g["unique_pix"])
        assert np.array_equal(out["track_pixel_map"], g["track_pixel_map"])
        ref = g[f"adc_integral_{name}"]
        assert np.array_equal(out["adc_list"] != 0, ref != 0)
        np.testing.assert_allclose(out["adc_list"], ref, rtol=1e-5)
        assert np.array_equal(out["adc_ticks_list"], g[f"adc_ticks_{name}"])
        assert np.array_equal(out["adc_digit"], g[f"adc_digit_{name}"])
        hit = ref != 0
        np.testing.assert_allclose(out["current_fractions"][hit], g[f"adc_fractions_{name}"][hit], rtol=1e-5,
                                   atol=1e-9)
        assert st.n_unique == len(g["unique_pix"]) and st.max_length == int(g["max_length"])


def _oracle_chain(seg, response, thr_of_pixel=None, gain_of_pixel=None, rng_states=None):
    """Reference dataflow on the oracle; `thr_of_pixel` / `gain_of_pixel` are dense arrays over pixel ids standing for
    the driver's pixel_thresholds_lut[unique_pix] / pixel_gains_lut[unique_pix] (cli/simulate_pixels.py:1079-1100)."""
    ref = seg.copy()
    O.quench(ref, consts.physics.BIRKS)
    O.drift(ref)
    nmax = O.max_pixels(ref)
    r = int(np.ceil(ref["tran_diff"].max() * 5 / consts.detector.PIXEL_PITCH))
    P = (2 * r + 1) * nmax + (1 + 2 * r) * r * 2
    _, neigh, nrad, _ = O.get_pixels(ref, nmax, P, r)
    upix = O.unique_pixels(neigh)
    starts, T = O.time_intervals(ref)
    sig = O.tracks_current(ref, neigh, T, response)
    pim = O.pixel_index_map(neigh, upix)
    tpm = O.track_pixel_map(upix, neigh, nrad, int(nrad.max()) + 1, consts.sim.MAX_TRACKS_PER_PIXEL)
    ps, pts, ovf = O.sum_pixel_signals(sig, starts, pim, tpm, len(upix))
    tt = np.linspace(0, consts.detector.TIME_INTERVAL[1], ps.shape[1] + 1)
    thr = (np.full(len(upix), consts.detector.DISCRIMINATION_THRESHOLD) if thr_of_pixel is None
           else np.ascontiguousarray(thr_of_pixel[upix]))
    adc, ticks, frac = O.get_adc_values(ps, pts, tt, thr, rng_states=rng_states)
    gain = None if gain_of_pixel is None else gain_of_pixel[upix][:, None] * np.ones((1, adc.shape[1]))
    return dict(unique_pix=upix, tpm=tpm, adc=adc, ticks=ticks, frac=frac, digit=O.digitize(adc, gain), ref=ref)


@pytest.mark.parametrize("cfg,kind", [("module0", "survey"), ("2x2_no_modvar", "dense"), ("ndlar", "golden")])
def test_fused_chain_vs_oracle_two_batches(cfg, kind):
    """Two events in ONE chain call == the oracle run per event (batches never mix)."""
    H.load_cfg(cfg)
    seg = synth.make_segments(16, seed=21, segs_per_event=8, spill=bool(consts.sim.IS_SPILL_SIM))
    if consts.sim.IS_SPILL_SIM:        # the driver removes the spill offset (cli/simulate_pixels.py:574-582)
        loc = seg["event_id"] % consts.sim.MAX_EVENTS_PER_FILE
        for f in ("t0", "t0_start", "t0_end"):
            seg[f] = seg[f] - loc * consts.sim.SPILL_PERIOD
    batching.swap_coordinates(seg)
    bid, order, table = batching.assign_batches(seg)
    seg, bid = seg[order], bid[order]
    resp = H.response_for(kind)
    ch = ChargeChain(resp)
    ch.upload(seg, bid)
    ch.quench_drift()
    ch.run(0, len(seg), want_fractions=True)
    out = ch.download()
    assert len(table) >= 2
    for b in range(len(table)):
        o = _oracle_chain(seg[bid == b], resp)
        m = out["batch"] == b
        assert np.array_equal(out["unique_pix"][m], o["unique_pix"])
        assert np.array_equal(out["track_pixel_map"][m], o["tpm"])
        assert np.array_equal(out["adc_list"][m] != 0, o["adc"] != 0)
        np.testing.assert_allclose(out["adc_list"][m], o["adc"], rtol=1e-5)
        assert np.array_equal(out["adc_ticks_list"][m], o["ticks"])
        assert np.array_equal(out["adc_digit"][m], o["digit"])
        hit = o["adc"] != 0
        np.testing.assert_allclose(out["current_fractions"][m][hit], o["frac"][hit], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("cfg,kind", [("module0", "survey"), ("2x2_no_modvar", "dense")])
def test_fused_chain_pixel_thresholds_and_gains_vs_oracle(cfg, kind):
    """per-pixel discrimination thresholds and gains (the driver's CudaDict lookups with a default,
    cli/simulate_pixels.py:1079-1100) in the fused chain == the oracle fed pixel_thresholds_lut[unique_pix] /
    pixel_gains_lut[unique_pix].  Half of the pixel ids get their own value, the rest the table's default; thresholds
    spread over 0.4-3x the nominal one so that hits appear, vanish and move."""
    H.load_cfg(cfg)
    seg = synth.make_segments(24, seed=23, segs_per_event=12, spill=bool(consts.sim.IS_SPILL_SIM))
    if consts.sim.IS_SPILL_SIM:
        loc = seg["event_id"] % consts.sim.MAX_EVENTS_PER_FILE
        for f in ("t0", "t0_start", "t0_end"):
            seg[f] = seg[f] - loc * consts.sim.SPILL_PERIOD
    batching.swap_coordinates(seg)
    bid, order, table = batching.assign_batches(seg)
    seg, bid = seg[order], bid[order]
    resp = H.response_for(kind)
    det = consts.detector
    n_ids = int(det.N_PIXELS[0] * det.N_PIXELS[1] * det.TPC_BORDERS.shape[0])
    rng = np.random.default_rng(77)
    keys = rng.choice(n_ids, size=n_ids // 2, replace=False).astype(np.int32)
    nominal_gain = det.GAIN * consts.units.mV / consts.units.e
    thr_default, gain_default = 1.3 * det.DISCRIMINATION_THRESHOLD, 0.9 * nominal_gain
    thr_vals = det.DISCRIMINATION_THRESHOLD * rng.uniform(0.4, 3.0, keys.size)
    gain_vals = nominal_gain * rng.uniform(0.5, 1.5, keys.size)
    thr_of_pixel = np.full(n_ids, thr_default); thr_of_pixel[keys] = thr_vals
    gain_of_pixel = np.full(n_ids, gain_default); gain_of_pixel[keys] = gain_vals
    ch = ChargeChain(resp)
    ch.upload(seg, bid)
    ch.quench_drift()
    ch.run(0, len(seg), want_fractions=True)
    plain = ch.download()
    try:
        # keys of another geometry can never be looked up and are skipped
        ch.set_pixel_thresholds(np.r_[keys, np.int32(n_ids + 5)], np.r_[thr_vals, 1e30], thr_default)
        ch.set_pixel_gains(keys, gain_vals, gain_default)
        ch.run(0, len(seg), want_fractions=True)
        out = ch.download()
        # tables built for this geometry must not be applied to another one silently ...
        H.load_cfg("ndlar" if cfg != "ndlar" else "module0")
        with pytest.raises(lib.LdsimError, match="another pixel geometry"):
            ChargeChain(None).run(0, len(seg))
        # ... and that second chain froze other constants in the process-wide context: the first one must refuse to go on
        # instead of computing with them
        H.load_cfg(cfg)
        with pytest.raises(lib.LdsimError, match="create a new ChargeChain"):
            ch.run(0, len(seg))
    finally:
        lib.check(lib.load().ldsim_clear_pixel_tables(lib.context()))
    assert not np.array_equal(out["adc_list"] != 0, plain["adc_list"] != 0)      # the tables did change the hits
    assert len(table) >= 2
    for b in range(len(table)):
        o = _oracle_chain(seg[bid == b], resp, thr_of_pixel, gain_of_pixel)
        m = out["batch"] == b
        assert np.array_equal(out["unique_pix"][m], o["unique_pix"])
        assert np.array_equal(out["adc_list"][m] != 0, o["adc"] != 0)
        np.testing.assert_allclose(out["adc_list"][m], o["adc"], rtol=1e-5)
        assert np.array_equal(out["adc_ticks_list"][m], o["ticks"])
        assert np.array_equal(out["adc_digit"][m], o["digit"])
        hit = o["adc"] != 0
        np.testing.assert_allclose(out["current_fractions"][m][hit], o["frac"][hit], rtol=1e-5, atol=1e-9)
    # and after clearing the tables a new chain is back on the constants
    ch = ChargeChain(resp)
    ch.upload(seg, bid)
    ch.quench_drift()
    ch.run(0, len(seg), want_fractions=True)
    again = ch.download()
    assert np.array_equal(again["unique_pix"], plain["unique_pix"])
    assert np.array_equal(again["adc_digit"], plain["adc_digit"]) and np.array_equal(again["adc_ticks_list"], plain["adc_ticks_list"])


@pytest.mark.parametrize("cfg", ["module0", "2x2_no_modvar"])
def test_light_golden(cfg):
    H.load_cfg(cfg)
    g = H.gold(f"light_{cfg}.npz")
    r = H.quench_drift(O, g["segments_in"])
    n = len(r)
    lut = synth.make_lut((14, 26, 8), 48, int(g["n_prof"]), int(g["lut_seed"]))
    n_op = consts.light.N_OP_CHANNEL
    inc = np.zeros((n, n_op), dtype=[('segment_id', 'u4'), ('n_photons_det', 'f4'), ('t0_det', 'f4')])
    vox = np.zeros((n, 3), dtype='i4')
    lightLUT.calculate_light_incidence[1, 256](r, lut, inc, vox)
    assert np.array_equal(vox, g["voxel"])
    np.testing.assert_allclose(inc['n_photons_det'], g["n_photons_det"], rtol=1e-6)
    np.testing.assert_allclose(inc['t0_det'], g["t0_det"], rtol=1e-6)
    inc['n_photons_det'] = g["n_photons_det"]; inc['t0_det'] = g["t0_det"]
    n_ticks, t_start = light_sim.get_nticks(inc)
    n_ticks = min(n_ticks, int(g["n_ticks"]))
    assert t_start == pytest.approx(float(g["t_start"]))
    opc = g["op_channel"]
    out = np.zeros((len(opc), n_ticks), dtype='f4')
    Mt = g["true_id"].shape[-1]
    tid = np.full((len(opc), n_ticks, Mt), -1, dtype='i8'); tph = np.zeros((len(opc), n_ticks, Mt))
    light_sim.sum_light_signals[1, 64](r, vox, np.arange(n, dtype='i8'), inc, opc, lut, float(g["t_start"]), out, tid,
                                       tph, g["sorted_indices"], int(g["n_prof"]))
    ref = g["light_sample_inc"][:, :n_ticks]
    np.testing.assert_allclose(out, ref, rtol=2e-6, atol=0)
    assert np.array_equal(tid, g["true_id"][:, :n_ticks])
    assert ref.sum() > 0
    # without truth slots the scatter kernel runs (f64 tile in LDS, one f32 rounding): same photons per tick
    out2 = np.zeros_like(out)
    light_sim.sum_light_signals[1, 64](r, vox, np.arange(n, dtype='i8'), inc, opc, lut, float(g["t_start"]), out2,
                                       np.zeros((len(opc), n_ticks, 0), dtype='i8'), np.zeros((len(opc), n_ticks, 0)),
                                       g["sorted_indices"], int(g["n_prof"]))
    assert np.array_equal(out2 != 0, ref != 0)
    np.testing.assert_allclose(out2, ref, rtol=1e-5, atol=0)


def test_chain_properties_full_event():
    """BASELINE-size event (5000 segments): size-independent properties instead of the (slow) oracle."""
    H.load_cfg("module0")
    seg = synth.make_segments(10000, seed=20241016 + 2)
    batching.swap_coordinates(seg)
    bid, order, table = batching.assign_batches(seg)
    seg, bid = seg[order], bid[order]
    resp = synth.make_response("survey")
    ch = ChargeChain(resp)
    ch.upload(seg, bid)
    ch.quench_drift()
    st = ch.run(0, len(seg))
    both = ch.download()
    # the page-locked download buffers (ldsim_host_alloc) hold the same arrays; they are the chain's, reused by the next call
    pin = ch.download(pinned=True)
    for k in both:
        assert np.array_equal(pin[k], both[k]), k
    q = ch.download_segments(seg.copy())
    # (1) sortedness: rows ordered by (batch, pixel id), pixel ids unique inside a batch
    key = both["batch"].astype(np.int64) * (1 << 32) + both["unique_pix"]
    assert (np.diff(key) > 0).all()
    # (2) batches are independent: running the second event alone gives the same rows -- pixels, hit slots, tick stamps
    #     and ADC counts exactly; charges to 1e-12 (weights_kernel's f64 LDS adds from four waves are order-dependent
    #     in the last bit, see test_chain_properties_baseline_sizes)
    n0 = int((bid == 0).sum())
    ch.run(n0, len(seg))
    second = ch.download()
    m = both["batch"] == 1
    assert np.array_equal(second["unique_pix"], both["unique_pix"][m])
    assert np.array_equal(second["adc_ticks_list"], both["adc_ticks_list"][m])
    assert np.array_equal(second["adc_digit"], both["adc_digit"][m])
    np.testing.assert_allclose(second["adc_list"], both["adc_list"][m], rtol=1e-12, atol=0)
    # (3) charge closure: the survey response integrates to 1 per unit charge at (i,j)=(0,0) and falls off,
    #     so collected charge is positive and below the drifted charge
    tot_q = both["adc_list"].sum()
    assert 0 < tot_q < q["n_electrons"].astype(np.float64).sum() * 30
    # (4) idempotence: same call again -> same discrete outputs exactly, charges and fractions to 1e-12
    #     (no global atomics on the data path; the only order-dependent sums are the LDS ones named above)
    ch.run(n0, len(seg))
    again = ch.download()
    for k in second:
        if second[k].dtype.kind == "f" and k not in ("adc_ticks_list", "adc_digit"):
            np.testing.assert_allclose(again[k], second[k], rtol=1e-12, atol=0, err_msg=k)
        else:
            assert np.array_equal(second[k], again[k]), k
    assert st.n_overflow == 0 and st.n_batches == 2


def test_more_than_max_tracks_per_pixel_vs_oracle():
    """70 short segments over the same pixels: get_track_pixel_map2 keeps the first MAX_TRACKS_PER_PIXEL = 50 in
    (ring distance, segment index) order, sum_pixel_signals leaves the others out of the pixel sum and raises the
    overflow flag (detsim.py:510-524, 582-607).  The fused chain (slot order from the sort, compact per-pair waveforms)
    must keep the same 50 and sum the same charge as the oracle's literal dataflow."""
    H.load_cfg("module0")
    det = consts.detector
    rng = np.random.default_rng(12)
    n = 70
    seg = np.zeros(n, dtype=segments_dtype)
    b = np.sort(det.TPC_BORDERS[0], axis=-1)
    # centre of a pixel well inside TPC 0 (TPC frame: x, y in the pixel plane, z the drift axis)
    px, py = 40, 100
    xc = b[0][0] + (px + 0.5) * det.PIXEL_PITCH
    yc = b[1][0] + (py + 0.5) * det.PIXEL_PITCH
    z0 = 0.5 * (b[2][0] + b[2][1])
    half = 0.35 * det.PIXEL_PITCH
    xs = xc + rng.uniform(-half, half, n); xe = xs + rng.uniform(0.02, 0.1, n) * rng.choice([-1, 1], n)
    ys = yc + rng.uniform(-half, half, n); ye = ys + rng.uniform(-0.1, 0.1, n)
    zs = z0 + rng.uniform(-1.0, 1.0, n); ze = zs + rng.uniform(-0.15, 0.15, n)
    seg["x_start"], seg["x_end"], seg["y_start"], seg["y_end"], seg["z_start"], seg["z_end"] = xs, xe, ys, ye, zs, ze
    for a in "xyz":
        seg[a] = 0.5 * (seg[a + "_start"] + seg[a + "_end"])
    seg["dx"] = np.sqrt((xe - xs) ** 2 + (ye - ys) ** 2 + (ze - zs) ** 2)
    seg["dEdx"] = 2.1
    seg["dE"] = seg["dEdx"] * seg["dx"]
    seg["segment_id"] = np.arange(n); seg["event_id"] = 0; seg["pdg_id"] = 13
    bid = np.zeros(n, dtype=np.int32)
    resp = synth.make_response("survey")
    ch = ChargeChain(resp)
    ch.upload(seg, bid)
    ch.quench_drift()
    st = ch.run(0, n, want_fractions=True)
    out = ch.download()
    o = _oracle_chain(seg, resp)
    assert st.n_overflow > 0 and st.max_active >= 1
    full = (o["tpm"] >= 0).sum(axis=1) == consts.sim.MAX_TRACKS_PER_PIXEL
    assert full.any()                                       # some pixel really filled all 50 slots
    assert np.array_equal(out["unique_pix"], o["unique_pix"])
    assert np.array_equal(out["track_pixel_map"], o["tpm"])
    assert np.array_equal(out["adc_list"] != 0, o["adc"] != 0)
    np.testing.assert_allclose(out["adc_list"], o["adc"], rtol=1e-5)
    assert np.array_equal(out["adc_ticks_list"], o["ticks"])
    assert np.array_equal(out["adc_digit"], o["digit"])
    hit = o["adc"] != 0
    np.testing.assert_allclose(out["current_fractions"][hit], o["frac"][hit], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("cfg,seeds", [("module0", (0, 1, 2, 3, 4, 5, 6, 7)), ("ndlar", (202, 203))])
def test_fused_chain_fuzz_slice_vs_oracle(cfg, seeds):
    """A slice of tools/fuzz_chain.py (the full runs: 700 cases over module0 / 2x2 dense / ndlar, 1.7e5 pairs, 3.2e4 hits,
    no mismatch): two events of ten segments per seed, the seed picks the flavour -- plain, short tracks, long segments
    (several slice chunks, overflow fallback), heavily ionising (hits on neighbours, several per pixel), medium, hugging
    the TPC faces (pixel ids off the plane), very short segments, nearly along the drift axis -- and the fused chain must
    match the oracle: pixels, track map, hit slots, ticks and ADC counts exactly, charges and
    fractions to 1e-5."""
    import importlib.util
    import os
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fuzz_chain", os.path.join(repo, "tools", "fuzz_chain.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    H.load_cfg(cfg)
    resp = H.response_for("survey")
    n_hits = 0
    for seed in seeds:
        problems, st, nh = fz.check_case(seed, cfg, resp)
        assert not problems, f"seed {seed}: {problems}"
        n_hits += nh
    assert n_hits > 20


def test_boundary_refuses_misuse_and_stays_usable():
    """Misuse of the C-ABI through the Python layer fails loudly with a message that names the problem (status code +
    ldsim_last_error -> LdsimError), and the context computes correctly afterwards."""
    H.load_cfg("module0")
    seg = synth.make_segments(40, seed=4, segs_per_event=20)
    batching.swap_coordinates(seg)
    bid, order, table = batching.assign_batches(seg)
    seg, bid = seg[order], bid[order]
    resp = synth.make_response("survey")
    ch = ChargeChain(resp)
    ch.upload(seg, bid)
    ch.quench_drift()
    ch.run(0, len(seg))
    good = ch.download()
    assert (good["adc_list"] != 0).sum() > 5
    with pytest.raises(lib.LdsimError, match="outside the resident store"):
        ch.run(0, len(seg) + 1)
    with pytest.raises(lib.LdsimError, match="outside the resident store"):
        ch.run(10, 5)
    with pytest.raises(lib.LdsimError, match="record count differs"):
        ch.download_segments(seg[:7].copy())
    with pytest.raises(lib.LdsimError, match="unknown option"):
        lib.set_option("no_such_option", 1)
    with pytest.raises(ValueError, match="differ in length"):
        ch.set_pixel_thresholds(np.arange(3), np.ones(2), 1.0)
    # FEE noise needs a seeded state table (ldsim_rng_seed): without one the call is refused, not run noise-free
    consts.detector.RESET_NOISE_CHARGE = 900.0
    try:
        noisy = ChargeChain(resp)
        lib.check(lib.load().ldsim_rng_clear(noisy.ctx))
        noisy.upload(seg, bid)
        noisy.quench_drift()
        with pytest.raises(lib.LdsimError, match="ldsim_rng_seed first"):
            noisy.run(0, len(seg))
        noisy.seed_rng(3)
        noisy.run(0, len(seg))
    finally:
        consts.detector.RESET_NOISE_CHARGE = 0
    # decreasing batch ids are refused at upload, before anything of the ctx is touched: the segments of the upload before stay
    # resident WITH their own batch ids (the ids used to be overwritten first), so the chain still runs on them
    fresh = ChargeChain(resp)
    fresh.upload(seg, bid)
    fresh.quench_drift()
    with pytest.raises(lib.LdsimError, match="non-decreasing"):
        fresh.upload(seg[:40], np.r_[np.ones(20), np.zeros(20)].astype(np.int32))
    fresh.n = len(seg)           # (the Python wrapper's own count follows its last call; the ctx still holds all segments)
    fresh.run(0, len(seg))
    kept = fresh.download()
    assert np.array_equal(kept["unique_pix"], good["unique_pix"]) and np.array_equal(kept["adc_digit"], good["adc_digit"])
    fresh.upload(seg, bid)
    fresh.quench_drift()
    fresh.run(0, len(seg))
    again = fresh.download()
    assert np.array_equal(again["unique_pix"], good["unique_pix"]) and np.array_equal(again["adc_digit"], good["adc_digit"])
    assert np.array_equal(again["adc_ticks_list"], good["adc_ticks_list"])


def test_second_thread_entering_a_busy_ctx_is_refused():
    """A ctx is thread-compatible, not thread-safe.  In round 3 two host threads inside one ctx freed the same scratch buffer twice
    under running kernels (GPU memory access fault, gpurun_out/r03_h25.log).  Every C-ABI entry now claims the ctx: a thread that
    enters while another thread's call is in progress gets LDSIM_ESTATE before it touches anything, and the call in progress is
    not disturbed.  The Python layer refuses the set-up itself: a ChargeChain made on a second thread while the first is alive."""
    import ctypes as C
    import threading
    H.load_cfg("module0")
    seg = synth.make_segments(12000, seed=5, segs_per_event=3000)
    batching.swap_coordinates(seg)
    bid, order, table = batching.assign_batches(seg)
    seg, bid = seg[order], bid[order]
    ch = ChargeChain(synth.make_response("survey"))
    ch.upload(seg, bid)
    ch.quench_drift()
    ch.run(0, len(seg), want_fractions=True)
    good = ch.download()
    L = lib.load()
    refused, other, second, done = [], [], [], threading.Event()

    def intruder():                          # hammers the ctx from another thread while the main thread is inside ldsim_charge_chain
        while not done.is_set():
            rc = L.ldsim_synchronize(ch.ctx)
            if rc == -4:
                refused.append(L.ldsim_last_error().decode())       # (thread-local message of this thread's call)
            elif rc != 0:
                other.append(rc)
        try:
            ChargeChain(None)               # (a second chain from this thread while `ch` lives on the main thread)
            second.append("created")
        except lib.LdsimError as e:
            second.append(str(e))

    th = threading.Thread(target=intruder)
    th.start()
    mine = 0
    try:
        for _ in range(6):                   # ctypes releases the GIL inside the call: the intruder runs beside it
            while True:                      # (the guard is symmetric: when the intruder is inside, this thread is the one refused --
                try:                         #  a refused call has touched nothing and is simply made again)
                    ch.run(0, len(seg), want_fractions=True)
                    break
                except lib.LdsimError as e:
                    assert "in use by another thread" in str(e)
                    mine += 1
                    assert mine < 100000
    finally:
        done.set()
        th.join(timeout=60)
    assert not th.is_alive() and not other
    assert refused, "the intruder never met the call in progress"
    assert all("in use by another thread" in m for m in refused)
    assert len(second) == 1 and "still alive" in second[0]
    out = ch.download()
    for k in good:
        assert np.array_equal(good[k], out[k]), f"{k}: the call in progress was disturbed"


def test_chain_run_async_equals_run():
    """ChargeChain.run_async / wait: the launch on a worker thread (the interpreter is free for host work meanwhile; the bundled
    driver builds the previous launch's packets there) gives the same statistics and results as run(); a second launch before
    wait() and a wait() without a launch are refused; a ctx call from the launching thread during the flight is refused by the
    library's guard or runs after it -- never beside it."""
    H.load_cfg("module0")
    seg = synth.make_segments(6000, seed=6, segs_per_event=1500)
    batching.swap_coordinates(seg)
    bid, order, table = batching.assign_batches(seg)
    seg, bid = seg[order], bid[order]
    ch = ChargeChain(synth.make_response("survey"))
    ch.upload(seg, bid)
    ch.quench_drift()
    st0 = ch.run(0, len(seg), want_fractions=True)
    good = ch.download()
    ch.run_async(0, len(seg), want_fractions=True)
    with pytest.raises(lib.LdsimError, match="already in flight"):
        ch.run_async(0, len(seg))
    host_work = sum(i * i for i in range(200000))          # (anything that does not enter the ctx)
    st1 = ch.wait()
    assert host_work > 0 and st1.n_pairs == st0.n_pairs and st1.n_unique == st0.n_unique
    out = ch.download()
    for k in good:
        assert np.array_equal(good[k], out[k]), k
    with pytest.raises(lib.LdsimError, match="no chain launch in flight"):
        ch.wait()


@pytest.mark.parametrize("cfg", ["module0", "ndlar"])
def test_chain_in_pair_ranges_on_two_streams_same_bits(cfg):
    """Option gform_chunks: tables and correlation of the node-separable form in K pair ranges, the tables of range c + 1 on a second
    stream beside the correlation of range c (measured: no faster, the default stays 1).  Same results bit for bit as the plain
    order, for a K that divides the pairs unevenly and with listed pairs (wide wave tables, larger LDS classes) in the launch."""
    H.load_cfg(cfg)
    seg = synth.make_segments(20000, seed=9, segs_per_event=2500)
    batching.swap_coordinates(seg)
    bid, order, table = batching.assign_batches(seg)
    seg, bid = seg[order], bid[order]
    ch = ChargeChain(synth.make_response("survey"))
    ch.upload(seg, bid)
    ch.quench_drift()
    try:
        ch.run(0, len(seg), want_fractions=True)
        good = ch.download()
        for K in (3, 7):
            lib.set_option("gform_chunks", K)
            st = ch.run(0, len(seg), want_fractions=True)
            out = ch.download()
            assert st.n_pairs > 4096 * K
            for k in good:
                assert np.array_equal(good[k], out[k]), (K, k)
    finally:
        lib.set_option("gform_chunks", 1)


def test_chain_with_nothing_to_simulate():
    """Launches that yield no (segment, pixel) pair at all -- every midpoint outside the TPCs (pixel_plane = 0xBEEF,
    drifting.py:34-39), and an empty segment range -- return empty results instead of launching zero-sized grids."""
    H.load_cfg("module0")
    seg = synth.make_segments(64, seed=3, segs_per_event=64)
    batching.swap_coordinates(seg)
    for f in ("x", "x_start", "x_end"):
        seg[f] += 1.0e4                                   # far outside every TPC box
    bid = np.zeros(len(seg), dtype=np.int32)
    ch = ChargeChain(synth.make_response("survey"))
    ch.upload(seg, bid)
    ch.quench_drift()
    q = ch.download_segments(seg.copy())
    assert (q["pixel_plane"] == consts.detector.DEFAULT_PLANE_INDEX).all()
    st = ch.run(0, len(seg), want_fractions=True)
    out = ch.download()
    assert st.n_pairs == 0 and st.n_unique == 0 and len(out["unique_pix"]) == 0 and out["adc_list"].shape == (0, 30)
    assert ch.compact_hits()[1] == 0
    st = ch.run(5, 5)
    assert st.n_segments == 0 and st.n_unique == 0 and len(ch.download()["unique_pix"]) == 0


def _prepared_set(cfg, n, seed_index):
    """SURVEY §8d synthetic set of BASELINE config `seed_index`, in the driver's frame, batch-sorted."""
    H.load_cfg(cfg)
    seg = synth.make_segments(n, seed=20241016 + seed_index, spill=bool(consts.sim.IS_SPILL_SIM))
    if consts.sim.IS_SPILL_SIM:     # the driver subtracts the spill offset again (cli/simulate_pixels.py:574-582)
        loc = seg["event_id"] % consts.sim.MAX_EVENTS_PER_FILE
        for f in ("t0", "t0_start", "t0_end"):
            seg[f] = seg[f] - loc * consts.sim.SPILL_PERIOD
    batching.swap_coordinates(seg)
    bid, order, table = batching.assign_batches(seg)
    return seg[order], bid[order]


def _per_batch(rows, keep=None):
    """{batch id: that batch's (pixel, charge, tick, ADC) rows} -- a batch is the independent unit of the path."""
    out = {}
    b = rows["batch"]
    edges = np.flatnonzero(np.r_[True, b[1:] != b[:-1], True])
    for lo, hi in zip(edges[:-1], edges[1:]):
        if keep is None or int(b[lo]) in keep:
            out[int(b[lo])] = {k: rows[k][lo:hi].copy() for k in ("unique_pix", "adc_list", "adc_ticks_list", "adc_digit")}
    return out


@pytest.mark.parametrize("cfg,seed_index,n", [("module0", 2, 100_000), ("2x2_no_modvar", 3, 1_000_000),
                                              ("ndlar", 5, 1_000_000)])
def test_chain_properties_baseline_sizes(cfg, seed_index, n):
    """BASELINE.json configs 2, 3 and 5 at their full segment counts (the oracle would need days): properties that
    do not depend on size.  A batch (event x TPC group) is the independent unit of the path, so its rows must not
    depend on what else shares the launch: a 50 k-segment chunking, a 20 k-segment chunking and single-batch launches
    must give the same pixels, hit slots, tick stamps and ADC counts exactly, and the same charges to 1e-12.
    (Not bit for bit: weights_kernel adds into its LDS bins with f64 ds_add from four waves, so the summation order
    depends on timing; measured on MI355X, one charge in ~2e5 segments moves, by 2e-15.  The reference's own pixel sum
    uses f64 atomics, detsim.py:510-524, and is no more reproducible than that.)"""
    seg, bid = _prepared_set(cfg, n, seed_index)
    assert (bid >= 0).all() and (np.diff(bid) >= 0).all()
    ch = ChargeChain(H.response_for("survey"))
    ch.upload(seg, bid)
    ch.quench_drift()
    q = ch.download_segments(seg.copy())
    assert (q["n_electrons"] > 0).mean() > 0.99 and np.isfinite(q["tran_diff"]).all()
    # launches of the second pass: another chunking of the first and last ~100 k segments, three batches alone
    second = [r for r in batching.chunk_ranges(bid, 20_000) if r[1] <= 100_000 or r[0] >= n - 100_000]
    ub = np.unique(bid)
    for bsel in (ub[0], ub[len(ub) // 2], ub[-1]):
        w = np.flatnonzero(bid == bsel)
        second.append((int(w[0]), int(w[-1]) + 1))
    revisit = set()
    for b0, e0 in second:
        revisit |= set(np.unique(bid[b0:e0]).tolist())
    kept, n_seen, n_rows, n_hits, tot_q = {}, 0, 0, 0, 0.0
    stats = dict(S=0, pairs=0, ovf=0, amb=0, fallback=0, batches=0)
    for b0, e0 in batching.chunk_ranges(bid, 50_000):
        st = ch.run(b0, e0)
        rows = ch.download()
        # (1) sortedness / uniqueness: rows ascend in (batch, pixel id)
        key = rows["batch"].astype(np.int64) * (1 << 32) + rows["unique_pix"]
        assert (np.diff(key) > 0).all()
        assert set(np.unique(rows["batch"])) <= set(np.unique(bid[b0:e0]))
        # (2) every written slot is a valid hit: ADC counts are integers in [0, 255], tick stamps ascend per pixel,
        #     a slot is written in all three arrays or in none, and the compact hit list has exactly those slots
        hit = rows["adc_list"] != 0
        assert np.array_equal(hit, rows["adc_ticks_list"] != 0)
        d = rows["adc_digit"][hit]
        assert ((d >= 0) & (d <= 255) & (d == np.round(d))).all()
        assert (hit[:, :-1] | ~hit[:, 1:]).all()            # slots fill from the front
        assert (np.diff(rows["adc_ticks_list"], axis=1)[hit[:, 1:]] > 0).all()
        assert ch.compact_hits()[1] == int(hit.sum())
        assert len(rows["unique_pix"]) == st.n_unique and st.n_segments == e0 - b0
        kept.update(_per_batch(rows, revisit))
        n_seen += len(np.unique(rows["batch"]))
        n_rows += len(key); n_hits += int(hit.sum()); tot_q += float(rows["adc_list"].sum())
        stats["S"] += st.n_segments; stats["pairs"] += st.n_pairs; stats["ovf"] += st.n_overflow
        stats["amb"] += st.n_ambiguous; stats["fallback"] += st.n_fallback; stats["batches"] += st.n_batches
    assert stats["S"] == n and n_rows > n // 2 and n_hits > n // 10
    assert n_seen <= int(bid.max()) + 1 and stats["batches"] == int(bid.max()) + 1
    # pixels hit by more than MAX_TRACKS_PER_PIXEL = 50 segments drop the excess and raise the reference's overflow
    # flag (detsim.py:510-524); the dense synthetic events produce a few, they must stay rare
    assert stats["ovf"] < 1e-3 * n_rows
    print(f"{cfg}: pairs {stats['pairs']} fallback {stats['fallback']} overflow pixels {stats['ovf']} "
          f"ambiguous {stats['amb']} rows {n_rows} hits {n_hits}")
    assert stats["fallback"] < 1e-2 * stats["pairs"]          # the split path carries the load
    assert stats["amb"] < 1e-3 * n                            # shifts near a rounding boundary stay rare
    assert tot_q > 0
    # (3) the second pass: same batches inside other launches
    checked = 0
    for b0, e0 in second:
        ch.run(b0, e0)
        for b, v in _per_batch(ch.download()).items():
            a = kept[b]
            for k in ("unique_pix", "adc_ticks_list", "adc_digit"):
                assert np.array_equal(a[k], v[k]), f"batch {b}: {k} depends on its launch ({b0}, {e0})"
            assert np.array_equal(a["adc_list"] != 0, v["adc_list"] != 0)
            np.testing.assert_allclose(v["adc_list"], a["adc_list"], rtol=1e-12, atol=0,
                                       err_msg=f"batch {b}: charges depend on its launch ({b0}, {e0})")
            checked += 1
    assert checked >= 6


@pytest.mark.parametrize("cfg", ["module0", "2x2_no_modvar"])
def test_light_response_golden(cfg):
    """light_sim.calc_scintillation_effect / calc_light_detector_response through the C-ABI against the
    reference's own output: the weight tables are built with the reference's expressions on the host and every term is
    added in tick order with an f4 store, so waveforms, truth ids and truth photons are required to be bit-identical --
    RLC SiPM model (module0 case) and measured impulse with interpolation (2x2 case), with and without truth slots."""
    g = H.load_light_response_case(cfg)
    inc, tid, tph = g["light_sample_inc"], g["true_id"].astype(np.int64), g["true_photons"]
    D, T = inc.shape
    M = tid.shape[-1]
    grid = ((D, -(-T // 64)), (1, 64))
    scint = np.zeros((D, T), dtype='f4')
    s_id = np.full((D, T, M), -1, dtype='i8'); s_ph = np.zeros((D, T, M))
    light_sim.calc_scintillation_effect[grid[0], grid[1]](inc, tid, tph, scint, s_id, s_ph)
    assert np.array_equal(scint, g["scint"])
    assert np.array_equal(s_id, g["scint_true_id"]) and np.array_equal(s_ph, g["scint_true_photons"])
    resp = np.zeros((D, T), dtype='f4')
    r_id = np.full((D, T, M), -1, dtype='i8'); r_ph = np.zeros((D, T, M))
    light_sim.calc_light_detector_response[grid[0], grid[1]](g["disc"], s_id, s_ph, resp, r_id, r_ph)
    assert np.array_equal(resp, g["response"])
    assert np.array_equal(r_id, g["response_true_id"]) and np.array_equal(r_ph, g["response_true_photons"])
    # no truth slots: same waveforms; and the kernels add to what the caller put in the output (the driver passes zeros)
    none_i, none_p = np.zeros((D, T, 0), dtype='i8'), np.zeros((D, T, 0))
    s2 = np.ones((D, T), dtype='f4')
    light_sim.calc_scintillation_effect[grid[0], grid[1]](inc, none_i, none_p, s2, none_i, none_p)
    assert not np.array_equal(s2, g["scint"]) and (s2[-1] == 1).all()            # the empty channel keeps the caller's 1
    s3 = np.zeros((D, T), dtype='f4')
    light_sim.calc_scintillation_effect[grid[0], grid[1]](inc, none_i, none_p, s3, none_i, none_p)
    r3 = np.zeros((D, T), dtype='f4')
    light_sim.calc_light_detector_response[grid[0], grid[1]](g["disc"], none_i, none_p, r3, none_i, none_p)
    assert np.array_equal(s3, g["scint"]) and np.array_equal(r3, g["response"])


def test_light_response_full_window_vs_oracle():
    """The configurations' real convolution windows (module0: 9000 ticks over an 11000-tick waveform) on a few channels
    against the oracle: covers the multi-chunk walk of the kernel and the lower bound max(itick - conv_ticks, 0)."""
    H.load_cfg("module0")
    light = consts.light
    assert light.SIPM_RESPONSE_MODEL == 1 and light.IMPULSE_MODEL.shape[0] > 2   # the reference's measured impulse
    rng = np.random.default_rng(8)
    D, T = 3, 11000
    inc = np.zeros((D, T), dtype='f4')
    hit = rng.random((D, T)) < 0.002
    inc[hit] = rng.uniform(1.0, 300.0, hit.sum()).astype('f4')
    inc[0, :40] = 50.0
    none_i, none_p = np.zeros((D, T, 0), dtype='i8'), np.zeros((D, T, 0))
    grid = ((D, -(-T // 64)), (1, 64))
    scint = np.zeros((D, T), dtype='f4')
    light_sim.calc_scintillation_effect[grid[0], grid[1]](inc, none_i, none_p, scint, none_i, none_p)
    ref_s = O.scintillation_effect(inc)[0]
    assert np.array_equal(scint, ref_s) and scint[0, 9500] != 0
    resp = np.zeros((D, T), dtype='f4')
    light_sim.calc_light_detector_response[grid[0], grid[1]](scint, none_i, none_p, resp, none_i, none_p)
    assert np.array_equal(resp, O.light_detector_response(scint, light.LIGHT_GAIN, light.IMPULSE_MODEL)[0])
    assert np.abs(resp).max() > 0


def _truth_rows(rng, D, T, M, density, kind):
    """[D][T][M] truth rows for the light response tests: `kind` 0 -- the front filled with distinct ids (what the stages produce),
    1 -- some rows with a repeated id, 2 -- some rows with a hole (-1) in front of filled slots as well, 3 -- ids that do not fit 32
    bits (2^40 + k, -7) in the later half of the ticks (the scintillation stage keeps its rows' ids as 32-bit words in LDS and has
    to fall back to the rows in memory when it meets one)."""
    ids = np.full((D, T, M), -1, dtype='i8')
    ph = np.zeros((D, T, M))
    for d in range(D):
        for t in np.flatnonzero(rng.random(T) < density):
            n = int(rng.integers(1, M + 1))
            ids[d, t, :n] = rng.choice(3 * M, size=n, replace=False) + 10
            ph[d, t, :n] = rng.uniform(0.05, 40.0, n) * rng.choice([1.0, 1.0, 1.0, 1e-3], n)
            r = rng.random()
            if kind in (1, 2) and n >= 3 and r < 0.3:
                ids[d, t, n - 1] = ids[d, t, 0]
            if kind == 2 and n >= 3 and 0.3 <= r < 0.6:
                ids[d, t, int(rng.integers(0, n - 1))] = -1
            if kind == 3 and t > T // 2 and r < 0.5:
                ids[d, t, 0] = (1 << 40) + ids[d, t, 0] if r < 0.3 else -7
    return ids, ph


@pytest.mark.parametrize("M,T,kind,win", [(3, 200, 0, 150.3), (50, 330, 0, 150.3), (50, 200, 2, 150.3), (64, 130, 1, 150.3),
                                          (7, 330, 2, 150.3), (20, 330, 3, 150.3), (1, 70, 0, 2.5), (5, 40, 2, 0.0)])
def test_light_truth_rows_in_lds_vs_oracle(M, T, kind, win):
    """The truth slots of both response stages by `light_truth_lds_kernel` (a wave's 64 output rows in LDS, the default) against the
    oracle's literal walk, bit for bit, and against `light_conv_kernel`'s rows in memory (option light_truth_lds 0) where that path
    is literal: a window shorter than the waveform (the bound max(itick - conv_ticks, 0)), a waveform that is no multiple of 64 ticks,
    zero samples (the scintillation stage skips their truth, :166), negative photons in the SiPM stage's magnitude test, rows that
    fill up and drop ids, and -- `kind` 1, 2 -- input rows with repeated ids / holes and output rows that arrive partly filled."""
    import ctypes as C
    H.load_cfg("module0")
    light, sim = consts.light, consts.sim
    keep = (tuple(light.LIGHT_WINDOW), sim.MC_TRUTH_THRESHOLD)
    try:
        light.LIGHT_WINDOW = (0.0, win * light.LIGHT_TICK_SIZE)          # (conv_ticks = ceil(win): 151, 3 or 0)
        sim.MC_TRUTH_THRESHOLD = 0.02
        rng = np.random.default_rng(100 * M + kind)
        D = 3
        inc = np.zeros((D, T), dtype='f4')
        hit = rng.random((D, T)) < 0.5
        inc[hit] = rng.uniform(1.0, 300.0, hit.sum()).astype('f4')
        tid, tph = _truth_rows(rng, D, T, M, 0.12, kind)
        grid = ((D, -(-T // 64)), (1, 64))

        def start(filled):
            o_id, o_ph = np.full((D, T, M), -1, dtype='i8'), np.zeros((D, T, M))
            if filled:
                o_id, o_ph = _truth_rows(rng, D, T, M, 0.3, kind)
            return np.zeros((D, T), dtype='f4'), o_id, o_ph

        for filled in ([False, True] if kind else [False]):
            s0 = start(filled)
            # oracle: the C restatement on copies of the same in / out arrays
            ref = [a.copy() for a in s0]
            O.lib().o_scintillation_effect(O._p(inc), O._p(tid), O._p(tph), C.c_int32(D), C.c_int32(T), C.c_int32(M), O._p(ref[0]),
                                           O._p(ref[1]), O._p(ref[2]), C.byref(O._consts()))
            got = {}
            for mode in (1, 0):
                lib.set_option("light_truth_lds", mode)
                g = [a.copy() for a in s0]
                light_sim.calc_scintillation_effect[grid[0], grid[1]](inc, tid, tph, g[0], g[1], g[2])
                got[mode] = g
            lib.set_option("light_truth_lds", 1)
            for mode in (1, 0):
                for k, name in enumerate(("scint", "ids", "photons")):
                    assert np.array_equal(got[mode][k], ref[k]), f"scintillation stage, light_truth_lds {mode}, {name} (filled {filled})"
            assert (ref[1][:, :, -1] != -1).any() or M > 10 or win < 3          # (small M: rows fill up and drop ids)
            assert (ref[1] != s0[1]).any()
            # SiPM stage on the scintillation stage's output (negative photons added: the magnitude test), literal rows in `tid`
            s_in, s_id, s_ph = ref[0], ref[1].copy(), ref[2].copy()
            neg = rng.random(s_ph.shape) < 0.2
            s_ph[neg] = -s_ph[neg]
            if kind:
                s_id[:, ::7] = tid[:, ::7]
            r0 = start(filled)
            gain = np.ascontiguousarray(light.LIGHT_GAIN, dtype=np.float64)
            imp = np.ascontiguousarray(light.IMPULSE_MODEL, dtype=np.float64)
            rref = [a.copy() for a in r0]
            O.lib().o_light_detector_response(O._p(s_in), O._p(s_id), O._p(s_ph), C.c_int32(D), C.c_int32(T), C.c_int32(M), O._p(gain),
                                              O._p(imp), C.c_int32(imp.shape[0]), O._p(rref[0]), O._p(rref[1]), O._p(rref[2]),
                                              C.byref(O._consts()))
            for mode in ((1, 0) if kind in (0, 3) else (1,)):      # (rows in memory: literal for distinct ids in front of the first -1 only)
                lib.set_option("light_truth_lds", mode)
                g = [a.copy() for a in r0]
                light_sim.calc_light_detector_response[grid[0], grid[1]](s_in, s_id, s_ph, g[0], g[1], g[2])
                lib.set_option("light_truth_lds", 1)
                for k, name in enumerate(("response", "ids", "photons")):
                    assert np.array_equal(g[k], rref[k]), f"SiPM stage, light_truth_lds {mode}, {name} (filled {filled})"
            assert (rref[2] != r0[2]).any() or win < 3
    finally:
        lib.set_option("light_truth_lds", 1)
        light.LIGHT_WINDOW, sim.MC_TRUTH_THRESHOLD = keep


def test_light_properties_baseline_event():
    """BASELINE.json config 5's light leg (ndlar, synthetic light set-up of SURVEY §8d: 48 channels per TPC) on one full
    event: channel masking, voxel bounds, and exact linearity of the photon sum -- doubling every n_photons_det doubles
    every tick bit for bit (power-of-two scaling commutes with f32/f64 rounding)."""
    seg, bid = _prepared_set("ndlar", 5000, 5)
    n_op = synth.set_synthetic_light(48)
    lut = synth.make_lut((14, 26, 8), 48, 100, synth.SEED_BASE + 5)
    r = H.quench_drift(_HipQD, seg)
    n = len(r)
    inc = np.zeros((n, n_op), dtype=[('segment_id', 'u4'), ('n_photons_det', 'f4'), ('t0_det', 'f4')])
    vox = np.full((n, 3), -1, dtype='i4')
    lightLUT.calculate_light_incidence[1, 256](r, lut, inc, vox)
    in_tpc = r["pixel_plane"] != consts.detector.DEFAULT_PLANE_INDEX
    assert in_tpc.sum() > 0.9 * n
    assert ((vox[in_tpc] >= 0) & (vox[in_tpc] < np.array([14, 26, 8]))).all()
    own = consts.light.OP_CHANNEL_TO_TPC[None, :] == r["pixel_plane"][:, None]
    assert (inc['n_photons_det'][~own] == 0).all()                      # a segment only lights its own TPC's channels
    lit = own & in_tpc[:, None] & (r["n_photons"] > 0)[:, None]
    assert (inc['n_photons_det'][lit] > 0).all()
    assert (inc['n_photons_det'] <= r["n_photons"][:, None] * 1e-2 * (1 + 1e-6)).all()   # vis <= 1e-2, efficiency 1
    # photon sum on the channels of the busiest TPC
    tpc = int(np.bincount(r["pixel_plane"][in_tpc]).argmax())
    opc = consts.light.TPC_TO_OP_CHANNEL[tpc].astype('i4')
    n_ticks, t_start = light_sim.get_nticks(inc)
    n_ticks = min(n_ticks, 50_000)
    order = np.argsort(r["t0"], kind="stable").astype('i4')
    srt = np.tile(order, (len(opc), 1))
    no_id, no_ph = np.zeros((len(opc), n_ticks, 0), dtype='i8'), np.zeros((len(opc), n_ticks, 0))

    def photon_sum(scale):
        scaled = inc.copy()
        scaled['n_photons_det'] = inc['n_photons_det'] * np.float32(scale)
        out = np.zeros((len(opc), n_ticks), dtype='f4')
        light_sim.sum_light_signals[1, 64](r, vox, np.arange(n, dtype='i8'), scaled, opc, lut, t_start, out, no_id,
                                           no_ph, srt, 100)
        return out

    one, two = photon_sum(1.0), photon_sum(2.0)
    assert one.sum() > 0 and (one >= 0).all()
    assert np.array_equal(two, np.float32(2) * one)
    # a channel of another TPC sees nothing from this TPC's segments, and nothing arrives before the first deposit
    seen = inc['n_photons_det'][:, opc].sum(axis=0) > 0
    assert np.array_equal(one.sum(axis=1) > 0, seen)


@pytest.mark.parametrize("cfg", ["module0", "ndlar"])
def test_light_incidence_four_channel_kernel_is_bitwise_the_scalar_one(cfg):
    """light_incidence4_kernel (four channels per lane, float4 stores, channel tables in registers) against the
    one-channel-per-lane kernel and the oracle: n_photons_det, t0_det (threshold-trigger configs) and voxels identical,
    on the resident path and through the stage call; rows of segments outside the TPCs read zero."""
    seg, bid = _prepared_set(cfg, 3000, 7)
    n_op = synth.set_synthetic_light(48)
    lut = synth.make_lut((14, 26, 8), 48, 100, synth.SEED_BASE + 7)
    r = H.quench_drift(_HipQD, seg)
    r["pixel_plane"][::97] = consts.detector.DEFAULT_PLANE_INDEX          # some segments outside every TPC
    n = len(r)
    lib.context()
    got = {}
    try:
        for scalar in (1, 0):
            lib.set_option("light_incidence_scalar", scalar)
            inc = np.zeros((n, n_op), dtype=[('segment_id', 'u4'), ('n_photons_det', 'f4'), ('t0_det', 'f4')])
            vox = np.full((n, 3), -1, dtype='i4')
            lightLUT.calculate_light_incidence[1, 256](r, lut, inc, vox)
            ch = ChargeChain()
            ch.upload(r, np.zeros(n, dtype=np.int32))
            ch.light_incidence(lut)
            rinc, rvox = ch.download_light_incidence()
            got[scalar] = (inc, vox, rinc, rvox)
    finally:
        lib.set_option("light_incidence_scalar", 0)
    for a, b in zip(got[1], got[0]):
        assert a.tobytes() == b.tobytes()
    inc, vox, rinc, rvox = got[0]
    out = r["pixel_plane"] == consts.detector.DEFAULT_PLANE_INDEX
    assert (inc['n_photons_det'][out] == 0).all() and (rinc['n_photons_det'][out] == 0).all()
    assert np.array_equal(inc['n_photons_det'][~out], rinc['n_photons_det'][~out])
    onph, ot0, ovox = O.light_incidence(r, lut)
    assert np.array_equal(vox[~out], ovox[~out])
    assert np.array_equal(inc['n_photons_det'][~out], onph[~out])
    if consts.light.LIGHT_TRIG_MODE == 0:
        assert np.array_equal(inc['t0_det'][~out], ot0[~out])
    assert (inc['n_photons_det'][~out] > 0).sum() > 1000
    # a per-module slice of the channels (n_out < N_OP_CHANNEL: the tables' slice then follows the segment's TPC,
    # lightLUT.py:93-97) takes the kernel's other instantiation
    n_mod = 2 * n_op // consts.detector.TPC_BORDERS.shape[0]
    if n_mod < n_op:
        sl = {}
        try:
            for scalar in (1, 0):
                lib.set_option("light_incidence_scalar", scalar)
                inc2 = np.zeros((n, n_mod), dtype=inc.dtype)
                vox2 = np.full((n, 3), -1, dtype='i4')
                lightLUT.calculate_light_incidence[1, 256](r, lut, inc2, vox2)
                sl[scalar] = inc2
        finally:
            lib.set_option("light_incidence_scalar", 0)
        assert sl[1].tobytes() == sl[0].tobytes()
        onph2, _, _ = O.light_incidence(r, lut, n_out=n_mod)
        assert np.array_equal(sl[0]['n_photons_det'][~out], onph2[~out]) and (onph2 > 0).sum() > 1000


@pytest.mark.parametrize("cfg", ["module0", "2x2_no_modvar"])
def test_resident_light_leg_golden(cfg):
    """Device-resident light leg (ldsim_dev_light_incidence / ldsim_dev_sum_light: nothing leaves HBM between the stages)
    against the reference's goldens and against the host-buffer stage calls: voxels and truth ids exact, photons to f4."""
    H.load_cfg(cfg)
    g = H.gold(f"light_{cfg}.npz")
    r = H.quench_drift(O, g["segments_in"])
    n = len(r)
    lut = synth.make_lut((14, 26, 8), 48, int(g["n_prof"]), int(g["lut_seed"]))
    ch = ChargeChain()
    ch.upload(r, np.zeros(n, dtype=np.int32))
    ch.light_incidence(lut)
    inc, vox = ch.download_light_incidence()
    assert np.array_equal(vox, g["voxel"])
    np.testing.assert_allclose(inc['n_photons_det'], g["n_photons_det"], rtol=1e-6)
    if consts.light.LIGHT_TRIG_MODE == 0:
        np.testing.assert_allclose(inc['t0_det'], g["t0_det"], rtol=1e-6)
    opc = g["op_channel"]
    Mt = g["true_id"].shape[-1]
    n_ticks, t_start = ch.sum_light(0, n, opc, np.arange(n, dtype='i8'), max_truth=Mt, max_ticks=int(g["n_ticks"]))
    assert n_ticks == int(g["n_ticks"]) and t_start == pytest.approx(float(g["t_start"]))
    out, tid, tph = ch.download_light()
    ref = g["light_sample_inc"][:, :n_ticks]
    assert ref.sum() > 0
    np.testing.assert_allclose(out, ref, rtol=3e-6, atol=0)
    assert np.array_equal(tid, g["true_id"][:, :n_ticks])
    np.testing.assert_allclose(tph, g["true_photons"][:, :n_ticks], rtol=1e-5)
    # the host-buffer stage calls on the same records (reference call sites) give the same arrays bit for bit: the same
    # kernels run, the stage call with the driver's sorted_indices, the resident one with its own descending order
    inc2 = np.zeros_like(inc); vox2 = np.zeros_like(vox)
    lightLUT.calculate_light_incidence[1, 256](r, lut, inc2, vox2)
    assert np.array_equal(inc2['n_photons_det'], inc['n_photons_det']) and np.array_equal(vox2, vox)
    out2 = np.zeros_like(out); tid2 = np.full_like(tid, -1); tph2 = np.zeros_like(tph)
    light_sim.sum_light_signals[1, 64](r, vox2, np.arange(n, dtype='i8'), inc2, opc, lut, float(t_start), out2, tid2, tph2,
                                       g["sorted_indices"], int(g["n_prof"]))
    assert np.array_equal(out2, out) and np.array_equal(tid2, tid) and np.array_equal(tph2, tph)
    # without truth slots the scatter kernel runs (f64 tile, one f4 rounding)
    ch.upload(r, np.zeros(n, dtype=np.int32))       # the stage calls above took the segment store over
    ch.light_incidence(lut)
    ch.sum_light(0, n, opc, max_truth=0, max_ticks=int(g["n_ticks"]))
    out3, _, _ = ch.download_light()
    assert np.array_equal(out3 != 0, ref != 0)
    np.testing.assert_allclose(out3, ref, rtol=1e-5, atol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("smear", [True, False])
@pytest.mark.parametrize("n_tracks,with_none", [(3, False), (7, True), (40, True)])
def test_resident_photon_sum_slot_walk_equals_the_record_walk(n_tracks, with_none, smear):
    """The resident photon sum gives out truth slots 64 records at a time (light_replay_wave_kernel: slot lookup by all lanes, new
    tracks in order of first appearance, two sequential chains); the host-array stage call accumulates into the caller's arrays
    and walks record by record as light_sim.py:101-127 is written.  Same arrays bit for bit when many segments share a track
    (slots re-used), when there are more tracks than slots (records dropped), and with track id -1 among them (a slot holding -1
    looks empty and is taken over by the next track, its photons staying).  The two sums also build their records differently: the
    resident one from the compacted pairs with photons, a wave per pair (light_emit_wave_kernel), the stage call from the caller's
    sorted_indices, a thread per pair after a count pass -- with and without LUT smearing."""
    cfg = "2x2_no_modvar" if os.path.exists(os.path.join(os.path.dirname(__file__), "golden", "light_2x2_no_modvar.npz")) else "module0"
    H.load_cfg(cfg)
    g = H.gold(f"light_{cfg}.npz")
    r = np.concatenate([H.quench_drift(O, g["segments_in"])] * 6)      # (six segments on every spot: each cell meets several tracks)
    n = len(r)
    lut = synth.make_lut((14, 26, 8), 48, int(g["n_prof"]), int(g["lut_seed"]))
    opc = g["op_channel"]
    nt = int(g["n_ticks"])
    rng = np.random.default_rng(100 + n_tracks)
    ids = rng.integers(0, n_tracks, n).astype('i8') * 11 + 5
    if with_none:
        ids[rng.random(n) < 0.15] = -1
    Mt = 2
    consts.sim.MC_TRUTH_THRESHOLD = 1e-9          # (the golden set is small: every deposit counts for the slots)
    consts.light.ENABLE_LUT_SMEARING = smear       # (with: a record per profile bin; without: one per pair, at the LUT's mean arrival time)
    ch = ChargeChain()
    ch.upload(r, np.zeros(n, dtype=np.int32))
    ch.light_incidence(lut)
    inc, vox = ch.download_light_incidence()
    n_ticks, t_start = ch.sum_light(0, n, opc, ids, max_truth=Mt, max_ticks=nt)
    out, tid, tph = ch.download_light()
    filled = (tid >= 0).sum(axis=-1)
    assert out.sum() > 0 and filled.max() == Mt, np.bincount(filled.ravel())          # some cell's slots are all taken
    # the resident sum's visiting order: descending photons per detector, ties by descending index
    order = np.ascontiguousarray(np.stack([np.argsort(inc['n_photons_det'][:, c], kind="stable")[::-1] for c in opc]), dtype=np.int32)
    out2 = np.zeros_like(out); tid2 = np.full_like(tid, -1); tph2 = np.zeros_like(tph)
    light_sim.sum_light_signals[1, 64](r, vox, ids, inc, opc, lut, float(t_start), out2, tid2, tph2, order, int(g["n_prof"]))
    assert np.array_equal(out2, out) and np.array_equal(tid2, tid) and np.array_equal(tph2, tph)


@pytest.mark.gpu
def test_resident_photon_sums_in_a_row_start_from_clean_arrays():
    """ldsim_dev_sum_light with truth slots does not clear its arrays when the same buffers served the sum before: it resets the
    cells that sum wrote.  Sums of different batches, tick counts and slot counts in a row on one context must equal the same sums
    each taken right after a sum without truth slots (which makes the next one clear everything): bit for bit."""
    cfg = "2x2_no_modvar" if os.path.exists(os.path.join(os.path.dirname(__file__), "golden", "light_2x2_no_modvar.npz")) else "module0"
    H.load_cfg(cfg)
    g = H.gold(f"light_{cfg}.npz")
    r = H.quench_drift(O, g["segments_in"])
    n = len(r)
    lut = synth.make_lut((14, 26, 8), 48, int(g["n_prof"]), int(g["lut_seed"]))
    opc = g["op_channel"]
    nt = int(g["n_ticks"])
    ch = ChargeChain()
    ch.upload(r, np.zeros(n, dtype=np.int32))
    ch.light_incidence(lut)
    ids = np.arange(n, dtype='i8')
    # (first, last, truth slots, ticks): overlapping and disjoint segment ranges, a shorter tick axis, another slot count
    calls = [(0, n, 4, nt), (0, n // 2, 4, nt), (n // 3, n, 4, max(nt // 2, 64)), (0, n, 4, nt), (0, n, 2, nt), (n // 2, n, 2, nt)]

    def one(b, e, mt, ticks):
        ch.sum_light(b, e, opc, ids[b:e], max_truth=mt, max_ticks=ticks)
        return ch.download_light()

    got = [one(*c) for c in calls]                       # in a row: every sum after the first may take the lazy path
    for c, (out, tid, tph) in zip(calls, got):
        ch.sum_light(0, n, opc, max_truth=0, max_ticks=nt)          # no truth slots: invalidates the lazy state
        ref_out, ref_tid, ref_tph = one(*c)
        assert np.array_equal(out, ref_out) and np.array_equal(tid, ref_tid) and np.array_equal(tph, ref_tph), c
    assert any((t[1] != -1).any() for t in got)


@pytest.mark.gpu
def test_resident_photon_sums_without_truth_slots_over_the_lit_tiles():
    """ldsim_dev_sum_light without truth slots sums over a device-built list of the (detector, tick tile) cells some deposit falls
    into and, when the same buffer served such a sum before, clears only the tiles that sum listed.  ndlar batches (each lights its
    own TPCs' rows of 3360), tick axes of one to six tiles (whole and ragged), a shorter channel list and an empty range in a row on one context:
    every array equals the one the grid over all (detector, tile) cells writes into a fully cleared array
    (option light_sum_no_list) -- the same cells non-zero, values to the order of the f64 additions."""
    seg, bid = _prepared_set("ndlar", 20_000, 5)
    synth.set_synthetic_light(48)
    lut = synth.make_lut((14, 26, 8), 48, 100, synth.SEED_BASE + 5)
    opc = consts.light.TPC_TO_OP_CHANNEL[:].ravel().astype('i4')
    n_sim = int((bid >= 0).sum())
    edges = np.flatnonzero(np.r_[True, bid[1:n_sim] != bid[:n_sim - 1], True])
    assert len(edges) > 40
    rng = np.random.default_rng(11)
    calls = []
    for k, (b, e) in enumerate(zip(edges[:-1], edges[1:])):
        if k >= 36:
            break
        ticks = [11000, 2048, 5000, 8192, 2049][k % 5]
        ch_list = opc if k % 7 else np.ascontiguousarray(opc[rng.permutation(len(opc))[:1000]])
        calls.append((int(b), int(e), ch_list, ticks))
    calls.insert(5, (int(edges[3]), int(edges[3]), opc, 11000))            # an empty range: everything back to zero
    calls.append((0, n_sim, opc, 11000))                                    # every batch at once: many rows lit

    def run(no_list, own_stream=False):
        ch = ChargeChain(H.response_for("survey")) if own_stream else ChargeChain()
        lib.set_option("light_sum_no_list", 1 if no_list else 0, ch.ctx)
        lib.set_option("light_sum_async", 1 if own_stream else 0, ch.ctx)
        try:
            ch.upload(seg, bid)
            ch.quench_drift()
            ch.light_incidence(lut)
            outs = []
            for k, (b, e, cl, ticks) in enumerate(calls):
                n_ticks, _ = ch.sum_light(b, e, cl, max_truth=0, max_ticks=ticks)
                assert n_ticks == ticks
                if own_stream and k % 3 == 0 and e > b:      # the charge chain of the same batch in the ctx's stream meanwhile
                    ch.run(b, min(e, b + 300))
                if own_stream and k % 4 == 1:                # sums in a row on the light stream, no consumer between them
                    continue
                outs.append((k, ch.download_light()[0]))
            return outs
        finally:
            lib.set_option("light_sum_no_list", 0, ch.ctx)
            lib.set_option("light_sum_async", 0, ch.ctx)

    got, ref = run(False), run(True)
    lit_rows = set()
    for c, (_, a), (_, r) in zip(calls, got, ref):
        assert a.shape == r.shape == (len(c[2]), c[3])
        assert np.array_equal(a != 0, r != 0), c[:2]
        np.testing.assert_allclose(a, r, rtol=1e-6, atol=0)
        lit_rows.add(tuple(np.flatnonzero((r != 0).any(axis=1))[:4]))
    assert not got[5][1].any() and got[-1][1].any()
    assert len(lit_rows) > 10           # the batches really light different rows: a tile left uncleared would have shown
    # the same sums on the light stream (option light_sum_async), charge-chain launches and further sums in between: every
    # array a consumer fetches is the one the sum in the ctx's stream gave
    ref_by_k = dict(ref)
    own = run(False, own_stream=True)
    assert 20 < len(own) < len(calls)
    for k, a in own:
        assert np.array_equal(a != 0, ref_by_k[k] != 0), calls[k][:2]
        np.testing.assert_allclose(a, ref_by_k[k], rtol=1e-6, atol=0)


def _lsb_mismatch(got, ref, lsb):
    """fraction of values that differ, and whether every difference is exactly one digitiser LSB"""
    d = np.abs(np.asarray(got) - np.asarray(ref))
    bad = d != 0
    return bad.mean(), bool(np.all(d[bad] == lsb))


@pytest.mark.parametrize("name", H.LIGHT_WVFM_CASES)
def test_light_waveform_chain_golden(name):
    """Second half of the light chain through the C-ABI against the reference's own functions (light_sim.py:186-238,
    339-619; fixtures of oracle/gen_golden.py gen_light_wvfm): Poisson counts and the advanced random states, trigger ticks
    (module by module, with the reference's re-slicing bookkeeping), noise from given phases, digitised waveforms and their
    truth slots.  Exact where the arithmetic is integer or a copy; where a libm call of the device (exp, log, cos, sincos)
    sits in front of an integer rounding, a difference of one count / one LSB is tolerated in < 1e-3 of the values."""
    from larndsim_amd import rng as lrng
    g = H.load_light_wvfm_case(name)
    icase = H.LIGHT_WVFM_CASES.index(name)
    light = consts.light
    # -- calc_stat_fluctuations
    inc = g["fluct_inc"]
    D0, T0 = inc.shape
    st = lrng.create_xoroshiro128p_states(D0 * T0, seed=4242 + icase)
    assert np.array_equal(st.copy_to_host().view('u8').reshape(-1, 2), g["fluct_states_before"])
    disc = np.full((D0, T0), -1, dtype='f4')
    light_sim.calc_stat_fluctuations[(D0, -(-T0 // 64)), (1, 64)](inc, disc, st)
    assert np.array_equal(st.copy_to_host().view('u8').reshape(-1, 2), g["fluct_states_after"])
    frac, one = _lsb_mismatch(disc, g["fluct_disc"], np.float32(1.0 / light.LIGHT_TICK_SIZE))
    assert frac < 1e-3 and one, (frac, one)
    assert np.array_equal(disc[inc <= 0], np.zeros((inc <= 0).sum(), dtype='f4'))
    with pytest.raises(IndexError):
        light_sim.calc_stat_fluctuations[1, 1](np.ones((D0 + 1, T0), dtype='f4'), np.zeros((D0 + 1, T0), dtype='f4'), st)
    # -- get_triggers
    trig, trig_op, trig_type = light_sim.get_triggers(g["response"], g["group_threshold"], g["op_channel"], 0)
    assert np.array_equal(trig, g["trigger_idx"]) and np.array_equal(trig_op, g["trigger_op_channel_idx"])
    assert np.array_equal(trig_type, g["trigger_type"]) and len(trig) > 0
    assert len(light_sim.get_triggers(g["response"], g["group_threshold"], g["op_channel"], 1)[0]) == int(g["n_trig_subbatch1"])
    quiet = np.zeros_like(g["response"])
    assert len(light_sim.get_triggers(quiet, g["group_threshold"], g["op_channel"], 0)[0]) == (0 if light.LIGHT_TRIG_MODE == 0 else 1)
    # -- gen_light_detector_noise with the recorded phases
    lsb = 2.0 ** (16 - light.LIGHT_NBIT)
    for n in (1500, 1501):
        ref = g[f"noise_{n}"]
        ph = H.det_phases((ref.shape[0], n // 2 + 1), int(g[f"noise_{n}_phase_seed"]))
        got = light_sim.gen_light_detector_noise(ref.shape, g["noise_spectrum"][:ref.shape[0]], phases=ph)
        frac, one = _lsb_mismatch(got, ref, lsb)
        assert frac < 1e-3 and one and np.abs(ref).max() > 100 * lsb, (n, frac, one)
    with pytest.raises(lib.LdsimError, match="at least 2 samples"):
        light_sim.gen_light_detector_noise((2, 1), g["noise_spectrum"][:2])
    # -- sim_triggers + digitize_signal: zero spectrum is deterministic
    keep = g["wvfm_keep_rows"]
    ns = int(g["digit_samples"])
    args = (g["response"][keep], g["op_channel"][keep], g["response_true_id"][keep].astype('i8'),
            g["response_true_photons"][keep], g["trigger_idx"], g["trigger_op_channel_idx"], ns)
    d, dt, dp = light_sim.sim_triggers(None, None, *args, np.zeros_like(g["noise_spectrum"]))
    assert np.array_equal(d, g["wvfm_quiet"]) and np.array_equal(dt, g["wvfm_true_id"])
    np.testing.assert_allclose(dp, g["wvfm_true_photons"], rtol=1e-15, atol=0)
    d0, _, _ = light_sim.sim_triggers(None, None, *args, None)
    assert np.array_equal(d0, d)
    # with the recorded phases of both noise calls
    pre = int(np.ceil(light.LIGHT_TRIG_WINDOW[0] / light.LIGHT_TICK_SIZE)); post = int(np.ceil(light.LIGHT_TRIG_WINDOW[1] / light.LIGHT_TICK_SIZE))
    tmin, tmax = int(g["trigger_idx"].min()), int(g["trigger_idx"].max())
    n0 = max(pre - tmin, 0)
    Tp = g["response"].shape[1] + n0
    Tp += max(post + tmax + n0 - Tp, 0)
    seeds = g["wvfm_noisy_phase_seeds"]
    n_missing = len(np.setdiff1d(np.unique(g["trigger_op_channel_idx"]), g["op_channel"][keep]))
    ph_sig = H.det_phases((int(keep.sum()), Tp // 2 + 1), int(seeds[0]))
    ph_mis = H.det_phases((n_missing, Tp // 2 + 1), int(seeds[1])) if len(seeds) > 1 else None
    d2, dt2, _ = light_sim.sim_triggers(None, None, *args, g["noise_spectrum"], phases_signal=ph_sig, phases_missing=ph_mis)
    frac, one = _lsb_mismatch(d2, g["wvfm_noisy"], lsb)
    assert frac < 1e-3 and one, (frac, one)
    assert np.array_equal(dt2, dt)
    # internal phases: needs a seeded generator, is reproducible with it, and has the same power
    lib.check(lib.load().ldsim_rng_clear(lib.context()))
    with pytest.raises(lib.LdsimError, match="ldsim_rng_seed first"):
        light_sim.sim_triggers(None, None, *args, g["noise_spectrum"])
    lrng.create_xoroshiro128p_states(16, seed=5)
    a = light_sim.sim_triggers(None, None, *args, g["noise_spectrum"])[0]
    lrng.create_xoroshiro128p_states(16, seed=5)
    b = light_sim.sim_triggers(None, None, *args, g["noise_spectrum"])[0]
    assert np.array_equal(a, b) and not np.array_equal(a, d2)
    assert 0.8 < (a - d).std() / (g["wvfm_noisy"] - g["wvfm_quiet"]).std() < 1.25


def test_resident_light_waveform_chain_vs_oracle():
    """The light leg from the resident segments to digitised waveforms without leaving HBM between the stages
    (sum_light -> light_response -> get_triggers(None) -> sim_triggers(None)) against the oracle fed with the downloaded
    photon sum: scintillation, Poisson counts (same state table), detector response, trigger ticks, waveforms, truth."""
    from larndsim_amd import rng as lrng
    cfg = "module0"
    H.load_cfg(cfg)
    light = consts.light
    light.LIGHT_WINDOW = (0.2, 1.2)                       # 1000-tick convolutions keep the oracle's O(T*C) loops short
    light.LIGHT_TRIG_WINDOW = (0.2, 0.5)
    consts.sim.MAX_MC_TRUTH_IDS = 2
    g = H.gold(f"light_{cfg}.npz")
    r = H.quench_drift(O, g["segments_in"])
    n = len(r)
    lut = synth.make_lut((14, 26, 8), 48, int(g["n_prof"]), int(g["lut_seed"]))
    ch = ChargeChain()
    ch.upload(r, np.zeros(n, dtype=np.int32))
    ch.light_incidence(lut)
    opc = light.TPC_TO_OP_CHANNEL[:].ravel()
    n_ticks, t_start = ch.sum_light(0, n, opc, np.arange(n, dtype='i8'))
    inc, tid, tph = ch.download_light()
    assert inc.sum() > 0 and n_ticks > 1400
    nd = opc.shape[0]
    ch.seed_rng(77, nd * n_ticks)
    states = O.rng_create_states(nd * n_ticks, 77)
    ch.light_response(fluctuate=True)
    sc, di, resp, rtid, rtph = ch.download_light_response(stages=True)
    o_sc, o_stid, o_stph = O.scintillation_effect(inc, tid, tph)
    assert np.array_equal(sc, o_sc)
    o_di = O.stat_fluctuations(o_sc, states)
    frac, one = _lsb_mismatch(di, o_di, np.float32(1.0 / light.LIGHT_TICK_SIZE))
    assert frac < 1e-3 and one and (o_di > 0).sum() > 100, (frac, one)
    # downstream of the fluctuations compare on the device's own counts, so that a one-count difference cannot propagate
    o_resp, o_rtid, o_rtph = O.light_detector_response(di, light.LIGHT_GAIN, light.IMPULSE_MODEL, o_stid, o_stph)
    assert np.array_equal(resp, o_resp) and np.array_equal(rtid, o_rtid)
    np.testing.assert_allclose(rtph, o_rtph, rtol=1e-12, atol=0)
    per = light.OP_CHANNEL_PER_TRIG
    gsum = resp.reshape(-1, per, n_ticks).sum(axis=1)
    thr = np.full(nd // per, float(np.sort(gsum.min(axis=1))[nd // per // 2]) * 0.5)       # about half of the groups fire
    trig, trig_op, trig_type = light_sim.get_triggers(None, thr, opc, 0)
    o_trig, o_op, o_type = O.get_triggers(resp, thr, opc, 0)
    assert len(trig) > 0 and np.array_equal(trig, o_trig) and np.array_equal(trig_op, o_op) and np.array_equal(trig_type, o_type)
    ns = int(np.ceil((light.LIGHT_TRIG_WINDOW[1] + light.LIGHT_TRIG_WINDOW[0]) / light.LIGHT_DIGIT_SAMPLE_SPACING))
    d, dt, dp = light_sim.sim_triggers(None, None, None, opc, None, None, trig, trig_op, ns, None)
    o_d, o_dt, o_dp = O.sim_triggers(resp, opc, rtid, rtph, trig, trig_op, ns, np.zeros((light.N_OP_CHANNEL, 4)))
    assert np.array_equal(d, o_d) and np.array_equal(dt, o_dt) and (d != 0).sum() > 50 and (dt >= 0).sum() > 10
    np.testing.assert_allclose(dp, o_dp, rtol=1e-12, atol=0)
    # the host-buffer calls on the downloaded arrays are the same kernels
    assert np.array_equal(light_sim.get_triggers(resp, thr, opc, 0)[0], trig)
    d_h = light_sim.sim_triggers(None, None, resp, opc, rtid, rtph, trig, trig_op, ns, None)[0]
    assert np.array_equal(d_h, d)
    ms = ch.light_response_ms()
    assert all(v > 0 for v in ms.values())


@pytest.mark.parametrize("cfg,kind,fractions", [("module0", "survey", True), ("ndlar", "golden", True), ("2x2_no_modvar", "survey", False)])
def test_compact_download_expands_to_the_dense_rows(cfg, kind, fractions):
    """ldsim_chain_compact_build / _download: hit pixels, hits, track slots and per-hit fractions gathered on the device;
    chain.expand_compact gives back the dense rows of the hit pixels, equal to the rows ldsim_chain_download returns (the
    fractions on the written slots -- the slot after a pixel's last hit holds an un-normalised residue in the dense array,
    fee.py:572-573, which no exporter reads)."""
    from larndsim_amd.chain import expand_compact
    seg, bid = _two_event_set(cfg, 43)
    ch = ChargeChain(H.response_for(kind))
    ch.upload(seg, bid)
    ch.quench_drift()
    st = ch.run(0, len(seg), want_fractions=fractions)
    dense = ch.download()
    c = ch.download_compact()
    e = expand_compact(c)
    rows = e["row"]
    has_hit = dense["adc_list"][:, 0] != 0
    assert np.array_equal(np.flatnonzero(has_hit), rows) and len(rows) > 50
    assert len(c["hit_rows"]) == int((dense["adc_list"] != 0).sum()) == ch.compact_hits()[1]
    for k in ("unique_pix", "batch", "adc_list", "adc_ticks_list", "adc_digit", "track_pixel_map"):
        assert np.array_equal(e[k], dense[k][rows]), k
    if fractions:
        written = dense["adc_list"][rows] != 0
        assert np.array_equal(e["current_fractions"][written], dense["current_fractions"][rows][written])
        assert not e["current_fractions"][~written].any()
        nbytes = sum(v.nbytes for v in c.values() if hasattr(v, "nbytes"))
        assert nbytes < 0.02 * sum(v.nbytes for v in dense.values())          # a few percent of the dense arrays at most
    else:
        assert "current_fractions" not in e and len(c["fractions"]) == 0
    # built for the last launch only
    ch.run(0, len(seg) // 2, want_fractions=fractions)
    with pytest.raises(lib.LdsimError, match="compact_build"):
        lib.check(lib.load().ldsim_chain_compact_download(ch.ctx, None, None, None, None, None))


def test_compact_hit_rows_decode_to_the_downloaded_arrays():
    """The 24-byte rows the multi-GPU exchange moves ({batch, pixel, adc, slot, tick}: ldsim_chain_compact_hits ->
    ldsim_hits_accumulate -> ldsim_comm_allgather_hits -> ldsim_comm_gathered_download, here through a one-rank RCCL
    communicator) hold exactly the written slots of ldsim_chain_download, over two chain launches."""
    from larndsim_amd.comm import Communicator
    H.load_cfg("module0")
    seg = synth.make_segments(600, seed=31, segs_per_event=200)
    batching.swap_coordinates(seg)
    bid, order, table = batching.assign_batches(seg)
    seg, bid = seg[order], bid[order]
    nsim = int((bid >= 0).sum())
    ch = ChargeChain(synth.make_response("survey"))
    ch.upload(seg, bid)
    ch.quench_drift()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    cm = Communicator(ch.ctx, 0, 1)
    try:
        cut = int(np.searchsorted(bid[:nsim], 2))
        want = []
        for i, (b0, b1) in enumerate(((0, cut), (cut, nsim))):
            ch.run(b0, b1)
            res = ch.download()
            assert ch.compact_hits()[1] == int((res["adc_list"] != 0).sum()) and ch.compact_hits()[2] == 24
            cm.accumulate_hits(reset=(i == 0))
            u, h = np.nonzero(res["adc_list"] != 0)
            want.append(np.stack([res["batch"][u], res["unique_pix"][u], res["adc_digit"][u, h].astype(np.int64), h,
                                  res["adc_ticks_list"][u, h].view(np.int64)], axis=1))
        want = np.concatenate(want)
        total, counts, rows = cm.allgather_hits(download=True)
        assert total == len(want) == counts[0] and total > 50
        got = np.stack([rows["batch"], rows["pixel"], rows["adc"], rows["slot"], rows["tick"].view(np.int64)], axis=1)
        assert np.array_equal(got, want)
        assert cm.allreduce(3.5, "max") == 3.5 and cm.allreduce(2.0) == 2.0
    finally:
        cm.destroy()


def test_light_waveform_chain_properties_full_spill():
    """One full 2x2 spill (5000 segments, all 384 channels, 16000 ticks, the shipped 16 us window) through the resident
    light waveform chain, checked by properties that do not need the O(T*C) oracle loops: the scintillation stage conserves
    photons up to the window's end, the Poisson stage conserves them statistically, the SiPM stage is the gain-scaled linear
    convolution with the impulse model (FFT reference), the beam trigger digitises every channel, samples are LSB multiples."""
    H.load_cfg("2x2_no_modvar")
    consts.sim.MAX_MC_TRUTH_IDS = 0
    light = consts.light
    seg = synth.make_segments(5000, seed=20241016 + 3, segs_per_event=5000, spill=True)
    batching.swap_coordinates(seg)
    seg = seg[batching.select_active_volume(seg, consts.detector.TPC_BORDERS)]
    bid, order, table = batching.assign_batches(seg)
    seg, bid = np.ascontiguousarray(seg[order]), bid[order]
    assert len(table) == 1
    lut = synth.make_lut((14, 26, 8), 48, 100, 3)
    ch = ChargeChain()
    ch.upload(seg, bid)
    ch.quench_drift()
    ch.light_incidence(lut)
    opc = light.TPC_TO_OP_CHANNEL[:].ravel()
    n_ticks, t_start = ch.sum_light(0, len(seg), opc)
    assert n_ticks == 16000 and t_start == 0
    inc, _, _ = ch.download_light()
    ch.seed_rng(11, opc.shape[0] * n_ticks)
    ch.light_response(fluctuate=True)
    sc, di, resp, _, _ = ch.download_light_response(stages=True)
    tick = light.LIGHT_TICK_SIZE
    # scintillation: every photon is spread over the profile; what falls past the end of the array is lost, nothing is gained
    C = int(np.ceil((light.LIGHT_WINDOW[1] - light.LIGHT_WINDOW[0]) / tick))
    n = np.arange(C + 1)
    w = (light.SINGLET_FRACTION * np.exp(-n * tick / light.TAU_S) * (1 - np.exp(-tick / light.TAU_S))
         + (1 - light.SINGLET_FRACTION) * np.exp(-n * tick / light.TAU_T) * (1 - np.exp(-tick / light.TAU_T)))
    tot_in, tot_sc = inc.sum(axis=1, dtype=np.float64), sc.sum(axis=1, dtype=np.float64)
    live = tot_in > 0
    assert live.sum() > 300 and (tot_sc[live] <= tot_in[live] * (w.sum() + 1e-5)).all()
    assert (tot_sc[live] > 0.95 * tot_in[live]).all()            # arrivals sit in the first microseconds of a 16 us array
    ref_sc = np.array([np.convolve(inc[d].astype(np.float64), w)[:n_ticks] for d in np.flatnonzero(live)[:4]])
    np.testing.assert_allclose(sc[np.flatnonzero(live)[:4]], ref_sc, rtol=2e-4, atol=1e-6 * ref_sc.max())
    # Poisson: integer counts per tick, totals within 6 sigma of the expectation
    counts = di.astype(np.float64) * tick
    assert np.array_equal(counts, np.round(counts)) and (di[sc <= 0] == 0).all()
    mean = tot_sc * tick
    assert (np.abs(counts.sum(axis=1) - mean)[live] < 6 * np.sqrt(mean[live]) + 6).all()
    # SiPM response = LIGHT_GAIN[row] * (disc convolved with the impulse model resampled to the tick), f4 accumulation
    imp = np.asarray(light.IMPULSE_MODEL, dtype=float)
    idx = n * tick / light.IMPULSE_TICK_SIZE
    i0 = np.floor(idx).astype(int)
    wi = np.where(i0 > len(imp) - 2, np.where((i0 == idx) & (i0 <= len(imp) - 1), imp[np.minimum(i0, len(imp) - 1)], 0.0),
                  imp[np.minimum(i0, len(imp) - 2)] + (imp[np.minimum(i0 + 1, len(imp) - 1)] - imp[np.minimum(i0, len(imp) - 2)]) * (idx - i0))
    wi = wi / (light.IMPULSE_TICK_SIZE / tick)
    rows = np.flatnonzero(live)[:4]
    ref_r = np.array([light.LIGHT_GAIN[d] * np.convolve(di[d].astype(np.float64), wi)[:n_ticks] for d in rows])
    np.testing.assert_allclose(resp[rows], ref_r, rtol=5e-4, atol=2e-6 * np.abs(ref_r).max())
    assert resp.min() < 0 <= resp.max() + 1e-3                   # negative-going pulses (gain < 0)
    # beam trigger mode: one trigger at tick 0 on every channel; LSB-quantised waveforms that carry the pulses
    thr = np.repeat(np.array(light.LIGHT_TRIG_THRESHOLD)[..., None], light.OP_CHANNEL_PER_TRIG, axis=-1).ravel()[opc]
    thr = thr.reshape(-1, light.OP_CHANNEL_PER_TRIG)[..., 0].copy()
    trig, trig_op, trig_type = light_sim.get_triggers(None, thr, opc, 0)
    assert trig.tolist() == [0] and trig_type.tolist() == [1] and trig_op.shape == (1, 384)
    ns = int(np.ceil((light.LIGHT_TRIG_WINDOW[1] + light.LIGHT_TRIG_WINDOW[0]) / light.LIGHT_DIGIT_SAMPLE_SPACING))
    wv, _, _ = light_sim.sim_triggers(None, None, None, opc, None, None, trig, trig_op, ns, None)
    lsb = 2.0 ** (16 - light.LIGHT_NBIT)
    assert wv.shape == (1, 384, ns) and np.array_equal(wv, np.round(wv / lsb) * lsb)
    # sample i reads padded tick 16 i; the padding in front is LIGHT_TRIG_WINDOW[0] / tick ticks of zeros
    pre = int(np.ceil(light.LIGHT_TRIG_WINDOW[0] / tick))
    step = int(round(light.LIGHT_DIGIT_SAMPLE_SPACING / tick))
    k = np.arange(ns) * step - pre
    ok = (k >= 0) & (k < n_ticks)
    expect = np.zeros((384, ns))
    expect[:, ok] = resp[:, k[ok]]
    assert np.array_equal(wv[0], np.round(expect / lsb) * lsb)


def test_stage_call_between_upload_and_run_is_refused():
    """The host-buffer stage calls upload their records into the store the resident chain uses.  A chain call that finds
    the store taken over by a stage call must refuse (LDSIM_ESTATE) instead of simulating the stage call's records."""
    H.load_cfg("module0")
    seg = synth.make_segments(40, seed=3, segs_per_event=40)
    batching.swap_coordinates(seg)
    other = synth.make_segments(64, seed=4, segs_per_event=64)
    batching.swap_coordinates(other)
    ch = ChargeChain(H.response_for("survey"))
    ch.upload(seg, np.zeros(len(seg), dtype=np.int32))
    ch.quench_drift()
    good = ch.run(0, len(seg))
    ref = ch.download()
    quenching.quench[1, 256](other, consts.physics.BIRKS)           # a stage call in between (the reference loop does this)
    for call in (lambda: ch.run(0, len(seg)), ch.quench_drift, ch.reset, lambda: ch.download_segments(seg.copy())):
        with pytest.raises(lib.LdsimError, match="stage call replaced the resident segments"):
            call()
    ch.upload(seg, np.zeros(len(seg), dtype=np.int32))
    ch.quench_drift()
    again = ch.run(0, len(seg))
    assert again.n_unique == good.n_unique
    out = ch.download()
    for k in ref:
        assert np.array_equal(ref[k], out[k]), k
    # batch ids must be non-decreasing over the simulated segments, also across skipped (negative) ones
    with pytest.raises(lib.LdsimError, match="non-decreasing"):
        ch.upload(seg, np.array([5, -1, 3] + [6] * (len(seg) - 3), dtype=np.int32))


def test_light_properties_baseline_sizes():
    """BASELINE.json config 5, light leg at full size: ndlar, 1 M segments, synthetic light set-up of SURVEY 8d (48 channels
    per TPC, 3360 channels), device-resident.  Per event: a segment lights only its own TPC's channels, voxels are inside
    the LUT, and the photon sum closes -- every channel's waveform integrates to the photons incident on it."""
    seg, bid = _prepared_set("ndlar", 1_000_000, 5)
    n_op = synth.set_synthetic_light(48)
    lut = synth.make_lut((14, 26, 8), 48, 100, synth.SEED_BASE + 5)
    ch = ChargeChain()
    ch.upload(seg, bid)
    ch.quench_drift()
    ch.light_incidence(lut)
    got = ch.download_segments(seg.copy())
    opc = consts.light.TPC_TO_OP_CHANNEL[:].ravel().astype('i4')
    c2t = np.asarray(consts.light.OP_CHANNEL_TO_TPC)
    tick = consts.light.LIGHT_TICK_SIZE
    n_sim = int((bid >= 0).sum())            # unsimulated segments (none in the synthetic set) sit behind the batches
    assert n_sim > 0.99 * len(seg)
    ev = got["event_id"][:n_sim]
    edges = np.flatnonzero(np.r_[True, ev[1:] != ev[:-1], True])
    assert len(edges) - 1 == 200
    checked = 0
    for b, e in zip(edges[:-1], edges[1:]):
        inc, vox = ch.download_light_incidence(b, e)
        r = got[b:e]
        in_tpc = r["pixel_plane"] != consts.detector.DEFAULT_PLANE_INDEX
        assert ((vox[in_tpc] >= 0) & (vox[in_tpc] < np.array([14, 26, 8]))).all()
        nph = inc['n_photons_det']
        own = c2t[None, :] == r["pixel_plane"][:, None]
        assert (nph[~own] == 0).all()
        lit = own & in_tpc[:, None] & (r["n_photons"] > 0)[:, None]
        assert (nph[lit] > 0).all()
        n_ticks, t_start = ch.sum_light(b, e, opc, max_truth=0)
        out, _, _ = ch.download_light()
        assert n_ticks == 11000 and t_start == 0
        np.testing.assert_allclose(out.sum(axis=1, dtype=np.float64) * tick, nph.sum(axis=0, dtype=np.float64), rtol=2e-5,
                                   atol=1e-3)
        checked += e - b
    assert checked == n_sim


def test_rng_table_growth_follows_maybe_create_rng_states():
    """rng.maybe_create_rng_states (cli/simulate_pixels.py:92-104): a table that is long enough is untouched; a shorter one
    keeps its states (advanced or not) and gets a fresh create_xoroshiro128p_states(n - len, seed) chain appended; the
    chain's own growth (more pixel rows than states) continues the last chain instead."""
    H.load_cfg("module0")
    st = lrng.create_xoroshiro128p_states(100, 3)
    first = st.copy_to_host().copy()
    assert np.array_equal(first.view(np.uint64), O.rng_create_states(100, 3).view(np.uint64))
    assert lrng.maybe_create_rng_states(80, 5, st) is st and len(st) == 100
    assert np.array_equal(st.copy_to_host().view(np.uint64), first.view(np.uint64))
    lrng.maybe_create_rng_states(150, 9, st)
    grown = st.copy_to_host()
    assert len(st) == 150 and grown.shape[0] == 150
    assert np.array_equal(grown[:100].view(np.uint64), first.view(np.uint64))
    assert np.array_equal(grown[100:].view(np.uint64), O.rng_create_states(50, 9).view(np.uint64))
    fresh = lrng.maybe_create_rng_states(7, 11, None)
    assert np.array_equal(fresh.copy_to_host().view(np.uint64), O.rng_create_states(7, 11).view(np.uint64))


def test_fee_noise_stream_vs_oracle():
    """FEE noise (fee.py:557,583-584,616-617,621,649) with the xoroshiro128p + Box-Muller generator Numba documents
    (csrc/rng.h; oracle/ldsim_oracle.c holds the same restatement -- the stream is third-party and unpinned).  On the
    golden chain's pixel waveforms: the states after the call are bit-identical to the oracle's, i.e. every pixel drew
    exactly as many numbers as the oracle did (same triggers, failed triggers and resets); hit slots and tick stamps
    agree; charges agree to the float32 normals' last bit (device vs host logf / cosf)."""
    H.load_cfg("module0", noise_zero=False)
    d = consts.detector
    assert d.RESET_NOISE_CHARGE > 0 and d.UNCORRELATED_NOISE_CHARGE > 0 and d.DISCRIMINATOR_NOISE > 0
    g = H.gold("chain_module0.npz")
    ps = np.ascontiguousarray(g["pixels_signals"], dtype=np.float64)
    U, NT = ps.shape
    # more pixels, with charges around the threshold so that the noise decides: scaled copies of the golden waveforms
    scale = np.r_[np.ones(U), np.linspace(0.05, 1.5, 40 * U)]
    ps = np.concatenate([ps] + [ps] * 40) * scale[:, None]
    pts = None
    U = ps.shape[0]
    tt = np.linspace(0, d.TIME_INTERVAL[1], NT + 1)
    thr = np.full(U, d.DISCRIMINATION_THRESHOLD * 1.0)
    st = O.rng_create_states(U, 20241016)
    adc_o, ticks_o, _ = O.get_adc_values(ps, pts, tt, thr, rng_states=st)
    A = consts.sim.MAX_ADC_VALUES
    adc = np.zeros((U, A)); ticks = np.zeros((U, A))
    states = lrng.create_xoroshiro128p_states(U, 20241016)
    assert np.array_equal(states.copy_to_host().view(np.uint64), O.rng_create_states(U, 20241016).view(np.uint64))
    fee.get_adc_values[1, 128](ps, pts, tt, adc, ticks, 0, states, None, thr)
    after = states.copy_to_host()
    assert np.array_equal(after.view(np.uint64).ravel(), st.view(np.uint64).ravel()), "draw counts differ from the oracle"
    assert np.array_equal(adc != 0, adc_o != 0) and (adc_o != 0).sum() > 100
    assert np.array_equal(ticks, ticks_o)
    np.testing.assert_allclose(adc, adc_o, rtol=1e-6, atol=1e-2)
    # the noise really acted: against the noise-free scan some hits appear / vanish and the charges differ
    H.load_cfg("module0", noise_zero=True)
    adc0 = np.zeros((U, A)); ticks0 = np.zeros((U, A))
    fee.get_adc_values[1, 128](ps, pts, tt, adc0, ticks0, 0, None, None, thr)
    assert not np.array_equal(adc != 0, adc0 != 0)
    both = (adc != 0) & (adc0 != 0)
    assert np.abs(adc[both] - adc0[both]).mean() > 300          # ~ sqrt(900^2 + 500^2) electrons of reset + uncorrelated noise


def test_fee_noise_statistical_closure():
    """Noise closure on 40 000 identical pixels carrying one 20 ke- pulse: every pixel triggers once at the pulse, the
    reported charge is the pulse plus reset noise (drawn at the previous reset) plus uncorrelated noise, i.e. spread
    sqrt(RESET^2 + UNCORRELATED^2) = 1030 e-; pixels without signal do not trigger (threshold at 5.8 sigma)."""
    H.load_cfg("module0", noise_zero=False)
    d = consts.detector
    NT = len(d.TIME_TICKS)
    U = 40_000
    q_pulse = 20_000.0
    ps = np.zeros((U, NT))
    ps[: U // 2, 700] = q_pulse / d.TIME_SAMPLING
    tt = np.linspace(0, d.TIME_INTERVAL[1], NT + 1)
    thr = np.full(U, d.DISCRIMINATION_THRESHOLD * 1.0)
    A = consts.sim.MAX_ADC_VALUES
    adc = np.zeros((U, A)); ticks = np.zeros((U, A))
    states = lrng.create_xoroshiro128p_states(U, 11)
    fee.get_adc_values[1, 128](ps, None, tt, adc, ticks, 0, states, None, thr)
    sig, empty = adc[: U // 2], adc[U // 2:]
    assert (empty != 0).sum() <= 2
    first = sig[:, 0]
    assert (first != 0).all()
    assert ((sig[:, 1:] != 0).sum(axis=1) == 0).mean() > 0.995       # no second hit (the tail after the reset is ~1 % of the pulse)
    sigma = np.hypot(d.RESET_NOISE_CHARGE, d.UNCORRELATED_NOISE_CHARGE)
    assert abs(first.std() - sigma) < 0.03 * sigma
    assert abs(first.mean() - q_pulse) < 0.01 * q_pulse + 4 * sigma / np.sqrt(U // 2)


def test_fused_chain_with_fee_noise_vs_oracle():
    """The fused chain with the shipped noise charges: row u of the launch draws from state u of the table, so the oracle
    run per batch with the matching slice of states reproduces the launch: same hits, ticks and codes, charges to the
    float32 normals' last bit, and the table ends in the same states."""
    H.load_cfg("module0", noise_zero=False)
    seg = synth.make_segments(16, seed=21, segs_per_event=8)
    batching.swap_coordinates(seg)
    bid, order, table = batching.assign_batches(seg)
    seg, bid = seg[order], bid[order]
    resp = H.response_for("survey")
    ch = ChargeChain(resp)
    ch.upload(seg, bid)
    ch.quench_drift()
    ch.seed_rng(5)
    st = ch.run(0, len(seg), want_fractions=True)
    out = ch.download()
    states = O.rng_create_states(int(st.n_unique), 5)
    u0 = 0
    assert len(table) >= 2
    for b in range(len(table)):
        m = out["batch"] == b
        nb = int(m.sum())
        sl = states[u0:u0 + nb]
        o = _oracle_chain(seg[bid == b], resp, rng_states=sl)
        states[u0:u0 + nb] = sl
        u0 += nb
        assert np.array_equal(out["unique_pix"][m], o["unique_pix"])
        assert np.array_equal(out["adc_list"][m] != 0, o["adc"] != 0)
        np.testing.assert_allclose(out["adc_list"][m], o["adc"], rtol=1e-5, atol=1e-2)
        assert np.array_equal(out["adc_ticks_list"][m], o["ticks"])
        hit = o["adc"] != 0
        np.testing.assert_allclose(out["current_fractions"][m][hit], o["frac"][hit], rtol=1e-5, atol=1e-9)
    assert np.array_equal(ch.rng_states.copy_to_host(u0).view(np.uint64).ravel(), states.view(np.uint64).ravel())


def test_tracks_current_mc_vs_oracle_and_closure():
    """detsim.tracks_current_mc (what the reference driver calls, cli/simulate_pixels.py:1016).  Every (segment, pixel, tick)
    has its own stream here (the reference's tick threads race on one state), so parity is (a) with the oracle's
    restatement of the same rule: the table states end bit-identical and the currents agree except where a float32 normal's
    last bit moves a sample across a response-cell edge; (b) statistical closure with the deterministic integral
    tracks_current: the same charge is induced."""
    H.load_cfg("module0")
    seg = synth.make_segments(4, seed=17, segs_per_event=4)
    batching.swap_coordinates(seg)
    r = H.quench_drift(O, seg)
    nmax = O.max_pixels(r)
    P = 3 * nmax + 6
    _, neigh, nrad, _ = O.get_pixels(r, nmax, P, 1)
    _, T = O.time_intervals(r)
    resp = synth.make_response("golden")
    S = len(r)
    st = O.rng_create_states(S * P, 99)
    ref = O.tracks_current_mc(r, neigh, T, resp, st)
    states = lrng.create_xoroshiro128p_states(S * P, 99)
    sig = np.zeros((S, P, T), dtype=np.float32)
    detsim.tracks_current_mc[(S, P, 1), (1, 1, 64)](sig, neigh, r, resp, states)
    assert np.array_equal(states.copy_to_host().view(np.uint64).ravel(), st.view(np.uint64).ravel())
    assert np.array_equal(sig != 0, ref != 0) and (ref != 0).sum() > 5000
    peak = np.abs(ref).max(axis=-1, keepdims=True) + 1e-30
    close = np.abs(sig - ref) <= 1e-4 * np.abs(ref) + 1e-6 * peak
    assert close.mean() > 0.999
    # a second call continues the streams: different samples, same physics
    sig2 = np.zeros_like(sig)
    detsim.tracks_current_mc[(S, P, 1), (1, 1, 64)](sig2, neigh, r, resp, None)
    assert not np.array_equal(sig2, sig)
    det = O.tracks_current(r, neigh, T, resp)
    q_mc, q_mc2, q_det = sig.sum(dtype=np.float64), sig2.sum(dtype=np.float64), det.sum(dtype=np.float64)
    assert abs(q_mc / q_det - 1) < 0.03 and abs(q_mc2 / q_det - 1) < 0.03
    # per pair with a sizeable signal the two estimates of the induced charge agree within the sampling error
    big = np.abs(det).sum(axis=-1) > 0.05 * np.abs(det).sum(axis=-1).max()
    ratio = sig.sum(axis=-1, dtype=np.float64)[big] / det.sum(axis=-1, dtype=np.float64)[big]
    assert np.abs(ratio - 1).max() < 0.15


def test_fused_chain_with_mc_currents():
    """Option "mc_current": the fused chain takes its induced currents from tracks_current_mc (the reference driver's
    configuration) -- same pixels as the deterministic chain, charges equal within the Monte-Carlo error."""
    H.load_cfg("module0")
    seg = synth.make_segments(24, seed=27, segs_per_event=12)
    batching.swap_coordinates(seg)
    bid, order, table = batching.assign_batches(seg)
    seg, bid = seg[order], bid[order]
    ch = ChargeChain(H.response_for("survey"))
    ch.upload(seg, bid)
    ch.quench_drift()
    ch.run(0, len(seg))
    det = ch.download()
    try:
        lib.set_option("mc_current", 1)
        ch.seed_rng(123)
        ch.run(0, len(seg))
        mc = ch.download()
        ch.seed_rng(123)
        ch.run(0, len(seg))
        again = ch.download()
    finally:
        lib.set_option("mc_current", 0)
    for k in mc:
        assert np.array_equal(mc[k], again[k]), f"{k}: not reproducible with the same seed"
    assert np.array_equal(mc["unique_pix"], det["unique_pix"]) and np.array_equal(mc["track_pixel_map"], det["track_pixel_map"])
    q_det, q_mc = det["adc_list"].sum(), mc["adc_list"].sum()
    assert q_det > 0 and abs(q_mc / q_det - 1) < 0.03
    both = (det["adc_list"][:, 0] > 2e4) & (mc["adc_list"][:, 0] != 0)
    assert both.sum() >= 5
    assert np.abs(mc["adc_list"][both, 0] / det["adc_list"][both, 0] - 1).max() < 0.2


def _two_event_set(cfg, seed, n=1200):
    H.load_cfg(cfg)
    seg = synth.make_segments(n, seed=seed, segs_per_event=n // 2, spill=bool(consts.sim.IS_SPILL_SIM))
    if consts.sim.IS_SPILL_SIM:
        loc = seg["event_id"] % consts.sim.MAX_EVENTS_PER_FILE
        for f in ("t0", "t0_start", "t0_end"):
            seg[f] = seg[f] - loc * consts.sim.SPILL_PERIOD
    batching.swap_coordinates(seg)
    bid, order, table = batching.assign_batches(seg)
    return seg[order], bid[order]


def _reset_current_options():
    for name, v in (("split_kernels", 1), ("weights_mode", 2), ("wbuf_doubles_per_pair", 6144), ("split_max_items", 0),
                    ("quad_max_nodes", 256), ("numba_f32", 0), ("tail_log", 14.0), ("prune_log", 23.0), ("mac_mode", 1),
                    ("quad_accuracy_log10", 7), ("gform_max_support", 1e9), ("trim_response_log", 23.0),
                    ("gform_wave_tables", 1)):
        lib.set_option(name, v)


@pytest.mark.parametrize("cfg,kind", [("module0", "dense"), ("ndlar", "golden")])
@pytest.mark.parametrize("mode", [2, 1, 0])
def test_split_kernels_equal_monolithic(cfg, kind, mode):
    """weights stage + mac_kernel (default) vs the monolithic current_kernel on 2 x 600 segments, for both weights stages
    (weights_mode 1 = qweights_kernel, Gauss-Legendre along the segment; 0 = weights_kernel, the closed form per sample):
    same hits, charges equal to rounding (summation order / quadrature differ), incl. an item cap that sends many pairs
    through the overflow fallback.  A weight pool that starts far too small is grown and the launch repeated."""
    seg, bid = _two_event_set(cfg, 33)
    ch = ChargeChain(H.response_for(kind))
    ch.upload(seg, bid)
    ch.quench_drift()
    res = {}
    try:
        lib.set_option("weights_mode", mode)
        lib.set_option("gform_max_support", 1e9)       # mode 2: the matrix form whatever the table's support
        for name, split, cap, max_items in (("mono", 0, 6144, 0), ("split", 1, 65536, 0), ("tiny", 1, 50, 0),
                                            ("capped", 1, 6144, 60)):
            lib.set_option("split_kernels", split)
            lib.set_option("wbuf_doubles_per_pair", cap)
            lib.set_option("split_max_items", max_items)
            st = ch.run(0, len(seg), want_fractions=True)
            res[name] = ch.download()
            if name == "split":
                assert st.n_fallback < 0.01 * st.n_pairs
            if name == "tiny" and mode != 2:
                assert st.n_wbuf > 50 * st.n_pairs        # the pool was grown past its initial budget
            if name == "capped" and mode != 2:            # (the node-separable form has no weight pool and no item lists)
                assert st.n_fallback > 0.2 * st.n_pairs   # the fallback really carried a good share
    finally:
        _reset_current_options()
    a = res["mono"]
    assert (a["adc_list"] != 0).sum() > 100
    # regrown pool == roomy pool: the same pairs take the same path.  qweights_kernel is bitwise reproducible (every bin
    # is owned by one thread); weights_kernel's LDS atomics may move the last bit (see test_chain_properties_baseline_sizes)
    for k in res["split"]:
        if mode == 0 and k in ("adc_list", "current_fractions"):
            np.testing.assert_allclose(res["tiny"][k], res["split"][k], rtol=1e-12, atol=1e-15, err_msg=k)
        else:
            assert np.array_equal(res["split"][k], res["tiny"][k]), k
    for name in ("split", "capped"):
        b = res[name]
        assert np.array_equal(a["unique_pix"], b["unique_pix"]) and np.array_equal(a["track_pixel_map"], b["track_pixel_map"])
        assert np.array_equal(a["adc_list"] != 0, b["adc_list"] != 0)
        # the two paths drop different sets of weights below exp(-prune_log) = 1e-10 of the peak: charges within 2e-8
        np.testing.assert_allclose(b["adc_list"], a["adc_list"], rtol=2e-8)
        assert np.array_equal(a["adc_ticks_list"], b["adc_ticks_list"])
        assert np.array_equal(a["adc_digit"], b["adc_digit"])
        np.testing.assert_allclose(b["current_fractions"], a["current_fractions"], rtol=1e-7, atol=1e-10)


def test_overlapped_download_returns_the_rows_of_its_own_launch():
    """ldsim_chain_download_async: launch k's rows are copied on the second stream while launch k + 1 computes into the
    other set of output buffers; every chunk's arrays equal the synchronous download of the same chunk, also when a wait is
    skipped (the launch two later waits for the copy itself) and when synchronous and overlapped downloads are mixed."""
    H.load_cfg("module0")
    seg = synth.make_segments(4000, seed=23, segs_per_event=800)          # five events -> five batches
    batching.swap_coordinates(seg)
    bid, order, table = batching.assign_batches(seg)
    seg, bid = seg[order], bid[order]
    ch = ChargeChain(H.response_for("survey"))
    ch.upload(seg, bid)
    ch.quench_drift()
    cuts = [0] + [int(i) for i in np.flatnonzero(np.diff(bid)) + 1] + [len(seg)]     # one launch per batch
    assert len(cuts) >= 5
    ref = []
    for b, e in zip(cuts[:-1], cuts[1:]):
        ch.run(b, e, want_fractions=True)
        ref.append({k: v.copy() for k, v in ch.download().items()})
    got, prev = [], None
    for i, (b, e) in enumerate(zip(cuts[:-1], cuts[1:])):
        ch.run(b, e, want_fractions=True)          # overlaps with the copy started one iteration ago
        if prev is not None and i != 2:
            ch.wait_download()                       # (i == 2: no explicit wait; the download_async below waits for the copy)
        new = ch.download_async()
        if prev is not None:
            got.append({k: v.copy() for k, v in prev.items()})
        prev = new
    ch.wait_download()
    got.append({k: v.copy() for k, v in prev.items()})
    assert len(got) == len(ref) >= 3
    for a, r in zip(got, ref):
        assert set(a) == set(r)
        for k in r:
            assert np.array_equal(a[k], r[k]), k
    # a synchronous download after overlapped ones still returns the last launch
    ch.run(cuts[0], cuts[1], want_fractions=True)
    last = ch.download()
    for k in ref[0]:
        assert np.array_equal(last[k], ref[0][k]), k


@pytest.mark.parametrize("cfg,kind", [("module0", "survey"), ("module0", "dense"), ("ndlar", "golden"), ("ndlar", "dense")])
def test_mac_shift_kernel_is_bitwise_the_lds_kernel(cfg, kind):
    """mac_shift_kernel / mac_shift2_kernel (default: window slid through the wave with DPP shifts, weights through the
    scalar cache, zero-padded response rows) against mac_kernel<M> (rows staged in LDS), M = 1 (module0) and M = 2 (ndlar):
    same products in the same order, so every output is bit-identical -- incl. waveform windows that hang over either end
    of the response support (survey table: support narrower than a tile) and pairs whose items fill all 8 blocks."""
    H.load_cfg(cfg)
    seg, bid = _two_event_set(cfg, 35)
    ch = ChargeChain(H.response_for(kind))
    ch.upload(seg, bid)
    ch.quench_drift()
    res = {}
    try:
        lib.set_option("weights_mode", 1)              # the weights stage whose pool and item lists these kernels read
        for mode in (1, 0):
            lib.set_option("mac_mode", mode)
            st = ch.run(0, len(seg), want_fractions=True)
            assert st.n_dfma > 0 and st.n_wbuf > 0
            res[mode] = ch.download()
    finally:
        _reset_current_options()
    assert (res[1]["adc_list"] != 0).sum() > 100
    for k in res[1]:
        assert np.array_equal(res[1][k], res[0][k]), k
    with pytest.raises(lib.LdsimError, match="mac_mode"):
        lib.set_option("mac_mode", 2)


@pytest.mark.parametrize("cfg,kind", [("module0", "survey"), ("ndlar", "golden")])
@pytest.mark.parametrize("wmode", [2, 1])
def test_quadrature_weights_node_cap_and_pruning(cfg, kind, wmode):
    """The quadrature stages (weights_mode 2: gtables_kernel + gcorr_kernel, the default; 1: qweights_kernel + the shifted-window
    correlation) hand pairs that need more Gauss-Legendre nodes than "quad_max_nodes" to the monolithic kernel
    (the shipped cap of 256 covers segments up to ~130 Gaussian widths long): with a cap of 12 most pairs go that way and
    the result must not change; prune_log = 0 (every bin kept) must not change it either; and two runs are bitwise equal."""
    seg, bid = _two_event_set(cfg, 37)
    ch = ChargeChain(H.response_for(kind))
    ch.upload(seg, bid)
    ch.quench_drift()
    res = {}
    try:
        lib.set_option("weights_mode", wmode)
        lib.set_option("gform_max_support", 1e9)
        for name, cap, prune in (("default", 256, 23.0), ("again", 256, 23.0), ("cap12", 12, 23.0), ("keepall", 256, 0.0),
                                 ("acc12", 256, 23.0), ("acc10", 256, 23.0), ("prune30", 256, 30.0)):
            lib.set_option("quad_max_nodes", cap)
            lib.set_option("prune_log", prune)
            lib.set_option("quad_accuracy_log10", {"acc12": 12, "acc10": 10}.get(name, 7))
            st = ch.run(0, len(seg), want_fractions=True)
            res[name] = ch.download()
            if name == "default":
                assert st.n_fallback < 0.01 * st.n_pairs
                assert 6 <= st.n_samples / st.n_pairs < 64       # n_samples counts quadrature nodes in this mode
                # of the issued FMA lanes, the ones that are neither padding (8-shift blocks and the 512-tick tile of the
                # shifted-window kernels -- a window of the survey table fills a fifth of a tile once the ticks below 1e-10 of
                # the table's peak are not read; nodes to 16, cells to 4, ticks to 16 in the matrix form) nor outside the window
                assert (0.1 if wmode == 1 else 0.3) * st.n_dfma < st.n_dfma_useful <= st.n_dfma
            if name == "cap12":
                assert st.n_fallback > 0.3 * st.n_pairs
    finally:
        _reset_current_options()
    a = res["default"]
    assert (a["adc_list"] != 0).sum() > 100
    for k in a:
        assert np.array_equal(a[k], res["again"][k]), f"{k}: not bitwise reproducible"
    # the node rules against each other: 1e-10 of the peak weight (round 3's default) against 1e-12 as before; the shipped rule
    # (1e-7 of the peak weight: round 4, tools/quad_sweep.py -- per tick it sits at 0.012 of the parity tolerance against the
    # reference's goldens, where the f4 rounding of the stored currents already is) moves an ADC charge by less than 2e-7
    # relative, fifty times inside the 1e-5 bar; discrete outputs identical
    np.testing.assert_allclose(res["acc12"]["adc_list"], res["acc10"]["adc_list"], rtol=2e-9, atol=0)
    np.testing.assert_allclose(res["acc12"]["adc_list"], a["adc_list"], rtol=2e-7, atol=0)
    for name in ("acc12", "acc10"):
        assert np.array_equal(res[name]["adc_digit"], a["adc_digit"]) and np.array_equal(res[name]["adc_ticks_list"], a["adc_ticks_list"])
    # the shipped pruning (weights below exp(-23) = 1e-10 of the pair's peak dropped) against keeping every weight and against
    # the earlier exp(-30): charges within 2e-8, three orders inside the bar and below the f32 resolution of the reference's
    # own currents; discrete outputs identical
    for name in ("cap12", "keepall", "prune30"):
        b = res[name]
        assert np.array_equal(a["unique_pix"], b["unique_pix"]) and np.array_equal(a["track_pixel_map"], b["track_pixel_map"])
        assert np.array_equal(a["adc_list"] != 0, b["adc_list"] != 0)
        np.testing.assert_allclose(b["adc_list"], a["adc_list"], rtol=2e-8)
        assert np.array_equal(a["adc_ticks_list"], b["adc_ticks_list"])
        assert np.array_equal(a["adc_digit"], b["adc_digit"])
        np.testing.assert_allclose(b["current_fractions"], a["current_fractions"], rtol=1e-7, atol=1e-10)


@pytest.mark.parametrize("mode", [2, 1, 0])
def test_numba_f32_typing_mode_vs_oracle(mode):
    """Option "numba_f32": the sub-expressions Numba types float32 for f4 record fields (detsim.py:74-79,116-118,141,387)
    are evaluated in float, in both weights stages and the monolithic kernel; checked against the oracle's switch
    (o_set_numba_f32) on the materialising tracks_current and through the fused chain."""
    H.load_cfg("module0")
    response = H.response_for("golden")
    seg = synth.make_segments(24, seed=41, segs_per_event=24)
    batching.swap_coordinates(seg)
    ref = H.quench_drift(O, seg)
    nmax = O.max_pixels(ref)
    r = int(np.ceil(ref["tran_diff"].max() * 5 / consts.detector.PIXEL_PITCH))
    P = (2 * r + 1) * nmax + (1 + 2 * r) * r * 2
    _, neigh, nrad, _ = O.get_pixels(ref, nmax, P, r)
    starts, T = O.time_intervals(ref)
    plain = O.tracks_current(ref, neigh, T, response)
    O.lib().o_set_numba_f32(1)
    try:
        typed = O.tracks_current(ref, neigh, T, response)
    finally:
        O.lib().o_set_numba_f32(0)
    assert np.abs(typed - plain).max() > 0          # the switch really changes the waveforms
    try:
        lib.set_option("numba_f32", 1)
        got = np.zeros_like(typed)
        for split in (0, 1):                             # the monolithic kernel, then the weights stage under test
            lib.set_option("split_kernels", split)
            lib.set_option("weights_mode", mode)
            lib.set_option("gform_max_support", 1e9)
            detsim.tracks_current[1, 1](got, neigh, ref, response)
            H.assert_wave_close(got, typed, what=f"numba_f32 stage call, split_kernels {split}")
            assert (detsim.tracks_current_stats().n_wbuf > 0) == bool(split)
        # fused chain (weights stage `mode` + mac_kernel) on the same records: compare the pixel charges with a chain
        # assembled from the typed oracle waveforms
        lib.set_option("weights_mode", mode)
        ch = ChargeChain(response)
        ch.upload(ref, np.zeros(len(ref), dtype=np.int32))
        st = ch.run(0, len(ref), want_fractions=False)
        out = ch.download()
    finally:
        _reset_current_options()
    upix = O.unique_pixels(neigh)
    pim = O.pixel_index_map(neigh, upix)
    tpm = O.track_pixel_map(upix, neigh, nrad, int(nrad.max()) + 1, consts.sim.MAX_TRACKS_PER_PIXEL)
    ps, pts, _ = O.sum_pixel_signals(typed, starts, pim, tpm, len(upix))
    tt = np.linspace(0, consts.detector.TIME_INTERVAL[1], ps.shape[1] + 1)
    adc, ticks, _ = O.get_adc_values(ps, pts, tt, np.full(len(upix), consts.detector.DISCRIMINATION_THRESHOLD))
    assert np.array_equal(upix, out["unique_pix"])
    assert np.array_equal(adc != 0, out["adc_list"] != 0) and (adc != 0).sum() > 10
    np.testing.assert_allclose(out["adc_list"], adc, rtol=1e-6, atol=1e-6)
    assert np.array_equal(ticks, out["adc_ticks_list"])
    assert np.array_equal(O.digitize(adc), out["adc_digit"])


@pytest.mark.parametrize("cfg,kind", [("module0", "survey"), ("ndlar", "golden")])
def test_f32_tail_class_equals_all_f64(cfg, kind):
    """weights_kernel evaluates samples bounded by exp(-tail_log) of the segment's peak density in f32 (default 14).
    Against tail_log = 0 (every sample in f64): same hits, identical tick stamps and ADC counts, charges within 1e-9
    (measured on MI355X: 3e-13 at 14, 3e-9 at 8), i.e. far inside the 1e-5 tolerance of north_star."""
    H.load_cfg(cfg)
    seg = synth.make_segments(1200, seed=35, segs_per_event=600, spill=bool(consts.sim.IS_SPILL_SIM))
    if consts.sim.IS_SPILL_SIM:
        loc = seg["event_id"] % consts.sim.MAX_EVENTS_PER_FILE
        for f in ("t0", "t0_start", "t0_end"):
            seg[f] = seg[f] - loc * consts.sim.SPILL_PERIOD
    batching.swap_coordinates(seg)
    bid, order, table = batching.assign_batches(seg)
    seg, bid = seg[order], bid[order]
    ch = ChargeChain(H.response_for(kind))
    ch.upload(seg, bid)
    ch.quench_drift()
    res = {}
    try:
        lib.set_option("weights_mode", 0)      # the tail class belongs to the per-sample weights_kernel
        for tl in (0.0, 14.0):
            lib.set_option("tail_log", tl)
            ch.run(0, len(seg), want_fractions=True)
            res[tl] = ch.download()
    finally:
        _reset_current_options()
    a, b = res[0.0], res[14.0]
    assert (a["adc_list"] != 0).sum() > 100
    assert np.array_equal(a["unique_pix"], b["unique_pix"])
    assert np.array_equal(a["adc_list"] != 0, b["adc_list"] != 0)
    np.testing.assert_allclose(b["adc_list"], a["adc_list"], rtol=1e-9)
    assert np.array_equal(a["adc_ticks_list"], b["adc_ticks_list"])
    assert np.array_equal(a["adc_digit"], b["adc_digit"])
    np.testing.assert_allclose(b["current_fractions"], a["current_fractions"], rtol=1e-7, atol=1e-10)


def _load_cli():
    import importlib.util
    import os
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("sp_cli", os.path.join(repo, "larnd-sim_amd", "cli", "simulate_pixels.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    return cli


@pytest.mark.parametrize("f32_flag,f8_records", [("auto", False), ("auto", True), ("0", False), ("1", True)])
def test_cli_end_to_end(tmp_path, f32_flag, f8_records):
    """simulate_pixels CLI on a .npy segment file (edep-sim frame), shipped noise charges, seeded: the per-pixel arrays equal
    a ChargeChain run by hand with the same seed, the LArPix packets are what packets.build_packets makes of them, the
    segments come back in the edep-sim frame, and an unknown keyword is refused.
    --numba_f32: `auto` turns the f32 typing mode on for the 152-byte f4 schema and off for records with f8 fields, 0 / 1 force it;
    the by-hand chain runs in the mode the CLI must have chosen (the mode itself against the oracle:
    test_numba_f32_typing_mode_vs_oracle)."""
    from larndsim_amd import packets
    cli = _load_cli()
    H.load_cfg("module0")
    seg = synth.make_segments(40, seed=9, segs_per_event=20)
    if f8_records:          # the same values in f8 fields (what the reference's tests and the goldens use)
        seg = seg.astype(np.dtype([(n, "<f8" if np.dtype(seg.dtype[n]).kind == "f" else seg.dtype[n]) for n in seg.dtype.names]))
    f32_mode = {"auto": 0 if f8_records else 1, "0": 0, "1": 1}[f32_flag]
    assert cli.numba_f32_mode(f32_flag, seg.dtype) == f32_mode
    np.save(tmp_path / "in.npy", seg)
    resp = synth.make_response("survey")
    np.save(tmp_path / "resp.npy", resp)
    res = cli.run_simulation(str(tmp_path / "in.npy"), str(tmp_path / "out.npz"), config="module0",
                             response_file=str(tmp_path / "resp.npy"), rand_seed=5, raw_arrays=True, numba_f32=f32_flag)
    out = np.load(tmp_path / "out.npz")
    assert out["segments"].shape[0] <= 40 and (out["segments"]["n_electrons"] > 0).any()
    # same thing by hand: the CLI keeps the shipped noise charges and seeds the state table with rand_seed
    H.load_cfg("module0", noise_zero=False)
    assert consts.detector.RESET_NOISE_CHARGE > 0
    tr = cli.prepare_tracks(seg.copy())
    tr = tr[batching.select_active_volume(tr, consts.detector.TPC_BORDERS)]
    bid, order, table = batching.assign_batches(tr)
    tr, bid = np.ascontiguousarray(tr[order]), bid[order]
    ch = ChargeChain(resp)
    ch.seed_rng(5)
    try:
        lib.set_option("numba_f32", f32_mode)
        ch.upload(tr, bid); ch.quench_drift(); ch.run(0, len(tr), want_fractions=True)
    finally:
        lib.set_option("numba_f32", 0)
    ref = ch.download()
    assert np.array_equal(out["raw__unique_pix"], ref["unique_pix"]) and np.array_equal(out["raw__adc_digit"], ref["adc_digit"])
    assert np.array_equal(out["raw__adc_ticks_list"], ref["adc_ticks_list"])
    np.testing.assert_allclose(out["raw__adc_list"], ref["adc_list"], rtol=1e-12, atol=0)
    assert np.array_equal(out["raw__event_id"], np.array([t[0] for t in table])[ref["batch"]])
    # packets: one data packet per hit above the pedestal code, the association rows travel with them
    pk, assn = out["packets"], out["mc_packets_assn"]
    assert pk.dtype == packets.packets_dtype and len(pk) == len(assn) == res["n_packets"]
    n_data = int((ref["adc_digit"] > packets._digitize0()).sum())
    assert (pk["packet_type"] == 0).sum() == n_data > 0
    seg_ids = set(int(x) for x in seg["segment_id"])
    used = assn["segment_ids"][pk["packet_type"] == 0]
    assert set(int(x) for x in used[used >= 0]) <= seg_ids and (assn["fraction"][pk["packet_type"] == 0][:, 0] > 0).all()
    # stored un-swapped: x is the drift axis again
    by_id = {int(r["segment_id"]): r for r in seg}
    for r in out["segments"]:
        assert r["x"] == by_id[int(r["segment_id"])]["x"] and r["z"] == by_id[int(r["segment_id"])]["z"]
    # an unknown keyword is refused
    with pytest.raises(KeyError, match="not in supported keywords"):
        cli.run_simulation(str(tmp_path / "in.npy"), str(tmp_path / "out2.npz"), config="2x2_mpvmpr")


def test_cli_write_batch_size(tmp_path, monkeypatch):
    """sim.WRITE_BATCH_SIZE (cli/simulate_pixels.py:1207-1222): the driver gathers that many batches per export.  module0
    (self-triggered light: an export starts every event it holds with timestamp / sync / trigger packets; events cut into three
    batches each).  With one batch per export and
    with two (exports that cut events): the same data packets and association rows in the same order, fewer per-export timestamp / sync / trigger packets.
    With more batches per export than the run has, everything is exported once after the event loop: the file ends with exactly
    what the reference's hit loop (tests/packets_loop.py, pinned by the packet goldens) makes of the concatenated per-pixel
    arrays."""
    from larndsim_amd import packets
    from packets_loop import build_packets_loop
    cli = _load_cli()
    H.load_cfg("module0")
    assert consts.light.LIGHT_TRIG_MODE == 0 and not consts.sim.IS_SPILL_SIM
    seg = synth.make_segments(200, seed=12, segs_per_event=50, max_track_len=6.0)
    np.save(tmp_path / "in.npy", seg)
    resp = synth.make_response("survey", response_sampling=consts.detector.RESPONSE_SAMPLING)
    np.save(tmp_path / "resp.npy", resp)
    real_load = consts.load_snapshot

    def run(k, name):
        def load(snap):
            r = real_load(snap)
            consts.sim.WRITE_BATCH_SIZE = k
            consts.sim.BATCH_SIZE = 20           # an event's 50 segments in three batches: exports can cut an event
            return r
        monkeypatch.setattr(consts, "load_snapshot", load)
        res = cli.run_simulation(str(tmp_path / "in.npy"), str(tmp_path / name), config="module0",
                                 response_file=str(tmp_path / "resp.npy"), rand_seed=5, raw_arrays=True,
                                 chunk_segments=60)       # four chain launches: exports gather batches of different launches
        monkeypatch.setattr(consts, "load_snapshot", real_load)
        return res, np.load(tmp_path / name)

    (_, o1), (_, o3), (_, oall) = run(1, "o1.npz"), run(2, "o2.npz"), run(1000, "oall.npz")
    n_batches = len(np.unique(o1["raw__batch"]))
    assert n_batches >= 6
    for k in ("raw__unique_pix", "raw__adc_digit", "raw__adc_ticks_list", "raw__batch"):
        assert np.array_equal(o1[k], o3[k]) and np.array_equal(o1[k], oall[k])
    d1, d3, da = (o["packets"]["packet_type"] == 0 for o in (o1, o3, oall))
    assert d1.sum() == d3.sum() == da.sum() > 100
    for o, d in ((o3, d3), (oall, da)):
        assert o["packets"][d].tobytes() == o1["packets"][d1].tobytes()
        assert o["mc_packets_assn"][d].tobytes() == o1["mc_packets_assn"][d1].tobytes()
    assert len(o1["packets"]) > len(o3["packets"]) > len(oall["packets"])
    # one export of everything: rebuild its arguments from the raw arrays the way the driver does
    H.load_cfg("module0", noise_zero=False)
    consts.sim.BATCH_SIZE = 20
    tr = cli.prepare_tracks(seg.copy())
    tr = tr[batching.select_active_volume(tr, consts.detector.TPC_BORDERS)]
    bid, order, table = batching.assign_batches(tr)
    tr, bid = np.ascontiguousarray(tr[order]), bid[order]
    rb, tpm = oall["raw__batch"], oall["raw__track_pixel_map"]
    first_seg = np.searchsorted(bid[:int((bid >= 0).sum())], rb, side="left")     # a batch's rows count its segments from its first one
    seg_of = np.where(tpm >= 0, first_seg[:, None] + np.maximum(tpm, 0), 0)
    track_ids = np.where(tpm >= 0, tr["segment_id"].astype(np.int64)[seg_of], -1)
    traj_field = "file_traj_id" if "file_traj_id" in tr.dtype.names else "traj_id"
    traj_ids = np.where(tpm >= 0, tr[traj_field].astype(np.int64)[seg_of], -1)
    events = np.array([t[0] for t in table])[rb]
    ev_ids = np.repeat(events[:, None], oall["raw__adc_digit"].shape[1], axis=1)
    uniq = np.unique(events)
    num_evids = int(tr[consts.sim.EVENT_SEPARATOR].max() % consts.sim.MAX_EVENTS_PER_FILE) + 1
    event_times = cli.gen_event_times(num_evids, np.random.default_rng(5))          # the driver's, from rand_seed
    ev_time = event_times[uniq % consts.sim.MAX_EVENTS_PER_FILE]
    pk, assn = build_packets_loop(ev_ids, oall["raw__adc_digit"], oall["raw__adc_ticks_list"], oall["raw__unique_pix"],
                                  oall["raw__current_fractions"], track_ids, traj_ids, ev_time,
                                  light_trigger_times=np.zeros(len(uniq)), light_trigger_event_id=uniq,
                                  light_trigger_modules=np.ones(len(uniq)))
    assert len(pk) > da.sum()
    assert oall["packets"][-len(pk):].tobytes() == pk.tobytes()
    assert oall["mc_packets_assn"][-len(pk):].tobytes() == assn.tobytes()


def test_cli_write_batch_size_counts_batches_without_hits(tmp_path, monkeypatch):
    """sim.WRITE_BATCH_SIZE counts every simulated batch (cli/simulate_pixels.py:1207-1214), also one whose pixels hold
    no hit.  The compact download used to return no row for such a batch, so the default driver grouped its exports differently
    from its own --raw_arrays form.  An event in the middle of the run deposits almost nothing (several hit-less batches): both
    download forms write the same packets and association rows, byte for byte, for two batches per export."""
    cli = _load_cli()
    H.load_cfg("module0")
    seg = synth.make_segments(200, seed=12, segs_per_event=50, max_track_len=6.0)
    quiet = seg["event_id"] == np.unique(seg["event_id"])[1]
    seg["dEdx"][quiet] *= 1e-5
    seg["dE"][quiet] *= 1e-5
    np.save(tmp_path / "in.npy", seg)
    resp = synth.make_response("survey", response_sampling=consts.detector.RESPONSE_SAMPLING)
    np.save(tmp_path / "resp.npy", resp)
    real_load = consts.load_snapshot

    def load(snap):
        r = real_load(snap)
        consts.sim.WRITE_BATCH_SIZE = 2
        consts.sim.BATCH_SIZE = 20
        return r
    outs = {}
    for name, raw in (("compact", False), ("dense", True)):
        monkeypatch.setattr(consts, "load_snapshot", load)
        cli.run_simulation(str(tmp_path / "in.npy"), str(tmp_path / f"{name}.npz"), config="module0",
                           response_file=str(tmp_path / "resp.npy"), rand_seed=5, raw_arrays=raw, chunk_segments=60)
        monkeypatch.setattr(consts, "load_snapshot", real_load)
        outs[name] = np.load(tmp_path / f"{name}.npz")
    d = outs["dense"]
    # the quiet event's batches really are there, and really hold no hit
    ped = float(__import__("larndsim_amd.packets", fromlist=["_digitize0"])._digitize0())
    hits_of_batch = {int(b): int((d["raw__adc_digit"][d["raw__batch"] == b] > ped).sum()) for b in np.unique(d["raw__batch"])}
    assert sum(1 for v in hits_of_batch.values() if v == 0) >= 2 and sum(1 for v in hits_of_batch.values() if v > 0) >= 4
    assert (d["packets"]["packet_type"] == 0).sum() > 50
    assert outs["compact"]["packets"].tobytes() == d["packets"].tobytes()
    assert outs["compact"]["mc_packets_assn"].tobytes() == d["mc_packets_assn"].tobytes()


def test_cli_module_variation(tmp_path):
    """--config 2x2: the reference's module-variation keyword (per-module constants: module 3 has the 3.88 mm pixel layout,
    all four 50 ns response sampling).  The driver's module loop (cli/simulate_pixels.py:676-715): every module simulated
    with its own constants on its own segments.  Checked: each module's per-pixel arrays equal a ChargeChain run by hand
    under that module's snapshot; data packets sit on the module's io groups; the light datasets come per module and the
    per-module waveforms are merged side by side; the segments of all modules come back once."""
    from larndsim_amd import packets
    cli = _load_cli()
    consts.load_snapshot("2x2_mod1")
    seg = synth.make_segments(360, seed=21, segs_per_event=120, spill=True, max_track_len=4.0)   # short tracks: every module hit
    np.save(tmp_path / "in.npy", seg)
    luts = []
    for i, sd in enumerate((3, 4)):
        np.savez(tmp_path / f"lut{i}.npz", arr=synth.make_lut((14, 26, 8), 48, 40, sd))
        luts.append(str(tmp_path / f"lut{i}.npz"))
    res = cli.run_simulation(str(tmp_path / "in.npy"), str(tmp_path / "out.npz"), config="2x2", rand_seed=5, raw_arrays=True,
                             light_lut_filename=luts)
    out = np.load(tmp_path / "out.npz")
    assert res["n_segments"] == out["segments"].shape[0] > 150 and res["n_hits"] > 50
    assert len(np.unique(out["segments"]["segment_id"])) == out["segments"].shape[0]
    pk = out["packets"]
    data = pk[pk["packet_type"] == 0]
    n_data = 0
    tr_all = cli.prepare_tracks(seg.copy())
    for i_mod in (1, 2, 3, 4):
        consts.load_snapshot(f"2x2_mod{i_mod}")
        det = consts.detector
        if i_mod == 3:
            assert det.N_PIXELS == (160, 320) and abs(det.PIXEL_PITCH - 0.387975) < 1e-12
        act = tr_all[batching.select_active_volume(tr_all, det.TPC_BORDERS)]
        borders = det.TPC_BORDERS[(i_mod - 1) * 2: i_mod * 2]
        tr = act[batching.select_active_volume(act, borders)]
        bid, order, table = batching.assign_batches(tr, tpc_borders=borders)
        tr, bid = np.ascontiguousarray(tr[order]), bid[order]
        ch = ChargeChain(synth.make_response("survey", response_sampling=det.RESPONSE_SAMPLING))
        ch.upload(tr, bid); ch.quench_drift(); ch.run(0, len(tr), want_fractions=True)
        ref = ch.download()
        k = f"raw_mod{i_mod - 1}__"
        assert np.array_equal(out[k + "unique_pix"], ref["unique_pix"]), i_mod
        # the FEE noise draws differ (the CLI's state table has been advanced by the modules before), the pixel sets do not
        planes = ref["unique_pix"] // (det.N_PIXELS[0] * det.N_PIXELS[1])
        assert set(np.unique(planes)) <= {2 * (i_mod - 1), 2 * (i_mod - 1) + 1} and len(ref["unique_pix"]) > 10
        n_mod_data = int((out[k + "adc_digit"] > packets._digitize0()).sum())
        n_data += n_mod_data
        dat = out[f"light_dat__light_dat_module{i_mod - 1}"]
        assert dat.shape == (len(tr), 96) and (dat["n_photons_det"] > 0).any()
    assert len(data) == n_data
    io = consts.detector.MODULE_TO_IO_GROUPS
    assert set(np.unique(data["io_group"])) <= set(g for v in io.values() for g in v) and len(np.unique(data["io_group"])) >= 4
    # beam trigger mode: one waveform row per (event, module group) and module, merged to all 384 channels; one light_trig row
    # per spill
    light = consts.light
    ns = int(np.ceil((light.LIGHT_TRIG_WINDOW[1] + light.LIGHT_TRIG_WINDOW[0]) / light.LIGHT_DIGIT_SAMPLE_SPACING))
    wv = out["light_wvfm"]
    assert wv.shape[1:] == (384, ns) and wv.shape[0] == res["n_light_triggers"] // 4 >= 2 and (wv != 0).any()
    assert not any(k.startswith("light_wvfm/") or k.startswith("light_wvfm__") for k in out.files)
    assert out["light_trig"].shape[0] == len(np.unique(seg["event_id"])) and out["light_trig"]["op_channel"].shape[1] == 384
    assert (out["light_wvfm_mc_assn"]["op_channel_id"] >= 96).any()          # channel ids of modules past the first


def test_cli_overlapped_downloads_write_the_same_file(tmp_path):
    """The driver's pipelined form (a launch's rows on the copy stream while the previous launch's packets are built; chosen
    from eight launches on) against the synchronous one on the same launches: every dataset of the output is identical."""
    cli = _load_cli()
    H.load_cfg("module0")
    seg = synth.make_segments(480, seed=12, segs_per_event=40)          # 12 events
    np.save(tmp_path / "in.npy", seg)
    np.save(tmp_path / "resp.npy", synth.make_response("survey"))
    outs = []
    for name, ov in (("sync", False), ("ovl", True), ("auto", None)):
        res = cli.run_simulation(str(tmp_path / "in.npy"), str(tmp_path / f"{name}.npz"), config="module0", rand_seed=4,
                                 response_file=str(tmp_path / "resp.npy"), chunk_segments=40, raw_arrays=True,
                                 overlap_downloads=ov)
        assert res["n_batches"] >= 8 and res["n_packets"] > 100
        outs.append(dict(np.load(tmp_path / f"{name}.npz")))
    for other in outs[1:]:
        assert set(other) == set(outs[0])
        for k in outs[0]:
            a, b = outs[0][k], other[k]
            assert a.shape == b.shape and a.dtype == b.dtype, k
            if a.dtype.names:                      # field by field: padding bytes of an aligned record are not data
                for f in a.dtype.names:
                    assert np.array_equal(a[f], b[f], equal_nan=True), (k, f)
            else:
                assert np.array_equal(a, b, equal_nan=True), k
    # the default driver path downloads the compact form (hit pixels only) instead of every unique pixel's arrays: the same
    # packets, association rows and segments
    res = cli.run_simulation(str(tmp_path / "in.npy"), str(tmp_path / "compact.npz"), config="module0", rand_seed=4,
                             response_file=str(tmp_path / "resp.npy"), chunk_segments=40)
    cpt = dict(np.load(tmp_path / "compact.npz"))
    assert res["n_packets"] == len(outs[0]["packets"]) and not any(k.startswith("raw") for k in cpt)
    for k in ("packets", "mc_packets_assn", "segments"):
        for f in cpt[k].dtype.names:
            assert np.array_equal(cpt[k][f], outs[0][k][f], equal_nan=True), (k, f)


def test_cli_light_leg(tmp_path):
    """--light_lut_filename: the CLI runs the device-resident light leg and writes light_dat like the reference's driver."""
    cli = _load_cli()
    H.load_cfg("module0")
    seg = synth.make_segments(60, seed=10, segs_per_event=30)
    seg = seg[np.random.default_rng(4).permutation(len(seg))]      # events interleaved: batch order is not file order
    np.save(tmp_path / "in.npy", seg)
    lut = synth.make_lut((14, 26, 8), 48, 40, 3)
    np.savez(tmp_path / "lut.npz", arr=lut)
    res = cli.run_simulation(str(tmp_path / "in.npy"), str(tmp_path / "out.npz"), config="module0", rand_seed=1,
                             light_lut_filename=str(tmp_path / "lut.npz"), raw_arrays=True)
    out = np.load(tmp_path / "out.npz")
    dat = out["light_dat__light_dat_allmodules"]
    assert dat.shape == (out["segments"].shape[0], consts.light.N_OP_CHANNEL)
    assert (dat["n_photons_det"] > 0).any() and np.array_equal(dat["segment_id"][:, 0], out["segments"]["segment_id"])
    # `segments` and `light_dat` rows: the active-volume subsequence of the input in FILE order, like the reference's
    # segments_to_files = tracks (cli/simulate_pixels.py:1230-1234, 759-760) -- not the batch order the device works in
    tr = cli.prepare_tracks(seg.copy())
    act = tr[batching.select_active_volume(tr, consts.detector.TPC_BORDERS)]
    assert np.array_equal(out["segments"]["segment_id"], act["segment_id"])
    assert not np.array_equal(act["segment_id"], act["segment_id"][batching.assign_batches(act)[1]])
    hand = act.copy()
    quenching.quench[1, 64](hand, consts.physics.BIRKS)
    drifting.drift[1, 64](hand)
    for f in ("n_electrons", "n_photons", "t", "long_diff"):
        assert np.array_equal(out["segments"][f], hand[f]), f
    ch = ChargeChain(synth.make_response("survey"))
    ch.upload(act.copy(), np.zeros(len(act), dtype=np.int32)); ch.quench_drift()
    mask = lut["vis"] > 0
    lut2 = lut.copy(); lut2["vis"][~mask] = lut2["vis"][mask].min()
    ch.light_incidence(lut2, n_out=consts.light.N_OP_CHANNEL)
    inc_hand, _ = ch.download_light_incidence(0, len(hand))
    assert np.array_equal(dat["n_photons_det"], inc_hand["n_photons_det"])
    inc = out["light_sample_inc"]
    assert inc.shape[0] == res["n_batches"] and inc.sum() > 0
    # the waveform chain: one light_wvfm row per trigger (threshold mode: also one light_trig row), LSB-quantised samples;
    # (event, TPC group) combinations without segments contribute the reference's "null" waveform of an empty response
    light = consts.light
    wv, trig = out["light_wvfm"], out["light_trig"]
    ns = int(np.ceil((light.LIGHT_TRIG_WINDOW[1] + light.LIGHT_TRIG_WINDOW[0]) / light.LIGHT_DIGIT_SAMPLE_SPACING))
    assert wv.shape[1:] == (light.N_OP_CHANNEL, ns) and wv.shape[0] == res["n_light_triggers"] == trig.shape[0] >= 2
    lsb = 2.0 ** (16 - light.LIGHT_NBIT)
    assert np.array_equal(wv, np.round(wv / lsb) * lsb) and (wv != 0).any()
    assert trig["op_channel"].shape == (wv.shape[0], light.N_OP_CHANNEL) and (np.diff(trig["ts_s"]) >= 0).all()
    # with a detector-noise spectrum file every sample carries noise, reproducibly for a seed
    noise = np.abs(np.random.default_rng(3).normal(0, 4000.0, (light.N_OP_CHANNEL, 65)))
    np.save(tmp_path / "noise.npy", noise)
    runs = []
    for i in range(2):
        cli.run_simulation(str(tmp_path / "in.npy"), str(tmp_path / f"outn{i}.npz"), config="module0", rand_seed=1,
                           light_lut_filename=str(tmp_path / "lut.npz"), light_det_noise_filename=str(tmp_path / "noise.npy"))
        runs.append(np.load(tmp_path / f"outn{i}.npz")["light_wvfm"])
    assert np.array_equal(runs[0], runs[1]) and runs[0].shape == wv.shape
    assert (runs[0] != 0).mean() > 0.5 and (runs[0] - wv).std() > 4 * lsb


def test_cli_pixel_threshold_and_gain_files(tmp_path):
    """--pixel_thresholds_file / --pixel_gains_file (the reference's keys / values / default .npz, cli/simulate_pixels.py:
    439-449) reach the fused chain: same result as setting the tables by hand, different from running without them, and
    the next run without the flags is back on the constants."""
    cli = _load_cli()
    H.load_cfg("module0")
    det = consts.detector
    seg = synth.make_segments(40, seed=9, segs_per_event=20)
    np.save(tmp_path / "in.npy", seg)
    resp = synth.make_response("survey")
    np.save(tmp_path / "resp.npy", resp)
    n_ids = int(det.N_PIXELS[0] * det.N_PIXELS[1] * det.TPC_BORDERS.shape[0])
    rng = np.random.default_rng(5)
    keys = rng.choice(n_ids, size=n_ids // 3, replace=False)
    thr = det.DISCRIMINATION_THRESHOLD * rng.uniform(0.4, 3.0, keys.size)
    gain = det.GAIN * consts.units.mV / consts.units.e * rng.uniform(0.5, 1.5, keys.size)
    np.savez(tmp_path / "thr.npz", keys=keys, values=thr, default=np.array([2.0 * det.DISCRIMINATION_THRESHOLD]))
    np.savez(tmp_path / "gain.npz", keys=keys, values=gain, default=np.array([det.GAIN * consts.units.mV / consts.units.e]))
    # (numba_f32 0: the chain run by hand below is in the library's default all-f64 mode; the CLI's `auto` is test_cli_end_to_end's)
    common = dict(config="module0", response_file=str(tmp_path / "resp.npy"), rand_seed=8, raw_arrays=True, numba_f32="0")
    cli.run_simulation(str(tmp_path / "in.npy"), str(tmp_path / "a.npz"), pixel_thresholds_file=str(tmp_path / "thr.npz"),
                       pixel_gains_file=str(tmp_path / "gain.npz"), **common)
    cli.run_simulation(str(tmp_path / "in.npy"), str(tmp_path / "b.npz"), **common)
    with_files = {k[5:]: v for k, v in np.load(tmp_path / "a.npz").items() if k.startswith("raw__")}
    without = {k[5:]: v for k, v in np.load(tmp_path / "b.npz").items() if k.startswith("raw__")}
    assert not np.array_equal(with_files["adc_digit"], without["adc_digit"])
    H.load_cfg("module0", noise_zero=False)
    tr = cli.prepare_tracks(seg.copy())
    tr = tr[batching.select_active_volume(tr, consts.detector.TPC_BORDERS)]
    bid, order, table = batching.assign_batches(tr)
    tr, bid = np.ascontiguousarray(tr[order]), bid[order]
    ch = ChargeChain(resp)
    ch.upload(tr, bid); ch.quench_drift()
    ch.seed_rng(8)
    ch.run(0, len(tr), want_fractions=True)
    plain = ch.download()
    try:
        ch.set_pixel_thresholds(keys, thr, 2.0 * det.DISCRIMINATION_THRESHOLD)
        ch.set_pixel_gains(keys, gain, det.GAIN * consts.units.mV / consts.units.e)
        ch.seed_rng(8)
        ch.run(0, len(tr), want_fractions=True)
        by_hand = ch.download()
    finally:
        ch.clear_pixel_tables()
    for res, ref in ((with_files, by_hand), (without, plain)):
        assert np.array_equal(res["unique_pix"], ref["unique_pix"]) and np.array_equal(res["adc_digit"], ref["adc_digit"])
        assert np.array_equal(res["adc_ticks_list"], ref["adc_ticks_list"])
        np.testing.assert_allclose(res["adc_list"], ref["adc_list"], rtol=1e-12, atol=0)


@pytest.mark.parametrize("path", list(CURRENT_PATHS))
def test_tracks_current_edge_cases_vs_oracle(path):
    """Degenerate and extreme segments: x_start == x_end (reference returns 0, detsim.py:62-69), z_start == z_end,
    a 6 cm steep segment (many slice chunks, waveform longer than one 2048-tick tile), a segment leaving the pixel
    plane (-1 gaps), and one outside every TPC."""
    H.load_cfg("module0")
    seg = synth.make_segments(6, seed=5, segs_per_event=6)
    batching.swap_coordinates(seg)
    B = consts.detector.TPC_BORDERS[0]
    zmid = 0.5 * (B[2][0] + B[2][1])
    def put(i, a, b):
        for k, ax in enumerate("xyz"):
            seg[ax + "_start"][i] = a[k]; seg[ax + "_end"][i] = b[k]
            seg[ax][i] = 0.5 * (np.float32(a[k]).astype(np.float64) + np.float32(b[k]))
        seg["dx"][i] = np.linalg.norm(np.array(b) - np.array(a)); seg["dE"][i] = 2.1 * seg["dx"][i]
    x0, y0 = B[0][0] + 20.0, B[1][0] + 50.0
    put(0, (x0, y0, zmid), (x0, y0 + 0.3, zmid + 0.2))                 # x_start == x_end
    put(1, (x0, y0, zmid), (x0 + 0.3, y0 + 0.1, zmid))                 # z_start == z_end
    put(2, (x0, y0, B[2][0] + np.sign(B[2][1] - B[2][0]) * 2.0), (x0 + 0.4, y0 + 0.3, B[2][0] + np.sign(B[2][1] - B[2][0]) * 8.0))   # 6 cm, steep
    put(3, (B[0][0] + 0.5, y0, zmid), (B[0][0] - 0.3, y0 + 0.2, zmid + 0.1))     # leaves the pixel plane
    put(4, (x0 + 500, y0, zmid), (x0 + 500.2, y0, zmid + 0.1))         # outside every TPC
    r = H.quench_drift(O, seg)
    assert r["pixel_plane"][4] == consts.detector.DEFAULT_PLANE_INDEX
    nmax = O.max_pixels(r)
    P = 3 * nmax + 6
    _, neigh, nrad, _ = O.get_pixels(r, nmax, P, 1)
    _, T = O.time_intervals(r)
    assert T > 2048                                                      # second tick tile exercised
    resp = synth.make_response("golden")
    ref = O.tracks_current(r, neigh, T, resp)
    sig, st = _tracks_current_on(path, neigh, r, resp, T)
    assert not ref[0].any() and not sig[0].any() and not sig[4].any()
    # z_start == z_end: direction[2] == 0 makes the reference's track_point divide 0/0; its waveform is NaN
    # (0 * NaN accumulates).  Recorded divergence: this build emits zeros for such a pair.
    assert np.isnan(ref[1]).any() and not sig[1].any()
    ok = [0, 2, 3, 4, 5]
    H.assert_wave_close(sig[ok], ref[ok], rtol=1e-5, atol_peak=1e-7, what="edge cases")
    assert np.abs(ref[2]).max() > 0 and np.abs(ref[3]).max() > 0
