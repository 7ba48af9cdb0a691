#!/usr/bin/env python3
"""
simulate_pixels.py -- command-line driver for the charge path, keeping the reference's flag surface
(cli/simulate_pixels.py:124-145, run through `fire` there; argparse here, fire is optional).

Covers what SURVEY §8 puts on the hot path: input checks, config resolution, segment preparation
(segment_id / n_photons / t0 columns, spill-time reset, x<->z swap), active-volume selection, batching,
quench -> drift -> [light incidence] -> charge chain, and a numbers-only output (npz, or HDF5 when h5py is
importable).  LArPix packet export, light waveforms and truth pass-through are out of scope:
flags that only concern them are accepted and ignored with a notice.
"""
import argparse
import os
import sys
import warnings

import numpy as np
import numpy.lib.recfunctions as rfn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from larndsim_amd import batching, consts, fee, synth  # noqa: E402
from larndsim_amd.chain import ChargeChain  # noqa: E402

CONFIG_SNAPSHOTS = {"module0": "module0", "2x2_no_modvar": "2x2_no_modvar", "2x2": "2x2_no_modvar", "ndlar": "ndlar"}
IGNORED = ("light_det_noise_filename", "bad_channels", "pixel_thresholds_id", "pixel_gains_id", "save_memory",
           "pixel_layout_id", "response_id", "light_lut_id")


def load_segments(path, dset="segments"):
    if path.endswith(".npy"):
        return np.load(path)
    if path.endswith(".npz"):
        return np.load(path)[dset]
    try:
        import h5py
    except ImportError as e:
        raise RuntimeError("HDF5 input needs h5py; .npy/.npz structured arrays with the same dtype are accepted") from e
    with h5py.File(path, "r") as f:
        return np.array(f[dset])


def prepare_tracks(tracks):
    """cli/simulate_pixels.py:482-494, 550-587: add missing columns, reset spill time, swap x<->z."""
    sim = consts.sim
    if "segment_id" not in tracks.dtype.names:
        tracks = rfn.append_fields(tracks, "segment_id", np.arange(tracks.shape[0], dtype="u4"), usemask=False)
    if "n_photons" not in tracks.dtype.names:
        tracks = rfn.append_fields(tracks, "n_photons", np.zeros(tracks.shape[0], dtype="f4"), usemask=False)
    if "t0" not in tracks.dtype.names:
        t0, t0s, t0e = tracks["t"].copy(), tracks["t_start"].copy(), tracks["t_end"].copy()
        tracks = rfn.append_fields(tracks, ["t0", "t0_start", "t0_end"], [t0, t0s, t0e], dtypes=["f4"] * 3,
                                   usemask=False)
        for f in ("t", "t_start", "t_end"):
            tracks[f] = 0
    if sim.IS_SPILL_SIM:
        ev = tracks[sim.EVENT_SEPARATOR]
        local = ev - (ev // sim.MAX_EVENTS_PER_FILE) * sim.MAX_EVENTS_PER_FILE
        for f in ("t0_start", "t0_end", "t0"):
            tracks[f] = tracks[f] - local * sim.SPILL_PERIOD
    return batching.swap_coordinates(tracks)


def run_simulation(input_filename, output_filename, config="module0", mod2mod_variation=None, pixel_layout=None,
                   detector_properties=None, simulation_properties=None, response_file=None, light_simulated=None,
                   light_lut_filename=None, n_events=None, rand_seed=None, chunk_segments=50000,
                   pixel_thresholds_file=None, pixel_gains_file=None, **ignored):
    if not os.path.exists(input_filename):
        raise Exception(f"Input file {input_filename} does not exist.")
    if os.path.exists(output_filename):
        raise Exception(f"Output file {output_filename} already exists.")
    for k, v in ignored.items():
        if v is not None:
            print(f"[simulate_pixels] --{k} concerns an out-of-scope stage and is ignored")
    if mod2mod_variation:
        raise NotImplementedError("mod2mod_variation is out of scope (SURVEY §8f)")
    if detector_properties and pixel_layout and simulation_properties:
        consts.load_properties(detector_properties, pixel_layout, simulation_properties)
    else:
        if config not in CONFIG_SNAPSHOTS:
            raise KeyError(f"Key {config} not in supported keywords {list(CONFIG_SNAPSHOTS)}")
        consts.load_snapshot(CONFIG_SNAPSHOTS[config])
    det, sim = consts.detector, consts.sim
    if any(getattr(det, k) for k in ("RESET_NOISE_CHARGE", "UNCORRELATED_NOISE_CHARGE", "DISCRIMINATOR_NOISE")):
        warnings.warn("FEE noise is switched off: the reference's Numba RNG stream is not reproduced")
        det.RESET_NOISE_CHARGE = det.UNCORRELATED_NOISE_CHARGE = det.DISCRIMINATOR_NOISE = 0
    if response_file:
        response = np.load(response_file)
    else:
        warnings.warn("no --response_file: using the synthetic survey response table")
        response = synth.make_response("survey")

    tracks = load_segments(input_filename, sim.TRACKS_DSET_NAME)
    if tracks.size == 0:
        print("Empty input dataset, exiting")
        return None
    if n_events:
        max_ev = np.unique(tracks[sim.EVENT_SEPARATOR])[n_events - 1]
        tracks = tracks[tracks[sim.EVENT_SEPARATOR] <= max_ev]
    tracks = prepare_tracks(tracks)
    tracks = tracks[batching.select_active_volume(tracks, det.TPC_BORDERS)]
    bid, order, table = batching.assign_batches(tracks)
    tracks, bid = np.ascontiguousarray(tracks[order]), bid[order]
    nsim = int((bid >= 0).sum())

    chain = ChargeChain(response)
    chain.clear_pixel_tables()
    if pixel_thresholds_file is not None:                          # cli/simulate_pixels.py:439-443, 1079-1084
        print("Pixel thresholds file:", pixel_thresholds_file)
        chain.set_pixel_thresholds(*fee.load_pixel_table(pixel_thresholds_file))
    if pixel_gains_file is not None:                               # :445-449, 1097-1100
        print("Pixel gains file:", pixel_gains_file)
        chain.set_pixel_gains(*fee.load_pixel_table(pixel_gains_file))
    chain.upload(tracks, bid)
    chain.quench_drift(consts.physics.BIRKS)
    chain.download_segments(tracks)
    parts = []
    edges = np.flatnonzero(np.r_[True, bid[1:nsim] != bid[:nsim - 1], True]) if nsim else np.array([0])
    b = 0
    for e in edges[1:]:
        if e - b >= chunk_segments or e == nsim:
            chain.run(int(b), int(e), want_fractions=True)
            parts.append(chain.download())
            b = e
    res = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]} if parts else {}
    if res:
        ev = np.array([t[0] for t in table])
        res["event_id"] = ev[res["batch"]]
    out_tracks = batching.swap_coordinates(tracks.copy())          # stored un-swapped like the reference (:1275)
    if output_filename.endswith((".h5", ".hdf5")):
        import h5py
        with h5py.File(output_filename, "w") as f:
            f.create_dataset("segments", data=out_tracks)
            for k, v in res.items():
                f.create_dataset(k, data=v)
    else:
        np.savez_compressed(output_filename, segments=out_tracks, **res)
    print(f"simulated {nsim} segments in {len(table)} batches -> {len(res.get('unique_pix', []))} pixel rows, "
          f"{int((res.get('adc_list', np.zeros(0)) != 0).sum())} hits")
    return res


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--input_filename", required=True)
    ap.add_argument("--output_filename", required=True)
    ap.add_argument("--config", default="module0")
    ap.add_argument("--mod2mod_variation", type=lambda s: s.lower() in ("1", "true"), default=None)
    for k in ("pixel_layout", "detector_properties", "simulation_properties", "response_file", "light_lut_filename",
              "pixel_thresholds_file", "pixel_gains_file", *IGNORED):
        ap.add_argument("--" + k, default=None)
    ap.add_argument("--light_simulated", default=None)
    ap.add_argument("--n_events", type=int, default=None)
    ap.add_argument("--rand_seed", type=int, default=None)
    a = vars(ap.parse_args(argv))
    run_simulation(**a)


if __name__ == "__main__":
    main()
