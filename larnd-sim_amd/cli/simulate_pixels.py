#!/usr/bin/env python3
"""
simulate_pixels.py -- command-line driver keeping the reference's flag surface (cli/simulate_pixels.py:124-145, run through
`fire` there; argparse here).

input checks -> configuration resolution (larndsim_amd.config, the reference's get_config rules) -> segment preparation
(segment_id / n_photons / t0 columns, spill-time reset, x<->z swap) -> active-volume selection -> batching ->
quench + drift -> [light incidence; per batch: photon sum -> scintillation profile -> Poisson fluctuations -> SiPM response ->
triggers -> digitised waveforms with detector noise] -> charge chain with FEE noise -> LArPix packets + mc_packets_assn per
batch -> output file (HDF5 when h5py is importable, else .npz with the same dataset names) with the updated segments,
light_dat, light_trig, light_wvfm, light_wvfm_mc_assn and the truth datasets of the input passed through.

Module-to-module variation (--mod2mod_variation or a keyword with MOD2MOD_VARIATION, e.g. `2x2`): the driver's module loop --
per-module constants, pixel layout, response, light LUT, thresholds and gains; per-module light datasets merged at the end.

The per-module pointer lists (--pixel_layout_id, --response_id, --light_lut_id, --pixel_thresholds_id, --pixel_gains_id: module m
uses file ids[m] of the corresponding file list) default to the keyword's <X>_ID entries.

Not built (the flag is accepted and reported): memory logging (--save_memory).
"""
import argparse
import os
from math import ceil
import sys
import warnings
from time import time

import numpy as np
import numpy.lib.recfunctions as rfn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from larndsim_amd import batching, consts, fee, light_sim, packets, synth  # noqa: E402
from larndsim_amd import config as cfgmod  # noqa: E402
from larndsim_amd.chain import ChargeChain  # noqa: E402

SEED = int(time())
IGNORED = ("save_memory",)
ID_FLAGS = ("pixel_layout_id", "response_id", "light_lut_id", "pixel_thresholds_id", "pixel_gains_id")
TRUTH_DATASETS = ("trajectories", "vertices", "mc_hdr", "mc_stack")


def _h5py():
    try:
        import h5py
        return h5py
    except ImportError:
        return None


def load_input(path, dset="segments"):
    """(segments, {truth dataset name: array}) from HDF5 (needs h5py), .npz (same dataset names) or .npy (segments only)."""
    if path.endswith(".npy"):
        return np.load(path), {}
    if path.endswith(".npz"):
        with np.load(path) as f:
            return f[dset], {k: f[k] for k in TRUTH_DATASETS if k in f.files}
    h5py = _h5py()
    if h5py is None:
        raise RuntimeError("HDF5 input needs h5py; .npy/.npz structured arrays with the same dtype are accepted")
    with h5py.File(path, "r") as f:
        return np.array(f[dset]), {k: np.array(f[k]) for k in TRUTH_DATASETS if k in f}


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


def gen_event_times(nevents, rng):
    """fee.gen_event_times (fee.py:70-82): cumulative exponential gaps after NON_BEAM_EVENT_GAP (numpy generator here, cupy's there)."""
    d = consts.detector
    return np.cumsum(rng.exponential(scale=d.EVENT_RATE, size=int(nevents))) + d.NON_BEAM_EVENT_GAP


class _Output:
    """HDF5 when h5py is importable, else one .npz with the same dataset names (written at close)."""

    def __init__(self, filename):
        self.filename = filename
        self.h5py = _h5py() if filename.endswith((".h5", ".hdf5")) else None
        if filename.endswith((".h5", ".hdf5")) and self.h5py is None:
            raise RuntimeError("HDF5 output needs h5py; give an .npz output name instead")
        self.parts = {}

    def append_packets(self, pk, assn):
        if self.h5py is not None:
            packets.write_hdf5(self.filename, pk, assn)
        else:
            self.parts.setdefault("packets", []).append(pk)
            self.parts.setdefault("mc_packets_assn", []).append(assn)

    def append(self, name, data):
        """resizable dataset grown along axis 0 (light_trig, light_wvfm, light_wvfm_mc_assn)"""
        if data.shape[0] == 0:
            return
        if self.h5py is not None:
            with self.h5py.File(self.filename, "a") as f:
                light_sim._append(f, name, data, (None,) * data.ndim)
        else:
            self.parts.setdefault(name, []).append(data)

    def merge_module_waveforms(self, module_indices):
        """light_sim.merge_module_light_wvfm_same_trigger (light_sim.py:759-775): the per-module waveform datasets side by side
        along the channel axis as `light_wvfm`."""
        names = [f"light_wvfm/light_wvfm_mod{i}" for i in module_indices]
        if self.h5py is not None:
            with self.h5py.File(self.filename, "r") as f:
                present = all(n in f for n in names)
            if present:
                light_sim.merge_module_light_wvfm_same_trigger(self.filename, [i + 1 for i in module_indices])
            return
        if not all(n in self.parts for n in names):
            return
        parts = [np.concatenate(self.parts.pop(n)) for n in names]
        if len({p.shape[0] for p in parts}) != 1:
            raise ValueError("The number of triggers should be the same in each module with light trigger mode 1 "
                             "(light waveform).")
        self.parts["light_wvfm"] = [np.concatenate(parts, axis=1)]

    def put(self, name, data, attrs=None):
        if self.h5py is not None:
            with self.h5py.File(self.filename, "a") as f:
                ds = f.create_dataset(name, data=data)
                for k, v in (attrs or {}).items():
                    ds.attrs[k] = v
        else:
            self.parts[name.replace("/", "__")] = [data]

    def close(self, pixel_layout=None):
        if self.h5py is not None:                                   # cli/simulate_pixels.py:1299-1301
            with self.h5py.File(self.filename, "a") as f:
                if "configs" in f.keys() and pixel_layout is not None:
                    f["configs"].attrs["pixel_layout"] = pixel_layout
        if self.h5py is None:
            # stored members like np.savez (uncompressed like the HDF5 datasets they stand in for), each written from its
            # per-launch pieces as they are, checksummed on all host threads (larndsim_amd/npz_stream.py)
            from larndsim_amd.npz_stream import NpzStream
            with NpzStream(self.filename if self.filename.endswith(".npz") else self.filename + ".npz") as z:
                for k, v in self.parts.items():
                    if len(v):
                        z.write(k, v)


def _prepend_t_event(arr):
    """The dataset with an f4 ``t_event`` field in front (aligned dtype, cli/simulate_pixels.py:618-625, 632-639)."""
    if "t_event" in arr.dtype.names:
        return arr
    descr = [("t_event", "f4")] + arr.dtype.descr
    new = np.empty(arr.shape, dtype=np.dtype(descr, align=True))
    for field in descr[1:]:
        if len(field[0]) == 0:
            continue
        new[field[0]] = arr[field[0]]
    return new


def attach_event_times(truth, event_times, sim):
    """Event times into the truth datasets that are passed through (cli/simulate_pixels.py:614-642): ``vertices['t_event']``
    for non-spill simulations -- one time per distinct event id, repeated over that event's rows in file order --, then
    copied row by row into ``mc_hdr``.  ``truth`` is changed in place."""
    if "vertices" in truth and not sim.IS_SPILL_SIM:
        vertices = _prepend_t_event(truth["vertices"])
        uniq_ev, counts = np.unique(vertices[sim.EVENT_SEPARATOR], return_counts=True)
        vertices["t_event"] = np.repeat(np.take(np.asarray(event_times), uniq_ev, mode="wrap"), counts)   # (cupy.take wraps)
        truth["vertices"] = vertices
    if "mc_hdr" in truth and "vertices" in truth and "t_event" in truth["vertices"].dtype.names:
        vertices, mc_hdr = truth["vertices"], _prepend_t_event(truth["mc_hdr"])
        if len(vertices) != len(mc_hdr):
            raise ValueError("vertices and mc_hdr datasets have different number of vertices! The number should be the same.")
        mc_hdr["t_event"] = vertices["t_event"]
        truth["mc_hdr"] = mc_hdr


def _as_list(v):
    """a flag given as 'a,b,c' (argparse) or a list (fire-style call) -> list; a plain value stays"""
    if isinstance(v, str) and "," in v:
        return [x.strip() for x in v.split(",") if x.strip()]
    return v


NUMBA_F32_FIELDS = ("x_start", "x_end", "y_start", "y_end", "z_start", "z_end", "tran_diff", "long_diff", "t_start")


def numba_f32_mode(flag, dtype):
    """--numba_f32 {auto,0,1} -> 0 / 1.  auto: 1 when the fields whose f4 storage makes Numba type sub-expressions of
    tracks_current as f32 are all f4 in the input records (the edep-sim schema), 0 when they are f8 (what the reference's own
    tests and the golden vectors use).  A mixed record keeps the all-f64 arithmetic and says so."""
    v = str(flag).strip().lower()
    if v in ("0", "false", "off"):
        return 0
    if v in ("1", "true", "on"):
        return 1
    if v != "auto":
        raise ValueError(f"--numba_f32 must be auto, 0 or 1, not {flag!r}")
    kinds = {np.dtype(dtype[f]).itemsize for f in NUMBA_F32_FIELDS if f in dtype.names and np.dtype(dtype[f]).kind == "f"}
    if kinds == {4}:
        return 1
    if kinds - {8}:
        print("[simulate_pixels] the segment records mix f4 and f8 hot-path fields: all-f64 arithmetic is used (--numba_f32 0)")
    return 0


def run_simulation(input_filename, output_filename, config="module0", mod2mod_variation=None, pixel_layout=None,
                   detector_properties=None, simulation_properties=None, response_file=None, light_simulated=None,
                   light_lut_filename=None, light_det_noise_filename=None, bad_channels=None, n_events=None,
                   pixel_thresholds_file=None, pixel_gains_file=None, rand_seed=None, config_root=None,
                   tracks_current_mc=False, chunk_segments=50000, raw_arrays=False, overlap_downloads=None,
                   pixel_layout_id=None, response_id=None, light_lut_id=None, pixel_thresholds_id=None, pixel_gains_id=None,
                   numba_f32="auto", **ignored):
    if not os.path.exists(input_filename):
        raise Exception(f"Input file {input_filename} does not exist.")
    if os.path.exists(output_filename):
        raise Exception(f"Output file {output_filename} already exists.")
    for k, v in ignored.items():
        if v is not None:
            print(f"[simulate_pixels] --{k} concerns a stage that is not built and is ignored")
    from larndsim_amd import lib

    # ---- configuration (cli/simulate_pixels.py:269-384) ------------------------------------------------------------------
    cfg = cfgmod.get_config(config, config_root)
    snapshot = cfg.get("SNAPSHOT")                                  # built-in keyword: numbers-only constants, no YAML files
    pixel_layout, response_file, light_lut_filename, pixel_thresholds_file, pixel_gains_file = (
        _as_list(v) for v in (pixel_layout, response_file, light_lut_filename, pixel_thresholds_file, pixel_gains_file))
    explicit_files = bool(detector_properties or pixel_layout or simulation_properties)
    if snapshot is None or explicit_files:
        pixel_layout = pixel_layout or cfg.get("PIXEL_LAYOUT")
        detector_properties = detector_properties or cfg.get("DET_PROPERTIES")
        simulation_properties = simulation_properties or cfg.get("SIM_PROPERTIES")
        if not (detector_properties and pixel_layout and simulation_properties) or snapshot is not None and not all(
                os.path.isfile(f) for f in [detector_properties, simulation_properties] + list(np.atleast_1d(pixel_layout))):
            raise AssertionError("pixel_layout, detector_properties and simulation_properties (files) must all be specified")
        snapshot = None
        n_modules = len(consts.get_n_modules(detector_properties))
    else:
        n_modules = len(snapshot) if isinstance(snapshot, list) else 1
    if response_file is None and snapshot is None:
        response_file = cfg.get("RESPONSE")
    if pixel_thresholds_file is None and cfg.get("PIXEL_THRESHOLDS_FILE"):   # :280-285
        pixel_thresholds_file = _as_list(cfg["PIXEL_THRESHOLDS_FILE"])
        if pixel_thresholds_id is None:
            pixel_thresholds_id = cfg.get("PIXEL_THRESHOLDS_ID")
    if pixel_gains_file is None and cfg.get("PIXEL_GAINS_FILE"):             # :286-291
        pixel_gains_file = _as_list(cfg["PIXEL_GAINS_FILE"])
        if pixel_gains_id is None:
            pixel_gains_id = cfg.get("PIXEL_GAINS_ID")
    if light_simulated is None:
        light_simulated = bool(cfg.get("LIGHT_SIMULATED", True))
    if light_simulated and light_lut_filename is None:
        light_lut_filename = cfg.get("LIGHT_LUT")
    m2m = cfgmod.module_variation_active(cfg, n_modules, mod2mod_variation,
                                         snapshot if snapshot is not None else pixel_layout, response_file, light_lut_filename)
    if m2m:                                                         # one entry per module (:374-384)
        # The reference reads the *_id flags (:272-290) and then looks the pointer lists up in the keyword's entry only
        # (load_mod2mod_variation_properties(cfg, ..), :106-122), so a list given on its command line never takes effect.
        # Here a given list is the pointer list -- what the flag is documented to be (:147-160) --, the keyword's otherwise.
        ids = {k: cfgmod.id_list(v) for k, v in (("PIXEL_LAYOUT_ID", pixel_layout_id), ("RESPONSE_ID", response_id),
                                                   ("LIGHT_LUT_ID", light_lut_id), ("PIXEL_THRESHOLD_ID", pixel_thresholds_id),
                                                   ("PIXEL_GAIN_ID", pixel_gains_id))}
        if snapshot is None:
            pixel_layout = cfgmod.module_files(cfg, pixel_layout, "PIXEL_LAYOUT_ID", n_modules, "pixel layout", ids["PIXEL_LAYOUT_ID"])
        response_file = cfgmod.module_files(cfg, response_file, "RESPONSE_ID", n_modules, "response files", ids["RESPONSE_ID"])
        pixel_thresholds_file = cfgmod.module_files(cfg, pixel_thresholds_file, "PIXEL_THRESHOLD_ID", n_modules,
                                                    "pixel threshold files", ids["PIXEL_THRESHOLD_ID"])
        pixel_gains_file = cfgmod.module_files(cfg, pixel_gains_file, "PIXEL_GAIN_ID", n_modules, "pixel gain files",
                                               ids["PIXEL_GAIN_ID"])
        light_lut_filename = cfgmod.module_files(cfg, light_lut_filename, "LIGHT_LUT_ID", n_modules, "light LUT", ids["LIGHT_LUT_ID"])
    else:
        if isinstance(snapshot, list):
            snapshot = snapshot[0]
        pixel_layout = cfgmod.single_file(pixel_layout, "pixel layout file")
        response_file = cfgmod.single_file(response_file, "response file")
        pixel_thresholds_file = cfgmod.single_file(pixel_thresholds_file, "pixel threshold file")
        pixel_gains_file = cfgmod.single_file(pixel_gains_file, "pixel gain file")
        light_lut_filename = cfgmod.single_file(light_lut_filename, "light lookup table")

    def load_constants(i_mod):
        """module i_mod's constants (i_mod = -1: the one configuration of all modules), :452-464, 678-682"""
        if snapshot is not None:
            consts.load_snapshot(snapshot[i_mod - 1] if m2m else snapshot)
        else:
            consts.load_properties(detector_properties, pixel_layout, simulation_properties, i_module=i_mod if m2m else -1)
        consts.sim.MOD2MOD_VARIATION = bool(m2m)

    def per_module(value, i_mod):
        return value[i_mod - 1] if (m2m and isinstance(value, (list, tuple))) else value

    load_constants(1 if m2m else -1)                               # geometry: any module's pixel layout will do (:657-661)
    det, sim, light = consts.detector, consts.sim, consts.light
    light_simulated = bool(light_simulated) and bool(light.LIGHT_SIMULATED) and light.N_OP_CHANNEL > 0
    if light_simulated and not all(f and os.path.isfile(f) for f in np.atleast_1d(light_lut_filename if light_lut_filename is not None else "")):
        print("light_lut_filename is not provided (required if light_simulated is True): light is not simulated")
        light_simulated = False
    if not rand_seed:
        rand_seed = SEED
    print("Random seed:", rand_seed)
    bad_list = None
    if bad_channels:
        import yaml
        with open(bad_channels) as f:
            bad_list = yaml.safe_load(f)

    # ---- input (:476-587) ---------------------------------------------------------------------------------------------------------
    tracks, truth = load_input(input_filename, sim.TRACKS_DSET_NAME)
    if tracks.size == 0:
        print("Empty input dataset, exiting")
        return None
    if n_events:
        max_ev = np.unique(tracks[sim.EVENT_SEPARATOR])[n_events - 1]
        tracks = tracks[tracks[sim.EVENT_SEPARATOR] <= max_ev]
        truth = {k: v[v[sim.EVENT_SEPARATOR] <= max_ev] for k, v in truth.items()}
    tracks = prepare_tracks(tracks)
    num_evids = int(tracks[sim.EVENT_SEPARATOR].max() % sim.MAX_EVENTS_PER_FILE) + 1
    host_rng = np.random.default_rng(rand_seed)
    event_times = (np.arange(num_evids) * sim.SPILL_PERIOD if sim.IS_SPILL_SIM else gen_event_times(num_evids, host_rng))
    attach_event_times(truth, event_times, sim)                                             # :614-642
    all_tracks = tracks[batching.select_active_volume(tracks, det.TPC_BORDERS)]            # :664-668
    all_events = np.unique(all_tracks[sim.EVENT_SEPARATOR])
    traj_field = "file_traj_id" if "file_traj_id" in all_tracks.dtype.names else "traj_id"
    mod_ids = list(det.MOD_IDS) if m2m else [-1]

    out = _Output(output_filename)
    totals = dict(n_segments=0, n_batches=0, n_hits=0, n_packets=0, n_light_triggers=0)
    kept_tracks, light_dat = [], {}
    rng_seeded = False
    lib.set_option("mc_current", 1 if tracks_current_mc else 0)
    # Numba types f32 (op) f32 as f32: with the real 152-byte schema (f4 coordinates, widths, times) a few sub-expressions of
    # tracks_current run in single precision in the reference (detsim.py:74-79,116-118,141,387); with f8 fields everything is f64.
    # "auto" follows the record's dtype.  The restatement of Numba's typing is UNPINNED (no Numba here to check it against:
    # INTEGRATION.md); it differs from all-f64 by up to 5e-5 of a waveform's peak.
    f32_mode = numba_f32_mode(numba_f32, all_tracks.dtype)
    print("Numba f32 typing mode:", "on" if f32_mode else "off", f"(--numba_f32 {numba_f32})")
    lib.set_option("numba_f32", f32_mode)
    try:
        for i_mod in mod_ids:                                       # convention: module ids count from 1 (:676-715)
            if m2m:
                print(f"Simulating module {i_mod - 1}")
                load_constants(i_mod)
                det, sim, light = consts.detector, consts.sim, consts.light
                det_borders = det.TPC_BORDERS[(i_mod - 1) * 2: i_mod * 2]
                tracks = all_tracks[batching.select_active_volume(all_tracks, det_borders)]
            else:
                det_borders = det.TPC_BORDERS
                tracks = all_tracks
            rf = per_module(response_file, i_mod)
            if rf and os.path.isfile(rf):
                response = np.load(rf)
            else:
                warnings.warn(f"response file {rf!r} not available (the reference checkout ships none): using the "
                              f"synthetic survey response table")
                response = synth.make_response("survey", response_sampling=det.RESPONSE_SAMPLING)
            chain = ChargeChain(response)
            chain.clear_pixel_tables()
            if not rng_seeded:
                chain.seed_rng(rand_seed)                           # create_xoroshiro128p_states(1024*256, seed) (:396), once
                rng_seeded = True
            thr_file, gain_file = per_module(pixel_thresholds_file, i_mod), per_module(pixel_gains_file, i_mod)
            if thr_file is not None:                                # :439-443, 698-706, 1079-1084
                print("Pixel thresholds file:", thr_file)
                chain.set_pixel_thresholds(*fee.load_pixel_table(thr_file))
            if gain_file is not None:                               # :445-449, 1097-1100
                print("Pixel gains file:", gain_file)
                chain.set_pixel_gains(*fee.load_pixel_table(gain_file))
            res = _simulate_module(chain, out, i_mod, m2m, tracks, all_events, det_borders, event_times, rand_seed, traj_field,
                                   per_module(light_lut_filename, i_mod) if light_simulated else None, light_det_noise_filename,
                                   cfg, len(mod_ids), bad_list, chunk_segments, raw_arrays, overlap_downloads)
            for k in totals:
                totals[k] += res[k]
            kept_tracks.append(res["tracks"])
            if res["light_dat"] is not None:
                light_dat[i_mod] = res["light_dat"]
        # ---- end of file (:1226-1297) --------------------------------------------------------------------------------------------------
        out_tracks = np.concatenate(kept_tracks) if len(kept_tracks) > 1 else kept_tracks[0].copy()
        if sim.IS_SPILL_SIM:
            ev = out_tracks[sim.EVENT_SEPARATOR]
            local = ev - (ev // sim.MAX_EVENTS_PER_FILE) * sim.MAX_EVENTS_PER_FILE
            for f in ("t0_start", "t0_end", "t0"):
                out_tracks[f] = out_tracks[f] + local * sim.SPILL_PERIOD
        if light_simulated and light.LIGHT_TRIG_MODE == 1:          # one beam trigger per spill / event (:1252-1259)
            ev_all = out_tracks[sim.EVENT_SEPARATOR]
            lev = np.unique(ev_all - (ev_all // sim.MAX_EVENTS_PER_FILE) * sim.MAX_EVENTS_PER_FILE) if sim.IS_SPILL_SIM \
                else (truth["vertices"]["event_id"] if "vertices" in truth else np.unique(ev_all))
            lt = lev * sim.SPILL_PERIOD if sim.IS_SPILL_SIM else event_times
            out.append("light_trig", light_sim.build_light_trig(lev, np.full(len(lev), 0), np.full(len(lev), 0),
                                                                light.TPC_TO_OP_CHANNEL[:].ravel(), lt))
        if light_simulated and m2m and light.LIGHT_TRIG_MODE == 1:  # merge_module_light_wvfm_same_trigger (:759-775)
            out.merge_module_waveforms([m - 1 for m in mod_ids])
        batching.swap_coordinates(out_tracks)
        out.put(sim.TRACKS_DSET_NAME, out_tracks, attrs={"zbeam": True})
        for i_mod, dat in light_dat.items():
            out.put(f"light_dat/light_dat_module{i_mod - 1}" if m2m else "light_dat/light_dat_allmodules", dat)
        for k, v in truth.items():
            out.put(k, v)
        out.close(pixel_layout if isinstance(pixel_layout, str) else None if pixel_layout is None else list(pixel_layout))
    finally:
        lib.set_option("mc_current", 0)
        lib.set_option("numba_f32", 0)
    print(f"simulated {totals['n_segments']} segments in {totals['n_batches']} batches -> {totals['n_hits']} hits, "
          f"{totals['n_packets']} packets" + (f", {totals['n_light_triggers']} light triggers" if light_simulated else ""))
    print("Output saved in:", output_filename)
    return totals


def _simulate_module(chain, out, i_mod, m2m, tracks, all_events, det_borders, event_times, rand_seed, traj_field, light_lut,
                     light_det_noise_filename, cfg, n_mod_ids, bad_list, chunk_segments, raw_arrays, overlap_downloads=None):
    """One pass of the driver's module loop body (cli/simulate_pixels.py:717-1232) on the device-resident chain: quench + drift,
    light leg, charge chain, packets.  ``tracks``: the module's active segments (all active segments without module
    variation); ``all_events``: event ids of every active segment (a module without segments in an event still reads out)."""
    det, sim, light = consts.detector, consts.sim, consts.light
    one = lambda v: v[0] if isinstance(v, (list, tuple)) else v          # noqa: E731
    bid, order, table = batching.assign_batches(tracks, tpc_borders=det_borders)
    tracks, bid = np.ascontiguousarray(tracks[order]), bid[order]
    nsim = int((bid >= 0).sum())
    res = dict(n_segments=nsim, n_batches=len(table), n_hits=0, n_packets=0, n_light_triggers=0, light_dat=None)
    if len(tracks):                                                 # a module may hold no segment at all: it still reads out
        chain.upload(tracks, bid)
        chain.quench_drift(consts.physics.BIRKS)
        chain.download_segments(tracks)
    # the batch-sorted copy is the device's; the file gets the module's segments in input order, like the reference's
    # `segments_to_files = tracks` (cli/simulate_pixels.py:1230-1234), with light_dat rows aligned to it (:759-760)
    file_rows = np.empty_like(tracks)
    file_rows[order] = tracks
    res["tracks"] = file_rows
    edges = np.flatnonzero(np.r_[True, bid[1:nsim] != bid[:nsim - 1], True]) if nsim else np.array([0])
    n_groups = int(np.ceil(np.asarray(det_borders).shape[0] / sim.EVENT_BATCH_SIZE))
    batch_of = {}
    for ib, (ev, grp, sub, _n) in enumerate(table):
        batch_of.setdefault((int(ev), int(grp)), []).append(ib)
    light_trig_of = {}                                              # (event, TPC group) -> trigger arrays for the packet stream
    if light_lut is not None:
        lut = np.load(light_lut)["arr"]
        mask = lut["vis"] > 0                                       # no voxel with 0 visibility (:778-780)
        lut["vis"][~mask] = lut["vis"][mask].min()
        n_light_channel = int(light.N_OP_CHANNEL / n_mod_ids) if m2m else light.N_OP_CHANNEL        # :750
        if len(tracks):
            chain.light_incidence(lut, n_out=n_light_channel)
        # in the module-variation case the channel indices of the first module stand for every module's (:1127-1131)
        op_channel = (light.TPC_TO_OP_CHANNEL[:2].ravel() if m2m else light.TPC_TO_OP_CHANNEL[:].ravel()).astype(np.int32)
        n_det = op_channel.shape[0]
        noise_file = light_det_noise_filename or one(cfg.get("LIGHT_DET_NOISE"))
        if noise_file and os.path.isfile(noise_file):
            print("Light detector noise: ", noise_file)
            light_noise = np.load(noise_file)
            if m2m:
                light_noise = light_noise[n_light_channel * (i_mod - 1): n_light_channel * i_mod]          # :789-790
        else:
            print("light_det_noise_filename is not provided (required if light_simulated is True): no detector noise")
            light_noise = None
        digit_samples = ceil((light.LIGHT_TRIG_WINDOW[1] + light.LIGHT_TRIG_WINDOW[0]) / light.LIGHT_DIGIT_SAMPLE_SPACING)
        # group thresholds of the active channels (:1183-1185)
        thr = np.repeat(np.array(light.LIGHT_TRIG_THRESHOLD)[..., np.newaxis], light.OP_CHANNEL_PER_TRIG, axis=-1)
        thr = thr.ravel()[op_channel].copy().reshape(-1, light.OP_CHANNEL_PER_TRIG)[..., 0]
        wvfm_name = f"light_wvfm/light_wvfm_mod{i_mod - 1}" if (m2m and light.LIGHT_TRIG_MODE == 1) else "light_wvfm"
        light_rows, i_trig = [], 0
        null_wvfm = None
        for ev in all_events:                                       # the reference's loop order: events, TPC groups (:864)
            ev_time = np.array([event_times[int(ev) % sim.MAX_EVENTS_PER_FILE]])
            for grp in range(n_groups):
                acc = dict(start=[], idx=[], typ=[], opc=[], wv=[], tid=[], tph=[])
                ibs = batch_of.get((int(ev), grp), [])
                if not ibs:
                    # nothing to simulate in this module group: waveforms of an empty response (:805-841, 894-899)
                    if null_wvfm is None:
                        nt0 = int((light.LIGHT_WINDOW[1] + light.LIGHT_WINDOW[0]) / light.LIGHT_TICK_SIZE)
                        zero = np.zeros((n_det, nt0), dtype=np.float32)
                        mt = sim.MAX_MC_TRUTH_IDS
                        null_wvfm = light_sim.sim_triggers(
                            None, None, zero, op_channel, np.full((n_det, nt0, mt), -1, dtype=np.int64),
                            np.zeros((n_det, nt0, mt)), np.array([0]), op_channel[None, :], digit_samples, light_noise)
                    acc["start"].append(np.full(1, 0.0)); acc["idx"].append(np.array([0]))
                    acc["typ"].append(np.full(1, light.LIGHT_TRIG_MODE)); acc["opc"].append(op_channel[None, :])
                    for k, v in zip(("wv", "tid", "tph"), null_wvfm):
                        acc[k].append(v)
                for ib in ibs:                                      # sub-batches of BATCH_SIZE segments (:902-905, 1120-1205)
                    b0, b1 = int(edges[ib]), int(edges[ib + 1])
                    n_ticks, t_start = chain.sum_light(b0, b1, op_channel,
                                                       segment_track_id=tracks["segment_id"][b0:b1].astype(np.int64))
                    if raw_arrays:
                        light_rows.append(chain.download_light(truth=False)[0])
                    chain.extend_rng(n_det * (-(-int(n_ticks) // 64)) * 64, rand_seed + int(ev) + table[ib][2] * sim.BATCH_SIZE)
                    chain.light_response(fluctuate=True)
                    t_idx, t_opc, t_type = light_sim.get_triggers(None, thr, op_channel, table[ib][2])
                    wv = light_sim.sim_triggers(None, None, None, op_channel, None, None, t_idx, t_opc, digit_samples,
                                                light_noise)
                    acc["start"].append(np.full(t_idx.shape[0], t_start)); acc["idx"].append(t_idx)
                    acc["typ"].append(t_type); acc["opc"].append(t_opc)
                    for k, v in zip(("wv", "tid", "tph"), wv):
                        acc[k].append(v)
                if not any(len(a) for a in acc["idx"]):
                    continue
                cat = {k: np.concatenate(v, axis=0) for k, v in acc.items()}
                ntr = cat["idx"].shape[0]
                lev = np.full(ntr, ev)
                if light.LIGHT_TRIG_MODE == 0:
                    out.append("light_trig", light_sim.build_light_trig(lev, cat["start"], cat["idx"], cat["opc"], ev_time))
                out.append(wvfm_name, cat["wv"])
                if sim.MAX_MC_TRUTH_IDS > 0:
                    out.append("light_wvfm_mc_assn",
                               light_sim.zero_suppress_waveform_truth(cat["tid"], cat["tph"], lev[0], i_trig, i_mod))
                i_trig += 1
                res["n_light_triggers"] += ntr
                if ibs:
                    mods = (cat["typ"] if light.LIGHT_TRIG_MODE == 1 else
                            np.array([det.TPC_TO_MODULE[int(t)] for t in light.OP_CHANNEL_TO_TPC[cat["opc"]][:, 0]]))
                    light_trig_of[(int(ev), grp)] = (cat["start"] + cat["idx"] * light.LIGHT_TICK_SIZE, lev, mods)
        if len(tracks):
            inc, _ = chain.download_light_incidence(0, len(tracks))
            inc["segment_id"] = tracks["segment_id"][:, None]
            inc_file = np.empty_like(inc)
            inc_file[order] = inc
            res["light_dat"] = inc_file
        if raw_arrays and light_rows:
            nt_max = max(r.shape[1] for r in light_rows)           # the tick count follows each batch's arrival times
            out.put("light_sample_inc" + (f"_mod{i_mod - 1}" if m2m else ""),
                    np.stack([np.pad(r, ((0, 0), (0, nt_max - r.shape[1]))) for r in light_rows]))

    # ---- sync / timestamp / trigger packets at every new event (:866-890), then the event's charge packets ------------------
    period = det.CLOCK_RESET_PERIOD * det.CLOCK_CYCLE
    sync_start = event_times[0] // period * period + period
    io_rows = np.array(list(det.MODULE_TO_IO_GROUPS.values()))
    trig_module = int(np.argwhere(io_rows == packets.get_trig_io())[0][0]) + 1
    announced = 0                                                   # events of all_events whose packets are out

    def announce_until(event):
        nonlocal sync_start, announced
        while announced < len(all_events) and all_events[announced] <= event:
            t_ev = event_times[int(all_events[announced]) % sim.MAX_EVENTS_PER_FILE]
            if t_ev - sync_start >= 0:
                sync_times = np.arange(sync_start, t_ev + 1, period)
                if len(sync_times):
                    spk = packets.build_sync_packets(np.full(sync_times.shape, period), i_mod)
                    out.append_packets(*spk)
                    res["n_packets"] += len(spk[0])
                    sync_start = sync_times[-1] + period
            if i_mod == trig_module or i_mod == -1:                 # the trigger is forwarded to one PACMAN only
                tpk = packets.build_timestamp_trigger_packets([t_ev], i_mod)
                out.append_packets(*tpk)
                res["n_packets"] += len(tpk[0])
            announced += 1

    parts = []
    seg_ids_all = tracks["segment_id"].astype(np.int64)            # (once: a copy of the tail per batch is quadratic in a spill)
    trj_ids_all = tracks[traj_field].astype(np.int64)

    # Batches per export (sim.WRITE_BATCH_SIZE, default 1; cli/simulate_pixels.py:1207-1214): the reference gathers that many
    # batches' arrays and hands fee.export_to_hdf5 their concatenation (save_results, :179-258) -- one set of timestamp / sync /
    # trigger packets per event of the export, clock-rollover state from its first row -- and exports what is left after the event
    # loop.  A batch counts when it holds a pixel the chain returns rows for.
    write_batch = max(int(getattr(sim, "WRITE_BATCH_SIZE", 1)), 1)
    pending = []            # (event, arrays of the batch, its light triggers or None)

    event_of_batch = np.array([t[0] for t in table], dtype=np.int64)
    first_seg_of_batch = np.searchsorted(bid[:nsim], np.arange(len(table)), side="left").astype(np.int64)

    def export_triggers(events_in_order):
        """light triggers embedded in the charge stream of an export (:206-222)"""
        uniq = np.unique(np.asarray(events_in_order))
        ev_time = np.array([event_times[int(e) % sim.MAX_EVENTS_PER_FILE] for e in uniq])
        if all(p[2] is None for p in pending):
            # each event triggers once at perfect t0 (:218-222)
            return ev_time, (np.zeros(len(uniq)), uniq, np.ones(len(uniq)))
        # the simulated triggers of the export's batches (:209-216); a batch without any keeps the perfect one
        trip = [p[2] if p[2] is not None else (np.zeros(1), np.array([p[0]]), np.ones(1)) for p in pending]
        return ev_time, tuple(np.concatenate([np.atleast_1d(t[k]) for t in trip]) for k in range(3))

    def flush_pending_compact():
        """the same export from the compact rows of the chain, through the native hit loop (ldsim_packets_build): no dense
        [pixel][30][50] fraction rows are rebuilt to feed an array-shaped exporter"""
        pieces = [packets.compact_to_rows(p[1][1], event_of_batch, first_seg_of_batch, seg_ids_all, trj_ids_all, rows=p[1][2:4])
                  for p in pending]
        rows = pieces[0] if len(pieces) == 1 else {k: np.concatenate([q[k] for q in pieces]) for k in pieces[0]}
        ev_time, (lt_times, lt_events, lt_mods) = export_triggers([p[0] for p in pending])
        pk, assn = packets.build_packets_compact(**rows, event_start_times=ev_time, light_trigger_times=lt_times,
                                                 light_trigger_event_id=lt_events, light_trigger_modules=lt_mods,
                                                 bad_channels=bad_list, i_mod=i_mod)
        out.append_packets(pk, assn)
        res["n_packets"] += len(pk)
        pending.clear()

    def flush_pending():
        if not pending:
            return
        if isinstance(pending[0][1][0], str):           # ("compact", download, first row, row after the last)
            return flush_pending_compact()
        events = [p[0] for p in pending]
        cat = [np.concatenate([p[1][k] for p in pending]) if len(pending) > 1 else pending[0][1][k] for k in range(7) if k != 4]
        # (the fraction rows, 12 KB per pixel: batches that follow each other in one launch's array are one slice of it)
        src = [p[3] for p in pending]
        one_run = all(a is not None and b is not None and a[0] is b[0] and a[2] == b[1] for a, b in zip(src[:-1], src[1:]))
        frac = src[0][0][src[0][1]:src[-1][2]] if (one_run and src[0] is not None) else \
            (np.concatenate([p[1][4] for p in pending]) if len(pending) > 1 else pending[0][1][4])
        cat.insert(4, frac)
        uniq = np.unique(np.asarray(events))
        ev_time = np.array([event_times[int(e) % sim.MAX_EVENTS_PER_FILE] for e in uniq])
        if all(p[2] is None for p in pending):
            # each event triggers once at perfect t0 (:218-222)
            lt_times, lt_events, lt_mods = np.zeros(len(uniq)), uniq, np.ones(len(uniq))
        else:
            # the simulated triggers of the export's batches (:209-216); a batch without any keeps the perfect one
            trip = [p[2] if p[2] is not None else (np.zeros(1), np.array([p[0]]), np.ones(1)) for p in pending]
            lt_times, lt_events, lt_mods = (np.concatenate([np.atleast_1d(t[k]) for t in trip]) for k in range(3))
        pk, assn = packets.build_packets(*cat, ev_time, light_trigger_times=lt_times, light_trigger_event_id=lt_events,
                                         light_trigger_modules=lt_mods, bad_channels=bad_list, i_mod=i_mod)
        out.append_packets(pk, assn)
        res["n_packets"] += len(pk)
        pending.clear()

    def export_chunk_compact(c):
        """packets and association rows of one chain launch from its compact download: a batch is a run of hit-pixel rows (its
        first unique pixel always among them, hits or not: kernels_compact.hip)"""
        hp = c["hit_pixels"]
        rb = hp[:, 2]
        starts = np.flatnonzero(np.r_[True, rb[1:] != rb[:-1]]) if len(rb) else np.zeros(0, dtype=np.int64)
        ends = np.r_[starts[1:], len(rb)]
        for a, b in zip(starts, ends):
            bb = int(rb[a])
            event = table[bb][0]
            announce_until(event)
            pending.append((event, ("compact", c, int(a), int(b)), light_trig_of.get((int(event), int(table[bb][1]))), None))
            if len(pending) >= write_batch:
                flush_pending()
        res["n_hits"] += len(c["hit_rows"])

    def export_chunk(r):
        """packets and association rows of one chain launch (`r`: the launch's per-pixel arrays)"""
        # the chain returns the unique pixels ordered by batch: a batch is a contiguous run (a view, not a 12 KB-per-pixel copy
        # of the backtracking array); an unordered result falls back to masks
        rb = r["batch"]
        ordered = len(rb) < 2 or bool((rb[1:] >= rb[:-1]).all())
        for bb in np.unique(rb):
            m = slice(int(np.searchsorted(rb, bb, side="left")), int(np.searchsorted(rb, bb, side="right"))) if ordered \
                else rb == bb
            lo = int(np.searchsorted(bid[:nsim], bb, side="left"))
            seg_ids, trj_ids = seg_ids_all[lo:], trj_ids_all[lo:]
            tpm = r["track_pixel_map"][m]
            digit, ticks_b, upix_b, frac_b = r["adc_digit"][m], r["adc_ticks_list"][m], r["unique_pix"][m], r["current_fractions"][m]
            # (compact form: a batch whose first unique pixel holds no hit comes with that hit-less row in front --
            # expand_compact(lead_rows=True) -- because the reference's exporter keeps its clock-rollover state in row 0 of what
            # it is handed, fee.py:164-183, 267-277)
            track_ids = np.where(tpm >= 0, seg_ids[np.maximum(tpm, 0)], -1)
            traj_ids = np.where(tpm >= 0, trj_ids[np.maximum(tpm, 0)], -1)
            event = table[int(bb)][0]
            announce_until(event)
            ev_ids = np.full(digit.shape, event)
            # light triggers embedded in the charge stream (:209-221): the simulated ones, else one perfect trigger per event
            pending.append((event, (ev_ids, digit, ticks_b, upix_b, frac_b, track_ids, traj_ids),
                            light_trig_of.get((int(event), int(table[int(bb)][1]))),
                            (r["current_fractions"], m.start, m.stop) if ordered else None))
            if len(pending) >= write_batch:
                flush_pending()
        # batches that wait for the next launch's: their rows are views of this launch's arrays, which a later download may
        # overwrite (page-locked buffers are reused) -- keep copies
        for i, p in enumerate(pending):
            if p[3] is not None:
                pending[i] = (p[0], tuple(np.array(a) for a in p[1]), p[2], None)
        res["n_hits"] += int((r["adc_list"] != 0).sum())
        if raw_arrays:
            r = {k: np.array(v) for k, v in r.items()}       # (views of page-locked buffers that later launches reuse)
            r["event_id"] = np.array([t[0] for t in table])[r["batch"]]
            parts.append(r)

    # What crosses PCIe per launch: the compact form (hit pixels, hits, per-hit fractions: ldsim_chain_compact_*), expanded on the
    # host to the dense rows of the hit pixels -- the only rows the exporter reads.  --raw_arrays wants every unique pixel's
    # arrays and takes the dense download: launch k's rows then travel on the library's copy stream (download_async) while
    # launch k - 1's packets are built (page-locking the two sets of host arrays costs ~0.5 s: from about eight launches on).
    from larndsim_amd.chain import expand_compact
    overlapped = raw_arrays and ((nsim >= 8 * chunk_segments) if overlap_downloads is None else bool(overlap_downloads))
    b = 0
    in_flight = None
    launches = [(int(b0), int(e0)) for b0, e0 in _launch_ranges(edges, nsim, chunk_segments)]
    if not raw_arrays:
        # Default path.  Launch k runs on a worker thread (ChargeChain.run_async: the C call blocks on its size read-backs but
        # releases the interpreter) while launch k - 1's packets are built here from its compact download -- host work that never
        # enters the ctx.
        prev = None
        for b0, e0 in launches:
            chain.run_async(b0, e0, want_fractions=True)
            if prev is not None:
                export_chunk_compact(prev)
            chain.wait()
            prev = chain.download_compact()
        if prev is not None:
            export_chunk_compact(prev)
        launches = []
    for b, e in launches:
        chain.run(int(b), int(e), want_fractions=True)
        if not overlapped:
            export_chunk(chain.download())
            continue
        arriving = chain.download_async()
        if in_flight is not None:
            export_chunk(in_flight)
        in_flight = arriving
    if in_flight is not None:
        chain.wait_download()
        export_chunk(in_flight)
    if len(all_events):
        announce_until(all_events[-1])
    flush_pending()                                  # (what is left after the event loop, :1219-1222)
    if raw_arrays and parts:
        for k in parts[0]:
            out.put(("raw/" if not m2m else f"raw_mod{i_mod - 1}/") + k, np.concatenate([p[k] for p in parts]))
    return res


def _launch_ranges(edges, nsim, chunk_segments):
    """segment ranges of the chain launches: whole batches, at least chunk_segments each (the last one: what is left)"""
    b = 0
    for e in edges[1:]:
        if e - b >= chunk_segments or e == nsim:
            yield b, e
            b = e


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--input_filename", required=True)
    ap.add_argument("--output_filename", required=True)
    ap.add_argument("--config", default="module0")
    ap.add_argument("--config_root", default=None, help="larnd-sim tree holding config/config.yaml and the YAML families "
                                                        "(default: $LARNDSIM_ROOT, else the built-in snapshot keywords)")
    tf = lambda s: s.lower() in ("1", "true", "yes")                # noqa: E731
    ap.add_argument("--mod2mod_variation", type=tf, default=None,
                    help="per-module pixel layouts / responses / light LUTs / thresholds / gains (comma-separated lists), "
                         "default: the keyword's MOD2MOD_VARIATION")
    ap.add_argument("--light_simulated", type=tf, default=None)
    for k in ("pixel_layout", "detector_properties", "simulation_properties", "response_file", "light_lut_filename",
              "light_det_noise_filename", "bad_channels", "pixel_thresholds_file", "pixel_gains_file", *IGNORED):
        ap.add_argument("--" + k, default=None)
    for k in ID_FLAGS:
        ap.add_argument("--" + k, default=None, help="module variation: per-module index into the corresponding file list, "
                                                      "e.g. 0,0,1,0 (default: the keyword's entry)")
    ap.add_argument("--n_events", type=int, default=None)
    ap.add_argument("--rand_seed", type=int, default=None)
    ap.add_argument("--tracks_current_mc", action="store_true",
                    help="induced currents from tracks_current_mc like the reference driver (default: tracks_current)")
    ap.add_argument("--raw_arrays", action="store_true", help="also store the per-pixel arrays (raw/...) and light_sample_inc")
    ap.add_argument("--numba_f32", default="auto", choices=["auto", "0", "1"],
                    help="evaluate the sub-expressions Numba types as float32 for f4 record fields in single precision like the "
                         "reference (1), all in double (0), or by the input records' dtype (auto, default); unpinned restatement")
    a = vars(ap.parse_args(argv))
    run_simulation(**a)


if __name__ == "__main__":
    main()
