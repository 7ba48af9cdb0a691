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

Not built (a flag that only concerns them is accepted and reported): module-to-module variation, bad-channel lists by id,
memory logging.
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
IGNORED = ("pixel_thresholds_id", "pixel_gains_id", "save_memory", "pixel_layout_id",
           "response_id", "light_lut_id")
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

    def put(self, name, data, attrs=None):
        if self.h5py is not None:
            with self.h5py.File(self.filename, "a") as f:
                ds = f.create_dataset(name, data=data)
                for k, v in (attrs or {}).items():
                    ds.attrs[k] = v
        else:
            self.parts[name.replace("/", "__")] = [data]

    def close(self):
        if self.h5py is None:
            out = {k: (np.concatenate(v) if len(v) > 1 else v[0]) for k, v in self.parts.items() if len(v)}
            np.savez_compressed(self.filename, **out)


def run_simulation(input_filename, output_filename, config="module0", mod2mod_variation=None, pixel_layout=None,
                   detector_properties=None, simulation_properties=None, response_file=None, light_simulated=None,
                   light_lut_filename=None, light_det_noise_filename=None, bad_channels=None, n_events=None, pixel_thresholds_file=None,
                   pixel_gains_file=None, rand_seed=None, config_root=None, tracks_current_mc=False, chunk_segments=50000,
                   raw_arrays=False, **ignored):
    if not os.path.exists(input_filename):
        raise Exception(f"Input file {input_filename} does not exist.")
    if os.path.exists(output_filename):
        raise Exception(f"Output file {output_filename} already exists.")
    for k, v in ignored.items():
        if v is not None:
            print(f"[simulate_pixels] --{k} concerns a stage that is not built and is ignored")

    # ---- configuration (cli/simulate_pixels.py:269-384) ------------------------------------------------------------------
    cfg = cfgmod.get_config(config, config_root)
    cfgmod.check_single_configuration(config, cfg, mod2mod_variation)
    one = lambda v: v[0] if isinstance(v, (list, tuple)) else v          # noqa: E731  single-configuration lists
    pixel_layout = pixel_layout or (one(cfg.get("PIXEL_LAYOUT")) if "SNAPSHOT" not in cfg else None)
    detector_properties = detector_properties or (cfg.get("DET_PROPERTIES") if "SNAPSHOT" not in cfg else None)
    simulation_properties = simulation_properties or (cfg.get("SIM_PROPERTIES") if "SNAPSHOT" not in cfg else None)
    if detector_properties and pixel_layout and simulation_properties:
        consts.load_properties(detector_properties, pixel_layout, simulation_properties)
    elif "SNAPSHOT" in cfg and not (detector_properties or pixel_layout or simulation_properties):
        consts.load_snapshot(cfg["SNAPSHOT"])
    else:
        raise AssertionError("pixel_layout, detector_properties and simulation_properties (files) must all be specified")
    det, sim, light = consts.detector, consts.sim, consts.light
    if response_file is None and "SNAPSHOT" not in cfg:
        response_file = one(cfg.get("RESPONSE"))
    if response_file and os.path.isfile(response_file):
        response = np.load(response_file)
    else:
        warnings.warn(f"response file {response_file!r} not available (the reference checkout ships none): using the "
                      f"synthetic survey response table")
        response = synth.make_response("survey", response_sampling=det.RESPONSE_SAMPLING)
    if light_simulated is None:
        light_simulated = bool(cfg.get("LIGHT_SIMULATED", True))
    light_simulated = bool(light_simulated) and bool(light.LIGHT_SIMULATED) and light.N_OP_CHANNEL > 0
    lut = None
    if light_simulated:
        light_lut_filename = light_lut_filename or one(cfg.get("LIGHT_LUT"))
        if light_lut_filename and os.path.isfile(light_lut_filename):
            lut = np.load(light_lut_filename)["arr"]
            mask = lut["vis"] > 0                                   # no voxel with 0 visibility (cli/simulate_pixels.py:770-771)
            lut["vis"][~mask] = lut["vis"][mask].min()
        else:
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
    tracks = tracks[batching.select_active_volume(tracks, det.TPC_BORDERS)]
    bid, order, table = batching.assign_batches(tracks)
    tracks, bid = np.ascontiguousarray(tracks[order]), bid[order]
    nsim = int((bid >= 0).sum())
    traj_field = "file_traj_id" if "file_traj_id" in tracks.dtype.names else "traj_id"

    # ---- device-resident simulation ---------------------------------------------------------------------------------------------
    chain = ChargeChain(response)
    chain.clear_pixel_tables()
    chain.seed_rng(rand_seed)                                       # create_xoroshiro128p_states(1024*256, seed) (:396)
    if pixel_thresholds_file is not None:                          # :439-443, 1079-1084
        print("Pixel thresholds file:", pixel_thresholds_file)
        chain.set_pixel_thresholds(*fee.load_pixel_table(pixel_thresholds_file))
    if pixel_gains_file is not None:                               # :445-449, 1097-1100
        print("Pixel gains file:", pixel_gains_file)
        chain.set_pixel_gains(*fee.load_pixel_table(pixel_gains_file))
    from larndsim_amd import lib
    lib.set_option("mc_current", 1 if tracks_current_mc else 0)
    out = _Output(output_filename)
    try:
        chain.upload(tracks, bid)
        chain.quench_drift(consts.physics.BIRKS)
        chain.download_segments(tracks)
        edges = np.flatnonzero(np.r_[True, bid[1:nsim] != bid[:nsim - 1], True]) if nsim else np.array([0])
        light_trig_of = {}                                           # (event, TPC group) -> trigger arrays for the packet stream
        n_light_trig = 0
        if light_simulated:
            chain.light_incidence(lut)
            op_channel = light.TPC_TO_OP_CHANNEL[:].ravel().astype(np.int32)
            n_det = op_channel.shape[0]
            light_det_noise_filename = light_det_noise_filename or one(cfg.get("LIGHT_DET_NOISE"))
            if light_det_noise_filename and os.path.isfile(light_det_noise_filename):
                print("Light detector noise: ", light_det_noise_filename)
                light_noise = np.load(light_det_noise_filename)
            else:
                print("light_det_noise_filename is not provided (required if light_simulated is True): no detector noise")
                light_noise = None
            digit_samples = ceil((light.LIGHT_TRIG_WINDOW[1] + light.LIGHT_TRIG_WINDOW[0]) / light.LIGHT_DIGIT_SAMPLE_SPACING)
            # group thresholds of the active channels (:1183-1185)
            thr = np.repeat(np.array(light.LIGHT_TRIG_THRESHOLD)[..., np.newaxis], light.OP_CHANNEL_PER_TRIG, axis=-1)
            thr = thr.ravel()[op_channel].copy().reshape(-1, light.OP_CHANNEL_PER_TRIG)[..., 0]
            n_groups = int(np.ceil(det.TPC_BORDERS.shape[0] / sim.EVENT_BATCH_SIZE))
            batch_of = {}
            for ib, (ev, grp, sub, _n) in enumerate(table):
                batch_of.setdefault((int(ev), int(grp)), []).append(ib)
            light_rows, i_trig = [], 0
            null_wvfm = None
            for ev in np.unique(tracks[sim.EVENT_SEPARATOR]):       # the reference's loop order: events, TPC groups (:864)
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
                    for ib in ibs:                                  # sub-batches of BATCH_SIZE segments (:902-905, 1120-1205)
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
                    out.append("light_wvfm", cat["wv"])
                    if sim.MAX_MC_TRUTH_IDS > 0:
                        out.append("light_wvfm_mc_assn",
                                   light_sim.zero_suppress_waveform_truth(cat["tid"], cat["tph"], lev[0], i_trig, -1))
                    i_trig += 1
                    n_light_trig += ntr
                    if ibs:
                        mods = (cat["typ"] if light.LIGHT_TRIG_MODE == 1 else
                                np.array([det.TPC_TO_MODULE[int(t)] for t in light.OP_CHANNEL_TO_TPC[cat["opc"]][:, 0]]))
                        light_trig_of[(int(ev), grp)] = (cat["start"] + cat["idx"] * light.LIGHT_TICK_SIZE, lev, mods)
            inc, _ = chain.download_light_incidence(0, len(tracks))
            inc["segment_id"] = tracks["segment_id"][:, None]
            out.put("light_dat/light_dat_allmodules", inc)
            if raw_arrays and light_rows:
                nt_max = max(r.shape[1] for r in light_rows)       # the tick count follows each batch's arrival times
                out.put("light_sample_inc", np.stack([np.pad(r, ((0, 0), (0, nt_max - r.shape[1]))) for r in light_rows]))
        parts, n_hits, n_packets = [], 0, 0
        b = 0
        for e in edges[1:]:
            if not (e - b >= chunk_segments or e == nsim):
                continue
            st = chain.run(int(b), int(e), want_fractions=True)
            res = chain.download()
            # one export per batch, like save_results with WRITE_BATCH_SIZE = 1 (:179-258, 1207-1214)
            for bb in np.unique(res["batch"]):
                m = res["batch"] == bb
                lo = int(np.searchsorted(bid[:nsim], bb, side="left"))
                seg_ids = tracks["segment_id"][lo:].astype(np.int64)
                trj_ids = tracks[traj_field][lo:].astype(np.int64)
                tpm = res["track_pixel_map"][m]
                track_ids = np.where(tpm >= 0, seg_ids[np.maximum(tpm, 0)], -1)
                traj_ids = np.where(tpm >= 0, trj_ids[np.maximum(tpm, 0)], -1)
                event = table[int(bb)][0]
                ev_ids = np.full(res["adc_digit"][m].shape, event)
                ev_time = np.array([event_times[int(event) % sim.MAX_EVENTS_PER_FILE]])
                # light triggers embedded in the charge stream (:209-221): the simulated ones, else one perfect trigger
                lt_times, lt_events, lt_mods = light_trig_of.get((int(event), int(table[int(bb)][1])),
                                                                 (np.zeros(1), np.array([event]), np.ones(1)))
                pk, assn = packets.build_packets(ev_ids, res["adc_digit"][m], res["adc_ticks_list"][m], res["unique_pix"][m],
                                                 res["current_fractions"][m], track_ids, traj_ids, ev_time,
                                                 light_trigger_times=lt_times, light_trigger_event_id=lt_events,
                                                 light_trigger_modules=lt_mods, bad_channels=bad_list)
                out.append_packets(pk, assn)
                n_packets += len(pk)
            n_hits += int((res["adc_list"] != 0).sum())
            if raw_arrays:
                res["event_id"] = np.array([t[0] for t in table])[res["batch"]]
                parts.append(res)
            b = e
        if raw_arrays and parts:
            for k in parts[0]:
                out.put("raw/" + k, np.concatenate([p[k] for p in parts]))
        if light_simulated and light.LIGHT_TRIG_MODE == 1:          # one beam trigger per spill / event (:1252-1259)
            ev_all = tracks[sim.EVENT_SEPARATOR]
            lev = np.unique(ev_all - (ev_all // sim.MAX_EVENTS_PER_FILE) * sim.MAX_EVENTS_PER_FILE) if sim.IS_SPILL_SIM \
                else (truth["vertices"]["event_id"] if "vertices" in truth else np.unique(ev_all))
            lt = lev * sim.SPILL_PERIOD if sim.IS_SPILL_SIM else event_times
            out.append("light_trig", light_sim.build_light_trig(lev, np.full(len(lev), 0), np.full(len(lev), 0),
                                                                light.TPC_TO_OP_CHANNEL[:].ravel(), lt))
        # ---- truth pass-through (:1226-1297): true timing structure restored, edep-sim coordinate convention ---------------------------
        out_tracks = tracks.copy()
        if sim.IS_SPILL_SIM:
            ev = out_tracks[sim.EVENT_SEPARATOR]
            local = ev - (ev // sim.MAX_EVENTS_PER_FILE) * sim.MAX_EVENTS_PER_FILE
            for f in ("t0_start", "t0_end", "t0"):
                out_tracks[f] = out_tracks[f] + local * sim.SPILL_PERIOD
        batching.swap_coordinates(out_tracks)
        out.put(sim.TRACKS_DSET_NAME, out_tracks, attrs={"zbeam": True})
        for k, v in truth.items():
            out.put(k, v)
        out.close()
    finally:
        lib.set_option("mc_current", 0)
    print(f"simulated {nsim} segments in {len(table)} batches -> {n_hits} hits, {n_packets} packets"
          + (f", {n_light_trig} light triggers" if light_simulated else ""))
    print("Output saved in:", output_filename)
    return dict(n_segments=nsim, n_batches=len(table), n_hits=n_hits, n_packets=n_packets, n_light_triggers=n_light_trig)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--input_filename", required=True)
    ap.add_argument("--output_filename", required=True)
    ap.add_argument("--config", default="module0")
    ap.add_argument("--config_root", default=None, help="larnd-sim tree holding config/config.yaml and the YAML families "
                                                        "(default: $LARNDSIM_ROOT, else the built-in snapshot keywords)")
    tf = lambda s: s.lower() in ("1", "true", "yes")                # noqa: E731
    ap.add_argument("--mod2mod_variation", type=tf, default=None)
    ap.add_argument("--light_simulated", type=tf, default=None)
    for k in ("pixel_layout", "detector_properties", "simulation_properties", "response_file", "light_lut_filename",
              "light_det_noise_filename", "bad_channels", "pixel_thresholds_file", "pixel_gains_file", *IGNORED):
        ap.add_argument("--" + k, default=None)
    ap.add_argument("--n_events", type=int, default=None)
    ap.add_argument("--rand_seed", type=int, default=None)
    ap.add_argument("--tracks_current_mc", action="store_true",
                    help="induced currents from tracks_current_mc like the reference driver (default: tracks_current)")
    ap.add_argument("--raw_arrays", action="store_true", help="also store the per-pixel arrays (raw/...) and light_sample_inc")
    a = vars(ap.parse_args(argv))
    run_simulation(**a)


if __name__ == "__main__":
    main()
