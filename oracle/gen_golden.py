#!/usr/bin/env python3
"""
TEST INFRASTRUCTURE -- golden-vector generator (runs only in the build container).

Imports the reference's own, unmodified Python kernels from /root/reference and runs
them serially on small seeded inputs, writing NUMBERS-ONLY fixtures to tests/golden/
and constants snapshots to larnd-sim_amd/larndsim_amd/snapshots/.

The reference kernels are Numba ``@cuda.jit`` functions; numba / cupy / h5py / larpix
are not installed here (ordinary ModuleNotFoundError), so minimal stand-in modules are
placed in ``sys.modules``: ``numba.njit`` = identity, ``numba.cuda.jit`` =
a launcher that walks the launch grid serially and serves ``cuda.grid`` /
``cuda.gridsize`` / ``cuda.atomic``; ``cupy`` = numpy.  No reference source or bytecode
is written anywhere; the fixtures hold inputs and outputs only.

Faithfulness rules (Numba types arithmetic as f64; NumPy-2 scalars do not):
  * float record fields are f8 but hold f4-representable values; integer fields keep
    their real dtypes (u4 n_electrons -> truncation on store, i4 pixel_plane);
  * between stages the mutated float fields are rounded through f4 (the HDF5 schema);
  * FEE noise constants are 0 (the Numba RNG stream is third-party and unpinned).

Usage:  python oracle/gen_golden.py [--sets consts,qd,pixels,chain,sampled,light,light_response] [--jobs 8]
"""
import argparse
import importlib
import itertools
import json
import os
import sys
import types
from multiprocessing import Pool

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLD = os.path.join(REPO, "tests", "golden")
SNAP = os.path.join(REPO, "larnd-sim_amd", "larndsim_amd", "snapshots")
sys.path.insert(0, os.path.join(REPO, "larnd-sim_amd"))
sys.path.insert(0, REPO)

CONFIGS = {
    "module0": ("detector_properties/module0.yaml", "pixel_layouts/multi_tile_layout-2.3.16.yaml",
                "simulation_properties/singles_sim.yaml"),
    "2x2_no_modvar": ("detector_properties/2x2_no_modvar.yaml", "pixel_layouts/multi_tile_layout-2.4.16.yaml",
                      "simulation_properties/2x2_NuMI_sim_no_modvar.yaml"),
    "ndlar": ("detector_properties/ndlar-module.yaml", "pixel_layouts/multi_tile_layout-3.0.40.yaml",
              "simulation_properties/NDLAr_LBNF_sim.yaml"),
}


# --------------------------------------------------------------------------
# stand-in modules
# --------------------------------------------------------------------------
class _State:
    pos = (0, 0, 0)
    size = (1, 1, 1)
    z_only = None       # optional iterable restricting the 3rd grid axis (sampled ticks)


def _tup3(v):
    if isinstance(v, (int, np.integer)):
        return (int(v), 1, 1)
    v = tuple(int(x) for x in v)
    return v + (1,) * (3 - len(v))


class _Kernel:
    def __init__(self, fn):
        self.fn = fn

    def __getitem__(self, cfg):
        bpg, tpb = _tup3(cfg[0]), _tup3(cfg[1])
        size = tuple(b * t for b, t in zip(bpg, tpb))

        def launch(*args):
            _State.size = size
            zs = range(size[2]) if _State.z_only is None else [z for z in _State.z_only if z < size[2]]
            for x in range(size[0]):
                for y in range(size[1]):
                    for z in zs:
                        _State.pos = (x, y, z)
                        self.fn(*args)
        return launch


def _install_standins():
    numba = types.ModuleType("numba")
    cuda = types.ModuleType("numba.cuda")
    crandom = types.ModuleType("numba.cuda.random")
    nerrors = types.ModuleType("numba.core.errors")

    def njit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    def cjit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return _Kernel(a[0])
        return (lambda f: f) if k.get("device") else (lambda f: _Kernel(f))

    def grid(n):
        return _State.pos[0] if n == 1 else tuple(_State.pos[:n])

    def gridsize(n):
        return _State.size[0] if n == 1 else tuple(_State.size[:n])

    class _Atomic:
        @staticmethod
        def add(arr, idx, val):
            arr[idx] += val

        @staticmethod
        def max(arr, idx, val):
            if val > arr[idx]:
                arr[idx] = val

    numba.njit = njit
    numba.cuda = cuda
    cuda.jit = cjit
    cuda.grid = grid
    cuda.gridsize = gridsize
    cuda.atomic = _Atomic
    cuda.random = crandom
    cuda.to_device = lambda a: a
    cuda.device_array = lambda n, dtype=None: np.zeros(n, dtype=dtype)
    crandom.xoroshiro128p_normal_float32 = lambda states, i: 0.0
    crandom.xoroshiro128p_uniform_float32 = lambda states, i: 0.5
    crandom.create_xoroshiro128p_states = lambda n, seed=0: np.zeros(n)
    nerrors.NumbaPerformanceWarning = Warning
    numba.core = types.ModuleType("numba.core")
    numba.core.errors = nerrors

    cupy = types.ModuleType("cupy")
    for name in dir(np):
        if not name.startswith("__"):
            setattr(cupy, name, getattr(np, name))
    cupy.get_array_module = lambda *a: np
    cupy.asnumpy = np.asarray
    cupy.cuda = types.ModuleType("cupy.cuda")

    mods = {"numba": numba, "numba.cuda": cuda, "numba.cuda.random": crandom, "numba.core": numba.core,
            "numba.core.errors": nerrors, "cupy": cupy, "cupy.cuda": cupy.cuda, "h5py": types.ModuleType("h5py")}
    larpix = types.ModuleType("larpix")
    for sub, names in (("packet", ["Packet_v2", "TimestampPacket", "TriggerPacket", "SyncPacket", "PacketCollection"]),
                       ("key", ["Key"]), ("format", ["hdf5format"])):
        m = types.ModuleType("larpix." + sub)
        for nm in names:
            setattr(m, nm, type(nm, (), {}))
        setattr(larpix, sub, m)
        mods["larpix." + sub] = m
    mods["larpix"] = larpix
    sys.modules.update(mods)


class Ref:
    """The reference package loaded for one configuration."""

    def __init__(self, cfgname, noise_zero=True):
        _install_standins()
        if REF not in sys.path:
            sys.path.insert(0, REF)
        for m in [m for m in sys.modules if m == "larndsim" or m.startswith("larndsim.")]:
            del sys.modules[m]
        det, pix, simf = (os.path.join(REF, "larndsim", p) for p in CONFIGS[cfgname])
        from larndsim import consts
        consts.load_properties(det, pix, simf)
        if noise_zero:
            consts.detector.RESET_NOISE_CHARGE = 0
            consts.detector.UNCORRELATED_NOISE_CHARGE = 0
            consts.detector.DISCRIMINATOR_NOISE = 0
        self.consts = consts
        from larndsim.consts import physics, units
        consts.physics, consts.units = physics, units
        self.detector, self.light, self.sim, self.physics = consts.detector, consts.light, consts.sim, physics
        for name in ("quenching", "drifting", "pixels_from_track", "detsim", "fee", "lightLUT", "light_sim"):
            setattr(self, name, importlib.import_module("larndsim." + name))


# --------------------------------------------------------------------------
# record helpers
# --------------------------------------------------------------------------
F4_FIELDS = ["x_start", "y_start", "z_start", "x_end", "y_end", "z_end", "x", "y", "z", "dx", "dEdx", "dE",
             "t", "t_start", "t_end", "n_photons", "long_diff", "tran_diff"]
REF_DTYPE = np.dtype([("event_id", "u4"), ("segment_id", "u4"), ("traj_id", "u4"), ("n_electrons", "u4"),
                      ("pixel_plane", "i4")] + [(f, "f8") for f in F4_FIELDS] +
                     [("t0", "f8"), ("t0_start", "f8"), ("t0_end", "f8")])


def f4(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def to_ref(seg):
    """152-B schema (or any structured array) -> reference-run records (f8 floats holding f4 values)."""
    r = np.zeros(seg.shape[0], dtype=REF_DTYPE)
    for n in REF_DTYPE.names:
        if n in seg.dtype.names:
            r[n] = seg[n]
    return r


def round_f4_fields(r, fields):
    for n in fields:
        r[n] = f4(r[n])


def swap_xz(seg):
    for a, b in (("x_start", "z_start"), ("x_end", "z_end"), ("x", "z")):
        tmp = seg[a].copy(); seg[a] = seg[b]; seg[b] = tmp
    return seg


def hand_segments(det, n, seed, plane_choices=None, long_frac=0.2, steep_frac=0.25, dtype=None):
    """Diverse hand-made segments in the TPC frame (post-swap), f4-rounded."""
    from larndsim_amd.layout import segments_dtype
    rng = np.random.default_rng(seed)
    seg = np.zeros(n, dtype=segments_dtype)
    B = np.asarray(det.TPC_BORDERS)
    sb = np.sort(B, axis=-1)
    ntpc = B.shape[0]
    for i in range(n):
        p = int(rng.integers(0, ntpc)) if plane_choices is None else int(rng.choice(plane_choices))
        lo, hi = sb[p, :, 0] + 0.8, sb[p, :, 1] - 0.8
        c = rng.uniform(lo, hi)
        L = rng.uniform(0.3, 1.6) if rng.random() < long_frac else rng.uniform(0.03, 0.5)
        if rng.random() < steep_frac:
            cz = rng.choice([-1, 1]) * rng.uniform(0.9, 0.9995)
        else:
            cz = rng.uniform(-1, 1)
        ph = rng.uniform(0, 2 * np.pi)
        s = np.sqrt(1 - cz * cz)
        d = np.array([s * np.cos(ph), s * np.sin(ph), cz])
        a, b = c - 0.5 * L * d, c + 0.5 * L * d
        seg["x_start"][i], seg["y_start"][i], seg["z_start"][i] = a
        seg["x_end"][i], seg["y_end"][i], seg["z_end"][i] = b
        seg["dx"][i] = L
        seg["dEdx"][i] = np.clip(rng.normal(2.1, 0.6), 0.5, 12.0)
        seg["event_id"][i] = 0
    for ax in "xyz":
        seg[ax] = 0.5 * (seg[ax + "_start"].astype(np.float64) + seg[ax + "_end"])
    seg["dE"] = seg["dEdx"].astype(np.float64) * seg["dx"]
    seg["segment_id"] = np.arange(n)
    seg["traj_id"] = np.arange(n) // 3
    return seg


def corner_segments(det, n, seed):
    """Geometry corners for tracks_current (the shapes tools/fuzz_chain.py's last three flavours draw): segments hugging a TPC
    face in x / y (neighbour pixels fall off the plane -> pID -1 slots), very short ones (0.5-50 um), nearly along the drift
    axis (0.1-0.8 degrees off it) and nearly perpendicular to it (|dz| ~ 1e-4 of the length), plus a heavily ionising long one."""
    seg = hand_segments(det, n, seed)
    rng = np.random.default_rng(seed + 1000)
    B = np.sort(np.asarray(det.TPC_BORDERS), axis=-1)
    for i in range(n):
        flavour = i % 5
        p = int(rng.integers(0, B.shape[0]))
        lo, hi = B[p, :, 0] + 0.8, B[p, :, 1] - 0.8
        c = rng.uniform(lo, hi)
        L, cz, ph = rng.uniform(0.05, 0.6), rng.uniform(-0.9, 0.9), rng.uniform(0, 2 * np.pi)
        if flavour == 0:                         # hugging a face: centre 0.02-0.2 cm inside the border in x or y
            ax = int(rng.integers(0, 2))
            side = int(rng.integers(0, 2))
            c[ax] = B[p, ax, side] + (1 - 2 * side) * rng.uniform(0.02, 0.2)
            L = rng.uniform(0.02, 0.3)
        elif flavour == 1:                       # very short
            L = 10 ** rng.uniform(-4.3, -2.3)
        elif flavour == 2:                       # along the drift axis
            # 0.1-0.8 degrees off the axis: closer to it the reference's z_interval hands its slice loop ranges that take the
            # pure-Python launcher hours per segment (the HIP-vs-oracle fuzz covers 1e-7 off the axis in seconds)
            cz = rng.choice([-1, 1]) * (1 - 10 ** rng.uniform(-5.7, -4))
            L = rng.uniform(0.05, 0.4)
        elif flavour == 3:                       # perpendicular to it
            cz = rng.choice([-1, 1]) * 10 ** rng.uniform(-5, -3.5)
            L = rng.uniform(0.1, 1.0)
        else:                                    # heavily ionising, long
            L = rng.uniform(1.0, 1.6)
            seg["dEdx"][i] = rng.uniform(8, 30)
        sxy = np.sqrt(max(0.0, 1 - cz * cz))
        d = np.array([sxy * np.cos(ph), sxy * np.sin(ph), cz])
        a, b = c - 0.5 * L * d, c + 0.5 * L * d
        seg["x_start"][i], seg["y_start"][i], seg["z_start"][i] = a
        seg["x_end"][i], seg["y_end"][i], seg["z_end"][i] = b
        seg["dx"][i] = L
    for ax in "xyz":
        seg[ax] = 0.5 * (seg[ax + "_start"].astype(np.float64) + seg[ax + "_end"])
    seg["dE"] = seg["dEdx"].astype(np.float64) * seg["dx"]
    return seg


def run_quench_drift(ref, r, mode):
    n = r.shape[0]
    ref.quenching.quench[max(1, -(-n // 256)), 256](r, mode)
    round_f4_fields(r, ["n_photons"])
    ref.drifting.drift[max(1, -(-n // 256)), 256](r)
    round_f4_fields(r, ["long_diff", "tran_diff", "t", "t_start", "t_end"])
    return r


# --------------------------------------------------------------------------
# fixture sets
# --------------------------------------------------------------------------
def gen_consts():
    from larndsim_amd import consts as my
    os.makedirs(SNAP, exist_ok=True)
    for cfg in CONFIGS:
        ref = Ref(cfg, noise_zero=False)
        snap = my.snapshot_dict(ref.detector, ref.light, ref.sim)
        with open(os.path.join(SNAP, cfg + ".json"), "w") as f:
            json.dump(snap, f)
        print("snapshot", cfg, "TPCs", np.asarray(ref.detector.TPC_BORDERS).shape[0])
    # the module-variation configuration `2x2` of the reference's config.yaml: detector properties 2x2.yaml, pixel layouts
    # [2.4.16, 2.5.16] with PIXEL_LAYOUT_ID [0, 0, 1, 0], 2x2_NuMI_sim.yaml -- one snapshot per module, loaded the way the
    # driver's module loop does (cli/simulate_pixels.py:452-454, 678-682)
    ref = Ref("2x2_no_modvar", noise_zero=False)
    R = os.path.join(REF, "larndsim")
    layouts = [os.path.join(R, "pixel_layouts", f) for f in ("multi_tile_layout-2.4.16.yaml", "multi_tile_layout-2.5.16.yaml")]
    per_module = [layouts[i] for i in (0, 0, 1, 0)]
    det = os.path.join(R, "detector_properties", "2x2.yaml")
    for i_mod in (1, 2, 3, 4):
        ref.consts.light.set_light_properties(det)
        ref.consts.sim.set_simulation_properties(os.path.join(R, "simulation_properties", "2x2_NuMI_sim.yaml"))
        ref.consts.detector.set_detector_properties(det, per_module, i_mod)
        snap = my.snapshot_dict(ref.consts.detector, ref.consts.light, ref.consts.sim)
        with open(os.path.join(SNAP, f"2x2_mod{i_mod}.json"), "w") as f:
            json.dump(snap, f)
        print("snapshot 2x2 module", i_mod, "pitch", ref.consts.detector.PIXEL_PITCH, "bin", ref.consts.detector.RESPONSE_BIN_SIZE,
              "sampling", ref.consts.detector.RESPONSE_SAMPLING, "N_PIXELS", ref.consts.detector.N_PIXELS)


def gen_qd():
    """quench (Birks + Box) and drift on 152-B-schema values incl. edge cases."""
    for cfg, seed in (("module0", 11), ("2x2_no_modvar", 12), ("ndlar", 13)):
        ref = Ref(cfg)
        seg = hand_segments(ref.detector, 48, seed)
        # edge cases: dEdx = 0, huge dEdx, midpoint outside every TPC, nonzero t0
        seg["dEdx"][0] = 0.0; seg["dE"][0] = 1.0
        seg["dEdx"][1] = 1e10; seg["dE"][1] = 1e10
        for f in ("x", "x_start", "x_end"):
            seg[f][2] += 500.0
        seg["t0"][3:] = np.random.default_rng(seed).uniform(0, 5, 45)
        seg["t0_start"] = seg["t0"]; seg["t0_end"] = seg["t0"]
        out = {"segments_in": seg}
        for mode, name in ((ref.physics.BIRKS, "birks"), (ref.physics.BOX, "box")):
            r = to_ref(seg)
            ref.quenching.quench[1, 256](r, mode)
            round_f4_fields(r, ["n_photons"])
            out[f"{name}_n_electrons"] = r["n_electrons"].copy()
            out[f"{name}_n_photons"] = r["n_photons"].copy()
            if name == "birks":
                ref.drifting.drift[1, 256](r)
                for f in ("pixel_plane", "n_electrons", "long_diff", "tran_diff", "t", "t_start", "t_end"):
                    out["drift_" + f] = r[f].copy()          # f64 values before f4 narrowing
        np.savez_compressed(os.path.join(GOLD, f"qd_{cfg}.npz"), **out)
        print("qd", cfg, "planes", np.unique(out["drift_pixel_plane"])[:6])


def _pixel_stage(ref, r):
    det = ref.detector
    n = r.shape[0]
    bpg = max(1, -(-n // 128))
    max_radius = int(np.ceil(max(r["tran_diff"]) * 5 / det.PIXEL_PITCH))
    mp = np.array([0])
    ref.pixels_from_track.max_pixels[bpg, 128](r, mp)
    P = (2 * max_radius + 1) * mp[0] + (1 + 2 * max_radius) * max_radius * 2
    active = np.full((n, mp[0]), -1, dtype=np.int32)
    neigh = np.full((n, P), -1, dtype=np.int32)
    nrad = np.full((n, P), -1, dtype=np.int32)
    nlist = np.zeros(n)
    ref.pixels_from_track.get_pixels[bpg, 128](r, active, neigh, nrad, nlist, max_radius)
    starts = np.empty(n)
    tmax = np.array([0])
    ref.detsim.time_intervals[bpg, 128](starts, tmax, r)
    return dict(max_radius=max_radius, max_pixels=int(mp[0]), active=active, neigh=neigh, nrad=nrad,
                n_pixels_list=nlist, track_starts=starts, max_length=int(tmax[0]))


def gen_pixels():
    for cfg, seed in (("module0", 21), ("2x2_no_modvar", 22), ("ndlar", 23)):
        ref = Ref(cfg)
        seg = hand_segments(ref.detector, 64, seed, long_frac=0.35)
        # push a few segments across / beyond the pixel-plane edges so -1 gaps appear
        sb = np.sort(np.asarray(ref.detector.TPC_BORDERS), axis=-1)
        for i in (0, 1, 2):
            p = i % sb.shape[0]
            seg["x_start"][i] = sb[p, 0, 0] + 0.1; seg["x_end"][i] = sb[p, 0, 0] - 0.5 + 0.3 * i
            seg["y_start"][i] = sb[p, 1, 1] - 0.2; seg["y_end"][i] = sb[p, 1, 1] + 0.4
            seg["z_start"][i] = seg["z_end"][i] = 0.5 * (sb[p, 2, 0] + sb[p, 2, 1])
            for ax in "xyz":
                seg[ax][i] = 0.5 * (float(seg[ax + "_start"][i]) + float(seg[ax + "_end"][i]))
        seg["x"][0] = sb[0, 0, 0] + 0.05   # keep midpoints inside so drift assigns a plane
        seg["y"][0] = sb[0, 1, 1] - 0.05
        r = run_quench_drift(ref, to_ref(seg), ref.physics.BIRKS)
        keep = r["pixel_plane"] != ref.detector.DEFAULT_PLANE_INDEX   # reference indexes OOB otherwise
        r = r[keep]
        out = _pixel_stage(ref, r)
        out["segments_in"] = seg[keep]
        np.savez_compressed(os.path.join(GOLD, f"pixels_{cfg}.npz"), **out)
        print("pixels", cfg, "P", out["neigh"].shape, "max_length", out["max_length"])


def _current_job(args):
    cfg, r, neigh, T, itrk, z_only, resp_kind = args
    from larndsim_amd import synth
    ref = Ref(cfg)
    response = synth.make_response(resp_kind, response_sampling=ref.detector.RESPONSE_SAMPLING)
    P = neigh.shape[1]
    sig = np.zeros((1, P, T), dtype=np.float32)
    _State.z_only = z_only
    ref.detsim.tracks_current[(1, P, -(-T // 64)), (1, 1, 64)](sig, neigh[itrk:itrk + 1], r[itrk:itrk + 1], response)
    _State.z_only = None
    return itrk, sig[0]


def gen_chain(jobs):
    """Full chain quench -> ... -> digitize on a handful of segments (all ticks)."""
    from larndsim_amd import synth
    for cfg, seed, nseg in (("module0", 31, 5),):
        ref = Ref(cfg)
        det = ref.detector
        B = np.asarray(det.TPC_BORDERS)
        # one short "track" of 4 consecutive segments sharing pixels + 1 isolated segment
        seg = hand_segments(det, nseg, seed, plane_choices=[0])
        p0 = np.array([B[0, 0, 0] + 12.3, B[0, 1, 0] + 40.7, B[0, 2, 0] + 6.0 * np.sign(B[0, 2, 1] - B[0, 2, 0])])
        d = np.array([0.62, 0.35, 0.70]); d /= np.linalg.norm(d)
        cuts = np.array([0.0, 0.21, 0.47, 0.58, 0.93])
        for i in range(4):
            a, b = p0 + cuts[i] * d, p0 + cuts[i + 1] * d
            seg["x_start"][i], seg["y_start"][i], seg["z_start"][i] = a
            seg["x_end"][i], seg["y_end"][i], seg["z_end"][i] = b
            seg["dx"][i] = cuts[i + 1] - cuts[i]
            seg["dEdx"][i] = 2.0 + 0.3 * i
        for ax in "xyz":
            seg[ax] = 0.5 * (seg[ax + "_start"].astype(np.float64) + seg[ax + "_end"])
        seg["dE"] = seg["dEdx"].astype(np.float64) * seg["dx"]
        r = run_quench_drift(ref, to_ref(seg), ref.physics.BIRKS)
        pix = _pixel_stage(ref, r)
        neigh, nrad, T = pix["neigh"], pix["nrad"], pix["max_length"]
        n = r.shape[0]
        # tracks_current over ALL ticks costs ~20 min per (segment, pixel) in pure Python (64000 rho calls per tick),
        # so the full-tick `signals` of this set come from the oracle (oracle/ldsim_oracle.c), which the `sampled_*`
        # sets pin to the reference's tracks_current; every stage downstream runs the reference's own source.
        try:
            from oracle import oracle as ORC
        except ImportError:      # run as a script: the script directory shadows the package name
            import oracle as ORC
        from larndsim_amd import consts as my_consts
        my_consts.load_snapshot({"module0": "module0"}[cfg])
        for kk in ("RESET_NOISE_CHARGE", "UNCORRELATED_NOISE_CHARGE", "DISCRIMINATOR_NOISE"):
            setattr(my_consts.detector, kk, 0)
        signals = ORC.tracks_current(r, neigh, T, synth.make_response("golden"))
        # spot-check against the reference itself on a few ticks of every segment
        peak_tick = int(np.argmax(np.abs(signals).sum(axis=(0, 1))))
        spot = sorted({3, peak_tick - 7, peak_tick, peak_tick + 5, T - 2})
        with Pool(jobs) as pool:
            res = pool.map(_current_job, [(cfg, r, neigh, T, i, spot, "golden") for i in range(n)])
        for itrk, sref in res:
            np.testing.assert_allclose(signals[itrk][:, spot], sref[:, spot], rtol=3e-7, atol=0)
        unique_pix = np.unique(neigh.ravel())
        unique_pix = unique_pix[unique_pix != -1]
        pixel_index_map = np.full(neigh.shape, -1, dtype=np.int64)
        for i_ in range(n):
            compare = neigh[i_, ..., np.newaxis] == unique_pix
            idx = np.where(compare)
            pixel_index_map[i_, idx[0]] = idx[1]
        M = ref.sim.MAX_TRACKS_PER_PIXEL
        track_pixel_map = np.full((unique_pix.shape[0], M), -1, dtype=np.int64)
        U = unique_pix.shape[0]
        ref.detsim.get_track_pixel_map2[max(1, -(-U // 32)), 32](
            track_pixel_map[:0] if U == 0 else _PadRows(track_pixel_map, 32),
            _PadVec(unique_pix, 32), neigh, nrad, int(nrad.max()) + 1)
        NT = len(det.TIME_TICKS)
        pixels_signals = np.zeros((U, NT))
        pixels_tracks_signals = np.zeros((U, NT, M))
        overflow = np.zeros(U)
        ref.detsim.sum_pixel_signals[(n, neigh.shape[1], -(-T // 64)), (1, 1, 64)](
            pixels_signals, signals, pix["track_starts"], pixel_index_map, track_pixel_map,
            pixels_tracks_signals, overflow)
        A = ref.sim.MAX_ADC_VALUES
        out = dict(segments_in=seg, signals=signals, unique_pix=unique_pix, pixel_index_map=pixel_index_map,
                   track_pixel_map=track_pixel_map, pixels_signals=pixels_signals, overflow=overflow,
                   response_kind="golden", signals_source="oracle (spot-checked against the reference at 5 ticks per segment)", **pix)
        for thr_name, thr in (("default", det.DISCRIMINATION_THRESHOLD * ref.consts.units.e), ("low", 600.0)):
            time_ticks = np.linspace(0, 1 * det.TIME_INTERVAL[1], NT + 1)
            integral = np.zeros((U, A)); ticks = np.zeros((U, A)); frac = np.zeros((U, A, M))
            thresholds = np.full(U, thr)
            ref.fee.get_adc_values[max(1, -(-U // 128)), 128](
                pixels_signals, pixels_tracks_signals, time_ticks, integral, ticks, 0, np.zeros(1), frac, thresholds)
            out[f"adc_integral_{thr_name}"] = integral
            out[f"adc_ticks_{thr_name}"] = ticks
            out[f"adc_fractions_{thr_name}"] = frac
            out[f"adc_digit_{thr_name}"] = ref.fee.digitize(integral)
            out[f"threshold_{thr_name}"] = thr
        np.savez_compressed(os.path.join(GOLD, f"chain_{cfg}.npz"), **out)
        print("chain", cfg, "U", U, "T", T, "hits(default)", int((out["adc_integral_default"] != 0).sum()),
              "hits(low)", int((out["adc_integral_low"] != 0).sum()))


class _PadRows:
    """Lets a launch of ceil(U/32)*32 threads index rows >= U harmlessly (the reference
    kernel has no bounds check because real launches are padded the same way and cupy
    would fault; the serial launcher just needs the extra threads to be no-ops)."""

    def __init__(self, arr, mult):
        self.arr = arr
        self.shape = arr.shape
        self._scratch = np.full((arr.shape[1],), -1, dtype=arr.dtype)

    def __getitem__(self, i):
        return self.arr[i] if i < self.arr.shape[0] else self._scratch


class _PadVec:
    def __init__(self, arr, mult):
        self.arr = arr
        self.shape = arr.shape

    def __getitem__(self, i):
        return self.arr[i] if i < self.arr.shape[0] else -12345


def gen_sampled(jobs):
    """tracks_current at sampled ticks for many diverse (segment, pixel) pairs."""
    sets = (("module0", 41, 10, "golden", ""), ("2x2_no_modvar", 42, 6, "survey", ""), ("ndlar", 43, 6, "golden", ""),
            ("module0", 51, 10, "golden", "corners_"), ("ndlar", 53, 5, "golden", "corners_"))
    only = os.environ.get("GEN_SAMPLED_ONLY")            # e.g. "corners_" to make only the corner sets
    for cfg, seed, nseg, kind, tag in sets:
        if only is not None and tag != only:
            continue
        ref = Ref(cfg)
        seg = corner_segments(ref.detector, nseg, seed) if tag else hand_segments(ref.detector, nseg, seed, long_frac=0.3, steep_frac=0.35)
        if cfg == "2x2_no_modvar":                      # spill-style t0
            seg["t0"] = np.random.default_rng(seed).uniform(0, 10, nseg)
            seg["t0_start"] = seg["t0"]; seg["t0_end"] = seg["t0"]
        r = run_quench_drift(ref, to_ref(seg), ref.physics.BIRKS)
        keep = r["pixel_plane"] != ref.detector.DEFAULT_PLANE_INDEX
        r, seg = r[keep], seg[keep]
        pix = _pixel_stage(ref, r)
        T = pix["max_length"]
        # sampled ticks: a coarse comb + a dense comb where the synthetic response peaks
        K = 1950 if ref.detector.RESPONSE_SAMPLING >= 0.1 else 3800
        peak = int(round((K - 70) * ref.detector.RESPONSE_SAMPLING / ref.detector.TIME_SAMPLING
                         - (ref.detector.TIME_WINDOW - ref.detector.TIME_PADDING) / ref.detector.TIME_SAMPLING))
        ticks = sorted(set(list(range(3, T, 211)) + list(range(max(0, peak - 60), min(T, peak + 50), 4)) + [0, T - 1]))
        if tag:      # a segment along the drift axis costs the pure-Python launcher ~2 minutes per tick: a dozen ticks
            ticks = sorted(set(list(range(3, T, 811)) + list(range(max(0, peak - 48), min(T, peak + 40), 12)) + [0, T - 1]))
        with Pool(jobs) as pool:
            res = pool.map(_current_job, [(cfg, r, pix["neigh"], T, i, ticks, kind) for i in range(r.shape[0])])
        signals = np.zeros((r.shape[0], pix["neigh"].shape[1], len(ticks)), dtype=np.float32)
        for itrk, s in res:
            signals[itrk] = s[:, ticks]
        np.savez_compressed(os.path.join(GOLD, f"sampled_{tag}{cfg}.npz"), segments_in=seg, ticks=np.array(ticks),
                            signals=signals, response_kind=kind, **pix)
        print("sampled", tag + cfg, "pairs", int((pix["neigh"] >= 0).sum()), "ticks", len(ticks),
              "nonzero", int((signals != 0).sum()))


def gen_light():
    from larndsim_amd import synth
    for cfg, seed in (("module0", 51), ("2x2_no_modvar", 52)):
        ref = Ref(cfg)
        light = ref.light
        seg = hand_segments(ref.detector, 40, seed)
        seg["t0"] = np.random.default_rng(seed).uniform(0, 3, 40)
        seg["t0_start"] = seg["t0"]; seg["t0_end"] = seg["t0"]
        r = run_quench_drift(ref, to_ref(seg), ref.physics.BIRKS)
        n = r.shape[0]
        n_prof = 40
        lut = synth.make_lut((14, 26, 8), 48, n_prof, seed)
        n_op = light.N_OP_CHANNEL
        inc = np.zeros((n, n_op), dtype=[('segment_id', 'u4'), ('n_photons_det', 'f4'), ('t0_det', 'f4')])
        # f4 structured outputs: store through f8 mirrors to keep Numba's f64 arithmetic, then narrow
        inc8 = np.zeros((n, n_op), dtype=[('segment_id', 'u4'), ('n_photons_det', 'f8'), ('t0_det', 'f8')])
        lut8 = np.zeros(lut.shape, dtype=[('vis', 'f8'), ('t0', 'f8'), ('t0_avg', 'f8'), ('time_dist', 'f8', (n_prof,))])
        for f in lut.dtype.names:
            lut8[f] = lut[f]
        voxel = np.zeros((n, 3), dtype='i4')
        ref.lightLUT.calculate_light_incidence[max(1, -(-n // 256)), 256](r, lut8, inc8, voxel)
        inc['n_photons_det'] = inc8['n_photons_det']; inc['t0_det'] = inc8['t0_det']
        inc8['n_photons_det'] = inc['n_photons_det']; inc8['t0_det'] = inc['t0_det']     # narrowed values
        n_ticks, t_start = ref.light_sim.get_nticks(inc)
        n_ticks = min(n_ticks, 1200)
        op_channel = light.TPC_TO_OP_CHANNEL[:].ravel()
        n_det = op_channel.shape[0]
        sorted_indices = np.zeros((n_det, n), dtype=np.int32)
        for idet in range(n_det):
            sorted_indices[idet] = np.argsort(inc[:, idet]['n_photons_det'])[::-1]
        out_inc = np.zeros((n_det, n_ticks), dtype='f4')   # like the reference driver: f64 add, f4 store per update
        M = 4
        true_id = np.full((n_det, n_ticks, M), -1, dtype='i8')
        true_ph = np.zeros((n_det, n_ticks, M))
        ref.light_sim.sum_light_signals[(n_det, -(-n_ticks // 64)), (1, 64)](
            r, voxel, np.arange(n, dtype='i8'), inc8, op_channel, lut8, t_start, out_inc, true_id, true_ph,
            sorted_indices, n_prof)
        np.savez_compressed(os.path.join(GOLD, f"light_{cfg}.npz"), segments_in=seg, lut_seed=seed, n_prof=n_prof,
                            n_photons_det=inc['n_photons_det'], t0_det=inc['t0_det'], voxel=voxel,
                            n_ticks=n_ticks, t_start=t_start, light_sample_inc=out_inc, true_id=true_id.astype(np.int32),
                            true_photons=true_ph.astype(np.float32), sorted_indices=sorted_indices, op_channel=op_channel)
        print("light", cfg, "n_op", n_op, "ticks", n_ticks, "smearing", light.ENABLE_LUT_SMEARING,
              "sum", float(out_inc.sum()))


def gen_light_response():
    """light_sim.calc_scintillation_effect (:148-184) and calc_light_detector_response (:303-337) on small arrays.
    Inputs are f8 mirrors holding f4 values (Numba promotes f4 x float to f64, NumPy 2 does not), outputs are f4 so
    that every `+=` rounds like the real kernel's store.  LIGHT_WINDOW is shortened (the functions read it at call
    time) so that the lower bound max(itick - conv_ticks, 0) is exercised without a 9000-tick pure-Python loop."""
    cases = (("module0", 61, dict(LIGHT_WINDOW=(1.0, 1.4), SIPM_RESPONSE_MODEL=0)),
             ("2x2_no_modvar", 62, dict(LIGHT_WINDOW=(0.0, 0.25), SIPM_RESPONSE_MODEL=1, IMPULSE_TICK_SIZE=0.0025)))
    for cfg, seed, over in cases:
        ref = Ref(cfg)
        light, sim = ref.light, ref.sim
        rng = np.random.default_rng(seed)
        D, T, M = 6, 1300, 3
        for k, v in over.items():
            setattr(light, k, v)
        light.LIGHT_GAIN = -np.linspace(1.0, 3.5, light.N_OP_CHANNEL)          # row-indexed by the kernel: make rows differ
        if light.SIPM_RESPONSE_MODEL == 1:
            tt = np.arange(60) * light.IMPULSE_TICK_SIZE
            light.IMPULSE_MODEL = np.exp(-tt / 0.03) * np.sin(tt / 0.02)       # synthetic measured impulse
        conv_ticks = int(np.ceil((light.LIGHT_WINDOW[1] - light.LIGHT_WINDOW[0]) / light.LIGHT_TICK_SIZE))
        assert conv_ticks < T
        # sparse photon arrivals with a dense burst, f4 values
        inc = np.zeros((D, T), dtype='f4')
        hits = rng.random((D, T)) < 0.03
        inc[hits] = rng.uniform(0.5, 400.0, hits.sum()).astype('f4')
        inc[:, 200:230] = rng.uniform(1.0, 50.0, (D, 30)).astype('f4')
        inc[D - 1] = 0                                                          # an empty channel
        tid = np.full((D, T, M), -1, dtype='i8')
        tph = np.zeros((D, T, M))
        for d, t in zip(*np.nonzero(inc)):
            k = int(rng.integers(1, M + 1))
            tid[d, t, :k] = rng.choice(40, size=k, replace=False)
            tph[d, t, :k] = float(inc[d, t]) * rng.dirichlet(np.ones(k))
        grid = ((D, -(-T // 64)), (1, 64))
        scint = np.zeros((D, T), dtype='f4')
        s_tid = np.full((D, T, M), -1, dtype='i8'); s_tph = np.zeros((D, T, M))
        ref.light_sim.calc_scintillation_effect[grid[0], grid[1]](inc.astype('f8'), tid, tph, scint, s_tid, s_tph)
        # the driver feeds the Poisson-fluctuated array here (calc_stat_fluctuations, RNG); any array will do
        disc = np.rint(scint.astype('f8') * 8).astype('f4')
        resp = np.zeros((D, T), dtype='f4')
        r_tid = np.full((D, T, M), -1, dtype='i8'); r_tph = np.zeros((D, T, M))
        ref.light_sim.calc_light_detector_response[grid[0], grid[1]](disc.astype('f8'), s_tid, s_tph, resp, r_tid, r_tph)
        np.savez_compressed(
            os.path.join(GOLD, f"light_response_{cfg}.npz"), light_window=np.array(light.LIGHT_WINDOW),
            sipm_response_model=light.SIPM_RESPONSE_MODEL, impulse_tick_size=light.IMPULSE_TICK_SIZE,
            impulse_model=np.asarray(light.IMPULSE_MODEL, dtype='f8'), light_gain=light.LIGHT_GAIN,
            mc_truth_threshold=sim.MC_TRUTH_THRESHOLD, light_sample_inc=inc, true_id=tid.astype('i4'), true_photons=tph,
            scint=scint, scint_true_id=s_tid.astype('i4'), scint_true_photons=s_tph, disc=disc,
            response=resp, response_true_id=r_tid.astype('i4'), response_true_photons=r_tph)
        print("light_response", cfg, "conv_ticks", conv_ticks, "model", light.SIPM_RESPONSE_MODEL,
              "scint sum", float(scint.sum()), "response sum", float(resp.sum()),
              "truth slots used", int((s_tid >= 0).sum()), int((r_tid >= 0).sum()))


def det_phases(shape, seed):
    """Deterministic stand-in for cp.random.uniform(size=shape): a multiplicative hash of (row, column, seed) in [0, 1).
    tests/helpers.py holds the same formula."""
    shape = tuple(int(v) for v in np.atleast_1d(shape))
    i, k = np.meshgrid(np.arange(shape[0], dtype=np.uint64), np.arange(shape[1], dtype=np.uint64), indexing='ij')
    h = (i * np.uint64(7919) + k * np.uint64(104729) + np.uint64(seed)) * np.uint64(2654435761)
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF)
    return h.astype(np.float64) / 4294967296.0


class _GA(np.ndarray):
    """ndarray with cupy's .get()"""
    def get(self):
        return np.asarray(self)


def _ga(a):
    return np.asarray(a).view(_GA)


def gen_light_wvfm():
    """The second half of the light chain on small arrays: calc_stat_fluctuations (light_sim.py:186-238) with the restated
    xoroshiro128p generator supplied as numba.cuda.random (the generator itself is third-party and unpinned; this pins the
    Poisson logic around it), get_triggers (:339-443), gen_light_detector_noise (:445-478) with the uniform phases recorded,
    sim_triggers + digitize_signal (:480-619).  LIGHT_TRIG_WINDOW is shortened where the pure-Python grid walk of
    digitize_signal would otherwise take hours; the functions read it at call time."""
    import ctypes as C
    try:
        from oracle import oracle as O
    except ImportError:
        import oracle as O
    ol = O.lib()
    ol.o_rng_uniform_f32.restype = C.c_float
    ol.o_rng_normal_f32.restype = C.c_float

    def rng_fn(name):
        f = getattr(ol, name)
        return lambda states, i: float(f(C.c_void_p(states.ctypes.data + int(i) * states.dtype.itemsize)))

    cases = (("module0", 71, dict()),
             ("2x2_no_modvar", 72, dict(LIGHT_TRIG_MODE=0, LIGHT_TRIG_WINDOW=(0.2, 0.6))),
             ("2x2_no_modvar", 73, dict(LIGHT_TRIG_WINDOW=(0.1, 0.7))))
    for icase, (cfg, seed, over) in enumerate(cases):
        ref = Ref(cfg)
        light, sim, ls = ref.light, ref.sim, ref.light_sim
        crandom = sys.modules["numba.cuda.random"]
        crandom.xoroshiro128p_uniform_float32 = rng_fn("o_rng_uniform_f32")
        crandom.xoroshiro128p_normal_float32 = rng_fn("o_rng_normal_f32")
        for k, v in over.items():
            setattr(light, k, v)
        rng = np.random.default_rng(seed)
        out = dict(light_trig_mode=light.LIGHT_TRIG_MODE, light_trig_window=np.array(light.LIGHT_TRIG_WINDOW),
                   mc_truth_threshold=sim.MC_TRUTH_THRESHOLD)

        # -- calc_stat_fluctuations: means on both sides of 30, zeros, a negative value
        D0, T0 = 5, 700
        inc = np.zeros((D0, T0), dtype='f4')
        on = rng.random((D0, T0)) < 0.6
        inc[on] = (10 ** rng.uniform(1.0, 5.3, on.sum())).astype('f4')         # PE/us: mean per tick 0.01 .. 200
        inc[0, :5] = (-3.0, 0.0, 29999.0, 30000.0, 30001.0)
        states = O.rng_create_states(D0 * T0, 4242 + icase)
        out["fluct_states_before"] = states.copy().view('u8').reshape(-1, 2)
        disc = np.zeros((D0, T0), dtype='f4')
        ls.calc_stat_fluctuations[(D0, -(-T0 // 64)), (1, 64)](inc.astype('f8'), disc, states)
        out.update(fluct_inc=inc, fluct_disc=disc, fluct_states_after=states.view('u8').reshape(-1, 2))

        # -- a detector response with pulses: get_triggers
        op_channel = light.TPC_TO_OP_CHANNEL[:].ravel()
        nd = op_channel.shape[0]
        digit_ticks = int(np.ceil((light.LIGHT_TRIG_WINDOW[1] + light.LIGHT_TRIG_WINDOW[0]) / light.LIGHT_TICK_SIZE))
        T = 3 * digit_ticks + 1777
        Mt = 2
        resp = np.zeros((nd, T), dtype='f4')        # quiet background except in the pulsed groups: keeps the fixture small
        tid = np.full((nd, T, Mt), -1, dtype='i8'); tph = np.zeros((nd, T, Mt))
        per = light.OP_CHANNEL_PER_TRIG
        thr = np.repeat(np.asarray(light.LIGHT_TRIG_THRESHOLD)[..., None], per, axis=-1).ravel()[op_channel].copy()
        thr = thr.reshape(-1, per)[..., 0]
        groups = rng.choice(nd // per, size=min(6, nd // per), replace=False)
        pulse_at = [137, 137 + digit_ticks + 55, 2 * digit_ticks + 600, 3 * digit_ticks + 1200]
        for g in groups:
            if thr[g] < -1e6:
                continue
            resp[g * per:(g + 1) * per] = rng.normal(0, 2.0, (per, T)).astype('f4')
            for t0 in pulse_at[:int(rng.integers(2, 5))]:
                t0 = t0 + int(rng.integers(0, 40))
                w = int(rng.integers(25, 90))
                amp = thr[g] / per * rng.uniform(1.2, 3.0)
                rows = slice(g * per, (g + 1) * per)
                shape = np.exp(-np.arange(w) / (w / 3.0))
                resp[rows, t0:t0 + w] += (amp * shape).astype('f4')
                for r in range(g * per, (g + 1) * per):
                    for t in range(t0, min(t0 + w, T)):
                        tid[r, t, 0] = 1000 + g
                        tph[r, t, 0] = -float(resp[r, t]) * 0.7
                        if (t + r) % 3 == 0:
                            tid[r, t, 1] = 2000 + g
                            tph[r, t, 1] = -float(resp[r, t]) * 0.3 if t % 5 else 1e-9
        trig, trig_op, trig_type = ls.get_triggers(resp, thr, _ga(op_channel), 0)
        trig2 = ls.get_triggers(resp, thr, _ga(op_channel), 1)
        out.update(response=resp, response_true_id=tid.astype('i4'), response_true_photons=tph, group_threshold=thr,
                   op_channel=op_channel.astype('i4'), trigger_idx=np.asarray(trig), trigger_op_channel_idx=np.asarray(trig_op),
                   trigger_type=np.asarray(trig_type), n_trig_subbatch1=len(trig2[0]))
        print("light_wvfm", cfg, over, "triggers", np.asarray(trig).tolist(), "types", np.asarray(trig_type).tolist())

        # -- gen_light_detector_noise with recorded phases
        nbins = 257
        noise_tab = np.abs(rng.normal(0, 1.0, (light.N_OP_CHANNEL, nbins))) * np.linspace(6000.0, 400.0, nbins)
        rec = []

        def uniform(size=None):
            u = det_phases(size, 100 * seed + len(rec))      # a formula, so the fixture need not store them
            rec.append(u)
            return u
        sys.modules["cupy"].random = types.SimpleNamespace(uniform=uniform)
        for shp in ((4, 1500), (3, 1501)):        # shape[1] < 2 divides by an empty mean in the reference (NaN)
            nz = ls.gen_light_detector_noise(shp, noise_tab[:shp[0]])
            out[f"noise_{shp[1]}"] = nz
            out[f"noise_{shp[1]}_phase_seed"] = 100 * seed + len(rec) - 1
        out["noise_spectrum"] = noise_tab

        # -- sim_triggers: zero spectrum (deterministic), then a real one with recorded phases
        digit_samples = int(np.ceil((light.LIGHT_TRIG_WINDOW[1] + light.LIGHT_TRIG_WINDOW[0]) / light.LIGHT_DIGIT_SAMPLE_SPACING))
        if len(trig) == 0:
            raise RuntimeError("no trigger in the golden case")
        for tag, tab in (("quiet", np.zeros_like(noise_tab)), ("noisy", noise_tab)):
            del rec[:]
            # drop a few rows of the signal so that sim_triggers has to add "missing" channels
            keep = np.ones(nd, bool)
            keep[[1, nd // 2, nd - 1]] = False
            args = [a[keep] for a in (resp, op_channel, tid, tph)]
            dsig, dtid, dtph = ls.sim_triggers(
                (max(len(trig), 1), max(np.asarray(trig_op).shape[1], 1), -(-digit_samples // 64)), (1, 1, 64),
                args[0].copy(), _ga(args[1]), args[2].copy(), args[3].copy(), np.asarray(trig), np.asarray(trig_op),
                digit_samples, tab)
            out[f"wvfm_{tag}"] = dsig
            if tag == "quiet":
                out["wvfm_true_id"] = dtid.astype('i4'); out["wvfm_true_photons"] = dtph
            else:
                out["wvfm_noisy_phase_seeds"] = np.array([100 * seed + i for i in range(len(rec))])
            print("  sim_triggers", tag, dsig.shape, "nonzero", int((dsig != 0).sum()), "truth", int((dtid >= 0).sum()),
                  "noise calls", len(rec))
        out["wvfm_keep_rows"] = keep
        out["digit_samples"] = digit_samples
        np.savez_compressed(os.path.join(GOLD, f"light_wvfm_{cfg}_{icase}.npz"), **out)


def gen_light_export():
    """light_sim.export_to_hdf5 / export_light_trig_to_hdf5 / export_light_wvfm_to_hdf5 / zero_suppress_waveform_truth
    (light_sim.py:621-757) with h5py.File replaced by an in-memory sink: what lands in light_trig, light_wvfm and
    light_wvfm_mc_assn after two appending calls (trigger mode 0) and after the separate calls of trigger mode 1."""
    class _DS:
        def __init__(self, data):
            self.a = np.array(data)

        shape = property(lambda self: self.a.shape)

        def resize(self, n, axis=0):
            assert axis == 0
            self.a = np.concatenate([self.a, np.zeros((n - self.a.shape[0],) + self.a.shape[1:], dtype=self.a.dtype)])

        def __setitem__(self, k, v):
            self.a[k] = v

        def __getitem__(self, k):
            return self.a[k]

    class _File:
        store = {}

        def __init__(self, *a, **k):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def __contains__(self, name):
            return name in self.store

        def create_dataset(self, name, data=None, **k):
            self.store[name] = _DS(data)

        def __getitem__(self, name):
            return self.store[name]

    for cfg, seed in (("module0", 81), ("2x2_no_modvar", 82)):
        ref = Ref(cfg)
        ls, light, sim = ref.light_sim, ref.light, ref.sim
        sim.MAX_MC_TRUTH_IDS = 3
        sim.MOD2MOD_VARIATION = False            # the driver sets it (cli/simulate_pixels.py:387)
        ls.h5py = types.SimpleNamespace(File=_File)
        _File.store = {}
        rng = np.random.default_rng(seed)
        all_ch = light.TPC_TO_OP_CHANNEL[:].ravel()
        ndm = 96 if light.LIGHT_TRIG_MODE == 0 else all_ch.shape[0]
        ns, Mt = 24, 3
        out = dict(light_trig_mode=light.LIGHT_TRIG_MODE)
        event_times = np.array([1000.5, 3.3e5, 1.21e6, 2.6e6])
        calls = []
        for icall, ntrig in enumerate((2, 1)):
            ev = np.full(ntrig, 1 + icall)
            start = np.full(ntrig, 0.25 * icall - 1.0)
            tidx = np.sort(rng.integers(0, 9000, ntrig))
            opc = np.stack([all_ch[:ndm]] * ntrig)
            wv = (rng.integers(-2000, 50, (ntrig, ndm, ns)) * 4).astype('f8')
            tid = np.full((ntrig, ndm, ns, Mt), -1, dtype='i8'); tph = np.zeros((ntrig, ndm, ns, Mt))
            hit = rng.random((ntrig, ndm, ns)) < 0.02
            for it, ic, isamp in zip(*np.nonzero(hit)):
                k = int(rng.integers(1, Mt + 1))
                tid[it, ic, isamp, :k] = rng.integers(0, 500, k)
                tph[it, ic, isamp, :k] = rng.uniform(0.1, 30.0, k)
            uniq_times = event_times[np.unique(ev) % sim.MAX_EVENTS_PER_FILE]
            i_trig = 5 + icall
            if light.LIGHT_TRIG_MODE == 0:
                ls.export_to_hdf5(ev, start, tidx, opc, wv, "x.h5", uniq_times, tid, tph, i_trig, -1)
            else:
                ls.export_light_wvfm_to_hdf5(ev, wv, "x.h5", tid, tph, i_trig, -1)
            calls.append(dict(event_id=ev, start_times=start, trigger_idx=tidx, op_channel_idx=opc, waveforms=wv,
                              true_track_id=tid.astype('i4'), true_photons=tph, event_times=uniq_times, i_trig=i_trig))
        if light.LIGHT_TRIG_MODE == 1:          # the once-per-file call, cli/simulate_pixels.py:1252-1259
            lev = np.array([0, 1, 3])
            ls.export_light_trig_to_hdf5(lev, np.full(3, 0), np.full(3, 0), all_ch, "x.h5", lev * sim.SPILL_PERIOD)
            out.update(trig1_event_id=lev, trig1_event_times=lev * sim.SPILL_PERIOD)
        for i, c in enumerate(calls):
            for k, v in c.items():
                out[f"call{i}_{k}"] = v
        trig = _File.store["light_trig"].a
        out.update(light_trig_op_channel=trig["op_channel"], light_trig_ts_s=trig["ts_s"], light_trig_ts_sync=trig["ts_sync"],
                   light_wvfm=_File.store["light_wvfm"].a)
        assn = _File.store["light_wvfm_mc_assn"].a
        for f in assn.dtype.names:
            out["assn_" + f] = assn[f]
        out["assn_dtype"] = np.array(str(assn.dtype.descr))
        np.savez_compressed(os.path.join(GOLD, f"light_export_{cfg}.npz"), **out)
        print("light_export", cfg, "mode", light.LIGHT_TRIG_MODE, "light_trig", trig.shape, trig.dtype, "wvfm",
              _File.store["light_wvfm"].a.shape, "assn", assn.shape)


def gen_packets():
    """fee.export_to_hdf5 (fee.py:84-356) on the golden chain's ADC arrays, replicated over events.  larpix-control is a
    third-party package that is not installed: its packet classes are replaced by attribute bags that record what the
    reference assigns (no logic of their own; assign_parity is a no-op), hdf5format.to_file and h5py.File by sinks.  The
    fixture therefore pins the reference's own logic -- which slots become packets, time ticks and rollover, the pixel ->
    io_group / io_channel / chip / channel mapping, the inserted timestamp / sync / trigger packets, the association rows --
    and nothing of larpix-control's byte format."""
    class _Bag:
        def __init__(self, kind, **kw):
            self.kind = kind
            self.__dict__.update(kw)

    class Packet_v2(_Bag):
        def __init__(self):
            super().__init__("data")

        def assign_parity(self):
            pass

    class TimestampPacket(_Bag):
        def __init__(self, timestamp=None):
            super().__init__("timestamp", timestamp=timestamp)

    class SyncPacket(_Bag):
        def __init__(self, sync_type=None, timestamp=None, io_group=None):
            super().__init__("sync", sync_type=sync_type, timestamp=timestamp, io_group=io_group)

    class TriggerPacket(_Bag):
        def __init__(self, io_group=None, trigger_type=None, timestamp=None):
            super().__init__("trigger", io_group=io_group, trigger_type=trigger_type, timestamp=timestamp)

    class Key:
        def __init__(self, io_group, io_channel, chip_id):
            self.io_group, self.io_channel, self.chip_id = io_group, io_channel, chip_id

    class _Attrs(dict):
        pass

    class _Node:
        def __init__(self):
            self.attrs = _Attrs()

    class _File:
        store = {}

        def __init__(self, *a, **k):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def keys(self):
            return self.store.keys()

        def create_dataset(self, name, data=None, **k):
            self.store[name] = data

        def __getitem__(self, name):
            return self.store.setdefault(name, _Node())

    chain = np.load(os.path.join(GOLD, "chain_module0.npz"))
    for cfg, case, t_events in (("module0", "a", (1000.0, 250000.0, 1.2e6)), ("2x2_no_modvar", "b", (0.0, 1.2e6, 2.4e6))):
        ref = Ref(cfg)
        fee = ref.fee
        fee.Packet_v2, fee.TimestampPacket, fee.SyncPacket, fee.TriggerPacket, fee.Key = (
            Packet_v2, TimestampPacket, SyncPacket, TriggerPacket, Key)
        fee.PacketCollection = lambda packets, read_id=0, message='': packets
        fee.hdf5format = types.SimpleNamespace(to_file=lambda *a, **k: None)
        _File.store = {}
        fee.h5py = types.SimpleNamespace(File=_File)
        U0 = chain["unique_pix"].shape[0]
        n_ev = len(t_events)
        rng = np.random.default_rng(71)
        adc = np.concatenate([chain["adc_digit_low"]] * n_ev)
        ticks = np.concatenate([chain["adc_ticks_low"]] * n_ev)
        frac = np.concatenate([chain["adc_fractions_low"]] * n_ev)
        upix = np.concatenate([chain["unique_pix"]] * n_ev)
        tpm = np.concatenate([chain["track_pixel_map"]] * n_ev)
        # segment ids / trajectory ids of the slots (the driver maps track_pixel_map through them, cli :1107-1113)
        seg_ids = np.where(tpm >= 0, 1000 + tpm, -1)
        traj_ids = np.where(tpm >= 0, 7 + tpm // 2, -1)
        ev = np.repeat(np.arange(n_ev), U0)
        event_id_list = np.repeat(ev[:, None], adc.shape[1], axis=1)
        event_times = np.array(t_events)
        bad = None
        if case == "a":      # one live channel of the golden set disabled through a bad-channels file
            import tempfile, yaml
            pk, _ = fee.export_to_hdf5(event_id_list, adc, ticks, upix, frac, seg_ids, traj_ids, "x.h5", event_times,
                                       light_trigger_times=np.zeros(n_ev), light_trigger_event_id=np.arange(n_ev),
                                       light_trigger_modules=np.ones(n_ev))
            first = [p for p in pk if p.kind == "data"][3]
            bad_dict = {first.chip_key: [int(first.channel_id)]}
            fd, bad = tempfile.mkstemp(suffix=".yaml")
            with os.fdopen(fd, "w") as fh:
                yaml.safe_dump(bad_dict, fh)
            _File.store = {}
        pk, assn = fee.export_to_hdf5(event_id_list, adc, ticks, upix, frac, seg_ids, traj_ids, "x.h5", event_times,
                                      light_trigger_times=np.zeros(n_ev) + 3.0, light_trigger_event_id=np.arange(n_ev),
                                      light_trigger_modules=np.ones(n_ev), bad_channels=bad)
        rows = np.zeros(len(pk), dtype=[("kind", "i4"), ("io_group", "i8"), ("io_channel", "i8"), ("chip_id", "i8"),
                                        ("channel_id", "i8"), ("timestamp", "f8"), ("dataword", "i8"),
                                        ("trigger_type", "i8"), ("receipt_timestamp", "i8"), ("first_packet", "i8")])
        for i, p in enumerate(pk):
            r = rows[i]
            if p.kind == "data":
                g, c, chip = (int(x) for x in p.chip_key.split("-"))
                r["kind"] = 0; r["io_group"] = g; r["io_channel"] = c; r["chip_id"] = chip
                r["channel_id"] = p.channel_id; r["timestamp"] = p.timestamp; r["dataword"] = p.dataword
                r["receipt_timestamp"] = p.receipt_timestamp; r["first_packet"] = p.first_packet
            elif p.kind == "timestamp":
                r["kind"] = 4; r["timestamp"] = float(p.timestamp); r["io_group"] = p.chip_key.io_group
            elif p.kind == "sync":
                r["kind"] = 6; r["timestamp"] = p.timestamp; r["io_group"] = p.io_group; r["trigger_type"] = p.sync_type[0]
            else:
                r["kind"] = 7; r["timestamp"] = p.timestamp; r["io_group"] = p.io_group; r["trigger_type"] = p.trigger_type[0]
        # the driver's own packets between events (cli/simulate_pixels.py:876-890): fee.export_sync_to_hdf5 and
        # fee.export_timestamp_trigger_to_hdf5, for all io groups (i_mod = -1) and for module 1's (i_mod = 1)
        extra = {}

        def bag_rows(pk):
            rr = np.zeros(len(pk), dtype=[("kind", "i4"), ("io_group", "i8"), ("timestamp", "f8"), ("trigger_type", "i8")])
            for i, p in enumerate(pk):
                if p.kind == "timestamp":
                    rr[i] = (4, p.chip_key.io_group, float(p.timestamp), 0)
                elif p.kind == "sync":
                    rr[i] = (6, p.io_group, p.timestamp, p.sync_type[0])
                else:
                    rr[i] = (7, p.io_group, p.timestamp, p.trigger_type[0])
            return rr
        period = ref.detector.CLOCK_RESET_PERIOD * ref.detector.CLOCK_CYCLE
        sync_times = np.array([period, period, 3 * period + 17.0])       # the last one is not a multiple: floored with a warning
        for i_mod in (-1, 1):
            import warnings
            saved = _File.store
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                _File.store = {}
                pk_s, as_s = fee.export_sync_to_hdf5("x.h5", sync_times, i_mod)
            _File.store = {}
            pk_t, as_t = fee.export_timestamp_trigger_to_hdf5("x.h5", event_times, i_mod)
            _File.store = saved
            extra[f"sync_rows_{i_mod}"] = bag_rows(pk_s)
            extra[f"tt_rows_{i_mod}"] = bag_rows(pk_t)
            extra[f"sync_assn_n_{i_mod}"] = len(as_s)
            assert (as_s["event_ids"] == -1).all() and (as_t["segment_ids"] == -1).all() and (as_t["fraction"] == 0).all()
        extra["sync_times"] = sync_times
        np.savez_compressed(os.path.join(GOLD, f"packets_{cfg}.npz"), event_id_list=event_id_list.astype("i4"), **extra,
                            adc=adc, ticks=ticks, fractions=frac, unique_pix=upix, segment_ids=seg_ids, traj_ids=traj_ids,
                            event_times=event_times, trig_times=np.zeros(n_ev) + 3.0,
                            bad_key=np.array(list(bad_dict.keys())[0] if bad else ""),
                            bad_channel=np.array(list(bad_dict.values())[0][0] if bad else -1),
                            rows=rows, assn_event_ids=assn["event_ids"], assn_segment_ids=assn["segment_ids"],
                            assn_fraction=assn["fraction"], assn_file_traj_ids=assn["file_traj_ids"],
                            assn_fraction_traj=assn["fraction_traj"],
                            config_attrs=np.array([_File.store["configs"].attrs[k] for k in
                                                   ("vdrift", "long_diff", "tran_diff", "lifetime", "drift_length")]))
        kinds = {k: int((rows["kind"] == k).sum()) for k in (0, 4, 6, 7)}
        print("packets", cfg, "trig mode", ref.light.LIGHT_TRIG_MODE, "packets", len(pk), kinds)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sets", default="consts,qd,pixels,light,sampled,chain")
    ap.add_argument("--jobs", type=int, default=8)
    a = ap.parse_args()
    if not os.path.isdir(REF):
        print("reference checkout not present; nothing to do")
        return 0
    os.makedirs(GOLD, exist_ok=True)
    for s in a.sets.split(","):
        {"consts": gen_consts, "qd": gen_qd, "pixels": gen_pixels, "light": gen_light, "light_response": gen_light_response,
         "sampled": lambda: gen_sampled(a.jobs), "chain": lambda: gen_chain(a.jobs), "packets": gen_packets, "light_wvfm": gen_light_wvfm, "light_export": gen_light_export}[s]()
    return 0


if __name__ == "__main__":
    sys.exit(main())
