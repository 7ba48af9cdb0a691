#!/opt/conda/bin/python3.9
"""
TEST INFRASTRUCTURE -- HDF5 layout fixtures (runs only in the build container, under an interpreter that has h5py:
/opt/conda/bin/python3.9 here; the system python the rest of the tooling uses has none).

Runs the REFERENCE's own exporters (larndsim/light_sim.py:647-775 export_light_wvfm_to_hdf5, export_light_trig_to_hdf5,
export_to_hdf5, merge_module_light_wvfm_same_trigger; larndsim/fee.py:284-356, the mc_packets_assn / configs-attribute half of
export_to_hdf5) with the REAL h5py into scratch files, on the inputs the committed tests/golden/light_export_*.npz and
packets_*.npz fixtures hold, and records what is left in the files -- object names, shapes, maxshapes, dtypes, attributes --
as tests/golden/h5_layout_<case>.json (names and numbers only).  tests/test_h5_io.py holds this package's writers to it.

numba / cupy are stood in for as in oracle/gen_golden.py.  larpix-control (third party, absent) is stood in for by a
``to_file`` that creates what the reference's next statements need to exist (`configs`, fee.py:350) and nothing else: the
`packets` dataset and `_header` are larpix-control's format, not the reference's, and stay unpinned.

Usage:  /opt/conda/bin/python3.9 oracle/gen_h5_layout.py
"""
import json
import os
import sys
import tempfile
import types

import h5py                      # the real one, before gen_golden's stand-in modules take its name
import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, os.path.join(REPO, "tests"))
import gen_golden as G          # noqa: E402
from h5_layout import describe  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")


def load_ref(cfg, **kw):
    ref = G.Ref(cfg, **kw)
    sys.modules["h5py"] = h5py      # Ref() puts an empty stand-in there (no h5py under the interpreter gen_golden runs on)
    return ref


def light_case(cfg, m2m):
    """the calls of gen_golden.gen_light_export again, into a real file; with `m2m` the per-module datasets and the merge"""
    ref = load_ref(cfg)
    ls, light, sim = ref.light_sim, ref.light, ref.sim
    ls.h5py = h5py
    g = np.load(os.path.join(GOLD, f"light_export_{cfg}.npz"))
    sim.MAX_MC_TRUTH_IDS = 3
    sim.MOD2MOD_VARIATION = bool(m2m)
    path = os.path.join(tempfile.mkdtemp(), "light.h5")
    mods = list(ref.detector.MOD_IDS) if m2m else [-1]
    n_per_mod = g["call0_waveforms"].shape[1] // len(mods)
    for i_mod in mods:
        sl = slice(None) if i_mod < 0 else slice((i_mod - 1) * n_per_mod, i_mod * n_per_mod)
        for icall in (0, 1):
            c = {k[len(f"call{icall}_"):]: g[k] for k in g.files if k.startswith(f"call{icall}_")}
            wv, tid, tph = c["waveforms"][:, sl], c["true_track_id"][:, sl].astype("i8"), c["true_photons"][:, sl]
            if light.LIGHT_TRIG_MODE == 0:
                ls.export_to_hdf5(c["event_id"], c["start_times"], c["trigger_idx"], c["op_channel_idx"], wv, path,
                                  c["event_times"], tid, tph, int(c["i_trig"]), i_mod)
            else:
                ls.export_light_wvfm_to_hdf5(c["event_id"], wv, path, tid, tph, int(c["i_trig"]), i_mod)
    if light.LIGHT_TRIG_MODE == 1:
        lev = g["trig1_event_id"]
        ls.export_light_trig_to_hdf5(lev, np.full(len(lev), 0), np.full(len(lev), 0), light.TPC_TO_OP_CHANNEL[:].ravel(), path,
                                     g["trig1_event_times"])
    before = describe(path)
    if m2m:
        ls.merge_module_light_wvfm_same_trigger(path)
    return dict(before_merge=before if m2m else None, final=describe(path))


def packets_case(cfg):
    """fee.export_to_hdf5 twice (append) with real h5py: what it leaves besides larpix-control's own objects"""
    ref = load_ref(cfg, noise_zero=False)
    fee = ref.fee
    g = np.load(os.path.join(GOLD, f"packets_{cfg}.npz"))

    class _Bag:                      # attribute bags, no logic (as in gen_golden.gen_packets)
        def __init__(self, *a, **kw):
            self.__dict__.update(kw)

        def assign_parity(self):
            pass

    def to_file(filename, packet_list, workers=1):
        # larpix-control's writer: here only what the reference's own next statements rely on (`configs` exists, fee.py:350)
        with h5py.File(filename, "a") as f:
            f.require_group("configs")

    fee.Packet_v2 = fee.TimestampPacket = fee.SyncPacket = fee.TriggerPacket = _Bag
    fee.Key = lambda *a: "-".join(str(int(x)) for x in a)
    fee.PacketCollection = lambda packets, read_id=0, message="": packets
    fee.hdf5format = types.SimpleNamespace(to_file=to_file)
    fee.h5py = h5py
    path = os.path.join(tempfile.mkdtemp(), "packets.h5")
    n_ev = len(g["event_times"])
    bad = None
    for _ in range(2):
        fee.export_to_hdf5(g["event_id_list"], g["adc"], g["ticks"], g["unique_pix"], g["fractions"], g["segment_ids"],
                           g["traj_ids"], path, g["event_times"], light_trigger_times=g["trig_times"],
                           light_trigger_event_id=np.arange(n_ev), light_trigger_modules=np.ones(n_ev), bad_channels=bad)
    return dict(final=describe(path), n_rows_per_call=int(len(g["rows"])) if bad is None else None)


def main():
    if not os.path.isdir(G.REF):
        print("reference checkout not present; nothing to do")
        return 0
    cases = {"light_module0": light_case("module0", False), "light_2x2_no_modvar": light_case("2x2_no_modvar", False),
             "light_2x2_no_modvar_m2m": light_case("2x2_no_modvar", True),
             "packets_module0": packets_case("module0")}
    for name, c in cases.items():
        with open(os.path.join(GOLD, f"h5_layout_{name}.json"), "w") as f:
            json.dump(c, f, indent=1, sort_keys=True)
        print(name, sorted(c["final"].keys()))
    return 0


if __name__ == "__main__":
    sys.exit(main())
