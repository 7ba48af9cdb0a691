"""
The HDF5 file surface, executed with the real h5py: input datasets in, every output dataset / attribute out.

The interpreter the test suite runs on has no h5py in this image; /opt/conda/bin/python3.9 has (h5py 3.3, numpy 1.26).  The
tests start ``tests/h5_driver.py`` there as a subprocess (CPU only): it runs this package's writers -- light_sim.export_*,
merge_module_light_wvfm_same_trigger, packets.write_hdf5, the driver's _Output sink and load_input -- into real files, checks
the contents, and reports the files' structure, which is compared here with tests/golden/h5_layout_*.json: the structure the
REFERENCE's own exporters left in real files for the same calls (oracle/gen_h5_layout.py; larndsim/light_sim.py:647-775,
larndsim/fee.py:284-356) -- names, shapes, maxshapes, dtypes, attributes.
"""
import json
import os
import subprocess
import sys

import pytest

import helpers as H

HERE = os.path.dirname(os.path.abspath(__file__))


def h5_python():
    """an interpreter that can import h5py and numpy, or None"""
    for exe in (sys.executable, "/opt/conda/bin/python3.9"):
        if exe and os.path.exists(exe):
            r = subprocess.run([exe, "-c", "import h5py, numpy, yaml"], capture_output=True)
            if r.returncode == 0:
                return exe
    return None


@pytest.fixture(scope="module")
def layouts(tmp_path_factory):
    exe = h5_python()
    if exe is None:
        pytest.skip("no interpreter with h5py in this environment")
    out = tmp_path_factory.mktemp("h5") / "layouts.json"
    env = {k: v for k, v in os.environ.items() if k not in ("PYTHONPATH", "PYTHONHOME")}
    r = subprocess.run([exe, os.path.join(HERE, "h5_driver.py"), str(out)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    with open(out) as f:
        return json.load(f)


def _gold(name):
    with open(os.path.join(H.GOLD, f"h5_layout_{name}.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("case", ["light_module0", "light_2x2_no_modvar", "light_2x2_no_modvar_m2m"])
def test_light_datasets_layout_equals_the_reference_exporters(layouts, case):
    """light_trig / light_wvfm / light_wvfm_mc_assn after two appending calls in threshold mode (module0) and beam mode
    (2x2), and the per-module light_wvfm/light_wvfm_mod<i> datasets before and after the merge: object for object what the
    reference's exporters wrote (contents are compared inside the driver with tests/golden/light_export_*.npz)."""
    ref, got = _gold(case), layouts[case]
    assert got["final"] == ref["final"]
    assert got["before_merge"] == ref["before_merge"]
    assert ref["final"]["light_wvfm"]["maxshape"] == [None, None, None] and ref["final"]["light_trig"]["maxshape"] == [None]
    if case.endswith("m2m"):
        assert "light_wvfm/light_wvfm_mod3" in ref["before_merge"] and "light_wvfm/light_wvfm_mod3" not in ref["final"]


def test_packets_file_layout(layouts):
    """packets.write_hdf5, called twice: `mc_packets_assn` (dtype, resizable, rows appended) and the `configs` attributes
    are what fee.export_to_hdf5 itself writes with h5py (fee.py:284-356) and equal the reference's; `packets` and `_header`
    are larpix-control's format (third party, absent: restated, unpinned) and are checked for presence and shape only."""
    ref, got = _gold("packets_module0")["final"], layouts["packets_module0"]["final"]
    assert got["mc_packets_assn"] == ref["mc_packets_assn"]
    assert got["configs"]["attrs"] == ref["configs"]["attrs"]
    assert set(ref["configs"]["attrs"]) == {"vdrift", "long_diff", "tran_diff", "lifetime", "drift_length"}
    assert got["packets"]["maxshape"] == [None] and got["packets"]["shape"] == got["mc_packets_assn"]["shape"]
    assert got["_header"]["attrs"]["version"] == "2.4"


def test_cli_output_file_layout(layouts):
    """What the driver leaves at the end of a file (reference cli/simulate_pixels.py:1272-1301): `segments` with the zbeam
    attribute, `light_dat/light_dat_allmodules`, the truth datasets of the input passed through, the resizable packet and light
    datasets, `configs.pixel_layout`.  load_input's round trip of the input datasets (:476-521) is asserted in the driver."""
    got = layouts["cli_output"]["final"]
    assert got["segments"]["attrs"] == {"zbeam": True}
    for name in ("trajectories", "vertices", "mc_hdr", "mc_stack", "light_dat/light_dat_allmodules"):
        assert got[name]["kind"] == "dataset" and got[name]["maxshape"] == got[name]["shape"]     # plain create_dataset
    for name, nd in (("packets", 1), ("mc_packets_assn", 1), ("light_trig", 1), ("light_wvfm", 3), ("light_wvfm_mc_assn", 1)):
        assert got[name]["maxshape"] == [None] * nd, name
    assert got["configs"]["attrs"]["pixel_layout"] == "multi_tile_layout-2.3.16.yaml"
    assert got["light_trig"]["dtype"] == _gold("light_module0")["final"]["light_trig"]["dtype"]
    assert got["light_wvfm_mc_assn"]["dtype"] == _gold("light_module0")["final"]["light_wvfm_mc_assn"]["dtype"]
    assert got["mc_packets_assn"]["dtype"] == _gold("packets_module0")["final"]["mc_packets_assn"]["dtype"]
