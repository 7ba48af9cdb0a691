"""
Configuration keywords -> file sets, mirroring ``larndsim.config.get_config`` (larndsim/config/config.py:40-69).

Two sources:
* a user-supplied larnd-sim tree (``root`` argument or the ``LARNDSIM_ROOT`` environment variable: the directory that holds
  ``config/config.yaml``, ``detector_properties/``, ``pixel_layouts/``, ``simulation_properties/`` and ``bin/``): the YAML
  map is read from there and resolved exactly like the reference does -- bare file names are joined to their family's
  directory, names containing ``/`` are kept, lists are resolved element by element, extra keys pass through;
* without one, the built-in keywords whose constants ship as numbers-only snapshots (``snapshots/*.json``).

A keyword whose entry has ``MOD2MOD_VARIATION: True`` (``2x2``, ``2x2_mpvmpr``, ``2x2_old_response``, ... in the reference's
config.yaml) describes a detector with per-module pixel layouts / responses / LUTs / thresholds / gains:
``module_variation_active`` and ``module_files`` restate the reference's decision and its ``<X>_ID`` pointer lists, and the
driver (cli/simulate_pixels.py) runs its module loop on them.
"""
import os

import yaml

CONFIG_DIR_NAMES = dict(SIM_PROPERTIES='simulation_properties', PIXEL_LAYOUT='pixel_layouts',
                        DET_PROPERTIES='detector_properties', RESPONSE='bin', LIGHT_LUT='bin', LIGHT_DET_NOISE='bin')

# built-in keywords: the file names are the ones the reference's own tests / config.yaml name for these detectors
BUILTIN = {
    'module0': dict(SNAPSHOT='module0', SIM_PROPERTIES='singles_sim.yaml', PIXEL_LAYOUT='multi_tile_layout-2.3.16.yaml',
                    DET_PROPERTIES='module0.yaml', RESPONSE='response_44.npy', LIGHT_SIMULATED=True),
    # the reference's module-variation keyword: constants snapshots per module (2x2.yaml, layouts 2.4.16 / 2.5.16 with
    # PIXEL_LAYOUT_ID [0, 0, 1, 0], 2x2_NuMI_sim.yaml); its response and LUT files are not part of the reference checkout
    '2x2': dict(SNAPSHOT=['2x2_mod1', '2x2_mod2', '2x2_mod3', '2x2_mod4'], SIM_PROPERTIES='2x2_NuMI_sim.yaml',
                PIXEL_LAYOUT=['multi_tile_layout-2.4.16.yaml', 'multi_tile_layout-2.5.16.yaml'], PIXEL_LAYOUT_ID=[0, 0, 1, 0],
                DET_PROPERTIES='2x2.yaml', RESPONSE=['response_44_v2a_50ns.npy', 'response_38_v2b_50ns.npy'],
                RESPONSE_ID=[0, 0, 1, 0], LIGHT_SIMULATED=True, LIGHT_LUT_ID=[0, 1, 1, 1], MOD2MOD_VARIATION=True),
    '2x2_no_modvar': dict(SNAPSHOT='2x2_no_modvar', SIM_PROPERTIES='2x2_NuMI_sim_no_modvar.yaml',
                          PIXEL_LAYOUT='multi_tile_layout-2.4.16.yaml', DET_PROPERTIES='2x2_no_modvar.yaml',
                          RESPONSE='response_44.npy', LIGHT_SIMULATED=True, MOD2MOD_VARIATION=False),
    'ndlar': dict(SNAPSHOT='ndlar', SIM_PROPERTIES='NDLAr_LBNF_sim.yaml', PIXEL_LAYOUT='multi_tile_layout-3.0.40.yaml',
                  DET_PROPERTIES='ndlar-module.yaml', RESPONSE='response_38.npy', LIGHT_SIMULATED=True, LIGHT_LUT='',
                  LIGHT_DET_NOISE=''),
}


def _root(root=None):
    root = root or os.environ.get('LARNDSIM_ROOT')
    if root and not os.path.isfile(os.path.join(root, 'config', 'config.yaml')):
        raise FileNotFoundError(f"{root} is not a larnd-sim tree: config/config.yaml not found")
    return root


def config_map(root=None):
    root = _root(root)
    if not root:
        return BUILTIN
    with open(os.path.join(root, 'config', 'config.yaml')) as f:
        return yaml.safe_load(f)


def list_config_keys(root=None):
    return config_map(root).keys()


def get_config(keyname, root=None):
    """Resolved entry of ``keyname``; raises KeyError like the reference for an unknown keyword."""
    root = _root(root)
    cmap = config_map(root)
    if keyname not in cmap:
        extra = "" if root else (" (built-in snapshots only: set LARNDSIM_ROOT or pass --config_root to resolve the keywords "
                                 "of a larnd-sim tree's config.yaml)")
        raise KeyError(f'Key {keyname} not in supported keywords {list(cmap.keys())}{extra}')
    cfg_map = cmap[keyname]
    if not root:
        return dict(cfg_map)
    res = {}
    for key, val in cfg_map.items():
        if key not in CONFIG_DIR_NAMES:
            res[key] = val
        elif isinstance(val, str):
            res[key] = val if '/' in val else os.path.join(root, CONFIG_DIR_NAMES[key], val)
        elif isinstance(val, list):
            res[key] = [v if '/' in v else os.path.join(root, CONFIG_DIR_NAMES[key], v) for v in val]
    return res


def module_variation_active(cfg, n_modules, mod2mod_variation=None, pixel_layout=None, response_file=None, light_lut=None):
    """The reference's decision whether per-module configurations are loaded (cli/simulate_pixels.py:355-372): the flag (or
    the keyword's MOD2MOD_VARIATION), switched off for a one-module detector or when only a single set of pixel layout /
    response / light LUT files is given."""
    m2m = cfg.get('MOD2MOD_VARIATION') if mod2mod_variation is None else mod2mod_variation
    if not m2m:
        return False
    if n_modules == 1:
        return False
    one = lambda v: v is None or isinstance(v, str) or len(v) == 1       # noqa: E731
    return not (one(pixel_layout) and one(response_file) and one(light_lut))


def id_list(value):
    """A pointer list as the command line gives it ('0,0,1,0', '[0, 0, 1, 0]') or as a list -> list of int; None stays."""
    if value is None:
        return None
    if isinstance(value, str):
        value = [v for v in value.replace("[", " ").replace("]", " ").replace(",", " ").split()]
    return [int(v) for v in value]


def module_files(cfg, files, id_name, n_modules, message="", ids=None):
    """One file per module from a list of files and a pointer list -- ``ids`` when given, else the keyword's ``<X>_ID``
    entry (``load_mod2mod_variation_properties``, cli/simulate_pixels.py:106-122): ``files[ids[m]]`` for module m; without a
    usable pointer list the files must already be one per module."""
    if files is None:
        return None
    ids = cfg.get(id_name) if ids is None else ids
    if ids is not None and isinstance(files, list) and len(ids) == n_modules and max(ids) < len(files):
        return [files[i] for i in ids]
    if isinstance(files, list) and len(files) != n_modules:
        raise KeyError(f"Simulation with module variation activated, but the number of {message} is incorrect!")
    return files


def single_file(value, message=""):
    """Without module variation a property may be a string or a one-element list (cli/simulate_pixels.py:403-426)."""
    if isinstance(value, (list, tuple)):
        if len(value) > 1:
            raise KeyError(f"Provided more than one {message} for the simulation with no module variation.")
        return value[0] if value else None
    return value
