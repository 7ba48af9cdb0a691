"""Light incidence from the look-up table -- mirrors larndsim/lightLUT.py:65-136."""
import ctypes as C

import numpy as np

from . import lib
from ._kernel import kernel
from .layout import make_layout

def _ensure_lut(lut):
    lib.set_light(lut)                # channel tables always, the LUT only when it is not already resident


@kernel
def calculate_light_incidence(tracks, lut, light_incidence, voxel):
    """``calculate_light_incidence[bpg, tpb](tracks, lut, light_incidence, voxel)``; ``light_incidence`` is the
    structured array with ``n_photons_det`` / ``t0_det`` (f4) the reference driver allocates."""
    lay = make_layout(tracks.dtype)
    n, n_out = light_incidence.shape
    lib.context()
    _ensure_lut(lut)
    nph = np.zeros((n, n_out), dtype=np.float32)
    t0d = np.zeros((n, n_out), dtype=np.float32)
    vox = np.zeros((n, 3), dtype=np.int32)
    lib.check(lib.load().ldsim_light_incidence(lib.context(refresh_consts=False), lib.ptr(tracks), C.c_int64(n),
                                               C.byref(lay), C.c_int32(n_out), lib.ptr(nph), lib.ptr(t0d),
                                               lib.ptr(vox)))
    light_incidence['n_photons_det'] = nph
    if 't0_det' in light_incidence.dtype.names:
        light_incidence['t0_det'] = t0d
    voxel[:] = vox
