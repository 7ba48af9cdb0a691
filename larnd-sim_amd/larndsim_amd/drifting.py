"""Electron drift to the anode -- mirrors larndsim/drifting.py:11-58 (kernel ``drift``)."""
import ctypes as C

from . import lib
from ._kernel import kernel
from .layout import make_layout


@kernel
def drift(tracks):
    """``drift[bpg, tpb](tracks)``: pixel_plane, lifetime-reduced n_electrons, diffusion sigmas, arrival times."""
    lay = make_layout(tracks.dtype)
    lib.check(lib.load().ldsim_drift(lib.context(), lib.ptr(tracks), C.c_int64(tracks.shape[0]), C.byref(lay)))
