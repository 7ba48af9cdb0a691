"""Pixels under / around each segment -- mirrors larndsim/pixels_from_track.py (max_pixels :43-65,
get_pixels :67-109, pixel2id/id2pixel :13-41)."""
import ctypes as C

import numpy as np

from . import consts, lib
from ._kernel import kernel
from .layout import make_layout

MAX_NEIGHBOR_BACKTRACK_DISTANCE = 4


def pixel2id(pixel_x, pixel_y, pixel_plane):
    d = consts.detector
    return pixel_x + d.N_PIXELS[0] * (pixel_y + d.N_PIXELS[1] * pixel_plane)


def id2pixel(pid):
    d = consts.detector
    return (pid % d.N_PIXELS[0], (pid // d.N_PIXELS[0]) % d.N_PIXELS[1], pid // (d.N_PIXELS[0] * d.N_PIXELS[1]))


@kernel
def max_pixels(tracks, n_max_pixels):
    """``max_pixels[bpg, tpb](tracks, n_max_pixels)``: atomic-max of the walk length into ``n_max_pixels[0]``."""
    lay = make_layout(tracks.dtype)
    v = C.c_int64(int(n_max_pixels[0]))
    lib.check(lib.load().ldsim_max_pixels(lib.context(), lib.ptr(tracks), C.c_int64(tracks.shape[0]), C.byref(lay),
                                          C.byref(v)))
    n_max_pixels[0] = v.value


@kernel
def get_pixels(tracks, active_pixels, neighboring_pixels, neighboring_radius, n_pixels_list, radius):
    """``get_pixels[bpg, tpb](tracks, active, neigh, nrad, n_pixels_list, radius)``.

    Output arrays are fully rewritten (-1 fill + results), i.e. they behave as if pre-filled with -1 the
    way the reference driver allocates them (cli/simulate_pixels.py:930-933)."""
    lay = make_layout(tracks.dtype)
    n = tracks.shape[0]
    for a in (active_pixels, neighboring_pixels, neighboring_radius):
        if a.dtype != np.int32 or not a.flags.c_contiguous or a.shape[0] != n:
            raise TypeError("pixel arrays must be C-contiguous int32 [n_tracks, width]")
    nl = n_pixels_list if (n_pixels_list is not None and n_pixels_list.dtype == np.float64) else None
    tmp = np.zeros(n) if nl is None and n_pixels_list is not None else nl
    lib.check(lib.load().ldsim_get_pixels(
        lib.context(), lib.ptr(tracks), C.c_int64(n), C.byref(lay), C.c_int32(int(radius)), lib.ptr(active_pixels),
        C.c_int32(active_pixels.shape[1]), lib.ptr(neighboring_pixels), lib.ptr(neighboring_radius),
        C.c_int32(neighboring_pixels.shape[1]), lib.ptr(tmp)))
    if n_pixels_list is not None and nl is None:
        n_pixels_list[:] = tmp
