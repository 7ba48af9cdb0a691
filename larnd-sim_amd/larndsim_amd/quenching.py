"""Electron recombination -- mirrors larndsim/quenching.py:11-44 (kernel ``quench``)."""
import ctypes as C

from . import lib
from ._kernel import kernel
from .layout import make_layout


@kernel
def quench(tracks, mode):
    """``quench[bpg, tpb](tracks, mode)``: fills ``n_electrons`` / ``n_photons`` in place.

    Raises ValueError for an invalid mode and RuntimeError for a NaN recombination factor, like the
    reference's in-kernel ``raise`` (quenching.py:37-41)."""
    lay = make_layout(tracks.dtype)
    ctx = lib.context()
    rc = lib.load().ldsim_quench(ctx, lib.ptr(tracks), C.c_int64(tracks.shape[0]), C.byref(lay), C.c_int32(int(mode)))
    if rc:
        msg = lib.load().ldsim_last_error().decode()
        if "mode" in msg:
            raise ValueError(msg)
        if "recombination value" in msg:
            raise RuntimeError(msg)
        lib.check(rc)
