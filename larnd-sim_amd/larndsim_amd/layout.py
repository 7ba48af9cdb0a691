"""
Record layouts of the ``segments`` dataset and the field table handed to the
C-ABI (``LdsimTrackLayout`` in include/ldsim.h).

``segments_dtype`` restates the edep-sim HDF5 schema the reference reads
(reference cli/dumpTree.py:17-28; aligned, itemsize 152).  The C-ABI does not
hard-code that schema: every entry point takes the record base pointer plus an
(offset, dtype) table, so the f8 test layouts the reference's own tests use
(tests/testQuenching.py:16-19) work through the same kernels.
"""
import ctypes

import numpy as np

segments_dtype = np.dtype([
    ("event_id", "u4"), ("vertex_id", "u8"), ("file_vertex_id", "u8"), ("segment_id", "u4"),
    ("z_end", "f4"), ("traj_id", "u4"), ("file_traj_id", "u4"), ("tran_diff", "f4"),
    ("z_start", "f4"), ("x_end", "f4"), ("y_end", "f4"), ("n_electrons", "u4"),
    ("pdg_id", "i4"), ("x_start", "f4"), ("y_start", "f4"), ("t_start", "f4"),
    ("t0_start", "f8"), ("t0_end", "f8"), ("t0", "f8"), ("dx", "f4"), ("long_diff", "f4"),
    ("pixel_plane", "i4"), ("t_end", "f4"), ("dEdx", "f4"), ("dE", "f4"), ("t", "f4"),
    ("y", "f4"), ("x", "f4"), ("z", "f4"), ("n_photons", "f4")], align=True)

# Field order == enum ldsim_field in include/ldsim.h
FIELDS = ["x_start", "y_start", "z_start", "x_end", "y_end", "z_end", "x", "y", "z",
          "dEdx", "dE", "t", "t_start", "t_end", "t0", "t0_start", "t0_end",
          "n_electrons", "n_photons", "long_diff", "tran_diff", "pixel_plane"]
NFIELDS = len(FIELDS)

# dtype codes == LDSIM_F4.. in include/ldsim.h
F4, F8, I4, U4, I8, U8 = 1, 2, 3, 4, 5, 6
_CODES = {np.dtype('f4'): F4, np.dtype('f8'): F8, np.dtype('i4'): I4, np.dtype('u4'): U4,
          np.dtype('i8'): I8, np.dtype('u8'): U8}


class LdsimTrackLayout(ctypes.Structure):
    _fields_ = [("itemsize", ctypes.c_int32),
                ("offset", ctypes.c_int32 * NFIELDS),
                ("dtype", ctypes.c_int32 * NFIELDS)]


def make_layout(dtype):
    """Build the (offset, dtype-code) table for a numpy structured dtype.

    Absent fields get offset -1 (read as 0, never written)."""
    dtype = np.dtype(dtype)
    lay = LdsimTrackLayout()
    lay.itemsize = dtype.itemsize
    for i, name in enumerate(FIELDS):
        if dtype.names is not None and name in dtype.names:
            ft, off = dtype.fields[name][:2]
            if ft not in _CODES:
                raise TypeError(f"field {name}: unsupported dtype {ft}")
            lay.offset[i] = off
            lay.dtype[i] = _CODES[ft]
        else:
            lay.offset[i] = -1
            lay.dtype[i] = F8
    return lay


def store_codes(dtype):
    """dtype code per field as it is *stored* in ``dtype`` (F8 when absent)."""
    lay = make_layout(dtype)
    return np.array(list(lay.dtype), dtype=np.int32), np.array(list(lay.offset), dtype=np.int32)
