"""
An .npz written dataset by dataset from pieces, without joining the pieces and without Python's byte-at-a-time checksum.

The driver's output file stands in for the reference's h5py datasets (reference cli/simulate_pixels.py:1240-1301, fee.py:346-372
`export_to_hdf5`): `packets`, `mc_packets_assn`, the light datasets and the passed-through truth, each grown along axis 0 launch
after launch.  ``numpy.savez`` needs every dataset as ONE array (a concatenate of ~1 GB per 10^6 segments) and ``zipfile`` then
checksums it with ``zlib.crc32`` at ~1 GB/s: 1.5 of 3.25 s per 10^6 module0 segments.  Here a member is the .npy header plus the
pieces as they are, stored (method 0) like savez stores them, and the CRC comes from ``ldsim_crc32_parts`` (csrc/crc32.hip, all
host threads).  ``numpy.load`` reads the result like any .npz; ZIP64 records are written when a size or offset needs them.
"""
import io
import struct
import time

import numpy as np

from . import lib

_LOCAL, _CENTRAL, _END, _END64, _LOC64 = 0x04034B50, 0x02014B50, 0x06054B50, 0x06064B50, 0x07064B50
_MAX32 = 0xFFFFFFFF
_ZIP64_AT = _MAX32          # sizes / offsets from here on go into ZIP64 records (a test lowers it)


def npy_header(dtype, shape):
    """The .npy preamble of a C-ordered array of ``dtype`` and ``shape`` (format 1.0, or 2.0 when the dtype description is long)."""
    d = {"descr": np.lib.format.dtype_to_descr(np.dtype(dtype)), "fortran_order": False, "shape": tuple(int(s) for s in shape)}
    b = io.BytesIO()
    try:
        np.lib.format.write_array_header_1_0(b, d)
    except ValueError:
        b = io.BytesIO()
        np.lib.format.write_array_header_2_0(b, d)
    return b.getvalue()


def _bytes_view(a):
    a = np.ascontiguousarray(a)
    return a.reshape(-1).view(np.uint8) if a.size else np.zeros(0, dtype=np.uint8)


class NpzStream:
    """``with NpzStream(path) as z: z.write(name, pieces)`` -- one member ``name.npy`` per call, ``pieces`` a list of arrays that
    share dtype and trailing shape (joined along axis 0)."""

    def __init__(self, path, n_threads=0):
        self.fp = open(path, "wb")
        self.n_threads = n_threads
        self.members = []          # (name bytes, crc, size, local header offset)
        t = time.localtime()
        self.dos_time = t.tm_hour << 11 | t.tm_min << 5 | t.tm_sec // 2
        self.dos_date = max(t.tm_year - 1980, 0) << 9 | t.tm_mon << 5 | t.tm_mday

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if et is None:
            self.close()
        else:
            self.fp.close()

    def write(self, name, pieces):
        pieces = [np.asarray(p) for p in pieces]
        if not pieces:
            raise ValueError(f"{name}: no piece")
        first = pieces[0]
        for p in pieces[1:]:
            if p.dtype != first.dtype or p.shape[1:] != first.shape[1:] or p.ndim != first.ndim:
                raise ValueError(f"{name}: pieces differ in dtype or trailing shape ({p.dtype}{p.shape} after {first.dtype}{first.shape})")
        if first.dtype.hasobject:
            raise ValueError(f"{name}: object arrays are not written")
        if first.ndim == 0:
            if len(pieces) != 1:
                raise ValueError(f"{name}: several 0-d pieces")
            shape = ()
        else:
            shape = (sum(p.shape[0] for p in pieces),) + first.shape[1:]
        parts = [np.frombuffer(npy_header(first.dtype, shape), dtype=np.uint8)] + [_bytes_view(p) for p in pieces]
        size = sum(p.nbytes for p in parts)
        crc = lib.crc32_parts(parts, self.n_threads)
        fname = (name + ".npy").encode("utf-8")
        offset = self.fp.tell()
        big = size >= _ZIP64_AT
        extra = struct.pack("<HHQQ", 1, 16, size, size) if big else b""
        s32 = _MAX32 if big else size
        self.fp.write(struct.pack("<IHHHHHIIIHH", _LOCAL, 45 if big else 20, 0x800, 0, self.dos_time, self.dos_date, crc, s32, s32,
                                  len(fname), len(extra)) + fname + extra)
        for p in parts:
            if p.nbytes:
                self.fp.write(p)
        self.members.append((fname, crc, size, offset))

    def close(self):
        cd_start = self.fp.tell()
        for fname, crc, size, offset in self.members:
            fields = [v for v, over in ((size, size >= _ZIP64_AT), (size, size >= _ZIP64_AT), (offset, offset >= _ZIP64_AT)) if over]
            extra = struct.pack("<HH" + "Q" * len(fields), 1, 8 * len(fields), *fields) if fields else b""
            s32, o32 = (_MAX32 if size >= _ZIP64_AT else size), (_MAX32 if offset >= _ZIP64_AT else offset)
            v = 45 if fields else 20
            self.fp.write(struct.pack("<IHHHHHHIIIHHHHHII", _CENTRAL, v, v, 0x800, 0, self.dos_time, self.dos_date, crc, s32, s32,
                                      len(fname), len(extra), 0, 0, 0, 0o600 << 16, o32) + fname + extra)
        cd_end = self.fp.tell()
        n, cd_size = len(self.members), cd_end - cd_start
        big = n >= 0xFFFF or cd_size >= _ZIP64_AT or cd_start >= _ZIP64_AT
        if big:
            self.fp.write(struct.pack("<IQHHIIQQQQ", _END64, 44, 45, 45, 0, 0, n, n, cd_size, cd_start))
            self.fp.write(struct.pack("<IIQI", _LOC64, 0, cd_end, 1))
        self.fp.write(struct.pack("<IHHHHIIH", _END, 0, 0, 0xFFFF if big else n, 0xFFFF if big else n, _MAX32 if big else cd_size,
                                  _MAX32 if big else cd_start, 0))
        self.fp.close()
