"""
Structure of an HDF5 file as plain data: for every group / dataset its attributes and, for datasets, shape, maxshape and
dtype (field names, base types, sub-array shapes).  Used on both sides of the HDF5 layout tests -- by
``oracle/gen_h5_layout.py`` on the files the REFERENCE's exporters write (the committed tests/golden/h5_layout_*.json) and by
``tests/h5_driver.py`` on the files this package writes -- so that the two are compared on the same terms.
Needs h5py: run under an interpreter that has it (this image: /opt/conda/bin/python3.9).
"""
import numpy as np


def dtype_plain(dt):
    dt = np.dtype(dt)
    if dt.names:
        return [[n, dtype_plain(dt.fields[n][0])] for n in dt.names]
    if dt.subdtype:
        return [dtype_plain(dt.subdtype[0]), [int(x) for x in dt.subdtype[1]]]
    return np.dtype(dt).newbyteorder('=').str.lstrip('=<|')


def _plain(v):
    if isinstance(v, bytes):
        return v.decode()
    if isinstance(v, np.ndarray):
        return [_plain(x) for x in v.tolist()]
    if isinstance(v, (np.generic,)):
        return v.item()
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    return v


def describe(path):
    """{object name: {kind, attrs[, shape, maxshape, dtype]}} of every object in the file ('/' = the root group)."""
    import h5py
    out = {}

    def visit(name, obj):
        ent = {"kind": "dataset" if isinstance(obj, h5py.Dataset) else "group",
               "attrs": {k: _plain(v) for k, v in sorted(obj.attrs.items())}}
        if isinstance(obj, h5py.Dataset):
            ent["shape"] = [int(s) for s in obj.shape]
            ent["maxshape"] = [None if m is None else int(m) for m in obj.maxshape]
            ent["dtype"] = dtype_plain(obj.dtype)
        out[name] = ent

    with h5py.File(path, "r") as f:
        out["/"] = {"kind": "group", "attrs": {k: _plain(v) for k, v in sorted(f.attrs.items())}}
        f.visititems(visit)
    return out


def read_all(path):
    """{dataset name: array} of every dataset in the file"""
    import h5py
    out = {}

    def visit(name, obj):
        if isinstance(obj, h5py.Dataset):
            out[name] = np.array(obj)

    with h5py.File(path, "r") as f:
        f.visititems(visit)
    return out
