"""
ctypes loader of ``libldsim_hip.so`` (the C-ABI in include/ldsim.h) and the process-wide context.

There is no CPU fallback: if the shared library is missing, or no MI355X is visible, every compute
entry point raises ``LdsimError``.
"""
import ctypes as C
import os

import numpy as np

from . import consts
from .abi import ABI_VERSION, LdsimChainStats, LdsimConsts, LdsimTrackLayout, pack_consts

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libldsim_hip.so")


class LdsimError(RuntimeError):
    pass


_lib = None
_ctx = None
_ctx_device = None

EXPORTS = [
    "ldsim_last_error", "ldsim_abi_version", "ldsim_device_count", "ldsim_ctx_create", "ldsim_ctx_destroy",
    "ldsim_host_alloc", "ldsim_host_free",
    "ldsim_set_consts", "ldsim_set_response", "ldsim_set_light_channels", "ldsim_set_light_lut", "ldsim_set_option",
    "ldsim_set_pixel_thresholds", "ldsim_set_pixel_gains", "ldsim_clear_pixel_tables",
    "ldsim_synchronize", "ldsim_quench", "ldsim_drift", "ldsim_max_pixels", "ldsim_get_pixels",
    "ldsim_time_intervals", "ldsim_tracks_current", "ldsim_tracks_current_stats", "ldsim_tracks_current_mc", "ldsim_track_pixel_map", "ldsim_sum_pixel_signals",
    "ldsim_get_adc_values", "ldsim_digitize", "ldsim_light_incidence", "ldsim_sum_light_signals",
    "ldsim_scintillation_effect", "ldsim_light_detector_response",
    "ldsim_segments_upload", "ldsim_segments_download", "ldsim_segments_reset", "ldsim_dev_quench_drift", "ldsim_charge_chain",
    "ldsim_chain_download", "ldsim_chain_download_async", "ldsim_chain_download_wait", "ldsim_chain_compact_hits", "ldsim_chain_compact_build", "ldsim_chain_compact_download", "ldsim_chain_kernel_ms", "ldsim_chain_kernel_ms_detail",
    "ldsim_dev_light_incidence", "ldsim_dev_light_incidence_download", "ldsim_dev_light_t0_range", "ldsim_dev_sum_light",
    "ldsim_dev_light_download", "ldsim_light_kernel_ms",
    "ldsim_rng_seed", "ldsim_rng_states_download", "ldsim_rng_clear", "ldsim_rng_extend", "ldsim_rng_count",
    "ldsim_stat_fluctuations", "ldsim_light_triggers", "ldsim_light_detector_noise", "ldsim_sim_triggers",
    "ldsim_dev_light_response", "ldsim_dev_light_response_download", "ldsim_light_response_ms",
    "ldsim_comm_unique_id", "ldsim_comm_init", "ldsim_comm_destroy", "ldsim_comm_count", "ldsim_comm_allreduce_f64", "ldsim_hits_accumulate",
    "ldsim_comm_allgather_hits", "ldsim_comm_gathered_download",
    "ldsim_packets_build", "ldsim_packets_row_bytes", "ldsim_packets_assn_row_bytes", "ldsim_crc32_parts",
]


def load():
    """dlopen the HIP library (does not touch the GPU)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LdsimError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(make -C larnd-sim_amd/csrc). There is no CPU fallback.")
        _lib = C.CDLL(LIB_PATH)
        _lib.ldsim_last_error.restype = C.c_char_p
        _lib.ldsim_packets_build.restype = C.c_int64
        _lib.ldsim_crc32_parts.restype = C.c_uint32
        _lib.ldsim_crc32_parts.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_int64, C.c_int32]
        for name in EXPORTS:
            getattr(_lib, name)   # fail loudly on a missing symbol
        if int(_lib.ldsim_abi_version()) != ABI_VERSION:
            v, _lib = int(_lib.ldsim_abi_version()), None
            raise LdsimError(f"{LIB_PATH} has ABI version {v}, this package expects {ABI_VERSION}: rebuild it "
                             "(make -C larnd-sim_amd/csrc)")
    return _lib


def device_count():
    return int(load().ldsim_device_count())


LDSIM_ENOSPC = -3

_light_shape = (0, 0, 0)


def set_light_shape(shape):
    """(n_det, n_ticks, max_truth) of the device-resident photon sum (ChargeChain.sum_light)."""
    global _light_shape
    _light_shape = tuple(int(v) for v in shape)


def set_light_shape_tuple(shape):
    """the same for a tuple of ints already made (the per-batch fast path of ChargeChain.sum_light)"""
    global _light_shape
    _light_shape = shape


def context_light_shape():
    return _light_shape


def check(rc):
    if rc != 0:
        raise LdsimError(f"ldsim error {rc}: {load().ldsim_last_error().decode()}")


def pinned_array(shape, dtype):
    """numpy array over page-locked host memory (ldsim_host_alloc / hipHostMalloc); freed with the array."""
    import weakref
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    p = C.c_void_p()
    check(load().ldsim_host_alloc(C.byref(p), C.c_size_t(max(n, 8))))
    buf = (C.c_char * max(n, 8)).from_address(p.value)
    arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
    weakref.finalize(buf, load().ldsim_host_free, C.c_void_p(p.value))
    return arr


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


_consts_bytes = None
_consts_generation = 0


def consts_generation():
    """Advances every time the constants frozen in the process-wide ctx actually change.  A ChargeChain freezes them once
    and remembers this number, so a second chain built under another configuration cannot swap them under it silently."""
    return _consts_generation


def context(device=None, refresh_consts=True, noise_zero=False):
    """The process-wide ldsim_ctx (created on first use); constants re-frozen from ``consts`` each call."""
    global _ctx, _ctx_device, _consts_bytes, _consts_generation
    lib = load()
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0")) if _ctx is None else _ctx_device
    if _ctx is not None and device != _ctx_device:
        destroy_context()
    c = pack_consts(noise_zero=noise_zero)
    if _ctx is None:
        h = C.c_void_p()
        check(lib.ldsim_ctx_create(C.c_int(device), C.byref(c), C.byref(h)))
        _ctx, _ctx_device = h, device
        _consts_bytes, _consts_generation = bytes(c), _consts_generation + 1
    elif refresh_consts:
        check(lib.ldsim_set_consts(_ctx, C.byref(c)))
        if bytes(c) != _consts_bytes:
            _consts_bytes, _consts_generation = bytes(c), _consts_generation + 1
    return _ctx


_chain_owner = None           # (thread ident, weak reference) of the ChargeChain that claimed the process-wide ctx last


def claim_chain(chain):
    """The process-wide ctx is thread-compatible, not thread-safe (include/ldsim.h): two ChargeChain objects driven from two threads
    would share its segment store and scratch buffers.  The library itself refuses a second thread INSIDE a call (LDSIM_ESTATE,
    csrc/ldsim_dev.h CtxEnter); this refuses the set-up that leads there: a ChargeChain made on one thread while a ChargeChain made
    on another, still running thread is alive.  One process per GPU, or one thread per ctx, is the supported shape."""
    global _chain_owner
    import threading
    import weakref
    me = threading.get_ident()
    if _chain_owner is not None:
        tid, ref = _chain_owner
        if tid != me and ref() is not None and any(t.ident == tid for t in threading.enumerate()):
            raise LdsimError(f"a ChargeChain created on thread {tid} is still alive: the process-wide GPU context serves one thread "
                             "at a time (use one process per GPU, or drop the other chain first)")
    _chain_owner = (me, weakref.ref(chain))


def destroy_context():
    global _ctx, _resp_token, _lut_token
    if _ctx is not None:
        load().ldsim_ctx_destroy(_ctx)
        _ctx = None
    _resp_token = _lut_token = None


_resp_token = None
_lut_token = None


def _token(arr, sample):
    # identity + a strided content checksum: a freed table's address can be reused by a different table
    return (arr.__array_interface__['data'][0], arr.shape, str(arr.dtype), float(sample))


def set_response(response, ctx=None, force=False):
    """Upload the induction response table unless this exact table is already resident (one token for every
    caller, so the stage API and ChargeChain can never see each other's stale table)."""
    global _resp_token
    ctx = ctx or context()
    r = np.ascontiguousarray(response, dtype=np.float64)
    if r.ndim != 3:
        raise ValueError("response must be [ni][nj][nk]")
    tok = _token(r, r.ravel()[::997].sum())
    if not force and tok == _resp_token:
        return
    check(load().ldsim_set_response(ctx, ptr(r), C.c_int32(r.shape[0]), C.c_int32(r.shape[1]), C.c_int32(r.shape[2])))
    _resp_token = tok


def set_option(name, value, ctx=None):
    ctx = ctx or context()
    check(load().ldsim_set_option(ctx, name.encode(), C.c_double(value)))


def set_light(lut=None, ctx=None):
    """Upload light channel tables (from ``consts.light``) and, if given and not already resident, the LUT."""
    global _lut_token
    ctx = ctx or context()
    lib = load()
    eff = np.ascontiguousarray(consts.light.OP_CHANNEL_EFFICIENCY, dtype=np.float64)
    c2t = np.ascontiguousarray(consts.light.OP_CHANNEL_TO_TPC, dtype=np.int32)
    check(lib.ldsim_set_light_channels(ctx, ptr(eff), ptr(c2t), C.c_int32(len(eff))))
    if lut is not None:
        tok = _token(lut, lut['vis'].ravel()[::97].sum())
        if tok == _lut_token:
            return
        _lut_token = tok
        vis = np.ascontiguousarray(lut['vis'], dtype=np.float32)
        t0 = np.ascontiguousarray(lut['t0'], dtype=np.float32)
        t0a = np.ascontiguousarray(lut['t0_avg'], dtype=np.float32)
        td = np.ascontiguousarray(lut['time_dist'], dtype=np.float32)
        nx, ny, nz, nd = lut.shape
        check(lib.ldsim_set_light_lut(ctx, ptr(vis), ptr(t0), ptr(t0a), ptr(td), C.c_int32(nx), C.c_int32(ny),
                                      C.c_int32(nz), C.c_int32(nd), C.c_int32(td.shape[-1])))


def crc32_parts(parts, n_threads=0):
    """zlib.crc32 of the concatenation of ``parts`` (bytes objects or C-contiguous numpy arrays) on the host threads of
    ``ldsim_crc32_parts`` -- the checksum of an .npz member written piece by piece."""
    import numpy as np
    views = [np.frombuffer(p, dtype=np.uint8) if isinstance(p, (bytes, bytearray, memoryview)) else p for p in parts]
    views = [v for v in views if v.nbytes]
    for v in views:
        if not v.flags["C_CONTIGUOUS"]:
            raise ValueError("crc32_parts needs C-contiguous pieces")
    n = len(views)
    ptrs = (C.c_void_p * max(n, 1))(*[v.ctypes.data for v in views])
    sizes = (C.c_uint64 * max(n, 1))(*[v.nbytes for v in views])
    return int(load().ldsim_crc32_parts(ptrs, sizes, C.c_int64(n), C.c_int32(n_threads)))
