"""Random states of the noisy stages -- mirrors ``numba.cuda.random.create_xoroshiro128p_states`` as the reference's driver
uses it (cli/simulate_pixels.py:92-104,396: one table of 262144 xoroshiro128p states, seeded once, advanced in place by
the kernels).  The table lives in the process-wide GPU context; state ``ip`` serves pixel row ``ip`` of a
``fee.get_adc_values`` call or of a chain launch.  The generator is third-party (module ``numba``) and restated from its
published algorithm (csrc/rng.h): noisy outputs are reproducible with a seed, not pinned to the reference's."""
import ctypes as C

import numpy as np

from . import lib

xoroshiro128p_dtype = np.dtype([("s0", "<u8"), ("s1", "<u8")], align=True)


class RngStates:
    """Handle to the device-resident state table."""

    def __init__(self, n, seed, ctx=None):
        self.n, self.seed = int(n), int(seed)
        self.ctx = ctx or lib.context()
        lib.check(lib.load().ldsim_rng_seed(self.ctx, C.c_uint64(self.seed & (2 ** 64 - 1)), C.c_int64(self.n)))

    def __len__(self):
        return self.n

    def copy_to_host(self, n=None):
        n = self.n if n is None else int(n)
        out = np.zeros(n, dtype=xoroshiro128p_dtype)
        lib.check(lib.load().ldsim_rng_states_download(self.ctx, lib.ptr(out), C.c_int64(n)))
        return out


def create_xoroshiro128p_states(n, seed=0, ctx=None):
    return RngStates(n, seed, ctx)


def maybe_create_rng_states(n, seed=0, rng_states=None, ctx=None):
    """cli/simulate_pixels.py:92-104: create the table, or extend a shorter one with a fresh
    ``create_xoroshiro128p_states(n - len, seed)`` chain; a long enough table is returned untouched."""
    if rng_states is None:
        return RngStates(n, seed, ctx)
    if int(n) > len(rng_states):
        lib.check(lib.load().ldsim_rng_extend(rng_states.ctx, C.c_int64(int(n)), C.c_uint64(int(seed) & (2 ** 64 - 1))))
        rng_states.n = int(n)
    return rng_states
