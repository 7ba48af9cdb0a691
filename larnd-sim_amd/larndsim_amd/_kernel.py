"""Launch-syntax shim: ``module.kernel[blocks, threads](arrays...)`` reads like the reference's Numba call
sites (cli/simulate_pixels.py:732 etc.); the launch configuration is accepted and ignored -- grids are
chosen by the HIP library."""
import functools


class Kernel:
    def __init__(self, fn):
        self._fn = fn
        functools.update_wrapper(self, fn)

    def __getitem__(self, launch_config):
        return self

    def __call__(self, *a, **k):
        return self._fn(*a, **k)


def kernel(fn):
    return Kernel(fn)
