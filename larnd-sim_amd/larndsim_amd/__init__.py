"""
larndsim_amd: MI355X-native (gfx950) implementation of larnd-sim's charge/light hot path.

Python host code over a C-ABI shared library of hand-written HIP kernels
(``csrc/`` -> ``libldsim_hip.so``); see DESIGN.md.
"""
__version__ = "0.1.0"
