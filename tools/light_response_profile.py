"""Time of the two light waveform response stages at the sizes the reference driver uses (all channels of the detector,
LIGHT_WINDOW worth of ticks), through the stage API (host buffers: the figures include H2D/D2H of the arrays)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'larnd-sim_amd')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np
import helpers as H
from larndsim_amd import consts, light_sim

for cfg in ("module0", "2x2_no_modvar"):
    H.load_cfg(cfg)
    light = consts.light
    D = int(light.N_OP_CHANNEL)
    T = int((light.LIGHT_WINDOW[1] + light.LIGHT_WINDOW[0]) / light.LIGHT_TICK_SIZE)
    conv = int(np.ceil((light.LIGHT_WINDOW[1] - light.LIGHT_WINDOW[0]) / light.LIGHT_TICK_SIZE))
    rng = np.random.default_rng(3)
    inc = np.zeros((D, T), dtype='f4')
    hit = rng.random((D, T)) < 0.01
    inc[hit] = rng.uniform(1, 200, hit.sum()).astype('f4')
    ni, npn = np.zeros((D, T, 0), dtype='i8'), np.zeros((D, T, 0))
    grid = ((D, -(-T // 64)), (1, 64))
    for rep in range(2):
        scint = np.zeros((D, T), dtype='f4'); resp = np.zeros((D, T), dtype='f4')
        t0 = time.perf_counter(); light_sim.calc_scintillation_effect[grid[0], grid[1]](inc, ni, npn, scint, ni, npn); t1 = time.perf_counter()
        light_sim.calc_light_detector_response[grid[0], grid[1]](scint, ni, npn, resp, ni, npn); t2 = time.perf_counter()
    terms = D * (T * (T + 1) // 2 if T <= conv else conv * (conv + 1) // 2 + (T - conv) * (conv + 1))
    print("%s: %d channels x %d ticks, window %d ticks (%.2e terms per stage): scintillation %.1f ms, detector response %.1f ms "
          "(%.1f G terms/s)" % (cfg, D, T, conv, terms, 1e3 * (t1 - t0), 1e3 * (t2 - t1), terms / (t2 - t1) / 1e9))
