"""Time of lightLUT.calculate_light_incidence and light_sim.sum_light_signals at the sizes the reference driver uses: one
event (5000 segments), every optical channel of the detector, LIGHT_WINDOW worth of ticks; without truth slots (scatter
kernel) and with a few (literal kernel).  Stage API: the figures include H2D/D2H of the arrays."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'larnd-sim_amd')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np
import helpers as H
from larndsim_amd import batching, consts, drifting, light_sim, lightLUT, quenching, synth

for cfg, seed in (("module0", 2), ("2x2_no_modvar", 3)):
    H.load_cfg(cfg)
    light = consts.light
    seg = synth.make_segments(5000, seed=20241016 + seed, segs_per_event=5000, spill=bool(consts.sim.IS_SPILL_SIM))
    if consts.sim.IS_SPILL_SIM:
        loc = seg["event_id"] % consts.sim.MAX_EVENTS_PER_FILE
        for f in ("t0", "t0_start", "t0_end"): seg[f] = seg[f] - loc * consts.sim.SPILL_PERIOD
    batching.swap_coordinates(seg)
    r = H.to_ref(seg)
    quenching.quench[1, 256](r, consts.physics.BIRKS); drifting.drift[1, 256](r)
    n, n_op = len(r), int(light.N_OP_CHANNEL)
    lut = synth.make_lut((14, 26, 8), 48, 100, 7)
    inc = np.zeros((n, n_op), dtype=[('segment_id', 'u4'), ('n_photons_det', 'f4'), ('t0_det', 'f4')])
    vox = np.zeros((n, 3), dtype='i4')
    t0 = time.perf_counter(); lightLUT.calculate_light_incidence[1, 256](r, lut, inc, vox); t_inc = time.perf_counter() - t0
    n_ticks, t_start = light_sim.get_nticks(inc)
    n_ticks = min(n_ticks, 50000)
    opc = light.TPC_TO_OP_CHANNEL[:].ravel().astype('i4')
    srt = np.argsort(-inc['n_photons_det'][:, opc], axis=0, kind="stable").T.astype('i4').copy()
    res = []
    for mt in (0, 4):
        out = np.zeros((len(opc), n_ticks), dtype='f4')
        tid = np.full((len(opc), n_ticks, mt), -1, dtype='i8'); tph = np.zeros((len(opc), n_ticks, mt))
        for rep in range(2):
            out[:] = 0; tid[:] = -1; tph[:] = 0
            t0 = time.perf_counter()
            light_sim.sum_light_signals[1, 64](r, vox, np.arange(n, dtype='i8'), inc, opc, lut, t_start, out, tid, tph, srt, 100)
            dt = time.perf_counter() - t0
        res.append((mt, dt, float(out.sum())))
    print("%s: %d segments, %d channels x %d ticks, smearing %s: incidence %.1f ms; photon sum %s" % (
        cfg, n, len(opc), n_ticks, light.ENABLE_LUT_SMEARING, 1e3 * t_inc,
        ", ".join("%d truth slots %.1f ms (sum %.4g)" % (m, 1e3 * d, s) for m, d, s in res)))
