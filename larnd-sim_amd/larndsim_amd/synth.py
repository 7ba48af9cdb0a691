"""
Synthetic inputs: straight-track segment sets in the edep-sim
``segments`` schema, induction-response tables and light LUTs.

The reference's large binary inputs (response_*.npy, lightLUT*.npz, the example
edep-sim file) are absent from the reference checkout, so every benchmark and
parity input is generated here from a formula + seed.
"""
import numpy as np

from . import consts
from .layout import segments_dtype

SEED_BASE = 20241016


# --------------------------------------------------------------------------
# induction response tables  R[i, j, k]  (f64, shape (I, J, K))
# --------------------------------------------------------------------------
def make_response(kind="survey", shape=None, response_sampling=None):
    """Synthetic field-response table.

    kind="survey": SURVEY §8d table  R = exp(-(i^2+j^2)/50) * g(k), g a Gaussian bump
        centred K-70 (sigma 6 ticks) normalised so sum_k g * RESPONSE_SAMPLING = 1.
        (g underflows to exactly 0 more than ~231 ticks from the centre.)
    kind="dense": same spatial part, g plus a 1/(d+10)^2 far-field tail so that no
        entry is exactly zero (the real tables have full support).
    kind="golden": deliberately asymmetric in (i, j) and bipolar in time so that an
        i/j swap, an off-by-one in k or a sign slip cannot cancel; used for fixtures.
    """
    dt = consts.detector.RESPONSE_SAMPLING if response_sampling is None else response_sampling
    if shape is None:
        shape = (45, 45, 1950) if dt >= 0.1 - 1e-12 else (45, 45, 3800)
    I, J, K = shape
    i = np.arange(I, dtype=np.float64)[:, None, None]
    j = np.arange(J, dtype=np.float64)[None, :, None]
    k = np.arange(K, dtype=np.float64)[None, None, :]
    centre = K - 70.0
    sigma = 6.0
    g = np.exp(-0.5 * ((k - centre) / sigma) ** 2)
    g = g / (g.sum() * dt)
    if kind == "survey":
        return np.exp(-(i * i + j * j) / 50.0) * g
    if kind == "dense":
        tail = 1.0 / (np.abs(centre - k) + 10.0) ** 2
        tail = tail / (tail.sum() * dt)
        return np.exp(-(i * i + j * j) / 50.0) * (0.8 * g + 0.2 * tail)
    if kind == "golden":
        g2 = np.exp(-0.5 * ((k - (centre - 25.0)) / 11.0) ** 2)
        g2 = g2 / (g2.sum() * dt)
        tail = 1.0 / (np.abs(centre - k) + 10.0) ** 2
        tail = tail / (tail.sum() * dt)
        spatial = np.exp(-(i * i + 2.0 * j * j) / 60.0)
        skew = (i + 2.0 * j) / (I + 2.0 * J)
        return spatial * (g * (1.0 - 0.3 * skew) - 0.45 * skew * g2 + 0.05 * tail)
    raise ValueError(kind)


# --------------------------------------------------------------------------
# light look-up table  (structured, shape (nx, ny, nz, n_det))
# --------------------------------------------------------------------------
def lut_dtype(n_prof=100):
    return np.dtype([('vis', 'f4'), ('t0', 'f4'), ('t0_avg', 'f4'), ('time_dist', 'f4', (n_prof,))])


def make_lut(vox_div=(14, 26, 8), n_det=48, n_prof=100, seed=SEED_BASE):
    """Synthetic LUT: vis~U(1e-4,1e-2), t0~U(0,20) ns, normalised exponential time profile."""
    rng = np.random.default_rng(seed)
    shape = tuple(vox_div) + (n_det,)
    lut = np.zeros(shape, dtype=lut_dtype(n_prof))
    lut['vis'] = rng.uniform(1e-4, 1e-2, shape)
    lut['t0'] = rng.uniform(0, 20, shape)
    tau = rng.uniform(5.0, 30.0, shape)[..., None]
    prof = np.exp(-np.arange(n_prof)[None, None, None, None, :] / tau)
    prof /= prof.sum(axis=-1, keepdims=True)
    lut['time_dist'] = prof
    lut['t0_avg'] = (prof * np.arange(n_prof)).sum(axis=-1) + lut['t0']
    return lut


def set_synthetic_light(channels_per_tpc=48):
    """a detector-properties file without light keys (ndlar) gets a synthetic light set-up --
    ``channels_per_tpc`` optical channels per TPC in TPC order, efficiency 1.  Writes the same ``consts.light`` fields
    ``consts.set_light_properties`` would have filled from ``n_op_channel`` / ``tpc_to_op_channel``."""
    n_tpc = int(np.asarray(consts.detector.TPC_BORDERS).shape[0])
    l = consts.light
    l.LIGHT_SIMULATED = True
    l.N_OP_CHANNEL = n_tpc * int(channels_per_tpc)
    l.OP_CHANNEL_EFFICIENCY = np.ones(l.N_OP_CHANNEL)
    l.TPC_TO_OP_CHANNEL = np.arange(l.N_OP_CHANNEL, dtype=int).reshape(n_tpc, int(channels_per_tpc))
    l.OP_CHANNEL_TO_TPC = np.repeat(np.arange(n_tpc, dtype=int), int(channels_per_tpc))
    return l.N_OP_CHANNEL


# --------------------------------------------------------------------------
# straight-track segment sets
# --------------------------------------------------------------------------
def _ray_box_exit(p, d, lo, hi):
    """Distance along unit direction d from p (inside the box) to the box surface."""
    with np.errstate(divide='ignore', invalid='ignore'):
        t1 = (lo - p) / d
        t2 = (hi - p) / d
    t = np.where(d > 0, t2, np.where(d < 0, t1, np.inf))
    return float(np.min(t))


def make_segments(n_segments, seed, segs_per_event=5000, spill=False, tpc_borders=None,
                  spill_period=None, dtype=segments_dtype, event_id0=0, max_track_len=60.0):
    """Straight tracks chopped into segments, edep-sim frame (x<->z swapped).

    The returned array is what an edep-sim ``segments`` dataset would hold: drift axis in
    ``x``; the driver's ``swap_coordinates`` brings it into the TPC_BORDERS frame.
    """
    rng = np.random.default_rng(seed)
    borders = np.asarray(consts.detector.TPC_BORDERS if tpc_borders is None else tpc_borders)
    sb = np.sort(borders, axis=-1)
    if spill_period is None:
        spill_period = consts.sim.SPILL_PERIOD
    out = np.zeros(n_segments, dtype=dtype)
    n = 0
    track_id = 0
    event = event_id0
    in_event = 0
    cols = {k: np.empty(n_segments) for k in
            ("xs", "ys", "zs", "xe", "ye", "ze", "dx", "dedx", "t0")}
    ev = np.empty(n_segments, dtype=np.int64)
    tid = np.empty(n_segments, dtype=np.int64)
    while n < n_segments:
        itpc = int(rng.integers(0, borders.shape[0]))
        lo = sb[itpc, :, 0] + 1.0
        hi = sb[itpc, :, 1] - 1.0
        p0 = rng.uniform(lo, hi)
        while True:
            cz = rng.uniform(-1.0, 1.0)
            ph = rng.uniform(0.0, 2.0 * np.pi)
            s = np.sqrt(1.0 - cz * cz)
            d = np.array([s * np.cos(ph), s * np.sin(ph), cz])
            if abs(d[0]) >= 1e-3:           # post-swap x component (avoids x_start == x_end)
                break
        L = min(_ray_box_exit(p0, d, lo, hi), max_track_len)
        kmax = int(L / 0.05) + 2
        cuts = np.concatenate(([0.0], np.cumsum(rng.uniform(0.05, 0.50, kmax))))
        cuts = cuts[cuts < L]
        ends = np.concatenate((cuts[1:], [L]))
        lens = ends - cuts
        keep = lens >= 0.01
        cuts, ends, lens = cuts[keep], ends[keep], lens[keep]
        m = len(cuts)
        dedx = np.clip(rng.normal(2.1, 0.2, m), 1.0, 10.0)
        t_trk = rng.uniform(0.0, 10.0) if spill else 0.0
        i = 0
        while i < m and n < n_segments:
            take = min(m - i, segs_per_event - in_event, n_segments - n)
            sl = slice(n, n + take)
            a, b = cuts[i:i + take], ends[i:i + take]
            cols["xs"][sl] = p0[0] + a * d[0]; cols["xe"][sl] = p0[0] + b * d[0]
            cols["ys"][sl] = p0[1] + a * d[1]; cols["ye"][sl] = p0[1] + b * d[1]
            cols["zs"][sl] = p0[2] + a * d[2]; cols["ze"][sl] = p0[2] + b * d[2]
            cols["dx"][sl] = lens[i:i + take]
            cols["dedx"][sl] = dedx[i:i + take]
            cols["t0"][sl] = t_trk + ((event % 1000) * spill_period if spill else 0.0)
            ev[sl] = event
            tid[sl] = track_id
            n += take; i += take; in_event += take
            if in_event >= segs_per_event:
                event += 1
                in_event = 0
                break               # rest of this track is dropped; next event starts fresh
        track_id += 1
    # edep-sim frame: drift (TPC z) axis is stored in "x", TPC x in "z"
    out["z_start"] = cols["xs"]; out["z_end"] = cols["xe"]
    out["y_start"] = cols["ys"]; out["y_end"] = cols["ye"]
    out["x_start"] = cols["zs"]; out["x_end"] = cols["ze"]
    for ax in "xyz":
        out[ax] = 0.5 * (out[ax + "_start"].astype(np.float64) + out[ax + "_end"])
    out["dx"] = cols["dx"]
    out["dEdx"] = cols["dedx"]
    out["dE"] = out["dEdx"].astype(np.float64) * out["dx"]
    for f in ("t0", "t0_start", "t0_end"):
        if f in out.dtype.names:
            out[f] = cols["t0"]
    out["event_id"] = ev
    for f in ("traj_id", "file_traj_id"):
        if f in out.dtype.names:
            out[f] = tid
    if "segment_id" in out.dtype.names:
        out["segment_id"] = np.arange(n_segments)
    if "pdg_id" in out.dtype.names:
        out["pdg_id"] = 13
    return out
