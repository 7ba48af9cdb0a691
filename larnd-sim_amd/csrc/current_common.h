// current_common.h -- device helpers shared by the monolithic induced-current kernel (kernels_current.hip)
// and the weights / correlation kernel pair (kernels_split.hip).
#pragma once
#include "ldsim_args.h"

#define CUR_THREADS 256
#define NWAVE 4
#define TPL 8                    // ticks per lane
#define WTILE (64 * TPL)         // ticks per wave tile
#define TILE_TICKS (NWAVE * WTILE)
#define ZC 64                    // max z slices per chunk
#define NU_MAX 64                // max distinct response shifts per chunk
#define NJ_MAX 48                // max distinct j cells (response table <= 48 wide in j)
#define NS_MAX 64                // max SAMPLED_POINTS
#define NEDGE 3                  // partially valid edge k's: k=0 and up to 2 at the top of the window
#define ARENA 5888               // f64 weight entries held in LDS per column group
#define QLEN 128                 // survivor queue entries per wave
#define CELLS_MAX 736            // response cells per column group


struct PairGeo {
  double x_p, y_p;
  double sx, sy, sz;       // z-ordered start
  double Dx, Dy, Dz, Dr;   // segment and its length
  double dirx, diry, dirz;
  double sT, sL, q;
  double sT2, sL2, s3;     // sT^2, sL^2 as rho's delta / a use them (f64) and the sigma product of its `factor`
  double rT, rL;           // sigma^2 / (sigma * sigma as _b types it): 1 unless numba_f32 (detsim.py:116-118)
  double z_start_int, z_step, x_step, y_step, x_start, y_start, sgnx, sgny;
  double t_start, z_anode, dV;
  int z_steps;
  bool ok;
};

__device__ __forceinline__ double sgn(double x) { return x >= 0 ? 1.0 : -1.0; }

// which (segment, pixel) a pair is: entry of the chain's sorted pair list, or cell [s][p] of the stage call's dense pixel array
// (pixel id -1 included: the reference indexes with it, see pair_geometry)
__device__ __forceinline__ void pair_ids(const CurArgs& A, int64_t pair, int64_t& seg, int64_t& pID) {
  if (A.pair_val) {
    const int32_t v = A.pair_val[pair];
    seg = A.seg_begin + v / A.P;
    pID = (int64_t)((A.pair_key[pair] >> 4) & 0xFFFFFFFFull);
  } else {
    seg = A.seg_begin + pair / A.P;
    pID = A.pixels[pair];
  }
}

// detsim.py:42-112
// nf32: Numba types f32 (op) f32 as f32 -- with f4 record fields the spots below are single precision
// (detsim.py:74-79; the oracle's F32SUB / F32MUL / F32DIV, oracle/ldsim_oracle.c:270-291)
__device__ __forceinline__ double f32sub(double a, double b, bool nf32) { return nf32 ? (double)((float)a - (float)b) : a - b; }
__device__ __forceinline__ double f32mul(double a, double b, bool nf32) { return nf32 ? (double)((float)a * (float)b) : a * b; }
__device__ __forceinline__ double f32div(double a, double b, bool nf32) { return nf32 ? (double)((float)a / (float)b) : a / b; }

__device__ __forceinline__ void z_interval(const double* sp, const double* ep, double x_p, double y_p, double tol,
                                           double& z_poca, double& z_lo, double& z_hi, bool nf32) {
  const double *start, *end;
  z_poca = z_lo = z_hi = 0;
  if (sp[0] > ep[0]) {
    start = ep;
    end = sp;
  } else if (sp[0] < ep[0]) {
    start = sp;
    end = ep;
  } else {
    return;
  }
  double xs = start[0], ys = start[1], xe = end[0], ye = end[1];
  double m = f32div(f32sub(ye, ys, nf32), f32sub(xe, xs, nf32), nf32);
  double q = f32div(f32sub(f32mul(xe, ys, nf32), f32mul(xs, ye, nf32), nf32), f32sub(xe, xs, nf32), nf32);
  double a = m, b = -1, cc = q;
  double x_poca = (b * (b * x_p - a * y_p) - f32mul(a, cc, nf32)) / (f32mul(a, a, nf32) + b * b);
  double dx = f32sub(end[0], start[0], nf32), dy = f32sub(end[1], start[1], nf32), dz = f32sub(end[2], start[2], nf32);
  double length = sqrt(dx * dx + dy * dy + dz * dz);
  double d0 = dx / length, d2 = dz / length;
  double doca;
  if (x_poca < start[0]) {
    doca = sqrt((x_p - start[0]) * (x_p - start[0]) + (y_p - start[1]) * (y_p - start[1]));
    x_poca = start[0];
  } else if (x_poca > end[0]) {
    doca = sqrt((x_p - end[0]) * (x_p - end[0]) + (y_p - end[1]) * (y_p - end[1]));
    x_poca = end[0];
  } else {
    doca = fabs(a * x_p + b * y_p + cc) / sqrt(f32mul(a, a, nf32) + b * b);
  }
  double zp = start[2] + (x_poca - start[0]) / d0 * d2;
  if (tol > doca) {
    double dxs = f32sub(xe, xs, nf32), dys = f32sub(ye, ys, nf32);
    double length2D = sqrt(dxs * dxs + dys * dys);
    double dir2x = f32sub(end[0], start[0], nf32) / length2D;
    double deltaL2D = sqrt(tol * tol - doca * doca);
    double x_plus = x_poca + deltaL2D * dir2x;
    double x_minus = x_poca - deltaL2D * dir2x;
    double plusL = (x_plus - start[0]) / d0;
    double minusL = (x_minus - start[0]) / d0;
    double plusZ = start[2] + d2 * plusL;
    double minusZ = start[2] + d2 * minusL;
    z_poca = zp;
    z_lo = fmin(minusZ, plusZ);
    z_hi = fmax(minusZ, plusZ);
  }
}

// detsim.py:366-414: everything that does not depend on the tick or the sample
static __device__ void pair_geometry(const CurArgs& A, int64_t seg, int64_t pID, PairGeo& g) {
  const LdsimConsts* c = A.c;
  const SegStore& s = A.s;
  g.ok = false;
  int64_t px, py, pplane;
  id2pixel(c, pID, px, py, pplane);
  if (!(px >= 0 && py >= 0)) return;
  int64_t bplane = pplane < 0 ? pplane + c->n_tpc : pplane;  // Python/Numba negative index wrap (pID == -1)
  if (bplane < 0 || bplane >= c->n_tpc) return;
  int32_t tplane = s.pixel_plane[seg];
  if (tplane < 0 || tplane >= c->n_tpc) return;
  const double(*pb)[2] = c->tpc_borders[bplane];
  double x_p = px * c->pixel_pitch + pb[0][0];
  double y_p = py * c->pixel_pitch + pb[1][0];
  x_p += c->pixel_pitch / 2;
  y_p += c->pixel_pitch / 2;
  double start[3], end[3];
  double xs = s.f[LDSIM_X_START][seg], ys = s.f[LDSIM_Y_START][seg], zs = s.f[LDSIM_Z_START][seg];
  double xe = s.f[LDSIM_X_END][seg], ye = s.f[LDSIM_Y_END][seg], ze = s.f[LDSIM_Z_END][seg];
  if (zs < ze) {
    start[0] = xs; start[1] = ys; start[2] = zs; end[0] = xe; end[1] = ye; end[2] = ze;
  } else {
    end[0] = xs; end[1] = ys; end[2] = zs; start[0] = xe; start[1] = ye; start[2] = ze;
  }
  const bool nf32 = A.numba_f32 != 0;
  g.Dx = f32sub(end[0], start[0], nf32); g.Dy = f32sub(end[1], start[1], nf32); g.Dz = f32sub(end[2], start[2], nf32);
  double length = sqrt(g.Dx * g.Dx + g.Dy * g.Dy + g.Dz * g.Dz);
  g.Dr = length;
  g.dirx = g.Dx / length; g.diry = g.Dy / length; g.dirz = g.Dz / length;
  g.sT = s.f[LDSIM_TRAN_DIFF][seg];
  g.sL = s.f[LDSIM_LONG_DIFF][seg];
  g.q = s.f[LDSIM_N_ELECTRONS][seg];
  g.sT2 = g.sT * g.sT; g.sL2 = g.sL * g.sL;
  g.s3 = f32mul(f32mul(g.sT, g.sT, nf32), g.sL, nf32);
  g.rT = g.sT2 / f32mul(g.sT, g.sT, nf32); g.rL = g.sL2 / f32mul(g.sL, g.sL, nf32);
  double impact = fmax(sqrt((5 * g.sT) * (5 * g.sT) + (5 * g.sT) * (5 * g.sT)),
                       sqrt(c->pixel_pitch * c->pixel_pitch + c->pixel_pitch * c->pixel_pitch) / 2) * 2;
  double z_poca, z_s, z_e;
  z_interval(start, end, x_p, y_p, impact, z_poca, z_s, z_e, nf32);
  if (z_poca == 0) return;
  g.x_p = x_p; g.y_p = y_p;
  g.sx = start[0]; g.sy = start[1]; g.sz = start[2];
  g.z_start_int = z_s - 4 * g.sL;
  double z_end_int = z_e + 4 * g.sL;
  double l0 = (z_s - start[2]) / g.dirz, l1 = (z_e - start[2]) / g.dirz;
  g.x_start = start[0] + l0 * g.dirx; g.y_start = start[1] + l0 * g.diry;
  double x_end = start[0] + l1 * g.dirx, y_end = start[1] + l1 * g.diry;
  const int NS = c->sampled_points;
  g.y_step = (fabs(y_end - g.y_start) + 8 * g.sT) / (NS - 1);
  g.x_step = (fabs(x_end - g.x_start) + 8 * g.sT) / (NS - 1);
  double z_sampling = c->time_sampling / 2.;
  double zs_f = ceil(fabs(z_end_int - g.z_start_int) / z_sampling);
  if (!(zs_f < 1.0e7)) return;   // NaN / absurd geometry: the reference's behaviour is undefined
  g.z_steps = (int)fmax((double)NS, zs_f);
  g.z_step = (z_end_int - g.z_start_int) / (g.z_steps - 1);
  g.t_start = py_round((s.f[LDSIM_T_START][seg] - s.f[LDSIM_T0_START][seg] - c->time_padding) / c->time_sampling) *
              c->time_sampling;
  g.z_anode = c->tpc_borders[tplane][2][0];
  g.sgnx = sgn(g.dirx); g.sgny = sgn(g.diry);
  g.dV = fabs(g.x_step) * fabs(g.y_step) * fabs(g.z_step);
  // anything non-finite -> no signal (reference: NaN propagation / undefined)
  double chk = g.x_step + g.y_step + g.z_step + g.x_start + g.y_start + g.t_start + g.q + g.Dr;
  if (!(fabs(chk) < 1e300) || !(g.sT > 0) || !(g.sL > 0) || !(g.Dr > 0)) return;
  g.ok = true;
}

// exact reference predicates for one slice at one tick (detsim.py:418-428 + get_closest_waveform :213)
__device__ __forceinline__ bool slice_valid_at(const LdsimConsts* c, double t_start, double t0, int it, int64_t& k) {
  double time_tick = t_start + it * c->time_sampling;
  k = (int64_t)py_round((time_tick - t0) / c->response_sampling);
  if (time_tick < 0.) return false;
  return (t0 < time_tick) && (time_tick < t0 + c->time_window);
}


// LDS row layout: one pad double per 8*M elements so that the 8*M-element lane stride of the sliding window
// becomes 8*M+1 doubles -> conflict-free ds_read_b64 (MI355X_MICROARCH.md, LDS banking)
template <int M>
__device__ __forceinline__ int rpos(int r) { return r + (r >> (M == 1 ? 3 : 4)); }

