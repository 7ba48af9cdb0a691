/*
 * ldsim_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C (f64, scalar, literal loop order) restatement of the reference's charge/light hot
 * path, used as the parity checker by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg.  Nothing in the product path may call into this file.
 *
 * Each function cites the reference lines it follows (paths relative to /root/reference).
 * Pinned against golden vectors generated from the reference's own source
 * (oracle/gen_golden.py writes the .npz files under tests/golden; checked by tests/test_oracle_golden.py).
 *
 * The only structural liberty taken: in o_tracks_current the tick-independent charge
 * rho(x,y,z)*dV is evaluated once per sample point and reused for every tick (the reference
 * re-evaluates it per tick); values and summation order per tick are unchanged.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/ldsim.h"

typedef struct {
  double x_start, y_start, z_start, x_end, y_end, z_end, x, y, z, dEdx, dE, t, t_start, t_end, t0,
      t0_start, t0_end, n_electrons, n_photons, long_diff, tran_diff;
  int32_t pixel_plane;
  int32_t pad_;
} OTrack;

/* Numba types f32 (op) f32 as f32.  With the 152-byte HDF5 schema (f4 coordinates and sigmas) this makes a few
 * sub-expressions of tracks_current single precision (detsim.py:387, :74-79, :116-118, :141).  g_numba_f32 = 1
 * restates exactly those spots in float; 0 = all-f64 (what the reference computes for f8 records). */
static int g_numba_f32 = 0;
void o_set_numba_f32(int on) { g_numba_f32 = on; }
#define F32SUB(a, b) (g_numba_f32 ? (double)((float)(a) - (float)(b)) : ((a) - (b)))
#define F32MUL(a, b) (g_numba_f32 ? (double)((float)(a) * (float)(b)) : ((a) * (b)))
#define F32DIV(a, b) (g_numba_f32 ? (double)((float)(a) / (float)(b)) : ((a) / (b)))

/* ---- Python / Numba scalar semantics -------------------------------------------------------- */
static double py_round(double x) { return nearbyint(x); } /* round-half-even, FE_TONEAREST */

/* Python float floor division (CPython float_divmod; Numba follows the same algorithm) */
static double py_floordiv(double vx, double wx) {
  double mod = fmod(vx, wx);
  double div = (vx - mod) / wx;
  if (mod != 0.0) {
    if ((wx < 0) != (mod < 0)) {
      mod += wx;
      div -= 1.0;
    }
  }
  double fd;
  if (div != 0.0) {
    fd = floor(div);
    if (div - fd > 0.5) fd += 1.0;
  } else {
    fd = copysign(0.0, vx / wx);
  }
  return fd;
}

/* Python integer floor-div / mod for possibly negative ids */
static int64_t ifloordiv(int64_t a, int64_t b) {
  int64_t q = a / b;
  if ((a % b != 0) && ((a < 0) != (b < 0))) q -= 1;
  return q;
}
static int64_t ifloormod(int64_t a, int64_t b) { return a - ifloordiv(a, b) * b; }

static double narrow(double v, int code) {
  switch (code) {
    case LDSIM_F4: return (double)(float)v;
    case LDSIM_F8: return v;
    case LDSIM_I4: return (double)(int32_t)v;
    case LDSIM_U4: return (double)(uint32_t)v;
    case LDSIM_I8: return (double)(int64_t)v;
    case LDSIM_U8: return (double)(uint64_t)v;
  }
  return v;
}

/* ---- a2: quenching.quench  (larndsim/quenching.py:11-44) ------------------------------------- */
int o_quench(OTrack* tr, int64_t n, const LdsimConsts* c, int mode, const int32_t* store) {
  int bad = 0;
  for (int64_t i = 0; i < n; i++) {
    double dEdx = tr[i].dEdx, dE = tr[i].dE, recomb = 0;
    if (mode == 1) { /* BOX: quenching.py:32-33 */
      double csi = c->box_beta * dEdx / (c->e_field * c->lar_density);
      double r = log(c->box_alpha + csi) / csi;
      recomb = (r > 0) ? r : 0; /* Python max(0, r) */
    } else if (mode == 2) { /* BIRKS: quenching.py:36 */
      recomb = c->birks_ab / (1 + c->birks_kb * dEdx / (c->e_field * c->lar_density));
    } else {
      return -1;
    }
    if (isnan(recomb)) { bad = 1; continue; }
    tr[i].n_electrons = narrow(recomb * dE / c->w_ion, store[LDSIM_N_ELECTRONS]);            /* :43 */
    tr[i].n_photons = narrow((dE / c->w_ph - tr[i].n_electrons) * c->scint_prescale,         /* :44 */
                             store[LDSIM_N_PHOTONS]);
  }
  return bad ? -2 : 0;
}

/* ---- a3: drifting.drift  (larndsim/drifting.py:11-58) ---------------------------------------- */
int o_drift(OTrack* tr, int64_t n, const LdsimConsts* c, const int32_t* store) {
  for (int64_t i = 0; i < n; i++) {
    OTrack* t = &tr[i];
    int32_t plane = c->default_plane_index;
    for (int ip = 0; ip < c->n_tpc; ip++) { /* :34-39 */
      const double(*p)[2] = c->tpc_borders[ip];
      double zlo = fmin(p[2][1] - 2e-2, p[2][0] - 2e-2), zhi = fmax(p[2][1] + 2e-2, p[2][0] + 2e-2);
      if (p[0][0] - 2e-2 <= t->x && t->x <= p[0][1] + 2e-2 && p[1][0] - 2e-2 <= t->y &&
          t->y <= p[1][1] + 2e-2 && zlo <= t->z && t->z <= zhi) {
        plane = ip;
        break;
      }
    }
    t->pixel_plane = plane;
    if (plane != c->default_plane_index) { /* :43-58 */
      double z_anode = c->tpc_borders[plane][2][0];
      double drift_distance = fabs(t->z - z_anode);
      double drift_start = fabs(fmin(t->z_start, t->z_end) - z_anode);
      double drift_end = fabs(fmax(t->z_start, t->z_end) - z_anode);
      double drift_time = drift_distance / c->v_drift;
      double lifetime_red = exp(-drift_time / c->electron_lifetime);
      t->n_electrons = narrow(t->n_electrons * lifetime_red, store[LDSIM_N_ELECTRONS]);
      t->long_diff = narrow(sqrt(drift_time * 2 * c->long_diff), store[LDSIM_LONG_DIFF]);
      t->tran_diff = narrow(sqrt(drift_time * 2 * c->tran_diff), store[LDSIM_TRAN_DIFF]);
      t->t = narrow(t->t + (drift_time + t->t0), store[LDSIM_T]);
      t->t_start = narrow(t->t_start + (fmin(drift_start, drift_end) / c->v_drift + t->t0), store[LDSIM_T_START]);
      t->t_end = narrow(t->t_end + (fmax(drift_start, drift_end) / c->v_drift + t->t0), store[LDSIM_T_END]);
    }
  }
  return 0;
}

/* ---- a4: pixel2id / id2pixel  (larndsim/pixels_from_track.py:13-41) --------------------------- */
static int64_t pixel2id(const LdsimConsts* c, int64_t px, int64_t py, int64_t plane) {
  return px + c->n_pixels[0] * (py + c->n_pixels[1] * plane);
}
static void id2pixel(const LdsimConsts* c, int64_t pid, int64_t* px, int64_t* py, int64_t* plane) {
  *px = ifloormod(pid, c->n_pixels[0]);
  *py = ifloormod(ifloordiv(pid, c->n_pixels[0]), c->n_pixels[1]);
  *plane = ifloordiv(pid, (int64_t)c->n_pixels[0] * c->n_pixels[1]);
}

static int in_range(const LdsimConsts* c, int64_t x, int64_t y, int64_t plane) {
  return 0 <= x && x < c->n_pixels[0] && 0 <= y && y < c->n_pixels[1] && 0 <= plane && plane < c->n_tpc;
}

static void start_end_pixels(const LdsimConsts* c, const OTrack* t, int64_t* x0, int64_t* y0, int64_t* x1,
                             int64_t* y1) {
  const double(*b)[2] = c->tpc_borders[t->pixel_plane];
  *x0 = (int64_t)py_floordiv(t->x_start - b[0][0], c->pixel_pitch);
  *y0 = (int64_t)py_floordiv(t->y_start - b[1][0], c->pixel_pitch);
  *x1 = (int64_t)py_floordiv(t->x_end - b[0][0], c->pixel_pitch);
  *y1 = (int64_t)py_floordiv(t->y_end - b[1][0], c->pixel_pitch);
}

/* Bresenham-without-diagonals walk (pixels_from_track.py:111-199). out may be NULL (count only). */
static int64_t walk_pixels(const LdsimConsts* c, int64_t x0, int64_t y0, int64_t x1, int64_t y1, int64_t plane,
                           int32_t* out, int64_t cap) {
  int64_t dx = llabs(x1 - x0), sx = x0 < x1 ? 1 : -1;
  int64_t dy = -llabs(y1 - y0), sy = y0 < y1 ? 1 : -1;
  int64_t err = dx + dy, n = 0, i = 0;
  if (in_range(c, x0, y0, plane)) {
    if (out && i < cap) out[i] = (int32_t)pixel2id(c, x0, y0, plane);
    n++;
  }
  while (x0 != x1 || y0 != y1) {
    i++;
    int64_t e2 = 2 * err;
    if (e2 - dy > dx - e2) {
      err += dy;
      x0 += sx;
    } else {
      err += dx;
      y0 += sy;
    }
    if (in_range(c, x0, y0, plane)) {
      if (out && i < cap) out[i] = (int32_t)pixel2id(c, x0, y0, plane);
      n++;
    }
  }
  return n;
}

/* ---- a5: max_pixels  (pixels_from_track.py:43-65) ---------------------------------------------- */
int o_max_pixels(const OTrack* tr, int64_t n, const LdsimConsts* c, int64_t* n_max) {
  for (int64_t i = 0; i < n; i++) {
    if (tr[i].pixel_plane < 0 || tr[i].pixel_plane >= c->n_tpc) continue; /* reference indexes OOB here */
    int64_t x0, y0, x1, y1;
    start_end_pixels(c, &tr[i], &x0, &y0, &x1, &y1);
    int64_t k = walk_pixels(c, x0, y0, x1, y1, tr[i].pixel_plane, NULL, 0);
    if (k > *n_max) *n_max = k;
  }
  return 0;
}

/* ring-distance code (pixels_from_track.py:246-269) */
static int32_t ring_code(int x_r, int y_r) {
  int dx = abs(x_r), dy = abs(y_r), dmax = dx > dy ? dx : dy, dmin = dx > dy ? dy : dx, dsum = dmax + dmin;
  if (dsum > 4) return -1;
  if (dsum <= 1) return dsum;
  if (dsum == 2) return dmax == 1 ? 2 : 3;
  if (dsum == 3) return dmax == 2 ? 4 : 5;
  return dmax == 2 ? 6 : (dmax == 3 ? 7 : 8);
}

/* ---- a6: get_pixels  (pixels_from_track.py:67-109,157-272) ------------------------------------- */
int o_get_pixels(const OTrack* tr, int64_t n, const LdsimConsts* c, int radius, int32_t* active, int64_t max_active,
                 int32_t* neigh, int32_t* nrad, int64_t P, double* n_list) {
  for (int64_t it = 0; it < n; it++) {
    const OTrack* t = &tr[it];
    int32_t* act = active + it * max_active;
    int32_t* ng = neigh + it * P;
    int32_t* nr = nrad + it * P;
    if (t->pixel_plane < 0 || t->pixel_plane >= c->n_tpc) { n_list[it] = 0; continue; }
    int64_t x0, y0, x1, y1;
    start_end_pixels(c, t, &x0, &y0, &x1, &y1);
    walk_pixels(c, x0, y0, x1, y1, t->pixel_plane, act, max_active);
    int64_t count = 0;
    for (int64_t p = 0; p < max_active; p++) {
      if (act[p] == -1) continue;
      for (int x_r = -radius; x_r <= radius; x_r++)
        for (int y_r = -radius; y_r <= radius; y_r++) {
          int64_t ax, ay, pl;
          id2pixel(c, act[p], &ax, &ay, &pl);
          int64_t nx = ax + x_r, ny = ay + y_r;
          if (!in_range(c, nx, ny, pl)) continue;
          int32_t np_ = (int32_t)pixel2id(c, nx, ny, pl);
          int uniq = 1;
          for (int64_t q = 0; q < P; q++)
            if (ng[q] == np_) { uniq = 0; break; }
          if (uniq && count < P) {
            ng[count] = np_;
            nr[count] = ring_code(x_r, y_r);
            count++;
          }
        }
    }
    n_list[it] = (double)count;
  }
  return 0;
}

/* ---- a8: time_intervals  (larndsim/detsim.py:18-40) -------------------------------------------- */
int o_time_intervals(const OTrack* tr, int64_t n, const LdsimConsts* c, double* starts, int64_t* tmax) {
  for (int64_t i = 0; i < n; i++) {
    double t_end = py_round((tr[i].t_end + 1) / c->time_sampling) * c->time_sampling;
    double t_start = py_round((tr[i].t_start - c->time_padding) / c->time_sampling) * c->time_sampling;
    double t_length = t_end - t_start;
    starts[i] = t_start;
    int64_t k = (int64_t)ceil(t_length / c->time_sampling);
    if (k > *tmax) *tmax = k;
  }
  return 0;
}

/* ---- a10: z_interval  (detsim.py:42-112) ------------------------------------------------------- */
static void z_interval(const double sp[3], const double ep[3], double x_p, double y_p, double tol, double* z_poca,
                       double* z_lo, double* z_hi) {
  const double *start, *end;
  *z_poca = *z_lo = *z_hi = 0;
  if (sp[0] > ep[0]) { start = ep; end = sp; }
  else if (sp[0] < ep[0]) { start = sp; end = ep; }
  else return;
  double xs = start[0], ys = start[1], xe = end[0], ye = end[1];
  double m = F32DIV(F32SUB(ye, ys), F32SUB(xe, xs));
  double q = F32DIV(F32SUB(F32MUL(xe, ys), F32MUL(xs, ye)), F32SUB(xe, xs));
  double a = m, b = -1, cc = q;
  double x_poca = (b * (b * x_p - a * y_p) - F32MUL(a, cc)) / (F32MUL(a, a) + b * b);
  double dx = F32SUB(end[0], start[0]), dy = F32SUB(end[1], start[1]), dz = F32SUB(end[2], start[2]);
  double length = sqrt(dx * dx + dy * dy + dz * dz);
  double dir3[3] = {dx / length, dy / length, dz / length};
  double doca;
  if (x_poca < start[0]) {
    doca = sqrt((x_p - start[0]) * (x_p - start[0]) + (y_p - start[1]) * (y_p - start[1]));
    x_poca = start[0];
  } else if (x_poca > end[0]) {
    doca = sqrt((x_p - end[0]) * (x_p - end[0]) + (y_p - end[1]) * (y_p - end[1]));
    x_poca = end[0];
  } else {
    doca = fabs(a * x_p + b * y_p + cc) / sqrt(F32MUL(a, a) + b * b);
  }
  double zp = start[2] + (x_poca - start[0]) / dir3[0] * dir3[2];
  if (tol > doca) {
    double dxs = F32SUB(xe, xs), dys = F32SUB(ye, ys);
    double length2D = sqrt(dxs * dxs + dys * dys);
    double dir2x = F32SUB(end[0], start[0]) / length2D;
    double deltaL2D = sqrt(tol * tol - doca * doca);
    double x_plus = x_poca + deltaL2D * dir2x;
    double x_minus = x_poca - deltaL2D * dir2x;
    double plusL = (x_plus - start[0]) / dir3[0];
    double minusL = (x_minus - start[0]) / dir3[0];
    double plusZ = start[2] + dir3[2] * plusL;
    double minusZ = start[2] + dir3[2] * minusL;
    *z_poca = zp;
    *z_lo = fmin(minusZ, plusZ);
    *z_hi = fmax(minusZ, plusZ);
  }
}

/* ---- a11: rho  (detsim.py:114-159) ------------------------------------------------------------- */
static double rho(double x, double y, double z, double q, const double start[3], const double sg[3],
                  const double seg[3]) {
  double Dx = seg[0], Dy = seg[1], Dz = seg[2];
  double Dr = sqrt(Dx * Dx + Dy * Dy + Dz * Dz);
  double a = ((Dx / Dr) * (Dx / Dr) / (2 * sg[0] * sg[0]) + (Dy / Dr) * (Dy / Dr) / (2 * sg[1] * sg[1]) +
              (Dz / Dr) * (Dz / Dr) / (2 * sg[2] * sg[2]));
  double factor = q / Dr / (F32MUL(F32MUL(sg[0], sg[1]), sg[2]) * sqrt(8 * M_PI * M_PI * M_PI));
  double sqrt_a_2 = 2 * sqrt(a);
  double b = -((x - start[0]) / F32MUL(sg[0], sg[0]) * (seg[0] / Dr) + (y - start[1]) / F32MUL(sg[1], sg[1]) * (seg[1] / Dr) +
               (z - start[2]) / F32MUL(sg[2], sg[2]) * (seg[2] / Dr));
  double delta = (x - start[0]) * (x - start[0]) / (2 * sg[0] * sg[0]) +
                 (y - start[1]) * (y - start[1]) / (2 * sg[1] * sg[1]) +
                 (z - start[2]) * (z - start[2]) / (2 * sg[2] * sg[2]);
  double integral = sqrt(M_PI) * (-erf(b / sqrt_a_2) + erf((b + 2 * a * Dr) / sqrt_a_2)) / sqrt_a_2;
  double expo = 0;
  if (factor != 0 && integral != 0) expo = exp(b * b / (4 * a) - delta + log(factor) + log(integral));
  return expo;
}

double o_rho(double x, double y, double z, double q, const double* start, const double* sigmas, const double* segment) {
  return rho(x, y, z, q, start, sigmas, segment);
}

static double signf(double x) { return x >= 0 ? 1.0 : -1.0; } /* detsim.py:455-466 */

/* ---- a9: tracks_current  (detsim.py:351-453; helpers :161-218) -------------------------------- */
// pairs are independent: OpenMP over (segment, pixel) only parallelises the cpu_baseline timing
int o_tracks_current(float* signals, const int32_t* pixels, const OTrack* tr, int64_t S, int64_t P, int64_t T,
                     const double* response, int64_t ni, int64_t nj, int64_t nk, const LdsimConsts* c) {
  const int NS = c->sampled_points;
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t pr = 0; pr < S * P; pr++) {
    {
      const int64_t itrk = pr / P, ipix = pr % P;
      const OTrack* t = &tr[itrk];
      int64_t pID = pixels[itrk * P + ipix];
      int64_t px, py, pplane;
      id2pixel(c, pID, &px, &py, &pplane);
      if (!(px >= 0 && py >= 0)) continue;
      /* get_pixel_coordinates: negative plane index wraps like a Python/Numba array index */
      int64_t bplane = pplane < 0 ? pplane + c->n_tpc : pplane;
      if (bplane < 0 || bplane >= c->n_tpc) continue;
      if (t->pixel_plane < 0 || t->pixel_plane >= c->n_tpc) continue;
      const double(*pb)[2] = c->tpc_borders[bplane];
      double x_p = px * c->pixel_pitch + pb[0][0];
      double y_p = py * c->pixel_pitch + pb[1][0];
      x_p += c->pixel_pitch / 2;
      y_p += c->pixel_pitch / 2;
      double start[3], end[3];
      if (t->z_start < t->z_end) {
        start[0] = t->x_start; start[1] = t->y_start; start[2] = t->z_start;
        end[0] = t->x_end; end[1] = t->y_end; end[2] = t->z_end;
      } else {
        end[0] = t->x_start; end[1] = t->y_start; end[2] = t->z_start;
        start[0] = t->x_end; start[1] = t->y_end; start[2] = t->z_end;
      }
      double seg[3] = {F32SUB(end[0], start[0]), F32SUB(end[1], start[1]), F32SUB(end[2], start[2])};
      double length = sqrt(seg[0] * seg[0] + seg[1] * seg[1] + seg[2] * seg[2]);
      double dir[3] = {seg[0] / length, seg[1] / length, seg[2] / length};
      double sg[3] = {t->tran_diff, t->tran_diff, t->long_diff};
      double impact = fmax(sqrt((5 * sg[0]) * (5 * sg[0]) + (5 * sg[1]) * (5 * sg[1])),
                           sqrt(c->pixel_pitch * c->pixel_pitch + c->pixel_pitch * c->pixel_pitch) / 2) * 2;
      double z_poca, z_s, z_e;
      z_interval(start, end, x_p, y_p, impact, &z_poca, &z_s, &z_e);
      if (z_poca == 0) continue;
      double z_start_int = z_s - 4 * sg[2], z_end_int = z_e + 4 * sg[2];
      double l0 = (z_s - start[2]) / dir[2], l1 = (z_e - start[2]) / dir[2];
      double x_start = start[0] + l0 * dir[0], y_start = start[1] + l0 * dir[1];
      double x_end = start[0] + l1 * dir[0], y_end = start[1] + l1 * dir[1];
      double y_step = (fabs(y_end - y_start) + 8 * sg[1]) / (NS - 1);
      double x_step = (fabs(x_end - x_start) + 8 * sg[0]) / (NS - 1);
      double z_sampling = c->time_sampling / 2.;
      double zs_f = ceil(fabs(z_end_int - z_start_int) / z_sampling);
      if (!(zs_f < 1e7)) continue; /* NaN / absurd: reference behaviour undefined */
      int64_t z_steps = (int64_t)fmax((double)NS, zs_f);
      double z_step = (z_end_int - z_start_int) / (z_steps - 1);
      double t_start = py_round((t->t_start - t->t0_start - c->time_padding) / c->time_sampling) * c->time_sampling;
      double z_anode = c->tpc_borders[t->pixel_plane][2][0];

      /* hoisted, tick-independent part */
      double* charge = (double*)malloc(sizeof(double) * z_steps * NS * NS);
      int32_t* ii = (int32_t*)malloc(sizeof(int32_t) * NS);
      int32_t* jj = (int32_t*)malloc(sizeof(int32_t) * NS);
      double* t0s = (double*)malloc(sizeof(double) * z_steps);
      double xs_[64], ys_[64];
      for (int ix = 0; ix < NS; ix++) {
        double x = x_start + signf(dir[0]) * (ix * x_step - 4 * sg[0]);
        double xd = fabs(x_p - x);
        xs_[ix] = x;
        ii[ix] = (xd > c->response_bin_size * ni) ? -2 : (int32_t)py_round(xd / c->response_bin_size - 0.5);
        double y = y_start + signf(dir[1]) * (ix * y_step - 4 * sg[1]);
        double yd = fabs(y_p - y);
        ys_[ix] = y;
        jj[ix] = (yd > c->response_bin_size * nj) ? -2 : (int32_t)py_round(yd / c->response_bin_size - 0.5);
      }
      for (int64_t iz = 0; iz < z_steps; iz++) {
        double z = z_start_int + iz * z_step;
        t0s[iz] = fabs(z - z_anode) / c->v_drift - c->time_window;
        for (int ix = 0; ix < NS; ix++)
          for (int iy = 0; iy < NS; iy++)
            charge[(iz * NS + ix) * NS + iy] =
                (ii[ix] == -2 || jj[iy] == -2) ? 0.0
                    : rho(xs_[ix], ys_[iy], z, t->n_electrons, start, sg, seg) * fabs(x_step) * fabs(y_step) * fabs(z_step);
      }
      float* out = signals + (itrk * P + ipix) * T;
      for (int64_t it = 0; it < T; it++) {
        double time_tick = t_start + it * c->time_sampling;
        if (time_tick < 0.) continue;
        double total = 0;
        int wrote = 0;
        for (int64_t iz = 0; iz < z_steps; iz++) {
          double t0 = t0s[iz];
          if (!(t0 < time_tick && time_tick < t0 + c->time_window)) continue;
          int64_t k = (int64_t)py_round((time_tick - t0) / c->response_sampling);
          for (int ix = 0; ix < NS; ix++) {
            if (ii[ix] == -2) continue;
            for (int iy = 0; iy < NS; iy++) {
              if (jj[iy] == -2) continue;
              double w = 0;
              if (0 <= ii[ix] && ii[ix] < ni && 0 <= jj[iy] && jj[iy] < nj && 0 <= k && k < nk)
                w = response[((int64_t)ii[ix] * nj + jj[iy]) * nk + k];
              total += w * charge[(iz * NS + ix) * NS + iy];
            }
            wrote = 1;
          }
        }
        if (wrote) out[it] = (float)total;
      }
      free(charge); free(ii); free(jj); free(t0s);
    }
  }
  return 0;
}

/* ---- a13: get_track_pixel_map2  (detsim.py:564-607) ------------------------------------------- */
int o_track_pixel_map(int64_t* map, const int32_t* unique_pix, int64_t U, const int32_t* pixels,
                      const int32_t* dist, int64_t S, int64_t P, int max_distance, int64_t M) {
  for (int64_t u = 0; u < U; u++) {
    int32_t upix = unique_pix[u];
    int64_t* row = map + u * M;
    for (int target = 0; target < max_distance; target++)
      for (int64_t itrk = 0; itrk < S; itrk++)
        for (int64_t ipix = 0; ipix < P; ipix++) {
          if (upix != pixels[itrk * P + ipix]) continue;
          if (dist[itrk * P + ipix] == target) {
            int64_t imap = 0;
            while (imap < M) {
              if (row[imap] == itrk) { imap = -1; break; }
              if (row[imap] == -1) break;
              imap++;
            }
            if (imap >= 0 && imap < M) row[imap] = itrk;
          }
          break;
        }
  }
  return 0;
}

/* ---- a14: sum_pixel_signals  (detsim.py:468-527) ----------------------------------------------- */
int o_sum_pixel_signals(double* pixels_signals, const float* signals, const double* track_starts,
                        const int64_t* pixel_index_map, const int64_t* track_pixel_map,
                        double* pixels_tracks_signals /* may be NULL */, double* overflow, int64_t S, int64_t P,
                        int64_t T, int64_t NT, int64_t M, const LdsimConsts* c) {
  for (int64_t itrk = 0; itrk < S; itrk++)
    for (int64_t ipix = 0; ipix < P; ipix++) {
      int64_t pidx = pixel_index_map[itrk * P + ipix];
      int64_t start_tick = (int64_t)py_round(track_starts[itrk] / c->time_sampling);
      if (pidx < 0) continue;
      int64_t counter = -99;
      for (int64_t k = 0; k < M; k++)
        if (itrk == track_pixel_map[pidx * M + k]) { counter = k; break; }
      if (counter < 0) { overflow[pidx] = 1; continue; }
      for (int64_t itick = 0; itick < T; itick++) {
        int64_t itime = start_tick + itick;
        if (itime < NT && itime > -1) {
          double v = signals[(itrk * P + ipix) * T + itick];
          pixels_signals[pidx * NT + itime] += v;
          if (pixels_tracks_signals) pixels_tracks_signals[(pidx * NT + itime) * M + counter] += v;
        }
      }
    }
  return 0;
}

/* ---- a15: fee.get_adc_values  (fee.py:517-655) ---------------------------------------------------------------
 * rng = NULL: noise terms 0 (the deterministic path every golden pins).  rng != NULL: rng[ip] is pixel ip's
 * xoroshiro128p state, advanced in place like the reference's rng_states[ip] (restatement above: unpinned). */
/* ---- numba.cuda.random (third-party, module `numba`, setup.py:5 `numba>=0.52`, unpinned; NOT present under /root/reference)
 * Restated from the published algorithm of numba/cuda/random.py: xoroshiro128+ ("xoroshiro128p"), state {s0, s1} u64;
 *   init_xoroshiro128p_state: SplitMix64 of the seed into BOTH words;  next: result = s0 + s1; s1 ^= s0;
 *   s0 = rotl(s0, 55) ^ s1 ^ (s1 << 14); s1 = rotl(s1, 36);  jump: the 2^64-step polynomial {0xbeac0467eba5facb,
 *   0xd86b048b86aa9922};  create_xoroshiro128p_states(n, seed): state 0 = init(seed), state i = state i-1 jumped once;
 *   uniform_float32 = float32((x >> 11) * 2^-53);  normal_float32 = Box-Muller in float32 from two uniforms, second value
 *   discarded:  sqrt(-2 log u1) * cos(2 pi u2).
 * Call sites: fee.py:557,583-584,616-617,621,649; detsim.py:331,336-337; seeding cli/simulate_pixels.py:92-104,396.
 * No reference test pins any RNG-dependent value and Numba cannot run here: PARITY UNPINNED for every noisy output -- the HIP
 * path is tested bit-identical to THIS restatement plus statistical closure. */
typedef struct { uint64_t s0, s1; } ORng;
static uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
uint64_t o_rng_next(ORng* st) {
  uint64_t s0 = st->s0, s1 = st->s1, result = s0 + s1;
  s1 ^= s0;
  st->s0 = rotl64(s0, 55) ^ s1 ^ (s1 << 14);
  st->s1 = rotl64(s1, 36);
  return result;
}
static void rng_jump(ORng* st) {
  static const uint64_t JUMP[2] = {0xbeac0467eba5facbULL, 0xd86b048b86aa9922ULL};
  uint64_t s0 = 0, s1 = 0;
  for (int i = 0; i < 2; i++)
    for (int b = 0; b < 64; b++) {
      if (JUMP[i] & (1ULL << b)) { s0 ^= st->s0; s1 ^= st->s1; }
      o_rng_next(st);
    }
  st->s0 = s0;
  st->s1 = s1;
}
void o_rng_create_states(ORng* states, int64_t n, uint64_t seed) {
  if (n < 1) return;
  uint64_t z = seed + 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z = z ^ (z >> 31);
  states[0].s0 = z;
  states[0].s1 = z;
  for (int64_t i = 1; i < n; i++) {
    states[i] = states[i - 1];
    rng_jump(&states[i]);
  }
}
float o_rng_uniform_f32(ORng* st) { return (float)((double)(o_rng_next(st) >> 11) * (1.0 / 9007199254740992.0)); }
float o_rng_normal_f32(ORng* st) {
  float u1 = o_rng_uniform_f32(st), u2 = o_rng_uniform_f32(st);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

/* ---- 8f row 1: tracks_current_mc (detsim.py:258-348), overlapping_segment (:220-256) ------------------------------------------
 * The reference's 64 tick threads race on rng_states[itrk + ntrk*ipix]; like the HIP kernel this restatement gives every
 * (segment, pixel, tick) its own stream derived from that state (SplitMix64 finaliser of the words offset by the tick):
 * reproducible, statistically equivalent, unpinned.  `states` has S*P entries; each is stepped once at the end. */
static void mc_stream(const ORng* b, uint32_t it, ORng* o) {
  uint64_t z0 = b->s0 + 0x9E3779B97F4A7C15ULL * (uint64_t)(it + 1u);
  uint64_t z1 = b->s1 ^ (0xD1B54A32D192ED03ULL * (uint64_t)(it + 1u));
  z0 = (z0 ^ (z0 >> 30)) * 0xBF58476D1CE4E5B9ULL; z0 = (z0 ^ (z0 >> 27)) * 0x94D049BB133111EBULL; z0 ^= z0 >> 31;
  z1 = (z1 ^ (z1 >> 30)) * 0xBF58476D1CE4E5B9ULL; z1 = (z1 ^ (z1 >> 27)) * 0x94D049BB133111EBULL; z1 ^= z1 >> 31;
  if ((z0 | z1) == 0) z0 = 1;
  o->s0 = z0;
  o->s1 = z1;
}
int o_tracks_current_mc(float* signals, const int32_t* pixels, const OTrack* tr, int64_t S, int64_t P, int64_t T,
                        const double* response, int64_t ni, int64_t nj, int64_t nk, const LdsimConsts* c, ORng* states) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t pr = 0; pr < S * P; pr++) {
    const int64_t itrk = pr / P, ipix = pr % P;
    const OTrack* t = &tr[itrk];
    float* out = signals + pr * T;
    int64_t pID = pixels[pr], px, py, pplane;
    id2pixel(c, pID, &px, &py, &pplane);
    if (!(px >= 0 && py >= 0)) continue;
    int64_t bplane = pplane < 0 ? pplane + c->n_tpc : pplane;
    if (bplane < 0 || bplane >= c->n_tpc || t->pixel_plane < 0 || t->pixel_plane >= c->n_tpc) continue;
    const double(*pb)[2] = c->tpc_borders[bplane];
    double x_p = px * c->pixel_pitch + pb[0][0] + c->pixel_pitch / 2, y_p = py * c->pixel_pitch + pb[1][0] + c->pixel_pitch / 2;
    double st[3], en[3];
    if (t->z_start < t->z_end) {
      st[0] = t->x_start; st[1] = t->y_start; st[2] = t->z_start; en[0] = t->x_end; en[1] = t->y_end; en[2] = t->z_end;
    } else {
      en[0] = t->x_start; en[1] = t->y_start; en[2] = t->z_start; st[0] = t->x_end; st[1] = t->y_end; st[2] = t->z_end;
    }
    double t_start = py_round((t->t_start - t->t0_start - c->time_padding) / c->time_sampling) * c->time_sampling;
    double sx = en[0] - st[0], sy = en[1] - st[1], sz = en[2] - st[2];
    double length = sqrt(sx * sx + sy * sy + sz * sz);
    double dir[3] = {sx / length, sy / length, sz / length};
    double impact = sqrt((double)ni * ni + (double)nj * nj) * c->response_bin_size;
    double dx = x_p - st[0], dy = y_p - st[1], vx = en[0] - st[0], vy = en[1] - st[1];
    double l = sqrt(vx * vx + vy * vy);
    vx /= l; vy /= l;
    double sp = (dx * vx + dy * vy) / l;
    double rx = dx - vx * sp * l, ry = dy - vy * sp * l, rr = sqrt(rx * rx + ry * ry);
    double ns[3], ne[3];
    if (rr > impact) {
      for (int k = 0; k < 3; k++) { ns[k] = st[k]; ne[k] = st[k]; }
    } else {
      double s_plus = sp + sqrt(impact * impact - rr * rr) / l, s_minus = sp - sqrt(impact * impact - rr * rr) / l;
      if (s_plus > 1) s_plus = 1; else if (s_plus < 0) s_plus = 0;
      if (s_minus > 1) s_minus = 1; else if (s_minus < 0) s_minus = 0;
      for (int k = 0; k < 3; k++) {
        ns[k] = st[k] * (1 - s_minus) + en[k] * s_minus;
        ne[k] = st[k] * (1 - s_plus) + en[k] * s_plus;
      }
    }
    double ux = ne[0] - ns[0], uy = ne[1] - ns[1], uz = ne[2] - ns[2];
    double sublen = sqrt(ux * ux + uy * uy + uz * uz);
    if (!(sublen > 0 && length > 0 && sublen < 1e6)) continue;
    double nstep_f = fmax(py_round(sublen / c->min_step_size), 1.0);
    if (!(nstep_f < 2.0e9)) continue;
    int64_t nstep = (int64_t)nstep_f;
    double step = sublen / nstep;
    double charge = t->n_electrons * (sublen / length) / ((double)nstep * c->mc_sample_multiplier);
    double z_anode = c->tpc_borders[t->pixel_plane][2][0];
    const ORng* base = &states[itrk + S * ipix];
    for (int64_t it = 0; it < T; it++) {
      double time_tick = t_start + it * c->time_sampling;
      if (time_tick < 0) continue;
      ORng rs;
      mc_stream(base, (uint32_t)it, &rs);
      double total = 0;
      for (int64_t istep = 0; istep < nstep; istep++)
        for (int m = 0; m < c->mc_sample_multiplier; m++) {
          double x = ns[0] + step * (istep + 0.5) * dir[0], y = ns[1] + step * (istep + 0.5) * dir[1];
          double z = ns[2] + step * (istep + 0.5) * dir[2];
          z += (double)o_rng_normal_f32(&rs) * t->long_diff;
          double t0 = fabs(z - z_anode) / c->v_drift - c->time_window;
          if (!(t0 < time_tick && time_tick < t0 + c->time_window)) continue;
          x += (double)o_rng_normal_f32(&rs) * t->tran_diff;
          y += (double)o_rng_normal_f32(&rs) * t->tran_diff;
          double xd = fabs(x_p - x), yd = fabs(y_p - y);
          if (xd > c->response_bin_size * ni) continue;
          if (yd > c->response_bin_size * nj) continue;
          int64_t i = (int64_t)py_round(xd / c->response_bin_size - 0.5), j = (int64_t)py_round(yd / c->response_bin_size - 0.5);
          int64_t k = (int64_t)py_round((time_tick - t0) / c->response_sampling);
          if (i >= 0 && i < ni && j >= 0 && j < nj && k >= 0 && k < nk) total += charge * response[(i * nj + j) * nk + k];
        }
      out[it] = (float)total;
    }
  }
  for (int64_t i = 0; i < S * P; i++) o_rng_next(&states[i]);
  return 0;
}

int o_get_adc_values_rng(const double* pixels_signals, const double* pixels_signals_tracks, const double* time_ticks,
                         int64_t n_time_ticks, double* adc_list, double* adc_ticks_list, double time_padding,
                         double* current_fractions, const double* thresholds, int64_t U, int64_t NT, int64_t M,
                         const LdsimConsts* c, ORng* rng);
int o_get_adc_values(const double* pixels_signals, const double* pixels_signals_tracks /* [U][NT][M] or NULL */,
                     const double* time_ticks, int64_t n_time_ticks, double* adc_list, double* adc_ticks_list,
                     double time_padding, double* current_fractions /* [U][A][M] or NULL */,
                     const double* thresholds, int64_t U, int64_t NT, int64_t M, const LdsimConsts* c) {
  return o_get_adc_values_rng(pixels_signals, pixels_signals_tracks, time_ticks, n_time_ticks, adc_list, adc_ticks_list,
                              time_padding, current_fractions, thresholds, U, NT, M, c, NULL);
}
#define NORMAL(ip) (rng ? (double)o_rng_normal_f32(&rng[ip]) : 0.0)
int o_get_adc_values_rng(const double* pixels_signals, const double* pixels_signals_tracks, const double* time_ticks,
                         int64_t n_time_ticks, double* adc_list, double* adc_ticks_list, double time_padding,
                         double* current_fractions, const double* thresholds, int64_t U, int64_t NT, int64_t M,
                         const LdsimConsts* c, ORng* rng) {
  const int64_t A = c->max_adc_values;
  const double dt = c->time_sampling, rt = c->buffer_risetime;
  for (int64_t ip = 0; ip < U; ip++) {
    const double* curre = pixels_signals + ip * NT;
    const double* trk = pixels_signals_tracks ? pixels_signals_tracks + ip * NT * M : NULL;
    double* frac = current_fractions ? current_fractions + ip * A * M : NULL;
    int64_t ic = 0, iadc = 0, adc_busy = 0, last_reset = 0;
    double true_q = 0, q_sum = NORMAL(ip) * c->reset_noise_charge;
    while (ic < NT || adc_busy > 0) {
      if (iadc >= A) break;
      double q = 0;
      if (rt > 0) {
        int64_t cs = (int64_t)floor(ic - 10 * rt / dt);
        if (cs < last_reset) cs = last_reset;
        int64_t ce = ic + 1 < NT ? ic + 1 : NT;
        for (int64_t jc = cs; jc < ce; jc++) {
          double w = exp((jc - ic) * dt / rt) * (1 - exp(-dt / rt));
          q += curre[jc] * dt * w;
          if (frac && trk)
            for (int64_t k = 0; k < M; k++) frac[iadc * M + k] += trk[jc * M + k] * dt * w;
        }
      } else if (ic < NT) {
        q += curre[ic] * dt;
        if (frac && trk)
          for (int64_t k = 0; k < M; k++) frac[iadc * M + k] += trk[ic * M + k] * dt;
      }
      q_sum += q;
      true_q += q;
      double q_noise = NORMAL(ip) * c->uncorrelated_noise_charge;
      double disc_noise = NORMAL(ip) * c->discriminator_noise;
      if (adc_busy > 0) adc_busy--;
      if (q_sum + q_noise >= thresholds[ip] + disc_noise && adc_busy == 0) {
        int64_t interval = (int64_t)py_round((3 * c->clock_cycle + c->adc_hold_delay * c->clock_cycle) / dt);
        int64_t integrate_end = ic + interval;
        ic++;
        while (ic <= integrate_end) {
          q = 0;
          if (rt > 0) {
            int64_t cs = (int64_t)floor(ic - 10 * rt / dt);
            if (cs < last_reset) cs = last_reset;
            int64_t ce = ic + 1 < NT ? ic + 1 : NT;
            for (int64_t jc = cs; jc < ce; jc++) {
              double w = exp((jc - ic) * dt / rt) * (1 - exp(-dt / rt));
              q += curre[jc] * dt * w;
              if (frac && trk)
                for (int64_t k = 0; k < M; k++) frac[iadc * M + k] += trk[jc * M + k] * dt * w;
            }
          } else if (ic < NT) {
            q += curre[ic] * dt;
            if (frac && trk)
              for (int64_t k = 0; k < M; k++) frac[iadc * M + k] += trk[ic * M + k] * dt;
          }
          q_sum += q;
          true_q += q;
          ic++;
        }
        double adc = q_sum + NORMAL(ip) * c->uncorrelated_noise_charge;
        disc_noise = NORMAL(ip) * c->discriminator_noise;
        if (adc < thresholds[ip] + disc_noise) {
          ic += (int64_t)py_round(c->reset_cycles * c->clock_cycle / dt);
          q_sum = NORMAL(ip) * c->reset_noise_charge;
          true_q = 0;
          if (frac)
            for (int64_t k = 0; k < M; k++) frac[iadc * M + k] = 0;
          last_reset = ic;
          continue;
        }
        if (true_q > 0 && frac)
          for (int64_t k = 0; k < M; k++) frac[iadc * M + k] /= true_q;
        adc_list[ip * A + iadc] = adc;
        int64_t crossing = ic < n_time_ticks - 1 ? ic : n_time_ticks - 1;
        int64_t post = ic - crossing > 0 ? ic - crossing : 0;
        adc_ticks_list[ip * A + iadc] = time_ticks[crossing] + time_padding - 2 + post;
        ic += (int64_t)py_round(c->reset_cycles * c->clock_cycle / dt);
        last_reset = ic;
        adc_busy = (int64_t)py_round(c->adc_busy_delay * c->clock_cycle / dt);
        q_sum = NORMAL(ip) * c->reset_noise_charge;
        true_q = 0;
        iadc++;
        continue;
      }
      ic++;
    }
  }
  return 0;
}

/* ---- a16: fee.digitize  (fee.py:499-515) ------------------------------------------------------- */
int o_digitize(const double* integral, int64_t n, const double* gain_list, double* out, const LdsimConsts* c) {
  const double mV = 1e-3 * (1e-6 * 1.0), e = 1.0; /* consts/units.py: mV = 1e-3*volt, volt = 1e-6*megavolt */
  for (int64_t i = 0; i < n; i++) {
    double gain = gain_list ? gain_list[i] : c->gain * mV / e;
    double v = integral[i] * gain + c->v_pedestal * mV - c->v_cm * mV;
    v = v > 0 ? v : 0;
    v = nearbyint(v * c->adc_counts / (c->v_ref * mV - c->v_cm * mV));
    out[i] = v < c->adc_counts - 1 ? v : c->adc_counts - 1;
  }
  return 0;
}

/* ---- a17: lightLUT.get_voxel + calculate_light_incidence  (lightLUT.py:15-136) ---------------- */
int o_light_incidence(const OTrack* tr, int64_t n, const float* vis, const float* t0lut, int nx, int ny, int nz,
                      int ndet, const double* eff, const int32_t* ch_to_tpc, int n_out, float* n_photons_det,
                      float* t0_det, int32_t* voxel, const LdsimConsts* c) {
  const double ns = 1.0, mus = 1e-6 * 1e9;
  for (int64_t it = 0; it < n; it++) {
    const OTrack* t = &tr[it];
    int32_t itpc = t->pixel_plane;
    if (itpc == c->default_plane_index) continue;
    int32_t imod = itpc / 2;
    const double(*b)[2] = c->tpc_borders[itpc];
    int is_even = b[2][1] > b[2][0];
    double x_min = b[0][0] - 2e-2, x_max = b[0][1] + 2e-2, y_min = b[1][0] - 2e-2, y_max = b[1][1] + 2e-2;
    double z_min = b[2][0] - 2e-2, z_max = b[2][1] + 2e-2;
    int i = is_even ? (int)((t->x - x_min) / (x_max - x_min) * nx) : (int)((x_max - t->x) / (x_max - x_min) * nx);
    int j = (int)((y_max - t->y) / (y_max - y_min) * ny);
    int k = (int)((t->z - z_min) / (z_max - z_min) * nz);
    i = i < 0 ? 0 : (i > nx - 1 ? nx - 1 : i);
    j = j < 0 ? 0 : (j > ny - 1 ? ny - 1 : j);
    k = k < 0 ? 0 : (k > nz - 1 ? nz - 1 : k);
    voxel[it * 3 + 0] = i; voxel[it * 3 + 1] = j; voxel[it * 3 + 2] = k;
    int64_t vbase = (((int64_t)i * ny + j) * nz + k) * ndet;
    int channel_offset = (n_out < c->n_op_channel) ? n_out * imod : 0;
    for (int o = 0; o < n_out; o++) {
      int op = o + channel_offset, li = o % ndet;
      double v = (double)vis[vbase + li] * (ch_to_tpc[op] == itpc ? 1 : 0);
      n_photons_det[it * n_out + o] = (float)(eff[op] * v * t->n_photons);
      if (c->light_trig_mode == 0)
        t0_det[it * n_out + o] = (float)(((double)t0lut[vbase + li] * ns + t->t0 * mus) / mus);
    }
  }
  return 0;
}

/* ---- a18: light_sim.sum_light_signals  (light_sim.py:58-129) ----------------------------------- */
int o_sum_light_signals(const OTrack* tr, int64_t n, const int32_t* voxel, const int64_t* track_id,
                        const float* n_photons_det, int n_inc, const int32_t* op_channel, int n_det,
                        const float* t0_avg, const float* time_dist, int nx, int ny, int nz, int ndet_lut, int nprof,
                        double start_time, const int32_t* sorted_indices, int64_t n_ticks, float* out,
                        int64_t* true_id, double* true_ph, int max_truth, const LdsimConsts* c) {
  const double ns = 1.0, mus = 1e-6 * 1e9, tick = c->light_tick_size;
  (void)nx;
  for (int idet = 0; idet < n_det; idet++)
    for (int64_t itick = 0; itick < n_ticks; itick++) {
      double st = itick * tick + start_time, en = st + tick;
      int idet_lut = op_channel[idet] % ndet_lut;
      float acc = out[idet * n_ticks + itick]; /* f32 accumulator: the output array is f4 */
      for (int64_t s = 0; s < n; s++) {
        int64_t itrk = sorted_indices[(int64_t)idet * n + s];
        float nph = n_photons_det[itrk * n_inc + op_channel[idet]];
        if (!(nph > 0)) continue;
        const int32_t* vx = voxel + itrk * 3;
        double track_time = tr[itrk].t0;
        double track_end = track_time + nprof * ns / mus;
        if (track_end < st || track_time > en) continue;
        int64_t lbase = ((((int64_t)vx[0] * ny + vx[1]) * nz + vx[2]) * ndet_lut + idet_lut);
        if (c->enable_lut_smearing) {
          const float* prof = time_dist + lbase * nprof;
          for (int ip = 0; ip < nprof; ip++) {
            double pt = track_time + ip * ns / mus;
            if (pt < en && pt > st) {
              double photons = (double)nph * (double)prof[ip] / tick;
              acc = (float)((double)acc + photons);
              if (photons > c->mc_truth_threshold)
                for (int k = 0; k < max_truth; k++) {
                  int64_t* tid = &true_id[((int64_t)idet * n_ticks + itick) * max_truth + k];
                  if (*tid == -1 || *tid == track_id[itrk]) {
                    *tid = track_id[itrk];
                    true_ph[((int64_t)idet * n_ticks + itick) * max_truth + k] += photons;
                    break;
                  }
                }
            }
          }
        } else {
          double pt = track_time + (double)t0_avg[lbase] * ns / mus;
          if (pt < en && pt > st) {
            double photons = (double)nph / tick;
            acc = (float)((double)acc + photons);
            if (photons > c->mc_truth_threshold)
              for (int k = 0; k < max_truth; k++) {
                int64_t* tid = &true_id[((int64_t)idet * n_ticks + itick) * max_truth + k];
                if (*tid == -1 || *tid == track_id[itrk]) {
                  *tid = track_id[itrk];
                  true_ph[((int64_t)idet * n_ticks + itick) * max_truth + k] += photons;
                  break;
                }
              }
          }
        }
      }
      out[idet * n_ticks + itick] = acc;
    }
  return 0;
}

/* ---- light waveform response: light_sim.py:131-184 and :241-337 ------------------------------------
 * Both kernels are one thread per (detector row, tick) adding into caller-initialised f4 arrays (zeros) and truth slots
 * (-1 / 0), term by term in ascending jtick: the f4 store after every term is part of the result. */

/* light_sim.scintillation_model (:131-146) */
static double o_scintillation_model(int64_t time_tick, const LdsimConsts* c) {
  const double tick = c->light_tick_size;
  double p1 = c->singlet_fraction * exp(-(double)time_tick * tick / c->tau_s) * (1 - exp(-tick / c->tau_s));
  double p3 = (1 - c->singlet_fraction) * exp(-(double)time_tick * tick / c->tau_t) * (1 - exp(-tick / c->tau_t));
  return (p1 + p3) * (time_tick >= 0 ? 1.0 : 0.0);
}

/* light_sim.interp (:241-271) */
static double o_interp(double idx, const double* arr, int64_t len, double low, double high) {
  int64_t i0 = (int64_t)floor(idx);
  if (i0 < 0) return low;
  if (i0 > len - 1) return high;
  if ((double)i0 == idx) return arr[i0];
  if (i0 > len - 2) return high;
  double v0 = arr[i0], v1 = arr[i0 + 1];
  return v0 + (v1 - v0) * (idx - (double)i0);
}

/* light_sim.sipm_response_model (:274-300); idet is unused by the reference too */
static double o_sipm_response_model(int64_t time_tick, const double* impulse_model, int64_t n_impulse,
                                    const LdsimConsts* c) {
  if (c->sipm_response_model == 0) {
    double t = (double)time_tick * c->light_tick_size;
    double impulse = (t >= 0 ? 1.0 : 0.0) * exp(-t / c->light_response_time) * sin(t / c->light_oscillation_period);
    impulse /= c->light_oscillation_period * (c->light_response_time * c->light_response_time);
    impulse *= c->light_oscillation_period * c->light_oscillation_period + c->light_response_time * c->light_response_time;
    return impulse * c->light_tick_size;
  }
  double impulse = o_interp((double)time_tick * c->light_tick_size / c->impulse_tick_size, impulse_model, n_impulse, 0, 0);
  impulse /= c->impulse_tick_size / c->light_tick_size;
  return impulse;
}

static int64_t o_conv_ticks(const LdsimConsts* c) {
  return (int64_t)ceil((c->light_window[1] - c->light_window[0]) / c->light_tick_size);
}

/* light_sim.calc_scintillation_effect (:148-184).  inc f4[D][T], truth i8/f8 [D][T][Mt] (Mt may be 0) */
int o_scintillation_effect(const float* inc, const int64_t* tid, const double* tph, int32_t D, int32_t T, int32_t Mt,
                           float* out, int64_t* out_tid, double* out_tph, const LdsimConsts* c) {
  const int64_t conv = o_conv_ticks(c);
  for (int64_t d = 0; d < D; d++)
    for (int64_t i = 0; i < T; i++) {
      float acc = out[d * T + i];
      int64_t j0 = i - conv > 0 ? i - conv : 0;
      for (int64_t j = j0; j <= i; j++) {
        float x = inc[d * T + j];
        if (x == 0) continue;
        double w = o_scintillation_model(i - j, c);
        acc = (float)((double)acc + w * (double)x);
        for (int a = 0; a < Mt; a++) {
          int64_t id = tid[(d * T + j) * Mt + a];
          if (id == -1) break;
          double ph = tph[(d * T + j) * Mt + a];
          if (w * ph < c->mc_truth_threshold) continue;
          for (int b = 0; b < Mt; b++) {
            int64_t* slot = &out_tid[(d * T + i) * Mt + b];
            if (*slot == id || *slot == -1) {
              *slot = id;
              out_tph[(d * T + i) * Mt + b] += w * ph;
              break;
            }
          }
        }
      }
      out[d * T + i] = acc;
    }
  return 0;
}

/* light_sim.calc_light_detector_response (:303-337).  light_gain is indexed by the ROW idet of the arrays, like the
 * reference (LIGHT_GAIN[idet], not by optical channel id).  The truth part follows the reference literally, including
 * that its slot test reads the INPUT ids at [idet, itick] (:331-333), not the output's and not at jtick. */
int o_light_detector_response(const float* inc, const int64_t* tid, const double* tph, int32_t D, int32_t T, int32_t Mt,
                              const double* light_gain, const double* impulse_model, int32_t n_impulse, float* out,
                              int64_t* out_tid, double* out_tph, const LdsimConsts* c) {
  const int64_t conv = o_conv_ticks(c);
  for (int64_t d = 0; d < D; d++)
    for (int64_t i = 0; i < T; i++) {
      float acc = out[d * T + i];
      int64_t j0 = i - conv > 0 ? i - conv : 0;
      for (int64_t j = j0; j <= i; j++) {
        double w = o_sipm_response_model(i - j, impulse_model, n_impulse, c);
        acc = (float)((double)acc + light_gain[d] * w * (double)inc[d * T + j]);
        for (int a = 0; a < Mt; a++) {
          if (tid[(d * T + j) * Mt + a] == -1) break;
          double ph = tph[(d * T + j) * Mt + a];
          if (fabs(w * ph) < c->mc_truth_threshold) continue;
          for (int b = 0; b < Mt; b++) {
            int64_t idb = tid[(d * T + i) * Mt + b], ida = tid[(d * T + i) * Mt + a];
            if (idb == ida || idb == -1) {
              out_tid[(d * T + i) * Mt + b] = ida;
              out_tph[(d * T + i) * Mt + b] += w * ph;
              break;
            }
          }
        }
      }
      out[d * T + i] = acc;
    }
  return 0;
}


/* ---- light: Poisson fluctuations and waveform digitisation (second half) ------------------------------ */
/* light_sim.xoroshiro128p_poisson_int32 (:186-216): inversion below a mean of 30 (one float32 uniform), else a truncated
 * normal (one float32 normal = two uniforms).  The generator is the restated third-party one above (unpinned). */
int32_t o_poisson_int32(double mean, ORng* st) {
  if (mean <= 0) return 0;
  if (mean < 30) {
    double u = (double)o_rng_uniform_f32(st);
    int32_t x = 0;
    double p = exp(-mean), s = p, prev_s = s;
    while (u > s) {
      x += 1;
      p = p * mean / x;
      prev_s = s;
      s = s + p;
      if (s == prev_s) break;
    }
    return x;
  }
  double v = (double)o_rng_normal_f32(st) * sqrt(mean) + mean;
  int64_t iv = (int64_t)v;                    /* int(): truncation toward zero */
  return iv > 0 ? (int32_t)iv : 0;
}

/* light_sim.calc_stat_fluctuations (:219-238): element e = idet*ntick + itick uses states[e]; the f4 input times the f64
 * tick size is f64 under Numba; the result is stored into the f4 output array. */
int o_stat_fluctuations(const float* inc, int64_t n, ORng* states, float* out, const LdsimConsts* c) {
  for (int64_t e = 0; e < n; e++) {
    if (inc[e] > 0)
      out[e] = (float)(1. / c->light_tick_size * (double)o_poisson_int32((double)inc[e] * c->light_tick_size, &states[e]));
    else
      out[e] = 0.f;
  }
  return 0;
}

/* light_sim.interp (:241-271) */
static double o_interp2(double idx, const double* arr, int64_t len, double low, double high, int f32diff) {
  int64_t i0 = (int64_t)floor(idx);
  if (i0 < 0) return low;
  if (i0 > len - 1) return high;
  if ((double)i0 == idx) return arr[i0];
  if (i0 > len - 2) return high;
  double d = f32diff ? (double)((float)arr[i0 + 1] - (float)arr[i0]) : arr[i0 + 1] - arr[i0];
  return arr[i0] + d * (idx - (double)i0);
}

/* light_sim.digitize_signal (:480-543).  signal [R][Tp] (f64 holding the values of the padded array; sig_is_f4 says the
 * reference's array was still f4 at that point, which only matters for Numba's f4 - f4 typing), signal_op [R],
 * trig_op [ntrig][ndm], truth [R][Tp][Mt]; outputs digit [ntrig][ndm][ns] (zero on entry), dtid (-1), dtph (0) [..][Mt].
 * Literal, including the row index `idet` (the optical channel id, not idet_signal) of the photons0 read (:520). */
int o_digitize_signal(const double* signal, int sig_is_f4, const int64_t* signal_op, int64_t R, int64_t Tp,
                      const int64_t* trig_op, int64_t ntrig, int64_t ndm, const int64_t* tid, const double* tph, int32_t Mt,
                      int64_t ns, double* digit, int64_t* dtid, double* dtph, const LdsimConsts* c) {
  for (int64_t itrig = 0; itrig < ntrig; itrig++)
    for (int64_t idm = 0; idm < ndm; idm++)
      for (int64_t is = 0; is < ns; is++) {
        double sample_tick = (double)is * c->light_digit_sample_spacing / c->light_tick_size;
        int64_t idet = trig_op[itrig * ndm + idm];
        int64_t s = 0;
        for (s = 0; s < R; s++)
          if (idet == signal_op[s]) break;
        /* Python's `for ... break` leaves idet_signal at R-1 when nothing matched (the `== signal.shape[0]` test of :497
         * can never fire), so an unmatched channel reads the LAST row */
        if (s == R) s = R - 1;
        if (R == 0) continue;
        const int64_t o = (itrig * ndm + idm) * ns + is;
        digit[o] = o_interp2(sample_tick, signal + s * Tp, Tp, 0, 0, sig_is_f4 && g_numba_f32);
        if (Mt == 0) continue;
        int64_t itick0 = (int64_t)floor(sample_tick), itick1 = (int64_t)ceil(sample_tick);
        if (itick0 < 0 || itick0 >= Tp) continue;       /* the reference would index out of bounds */
        int itrue = 0;
        for (int j = 0; j < Mt; j++) {
          if (itrue >= Mt) break;
          const int64_t id0 = tid[(s * Tp + itick0) * Mt + j];
          if (id0 == -1) break;
          double photons0 = 0, photons1 = 0;
          int64_t* slot = &dtid[o * Mt + itrue];
          if (id0 == *slot || *slot == -1) {
            *slot = id0;
            itrue += 1;
            photons0 = (idet >= 0 && idet < R) ? tph[(idet * Tp + itick0) * Mt + j] : 0.0;
            if (fabs(photons0) < c->mc_truth_threshold) continue;
            if (itick1 < Tp) {
              if (id0 == tid[(s * Tp + itick1) * Mt + j]) photons1 = tph[(s * Tp + itick1) * Mt + j];
              else
                for (int k = 0; k < Mt; k++)
                  if (id0 == tid[(s * Tp + itick1) * Mt + k]) { photons1 = tph[(s * Tp + itick1) * Mt + k]; break; }
            }
          }
          const int last = itrue - 1 < 0 ? Mt - 1 : itrue - 1;      /* Python's negative index wraps */
          if (dtid[o * Mt + last] != -1) {
            double pair[2] = {photons0, photons1};
            dtph[o * Mt + last] = o_interp2(sample_tick - (double)itick0, pair, 2, 0, 0, 0);
          }
        }
      }
  return 0;
}
