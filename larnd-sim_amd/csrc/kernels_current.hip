// kernels_current.hip -- a9-a12: induced current of one (segment, pixel) pair on all its time ticks.
// Reference: larndsim/detsim.py:351-453 (tracks_current), :42-112 (z_interval), :114-159 (rho),
// :161-218 (track_point, get_pixel_coordinates, get_closest_waveform), :455-466 (sign).
//
// The reference evaluates, for every tick, sum over (iz, ix, iy) of rho(x,y,z)*dV * R[i(ix)][j(iy)][k(it,iz)].
// rho does not depend on the tick, i only on ix, j only on iy and k = M*it + s(iz) (M = TIME_SAMPLING /
// RESPONSE_SAMPLING, s(iz) an integer shift per z slice).  So one workgroup per pair
//   (1) bins the charge samples once into weights  A[i][j][s]  (LDS, f64 ds_add),
//   (2) runs, per response cell (i,j), a register-tiled 1-D correlation  out[it] += sum_s A[s] * R[i][j][M*it+s]
//       with the R row staged in LDS and each lane holding 8 consecutive ticks (sliding window:
//       one LDS read per 8 DFMA),
//   (3) fixes the window-edge ticks (k = 0 and k ~ TIME_WINDOW/RESPONSE_SAMPLING), whose validity
//       depends on the individual slice, with exact per-slice predicates,
// and writes the f32 waveform.  All arithmetic is f64 (the reference's arithmetic type); only the
// store narrows to f32 like the reference's `signals` array.
#include "ldsim_args.h"
#include "erfcx_table.h"

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
#define ARENA 4352               // f64 weight entries held in LDS per column group
#define QLEN 128                 // survivor queue entries per wave
#define CELLS_MAX 512            // response cells per column group


struct PairGeo {
  double x_p, y_p;
  double sx, sy, sz;       // z-ordered start
  double Dx, Dy, Dz, Dr;   // segment and its length
  double dirx, diry, dirz;
  double sT, sL, q;
  double z_start_int, z_step, x_step, y_step, x_start, y_start, sgnx, sgny;
  double t_start, z_anode, dV;
  int z_steps;
  bool ok;
};

__device__ __forceinline__ double sgn(double x) { return x >= 0 ? 1.0 : -1.0; }

// detsim.py:42-112
__device__ __forceinline__ void z_interval(const double* sp, const double* ep, double x_p, double y_p, double tol,
                                           double& z_poca, double& z_lo, double& z_hi) {
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
  double m = (ye - ys) / (xe - xs);
  double q = (xe * ys - xs * ye) / (xe - xs);
  double a = m, b = -1, cc = q;
  double x_poca = (b * (b * x_p - a * y_p) - a * cc) / (a * a + b * b);
  double dx = end[0] - start[0], dy = end[1] - start[1], dz = end[2] - start[2];
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
    doca = fabs(a * x_p + b * y_p + cc) / sqrt(a * a + b * b);
  }
  double zp = start[2] + (x_poca - start[0]) / d0 * d2;
  if (tol > doca) {
    double length2D = sqrt((xe - xs) * (xe - xs) + (ye - ys) * (ye - ys));
    double dir2x = (end[0] - start[0]) / length2D;
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
__device__ void pair_geometry(const CurArgs& A, int64_t seg, int64_t pID, PairGeo& g) {
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
  g.Dx = end[0] - start[0]; g.Dy = end[1] - start[1]; g.Dz = end[2] - start[2];
  double length = sqrt(g.Dx * g.Dx + g.Dy * g.Dy + g.Dz * g.Dz);
  g.Dr = length;
  g.dirx = g.Dx / length; g.diry = g.Dy / length; g.dirz = g.Dz / length;
  g.sT = s.f[LDSIM_TRAN_DIFF][seg];
  g.sL = s.f[LDSIM_LONG_DIFF][seg];
  g.q = s.f[LDSIM_N_ELECTRONS][seg];
  double impact = fmax(sqrt((5 * g.sT) * (5 * g.sT) + (5 * g.sT) * (5 * g.sT)),
                       sqrt(c->pixel_pitch * c->pixel_pitch + c->pixel_pitch * c->pixel_pitch) / 2) * 2;
  double z_poca, z_s, z_e;
  z_interval(start, end, x_p, y_p, impact, z_poca, z_s, z_e);
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


// ---- rho without catastrophic cancellation and with one exp on the common path ----------------------------------
// The reference evaluates exp(b^2/4a - delta + log(factor) + log(integral)) with
// integral ~ erf(hi) - erf(lo) (detsim.py:150-157).  Same value, restated through erfcx(x) = exp(x^2) erfc(x)
// (piecewise degree-9 polynomials, 1 ulp, tools/gen_erfcx_table.py): for lo, hi of the same sign
//   exp(E) (erf(hi) - erf(lo)) = exp(E - A1^2) [erfcx(A1) - exp(-(A2^2 - A1^2)) erfcx(A2)],  A1 <= A2 the magnitudes,
// which is accurate where the literal difference cancels (checked against 40-digit arithmetic: <= 5e-14 relative).
__device__ __forceinline__ double tab_eval(const double (*T)[ERFCX_DEG + 1], int n, double x) {
  int i = (int)(x * 8.0);
  i = i < n - 1 ? i : n - 1;
  const double s = (x - (i * 0.125 + 0.0625)) * 16.0;
  const double* c = T[i];
  double p = c[ERFCX_DEG];
#pragma unroll
  for (int d = ERFCX_DEG - 1; d >= 0; d--) p = fma(p, s, c[d]);
  return p;
}
__device__ __forceinline__ double erfcx_pos(double x) {   // x >= 0
  if (x < 16.0) return tab_eval(erfcx_tab, ERFCX_N, x);
  const double t = 1.0 / (2 * x * x);
  double acc = 1.0, term = 1.0;
#pragma unroll
  for (int m = 1; m < 9; m++) {
    term = -term * (2 * m - 1) * t;
    acc += term;
  }
  return acc / (x * 1.7724538509055160273);
}
__device__ __forceinline__ double erf_pos(double x) { return x >= 6.0 ? 1.0 : tab_eval(erf_tab, ERF_N, x); }

// exp(E) * (erf(hi) - erf(lo)),  hi > lo
__device__ __forceinline__ double exp_erf_diff(double E, double lo, double hi) {
  if (lo < 0 && hi > 0) return exp(E) * (erf_pos(hi) + erf_pos(-lo));
  const double al = fabs(lo), ah = fabs(hi);
  const double A1 = fmin(al, ah), A2 = fmax(al, ah);
  if (A1 < 2.0) return exp(E) * (erf_pos(A2) - erf_pos(A1));
  const double e1 = exp(E - A1 * A1);
  const double t = (A2 - A1) * (A2 + A1);
  const double tail = t < 45.0 ? exp(-t) * erfcx_pos(A2) : 0.0;
  return e1 * (erfcx_pos(A1) - tail);
}

// LDS row layout: one pad double per 8*M elements so that the 8*M-element lane stride of the sliding window
// becomes 8*M+1 doubles -> conflict-free ds_read_b64 (MI355X_MICROARCH.md, LDS banking)
template <int M>
__device__ __forceinline__ int rpos(int r) { return r + (r >> (M == 1 ? 3 : 4)); }

template <int M>
__global__ void __launch_bounds__(CUR_THREADS) current_kernel(CurArgs A) {
  const LdsimConsts* c = A.c;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t pair = blockIdx.x;
  if (pair >= A.n_pairs) return;

  // ---- which (segment, pixel) -------------------------------------------------------------------
  int64_t seg, pID;
  if (A.pair_val) {
    int32_t v = A.pair_val[pair];
    seg = A.seg_begin + v / A.P;
    pID = (int64_t)((A.pair_key[pair] >> 4) & 0xFFFFFFFFull);
  } else {
    seg = A.seg_begin + pair / A.P;
    pID = A.pixels[pair];
  }
  float* out = A.out + pair * (int64_t)A.T;
  int T = A.T;
  if (A.tmax_batch) T = min(T, A.tmax_batch[A.s.batch[seg] - A.batch0]);

  PairGeo g;
  pair_geometry(A, seg, pID, g);
  if (!g.ok) {
    for (int it = tid; it < A.T; it += CUR_THREADS) out[it] = 0.f;
    return;
  }
  const int NS = c->sampled_points;
  const double dt = c->time_sampling, dtr = c->response_sampling, TW = c->time_window;
  const double bin = c->response_bin_size;

  // ---- LDS -----------------------------------------------------------------------------------------
  constexpr int ROWLEN = M * WTILE + NU_MAX + 16;
  constexpr int ROWBUF = ROWLEN + ROWLEN / (8 * M) + 8;
  __shared__ double s_row[NWAVE][ROWBUF];       // wave-private staged response rows
  __shared__ double s_A[ARENA];                 // weights A[cell][u] of the current column group
  __shared__ double s_Redge[NEDGE][CELLS_MAX];  // response at the partially valid edge k's, per cell
  __shared__ double s_C[NEDGE][NU_MAX];         // weight*response of slices that are NOT valid at an edge
  __shared__ double s_px[NS_MAX][2], s_py[NS_MAX][2], s_pz[ZC][2];   // separable parts of b and delta
  __shared__ int s_shift[ZC], s_inval[ZC];
  __shared__ short s_icell[NS_MAX], s_jcell[NS_MAX];
  __shared__ short s_colof[NS_MAX];             // column slot of every ix (-1: outside the table)
  __shared__ short s_coli[NS_MAX];              // response index i of every column slot
  __shared__ unsigned char s_ixord[NS_MAX];     // ix values ordered by column slot
  __shared__ short s_colstart[NS_MAX + 1];      // first position in s_ixord of every column
  __shared__ unsigned short s_list[CELLS_MAX];  // active cells of the group, deterministic order
  __shared__ unsigned char s_culo[CELLS_MAX], s_cuhi[CELLS_MAX];
  __shared__ unsigned int s_q[NWAVE][QLEN];     // per-wave queue of samples that passed the cheap bound
  __shared__ int s_misc[16];

  // ---- sample -> response cell maps (detsim.py:434-446, :211-212), columns = distinct i -------------------
  if (wv == 0) {
    int i = -1, j = -1;
    if (lane < NS) {
      double x = g.x_start + g.sgnx * (lane * g.x_step - 4 * g.sT);
      double xd = fabs(g.x_p - x);
      if (!(xd > bin * A.ni)) {
        i = (int)py_round(xd / bin - 0.5);
        if (i < 0 || i >= A.ni) i = -1;
      }
      double y = g.y_start + g.sgny * (lane * g.y_step - 4 * g.sT);
      double yd = fabs(g.y_p - y);
      if (!(yd > bin * A.nj)) {
        j = (int)py_round(yd / bin - 0.5);
        if (j < 0 || j >= A.nj) j = -1;
      }
      s_icell[lane] = (short)i;
      s_jcell[lane] = (short)j;
      // separable parts of rho's b and delta (detsim.py:114-118,146-148)
      double ddx = x - g.sx, ddy = y - g.sy;
      double iT2 = 1.0 / (g.sT * g.sT), i2T = 1.0 / (2 * g.sT * g.sT);
      s_px[lane][0] = ddx * iT2 * (g.Dx / g.Dr);
      s_px[lane][1] = ddx * ddx * i2T;
      s_py[lane][0] = ddy * iT2 * (g.Dy / g.Dr);
      s_py[lane][1] = ddy * ddy * i2T;
    }
    // leader of a column = first ix with that i; column slot = rank of the leader
    int leader = lane;
    for (int q = 0; q < NS; q++) {
      int iq = __shfl(i, q);
      if (q < leader && iq == i) leader = q;
    }
    bool is_leader = (lane < NS) && (i >= 0) && (leader == lane);
    unsigned long long lm = __ballot(is_leader);
    int slot = __popcll(lm & ((1ull << lane) - 1ull));
    int myslot = __shfl(slot, leader);
    if (i < 0 || lane >= NS) myslot = -1;
    int ncol = __popcll(lm);
    if (lane < NS) s_colof[lane] = (short)myslot;
    if (is_leader) s_coli[slot] = (short)i;
    // order ix by column slot: position = #ix with smaller slot + #earlier ix of the same slot
    int posn = 0;
    for (int q = 0; q < NS; q++) {
      int sq = __shfl(myslot, q);
      if (sq >= 0 && myslot >= 0 && (sq < myslot || (sq == myslot && q < lane))) posn++;
    }
    if (myslot >= 0) s_ixord[posn] = (unsigned char)lane;
    if (is_leader) s_colstart[slot] = (short)posn;
    int nvalid = __popcll(__ballot(myslot >= 0));
    int jmin = (j >= 0) ? j : (1 << 20), jmax = j;
    for (int off = 32; off > 0; off >>= 1) {
      jmin = min(jmin, __shfl_down(jmin, off));
      jmax = max(jmax, __shfl_down(jmax, off));
    }
    if (lane == 0) {
      s_colstart[ncol] = (short)nvalid;
      s_misc[0] = ncol;
      s_misc[1] = jmin;
      s_misc[2] = jmax;
    }
  }
  __syncthreads();
  const int ncol = s_misc[0], jmin = s_misc[1], jmax = s_misc[2];
  const int NJ = jmax - jmin + 1;
  if (ncol == 0 || NJ <= 0) {
    for (int it = tid; it < A.T; it += CUR_THREADS) out[it] = 0.f;
    return;
  }

  // ---- window-edge bookkeeping ----------------------------------------------------------------------
  // val = (time_tick - t0)/dtr lies in (0, V) while a slice is inside its window; k = round(val).
  // k = 0 is valid only for slices with val > 0; at the top, k is always valid if k + 0.5 < V, never if
  // k - 0.5 >= V, and depends on the slice in between (at most two such k).
  const double V = TW / dtr;
  int edge_k[NEDGE] = {0, -1, -1};
  int k_top;   // last response index any slice can use
  {
    int ka = (int)floor(V - 0.5 - 1e-6);          // k <= ka: always valid
    if ((double)ka + 0.5 >= V - 1e-6) ka--;
    int kn = (int)ceil(V + 0.5 + 1e-6);           // k >= kn: never valid
    k_top = kn - 1;
    int ne = 1;
    for (int k = ka + 1; k <= k_top && ne < NEDGE; k++) edge_k[ne++] = k;
    if (k_top - ka > NEDGE - 1) k_top = ka + NEDGE - 1;   // cannot happen for |fuzz| << 1
  }
  const int k_stage_hi = min(min(k_top, A.nk - 1), A.k_last);
  const int k_stage_lo = max(0, A.k_first);

  // first tick with time_tick >= 0 (detsim.py:418-420)
  int it0 = 0;
  if (g.t_start < 0) {
    int cand = (int)ceil(-g.t_start / dt) - 1;
    if (cand < 0) cand = 0;
    while (g.t_start + cand * dt < 0.) cand++;
    it0 = cand;
  }

  // active slices: rho <= q*N_T(0)*N_L(dist_z), so slices further than cz*sigma_L from the segment's z
  // range hold less than exp(-prune_log) of the peak density
  int iz_lo = 0, iz_hi = g.z_steps - 1;
  if (A.prune_log > 0 && g.z_step > 0) {
    double cz = sqrt(2.0 * A.prune_log) * g.sL;
    double zl = g.sz - cz, zh = g.sz + g.Dz + cz;
    double fl = floor((zl - g.z_start_int) / g.z_step) - 1, fh = ceil((zh - g.z_start_int) / g.z_step) + 1;
    if (fl > iz_lo) iz_lo = (int)fmin(fl, (double)g.z_steps);
    if (fh < iz_hi) iz_hi = (int)fmax(fh, -1.0);
  }

  // per-pair constants of rho (detsim.py:135-148)
  const double ux = g.Dx / g.Dr, uy = g.Dy / g.Dr, uz = g.Dz / g.Dr;
  const double i2T = 1.0 / (2 * g.sT * g.sT), i2L = 1.0 / (2 * g.sL * g.sL);
  const double iL2 = 1.0 / (g.sL * g.sL);
  const double a = ux * ux * i2T + uy * uy * i2T + uz * uz * i2L;
  const double factor = g.q / g.Dr / (g.sT * g.sT * g.sL * sqrt(8 * M_PI * M_PI * M_PI));
  const double sqrt_a_2 = 2 * sqrt(a);
  const double inv_sa2 = 1.0 / sqrt_a_2, inv4a = 1.0 / (4 * a);
  const double pref = factor * sqrt(M_PI) * inv_sa2 * g.dV;
  const double hi_off = 2 * a * g.Dr * inv_sa2;
  const bool do_prune = A.prune_log > 0;
  const double cut = -A.prune_log;

  // response shift of a slice from the exact expression at a reference tick in the middle of its window
  auto slice_shift = [&](int iz, double& z, double& t0, bool count) -> int {
    z = g.z_start_int + iz * g.z_step;
    t0 = fabs(z - g.z_anode) / c->v_drift - TW;
    int it_ref = (int)((t0 + 0.5 * TW - g.t_start) / dt);
    if (it_ref < 0) it_ref = 0;
    double tt = g.t_start + it_ref * dt;
    double val = (tt - t0) / dtr;
    double kr = py_round(val);
    if (count && fabs(val - kr) > 0.5 - 1e-7) atomicAdd(&A.counters[0], 1ull);
    return (int)kr - M * it_ref;
  };

  // ---- tick window that can see a non-zero response: fixed wave <-> tile assignment per pair --------------
  {
    int smin = 1 << 30, smax = -(1 << 30);
    for (int iz = iz_lo + tid; iz <= iz_hi; iz += CUR_THREADS) {
      double z, t0;
      int sh = slice_shift(iz, z, t0, false);
      smin = min(smin, sh);
      smax = max(smax, sh);
    }
    for (int off = 32; off > 0; off >>= 1) {
      smin = min(smin, __shfl_down(smin, off));
      smax = max(smax, __shfl_down(smax, off));
    }
    if (lane == 0) {
      s_misc[8 + wv] = smin;
      s_misc[12 + wv] = smax;
    }
  }
  __syncthreads();
  const int sh_min = min(min(s_misc[8], s_misc[9]), min(s_misc[10], s_misc[11]));
  const int sh_max = max(max(s_misc[12], s_misc[13]), max(s_misc[14], s_misc[15]));
  int it_w0 = it0, it_w1 = T;
  if (sh_min <= sh_max) {
    // k = M*it + s in [k_stage_lo, k_stage_hi]
    int lo = (k_stage_lo - sh_max) / M - 1, hi = (k_stage_hi - sh_min) / M + 2;
    it_w0 = max(it_w0, lo);
    it_w1 = min(it_w1, hi);
  }
  if (sh_min > sh_max || it_w1 <= it_w0 || k_stage_hi < k_stage_lo) {
    for (int it = tid; it < A.T; it += CUR_THREADS) out[it] = 0.f;
    return;
  }
  unsigned long long n_blocks = 0;   // 8-shift MAC blocks executed by this lane's wave (each = 64 DFMA per lane)

  for (int sup0 = it_w0; sup0 < it_w1; sup0 += TILE_TICKS) {
    const int wlen = min(it_w1 - sup0, TILE_TICKS);
    const int ntt = (wlen + WTILE - 1) / WTILE;            // 512-tick tiles in this super tile
    // wave -> (tile, share rank): 4 tiles: one each; 2 tiles: two waves per tile; 1 tile: all four waves
    int my_tile, share_rank, nshare;
    if (ntt >= 3) { my_tile = wv; share_rank = 0; nshare = 1; }
    else if (ntt == 2) { my_tile = wv >> 1; share_rank = wv & 1; nshare = 2; }
    else { my_tile = 0; share_rank = wv; nshare = 4; }
    const bool tile_live = my_tile < ntt;
    const int tb = sup0 + my_tile * WTILE;                   // first tick of this wave's tile
    double acc[TPL];
#pragma unroll
    for (int j = 0; j < TPL; j++) acc[j] = 0;

    int iz_next = iz_lo;
    while (iz_next <= iz_hi) {
      // ---- slice chunk: shifts, edge validity, separable z parts -------------------------------------------
      __syncthreads();
      if (wv == 0) {
        int nmax = min(ZC, iz_hi - iz_next + 1);
        int sh = 0, inval = 0;
        if (lane < nmax) {
          int iz = iz_next + lane;
          double z, t0;
          sh = slice_shift(iz, z, t0, sup0 == it_w0);
          double ddz = z - g.sz;
          s_pz[lane][0] = ddz * iL2 * uz;
          s_pz[lane][1] = ddz * ddz * i2L;
          s_shift[lane] = sh;
#pragma unroll
          for (int e = 0; e < NEDGE; e++) {
            bool ok = false;
            int num = edge_k[e] - sh;
            if (edge_k[e] >= 0 && num >= 0 && (num % M) == 0) {
              int64_t kk;
              ok = slice_valid_at(c, g.t_start, t0, num / M, kk) && kk == edge_k[e];
            }
            if (!ok) inval |= 1 << e;
          }
          s_inval[lane] = inval;
        }
        // longest prefix of slices whose shifts span <= NU_MAX values
        int pmin = lane < nmax ? sh : (1 << 30), pmax = lane < nmax ? sh : -(1 << 30);
        for (int off = 1; off < 64; off <<= 1) {
          int a1 = __shfl_up(pmin, off), a2 = __shfl_up(pmax, off);
          if (lane >= off) { pmin = min(pmin, a1); pmax = max(pmax, a2); }
        }
        bool fits = (lane < nmax) && (pmax - pmin + 1 <= NU_MAX);
        unsigned long long fm = __ballot(fits);
        int n = (fm == ~0ull) ? 64 : __ffsll((long long)~fm) - 1;
        int lo = __shfl(pmin, n - 1), hi = __shfl(pmax, n - 1);
        if (lane == 0) { s_misc[3] = n; s_misc[4] = lo; s_misc[5] = hi; }
      }
      __syncthreads();
      const int n_sl = s_misc[3], u_min = s_misc[4];
      const int NU = s_misc[5] - u_min + 1;
      const int NU8 = (NU + 7) & ~7;
      for (int i = tid; i < NEDGE * NU_MAX; i += CUR_THREADS) (&s_C[0][0])[i] = 0;
      const int cols_per_group = max(1, min(ARENA / (NJ * NU8), CELLS_MAX / NJ));

      for (int col0 = 0; col0 < ncol; col0 += cols_per_group) {
        const int gcols = min(cols_per_group, ncol - col0);
        const int ncell = gcols * NJ;
        const int g_ix0 = s_colstart[col0], g_nix = s_colstart[col0 + gcols] - g_ix0;
        __syncthreads();
        for (int i = tid; i < ncell * NU8; i += CUR_THREADS) s_A[i] = 0;
        for (int i = tid; i < NEDGE * ncell; i += CUR_THREADS) {
          int e = i / ncell, cell = i % ncell;
          int k = edge_k[e];
          double r = 0;
          if (k >= k_stage_lo && k <= k_stage_hi)
            r = A.resp[((int64_t)s_coli[col0 + cell / NJ] * A.nj + (jmin + cell % NJ)) * A.nk + k];
          s_Redge[e][cell] = r;
        }
        __syncthreads();
        // ---- (1) weights.  Pass A (all lanes busy): the cheap bound on every sample, survivors pushed to a
        //      wave-private queue.  Pass B (dense lanes): 2 erf + exp per survivor, f64 ds_add into A[cell][u]. ------
        if (A.debug_phases & 1) {
          unsigned int* q = s_q[wv];
          int qhead = 0, qn = 0;     // wave-uniform
          auto process = [&](int n) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (lane < n) {
              unsigned int en = q[(qhead + lane) & (QLEN - 1)];
              const int ix = en >> 12, iy = (en >> 6) & 63, sl = en & 63;
              double b = -(s_px[ix][0] + s_py[iy][0] + s_pz[sl][0]);
              double delta = s_px[ix][1] + s_py[iy][1] + s_pz[sl][1];
              double E = b * b * inv4a - delta;
              double lo = b * inv_sa2, hi = lo + hi_off;
              double w = (A.debug_phases & 4) ? pref * exp_erf_diff(E, lo, hi) : 1e-300 * E;
              if (w != 0 && (A.debug_phases & 8)) {
                const int cell = (s_colof[ix] - col0) * NJ + (s_jcell[iy] - jmin);
                const int u = s_shift[sl] - u_min;
                atomicAdd(&s_A[cell * NU8 + u], w);
                const int inval = s_inval[sl];
                if (inval) {
#pragma unroll
                  for (int e = 0; e < NEDGE; e++)
                    if (inval & (1 << e)) {
                      double r = s_Redge[e][cell];
                      if (r != 0) atomicAdd(&s_C[e][u], w * r);
                    }
                }
              }
            }
            qhead = (qhead + n) & (QLEN - 1);
            qn -= n;
          };
          const int npz = NS * n_sl;
          const int npz_pad = (npz + 63) & ~63;
          for (int p0 = wv * 64; p0 < npz_pad; p0 += CUR_THREADS) {
            const int p = p0 + lane;
            bool pv = p < npz;
            const int iy = pv ? p / n_sl : 0, sl = pv ? p - iy * n_sl : 0;
            pv = pv && (s_jcell[iy] >= 0);
            const double byz = s_py[iy][0] + s_pz[sl][0], dyz = s_py[iy][1] + s_pz[sl][1];
            const unsigned int tag = ((unsigned)iy << 6) | (unsigned)sl;
            for (int gi = 0; gi < g_nix; gi++) {
              const int ix = s_ixord[g_ix0 + gi];
              bool keep = pv;
              if (do_prune && keep) {
                double b = -(s_px[ix][0] + byz);
                double E2 = b * b * inv4a - (s_px[ix][1] + dyz);
                double lo = b * inv_sa2, hi = lo + hi_off;
                if (lo > 0) E2 -= lo * lo;
                else if (hi < 0) E2 -= hi * hi;
                keep = !(E2 < cut);
              }
              unsigned long long m = __ballot(keep);
              if (m) {
                if (keep) q[(qhead + qn + __popcll(m & ((1ull << lane) - 1ull))) & (QLEN - 1)] = ((unsigned)ix << 12) | tag;
                qn += __popcll(m);
                if (qn >= 64) process(64);
              }
            }
          }
          if (qn > 0) process(qn);
        } else if (tid < ncell) {
          s_A[tid * NU8] = 1.0;
        }
        __syncthreads();
        // ---- active cells + their shift range, in deterministic (cell index) order -----------------------------
        for (int cell = wv; cell < ncell; cell += NWAVE) {
          double v = (lane < NU8) ? s_A[cell * NU8 + lane] : 0.0;
          unsigned long long nz = __ballot(v != 0.0);
          if (lane == 0) {
            s_culo[cell] = nz ? (unsigned char)(__ffsll((long long)nz) - 1) : (unsigned char)255;
            s_cuhi[cell] = nz ? (unsigned char)(63 - __clzll((long long)nz)) : (unsigned char)0;
          }
        }
        __syncthreads();
        if (wv == 0) {
          int nact = 0;
          for (int base = 0; base < ncell; base += 64) {
            int cell = base + lane;
            bool act = (cell < ncell) && (s_culo[cell] != 255);
            unsigned long long am = __ballot(act);
            if (act) s_list[nact + __popcll(am & ((1ull << lane) - 1ull))] = (unsigned short)cell;
            nact += __popcll(am);
          }
          if (lane == 0) s_misc[6] = nact;
        }
        __syncthreads();
        // ---- (2) per cell: stage the response row (wave-private), sliding-window correlation ---------------------
        const int nact = s_misc[6];
        if (tile_live && (A.debug_phases & 2)) {
          double* rowp = s_row[wv];
          for (int li = share_rank; li < nact; li += nshare) {
            const int cell = s_list[li];
            const int ulo8 = s_culo[cell] & ~7;
            const int nblk = (s_cuhi[cell] - ulo8) / 8 + 1;
            const double* rrow = A.resp + ((int64_t)s_coli[col0 + cell / NJ] * A.nj + (jmin + cell % NJ)) * A.nk;
            // row element r  <->  response index k = kb + r
            const int kb = M * tb + u_min + ulo8;
            const int nrow = M * WTILE + nblk * 8 + 8;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            for (int r = lane; r < nrow; r += 64) {
              int k = kb + r;
              rowp[rpos<M>(r)] = (k >= k_stage_lo && k <= k_stage_hi) ? rrow[k] : 0.0;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const double* Aj = &s_A[cell * NU8 + ulo8];
            const int rl = M * TPL * lane;
            double w[M * (TPL - 1) + 8 + 1];
#pragma unroll
            for (int q = 0; q < M * (TPL - 1) + 1; q++) w[q] = rowp[rpos<M>(rl + q)];
            for (int b8 = 0; b8 < nblk; b8++) {
              const int u0 = b8 * 8;
#pragma unroll
              for (int q = 0; q < 8; q++) w[M * (TPL - 1) + 1 + q] = rowp[rpos<M>(rl + u0 + M * (TPL - 1) + 1 + q)];
#pragma unroll
              for (int du = 0; du < 8; du++) {
                double av = Aj[u0 + du];
#pragma unroll
                for (int j = 0; j < TPL; j++) acc[j] = fma(av, w[M * j + du], acc[j]);
              }
#pragma unroll
              for (int q = 0; q < M * (TPL - 1) + 1; q++) w[q] = w[q + 8];
            }
            n_blocks += nblk;
          }
        }
      }
      // ---- (3) window-edge corrections of this chunk: tick it <-> shift u = k_e - M*it - u_min ------------------------
      __syncthreads();
      if (tile_live && share_rank == 0) {
#pragma unroll
        for (int e = 0; e < NEDGE; e++) {
          if (edge_k[e] < 0) continue;
#pragma unroll
          for (int j = 0; j < TPL; j++) {
            int it = tb + TPL * lane + j;
            int u = edge_k[e] - M * it - u_min;
            if (u >= 0 && u < NU) acc[j] -= s_C[e][u];
          }
        }
      }
      iz_next += n_sl;
    }
    // ---- output: combine the waves sharing a tile, f32 store like the reference's `signals` ---------------------------
    __syncthreads();
    double* s_out = s_A;   // TILE_TICKS doubles
    for (int rnk = 0; rnk < nshare; rnk++) {
      if (tile_live && share_rank == rnk) {
#pragma unroll
        for (int j = 0; j < TPL; j++) {
          int idx = my_tile * WTILE + TPL * lane + j;
          s_out[idx] = (rnk == 0) ? acc[j] : s_out[idx] + acc[j];
        }
      }
      __syncthreads();
    }
    for (int i = tid; i < wlen; i += CUR_THREADS) {
      int it = sup0 + i;
      if (it < A.T) out[it] = (it >= it0 && it < T) ? (float)s_out[i] : 0.f;
    }
    __syncthreads();
  }
  // ticks outside the response-visible window are exactly zero
  for (int it = tid; it < A.T; it += CUR_THREADS)
    if (it < it_w0 || it >= it_w1) out[it] = 0.f;
  if (lane == 0 && n_blocks) atomicAdd(&A.counters[5], n_blocks * 64ull * 64ull);
}

extern "C++" int current_launch(ldsim_ctx* ctx, const CurArgs& args) {
  if (args.n_pairs == 0) return 0;
  const LdsimConsts& h = ctx->h_consts;
  double ratio = h.time_sampling / h.response_sampling;
  int M = (int)llround(ratio);
  if (M < 1 || M > 2 || fabs(ratio - M) > 1e-9) {
    ldsim_set_error("TIME_SAMPLING/RESPONSE_SAMPLING = %g unsupported (must be 1 or 2)", ratio);
    return LDSIM_EINVAL;
  }
  if (h.sampled_points > NS_MAX || args.nj > NJ_MAX || args.ni > NS_MAX * 2) {
    ldsim_set_error("response table / SAMPLED_POINTS too large for the kernel's static tiles");
    return LDSIM_EINVAL;
  }
  if (args.n_pairs > 0x7fffffffLL) {
    ldsim_set_error("too many pairs for one launch");
    return LDSIM_EINVAL;
  }
  dim3 grid((unsigned)args.n_pairs), block(CUR_THREADS);
  if (M == 1)
    hipLaunchKernelGGL(current_kernel<1>, grid, block, 0, ctx->stream, args);
  else
    hipLaunchKernelGGL(current_kernel<2>, grid, block, 0, ctx->stream, args);
  HIPCHK(hipGetLastError());
  return 0;
}
