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

#define CUR_THREADS 256
#define TPL 8                    // ticks per lane
#define TILE_TICKS (CUR_THREADS * TPL)
#define ZC 64                    // max z slices per chunk
#define NU_MAX 64                // max distinct response shifts per chunk
#define NJ_MAX 48                // max distinct j cells (response table <= 48 wide in j)
#define NS_MAX 64                // max SAMPLED_POINTS
#define NEDGE 4                  // edge k's: k=0 and up to 3 at the top of the window


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

template <int M>
__global__ void __launch_bounds__(CUR_THREADS) current_kernel(CurArgs A) {
  const LdsimConsts* c = A.c;
  const int tid = threadIdx.x;
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
  __shared__ double s_row[M * TILE_TICKS + NU_MAX + 16];
  __shared__ double s_A[NJ_MAX * NU_MAX];
  __shared__ double s_corr[TILE_TICKS];
  __shared__ double s_C[NEDGE][NU_MAX];
  __shared__ double s_Redge[NEDGE][NJ_MAX];
  __shared__ double s_t0[ZC];
  __shared__ double s_z[ZC];
  __shared__ int s_shift[ZC];
  __shared__ int s_inval[ZC];       // bit e set: slice NOT valid at edge e (needs a correction)
  __shared__ int s_icell[NS_MAX], s_jcell[NS_MAX];
  __shared__ int s_cols[NS_MAX], s_colcnt[NS_MAX];
  __shared__ unsigned char s_colix[NS_MAX][NS_MAX];
  __shared__ int s_jflag[NJ_MAX];
  __shared__ int s_misc[8];

  // ---- sample -> response cell maps (detsim.py:434-446, :211-212) ------------------------------------
  if (tid < NS) {
    double x = g.x_start + g.sgnx * (tid * g.x_step - 4 * g.sT);
    double xd = fabs(g.x_p - x);
    int i = -1;
    if (!(xd > bin * A.ni)) {
      i = (int)py_round(xd / bin - 0.5);
      if (i < 0 || i >= A.ni) i = -1;
    }
    s_icell[tid] = i;
    double y = g.y_start + g.sgny * (tid * g.y_step - 4 * g.sT);
    double yd = fabs(g.y_p - y);
    int j = -1;
    if (!(yd > bin * A.nj)) {
      j = (int)py_round(yd / bin - 0.5);
      if (j < 0 || j >= A.nj) j = -1;
    }
    s_jcell[tid] = j;
  }
  __syncthreads();
  if (tid == 0) {
    int ncol = 0, jmin = 1 << 30, jmax = -1;
    for (int ix = 0; ix < NS; ix++) {
      int i = s_icell[ix];
      if (i < 0) continue;
      int cidx = -1;
      for (int q = 0; q < ncol; q++)
        if (s_cols[q] == i) {
          cidx = q;
          break;
        }
      if (cidx < 0) {
        cidx = ncol++;
        s_cols[cidx] = i;
        s_colcnt[cidx] = 0;
      }
      s_colix[cidx][s_colcnt[cidx]++] = ix;
    }
    for (int iy = 0; iy < NS; iy++) {
      int j = s_jcell[iy];
      if (j < 0) continue;
      jmin = min(jmin, j);
      jmax = max(jmax, j);
    }
    s_misc[0] = ncol;
    s_misc[1] = jmin;
    s_misc[2] = jmax;
  }
  __syncthreads();
  const int ncol = s_misc[0], jmin = s_misc[1], jmax = s_misc[2];
  const int NJ = jmax - jmin + 1;

  // ---- window-edge bookkeeping ----------------------------------------------------------------------
  // val = (time_tick - t0)/dtr lies in (0, V) when the slice is in its window; k = round(val).
  // k in [1, KC] is valid for every slice; k = 0 and k in (KC, KC+3] depend on the slice.
  const double V = TW / dtr;
  const int KC = (int)ceil(V - 0.5) - 2;
  int edge_k[NEDGE] = {0, KC + 1, KC + 2, KC + 3};
  const int k_stage_hi = min(min(KC + 3, A.nk - 1), A.k_last);
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
  const double iT2 = 1.0 / (g.sT * g.sT), iL2 = 1.0 / (g.sL * g.sL);
  const double a = ux * ux * i2T + uy * uy * i2T + uz * uz * i2L;
  const double factor = g.q / g.Dr / (g.sT * g.sT * g.sL * sqrt(8 * M_PI * M_PI * M_PI));
  const double sqrt_a_2 = 2 * sqrt(a);
  const double inv_sa2 = 1.0 / sqrt_a_2, inv4a = 1.0 / (4 * a);
  const double pref = factor * sqrt(M_PI) * inv_sa2 * g.dV;
  const double hi_off = 2 * a * g.Dr * inv_sa2;
  const bool do_prune = A.prune_log > 0;
  const double cut = -A.prune_log;

  unsigned long long n_blocks = 0;   // 8-shift MAC blocks executed (each = 64 DFMA per lane)
  for (int tile0 = 0; tile0 < T; tile0 += TILE_TICKS) {
    double acc[TPL];
#pragma unroll
    for (int j = 0; j < TPL; j++) acc[j] = 0;
    for (int i = tid; i < TILE_TICKS; i += CUR_THREADS) s_corr[i] = 0;

    int iz_next = iz_lo;
    while (iz_next <= iz_hi) {
      // ---- slice chunk: shifts, validity ---------------------------------------------------------------
      __syncthreads();
      if (tid < ZC && iz_next + tid <= iz_hi) {
        int iz = iz_next + tid;
        double z = g.z_start_int + iz * g.z_step;
        double t0 = fabs(z - g.z_anode) / c->v_drift - TW;
        s_z[tid] = z;
        s_t0[tid] = t0;
        // shift from the exact expression at a reference tick in the middle of the window
        int it_ref = (int)((t0 + 0.5 * TW - g.t_start) / dt);
        if (it_ref < 0) it_ref = 0;
        double tt = g.t_start + it_ref * dt;
        double val = (tt - t0) / dtr;
        double kr = py_round(val);
        if (fabs(val - kr) > 0.5 - 1e-7) atomicAdd(&A.counters[0], 1ull);
        int sh = (int)kr - M * it_ref;
        s_shift[tid] = sh;
        int inval = 0;
#pragma unroll
        for (int e = 0; e < NEDGE; e++) {
          int num = edge_k[e] - sh;
          bool ok = false;
          if (num >= 0 && (num % M) == 0) {
            int64_t kk;
            ok = slice_valid_at(c, g.t_start, t0, num / M, kk) && kk == edge_k[e];
          }
          if (!ok) inval |= 1 << e;
        }
        s_inval[tid] = inval;
      }
      __syncthreads();
      if (tid == 0) {
        int nmax = min(ZC, iz_hi - iz_next + 1);
        int lo = s_shift[0], hi = s_shift[0], n = 1;
        while (n < nmax) {
          int sh = s_shift[n];
          int nlo = min(lo, sh), nhi = max(hi, sh);
          if (nhi - nlo + 1 > NU_MAX) break;
          lo = nlo;
          hi = nhi;
          n++;
        }
        s_misc[3] = n;
        s_misc[4] = lo;
        s_misc[5] = hi;
      }
      __syncthreads();
      const int n_sl = s_misc[3], u_min = s_misc[4];
      const int NU = s_misc[5] - u_min + 1;
      const int NU8 = (NU + 7) & ~7;
      for (int i = tid; i < NEDGE * NU_MAX; i += CUR_THREADS) (&s_C[0][0])[i] = 0;

      // ---- columns of response cells sharing i ------------------------------------------------------------
      for (int col = 0; col < ncol; col++) {
        const int ci = s_cols[col];
        const int nix = s_colcnt[col];
        __syncthreads();
        for (int i = tid; i < NJ * NU_MAX; i += CUR_THREADS) s_A[i] = 0;
        if (tid < NJ_MAX) s_jflag[tid] = 0;
        for (int i = tid; i < NEDGE * NJ; i += CUR_THREADS) {
          int e = i / NJ, jj = i % NJ;
          int k = edge_k[e];
          double r = 0;
          if (k >= k_stage_lo && k <= k_stage_hi) r = A.resp[((int64_t)ci * A.nj + (jmin + jj)) * A.nk + k];
          s_Redge[e][jj] = r;
        }
        __syncthreads();
        // ---- (1) weights: rho*dV of every sample of this column, binned by (j, shift) ---------------------
        const int total = (A.debug_phases & 1) ? nix * NS * n_sl : 0;
        if (!(A.debug_phases & 1) && tid < NJ) { s_jflag[tid] = 1; s_A[tid * NU_MAX] = 1.0; }
        for (int idx = tid; idx < total; idx += CUR_THREADS) {
          int sl = idx % n_sl;
          int rest = idx / n_sl;
          int iy = rest % NS;
          int ixc = rest / NS;
          int j = s_jcell[iy];
          if (j < 0) continue;
          int ix = s_colix[col][ixc];
          double x = g.x_start + g.sgnx * (ix * g.x_step - 4 * g.sT);
          double y = g.y_start + g.sgny * (iy * g.y_step - 4 * g.sT);
          double z = s_z[sl];
          double ddx = x - g.sx, ddy = y - g.sy, ddz = z - g.sz;
          double b = -(ddx * iT2 * ux + ddy * iT2 * uy + ddz * iL2 * uz);
          double delta = ddx * ddx * i2T + ddy * ddy * i2T + ddz * ddz * i2L;
          double E = b * b * inv4a - delta;
          double lo = b * inv_sa2, hi = lo + hi_off;
          if (do_prune) {
            double E2 = E;
            if (lo > 0) E2 -= lo * lo;
            else if (hi < 0) E2 -= hi * hi;
            if (E2 < cut) continue;
          }
          double integral = erf(hi) - erf(lo);
          if (integral == 0) continue;
          double w = pref * integral * exp(E);
          int u = s_shift[sl] - u_min;
          int jj = j - jmin;
          atomicAdd(&s_A[jj * NU_MAX + u], w);
          s_jflag[jj] = 1;
          int inval = s_inval[sl];
          if (inval) {
#pragma unroll
            for (int e = 0; e < NEDGE; e++)
              if (inval & (1 << e)) {
                double r = s_Redge[e][jj];
                if (r != 0) atomicAdd(&s_C[e][u], w * r);
              }
          }
        }
        __syncthreads();
        // ---- (2) per cell: stage the response row, sliding-window correlation ---------------------------------
        for (int jj = 0; jj < NJ; jj++) {
          if (!s_jflag[jj] || !(A.debug_phases & 2)) continue;
          const double* rrow = A.resp + ((int64_t)ci * A.nj + (jmin + jj)) * A.nk;
          // row element r  <->  response index k = kb + r,  kb = M*tile0 + u_min
          const int kb = M * tile0 + u_min;
          const int nrow = M * TILE_TICKS + NU8 + 8;
          __syncthreads();
          for (int r = tid; r < nrow; r += CUR_THREADS) {
            int k = kb + r;
            s_row[r] = (k >= k_stage_lo && k <= k_stage_hi) ? rrow[k] : 0.0;
          }
          __syncthreads();
          n_blocks += NU8 / 8;
          const double* Aj = &s_A[jj * NU_MAX];
          const double* rw = &s_row[M * TPL * tid];
          double w[M * (TPL - 1) + 8 + 1];
#pragma unroll
          for (int q = 0; q < M * (TPL - 1) + 1; q++) w[q] = rw[q];
          for (int u0 = 0; u0 < NU8; u0 += 8) {
#pragma unroll
            for (int q = 0; q < 8; q++) w[M * (TPL - 1) + 1 + q] = rw[u0 + M * (TPL - 1) + 1 + q];
#pragma unroll
            for (int du = 0; du < 8; du++) {
              double av = Aj[u0 + du];
#pragma unroll
              for (int j = 0; j < TPL; j++) acc[j] = fma(av, w[M * j + du], acc[j]);
            }
#pragma unroll
            for (int q = 0; q < M * (TPL - 1) + 1; q++) w[q] = w[q + 8];
          }
        }
      }
      // ---- (3) window-edge corrections of this chunk -------------------------------------------------------------
      __syncthreads();
      for (int e = 0; e < NEDGE; e++) {
        if (tid < NU) {
          double cv = s_C[e][tid];
          if (cv != 0) {
            int num = edge_k[e] - (u_min + tid);
            if (num >= 0 && (num % M) == 0) {
              int it = num / M - tile0;
              if (it >= 0 && it < TILE_TICKS) s_corr[it] += cv;
            }
          }
        }
        __syncthreads();
      }
      iz_next += n_sl;
    }
    // ---- output: f32 store like the reference's `signals` (cli/simulate_pixels.py:1007-1009) ---------------------
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TPL; j++) s_row[TPL * tid + j] = acc[j] - s_corr[TPL * tid + j];
    __syncthreads();
    for (int i = tid; i < TILE_TICKS; i += CUR_THREADS) {
      int it = tile0 + i;
      if (it < A.T) out[it] = (it >= it0 && it < T) ? (float)s_row[i] : 0.f;
    }
    __syncthreads();
  }
  if (tid == 0 && n_blocks) atomicAdd(&A.counters[5], n_blocks * 64ull * CUR_THREADS);
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
