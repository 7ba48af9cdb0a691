// kernels_qsetup.hip -- per-pair set-up pass of the quadrature weights stage (kernels_qweights.hip).
//
// pair_setup_kernel: one THREAD per (segment, pixel) pair computes what is uniform inside the pair -- the geometry of
// detsim.py:366-414, the valid sample range on either axis, the slice range, the response-shift range and tick window, the
// part of the segment that can reach the sample box and the Gauss-Legendre node count -- and writes a 240-byte record.
// qweights_kernel (a 256-thread workgroup per pair) used to repeat this in each of its four waves; here 64 different pairs
// share a wave.  (A wave-per-pair form of the whole weights stage was tried on top of this record and measured slower:
// DESIGN.md section 4, profiles/r02_qwave_phase_timing.log.)
#include "gform.h"

// =============================================================================================================
template <int M>
__global__ void __launch_bounds__(256) pair_setup_kernel(SplitArgs S, PairParams* __restrict__ pp, int qn_max, double qn0,
                                                         double qslope, GInfo* __restrict__ gi, unsigned char* __restrict__ maps) {
  // per thread: response column / row of every sample, and the counters of the counting sorts ([k][thread]: a wave's 64 bytes of
  // one k are contiguous)
  __shared__ unsigned char s_si[G_MAP_NS][256], s_sj[G_MAP_NS][256], s_cnt[64][256];
  const int tx = threadIdx.x;
  const CurArgs& A = S.c;
  const LdsimConsts* c = A.c;
  const int64_t pair = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (pair >= A.n_pairs) return;
  PairParams P;
  memset(&P, 0, sizeof(P));
  GInfo Gi;
  memset(&Gi, 0, sizeof(Gi));
  if (gi) gi[pair] = Gi;            // overwritten below when the pair has something to compute
  int64_t seg, pID;
  pair_ids(A, pair, seg, pID);
  int T = A.T;
  if (A.tmax_batch) T = min(T, A.tmax_batch[A.s.batch[seg] - A.batch0]);
  PairGeo g;
  pair_geometry(A, seg, pID, g);
  if (!g.ok) { pp[pair] = P; return; }
  const int NS = c->sampled_points;
  const double dt = c->time_sampling, bin = c->response_bin_size;
  // valid samples of either axis (inside the response table) and their extent relative to the segment start
  double xlo = 1e300, xhi = -1e300, ylo = 1e300, yhi = -1e300;
  unsigned long long i_present = 0;     // response columns i that hold a sample (ni <= 64), and the j range
  int j_lo = 1 << 20, j_hi = -1;
  for (int s = 0; s < NS; s++) {
    const double x = g.x_start + g.sgnx * (s * g.x_step - 4 * g.sT);
    const double xd = fabs(g.x_p - x);
    int i_s = 0xFF, j_s = 0xFF;
    if (!(xd > bin * A.ni)) {
      const int i = (int)py_round(xd / bin - 0.5);
      if (i >= 0 && i < A.ni) { xlo = fmin(xlo, x - g.sx); xhi = fmax(xhi, x - g.sx); i_present |= 1ull << i; i_s = i; }
    }
    const double y = g.y_start + g.sgny * (s * g.y_step - 4 * g.sT);
    const double yd = fabs(g.y_p - y);
    if (!(yd > bin * A.nj)) {
      const int j = (int)py_round(yd / bin - 0.5);
      if (j >= 0 && j < A.nj) { ylo = fmin(ylo, y - g.sy); yhi = fmax(yhi, y - g.sy); j_lo = min(j_lo, j); j_hi = max(j_hi, j); j_s = j; }
    }
    if (maps && s < G_MAP_NS) { s_si[s][tx] = (unsigned char)i_s; s_sj[s][tx] = (unsigned char)j_s; }
  }
  if (xhi < xlo || yhi < ylo) { pp[pair] = P; return; }
  int edge_k[NEDGE], k_stage_lo, k_stage_hi;
  edge_ks(c, A, edge_k, k_stage_lo, k_stage_hi);
  int it0 = 0;
  if (g.t_start < 0) {
    int cand = (int)ceil(-g.t_start / dt) - 1;
    if (cand < 0) cand = 0;
    while (g.t_start + cand * dt < 0.) cand++;
    it0 = cand;
  }
  const double ux = g.Dx / g.Dr, uy = g.Dy / g.Dr, uz = g.Dz / g.Dr;
  const double i2T = 1.0 / (2 * g.sT2), i2L = 1.0 / (2 * g.sL2);
  const double a = ux * ux * i2T + uy * uy * i2T + uz * uz * i2L;
  // Pruning is relative to the largest weight the segment can put on a sample: factor dV sqrt(pi / a) on the axis of a
  // segment much longer than the Gaussian's width along it, factor dV Dr for one much shorter (its charge is q, not the
  // q / Dr per unit length of the long one).  Thresholds and clips below use exp(-prune_eff) of the on-axis density, with
  // prune_eff = prune_log + log(sqrt(pi / a) / Dr) for the short ones, so that "1e-10 of the peak weight" holds for
  // micrometre-long segments as well (tests: test_tracks_current_length_sweep_vs_oracle, r = 0.01).
  const double peak_len = fmin(sqrt(M_PI / a), g.Dr);
  const double prune_eff = A.prune_log > 0 ? A.prune_log + log(sqrt(M_PI / a) / peak_len) : 0.0;
  int iz_lo = 0, iz_hi = g.z_steps - 1;
  if (A.prune_log > 0 && g.z_step > 0) {
    double cz = sqrt(2.0 * prune_eff) * g.sL;
    double zl = g.sz - cz, zh = g.sz + g.Dz + cz;
    double fl = floor((zl - g.z_start_int) / g.z_step) - 1, fh = ceil((zh - g.z_start_int) / g.z_step) + 1;
    if (fl > iz_lo) iz_lo = (int)fmin(fl, (double)g.z_steps);
    if (fh < iz_hi) iz_hi = (int)fmax(fh, -1.0);
  }
  const double factor = g.q / g.Dr / (g.s3 * sqrt(8 * M_PI * M_PI * M_PI));
  // the part of the segment that can reach the sample box (kernels_qweights.hip)
  const double G = sqrt(2.0 * ((A.prune_log > 0 ? prune_eff : 43.0) + 7.0));
  double s_lo = 0, s_hi = g.Dr;
  {
    const double z0 = g.z_start_int + iz_lo * g.z_step - g.sz, z1 = g.z_start_int + iz_hi * g.z_step - g.sz;
    const double lo3[3] = {xlo, ylo, fmin(z0, z1)}, hi3[3] = {xhi, yhi, fmax(z0, z1)};
    const double u3[3] = {ux, uy, uz}, w3[3] = {sqrt(g.sT2), sqrt(g.sT2), sqrt(g.sL2)};
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const double lo = lo3[k] - G * w3[k], hi = hi3[k] + G * w3[k];
      if (u3[k] != 0.0) {
        const double sa = lo / u3[k], sb = hi / u3[k];
        s_lo = fmax(s_lo, fmin(sa, sb));
        s_hi = fmin(s_hi, fmax(sa, sb));
      } else if (lo > 0 || hi < 0) {
        s_hi = -1;
      }
    }
  }
  if (!(s_hi > s_lo) || iz_hi < iz_lo) { pp[pair] = P; return; }
  const double qlen = s_hi - s_lo;
  // nodes for a relative quadrature error of 1e-v of the peak weight: tools/quad_nodes.py (v = 7, the default: 3.4 + 1.38 r;
  // 1e-10: 4.8 + 1.6 r; 1e-12 needs 6 + 1.9 r), r = clipped length in Gaussian widths along the segment
  const double nq_f = ceil(qn0 + qslope * qlen * sqrt(2.0 * a));
  // response shifts of the slices -> the tick window in which any of them sees a staged response entry
  int sh_min = 1 << 30, sh_max = -(1 << 30);
  for (int iz = iz_lo; iz <= iz_hi; iz++) {
    double z, t0;
    bool amb;
    const int sh = slice_shift_of<M>(c, g.z_start_int, g.z_step, g.z_anode, g.t_start, iz, z, t0, amb);
    sh_min = min(sh_min, sh);
    sh_max = max(sh_max, sh);
  }
  int it_w0 = it0, it_w1 = T;
  {
    int lo = (k_stage_lo - sh_max) / M - 1, hi = (k_stage_hi - sh_min) / M + 2;
    it_w0 = max(it_w0, lo);
    it_w1 = min(it_w1, hi);
  }
  if (it_w1 <= it_w0 || k_stage_hi < k_stage_lo) { pp[pair] = P; return; }
  P.status = (nq_f <= (double)qn_max) ? 1 : 2;
  P.NQ = P.status == 1 ? (int)nq_f : 0;
  P.iz_lo = iz_lo; P.iz_hi = iz_hi; P.it0 = it0; P.T = T; P.it_w0 = it_w0; P.it_w1 = it_w1;
  P.x_p = g.x_p; P.y_p = g.y_p; P.x_start = g.x_start; P.y_start = g.y_start; P.x_step = g.x_step; P.y_step = g.y_step;
  P.sgnx = g.sgnx; P.sgny = g.sgny; P.sT = g.sT; P.sx = g.sx; P.sy = g.sy;
  P.z_start_int = g.z_start_int; P.z_step = g.z_step; P.z_anode = g.z_anode; P.t_start = g.t_start; P.sz = g.sz;
  // numba_f32 (kernels_qweights.hip): centres scaled by r = sigma^2 / (sigma*sigma)_f32, node factor exp(-s^2 kappa)
  P.uxr = ux * g.rT; P.uyr = uy * g.rT; P.uzr = uz * g.rL; P.i2T = i2T; P.i2L = i2L;
  P.kappa = (ux * ux + uy * uy) * (1.0 - g.rT * g.rT) * i2T + uz * uz * (1.0 - g.rL * g.rL) * i2L;
  P.s_lo = s_lo; P.qlen = qlen;
  P.wscale = factor * g.dV * 0.5 * qlen;
  P.thr = A.prune_log > 0 ? exp(-A.prune_log) * factor * g.dV * peak_len : 0.0;
  pp[pair] = P;
  if (!gi) return;
  // ---- node-separable form (gform.h): what sizes the pair's record, and whether the pair fits its kernels ---------------
  Gi.ncol = __popcll(i_present);
  Gi.NJ = j_hi - j_lo + 1;
  Gi.jmin = j_lo;
  Gi.u_min = sh_min;
  Gi.NU = sh_max - sh_min + 1;
  Gi.NB = (P.NQ + G_NODES - 1) / G_NODES;
  Gi.status = P.status;
  Gi.NQ = P.NQ; Gi.it0 = it0; Gi.T = T; Gi.it_w0 = it_w0; Gi.it_w1 = it_w1;
  Gi.wave_ok = 0;
  if (iz_hi - iz_lo + 1 <= ZC) {
    if (Gi.NU <= G_NUCAP && Gi.ncol + Gi.NJ <= 54) Gi.wave_ok = 1;
    else if (Gi.NU <= 2 * G_NUCAP && Gi.ncol + Gi.NJ <= 80) Gi.wave_ok = 2;
  }
  if (!maps || NS > G_MAP_NS) Gi.wave_ok = 0;      // (the wave kernel works from the maps below)
  if (P.status == 1 && (Gi.ncol > G_NCOL || Gi.NJ > NJ_MAX || Gi.ncol * Gi.NJ > G_CELLCAP - G_CELLPAD)) {
    Gi.status = 2;
    pp[pair].status = 2;
  }
  if (Gi.status == 1) {
    // the window edges that need a table of their own: a slice's weight would be used at a tick where the reference does not
    // use it (the predicates of qweights_kernel's chunk set-up)
    const bool wmap = Gi.wave_ok != 0;
    unsigned long long* mrec = (unsigned long long*)(maps + (wmap ? pair : 0) * G_MAPB);
    int eb = 0, n_amb = 0;
    unsigned long long w_sh = 0, w_inv = 0;
    for (int iz = iz_lo; iz <= iz_hi; iz++) {
      double z, t0;
      bool amb;
      const int sh = slice_shift_of<M>(c, g.z_start_int, g.z_step, g.z_anode, g.t_start, iz, z, t0, amb);
      n_amb += amb;
      int inval = 0;
      for (int e = 0; e < NEDGE; e++) {
        const int num = edge_k[e] - sh;
        if (edge_k[e] >= k_stage_lo && edge_k[e] <= k_stage_hi && num >= 0 && (num % M) == 0) {
          const int it_e = num / M;
          if (it_e >= max(it0, it_w0) && it_e < min(T, it_w1)) {
            int64_t kk;
            if (!(slice_valid_at(c, g.t_start, t0, it_e, kk) && kk == edge_k[e])) inval |= 1 << e;
          }
        }
      }
      eb |= inval;
      if (wmap) {          // (<= ZC = 64 slices, <= 256 shifts)
        const int k = iz - iz_lo;
        w_sh |= (unsigned long long)(unsigned)(sh - sh_min) << (8 * (k & 7));
        w_inv |= (unsigned long long)(unsigned)inval << (8 * (k & 7));
        if ((k & 7) == 7 || iz == iz_hi) {
          mrec[G_MAP_ZSH / 8 + (k >> 3)] = w_sh;
          mrec[G_MAP_ZINV / 8 + (k >> 3)] = w_inv;
          w_sh = w_inv = 0;
        }
      }
    }
    if (wmap) {
      mrec[G_MAP_AMB / 8] = (unsigned long long)n_amb;
      // ---- x samples ordered by (column, s): counting sort over the columns that are present ------------------------------------------
      for (unsigned long long m = i_present; m; m &= m - 1) s_cnt[__ffsll((long long)m) - 1][tx] = 0;
      for (int s = 0; s < NS; s++) {
        const int i = s_si[s][tx];
        if (i != 0xFF) s_cnt[i][tx]++;
      }
      {
        unsigned long long wc = 0, ws = 0;
        int slot = 0, running = 0;
        for (unsigned long long m = i_present; m; m &= m - 1) {
          const int col = __ffsll((long long)m) - 1, cn = s_cnt[col][tx];
          s_cnt[col][tx] = (unsigned char)running;
          wc |= (unsigned long long)(unsigned)col << (8 * (slot & 7));
          ws |= (unsigned long long)(unsigned)running << (8 * (slot & 7));
          running += cn;
          slot++;
          if ((slot & 7) == 0) {
            mrec[G_MAP_COLI / 8 + (slot >> 3) - 1] = wc;
            mrec[G_MAP_COLSTART / 8 + (slot >> 3) - 1] = ws;
            wc = ws = 0;
          }
        }
        ws |= (unsigned long long)(unsigned)running << (8 * (slot & 7));      // the count, behind the last column's start
        if (slot & 7) mrec[G_MAP_COLI / 8 + (slot >> 3)] = wc;
        mrec[G_MAP_COLSTART / 8 + (slot >> 3)] = ws;
      }
      for (int s8 = 0; s8 < G_MAP_NS; s8 += 8) {
        unsigned long long w = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const int s = s8 + k;
          int pos = 0xFF;
          if (s < NS) {
            const int i = s_si[s][tx];
            if (i != 0xFF) { pos = s_cnt[i][tx]; s_cnt[i][tx] = (unsigned char)(pos + 1); }
          }
          w |= (unsigned long long)(unsigned)pos << (8 * k);
        }
        mrec[G_MAP_XPOS / 8 + (s8 >> 3)] = w;
      }
      // ---- y samples ordered by (row, s); JSTART[k] = samples in rows below jmin + k -----------------------------------------------------------
      const int NJ = Gi.NJ;
      for (int k = 0; k < NJ; k++) s_cnt[k][tx] = 0;
      for (int s = 0; s < NS; s++) {
        const int j = s_sj[s][tx];
        if (j != 0xFF) s_cnt[j - j_lo][tx]++;
      }
      {
        unsigned long long ws = 0;
        int running = 0;
        for (int k = 0; k <= NJ; k++) {           // (NJ <= NJ_MAX = 48: entries 0 .. 48)
          ws |= (unsigned long long)(unsigned)running << (8 * (k & 7));
          if (k < NJ) {
            const int cn = s_cnt[k][tx];
            s_cnt[k][tx] = (unsigned char)running;
            running += cn;
          }
          if ((k & 7) == 7 || k == NJ) { mrec[G_MAP_JSTART / 8 + (k >> 3)] = ws; ws = 0; }
        }
      }
      for (int s8 = 0; s8 < G_MAP_NS; s8 += 8) {
        unsigned long long w = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const int s = s8 + k;
          int pos = 0xFF;
          if (s < NS) {
            const int j = s_sj[s][tx];
            if (j != 0xFF) { pos = s_cnt[j - j_lo][tx]; s_cnt[j - j_lo][tx] = (unsigned char)(pos + 1); }
          }
          w |= (unsigned long long)(unsigned)pos << (8 * k);
        }
        mrec[G_MAP_YPOS / 8 + (s8 >> 3)] = w;
      }
    }
    Gi.edge_bound = eb;
    Gi.size = g_record_doubles(Gi.NB, Gi.NQ, Gi.ncol, Gi.NJ, Gi.NU, eb);
  }
  gi[pair] = Gi;
}

extern "C++" size_t qpair_params_bytes(int64_t n_pairs) { return (size_t)n_pairs * sizeof(PairParams); }

// the per-pair records of both quadrature weight kernels
extern "C++" int qpair_setup_launch(ldsim_ctx* ctx, const SplitArgs& S, int M, void* params, void* ginfo, void* maps) {
  if (S.c.n_pairs == 0) return 0;
  if (!ctx->d_glx || !ctx->d_glw || !params) {
    ldsim_set_error("Gauss-Legendre tables / pair parameter buffer missing");
    return LDSIM_ESTATE;
  }
  PairParams* pp = (PairParams*)params;
  const unsigned g0 = (unsigned)((S.c.n_pairs + 255) / 256);
  GInfo* gi = (GInfo*)ginfo;
  if (M == 1) hipLaunchKernelGGL(pair_setup_kernel<1>, dim3(g0), dim3(256), 0, ctx->stream, S, pp, ctx->gl_nmax, ctx->quad_n0,
                                 ctx->quad_slope, gi, (unsigned char*)maps);
  else hipLaunchKernelGGL(pair_setup_kernel<2>, dim3(g0), dim3(256), 0, ctx->stream, S, pp, ctx->gl_nmax, ctx->quad_n0, ctx->quad_slope,
                          gi, (unsigned char*)maps);
  HIPCHK(hipGetLastError());
  return 0;
}

