// kernels_qwave.hip -- a9-a12, weights stage of the split path: a per-pair set-up pass and ONE WAVE PER (segment, pixel) PAIR
// ("weights_mode" 2, default).
//
// Same mathematics as kernels_qweights.hip (Gauss-Legendre quadrature along the segment turns the binned weights into a sum of
// separable terms  A[i][j][shift] = sum_n w_n X_n[i] Y_n[j] Z_n[shift]), different mapping.  With the quadrature a pair is
// ~2e3 exponentials and ~6e4 FMAs: too little for a 256-thread workgroup.  PMC counters of qweights_kernel (profiles/README.md):
// 23 k vector instructions per pair of which a quarter is the pair geometry repeated by each of its four waves, phases fenced
// by workgroup barriers, 65 % of the wave cycles parked.  Here
//   pair_setup_kernel   one THREAD per pair: geometry (detsim.py:366-414), valid sample range, slice range, response-shift
//                       range and tick window, the part of the segment that can reach the sample box and the node count ->
//                       a 240-byte record.  Work that is uniform inside a pair now runs with 64 different pairs per wave.
//   qwave_kernel        one WAVE per pair, no workgroup barrier anywhere.  The record arrives through scalar loads (the pair's
//                       constants live in SGPRs).  x / y sample maps (lane = sample) -> per 64-slice chunk: shifts and edge
//                       flags (lane = slice) -> tables X[n][column], Y[n][j] (one lane per (node, bin)) -> window-edge bilinear
//                       form Q[n] (lane = node) -> Z[n][shift] tasks, which also fold the flagged slices into the edge
//                       corrections -> per group of cells every lane owns up to 3 (cell, 8-shift) tiles in registers and adds
//                       the nodes -> active range of every cell, pool allocation, items + weight blocks to HBM.
// The tables of ALL nodes stay resident in the wave's LDS arena when they fit (typical: 18 nodes x 52 bins); row strides are odd
// so that the lane = node and the lane = bin accesses are both conflict-free.  exp() of the <= 0 Gaussian exponents is a
// 13-term polynomial after Cody-Waite reduction (<= 2 ulp; the weights differ from qweights_kernel's by ~1e-15 relative).
// Every bin is owned by one lane and the nodes are added in a fixed order: bitwise reproducible.
#include "qpair.h"

#define QW_WAVES 2            // pairs per workgroup (independent waves: the workgroup only shares an LDS allocation)
#define QW_ARENA 1152         // doubles of table space per wave
#define QW_TPL 3              // tiles per lane per group
#define QW_TILES (64 * QW_TPL)

struct QWaveLds {
  double par[32];
  double tab[QW_ARENA];
  double dx[NS_MAX], dy[NS_MAX], dz[ZC];
  double c[NEDGE][NU_MAX];
  int inval[ZC];
  short coli[NS_MAX], colstart[NS_MAX + 1], jstart[NJ_MAX + 1], ustart[NU_MAX + 1];
  unsigned short boff[QW_TILES], li[QW_TILES];
  unsigned char ixord[NS_MAX], iyord[NS_MAX], zord[ZC], tmask[QW_TILES], culo[QW_TILES], cuhi[QW_TILES];
};

// =============================================================================================================
template <int M>
__global__ void __launch_bounds__(256) pair_setup_kernel(SplitArgs S, PairParams* __restrict__ pp, int qn_max) {
  const CurArgs& A = S.c;
  const LdsimConsts* c = A.c;
  const int64_t pair = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (pair >= A.n_pairs) return;
  PairParams P;
  memset(&P, 0, sizeof(P));
  int64_t seg, pID;
  {
    int32_t v = A.pair_val[pair];
    seg = A.seg_begin + v / A.P;
    pID = (int64_t)((A.pair_key[pair] >> 4) & 0xFFFFFFFFull);
  }
  int T = A.T;
  if (A.tmax_batch) T = min(T, A.tmax_batch[A.s.batch[seg] - A.batch0]);
  PairGeo g;
  pair_geometry(A, seg, pID, g);
  if (!g.ok) { pp[pair] = P; return; }
  const int NS = c->sampled_points;
  const double dt = c->time_sampling, bin = c->response_bin_size;
  // valid samples of either axis (inside the response table) and their extent relative to the segment start
  double xlo = 1e300, xhi = -1e300, ylo = 1e300, yhi = -1e300;
  for (int s = 0; s < NS; s++) {
    const double x = g.x_start + g.sgnx * (s * g.x_step - 4 * g.sT);
    const double xd = fabs(g.x_p - x);
    if (!(xd > bin * A.ni)) {
      const int i = (int)py_round(xd / bin - 0.5);
      if (i >= 0 && i < A.ni) { xlo = fmin(xlo, x - g.sx); xhi = fmax(xhi, x - g.sx); }
    }
    const double y = g.y_start + g.sgny * (s * g.y_step - 4 * g.sT);
    const double yd = fabs(g.y_p - y);
    if (!(yd > bin * A.nj)) {
      const int j = (int)py_round(yd / bin - 0.5);
      if (j >= 0 && j < A.nj) { ylo = fmin(ylo, y - g.sy); yhi = fmax(yhi, y - g.sy); }
    }
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
  int iz_lo = 0, iz_hi = g.z_steps - 1;
  if (A.prune_log > 0 && g.z_step > 0) {
    double cz = sqrt(2.0 * A.prune_log) * g.sL;
    double zl = g.sz - cz, zh = g.sz + g.Dz + cz;
    double fl = floor((zl - g.z_start_int) / g.z_step) - 1, fh = ceil((zh - g.z_start_int) / g.z_step) + 1;
    if (fl > iz_lo) iz_lo = (int)fmin(fl, (double)g.z_steps);
    if (fh < iz_hi) iz_hi = (int)fmax(fh, -1.0);
  }
  const double ux = g.Dx / g.Dr, uy = g.Dy / g.Dr, uz = g.Dz / g.Dr;
  const double i2T = 1.0 / (2 * g.sT2), i2L = 1.0 / (2 * g.sL2);
  const double a = ux * ux * i2T + uy * uy * i2T + uz * uz * i2L;
  const double factor = g.q / g.Dr / (g.s3 * sqrt(8 * M_PI * M_PI * M_PI));
  // the part of the segment that can reach the sample box (kernels_qweights.hip)
  const double G = sqrt(2.0 * ((A.prune_log > 0 ? A.prune_log : 43.0) + 7.0));
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
  const double nq_f = ceil(6.0 + 1.9 * qlen * sqrt(2.0 * a));
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
  P.thr = A.prune_log > 0 ? exp(-A.prune_log) * factor * g.dV * sqrt(M_PI / a) : 0.0;
  pp[pair] = P;
}

// =============================================================================================================
template <int M>
__global__ void __launch_bounds__(64 * QW_WAVES, 4) qwave_kernel(SplitArgs S, const PairParams* __restrict__ pp,
                                                               const double* __restrict__ glx, const double* __restrict__ glw) {
  const CurArgs& A = S.c;
  const LdsimConsts* c = A.c;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t pair = (int64_t)blockIdx.x * QW_WAVES + wv;
  if (pair >= A.n_pairs) return;
  __shared__ QWaveLds lds_all[QW_WAVES];
  QWaveLds& L = lds_all[wv];
  int32_t* hdr = S.hdr + pair * HDR_INTS;
  const PairParams* __restrict__ P = pp + pair;
  const int status = (A.debug_phases & 0x100) ? 0 : P->status;      // 0x100 (timing tools): stop after the record load
  if (status != 1) {           // nothing to emit, or handed to the monolithic kernel
    if (lane < HDR_INTS) hdr[lane] = (status == 2 && lane == 7) ? 1 : 0;
    if (status == 2 && lane == 0) atomicAdd(&A.counters[6], 1ull);
    return;
  }
  const int NS = c->sampled_points;
  const double bin = c->response_bin_size;
  const int NQ = P->NQ, iz_lo = P->iz_lo, iz_hi = P->iz_hi, it0 = P->it0, T = P->T, it_w0 = P->it_w0, it_w1 = P->it_w1;
  if (lane < PP_COUNT) L.par[lane] = ((const double*)((const char*)P + 32))[lane];
  wsync();

  // ---- sample -> response cell maps; member lists ordered by response index (lane = sample) --------------------------
  int ncol, jmin, jmax;
  {
    int i = -1;
    if (lane < NS) {
      const double x = L.par[PP_X_START] + L.par[PP_SGNX] * (lane * L.par[PP_X_STEP] - 4 * L.par[PP_ST]);
      const double xd = fabs(L.par[PP_X_P] - x);
      if (!(xd > bin * A.ni)) {
        i = (int)py_round(xd / bin - 0.5);
        if (i < 0 || i >= A.ni) i = -1;
      }
      L.dx[lane] = x - L.par[PP_SX];
    }
    int leader = lane;
    for (int q = 0; q < NS; q++) {
      int iq = __shfl(i, q);
      if (q < leader && iq == i) leader = q;
    }
    bool is_leader = (lane < NS) && (i >= 0) && (leader == lane);
    int slot = 0;   // rank of this column's i among the distinct i  -> cells come out sorted by (i, j)
    for (int q = 0; q < NS; q++) {
      int iq = __shfl(i, q);
      bool lq = __shfl((int)is_leader, q);
      if (lq && iq < i) slot++;
    }
    int myslot = (i < 0 || lane >= NS) ? -1 : slot;
    ncol = __popcll(__ballot(is_leader));
    if (is_leader) L.coli[slot] = (short)i;
    int posn = 0;
    for (int q = 0; q < NS; q++) {
      int sq = __shfl(myslot, q);
      if (sq >= 0 && myslot >= 0 && (sq < myslot || (sq == myslot && q < lane))) posn++;
    }
    if (myslot >= 0) L.ixord[posn] = (unsigned char)lane;
    if (is_leader) L.colstart[slot] = (short)posn;
    int nvalid = __popcll(__ballot(myslot >= 0));
    if (lane == 0) L.colstart[ncol] = (short)nvalid;
  }
  {
    int j = -1;
    if (lane < NS) {
      const double y = L.par[PP_Y_START] + L.par[PP_SGNY] * (lane * L.par[PP_Y_STEP] - 4 * L.par[PP_ST]);
      const double yd = fabs(L.par[PP_Y_P] - y);
      if (!(yd > bin * A.nj)) {
        j = (int)py_round(yd / bin - 0.5);
        if (j < 0 || j >= A.nj) j = -1;
      }
      L.dy[lane] = y - L.par[PP_SY];
    }
    jmin = (j >= 0) ? j : (1 << 20);
    jmax = j;
    for (int off = 32; off > 0; off >>= 1) {
      jmin = min(jmin, __shfl_xor(jmin, off));
      jmax = max(jmax, __shfl_xor(jmax, off));
    }
    int posn = 0, below = 0;
    for (int q = 0; q < NS; q++) {
      int jq = __shfl(j, q);
      if (jq >= 0 && j >= 0 && (jq < j || (jq == j && q < lane))) posn++;
      if (jq >= 0 && jq < jmin + lane) below++;
    }
    if (j >= 0) L.iyord[posn] = (unsigned char)lane;
    if (jmax >= jmin && lane <= jmax - jmin + 1 && lane <= NJ_MAX) L.jstart[lane] = (short)below;
  }
  wsync();
  if (A.debug_phases & 0x200) {      // timing tools: stop after the sample maps
    if (lane < HDR_INTS) hdr[lane] = 0;
    return;
  }
  const int NJ = jmax - jmin + 1;
  if (ncol == 0 || NJ <= 0 || NJ > NJ_MAX) {     // cannot happen for status 1 (the set-up pass saw valid samples); stay safe
    if (lane < HDR_INTS) hdr[lane] = 0;
    return;
  }
  int edge_k[NEDGE], k_stage_lo, k_stage_hi;
  edge_ks(c, A, edge_k, k_stage_lo, k_stage_hi);
  const double* __restrict__ gx_tab = glx + (int64_t)NQ * (NQ - 1) / 2;
  const double* __restrict__ gw_tab = glw + (int64_t)NQ * (NQ - 1) / 2;
  const bool do_prune = A.prune_log > 0;

  constexpr int IMAX = ItemCap<M>::value;
  Item* items = S.items + pair * IMAX;
  Corr* corr = S.corr + pair * CMAX;
  int n_items = 0, n_corr = 0, n_runs = 0, overflow = 0;      // wave-uniform

  int iz_next = iz_lo;
  while (iz_next <= iz_hi) {
    // ---- this chunk's slices: response shift, edge flags, member lists ordered by shift (lane = slice) ---------------------------
    int n_sl, u_min, NU, edge_mask;
    {
      const int nmax = min(ZC, iz_hi - iz_next + 1);
      int sh = 0, inval = 0;
      bool amb = false;
      if (lane < nmax) {
        double z, t0;
        const double t_start = L.par[PP_T_START];
        sh = slice_shift_of<M>(c, L.par[PP_Z_START_INT], L.par[PP_Z_STEP], L.par[PP_Z_ANODE], t_start, iz_next + lane, z, t0, amb);
        L.dz[lane] = z - L.par[PP_SZ];
#pragma unroll
        for (int e = 0; e < NEDGE; e++) {
          // a correction is needed only where the correlation would use this slice's weight at a tick the reference does
          // not: the edge index inside the staged response range, reachable by this shift and inside the stored window
          bool need = false;
          const int num = edge_k[e] - sh;
          if (edge_k[e] >= k_stage_lo && edge_k[e] <= k_stage_hi && num >= 0 && (num % M) == 0) {
            const int it_e = num / M;
            if (it_e >= max(it0, it_w0) && it_e < min(T, it_w1)) {
              int64_t kk;
              need = !(slice_valid_at(c, t_start, t0, it_e, kk) && kk == edge_k[e]);
            }
          }
          if (need) inval |= 1 << e;
        }
        L.inval[lane] = inval;
      }
      int pmin = lane < nmax ? sh : (1 << 30), pmax = lane < nmax ? sh : -(1 << 30);
      for (int off = 1; off < 64; off <<= 1) {
        int a1 = __shfl_up(pmin, off), a2 = __shfl_up(pmax, off);
        if (lane >= off) { pmin = min(pmin, a1); pmax = max(pmax, a2); }
      }
      bool fits = (lane < nmax) && (pmax - pmin + 1 <= NU_MAX);
      unsigned long long fm = __ballot(fits);
      n_sl = (fm == ~0ull) ? 64 : __ffsll((long long)~fm) - 1;
      const unsigned long long am = __ballot(amb && lane < n_sl);
      if (am && lane == 0) atomicAdd(&A.counters[0], (unsigned long long)__popcll(am));
      int lo = __shfl(pmin, n_sl - 1), hi = __shfl(pmax, n_sl - 1);
      int posn = 0, below = 0;
      for (int q = 0; q < n_sl; q++) {
        int sq = __shfl(sh, q);
        if (lane < n_sl && (sq < sh || (sq == sh && q < lane))) posn++;
        if (sq < lo + lane) below++;
      }
      if (lane < n_sl) L.zord[posn] = (unsigned char)lane;
      if (lane < hi - lo + 1) L.ustart[lane] = (short)below;
      if (lane == 0) L.ustart[hi - lo + 1] = (short)n_sl;
      int anyinv = (lane < n_sl) ? inval : 0;
      for (int off = 32; off > 0; off >>= 1) anyinv |= __shfl_xor(anyinv, off);
      edge_mask = anyinv;
      u_min = lo;
      NU = hi - lo + 1;
      if (lane == 0 && n_runs < RUNS_MAX) hdr[8 + n_runs] = n_items;
      if (n_runs >= RUNS_MAX) overflow = 1;
      n_runs++;
    }
    for (int i = lane; i < NEDGE * NU_MAX; i += 64) (&L.c[0][0])[i] = 0;
    wsync();
    const int NU8 = (NU + 7) & ~7, NB8 = NU8 >> 3;
    // table rows: X[n][ncol], Y[n][NJ], Z[n][NU8], odd strides (lane = node and lane = bin both conflict-free)
    const int XS = ncol | 1, YS = NJ | 1, ZS = NU8 + 1;
    // Z tasks: bins padded to a power of two so that a lane keeps its bin across task rounds (edge sums stay in a register)
    const int ZP = NU8 <= 8 ? 8 : NU8 <= 16 ? 16 : NU8 <= 32 ? 32 : 64;
    const int PN = XS + YS + ZS;
    const int nres = max(1, min(64, QW_ARENA / PN));      // nodes whose tables fit the arena (lane = node for the edge form)
    const bool single_pass = NQ <= nres;
    bool tables_valid = false;
    const int cells_per_group = QW_TILES / NB8;           // >= 24; a group is a run of cells in (column, j) order
    const int ncell_tot = ncol * NJ;

    for (int cell0 = 0; cell0 < ncell_tot; cell0 += cells_per_group) {
      const int ncell = min(cells_per_group, ncell_tot - cell0);
      const int ntiles = ncell * NB8;
      int tcell[QW_TPL], tblk[QW_TPL];
      double acc[QW_TPL][8];
#pragma unroll
      for (int r = 0; r < QW_TPL; r++) {
        const int t = lane + r * 64;
        tcell[r] = t < ntiles ? t / NB8 : -1;
        tblk[r] = t < ntiles ? t - tcell[r] * NB8 : 0;
#pragma unroll
        for (int q = 0; q < 8; q++) acc[r][q] = 0;
      }
      for (int n0 = 0; n0 < NQ; n0 += nres) {
        const int nb = min(nres, NQ - n0);
        double* const tX = L.tab;
        double* const tY = L.tab + nb * XS;
        double* const tZ = tY + nb * YS;
        if (!tables_valid) {
          wsync();      // the previous pass's tables are no longer read
          // ---- X and Y tables: one task per (node, bin), bin fastest ---------------------------------------------------------
          const int nX = ncol * nb, nY = NJ * nb;
          const double s_lo = L.par[PP_S_LO], hq = 0.5 * L.par[PP_QLEN], i2T = L.par[PP_I2T];
          const double uxr = L.par[PP_UXR], uyr = L.par[PP_UYR];
          for (int task = lane; task < nX + nY && (A.debug_phases & 1); task += 64) {
            const bool isx = task < nX;
            const int rel = isx ? task : task - nX;
            const int nbins = isx ? ncol : NJ;
            const int n = rel / nbins, b = rel - n * nbins;
            const double sn = s_lo + hq * (1.0 + gx_tab[n0 + n]);
            double sum = 0;
            if (isx) {
              const double cen = sn * uxr;
              for (int k = L.colstart[b]; k < L.colstart[b + 1]; k++) {
                const double d = L.dx[L.ixord[k]] - cen;
                sum += exp_neg(-d * d * i2T);
              }
              tX[n * XS + b] = sum;
            } else {
              const double cen = sn * uyr;
              for (int k = L.jstart[b]; k < L.jstart[b + 1]; k++) {
                const double d = L.dy[L.iyord[k]] - cen;
                sum += exp_neg(-d * d * i2T);
              }
              tY[n * YS + b] = sum;
            }
          }
          wsync();
          // ---- window-edge bilinear form, lane = node:  Q_e[n] = sum_cells X[n][col] Y[n][j] R[cell][edge_k[e]] ----------------
          double qn[NEDGE] = {0, 0, 0};
          const bool edges_now = edge_mask && cell0 == 0;     // once per pass: the X table holds every column
          if (edges_now && lane < nb && (A.debug_phases & 4)) {
#pragma unroll
            for (int e = 0; e < NEDGE; e++) {
              if (!(edge_mask & (1 << e))) continue;
              double qv = 0;
              for (int col = 0; col < ncol; col++) {
                const double xv = tX[lane * XS + col];
                const int64_t rbase = ((int64_t)L.coli[col] * A.nj + jmin) * A.nk + edge_k[e];
                for (int jj = 0; jj < NJ; jj++) qv = fma(xv * tY[lane * YS + jj], A.resp[rbase + (int64_t)jj * A.nk], qv);
              }
              qn[e] = qv;
            }
          }
          // ---- Z table: one task per (node, padded bin), bin fastest; flagged slices feed the edge corrections -----------------
          double cacc[NEDGE] = {0, 0, 0};
          const double uzr = L.par[PP_UZR], i2L = L.par[PP_I2L], wscale = L.par[PP_WSCALE], kappa = L.par[PP_KAPPA];
          const int zb = lane & (ZP - 1);                     // this lane's bin in every round
          const int zrounds = (A.debug_phases & 2) ? (ZP * nb + 63) >> 6 : 0;
          for (int rr = 0; rr < zrounds; rr++) {              // uniform trip count: the broadcast of Q below needs every lane
            const int n = (lane + 64 * rr) / ZP;
            const bool live = n < nb && zb < NU8;
            double sumi[NEDGE] = {0, 0, 0};
            double wn = 0;
            if (live) {
              double sum = 0;
              if (zb < NU) {
                const double sn = s_lo + hq * (1.0 + gx_tab[n0 + n]);
                const double cen = sn * uzr;
                for (int k = L.ustart[zb]; k < L.ustart[zb + 1]; k++) {
                  const int sl = L.zord[k];
                  const double d = L.dz[sl] - cen;
                  const double ev = exp_neg(-d * d * i2L);
                  sum += ev;
                  if (edges_now) {
                    const int iv = L.inval[sl];
#pragma unroll
                    for (int e = 0; e < NEDGE; e++)
                      if (iv & (1 << e)) sumi[e] += ev;
                  }
                }
                wn = wscale * gw_tab[n0 + n];
                if (kappa != 0.0) wn *= exp_neg(-sn * sn * kappa);
              }
              tZ[n * ZS + zb] = wn * sum;       // shifts beyond NU inside the last 8-block read as zero
            }
            if (edges_now) {
#pragma unroll
              for (int e = 0; e < NEDGE; e++)
                if (edge_mask & (1 << e)) {
                  const double qb = __shfl(qn[e], live ? n : 0);
                  if (live) cacc[e] = fma(wn * sumi[e], qb, cacc[e]);
                }
            }
          }
          if (edges_now) {
#pragma unroll
            for (int e = 0; e < NEDGE; e++) {
              if (!(edge_mask & (1 << e))) continue;
              double v = cacc[e];
              for (int off = ZP; off < 64; off <<= 1) v += __shfl_xor(v, off);     // lanes with equal lane % ZP
              if (lane < NU) L.c[e][lane] += v;
            }
          }
          wsync();
          tables_valid = single_pass;
        }
        // ---- this pass's share of the bins ------------------------------------------------------------------------------------
#pragma unroll
        for (int r = 0; r < QW_TPL; r++) {
          if (tcell[r] >= 0 && (A.debug_phases & 8)) {
            const int col = (cell0 + tcell[r]) / NJ, jj = (cell0 + tcell[r]) - col * NJ;
            const double* xr = tX + col;
            const double* yr = tY + jj;
            const double* zr = tZ + tblk[r] * 8;
            for (int n = 0; n < nb; n++) {
              const double xy = xr[n * XS] * yr[n * YS];
#pragma unroll
              for (int q = 0; q < 8; q++) acc[r][q] = fma(xy, zr[n * ZS + q], acc[r][q]);
            }
          }
        }
      }
      // ---- active bins of every cell, items, pool offsets ---------------------------------------------------------------------------
#pragma unroll
      for (int r = 0; r < QW_TPL; r++) {
        if (tcell[r] >= 0) {
          unsigned m = 0;
#pragma unroll
          for (int q = 0; q < 8; q++) {
            const bool in = tblk[r] * 8 + q < NU;
            if (in && (do_prune ? acc[r][q] > L.par[PP_THR] : acc[r][q] != 0.0)) m |= 1u << q;
          }
          L.tmask[lane + r * 64] = (unsigned char)m;
        }
      }
      wsync();
      int nact = 0, nblk_tot = 0;
      for (int base = 0; base < ncell; base += 64) {
        const int cell = base + lane;
        unsigned long long mk = 0;
        if (cell < ncell)
          for (int b8 = 0; b8 < NB8; b8++) mk |= (unsigned long long)L.tmask[cell * NB8 + b8] << (8 * b8);
        const bool act = mk != 0;
        const int ulo = act ? __ffsll((long long)mk) - 1 : 0, uhi = act ? 63 - __clzll((long long)mk) : 0;
        unsigned long long am = __ballot(act);
        int nb8 = act ? ((uhi - (ulo & ~7)) / 8 + 1) : 0;
        int sc = nb8;
        for (int off = 1; off < 64; off <<= 1) {
          int o = __shfl_up(sc, off);
          if (lane >= off) sc += o;
        }
        if (cell < ncell) {
          L.culo[cell] = act ? (unsigned char)ulo : (unsigned char)255;
          L.cuhi[cell] = (unsigned char)uhi;
          L.li[cell] = (unsigned short)(nact + __popcll(am & ((1ull << lane) - 1ull)));
          L.boff[cell] = (unsigned short)(nblk_tot + sc - nb8);
        }
        nact += __popcll(am);
        nblk_tot += __shfl(sc, 63);
      }
      unsigned long long base = 0;
      {
        const int item_cap = (A.split_max_items > 0 && A.split_max_items < IMAX) ? A.split_max_items : IMAX;
        int ok = (n_items + nact <= item_cap) && !overflow;
        const unsigned long long need = (unsigned long long)nblk_tot * 8ull;
        if (ok && need) {
          unsigned long long b0 = 0;
          if (lane == 0) b0 = atomicAdd(S.cursor, need);
          base = (unsigned long long)__shfl((long long)b0, 0);
          if (base + need > S.wbuf_cap) ok = 0;
        }
        if (!ok) overflow = 1;
      }
      wsync();
      if (!overflow && nact > 0) {
#pragma unroll
        for (int r = 0; r < QW_TPL; r++) {
          const int cell = tcell[r];
          if (cell < 0 || L.culo[cell] == 255) continue;
          const int ulo = L.culo[cell], uhi = L.cuhi[cell];
          const int ulo8 = ulo & ~7, u0 = tblk[r] * 8;
          if (u0 < ulo8 || u0 > uhi) continue;
          const unsigned long long wo = base + (unsigned long long)L.boff[cell] * 8ull;
          double* dst = S.wbuf + wo + (u0 - ulo8);
#pragma unroll
          for (int q = 0; q < 8; q++) dst[q] = (u0 + q >= ulo && u0 + q <= uhi) ? acc[r][q] : 0.0;
          if (u0 == ulo8) {
            const int col = (cell0 + cell) / NJ, jj = (cell0 + cell) - col * NJ;
            Item itx;
            itx.cell_nblk = (L.coli[col] * A.nj + (jmin + jj)) | (((uhi - ulo8) / 8 + 1) << 16);
            itx.sbase = u_min + ulo8;
            itx.woff_lo = (uint32_t)(wo & 0xFFFFFFFFull);
            itx.woff_hi = (uint32_t)(wo >> 32);
            items[n_items + L.li[cell]] = itx;
          }
        }
        n_items += nact;
      }
      wsync();
    }
    // ---- window-edge corrections of this chunk -> (tick, value) list ------------------------------------------------
    if (edge_mask) {
      wsync();
      for (int e = 0; e < NEDGE; e++) {
        double cv = (lane < NU) ? L.c[e][lane] : 0.0;
        int num = edge_k[e] - (u_min + lane);
        bool ok = (edge_k[e] >= 0) && (lane < NU) && cv != 0.0 && num >= 0 && (num % M) == 0;
        unsigned long long om = __ballot(ok);
        int cnt = __popcll(om);
        if (n_corr + cnt > CMAX) {
          overflow = 1;
        } else {
          if (ok) {
            Corr cr;
            cr.tick = num / M;
            cr.pad = 0;
            cr.val = cv;
            corr[n_corr + __popcll(om & ((1ull << lane) - 1ull))] = cr;
          }
          n_corr += cnt;
        }
      }
    }
    iz_next += n_sl;
  }
  if (lane == 0) {
    atomicAdd(&A.counters[1], (unsigned long long)NQ);
    hdr[0] = overflow ? 0 : n_items;
    hdr[1] = overflow ? 0 : n_corr;
    hdr[2] = it0;
    hdr[3] = T;
    hdr[4] = it_w0;
    hdr[5] = it_w1;
    hdr[6] = overflow ? 0 : min(n_runs, RUNS_MAX);
    hdr[7] = overflow;              // 1 = capacity overflow: the monolithic kernel recomputes this pair
    hdr[8 + min(n_runs, RUNS_MAX)] = n_items;
    if (overflow) atomicAdd(&A.counters[6], 1ull);
  }
}

extern "C++" size_t qwave_params_bytes(int64_t n_pairs) { return (size_t)n_pairs * sizeof(PairParams); }

// the per-pair records of both quadrature weight kernels
extern "C++" int qpair_setup_launch(ldsim_ctx* ctx, const SplitArgs& S, int M, void* params) {
  if (S.c.n_pairs == 0) return 0;
  if (!ctx->d_glx || !ctx->d_glw || !params) {
    ldsim_set_error("Gauss-Legendre tables / pair parameter buffer missing");
    return LDSIM_ESTATE;
  }
  PairParams* pp = (PairParams*)params;
  const unsigned g0 = (unsigned)((S.c.n_pairs + 255) / 256);
  if (M == 1) hipLaunchKernelGGL(pair_setup_kernel<1>, dim3(g0), dim3(256), 0, ctx->stream, S, pp, ctx->gl_nmax);
  else hipLaunchKernelGGL(pair_setup_kernel<2>, dim3(g0), dim3(256), 0, ctx->stream, S, pp, ctx->gl_nmax);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C++" int qwave_launch(ldsim_ctx* ctx, const SplitArgs& S, int M, void* params) {
  if (S.c.n_pairs == 0) return 0;
  int rc = qpair_setup_launch(ctx, S, M, params);
  if (rc) return rc;
  const PairParams* pp = (const PairParams*)params;
  const unsigned g1 = (unsigned)((S.c.n_pairs + QW_WAVES - 1) / QW_WAVES);
  if (M == 1)
    hipLaunchKernelGGL(qwave_kernel<1>, dim3(g1), dim3(64 * QW_WAVES), 0, ctx->stream, S, pp, ctx->d_glx, ctx->d_glw);
  else
    hipLaunchKernelGGL(qwave_kernel<2>, dim3(g1), dim3(64 * QW_WAVES), 0, ctx->stream, S, pp, ctx->d_glx, ctx->d_glw);
  HIPCHK(hipGetLastError());
  return 0;
}
