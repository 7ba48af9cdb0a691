// kernels_qweights.hip -- a9-a12, weights stage of the split path, default form ("weights_mode" 1).
//
// The reference evaluates, for every charge sample (ix, iy, iz) of a (segment, pixel) pair, the closed form of the line
// integral of a 3-D Gaussian (detsim.py:120-159: 2 erf, exp, 2 log) -- ~6.4e4 samples per pair.  The closed form is
//      rho(x, y, z) = q / (Dr (2 pi)^(3/2) sT sT sL)  *  Int_0^Dr ds  gT(x - sx - s ux) gT(y - sy - s uy) gL(z - sz - s uz)
// and the integrand is a product of three 1-D Gaussians.  A Gauss-Legendre rule along the segment (N nodes, chosen from
// the segment length in units of the Gaussian's width along it so that the rule is exact to 1e-12 of the on-axis density;
// measured in tools/quad_nodes.py) therefore turns the binned weights into a sum of N separable terms
//      A[i][j][shift] = sum_n  w_n  X_n[i] Y_n[j] Z_n[shift],     X_n[i] = sum_{ix in response column i} gT(x_ix - sx - s_n ux), ...
// i.e. ~120 N one-dimensional exponentials and a small (cells x shifts x N) product per pair instead of 6.4e4 closed-form
// evaluations.  All terms are positive (no erf cancellation), every bin is owned by one thread and the nodes are added in
// a fixed order, so the weights are bitwise reproducible.  The sample grid, the response cell of every sample, the
// response shift of every z slice and the window-edge predicates are the reference's own (shared with weights_kernel).
//
// Pairs whose segment is longer than ~130 widths (N > 256: sT -> 0 next to the anode) and pairs that exceed the item /
// correction / run capacities are flagged and recomputed by the monolithic current_kernel, like in weights_kernel.
#include "qpair.h"
#include "wave_ops.h"

#define QNB 16            // quadrature nodes per batch (tables of one batch live in LDS)
#define QTILES 512        // (cell, 8-shift block) tiles per column group: two per thread, accumulated in registers
#define QCOLS 32          // response columns per group
#define QN_LDS 64         // nodes of a rule kept in LDS (longer rules -- segments of > 37 Gaussian widths -- read the table)
#define Q_CELLS 512       // response cells per group

template <int M>
__global__ void __launch_bounds__(CUR_THREADS, 4) qweights_kernel(SplitArgs S, const PairParams* __restrict__ pp,
                                                                  const double* __restrict__ glx,
                                                                  const double* __restrict__ glw) {
  const CurArgs& A = S.c;
  const LdsimConsts* c = A.c;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t pair = blockIdx.x;
  if (pair >= A.n_pairs) return;
  int32_t* hdr = S.hdr + pair * HDR_INTS;

  // the pair's constants come from pair_setup_kernel's record (one thread per pair there; here every wave of the workgroup
  // used to repeat the geometry): the doubles sit in LDS and are read where a phase needs them
  __shared__ double s_par[32];
  const PairParams* __restrict__ P = pp + pair;
  const int status = (A.debug_phases & 0x100) ? 0 : P->status;
  if (status != 1) {           // nothing to emit, or handed to the monolithic kernel
    if (tid < HDR_INTS) hdr[tid] = (status == 2 && tid == 7) ? 1 : 0;
    if (status == 2 && tid == 0) stat_add(A.counters, 6, 1ull);
    return;
  }
  const int NQ = P->NQ, iz_lo = P->iz_lo, iz_hi = P->iz_hi, it0 = P->it0, T = P->T, it_w0 = P->it_w0, it_w1 = P->it_w1;
  if (tid < PP_COUNT) s_par[tid] = ((const double*)((const char*)P + 32))[tid];
  __syncthreads();
  auto write_empty = [&]() {
    if (tid < HDR_INTS) hdr[tid] = 0;
  };
  // the pair's Gauss-Legendre nodes and weights (rule NQ of the table) into LDS: the table tasks read them per task, and
  // from global memory every one of those reads was a full L2 round trip inside a latency-bound phase
  __shared__ double s_gl[2][QN_LDS];
  {
    const int64_t off = (int64_t)NQ * (NQ - 1) / 2;
    if (tid < NQ && tid < QN_LDS) {
      s_gl[0][tid] = glx[off + tid];
      s_gl[1][tid] = glw[off + tid];
    }
  }
  const int NS = c->sampled_points;
  const double bin = c->response_bin_size;

  __shared__ double s_X[QNB][QCOLS], s_Y[QNB][NJ_MAX], s_Z[QNB][ZC], s_Zi[QNB][ZC];
  __shared__ double s_C[NEDGE][NU_MAX];
  __shared__ double s_dx[NS_MAX], s_dy[NS_MAX], s_dz[ZC], s_Q[QNB];
  // the same offsets in member-list order (position k of s_ixord / s_iyord / s_zord): the table tasks walk a bin's members
  // with one LDS read per member instead of an index read followed by a dependent value read
  __shared__ double s_dxs[NS_MAX], s_dys[NS_MAX], s_dzs[ZC];
  __shared__ unsigned char s_invs[ZC];
  __shared__ int s_shift[ZC], s_inval[ZC];
  __shared__ short s_icell[NS_MAX], s_jcell[NS_MAX], s_colof[NS_MAX], s_coli[NS_MAX], s_colstart[NS_MAX + 1];
  __shared__ short s_jstart[NJ_MAX + 1], s_ustart[NU_MAX + 1];
  __shared__ unsigned char s_ixord[NS_MAX], s_iyord[NS_MAX], s_zord[ZC];
  __shared__ unsigned char s_tmask[QTILES], s_culo[Q_CELLS], s_cuhi[Q_CELLS];
  __shared__ unsigned short s_boff[Q_CELLS], s_li[Q_CELLS];
  __shared__ double s_rng[8];
  __shared__ int s_misc[24];
  __shared__ unsigned long long s_base64;

  // ---- one chunk's slices: response shift, edge flags, member lists ordered by shift.  One wave: wave 2 for the first
  // chunk, while waves 0 / 1 build the sample maps (nothing here depends on them); wave 0 for the chunks after it.
  int edge_k[NEDGE], k_stage_lo, k_stage_hi;
  edge_ks(c, A, edge_k, k_stage_lo, k_stage_hi);
  auto chunk_setup = [&](int iz_first, bool first) {
    int nmax = min(ZC, iz_hi - iz_first + 1);
    int sh = 0, inval = 0;
    double dzv = 0;
    if (lane < nmax) {
      double z, t0;
      bool amb;
      sh = slice_shift_of<M>(c, s_par[PP_Z_START_INT], s_par[PP_Z_STEP], s_par[PP_Z_ANODE], s_par[PP_T_START], iz_first + lane, z,
                             t0, amb);
      if (amb) stat_add(A.counters, 0, 1ull);
      dzv = z - s_par[PP_SZ];
      s_dz[lane] = dzv;
      s_shift[lane] = sh;
#pragma unroll
      for (int e = 0; e < NEDGE; e++) {
        // a correction is needed only where the correlation would use this slice's weight at a tick the reference
        // does not: the edge index must be inside the staged response range, reachable by this shift (an integer tick)
        // and that tick inside the stored window; everything else is dropped at the end anyway
        bool need = false;
        const int num = edge_k[e] - sh;
        if (edge_k[e] >= k_stage_lo && edge_k[e] <= k_stage_hi && num >= 0 && (num % M) == 0) {
          const int it_e = num / M;
          if (it_e >= max(it0, it_w0) && it_e < min(T, it_w1)) {
            int64_t kk;
            need = !(slice_valid_at(c, s_par[PP_T_START], t0, it_e, kk) && kk == edge_k[e]);
          }
        }
        if (need) inval |= 1 << e;
      }
      s_inval[lane] = inval;
    }
    const int pmin = wave_scan_i32(lane < nmax ? sh : (1 << 30), 0x7fffffff, [](int a, int b) { return a < b ? a : b; });
    const int pmax = wave_scan_i32(lane < nmax ? sh : -(1 << 30), (int)0x80000000, [](int a, int b) { return a > b ? a : b; });
    bool fits = (lane < nmax) && (pmax - pmin + 1 <= NU_MAX);
    unsigned long long fm = __ballot(fits);
    int n = (fm == ~0ull) ? 64 : __ffsll((long long)~fm) - 1;
    int lo = __builtin_amdgcn_readlane(pmin, n - 1), hi = __builtin_amdgcn_readlane(pmax, n - 1);
    // slices of the chunk ordered by shift (|z - z_anode| need not be monotone in iz): s_zord, s_ustart[u]
    int posn = 0, below = 0, anyinv = 0;
    {
      const unsigned long long lane_lt = (1ull << lane) - 1ull;
      for (int bs = lo; bs <= hi; bs++) {                    // the chunk's distinct shifts (<= NU_MAX), one ballot each
        const unsigned long long bal = __ballot(lane < n && sh == bs);
        const int cnt = __popcll(bal);
        if (lane < n && sh > bs) posn += cnt;
        if (lane < n && sh == bs) posn += __popcll(bal & lane_lt);
        if (bs < lo + lane) below += cnt;
      }
    }
    if (lane < n) {
      s_zord[posn] = (unsigned char)lane;
      s_dzs[posn] = dzv;
      s_invs[posn] = (unsigned char)inval;
    }
    if (lane < hi - lo + 1) s_ustart[lane] = (short)below;
    if (lane == 0) s_ustart[hi - lo + 1] = (short)n;
    anyinv = 0;
#pragma unroll
    for (int e = 0; e < NEDGE; e++)
      if (__ballot(lane < n && (inval & (1 << e)))) anyinv |= 1 << e;
    if (lane == 0) {
      s_misc[3] = n; s_misc[4] = lo; s_misc[5] = hi;
      s_misc[20] = anyinv;
      const int r = first ? 0 : s_misc[19];            // the first chunk is set up next to the maps, before anyone wrote these
      if (r < RUNS_MAX) hdr[8 + r] = first ? 0 : s_misc[16]; else s_misc[18] = 1;
      s_misc[19] = r + 1;
    }
  };

  // ---- sample -> response cell maps; member lists ordered by response index (x: wave 0, y: wave 1) ----------------
  // (lane values are fetched with v_readlane -- the lane index is the loop counter, wave-uniform -- not with a shuffle through
  // the LDS crossbar, whose latency a non-unrolled loop pays on every iteration)
  if (wv == 0) {
    int i = -1;
    double ddx = 0;
    if (lane < NS) {
      double x = s_par[PP_X_START] + s_par[PP_SGNX] * (lane * s_par[PP_X_STEP] - 4 * s_par[PP_ST]);
      double xd = fabs(s_par[PP_X_P] - x);
      if (!(xd > bin * A.ni)) {
        i = (int)py_round(xd / bin - 0.5);
        if (i < 0 || i >= A.ni) i = -1;
      }
      s_icell[lane] = (short)i;
      ddx = x - s_par[PP_SX];
      s_dx[lane] = ddx;
    }
    // Cells present, as a bit mask over i (ni < 64); the loop below runs over its set bits -- the handful of distinct cells
    // the 40 samples fall into -- with one ballot each, instead of over the 40 lanes three times.
    const bool valid = lane < NS && i >= 0;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    unsigned long long present = 0;
    {
      const int mlo = wave_lane_i32(wave_scan_i32(valid && i < 32 ? (1 << i) : 0, 0, [](int a, int b) { return a | b; }), 63);
      const int mhi = wave_lane_i32(wave_scan_i32(valid && i >= 32 ? (1 << (i - 32)) : 0, 0, [](int a, int b) { return a | b; }), 63);
      present = (unsigned long long)(unsigned)mlo | ((unsigned long long)(unsigned)mhi << 32);
    }
    const int slot = valid ? __popcll(present & ((1ull << i) - 1ull)) : 0;    // rank of this column's i among the distinct i
    const int ncol = __popcll(present);
    int posn = 0;
    bool is_leader = false;
    for (unsigned long long m = present; m; m &= m - 1) {
      const int bcell = __ffsll((long long)m) - 1;
      const unsigned long long bal = __ballot(valid && i == bcell);
      if (valid && i > bcell) posn += __popcll(bal);
      if (valid && i == bcell) {
        const int before = __popcll(bal & lane_lt);
        posn += before;
        is_leader = before == 0;
      }
    }
    const int myslot = valid ? slot : -1;
    if (lane < NS) s_colof[lane] = (short)myslot;
    if (is_leader) s_coli[slot] = (short)i;
    if (myslot >= 0) {
      s_ixord[posn] = (unsigned char)lane;
      s_dxs[posn] = ddx;
    }
    if (is_leader) s_colstart[slot] = (short)posn;
    int nvalid = __popcll(__ballot(myslot >= 0));
    const double lo = wave_min_f64(myslot >= 0 ? ddx : 1e300), hi = wave_max_f64(myslot >= 0 ? ddx : -1e300);
    if (lane == 0) {
      s_colstart[ncol] = (short)nvalid;
      s_misc[0] = ncol;
      s_rng[0] = lo;
      s_rng[1] = hi;
      s_misc[16] = 0;   // items emitted
      s_misc[17] = 0;   // corrections emitted
      s_misc[18] = 0;   // overflow (s_misc[19], the run count, is set by the first chunk's set-up on wave 2)
      s_misc[21] = 0;   // weights emitted that are not block padding
    }
  } else if (wv == 1) {
    int j = -1;
    double ddy = 0;
    if (lane < NS) {
      double y = s_par[PP_Y_START] + s_par[PP_SGNY] * (lane * s_par[PP_Y_STEP] - 4 * s_par[PP_ST]);
      double yd = fabs(s_par[PP_Y_P] - y);
      if (!(yd > bin * A.nj)) {
        j = (int)py_round(yd / bin - 0.5);
        if (j < 0 || j >= A.nj) j = -1;
      }
      s_jcell[lane] = (short)j;
      ddy = y - s_par[PP_SY];
      s_dy[lane] = ddy;
    }
    const int jmin = wave_min_i32((j >= 0) ? j : (1 << 20)), jmax = wave_max_i32(j);
    const double lo = wave_min_f64(j >= 0 ? ddy : 1e300), hi = wave_max_f64(j >= 0 ? ddy : -1e300);
    // members of one j are contiguous in s_iyord; s_jstart[j'] = number of valid samples with j < jmin + j'
    int posn = 0, below = 0;
    if (jmax >= jmin) {
      const unsigned long long lane_lt = (1ull << lane) - 1ull;
      for (int bj = jmin; bj <= jmax; bj++) {               // the few distinct j, one ballot each
        const unsigned long long bal = __ballot(j == bj);
        const int cnt = __popcll(bal);
        if (j > bj) posn += cnt;
        if (j == bj) posn += __popcll(bal & lane_lt);
        if (bj < jmin + lane) below += cnt;
      }
    }
    if (j >= 0) {
      s_iyord[posn] = (unsigned char)lane;
      s_dys[posn] = ddy;
    }
    if (jmax >= jmin && lane <= jmax - jmin + 1 && lane <= NJ_MAX) s_jstart[lane] = (short)below;   // nj <= NJ_MAX < 64
    if (lane == 0) {
      s_misc[1] = jmin;
      s_misc[2] = jmax;
      s_rng[2] = lo;
      s_rng[3] = hi;
    }
  } else if (wv == 2) {
    if (iz_lo <= iz_hi) chunk_setup(iz_lo, true);
    else if (lane == 0) s_misc[19] = 0;
  }
  __syncthreads();
  const int ncol = s_misc[0], jmin = s_misc[1], jmax = s_misc[2];
  const int NJ = jmax - jmin + 1;
  if (ncol == 0 || NJ <= 0 || NJ > NJ_MAX) { write_empty(); return; }
  if (A.debug_phases & 0x200) { write_empty(); return; }      // timing tools: stop after the sample maps

  // this pair's rule: staged in LDS before the maps, or (more than QN_LDS nodes) read from the table
  const double* gx_tab = NQ <= QN_LDS ? (const double*)s_gl[0] : glx + (int64_t)NQ * (NQ - 1) / 2;
  const double* gw_tab = NQ <= QN_LDS ? (const double*)s_gl[1] : glw + (int64_t)NQ * (NQ - 1) / 2;
  const bool do_prune = A.prune_log > 0;
  // numba_f32: _b divides by sigma*sigma typed f32 while delta and a use the f64 square (detsim.py:116-118,141-148), so per
  // axis the exponent is -[(d - r u s)^2 + u^2 s^2 (1 - r^2)] / (2 sigma^2), r = sigma^2 / (sigma*sigma)_f32: still one
  // Gaussian per axis (centre scaled by r) times a factor that depends on the node only (pair_setup_kernel: uxr.., kappa).
  const double uxr = s_par[PP_UXR], uyr = s_par[PP_UYR], uzr = s_par[PP_UZR], i2T = s_par[PP_I2T], i2L = s_par[PP_I2L];
  const double kappa = s_par[PP_KAPPA], s_lo = s_par[PP_S_LO], qlen = s_par[PP_QLEN], wscale = s_par[PP_WSCALE];
  // bins below exp(-prune_log) of the weight of an on-axis interior sample are not emitted
  const double thr = s_par[PP_THR];
  if (A.debug_phases & 0x400) { write_empty(); return; }      // timing tools: stop before the chunk loop

  constexpr int IMAX = ItemCap<M>::value;
  Item* items = S.items + pair * IMAX;
  Corr* corr = S.corr + pair * CMAX;

  int iz_next = iz_lo;
  while (iz_next <= iz_hi) {
    __syncthreads();
    if (wv == 0 && iz_next != iz_lo) chunk_setup(iz_next, false);
    for (int i = tid; i < NEDGE * NU_MAX; i += CUR_THREADS) (&s_C[0][0])[i] = 0;
    __syncthreads();
    const int n_sl = s_misc[3], u_min = s_misc[4];
    const int NU = s_misc[5] - u_min + 1;
    const int NU8 = (NU + 7) & ~7, NB8 = NU8 >> 3;
    const int edge_mask = s_misc[20];
    const int cols_per_group = max(1, min(min(QTILES / (NJ * NB8), Q_CELLS / NJ), QCOLS));

    for (int col0 = 0; col0 < ncol; col0 += cols_per_group) {
      const int gcols = min(cols_per_group, ncol - col0);
      const int ncell = gcols * NJ;
      const int ntiles = ncell * NB8;
      // this thread's tiles: (cell, 8-shift block), block index fastest
      int tcell[2], tblk[2];
      double acc[2][8];
#pragma unroll
      for (int r = 0; r < 2; r++) {
        const int t = tid + r * CUR_THREADS;
        tcell[r] = t < ntiles ? t / NB8 : -1;
        tblk[r] = t < ntiles ? t - tcell[r] * NB8 : 0;
#pragma unroll
        for (int q = 0; q < 8; q++) acc[r][q] = 0;
      }
      for (int n0 = 0; n0 < NQ; n0 += QNB) {
        const int nb = min(QNB, NQ - n0);
        __syncthreads();      // the previous batch's tables are no longer read
        // ---- tables of this node batch: one task per (table bin, node), node fastest -------------------------------------------
        {
          const int nX = gcols * nb, nY = NJ * nb, nZ = NU * nb;
          const int ntask = nX + nY + nZ;
          for (int task = tid; task < ntask && (A.debug_phases & 1); task += CUR_THREADS) {
            int kind, rel;
            if (task < nX) { kind = 0; rel = task; }
            else if (task < nX + nY) { kind = 1; rel = task - nX; }
            else { kind = 2; rel = task - nX - nY; }
            const int b = rel / nb, n = rel - b * nb;
            const double sn = s_lo + 0.5 * qlen * (1.0 + gx_tab[n0 + n]);
            if (kind == 0) {
              const double cen = sn * uxr;
              double sum = 0;
              for (int k = s_colstart[col0 + b]; k < s_colstart[col0 + b + 1]; k++) {
                const double d = s_dxs[k] - cen;
                sum += exp_neg(-d * d * i2T);
              }
              s_X[n][b] = sum;
            } else if (kind == 1) {
              const double cen = sn * uyr;
              double sum = 0;
              for (int k = s_jstart[b]; k < s_jstart[b + 1]; k++) {
                const double d = s_dys[k] - cen;
                sum += exp_neg(-d * d * i2T);
              }
              s_Y[n][b] = sum;
            } else {
              const double cen = sn * uzr;
              double sum = 0, sumi = 0;
              for (int k = s_ustart[b]; k < s_ustart[b + 1]; k++) {
                const double d = s_dzs[k] - cen;
                const double e = exp_neg(-d * d * i2L);
                sum += e;
                if (s_invs[k]) sumi += e;        // refined per edge below when several edges are flagged
              }
              double wn = wscale * gw_tab[n0 + n];
              if (kappa != 0.0) wn *= exp_neg(-sn * sn * kappa);
              s_Z[n][b] = wn * sum;
              s_Zi[n][b] = wn * sumi;
            }
          }
          // shifts beyond NU inside the last 8-block read as zero
          for (int i = tid; i < nb * (NU8 - NU); i += CUR_THREADS) {
            const int n = i / (NU8 - NU), u = NU + i % (NU8 - NU);
            s_Z[n][u] = 0;
          }
        }
        __syncthreads();
        // ---- this batch's share of the bins -----------------------------------------------------------------------------------
#pragma unroll
        for (int r = 0; r < 2; r++) {
          if (tcell[r] >= 0 && (A.debug_phases & 2)) {
            const int col = tcell[r] / NJ, jj = tcell[r] - col * NJ;
            const int u0 = tblk[r] * 8;
            for (int n = 0; n < nb; n++) {
              const double xy = s_X[n][col] * s_Y[n][jj];
              const double* zr = &s_Z[n][u0];
#pragma unroll
              for (int q = 0; q < 8; q++) acc[r][q] = fma(xy, zr[q], acc[r][q]);
            }
          }
        }
        // ---- window-edge corrections: C_e[u] += sum_n Zi_e[n][u] * sum_cells X[n][col] Y[n][j] R[cell][edge_k[e]] ----------------
        if (edge_mask && (A.debug_phases & 4)) {
#pragma unroll
          for (int e = 0; e < NEDGE; e++) {
            if (!(edge_mask & (1 << e))) continue;
            const bool single = (edge_mask & (edge_mask - 1)) == 0;   // one flagged edge: s_Zi already holds its table
            if (!single) {
              __syncthreads();
              for (int task = tid; task < NU * nb; task += CUR_THREADS) {
                const int b = task / nb, n = task - b * nb;
                const double sn = s_lo + 0.5 * qlen * (1.0 + gx_tab[n0 + n]);
                const double cen = sn * uzr;
                double sumi = 0;
                for (int k = s_ustart[b]; k < s_ustart[b + 1]; k++) {
                  const int sl = s_zord[k];
                  if (s_inval[sl] & (1 << e)) {
                    const double d = s_dz[sl] - cen;
                    sumi += exp_neg(-d * d * i2L);
                  }
                }
                double wn = wscale * gw_tab[n0 + n];
                if (kappa != 0.0) wn *= exp_neg(-sn * sn * kappa);
                s_Zi[n][b] = wn * sumi;
              }
            }
            {
              // every wave takes the nodes n = wv, wv + 4, ..: one response read per cell serves all of them
              double part[QNB / NWAVE];
#pragma unroll
              for (int m = 0; m < QNB / NWAVE; m++) part[m] = 0;
              for (int cl = lane; cl < ncell; cl += 64) {
                const int col = cl / NJ, jj = cl - col * NJ;
                const double r = A.resp[((int64_t)s_coli[col0 + col] * A.nj + (jmin + jj)) * A.nk + edge_k[e]];
#pragma unroll
                for (int m = 0; m < QNB / NWAVE; m++) {
                  const int n = wv + NWAVE * m;
                  if (n < nb) part[m] = fma(s_X[n][col] * s_Y[n][jj], r, part[m]);
                }
              }
#pragma unroll
              for (int m = 0; m < QNB / NWAVE; m++) {
                const double p = wave_add_f64(part[m]);
                if (lane == 0 && wv + NWAVE * m < nb) s_Q[wv + NWAVE * m] = p;
              }
            }
            __syncthreads();
            if (tid < NU) {
              double cv = s_C[e][tid];
              for (int n = 0; n < nb; n++) cv = fma(s_Zi[n][tid], s_Q[n], cv);
              s_C[e][tid] = cv;
            }
          }
        }
      }
      // ---- active bins of every cell, items, pool offsets ---------------------------------------------------------------------------
#pragma unroll
      for (int r = 0; r < 2; r++) {
        if (tcell[r] >= 0) {
          unsigned m = 0;
#pragma unroll
          for (int q = 0; q < 8; q++) {
            const bool in = tblk[r] * 8 + q < NU;
            if (in && (do_prune ? acc[r][q] > thr : acc[r][q] != 0.0)) m |= 1u << q;
          }
          s_tmask[tid + r * CUR_THREADS] = (unsigned char)m;
        }
      }
      __syncthreads();
      if (wv == 0) {
        int nact = 0, nblk_tot = 0, nrun_tot = 0;
        for (int base = 0; base < ncell; base += 64) {
          const int cell = base + lane;
          unsigned long long mk = 0;
          if (cell < ncell)
            for (int b8 = 0; b8 < NB8; b8++) mk |= (unsigned long long)s_tmask[cell * NB8 + b8] << (8 * b8);
          const bool act = mk != 0;
          const int ulo = act ? __ffsll((long long)mk) - 1 : 0, uhi = act ? 63 - __clzll((long long)mk) : 0;
          unsigned long long am = __ballot(act);
          int nb8 = act ? ((uhi - ulo) / 8 + 1) : 0;      // blocks of 8 shifts counted from the first kept shift, not from a multiple of 8
          // one scan for two sums: blocks (low half) and weights inside [ulo, uhi] (high half; statistics: the FMAs that are
          // not padding of an 8-shift block)
          const int scp = wave_scan_i32(nb8 + ((act ? uhi - ulo + 1 : 0) << 16), 0, [](int a, int b) { return a + b; });
          const int sc = scp & 0xFFFF;
          if (cell < ncell) {
            s_culo[cell] = act ? (unsigned char)ulo : (unsigned char)255;
            s_cuhi[cell] = (unsigned char)uhi;
            s_li[cell] = (unsigned short)(nact + __popcll(am & ((1ull << lane) - 1ull)));
            s_boff[cell] = (unsigned short)(nblk_tot + sc - nb8);
          }
          nact += __popcll(am);
          const int tot = __builtin_amdgcn_readlane(scp, 63);
          nblk_tot += tot & 0xFFFF;
          nrun_tot += tot >> 16;
        }
        if (lane == 0) {
          s_misc[21] += nrun_tot;
          s_misc[6] = nact;
          s_misc[7] = nblk_tot;
          int have = s_misc[16];
          unsigned long long need = (unsigned long long)nblk_tot * 8ull;
          unsigned long long base = 0;
          const int item_cap = (A.split_max_items > 0 && A.split_max_items < IMAX) ? A.split_max_items : IMAX;
          bool ok = (have + nact <= item_cap) && !s_misc[18];
          if (ok && need) {
            base = atomicAdd(S.cursor, need);        // (its round trip is not what the kernel waits for: measured, < 1 %)
            if (base + need > S.wbuf_cap) ok = false;
          }
          if (!ok) s_misc[18] = 1;
          s_base64 = base;
        }
      }
      __syncthreads();
      if (!s_misc[18] && s_misc[6] > 0 && (A.debug_phases & 8)) {
        const int have = s_misc[16];
        const unsigned long long base = s_base64;
#pragma unroll
        for (int r = 0; r < 2; r++) {
          const int cell = tcell[r];
          if (cell < 0 || s_culo[cell] == 255) continue;
          const int ulo = s_culo[cell], uhi = s_cuhi[cell];
          const int u0 = tblk[r] * 8;
          if (u0 + 7 < ulo || u0 > uhi) continue;
          // the cell's run starts at its first kept shift (an 8-aligned start padded 30 % of the blocks' FMAs away); the zeros
          // behind the last kept shift fill the last block
          const unsigned long long wo = base + (unsigned long long)s_boff[cell] * 8ull;
          double* dst = S.wbuf + wo;
#pragma unroll
          for (int q = 0; q < 8; q++) {
            const int u = u0 + q;
            if (u >= ulo && u <= uhi) dst[u - ulo] = acc[r][q];
          }
          const int nblk = (uhi - ulo) / 8 + 1;
          if (uhi < u0 + 8) {
            const int run = uhi - ulo + 1;
#pragma unroll
            for (int k = 0; k < 7; k++)
              if (run + k < nblk * 8) dst[run + k] = 0.0;
          }
          if (u0 <= ulo) {
            const int col = cell / NJ, jj = cell - col * NJ;
            Item itx;
            itx.cell_nblk = (s_coli[col0 + col] * A.nj + (jmin + jj)) | (nblk << 16);
            itx.sbase = u_min + ulo;
            itx.woff_lo = (uint32_t)(wo & 0xFFFFFFFFull);
            itx.woff_hi = (uint32_t)(wo >> 32);
            items[have + s_li[cell]] = itx;
          }
        }
      }
      __syncthreads();
      if (tid == 0 && !s_misc[18]) s_misc[16] += s_misc[6];
    }
    // ---- window-edge corrections of this chunk -> (tick, value) list ------------------------------------------------
    __syncthreads();
    if (wv == 0 && edge_mask) {
      for (int e = 0; e < NEDGE; e++) {
        double cv = (lane < NU) ? s_C[e][lane] : 0.0;
        int num = edge_k[e] - (u_min + lane);
        bool ok = (edge_k[e] >= 0) && (lane < NU) && cv != 0.0 && num >= 0 && (num % M) == 0;
        unsigned long long om = __ballot(ok);
        int have = s_misc[17];
        int cnt = __popcll(om);
        if (have + cnt > CMAX) {
          if (lane == 0) s_misc[18] = 1;
        } else if (ok) {
          Corr cr;
          cr.tick = num / M;
          cr.pad = 0;
          cr.val = cv;
          corr[have + __popcll(om & ((1ull << lane) - 1ull))] = cr;
        }
        if (lane == 0 && have + cnt <= CMAX) s_misc[17] = have + cnt;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
    }
    iz_next += n_sl;
  }
  if (tid == 0) stat_add(A.counters, 1, (unsigned long long)NQ);
  __syncthreads();
  if (tid == 0) {
    hdr[0] = s_misc[18] ? 0 : s_misc[16];
    hdr[1] = s_misc[18] ? 0 : s_misc[17];
    hdr[2] = it0;
    hdr[3] = T;
    hdr[4] = it_w0;
    hdr[5] = it_w1;
    hdr[6] = s_misc[18] ? 0 : min(s_misc[19], RUNS_MAX);   // an overflowed pair exposes no runs to mac_kernel
    hdr[7] = s_misc[18];            // 1 = capacity overflow: the monolithic kernel recomputes this pair
    int r = min(s_misc[19], RUNS_MAX);
    hdr[8 + r] = s_misc[16];
    if (s_misc[18]) stat_add(A.counters, 6, 1ull);
    else {
      const int ticks = min(T, it_w1) - max(it0, it_w0);
      if (ticks > 0 && s_misc[21] > 0) stat_add(A.counters, 8, (unsigned long long)s_misc[21] * (unsigned long long)ticks);
    }
  }
}

extern "C++" int qweights_launch(ldsim_ctx* ctx, const SplitArgs& S, int M, void* params) {
  if (S.c.n_pairs == 0) return 0;
  int rc = qpair_setup_launch(ctx, S, M, params, nullptr, nullptr);
  if (rc) return rc;
  const PairParams* pp = (const PairParams*)params;
  if (M == 1)
    hipLaunchKernelGGL(qweights_kernel<1>, dim3((unsigned)S.c.n_pairs), dim3(CUR_THREADS), 0, ctx->stream, S, pp, ctx->d_glx,
                       ctx->d_glw);
  else
    hipLaunchKernelGGL(qweights_kernel<2>, dim3((unsigned)S.c.n_pairs), dim3(CUR_THREADS), 0, ctx->stream, S, pp, ctx->d_glx,
                       ctx->d_glw);
  HIPCHK(hipGetLastError());
  return 0;
}
