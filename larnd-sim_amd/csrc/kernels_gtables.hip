// kernels_gtables.hip -- a9-a12, node-separable form ("weights_mode" 2, gform.h): the tables stage.
//
// gtables_kernel: one 256-thread workgroup per (segment, pixel) pair.  It does what qweights_kernel does up to its tables --
// sample -> response-cell maps, response shift and window-edge flags of every z slice (the reference's own expressions,
// detsim.py:414-446, shared with the other weight kernels), one Gaussian per (table bin, quadrature node) -- and stops there:
// per node batch it writes X[n][col], Y[n][j], Z[n][shift] (and one Zi[n][shift] table per window edge some slice is invalid
// at) and the list of response cells whose total weight can exceed the pruning threshold.  The rank-N accumulation into
// A[cell][shift], the weight pool and the item lists of the other split paths do not exist here; gcorr_kernel
// (kernels_gcorr.hip) correlates straight from the tables.
//
// Deterministic: every table entry has one owner thread and a fixed order of its terms.
#include <type_traits>
#include "gform.h"
#include "wave_ops.h"

#define GQN_LDS 64        // nodes of a rule kept in LDS (longer rules read the table)

template <int M>
__global__ void __launch_bounds__(CUR_THREADS, 4) gtables_kernel(GArgs GA, const int32_t* __restrict__ list) {
  const CurArgs& A = GA.c;
  const LdsimConsts* c = A.c;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t pair = list[blockIdx.x];       // the pairs gtables_wave_kernel does not take (it also flags the ones without a record)
  if (pair >= A.n_pairs) return;
  GInfo* __restrict__ gip = GA.gi + pair;
  const int status = (A.debug_phases & 0x100) ? 0 : gip->status;
  if (status != 1) {
    if (tid == 0) {
      GA.flags[pair] = status == 2;
      if (status == 2) stat_add(A.counters, 6, 1ull);
    }
    return;
  }
  const int ncol_g = gip->ncol, NJ_g = gip->NJ, u_min = gip->u_min, NU = gip->NU, ebound = gip->edge_bound, NB = gip->NB;
  double* __restrict__ rec = GA.rec + gip->off;
  const PairParams* __restrict__ P = GA.pp + pair;
  const int NQ = P->NQ, iz_lo = P->iz_lo, iz_hi = P->iz_hi, it0 = P->it0, T = P->T, it_w0 = P->it_w0, it_w1 = P->it_w1;
  const int NUr = g_nur(NU);

  __shared__ double s_par[32];
  __shared__ double s_gl[2][GQN_LDS];
  __shared__ double s_X[G_NODES][G_XS], s_Y[G_NODES][G_YS], s_Z[G_NODES][G_ZS];
  __shared__ double s_dxs[NS_MAX], s_dys[NS_MAX], s_dz[ZC], s_dzs[ZC], s_zs[G_NODES];
  __shared__ unsigned char s_invs[ZC], s_zord[ZC];
  __shared__ int s_inval[ZC];
  __shared__ short s_coli[NS_MAX], s_colstart[NS_MAX + 1], s_jstart[NJ_MAX + 1], s_ustart[NU_MAX + 1];
  __shared__ int s_misc[24], s_wcnt[NWAVE];
  __shared__ unsigned short s_cells[G_CELLCAP];     // col | j << 6 of the listed cells

  if (tid < PP_COUNT) s_par[tid] = ((const double*)((const char*)P + 32))[tid];
  {
    const int64_t off = (int64_t)NQ * (NQ - 1) / 2;
    if (tid < NQ && tid < GQN_LDS) {
      s_gl[0][tid] = GA.glx[off + tid];
      s_gl[1][tid] = GA.glw[off + tid];
    }
  }
  __syncthreads();
  const int NS = c->sampled_points;
  const double bin = c->response_bin_size;
  const int edge_k[NEDGE] = {GA.edge_k[0], GA.edge_k[1], GA.edge_k[2]};      // (edge_ks of the launch's constants, from the host)
  const int k_stage_lo = GA.k_stage_lo, k_stage_hi = GA.k_stage_hi;

  // ---- one chunk's slices (<= 64 slices, <= 64 distinct shifts): response shift, edge flags, member lists ordered by shift.
  // Run by one wave; results in s_dz / s_dzs / s_invs / s_inval / s_zord / s_ustart and s_misc[3..5], [20].
  auto chunk_setup = [&](int iz_first) {
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
#pragma unroll
      for (int e = 0; e < NEDGE; e++) {
        // a table of its own is needed only where the correlation would use this slice's weight at a tick the reference does
        // not: the edge index inside the staged response range, reachable by this shift (an integer tick), that tick inside
        // the stored window
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
    int posn = 0, below = 0;
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
    int anyinv = 0;
#pragma unroll
    for (int e = 0; e < NEDGE; e++)
      if (__ballot(lane < n && (inval & (1 << e)))) anyinv |= 1 << e;
    if (lane == 0) {
      s_misc[3] = n; s_misc[4] = lo; s_misc[5] = hi;
      s_misc[20] = anyinv;
    }
  };

  // ---- sample -> response cell maps; member lists ordered by response index (x: wave 0, y: wave 1); first chunk: wave 2 ---------
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
      ddx = x - s_par[PP_SX];
    }
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
    if (is_leader) s_coli[slot] = (short)i;
    if (valid) s_dxs[posn] = ddx;
    if (is_leader) s_colstart[slot] = (short)posn;
    int nvalid = __popcll(__ballot(valid));
    if (lane == 0) {
      s_colstart[ncol] = (short)nvalid;
      s_misc[0] = ncol;
      s_misc[18] = 0;   // overflow / inconsistency -> monolithic kernel
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
      ddy = y - s_par[PP_SY];
    }
    const int jmin = wave_min_i32((j >= 0) ? j : (1 << 20)), jmax = wave_max_i32(j);
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
    if (j >= 0) s_dys[posn] = ddy;
    if (jmax >= jmin && lane <= jmax - jmin + 1 && lane <= NJ_MAX) s_jstart[lane] = (short)below;   // nj <= NJ_MAX < 64
    if (lane == 0) {
      s_misc[1] = jmin;
      s_misc[2] = jmax;
    }
  } else if (wv == 2) {
    chunk_setup(iz_lo);
  }
  __syncthreads();
  const int ncol = s_misc[0], jmin = s_misc[1], jmax = s_misc[2];
  const int NJ = jmax - jmin + 1;
  const bool one_chunk = s_misc[3] >= iz_hi - iz_lo + 1;          // every slice in the chunk set up above
  const bool chunk_ready = true;                                  // (the set-up above is the state a one-chunk pair keeps)
  // the set-up pass sized the record from the same expressions: anything else is a bug, not a case -- flagged, recomputed by the
  // monolithic kernel and counted
  if (ncol != ncol_g || NJ != NJ_g || jmin != gip->jmin) {
    if (tid == 0) {
      GA.flags[pair] = 1;
      stat_add(A.counters, 6, 1ull);
    }
    return;
  }

  const double* gx_tab = NQ <= GQN_LDS ? (const double*)s_gl[0] : GA.glx + (int64_t)NQ * (NQ - 1) / 2;
  const double* gw_tab = NQ <= GQN_LDS ? (const double*)s_gl[1] : GA.glw + (int64_t)NQ * (NQ - 1) / 2;
  const bool do_prune = A.prune_log > 0;
  // numba_f32 (kernels_qweights.hip): centres scaled by r = sigma^2 / (sigma*sigma)_f32, node factor exp(-s^2 kappa)
  const double uxr = s_par[PP_UXR], uyr = s_par[PP_UYR], uzr = s_par[PP_UZR], i2T = s_par[PP_I2T], i2L = s_par[PP_I2L];
  const double kappa = s_par[PP_KAPPA], s_lo = s_par[PP_S_LO], qlen = s_par[PP_QLEN], wscale = s_par[PP_WSCALE];
  const double thr = s_par[PP_THR];
  const unsigned long long cells_d = g_cells_doubles(ncol, NJ);
  const unsigned long long batch_d = g_batch_doubles(ncol, NJ, NU, ebound);
  int emask_seen = 0;

  if (A.debug_phases & 0x1000000) return;      // timing tools: stop after the maps
  for (int b = 0; b < NB; b++) {
    const int n0 = b * G_NODES, nb = min(G_NODES, NQ - n0);
    double* brec = rec + G_HDR / 2 + (unsigned long long)b * batch_d;
    int32_t* cells = (int32_t*)brec;                         // [0] count, [2..] entries (entry e at cells[G_CELL0 + e])
    const int rows = g_rows(NQ, b);                          // node rows the record keeps of this batch
    double* gX = brec + cells_d;
    double* gY = gX + rows * ncol;
    double* gZ = gY + rows * NJ;
    __syncthreads();          // the previous batch's tables are no longer read
    // ---- X and Y tables of this node batch: one task per (bin, node), node fastest ---------------------------------------------
    {
      const int nX = ncol * nb, nY = NJ * nb;
      for (int task = tid; task < nX + nY && !(A.debug_phases & 0x2000000); task += CUR_THREADS) {
        const bool isx = task < nX;
        const int rel = isx ? task : task - nX;
        const int bb = rel / nb, n = rel - bb * nb;
        const double sn = s_lo + 0.5 * qlen * (1.0 + gx_tab[n0 + n]);
        double sum = 0;
        if (isx) {
          const double cen = sn * uxr;
          for (int k = s_colstart[bb]; k < s_colstart[bb + 1]; k++) {
            const double d = s_dxs[k] - cen;
            sum += exp_neg(-d * d * i2T);
          }
          s_X[n][bb] = sum;
        } else {
          const double cen = sn * uyr;
          for (int k = s_jstart[bb]; k < s_jstart[bb + 1]; k++) {
            const double d = s_dys[k] - cen;
            sum += exp_neg(-d * d * i2T);
          }
          s_Y[n][bb] = sum;
        }
      }
      // node rows past the batch's last node read as zero in the matrix product
      for (int i = tid; i < (G_NODES - nb) * (ncol + NJ); i += CUR_THREADS) {
        const int n = nb + i / (ncol + NJ), bb = i % (ncol + NJ);
        if (bb < ncol) s_X[n][bb] = 0; else s_Y[n][bb - ncol] = 0;
      }
    }
    // ---- Z tables: slices in chunks, every (shift bin, node) entry owned by one thread; shift range in parts of G_NUCAP bins
    // (one part for all but the steepest centimetre-long segments).  e < 0: all slices (and the per-node totals for the cell
    // test); e >= 0: the slices that are invalid at window edge e.
    auto ztable = [&](int e, double* gdst) {      // gdst: Z[16][NUr] (e < 0: all slices) or Zi_e[16][NUr]
      for (int pu0 = 0; pu0 < NU; pu0 += G_NUCAP) {
        const int pw = min(G_NUCAP, NUr - pu0);                  // columns of this part incl. the zero padding to 16
        __syncthreads();
        for (int i = tid; i < G_NODES * pw; i += CUR_THREADS) s_Z[i / pw][i % pw] = 0;
        for (int iz_next = iz_lo; iz_next <= iz_hi;) {
          __syncthreads();
          if (!(one_chunk && chunk_ready)) {
            if (wv == 0) chunk_setup(iz_next);
            __syncthreads();
          }
          const int n_sl = s_misc[3], lo_c = s_misc[4];
          const int NUc = s_misc[5] - lo_c + 1;
          if (e < 0) emask_seen |= s_misc[20];
          if ((e < 0 || (s_misc[20] & (1 << e))) && !(A.debug_phases & 0x4000000)) {
            for (int task = tid; task < NUc * nb; task += CUR_THREADS) {
              const int bb = task / nb, n = task - bb * nb;
              const int ub = lo_c - u_min + bb - pu0;
              if (ub < 0 || ub >= G_NUCAP) continue;
              const double sn = s_lo + 0.5 * qlen * (1.0 + gx_tab[n0 + n]);
              const double cen = sn * uzr;
              double sum = 0;
              for (int k = s_ustart[bb]; k < s_ustart[bb + 1]; k++) {
                if (e < 0 || (s_invs[k] & (1 << e))) {
                  const double d = s_dzs[k] - cen;
                  sum += exp_neg(-d * d * i2L);
                }
              }
              double wn = wscale * gw_tab[n0 + n];
              if (kappa != 0.0) wn *= exp_neg(-sn * sn * kappa);
              s_Z[n][ub] += wn * sum;
            }
          }
          iz_next += n_sl;
        }
        __syncthreads();
        if (!(GA.dbg & 16))
          for (int i = tid; i < rows * pw; i += CUR_THREADS) gdst[(i / pw) * NUr + pu0 + i % pw] = s_Z[i / pw][i % pw];
        if (e < 0 && tid < G_NODES) {
          double t = pu0 ? s_zs[tid] : 0.0;
          for (int u = 0; u < pw; u++) t += s_Z[tid][u];
          s_zs[tid] = t;
        }
      }
    };
    ztable(-1, gZ);
    __syncthreads();
    // ---- tables to the record -----------------------------------------------------------------------------------------------------
    if (!(GA.dbg & 16)) {
      for (int i = tid; i < rows * ncol; i += CUR_THREADS) gX[i] = s_X[i / ncol][i % ncol];
      for (int i = tid; i < rows * NJ; i += CUR_THREADS) gY[i] = s_Y[i / NJ][i % NJ];
    }
    // ---- cells that can carry weight: sum over the nodes and all shifts of X Y Z above the pruning threshold (every
    // (cell, shift) bin the weight kernels would keep lies in such a cell); list in (column, j) order ------------------------------------
    {
      const int ncand = ncol * NJ;
      int base = 0;
      for (int c0 = 0; c0 < ncand && !(A.debug_phases & 0x8000000); c0 += CUR_THREADS) {
        const int cc = c0 + tid;
        bool keep = false;
        int col = 0, jj = 0;
        if (cc < ncand) {
          col = cc / NJ; jj = cc - col * NJ;
          double w = 0;
#pragma unroll 4
          for (int n = 0; n < G_NODES; n++) w = fma(s_X[n][col] * s_Y[n][jj], s_zs[n], w);
          keep = do_prune ? w > thr : w != 0.0;
        }
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) s_wcnt[wv] = __popcll(bal);
        __syncthreads();
        int before = base;
        for (int w = 0; w < wv; w++) before += s_wcnt[w];
        if (keep) {
          const int pos = before + __popcll(bal & ((1ull << lane) - 1ull));
          const unsigned ce = (unsigned)(s_coli[col] * A.nj + (jmin + jj)) | ((unsigned)col << 16) | ((unsigned)jj << 24);
          cells[G_CELL0 + pos] = (int)ce;
          s_cells[pos] = (unsigned short)(col | (jj << 6));
        }
        base += s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
        __syncthreads();
      }
      __syncthreads();           // (s_cells[0] of another wave)
      const int padded = (base + G_CELLPAD - 1) & ~(G_CELLPAD - 1);
      if (tid < padded - base && base > 0) {
        // padding: copies of the first cell with the weightless flag (gcorr_kernel loads that cell's row and multiplies by 0)
        const unsigned c0 = s_cells[0] & 63u, j0 = (s_cells[0] >> 6) & 63u;
        cells[G_CELL0 + base + tid] = (int)(0x80000000u | (unsigned)(s_coli[c0] * A.nj + (jmin + (int)j0)) | (c0 << 16) | (j0 << 24));
      }
      if (tid == 0) { cells[0] = padded; cells[1] = base; }
    }
    // ---- window edges some slice of the pair is invalid at (detsim.py:418-428): the correlation uses every slice's weight at
    // every tick; the Z table over the invalid slices alone (the same table code) lets gcorr_kernel take their share of the one
    // tick an edge maps to back.
    {
      int et = 0;
      for (int e = 0; e < NEDGE; e++) {
        if (!(ebound & (1 << e)) || (GA.dbg & 32)) continue;
        double* gZi = gZ + (unsigned long long)rows * NUr * (unsigned long long)(1 + et);
        et++;
        if (!(emask_seen & (1 << e))) continue;      // (gcorr_kernel skips the table)
        ztable(e, gZi);
      }
    }
  }
  if (tid == 0) {
    gip->emask = emask_seen & ebound;
    GA.flags[pair] = 0;
    if (!(GA.dbg & 1)) stat_add(A.counters, 1, (unsigned long long)NQ);
  }
}


// ---- one wave per pair ---------------------------------------------------------------------------------------------------------
// gtables_wave_kernel: the same tables for the pairs whose slices fit one chunk (<= 64 slices, <= 128 shifts) and whose X and Y
// bins together fit 54 columns (GInfo.wave_ok: all but the longest segments).  A wave owns the pair: no workgroup barrier
// anywhere, 16 pairs in flight per CU instead of 4, and every phase fills its 64 lanes as (16 table bins) x (4 nodes), each lane
// carrying nodes q, 4 + q, 8 + q, 12 + q of its bin -- so a bin's member loop is shared by four nodes, the invalid-slice tables
// of the window edges accumulate in the same pass as Z from the same Gaussians, and no index needs a division.
// Entry for entry the same expressions and term order as gtables_kernel (the cell test sums the Z totals in another order).
// XYS = row stride of the joint X | Y table: 55 (ncol + NJ <= 54, <= 128 shifts, 10 KB of LDS, 16 pairs per CU) for the launch over
// all pairs, 81 (<= 80, <= 256 shifts, 13 KB, 12 per CU, no register spill at 3 waves per SIMD) over the list of the pairs that need
// it (wide diffusion or steep segments: a fifth of the ndlar pairs)

template <int M, int XYS>
__global__ void __launch_bounds__(64, (XYS <= 55 ? 4 : 3)) gtables_wave_kernel(GArgs GA, const int32_t* __restrict__ list, int pair0) {
  const CurArgs& A = GA.c;
  constexpr int NUW = XYS <= 55 ? G_NUCAP : 2 * G_NUCAP;       // shifts of a pair this instantiation takes
  const int lane = threadIdx.x, u16 = lane & 15, q = lane >> 4;
  const int64_t pair = list ? (int64_t)list[blockIdx.x] : (int64_t)blockIdx.x + pair0;      // (pair0: first pair of a range launch)
  if (pair >= A.n_pairs) return;
#ifdef LDSIM_GCORR_DEBUG
  const bool stamps = (GA.dbg & 2048) != 0;      // timing tools: cycle stamps into the statistics stripes 9..15 (chain.hip prints them)
#else
  constexpr bool stamps = false;
#endif
  unsigned long long ts0 = 0, ts_a = 0, ts_c = 0, ts_xy = 0, ts_z = 0, ts_cells = 0, ts_m = 0;
  if (stamps) ts0 = __builtin_amdgcn_s_memtime();
  GInfo* __restrict__ gip = GA.gi + pair;
  // (everything the pair's two records hold is requested before the first branch: one round trip to memory instead of three)
  const GInfo gi0 = *gip;
  const PairParams* __restrict__ P = GA.pp + pair;
  int p_i[4];
#pragma unroll
  for (int k = 0; k < 4; k++) p_i[k] = ((const int*)P)[k];
  const double par_l = lane < PP_COUNT ? ((const double*)((const char*)P + 32))[lane] : 0.0;
  const double e2_l = g_exp2_64[lane];
  const unsigned long long map_l = lane < G_MAPB / 8 ? ((const unsigned long long*)(GA.maps + pair * G_MAPB))[lane] : 0ull;
  const int status = GPHASE(0x100) ? 0 : gi0.status;
  if (status != 1) {           // nothing to compute, or handed to the monolithic kernel
    if (lane == 0 && !list) {
      GA.flags[pair] = status == 2;
      if (status == 2) stat_add(A.counters, 6, 1ull);
    }
    return;
  }
  if (gi0.wave_ok != (XYS <= 55 ? 1 : 2) || (GA.dbg & 64)) return;      // the other instantiation's or gtables_kernel's pair
  const int ncol_g = gi0.ncol, NJ_g = gi0.NJ, u_min = gi0.u_min, NU = gi0.NU, ebound = gi0.edge_bound, NB = gi0.NB;
  double* __restrict__ rec = GA.rec + gi0.off;
  // PairParams: status, NQ, iz_lo, iz_hi (, it0, T, it_w0, it_w1: in the edge flags of the maps already)
  const int NQ = p_i[1], iz_lo = p_i[2], iz_hi = p_i[3];
  const int NUr = g_nur(NU);
  // Gauss-Legendre nodes (lanes 0..15) and weights (16..31) of the first batch: one load per lane, in flight while the maps are
  // unpacked.  [Until round 4 each of a lane's four nodes was a conditional load followed by its own wait: eight trips to L2 per batch.]
  const double* gxw_tab = (lane < 16 ? GA.glx : GA.glw) + (int64_t)NQ * (NQ - 1) / 2;
  double node_l = 0.0;
  if (lane < 32 && u16 < min(G_NODES, NQ)) node_l = gxw_tab[u16];

  __shared__ double s_par[32], s_e2[64];
  __shared__ double s_dxs[NS_MAX], s_dys[NS_MAX], s_dzs[ZC];
  __shared__ double s_XY[G_NODES][XYS], s_zs[G_NODES];
  __shared__ unsigned char s_invs[ZC];
  __shared__ unsigned long long s_map64[G_MAPB / 8];
  __shared__ double s_gxw[2 * G_NODES];      // the batch's node positions | weights on [-1, 1] (0 past the rule's last node)
  __shared__ short s_ustart[NUW + 1];
  // the maps pair_setup_kernel made (gform.h): sample -> position in its column's / row's member list, the lists' starts, the
  // slices' shifts and edge flags.  [Until round 4 this wave derived them itself: two f64 divisions per sample and one ballot per
  // distinct column, row and shift -- 30 % of its cycles.]
  const unsigned char* s_mapb = (const unsigned char*)s_map64;
  const unsigned char* s_coli = s_mapb + G_MAP_COLI;
  const unsigned char* s_colstart = s_mapb + G_MAP_COLSTART;
  const unsigned char* s_jstart = s_mapb + G_MAP_JSTART;

  if (lane < PP_COUNT) s_par[lane] = par_l;
  s_e2[lane] = e2_l;
  if (lane < G_MAPB / 8) s_map64[lane] = map_l;
  if (lane < 2 * G_NODES) s_gxw[lane] = node_l;
  wsync();
  if (stamps) ts_a = __builtin_amdgcn_s_memtime();
  const unsigned long long ts_ld = ts_a - ts0;
  bool bad = false;          // (wave-uniform) inconsistent with the set-up pass: monolithic kernel
  const int ncol = ncol_g, NJ = NJ_g, jmin = gi0.jmin;
  if (lane < G_MAP_NS) {      // (samples past SAMPLED_POINTS carry 0xFF like the ones outside the table)
    const double x = s_par[PP_X_START] + s_par[PP_SGNX] * (lane * s_par[PP_X_STEP] - 4 * s_par[PP_ST]);
    const double y = s_par[PP_Y_START] + s_par[PP_SGNY] * (lane * s_par[PP_Y_STEP] - 4 * s_par[PP_ST]);
    const int px = s_mapb[G_MAP_XPOS + lane], py = s_mapb[G_MAP_YPOS + lane];
    if (px != 0xFF) s_dxs[px] = x - s_par[PP_SX];
    if (py != 0xFF) s_dys[py] = y - s_par[PP_SY];
  }
  if (ncol + NJ > XYS - 1) bad = true;

  // ---- the slices (one chunk): response shift, edge flags, member list ordered by shift --------------------------------------------------
  int NUc, anyinv = 0;
  {
    const int nmax = min(ZC, iz_hi - iz_lo + 1);
    int sh = 0, inval = 0;
    double dzv = 0;
    if (lane < nmax) {
      sh = u_min + (int)s_mapb[G_MAP_ZSH + lane];
      inval = s_mapb[G_MAP_ZINV + lane];
      const double z = s_par[PP_Z_START_INT] + (iz_lo + lane) * s_par[PP_Z_STEP];
      dzv = z - s_par[PP_SZ];
    }
    if (lane == 0 && s_mapb[G_MAP_AMB]) stat_add(A.counters, 0, (unsigned long long)s_mapb[G_MAP_AMB]);
    const int lo = wave_min_i32(lane < nmax ? sh : (1 << 30)), hi = wave_max_i32(lane < nmax ? sh : -(1 << 30));
    NUc = hi - lo + 1;
    if (nmax != iz_hi - iz_lo + 1 || NUc > NUW || NUc != NU || lo != u_min) bad = true;
    if (!bad) {
      const bool in = lane < nmax;
      const unsigned long long lane_lt = (1ull << lane) - 1ull, lane_le = lane_lt | (1ull << lane);
      // the slices follow the drift axis, so their shifts rise or fall with the lane: runs of equal shift are contiguous and the
      // ordered position follows from the run boundaries; any other order (none known) takes one ballot per distinct shift
      const int prev = __builtin_amdgcn_update_dpp(sh, sh, 0x138, 0xF, 0xF, false);        // wave_shr:1 (lane 0 keeps its own)
      const bool up = __ballot(in && lane > 0 && sh > prev) != 0, down = __ballot(in && lane > 0 && sh < prev) != 0;
      int posn = 0;
      if (!(up && down)) {
        const unsigned long long RS = __ballot(in && (lane == 0 || sh != prev));                // run starts
        const int run_first = 63 - __clzll((long long)(RS & lane_le));
        const unsigned long long above = RS & ~lane_le;
        const int run_end = above ? __ffsll((long long)above) - 1 : nmax;
        posn = down ? (nmax - run_end) + (lane - run_first) : lane;
        // bin starts: a run's first position at its shift, empty bins take the next start (a suffix minimum, 64 bins at a time)
        for (int k = lane; k <= NUc; k += 64) s_ustart[k] = -1;
        wsync();
        if (in && lane == run_first) s_ustart[sh - lo] = (short)posn;
        wsync();
        int carry = nmax;
        for (int part = 0; part < NUc; part += 64) {
          const int bi = NUc - 1 - part - lane;
          int v = 1 << 20;
          if (bi >= 0) {
            const int e = s_ustart[bi];
            v = e < 0 ? (1 << 20) : e;
          }
          v = min(wave_scan_i32(v, 1 << 20, [](int a, int b) { return a < b ? a : b; }), carry);
          if (bi >= 0) s_ustart[bi] = (short)v;
          carry = wave_lane_i32(v, 63);
        }
        if (lane == 0) s_ustart[NUc] = (short)nmax;
      } else {
        int below[NUW / 64];
#pragma unroll
        for (int k = 0; k < NUW / 64; k++) below[k] = 0;
        for (int bs = lo; bs <= hi; bs++) {                    // the distinct shifts, one ballot each
          const unsigned long long bal = __ballot(in && sh == bs);
          const int cnt = __popcll(bal);
          if (in && sh > bs) posn += cnt;
          if (in && sh == bs) posn += __popcll(bal & lane_lt);
#pragma unroll
          for (int k = 0; k < NUW / 64; k++)
            if (bs < lo + lane + 64 * k) below[k] += cnt;
        }
#pragma unroll
        for (int k = 0; k < NUW / 64; k++)
          if (lane + 64 * k < NUc) s_ustart[lane + 64 * k] = (short)below[k];
        if (lane == 0) s_ustart[NUc] = (short)nmax;
      }
      if (in) {
        s_dzs[posn] = dzv;
        s_invs[posn] = (unsigned char)inval;
      }
#pragma unroll
      for (int e = 0; e < NEDGE; e++)
        if (__ballot(lane < nmax && (inval & (1 << e)))) anyinv |= 1 << e;
    }
  }
  if (bad) {
    if (lane == 0) {
      GA.flags[pair] = 1;
      stat_add(A.counters, 6, 1ull);
    }
    return;
  }
  wsync();
  if (stamps) ts_c = __builtin_amdgcn_s_memtime();
  if (GPHASE(0x1000000)) return;      // timing tools: stop after the maps

  const bool do_prune = A.prune_log > 0;
  // (wave-uniform values read from LDS land in vector registers: moved to scalar ones, 20 VGPRs less)
  auto uni = [](double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
  };
  const double uxr = uni(s_par[PP_UXR]), uyr = uni(s_par[PP_UYR]), uzr = uni(s_par[PP_UZR]), i2T = uni(s_par[PP_I2T]);
  const double i2L = uni(s_par[PP_I2L]), kappa = uni(s_par[PP_KAPPA]), s_lo = uni(s_par[PP_S_LO]), qlen = uni(s_par[PP_QLEN]);
  const double wscale = uni(s_par[PP_WSCALE]), thr = uni(s_par[PP_THR]);
  const unsigned long long cells_d = g_cells_doubles(ncol, NJ);
  const unsigned long long batch_d = g_batch_doubles(ncol, NJ, NU, ebound);
  const int emask = anyinv & ebound;
  const int nbins = ncol + NJ;
  const unsigned nj_inv = (1u << 20) / (unsigned)NJ + 1u;      // cc / NJ = cc * nj_inv >> 20 for cc < 2048, NJ <= 48

  for (int b = 0; b < NB; b++) {
    unsigned long long ts_pro = 0;
    if (stamps) ts_pro = __builtin_amdgcn_s_memtime();
    const int n0 = b * G_NODES, nb = min(G_NODES, NQ - n0);
    double* brec = rec + G_HDR / 2 + (unsigned long long)b * batch_d;
    int32_t* cells = (int32_t*)brec;                         // [0] count, [2..] entries (entry e at cells[G_CELL0 + e])
    const int rows = g_rows(NQ, b);                          // node rows the record keeps of this batch
    double* gX = brec + cells_d;
    double* gY = gX + rows * ncol;
    double* gZ = gY + rows * NJ;
    // this lane's four nodes: position along the segment and weight (a node past the batch's last one: rows of zeros)
    if (b > 0) {           // (batch 0's were requested before the maps)
      wsync();
      if (lane < 2 * G_NODES) s_gxw[lane] = u16 < nb ? gxw_tab[n0 + u16] : 0.0;
      wsync();
    }
    double sn[4], wn[4];
#pragma unroll
    for (int m = 0; m < 4; m++) {
      const int n = 4 * m + q;
      const double gx = s_gxw[n], gw = s_gxw[G_NODES + n];
      sn[m] = s_lo + 0.5 * qlen * (1.0 + gx);
      double w = wscale * gw;
      if (kappa != 0.0) w *= exp_neg_tab(-sn[m] * sn[m] * kappa, s_e2);
      wn[m] = n < nb ? w : 0.0;
    }
    wsync();          // the previous batch's tables are no longer read
    // The two table phases, compiled for MC = 1 .. 4 node groups (a batch's last nodes fill fewer than four: 59 % of the survey
    // workload's pairs have a second batch of a few nodes): the MC Gaussians of a member are independent and interleave.
    double zsp[4] = {0, 0, 0, 0};
    auto tables = [&](auto mc_tag) {
      constexpr int MC = decltype(mc_tag)::value;
      // ---- X and Y tables: bins 0 .. ncol - 1 are the columns, ncol .. ncol + NJ - 1 the rows j ----------------------------------------------
      if (stamps) { ts_m = __builtin_amdgcn_s_memtime(); ts_a += ts_m - ts_pro; }
      for (int b0 = 0; b0 < nbins && !GPHASE(0x2000000); b0 += 16) {
        const int bi = b0 + u16;
        const bool act = bi < nbins, isx = bi < ncol;
        const int bb = isx ? bi : bi - ncol;
        int k = 0, ke = 0;
        if (act) {
          k = isx ? s_colstart[bb] : s_jstart[bb];
          ke = isx ? s_colstart[bb + 1] : s_jstart[bb + 1];
        }
        const double* src = isx ? s_dxs : s_dys;
        const double ur = isx ? uxr : uyr;
        double cen[4], sum[4] = {0, 0, 0, 0};
  #pragma unroll
        for (int m = 0; m < 4; m++) cen[m] = sn[m] * ur;
        double dd_next = k < ke ? src[k] : 0.0;          // (the next member's offset is read while this one's Gaussians run)
        for (; __ballot(k < ke); k++) {
          const double dd = dd_next;
          const bool on = k < ke;
          dd_next = k + 1 < ke ? src[k + 1] : 0.0;
          if (on) {
            double xg[4], g[4];
  #pragma unroll
            for (int m = 0; m < 4; m++) {
              const double d = dd - cen[m];
              xg[m] = -d * d * i2T;
            }
            exp_neg_tab_n<MC>(xg, g, s_e2);
  #pragma unroll
            for (int m = 0; m < 4; m++)
              if (m < MC) sum[m] += g[m];
          }
        }
        if (act) {
  #pragma unroll
          for (int m = 0; m < 4; m++) {
            const int n = 4 * m + q;
            const double v = n < nb ? sum[m] : 0.0;
            s_XY[n][bi] = v;
            if (!GDBG(16) && n < rows) {
              if (isx) gX[n * ncol + bb] = v; else gY[n * NJ + bb] = v;
            }
          }
        }
      }
      // ---- Z and, from the same Gaussians, the tables over the slices that are invalid at a window edge -----------------------------------------
      if (stamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); ts_xy += t - ts_m; ts_m = t; }
      {
        double cen[4];
  #pragma unroll
        for (int m = 0; m < 4; m++) cen[m] = sn[m] * uzr;
        // the (up to) two edges carried with Z, and where their tables go (an edge of the bound without invalid slices has a
        // table nobody reads)
        const int e0 = emask ? __ffs(emask) - 1 : 0, m1 = emask & (emask - 1), e1 = m1 ? __ffs(m1) - 1 : -1;
        const int m2 = m1 & (m1 - 1), e2 = m2 ? __ffs(m2) - 1 : -1;
        double* gZi0 = gZ + (unsigned long long)rows * NUr * (unsigned long long)(1 + g_popc3(ebound & ((1 << e0) - 1)));
        double* gZi1 = gZ + (unsigned long long)rows * NUr * (unsigned long long)(1 + (e1 >= 0 ? g_popc3(ebound & ((1 << e1) - 1)) : 0));
        for (int u0 = 0; u0 < NUr; u0 += 16) {
          const int ub = u0 + u16;
          int k = 0, ke = 0;
          if (ub < NUc) { k = s_ustart[ub]; ke = s_ustart[ub + 1]; }
          double z[4] = {0, 0, 0, 0}, zi[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
          double dd_next = k < ke ? s_dzs[k] : 0.0;
          if (!emask) {                  // (most pairs: no slice is invalid at a window edge, no edge tables)
            for (; __ballot(k < ke) && !GPHASE(0x4000000); k++) {
              const double dd = dd_next;
              const bool on = k < ke;
              dd_next = k + 1 < ke ? s_dzs[k + 1] : 0.0;
              if (on) {
                double xg[4], g[4];
  #pragma unroll
                for (int m = 0; m < 4; m++) {
                  const double d = dd - cen[m];
                  xg[m] = -d * d * i2L;
                }
                exp_neg_tab_n<MC>(xg, g, s_e2);
  #pragma unroll
                for (int m = 0; m < 4; m++)
                  if (m < MC) z[m] += g[m];
              }
            }
          } else {
            for (; __ballot(k < ke) && !GPHASE(0x4000000); k++) {
              const double dd = dd_next;
              const bool on = k < ke;
              dd_next = k + 1 < ke ? s_dzs[k + 1] : 0.0;
              if (on) {
                const int inv = s_invs[k];
                const bool i0 = (inv >> e0) & 1, i1 = e1 >= 0 && ((inv >> e1) & 1);
                double xg[4], g[4];
  #pragma unroll
                for (int m = 0; m < 4; m++) {
                  const double d = dd - cen[m];
                  xg[m] = -d * d * i2L;
                }
                exp_neg_tab_n<MC>(xg, g, s_e2);
  #pragma unroll
                for (int m = 0; m < 4; m++) {
                  if (m < MC) {
                    z[m] += g[m];
                    zi[0][m] += i0 ? g[m] : 0.0;
                    zi[1][m] += i1 ? g[m] : 0.0;
                  }
                }
              }
            }
          }
  #pragma unroll
          for (int m = 0; m < 4; m++) {
            const int n = 4 * m + q;
            const double v = 0.0 + wn[m] * z[m];
            zsp[m] += v;
            if (!GDBG(16) && n < rows) gZ[n * NUr + ub] = v;
          }
          if (!GDBG(48)) {
  #pragma unroll
            for (int m = 0; m < 4; m++) {
              if (emask && 4 * m < rows) gZi0[(4 * m + q) * NUr + ub] = 0.0 + wn[m] * zi[0][m];
              if (e1 >= 0 && 4 * m < rows) gZi1[(4 * m + q) * NUr + ub] = 0.0 + wn[m] * zi[1][m];
            }
          }
        }
        // a third edge (a response staged down to index 0 and up to the window's end): its table in a pass of its own
        if (e2 >= 0) {
          double* gZi2 = gZ + (unsigned long long)rows * NUr * (unsigned long long)(1 + g_popc3(ebound & ((1 << e2) - 1)));
          for (int u0 = 0; u0 < NUr; u0 += 16) {
            const int ub = u0 + u16;
            int k = 0, ke = 0;
            if (ub < NUc) { k = s_ustart[ub]; ke = s_ustart[ub + 1]; }
            double zi2[4] = {0, 0, 0, 0};
            for (; __ballot(k < ke); k++) {
              if (k < ke && ((s_invs[k] >> e2) & 1)) {
                const double dd = s_dzs[k];
  #pragma unroll
                for (int m = 0; m < 4; m++) {
                  if (m < MC) {
                    const double d = dd - cen[m];
                    zi2[m] += exp_neg_tab(-d * d * i2L, s_e2);
                  }
                }
              }
            }
  #pragma unroll
            for (int m = 0; m < 4; m++)
              if (4 * m < rows) gZi2[(4 * m + q) * NUr + ub] = 0.0 + wn[m] * zi2[m];
          }
        }
      }
    };
    switch ((nb + 3) >> 2) {
      case 1: tables(std::integral_constant<int, 1>{}); break;
      case 2: tables(std::integral_constant<int, 2>{}); break;
      case 3: tables(std::integral_constant<int, 3>{}); break;
      default: tables(std::integral_constant<int, 4>{}); break;
    }
    if (stamps) { const unsigned long long t = __builtin_amdgcn_s_memtime(); ts_z += t - ts_m; ts_m = t; }
    // per-node totals over the shifts (cell test): row sums over the 16 lanes of a node group
#pragma unroll
    for (int m = 0; m < 4; m++) {
      double v = zsp[m];
      v += wdpp_f64<0x111, 0xF>(0.0, v);
      v += wdpp_f64<0x112, 0xF>(0.0, v);
      v += wdpp_f64<0x114, 0xF>(0.0, v);
      v += wdpp_f64<0x118, 0xF>(0.0, v);
      if (u16 == 15) s_zs[4 * m + q] = v;
    }
    wsync();
    // ---- cells that can carry weight (as gtables_kernel), in (column, j) order ----------------------------------------------------------------
    {
      const int ncand = ncol * NJ;
      int base = 0;
      unsigned first_code = 0;
      for (int c0 = 0; c0 < ncand && !GPHASE(0x8000000); c0 += 64) {
        const int cc = c0 + lane;
        bool keep = false;
        int col = 0, jj = 0;
        if (cc < ncand) {
          col = (int)(((unsigned)cc * nj_inv) >> 20); jj = cc - col * NJ;
          // (rows past nb are zero: four nodes at a time, their twelve LDS reads in flight together -- one round trip per
          // group instead of one per node; the sum takes the nodes in the same order)
          double w = 0;
          for (int n = 0; n < ((nb + 3) & ~3); n += 4) {
            const double x0 = s_XY[n][col], x1 = s_XY[n + 1][col], x2 = s_XY[n + 2][col], x3 = s_XY[n + 3][col];
            const double y0 = s_XY[n][ncol + jj], y1 = s_XY[n + 1][ncol + jj], y2 = s_XY[n + 2][ncol + jj], y3 = s_XY[n + 3][ncol + jj];
            const double z0 = s_zs[n], z1 = s_zs[n + 1], z2 = s_zs[n + 2], z3 = s_zs[n + 3];
            w = fma(x0 * y0, z0, w);
            w = fma(x1 * y1, z1, w);
            w = fma(x2 * y2, z2, w);
            w = fma(x3 * y3, z3, w);
          }
          keep = do_prune ? w > thr : w != 0.0;
        }
        const unsigned long long bal = __ballot(keep);
        const unsigned ce = (unsigned)(s_coli[col] * A.nj + (jmin + jj)) | ((unsigned)col << 16) | ((unsigned)jj << 24);
        if (keep) cells[G_CELL0 + base + __popcll(bal & ((1ull << lane) - 1ull))] = (int)ce;
        if (base == 0 && bal) first_code = (unsigned)__builtin_amdgcn_readlane((int)ce, __ffsll((long long)bal) - 1);
        base += __popcll(bal);
      }
      const int padded = (base + G_CELLPAD - 1) & ~(G_CELLPAD - 1);
      // padding: copies of the first cell with the weightless flag (gcorr_kernel loads that cell's row and multiplies by 0)
      if (lane < padded - base && base > 0) cells[G_CELL0 + base + lane] = (int)(0x80000000u | first_code);
      if (lane == 0) { cells[0] = padded; cells[1] = base; }
    }
    if (stamps) ts_cells += __builtin_amdgcn_s_memtime() - ts_m;
  }
  if (stamps && lane == 0) {
    stat_add(A.counters, 9, ts_ld);
    stat_add(A.counters, 10, ts_c - (ts0 + ts_ld));
    stat_add(A.counters, 11, ts_a - (ts0 + ts_ld));
    stat_add(A.counters, 12, ts_xy);
    stat_add(A.counters, 13, ts_z);
    stat_add(A.counters, 14, ts_cells);
    stat_add(A.counters, 15, __builtin_amdgcn_s_memtime() - ts0);
  }
  if (lane == 0) {
    gip->emask = emask;
    GA.flags[pair] = 0;
    if (!GDBG(1)) stat_add(A.counters, 1, (unsigned long long)NQ);
  }
}

// the pairs gtables_wave_kernel's first launch leaves to its wide instantiation (wave_ok 2) and to gtables_kernel (0); one atomic
// per wave and list
__global__ void __launch_bounds__(256) gtables_list_kernel(const GInfo* __restrict__ gi, int64_t n, int all, int32_t* __restrict__ wg_list,
                                                          unsigned long long* __restrict__ wg_count, int32_t* __restrict__ w2_list,
                                                          unsigned long long* __restrict__ w2_count) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  int cls = 0;                 // 1: workgroup kernel, 2: wide wave kernel
  if (i < n && gi[i].status == 1) cls = (all || gi[i].wave_ok == 0) ? 1 : (gi[i].wave_ok == 2 ? 2 : 0);
  for (int c = 1; c <= 2; c++) {
    const unsigned long long m = __ballot(cls == c);
    if (!m) continue;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(c == 1 ? wg_count : w2_count, (unsigned long long)__popcll(m));
    base = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((int)base);
    if (cls == c) (c == 1 ? wg_list : w2_list)[(int64_t)base + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)i;
  }
}

extern "C++" int gtables_list_launch(ldsim_ctx* ctx, const GArgs& GA, int32_t* wg_list, unsigned long long* wg_count, int32_t* w2_list,
                                     unsigned long long* w2_count) {
  const int64_t n = GA.c.n_pairs;
  if (n == 0) return 0;
  hipLaunchKernelGGL(gtables_list_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, GA.gi, n, (GA.dbg & 64) ? 1 : 0,
                     wg_list, wg_count, w2_list, w2_count);
  HIPCHK(hipGetLastError());
  return 0;
}

// the wave kernel over the pair range [pair0, pair0 + n) on stream `ts`
extern "C++" int gtables_launch_range(ldsim_ctx* ctx, const GArgs& GA, int M, hipStream_t ts, int64_t pair0, int64_t n) {
  if (n <= 0) return 0;
  if (M == 1) hipLaunchKernelGGL((gtables_wave_kernel<1, 55>), dim3((unsigned)n), dim3(64), 0, ts, GA, (const int32_t*)nullptr, (int)pair0);
  else hipLaunchKernelGGL((gtables_wave_kernel<2, 55>), dim3((unsigned)n), dim3(64), 0, ts, GA, (const int32_t*)nullptr, (int)pair0);
  HIPCHK(hipGetLastError());
  return 0;
}
// its wide instantiation over the `n_w2` pairs of `w2_list`, and the workgroup kernel over the `n_wg` listed ones, on stream `ts`
extern "C++" int gtables_launch_lists(ldsim_ctx* ctx, const GArgs& GA, int M, hipStream_t ts, const int32_t* wg_list, int64_t n_wg,
                                      const int32_t* w2_list, int64_t n_w2) {
  if (n_w2 > 0) {
    if (M == 1) hipLaunchKernelGGL((gtables_wave_kernel<1, 81>), dim3((unsigned)n_w2), dim3(64), 0, ts, GA, w2_list, 0);
    else hipLaunchKernelGGL((gtables_wave_kernel<2, 81>), dim3((unsigned)n_w2), dim3(64), 0, ts, GA, w2_list, 0);
    HIPCHK(hipGetLastError());
  }
  if (n_wg > 0) {
    if (M == 1) hipLaunchKernelGGL(gtables_kernel<1>, dim3((unsigned)n_wg), dim3(CUR_THREADS), 0, ts, GA, wg_list);
    else hipLaunchKernelGGL(gtables_kernel<2>, dim3((unsigned)n_wg), dim3(CUR_THREADS), 0, ts, GA, wg_list);
    HIPCHK(hipGetLastError());
  }
  return 0;
}

// wave kernel over all pairs, its wide instantiation over the `n_w2` pairs of `w2_list`, then the workgroup kernel over the `n_wg`
// listed ones
extern "C++" int gtables_launch(ldsim_ctx* ctx, const GArgs& GA, int M, const int32_t* wg_list, int64_t n_wg, const int32_t* w2_list,
                                int64_t n_w2) {
  if (GA.c.n_pairs == 0) return 0;
  int rc = gtables_launch_range(ctx, GA, M, ctx->stream, 0, GA.c.n_pairs);
  if (rc) return rc;
  return gtables_launch_lists(ctx, GA, M, ctx->stream, wg_list, n_wg, w2_list, n_w2);
}
