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
#include "current_common.h"

template <int M>
__device__ __forceinline__ void current_pair(const CurArgs& A, const int64_t pair) {
  const LdsimConsts* c = A.c;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (pair >= A.n_pairs) return;
  if (A.only_flagged && A.only_flagged[pair * A.flag_stride + 7] == 0) return;

  // ---- which (segment, pixel) -------------------------------------------------------------------
  int64_t seg, pID;
  pair_ids(A, pair, seg, pID);
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
      double iT2 = g.rT / g.sT2, i2T = 1.0 / (2 * g.sT * g.sT);   // _b: sigma*sigma as typed (detsim.py:116)
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
  const double iL2 = g.rL / g.sL2;
  const double a = ux * ux * i2T + uy * uy * i2T + uz * uz * i2L;
  const double factor = g.q / g.Dr / (g.s3 * sqrt(8 * M_PI * M_PI * M_PI));
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
    if (count && fabs(val - kr) > 0.5 - 1e-7) stat_add(A.counters, 0, 1ull);
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
  unsigned long long n_surv = 0;     // samples that reached the transcendental pass (this wave)

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
            // a correction is needed only where the correlation would use this slice's weight at a tick the reference
            // does not: the edge index must be inside the staged response range, reachable by this shift (an integer tick)
            // and that tick inside the stored window; everything else is dropped at the end anyway
            bool need = false;
            const int num = edge_k[e] - sh;
            if (edge_k[e] >= k_stage_lo && edge_k[e] <= k_stage_hi && num >= 0 && (num % M) == 0) {
              const int it_e = num / M;
              if (it_e >= max(it0, it_w0) && it_e < min(T, it_w1)) {
                int64_t kk;
                need = !(slice_valid_at(c, g.t_start, t0, it_e, kk) && kk == edge_k[e]);
              }
            }
            if (need) inval |= 1 << e;
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
      const double cz_ = iL2 * uz;
      const double ddz0 = (g.z_start_int + iz_next * g.z_step) - g.sz;   // ddz of the chunk's first slice
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
              double integral = erf(hi) - erf(lo);          // detsim.py:150-152 (literal form)
              double w = (A.debug_phases & 4) ? pref * integral * exp(E) : 1e-300 * E;
              if (w != 0 && (A.debug_phases & 8)) {
                const int cell = (s_colof[ix] - col0) * NJ + (s_jcell[iy] - jmin);
                const int u = s_shift[sl] - u_min;
                atomicAdd(&s_A[cell * NU8 + u], w);
                const int inval = s_inval[sl];
                if (inval) {
#pragma unroll
                  for (int e = 0; e < NEDGE; e++)
                    if ((inval & (1 << e)) && edge_k[e] >= k_stage_lo && edge_k[e] <= k_stage_hi) {
                      // response at the edge index, straight from L2 (a few hundred distinct addresses per pair)
                      double r = A.resp[((int64_t)s_coli[col0 + cell / NJ] * A.nj + (jmin + cell % NJ)) * A.nk + edge_k[e]];
                      if (r != 0) atomicAdd(&s_C[e][u], w * r);
                    }
                }
              }
            }
            qhead = (qhead + n) & (QLEN - 1);
            qn -= n;
            n_surv += n;
          };
          // Pass A.  For a fixed (ix, iy) the exponent E is a concave quadratic in z, so the slices with E >= cut
          // form one interval: solve it once per (ix, iy) and enqueue only that interval (minus the samples the
          // erf bound removes beyond the segment ends).
          const double qa2 = cz_ * cz_ * inv4a - i2L;                 // <= 0
          const int nxy = g_nix * NS;
          const int nxy_pad = (nxy + 63) & ~63;
          for (int q0 = wv * 64; q0 < nxy_pad; q0 += CUR_THREADS) {
            const int qq = q0 + lane;
            int s_lo = 0, s_hi = -1;
            int ix = 0, iy = 0;
            double Bxy = 0, Dxy = 0;
            if (qq < nxy) {
              const int gi = qq / NS;
              iy = qq - gi * NS;
              ix = s_ixord[g_ix0 + gi];
              if (s_jcell[iy] >= 0) {
                Bxy = s_px[ix][0] + s_py[iy][0];
                Dxy = s_px[ix][1] + s_py[iy][1];
                s_lo = 0;
                s_hi = n_sl - 1;
                if (do_prune && qa2 < -1e-300 && g.z_step > 0) {
                  // E(ddz) = qa2 ddz^2 + qa1 ddz + qa0 >= cut
                  const double qa1 = 2 * Bxy * cz_ * inv4a, qa0 = Bxy * Bxy * inv4a - Dxy - cut;
                  const double disc = qa1 * qa1 - 4 * qa2 * qa0;
                  if (disc < 0) {
                    s_hi = -1;
                  } else {
                    const double sq = sqrt(disc);
                    const double r1 = (-qa1 + sq) / (2 * qa2), r2 = (-qa1 - sq) / (2 * qa2);   // r1 <= r2 (qa2 < 0)
                    // slice sl has ddz = ddz0 + sl*z_step
                    const double f_lo = floor((r1 - ddz0) / g.z_step) - 1, f_hi = ceil((r2 - ddz0) / g.z_step) + 1;
                    if (f_lo > s_lo) s_lo = (int)fmin(f_lo, (double)n_sl);
                    if (f_hi < s_hi) s_hi = (int)fmax(f_hi, -1.0);
                  }
                }
              }
            }
            int len = s_hi - s_lo + 1;
            if (len < 0) len = 0;
            int maxlen = len;
            for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off));
            const unsigned int tagxy = ((unsigned)ix << 12) | ((unsigned)iy << 6);
            for (int t = 0; t < maxlen; t++) {
              bool keep = t < len;
              const int sl = s_lo + t;
              if (do_prune && keep) {
                double b = -(Bxy + s_pz[sl][0]);
                double E2 = b * b * inv4a - (Dxy + s_pz[sl][1]);
                double lo = b * inv_sa2, hi = lo + hi_off;
                if (lo > 0) E2 -= lo * lo;
                else if (hi < 0) E2 -= hi * hi;
                keep = !(E2 < cut);
              }
              unsigned long long m = __ballot(keep);
              if (m) {
                if (keep) q[(qhead + qn + __popcll(m & ((1ull << lane) - 1ull))) & (QLEN - 1)] = tagxy | (unsigned)sl;
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
          // software pipeline: the NEXT cell's row segment is fetched into registers while the current cell is
          // correlated out of LDS (one wave-private row buffer; LDS ops of a wave execute in order)
          constexpr int NLOAD = (M * WTILE + NU_MAX + 8 + 63) / 64;
          double pre[NLOAD];
          auto fetch = [&](int li) {
            const int cell = s_list[li];
            const int ulo8 = s_culo[cell] & ~7;
            const int nblk = (s_cuhi[cell] - ulo8) / 8 + 1;
            const double* rrow = A.resp + ((int64_t)s_coli[col0 + cell / NJ] * A.nj + (jmin + cell % NJ)) * A.nk;
            const int kb = M * tb + u_min + ulo8;         // row element r  <->  response index k = kb + r
            const int nrow = M * WTILE + nblk * 8 + 8;
#pragma unroll
            for (int n = 0; n < NLOAD; n++) {
              const int r = lane + 64 * n, k = kb + r;
              pre[n] = (r < nrow && k >= k_stage_lo && k <= k_stage_hi) ? rrow[k] : 0.0;
            }
          };
          if (share_rank < nact) fetch(share_rank);
          for (int li = share_rank; li < nact; li += nshare) {
            const int cell = s_list[li];
            const int ulo8 = s_culo[cell] & ~7;
            const int nblk = (s_cuhi[cell] - ulo8) / 8 + 1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
            for (int n = 0; n < NLOAD; n++)
              if (lane + 64 * n < ROWLEN) rowp[rpos<M>(lane + 64 * n)] = pre[n];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (li + nshare < nact) fetch(li + nshare);
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
  if (lane == 0 && n_blocks) stat_add(A.counters, 5, n_blocks * 64ull * 64ull);
  if (lane == 0 && n_surv) stat_add(A.counters, 1, n_surv);
}

// one workgroup per pair, or -- the recompute pass of the split paths, where flagged pairs are few or none -- a fixed grid over
// the list of flagged pairs
template <int M>
__global__ void __launch_bounds__(CUR_THREADS, (M == 1 ? 2 : 1)) current_kernel(CurArgs A) {
  if (!A.flag_list) {
    current_pair<M>(A, blockIdx.x);
    return;
  }
  const int64_t cnt = (int64_t)*A.flag_count;
  for (int64_t idx = blockIdx.x; idx < cnt; idx += gridDim.x) {
    current_pair<M>(A, A.flag_list[idx]);
    __syncthreads();
  }
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
  dim3 grid((unsigned)(args.flag_list ? (args.n_pairs < 2048 ? args.n_pairs : 2048) : args.n_pairs)), block(CUR_THREADS);
  if (M == 1)
    hipLaunchKernelGGL(current_kernel<1>, grid, block, 0, ctx->stream, args);
  else
    hipLaunchKernelGGL(current_kernel<2>, grid, block, 0, ctx->stream, args);
  HIPCHK(hipGetLastError());
  return 0;
}
