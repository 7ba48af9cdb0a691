// kernels_split.hip -- a9-a12 as two kernels (the default path; option "split_kernels", 0 = monolithic kernels_current.hip):
//
//   weights_kernel<M>   one workgroup per (segment, pixel) pair: evaluates every charge sample once and emits the
//                       pair's binned weights  A[cell][shift]  as compact "items" (cell id, first shift, #8-shift
//                       blocks, offset into a shared f64 pool in HBM) plus the window-edge corrections.  It touches no
//                       response rows (the rare edge path reads single table entries): 128 VGPRs (+52 B/lane of
//                       scratch, outer code only), 37.7 KB of LDS -> 4 workgroups per CU.
//   mac_kernel<M>       one workgroup per pair: walks the pair's items; per item the response row segment and the
//                       item's weights are prefetched into registers one item ahead, staged in wave-private LDS and
//                       correlated with the register-tiled sliding window of kernels_current.hip.  M = 1: item list
//                       copied to LDS, 128 VGPRs, 31.8 KB -> 4 workgroups per CU; M = 2: descriptors read from HBM two
//                       items ahead, 168 VGPRs, 39.8 KB -> 3 workgroups per CU.
//
// Pairs that exceed a per-pair capacity (items, corrections, runs) are flagged and recomputed by the monolithic
// current_kernel (kernels_current.hip), which has no such limits.  The weight pool is sized by chain.hip from the
// demand of earlier launches; a launch that exhausted it is repeated, never used.
#include "split_common.h"

// =============================================================================================================
template <int M>
__global__ void __launch_bounds__(CUR_THREADS, 4) weights_kernel(SplitArgs S) {
  const CurArgs& A = S.c;
  const LdsimConsts* c = A.c;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t pair = blockIdx.x;
  if (pair >= A.n_pairs) return;
  int32_t* hdr = S.hdr + pair * HDR_INTS;

  int64_t seg, pID;
  pair_ids(A, pair, seg, pID);
  int T = A.T;
  if (A.tmax_batch) T = min(T, A.tmax_batch[A.s.batch[seg] - A.batch0]);

  PairGeo g;
  pair_geometry(A, seg, pID, g);
  auto write_empty = [&]() {
    if (tid < HDR_INTS) hdr[tid] = 0;
  };
  if (!g.ok) { write_empty(); return; }
  const int NS = c->sampled_points;
  const double dt = c->time_sampling, dtr = c->response_sampling, TW = c->time_window;
  const double bin = c->response_bin_size;

  __shared__ double s_A[W_ARENA];
  __shared__ double s_C[NEDGE][NU_MAX];
  __shared__ double s_px[NS_MAX][2], s_py[NS_MAX][2], s_pz[ZC][2];
  __shared__ int s_shift[ZC], s_inval[ZC];
  __shared__ short s_icell[NS_MAX], s_jcell[NS_MAX], s_colof[NS_MAX], s_coli[NS_MAX], s_colstart[NS_MAX + 1];
  __shared__ unsigned char s_ixord[NS_MAX];
  __shared__ unsigned short s_list[W_CELLS];
  __shared__ unsigned char s_culo[W_CELLS], s_cuhi[W_CELLS];
  __shared__ unsigned short s_boff[W_CELLS];    // first 8-shift block of every listed cell inside the group's allocation
  __shared__ unsigned int s_q[NWAVE][QLEN];
  __shared__ unsigned int s_qt[NWAVE][QLEN];    // tail class (f32 evaluation)
  __shared__ int s_misc[24];
  __shared__ unsigned long long s_base64;


  // ---- sample -> response cell maps; column slots ordered by response index i ---------------------------------
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
      double ddx = x - g.sx, ddy = y - g.sy;
      double iT2 = g.rT / g.sT2, i2T = 1.0 / (2 * g.sT * g.sT);   // _b: sigma*sigma as typed (detsim.py:116)
      s_px[lane][0] = ddx * iT2 * (g.Dx / g.Dr);
      s_px[lane][1] = ddx * ddx * i2T;
      s_py[lane][0] = ddy * iT2 * (g.Dy / g.Dr);
      s_py[lane][1] = ddy * ddy * i2T;
    }
    int leader = lane;
    for (int q = 0; q < NS; q++) {
      int iq = __shfl(i, q);
      if (q < leader && iq == i) leader = q;
    }
    bool is_leader = (lane < NS) && (i >= 0) && (leader == lane);
    // slot = rank of this column's i among the distinct i  -> cells come out sorted by (i, j)
    int slot = 0;
    for (int q = 0; q < NS; q++) {
      int iq = __shfl(i, q);
      bool lq = __shfl((int)is_leader, q);
      if (lq && iq < i) slot++;
    }
    int myslot = (i < 0 || lane >= NS) ? -1 : slot;
    int ncol = __popcll(__ballot(is_leader));
    if (lane < NS) s_colof[lane] = (short)myslot;
    if (is_leader) s_coli[slot] = (short)i;
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
      s_misc[16] = 0;   // items emitted
      s_misc[17] = 0;   // corrections emitted
      s_misc[18] = 0;   // overflow
      s_misc[19] = 0;   // runs
    }
  }
  __syncthreads();
  const int ncol = s_misc[0], jmin = s_misc[1], jmax = s_misc[2];
  const int NJ = jmax - jmin + 1;
  if (ncol == 0 || NJ <= 0) { write_empty(); return; }

  const double V = TW / dtr;
  int edge_k[NEDGE] = {0, -1, -1};
  int k_top;
  {
    int ka = (int)floor(V - 0.5 - 1e-6);
    if ((double)ka + 0.5 >= V - 1e-6) ka--;
    int kn = (int)ceil(V + 0.5 + 1e-6);
    k_top = kn - 1;
    int ne = 1;
    for (int k = ka + 1; k <= k_top && ne < NEDGE; k++) edge_k[ne++] = k;
    if (k_top - ka > NEDGE - 1) k_top = ka + NEDGE - 1;
  }
  const int k_stage_hi = min(min(k_top, A.nk - 1), A.k_last);
  const int k_stage_lo = max(0, A.k_first);

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
  // samples bounded by exp(-tail_log) of the segment's peak density are evaluated in f32 (relative error ~3e-7 of
  // a term that is itself < exp(-tail_log) of the peak); 0 = every sample in f64
  const bool do_tail = A.tail_log > 0;
  const double tcut = -A.tail_log;

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
    int lo = (k_stage_lo - sh_max) / M - 1, hi = (k_stage_hi - sh_min) / M + 2;
    it_w0 = max(it_w0, lo);
    it_w1 = min(it_w1, hi);
  }
  if (sh_min > sh_max || it_w1 <= it_w0 || k_stage_hi < k_stage_lo) { write_empty(); return; }

  constexpr int IMAX = ItemCap<M>::value;
  Item* items = S.items + pair * IMAX;
  Corr* corr = S.corr + pair * CMAX;

  unsigned long long n_surv = 0;
  int iz_next = iz_lo;
  while (iz_next <= iz_hi) {
    __syncthreads();
    if (wv == 0) {
      int nmax = min(ZC, iz_hi - iz_next + 1);
      int sh = 0, inval = 0;
      if (lane < nmax) {
        int iz = iz_next + lane;
        double z, t0;
        sh = slice_shift(iz, z, t0, true);
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
      int pmin = lane < nmax ? sh : (1 << 30), pmax = lane < nmax ? sh : -(1 << 30);
      for (int off = 1; off < 64; off <<= 1) {
        int a1 = __shfl_up(pmin, off), a2 = __shfl_up(pmax, off);
        if (lane >= off) { pmin = min(pmin, a1); pmax = max(pmax, a2); }
      }
      bool fits = (lane < nmax) && (pmax - pmin + 1 <= NU_MAX);
      unsigned long long fm = __ballot(fits);
      int n = (fm == ~0ull) ? 64 : __ffsll((long long)~fm) - 1;
      int lo = __shfl(pmin, n - 1), hi = __shfl(pmax, n - 1);
      if (lane == 0) {
        s_misc[3] = n; s_misc[4] = lo; s_misc[5] = hi;
        int r = s_misc[19];
        if (r < RUNS_MAX) hdr[8 + r] = s_misc[16]; else s_misc[18] = 1;
        s_misc[19] = r + 1;
      }
    }
    __syncthreads();
    const int n_sl = s_misc[3], u_min = s_misc[4];
    const int NU = s_misc[5] - u_min + 1;
    const int NU8 = (NU + 7) & ~7;
    for (int i = tid; i < NEDGE * NU_MAX; i += CUR_THREADS) (&s_C[0][0])[i] = 0;
    const int cols_per_group = max(1, min(W_ARENA / (NJ * NU8), W_CELLS / NJ));

    for (int col0 = 0; col0 < ncol; col0 += cols_per_group) {
      const int gcols = min(cols_per_group, ncol - col0);
      const int ncell = gcols * NJ;
      const int g_ix0 = s_colstart[col0], g_nix = s_colstart[col0 + gcols] - g_ix0;
      __syncthreads();
      for (int i = tid; i < ncell * NU8; i += CUR_THREADS) s_A[i] = 0;
      __syncthreads();
      {
        unsigned int* q = s_q[wv];
        unsigned int* qt = s_qt[wv];
        int qhead = 0, qn = 0, qthead = 0, qtn = 0;
        auto deposit = [&](int ix, int iy, int sl, double w) {
          const int cell = (s_colof[ix] - col0) * NJ + (s_jcell[iy] - jmin);
          const int u = s_shift[sl] - u_min;
          atomicAdd(&s_A[cell * NU8 + u], w);
          const int inval = s_inval[sl];
          if (inval) {
#pragma unroll
            for (int e = 0; e < NEDGE; e++)
              if (inval & (1 << e)) {
                // only slices flagged by the predicate above get here: edge_k[e] is inside the staged range
                double r = A.resp[((int64_t)s_coli[col0 + cell / NJ] * A.nj + (jmin + cell % NJ)) * A.nk + edge_k[e]];
                if (r != 0) atomicAdd(&s_C[e][u], w * r);
              }
          }
        };
        auto process = [&](int n) {
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          if (lane < n && (A.debug_phases & 1)) {
            unsigned int en = q[(qhead + lane) & (QLEN - 1)];
            const int ix = en >> 12, iy = (en >> 6) & 63, sl = en & 63;
            double b = -(s_px[ix][0] + s_py[iy][0] + s_pz[sl][0]);
            double delta = s_px[ix][1] + s_py[iy][1] + s_pz[sl][1];
            double E = b * b * inv4a - delta;
            double lo = b * inv_sa2, hi = lo + hi_off;
            double w = (A.debug_phases & 4) ? pref * (erf(hi) - erf(lo)) * exp(E)          // detsim.py:150-157 (literal form)
                                            : 1e-300 * (E + lo);
            if (w != 0 && (A.debug_phases & 8)) deposit(ix, iy, sl, w);
          }
          qhead = (qhead + n) & (QLEN - 1);
          qn -= n;
          n_surv += n;
        };
        // tail class: erf(hi) - erf(lo) = (sgn hi - sgn lo) - sgn(hi) erfc|hi| + sgn(lo) erfc|lo| with
        // exp(E) erfc(z) = t exp(E - z^2 + P(2t-1)), t = 1/(1+z/2) (tools/gen_erfc32.py); the exponent is reduced in f64
        // (k = rint(arg*log2e)) so the f32 part only sees |r| <= 0.5 + |P| log2e
        auto processT = [&](int n) {
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          if (lane < n && (A.debug_phases & 1)) {
            unsigned int en = qt[(qthead + lane) & (QLEN - 1)];
            const int ix = en >> 12, iy = (en >> 6) & 63, sl = en & 63;
            double b = -(s_px[ix][0] + s_py[iy][0] + s_pz[sl][0]);
            double delta = s_px[ix][1] + s_py[iy][1] + s_pz[sl][1];
            double E = b * b * inv4a - delta;
            double lo = b * inv_sa2, hi = lo + hi_off;
            const float L2E = 1.44269504088896341f;
            auto scaled_exp = [&](double arg, float extra) -> float {   // exp(arg + extra), arg <= 0
              double a2 = arg * 1.4426950408889634074;
              double kk = rint(a2);
              float r = (float)(a2 - kk) + extra * L2E;
              return ldexpf(__builtin_amdgcn_exp2f(r), (int)fmax(kk, -1000.0));
            };
            auto Tf = [&](double z) -> float {                          // exp(E) erfc(z), z >= 0
              float t = __builtin_amdgcn_rcpf(1.0f + 0.5f * (float)z);
              float u = 2.0f * t - 1.0f;
              float P = 3.337358939e-04f;
              P = fmaf(P, u, -2.079091206e-04f);
              P = fmaf(P, u, -2.048734760e-03f);
              P = fmaf(P, u, 1.776564635e-03f);
              P = fmaf(P, u, 8.703812035e-03f);
              P = fmaf(P, u, -9.873768747e-03f);
              P = fmaf(P, u, -4.687488818e-02f);
              P = fmaf(P, u, 4.734310629e-02f);
              P = fmaf(P, u, 6.726422172e-01f);
              P = fmaf(P, u, -6.717940761e-01f);
              return t * scaled_exp(E - z * z, P);
            };
            const float th = Tf(fabs(hi)), tl = Tf(fabs(lo));
            float wf = (hi < 0 ? th : -th) + (lo < 0 ? -tl : tl);
            if (lo < 0 && !(hi < 0)) wf += 2.0f * scaled_exp(E, 0.0f);
            double w = (A.debug_phases & 4) ? pref * (double)wf : 1e-300 * (E + lo);
            if (w != 0 && (A.debug_phases & 8)) deposit(ix, iy, sl, w);
          }
          qthead = (qthead + n) & (QLEN - 1);
          qtn -= n;
          n_surv += n;
        };
        const int npz = NS * n_sl;
        const int npz_pad = (npz + 63) & ~63;
        for (int p0 = wv * 64; p0 < npz_pad && (A.debug_phases & 2); p0 += CUR_THREADS) {
          const int p = p0 + lane;
          bool pv = p < npz;
          const int iy = pv ? p / n_sl : 0, sl = pv ? p - iy * n_sl : 0;
          pv = pv && (s_jcell[iy] >= 0);
          const double byz = s_py[iy][0] + s_pz[sl][0], dyz = s_py[iy][1] + s_pz[sl][1];
          const unsigned int tag = ((unsigned)iy << 6) | (unsigned)sl;
          for (int gi = 0; gi < g_nix; gi++) {
            const int ix = s_ixord[g_ix0 + gi];
            bool keep = pv, tail = false;
            if ((do_prune || do_tail) && keep) {
              double b = -(s_px[ix][0] + byz);
              double E2 = b * b * inv4a - (s_px[ix][1] + dyz);
              double lo = b * inv_sa2, hi = lo + hi_off;
              if (lo > 0) E2 -= lo * lo;
              else if (hi < 0) E2 -= hi * hi;
              keep = !(do_prune && E2 < cut);
              tail = do_tail && E2 < tcut;
            }
            const unsigned long long m = __ballot(keep && !tail), mt = __ballot(keep && tail);
            if (m) {
              if (keep && !tail)
                q[(qhead + qn + __popcll(m & ((1ull << lane) - 1ull))) & (QLEN - 1)] = ((unsigned)ix << 12) | tag;
              qn += __popcll(m);
              if (qn >= 64) process(64);
            }
            if (mt) {
              if (keep && tail)
                qt[(qthead + qtn + __popcll(mt & ((1ull << lane) - 1ull))) & (QLEN - 1)] = ((unsigned)ix << 12) | tag;
              qtn += __popcll(mt);
              if (qtn >= 64) processT(64);
            }
          }
        }
        if (qn > 0) process(qn);
        if (qtn > 0) processT(qtn);
      }
      __syncthreads();
      // ---- active cells, their shift range, and the compact emit -------------------------------------------------
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
        int nact = 0, nblk_tot = 0;
        for (int base = 0; base < ncell; base += 64) {
          int cell = base + lane;
          bool act = (cell < ncell) && (s_culo[cell] != 255);
          unsigned long long am = __ballot(act);
          int nb = act ? ((s_cuhi[cell] - (s_culo[cell] & ~7)) / 8 + 1) : 0;
          // inclusive scan of block counts within the 64 lanes
          int sc = nb;
          for (int off = 1; off < 64; off <<= 1) {
            int o = __shfl_up(sc, off);
            if (lane >= off) sc += o;
          }
          if (act) {
            int pos = nact + __popcll(am & ((1ull << lane) - 1ull));
            s_list[pos] = (unsigned short)cell;
            s_boff[pos] = (unsigned short)(nblk_tot + sc - nb);
          }
          nact += __popcll(am);
          nblk_tot += __shfl(sc, 63);
        }
        if (lane == 0) {
          s_misc[6] = nact;
          s_misc[7] = nblk_tot;
          int have = s_misc[16];
          unsigned long long need = (unsigned long long)nblk_tot * 8ull;
          unsigned long long base = 0;
          const int item_cap = (A.split_max_items > 0 && A.split_max_items < IMAX) ? A.split_max_items : IMAX;
          bool ok = (have + nact <= item_cap) && !s_misc[18];
          if (ok && need) {
            base = atomicAdd(S.cursor, need);
            if (base + need > S.wbuf_cap) ok = false;
          }
          if (!ok) s_misc[18] = 1;
          s_base64 = base;
        }
      }
      __syncthreads();
      {
        const int nact = s_misc[6];
        if (!s_misc[18] && nact > 0) {
          const int have = s_misc[16];
          const unsigned long long base = s_base64;
          const unsigned short* boff = s_boff;
          for (int li = wv; li < nact; li += NWAVE) {
            const int cell = s_list[li];
            const int ulo8 = s_culo[cell] & ~7;
            const int nblk = (s_cuhi[cell] - ulo8) / 8 + 1;
            const unsigned long long wo = base + (unsigned long long)boff[li] * 8ull;
            if (lane == 0) {
              Item itx;
              itx.cell_nblk = (s_coli[col0 + cell / NJ] * A.nj + (jmin + cell % NJ)) | (nblk << 16);
              itx.sbase = u_min + ulo8;
              itx.woff_lo = (uint32_t)(wo & 0xFFFFFFFFull);
              itx.woff_hi = (uint32_t)(wo >> 32);
              items[have + li] = itx;
            }
            if (lane < nblk * 8) S.wbuf[wo + lane] = s_A[cell * NU8 + ulo8 + lane];
          }
        }
      }
      __syncthreads();
      if (tid == 0 && !s_misc[18]) s_misc[16] += s_misc[6];
    }
    // ---- window-edge corrections of this chunk -> (tick, value) list ------------------------------------------------
    __syncthreads();
    if (wv == 0) {
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
  if (lane == 0 && n_surv) stat_add(A.counters, 1, n_surv);
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
  }
}

// =============================================================================================================
template <int M>
__global__ void __launch_bounds__(CUR_THREADS, (M == 1 ? 4 : 3)) mac_kernel(SplitArgs S) {
  const CurArgs& A = S.c;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t pair = blockIdx.x;
  if (pair >= A.n_pairs) return;
  const int32_t* hdr = S.hdr + pair * HDR_INTS;
  if (hdr[7]) return;                               // overflowed: the monolithic kernel writes this pair
  float* out = A.out + pair * (int64_t)A.T;
  const int n_items = hdr[0], n_corr = hdr[1], it0 = hdr[2], T = hdr[3], it_w0 = hdr[4], it_w1 = hdr[5];
  if (n_items <= 0 || it_w1 <= it_w0) {
    for (int it = tid; it < A.T; it += CUR_THREADS) out[it] = 0.f;
    return;
  }
  constexpr int ROWLEN = M * WTILE + NU_MAX + 16;
  constexpr int ROWBUF = ROWLEN + ROWLEN / (8 * M) + 8;
  // wave-private staged response rows; once every wave has left the correlation loop (the barrier that opens the
  // combine step) the same memory is the tile buffer s_out, and the barrier closing a supertile hands it back
  __shared__ double s_rowbuf[NWAVE * ROWBUF];
  static_assert(NWAVE * ROWBUF >= TILE_TICKS, "s_out aliases the row buffers");
  double* const s_out = s_rowbuf;
  __shared__ double s_w[NWAVE][64];                 // wave-private weights of the current item
  // Item descriptors (16 B) are read two items ahead of their use, the row + weight loads they describe one item ahead,
  // so both latencies sit behind a whole item of FMAs.  At M = 1 the list (<= 512 items, 8 KB) is copied to LDS first:
  // measured 2.5 % faster than reading descriptors from HBM.  At M = 2 the list is longer (<= 2048 items, 32 KB) and LDS
  // is what decides between 2 and 3 workgroups per CU, so descriptors are read straight from HBM (wave-uniform address):
  // 3 workgroups per CU, measured 5.7 % faster.
  const Item* __restrict__ gitems = S.items + pair * (int64_t)ItemCap<M>::value;
  constexpr int N_LDS_ITEMS = (M == 1) ? ItemCap<1>::value : 1;
  __shared__ Item s_items[N_LDS_ITEMS];
  if constexpr (M == 1) {
    for (int i = tid; i < n_items; i += CUR_THREADS) s_items[i] = gitems[i];
    __syncthreads();
  }
  const int k_stage_lo = max(0, A.k_first);
  int k_stage_hi;
  {
    const double V = A.c->time_window / A.c->response_sampling;
    int kn = (int)ceil(V + 0.5 + 1e-6);
    int ka = (int)floor(V - 0.5 - 1e-6);
    if ((double)ka + 0.5 >= V - 1e-6) ka--;
    int k_top = kn - 1;
    if (k_top - ka > NEDGE - 1) k_top = ka + NEDGE - 1;
    k_stage_hi = min(min(k_top, A.nk - 1), A.k_last);
  }
  unsigned long long n_blocks = 0;

  for (int sup0 = it_w0; sup0 < it_w1; sup0 += TILE_TICKS) {
    const int wlen = min(it_w1 - sup0, TILE_TICKS);
    const int ntt = (wlen + WTILE - 1) / WTILE;
    int my_tile, share_rank, nshare;
    if (ntt >= 3) { my_tile = wv; share_rank = 0; nshare = 1; }
    else if (ntt == 2) { my_tile = wv >> 1; share_rank = wv & 1; nshare = 2; }
    else { my_tile = 0; share_rank = wv; nshare = 4; }
    const bool tile_live = my_tile < ntt;
    const int tb = sup0 + my_tile * WTILE;
    double acc[TPL];
#pragma unroll
    for (int j = 0; j < TPL; j++) acc[j] = 0;

    if (tile_live) {
      double* rowp = s_rowbuf + wv * ROWBUF;
      double* wl = s_w[wv];
      constexpr int NLOAD = (M * WTILE + NU_MAX + 8 + 63) / 64;
      double pre[NLOAD], prew = 0;
      // padded LDS position of this lane's n-th staged element: cached in registers at M = 1; at M = 2 the row is twice
      // as long and the 18 registers are worth more than the few integer ops per item
      constexpr int NPOS = (M == 1) ? NLOAD : 1;
      int spos[NPOS];
      if constexpr (M == 1) {
#pragma unroll
        for (int n = 0; n < NLOAD; n++) spos[n] = rpos<M>(lane + 64 * n);
      }
      auto descriptor = [&](int li) -> Item {
        if constexpr (M == 1) return s_items[li];
        else return gitems[__builtin_amdgcn_readfirstlane(li)];
      };
      auto fetch = [&](const Item& itx) {
        const int nblk = (itx.cell_nblk >> 16) & 0xFF;
        const double* rrow = A.resp + (int64_t)(itx.cell_nblk & 0xFFFF) * A.nk;
        const int kb = M * tb + itx.sbase;           // row element r  <->  response index k = kb + r
        const int nrow = M * WTILE + nblk * 8 + 8;
        const double* src = rrow + kb + lane;
        if (kb >= k_stage_lo && kb + 64 * NLOAD - 1 <= k_stage_hi) {
          // the whole staged window lies inside the live response range: plain coalesced loads
#pragma unroll
          for (int n = 0; n < NLOAD; n++) pre[n] = src[64 * n];
        } else {
          const int r_lo = k_stage_lo - kb, r_hi = min(nrow, k_stage_hi - kb + 1);
#pragma unroll
          for (int n = 0; n < NLOAD; n++) {
            const int r = lane + 64 * n;
            pre[n] = (r >= r_lo && r < r_hi) ? src[64 * n] : 0.0;
          }
        }
        const unsigned long long wo = ((unsigned long long)itx.woff_hi << 32) | (unsigned long long)itx.woff_lo;
        prew = (lane < nblk * 8) ? S.wbuf[wo + lane] : 0.0;
      };
      Item d_cur{}, d_next{};
      if (share_rank < n_items) {
        d_cur = descriptor(share_rank);
        fetch(d_cur);
      }
      if (share_rank + nshare < n_items) d_next = descriptor(share_rank + nshare);
      for (int li = share_rank; li < n_items; li += nshare) {
        const int nblk = (d_cur.cell_nblk >> 16) & 0xFF;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
        for (int n = 0; n < NLOAD; n++)
          if (n < NLOAD - 1 || lane + 64 * n < ROWLEN) {
            if constexpr (M == 1) rowp[spos[n]] = pre[n];
            else rowp[rpos<M>(lane + 64 * n)] = pre[n];
          }
        wl[lane] = prew;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (li + nshare < n_items) fetch(d_next);
        d_cur = d_next;
        if (li + 2 * nshare < n_items) d_next = descriptor(li + 2 * nshare);
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
            const double av = wl[u0 + du];
#pragma unroll
            for (int j = 0; j < TPL; j++) acc[j] = fma(av, w[M * j + du], acc[j]);
          }
#pragma unroll
          for (int q = 0; q < M * (TPL - 1) + 1; q++) w[q] = w[q + 8];
        }
        n_blocks += nblk;
      }
    }
    // ---- combine waves sharing a tile, window-edge corrections, mask, f32 store -------------------------------------
    for (int rnk = 0; rnk < nshare; rnk++) {
      __syncthreads();
      if (tile_live && share_rank == rnk) {
#pragma unroll
        for (int j = 0; j < TPL; j++) {
          const int idx = my_tile * WTILE + TPL * lane + j;
          s_out[idx] = (rnk == 0) ? acc[j] : s_out[idx] + acc[j];
        }
      }
    }
    __syncthreads();
    {
      const Corr* cr = S.corr + pair * CMAX;
      for (int k = tid; k < n_corr; k += CUR_THREADS) {
        int i = cr[k].tick - sup0;
        if (i >= 0 && i < wlen) atomicAdd(&s_out[i], -cr[k].val);
      }
    }
    __syncthreads();
    for (int i = tid; i < wlen; i += CUR_THREADS) {
      int it = sup0 + i;
      if (it < A.T) out[it] = (it >= it0 && it < T) ? (float)s_out[i] : 0.f;
    }
    __syncthreads();
  }
  for (int it = tid; it < A.T; it += CUR_THREADS)
    if (it < it_w0 || it >= it_w1) out[it] = 0.f;
  if (lane == 0 && n_blocks) stat_add(A.counters, 5, n_blocks * 64ull * 64ull);
}

// =============================================================================================================
// M of the split path for these constants, 0 = configuration not covered (caller uses the monolithic kernel)
static int split_M(const ldsim_ctx* ctx, const CurArgs& args) {
  const LdsimConsts& h = ctx->h_consts;
  double ratio = h.time_sampling / h.response_sampling;
  int M = (int)llround(ratio);
  if (M < 1 || M > 2 || fabs(ratio - M) > 1e-9 || h.sampled_points > NS_MAX || args.nj > NJ_MAX || args.ni > 64 ||
      args.ni * args.nj > 65535 || args.n_pairs > 0x7fffffffLL)
    return 0;
  return M;
}

static SplitArgs split_args(const CurArgs& args, void* items, void* hdr, void* corr, double* wbuf,
                            unsigned long long wbuf_cap, unsigned long long* cursor) {
  SplitArgs S;
  S.c = args;
  S.items = (Item*)items;
  S.hdr = (int32_t*)hdr;
  S.corr = (Corr*)corr;
  S.wbuf = wbuf;
  S.wbuf_cap = wbuf_cap;
  S.cursor = cursor;
  return S;
}

extern "C++" int mac_shift_launch(ldsim_ctx* ctx, SplitArgs S, int M);

// returns 0 = launched, 1 = not covered by the split path, < 0 = error
extern "C++" int split_launch_weights(ldsim_ctx* ctx, const CurArgs& args, void* items, void* hdr, void* corr, double* wbuf,
                                      unsigned long long wbuf_cap, unsigned long long* cursor) {
  if (args.n_pairs == 0) return 0;
  const int M = split_M(ctx, args);
  if (!M) return 1;
  SplitArgs S = split_args(args, items, hdr, corr, wbuf, wbuf_cap, cursor);
  if (ctx->weights_mode) {
    int rc = ldsim_ensure(ctx, SB_PPAR, qpair_params_bytes(args.n_pairs));
    if (rc) return rc;
    return qweights_launch(ctx, S, M, ctx->scratch[SB_PPAR].p);
  }
  if (M == 1) hipLaunchKernelGGL(weights_kernel<1>, dim3((unsigned)args.n_pairs), dim3(CUR_THREADS), 0, ctx->stream, S);
  else hipLaunchKernelGGL(weights_kernel<2>, dim3((unsigned)args.n_pairs), dim3(CUR_THREADS), 0, ctx->stream, S);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C++" int split_launch_mac(ldsim_ctx* ctx, const CurArgs& args, void* items, void* hdr, void* corr, double* wbuf,
                                  unsigned long long wbuf_cap, unsigned long long* cursor) {
  if (args.n_pairs == 0) return 0;
  const int M = split_M(ctx, args);
  if (!M) return 1;
  SplitArgs S = split_args(args, items, hdr, corr, wbuf, wbuf_cap, cursor);
  if (ctx->mac_mode) return mac_shift_launch(ctx, S, M);            // kernels_macshift.hip
  if (M == 1) hipLaunchKernelGGL(mac_kernel<1>, dim3((unsigned)args.n_pairs), dim3(CUR_THREADS), 0, ctx->stream, S);
  else hipLaunchKernelGGL(mac_kernel<2>, dim3((unsigned)args.n_pairs), dim3(CUR_THREADS), 0, ctx->stream, S);
  HIPCHK(hipGetLastError());
  return 0;
}

// per-pair record sizes of the split path's HBM lists for these constants; returns M (0 = not covered)
extern "C++" int split_sizes(const ldsim_ctx* ctx, const CurArgs& args, size_t* item_bytes, size_t* hdr_bytes,
                             size_t* corr_bytes) {
  const int M = split_M(ctx, args);
  *item_bytes = sizeof(Item) * (M == 2 ? ItemCap<2>::value : ItemCap<1>::value);
  *hdr_bytes = sizeof(int32_t) * HDR_INTS;
  *corr_bytes = sizeof(Corr) * CMAX;
  return M;
}
