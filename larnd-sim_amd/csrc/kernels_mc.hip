// kernels_mc.hip -- detsim.tracks_current_mc (larndsim/detsim.py:258-348): Monte-Carlo estimate of the induced current that the
// reference driver runs (cli/simulate_pixels.py:1016).  Per (segment, pixel): the part of the segment within the response
// table's reach of the pixel centre (overlapping_segment, detsim.py:220-256) is cut into ~MIN_STEP_SIZE steps; for every
// tick each step's charge is smeared by the diffusion (three normals) and looked up in the response table.
//
// The reference gives all 64 tick threads of a block the same rng_states[itrk + ntrk*ipix] (detsim.py:324) and lets them
// race on it, so its output is not reproducible.  Here a workgroup owns the pair (geometry computed once), a thread owns
// ticks tid, tid + 256, .. and every (pair, tick) has its own xoroshiro128p stream derived from that state of the table:
// reproducible with a seed, statistically equivalent to the reference, never equal to it (unpinned).
#include "ldsim_args.h"
#include "rng.h"

__device__ __forceinline__ uint64_t splitmix_fin(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
__host__ __device__ inline void mc_stream_words(uint64_t b0, uint64_t b1, uint32_t it, uint64_t& s0, uint64_t& s1);

struct McArgs {
  CurArgs c;
  const RngState* states;
  int64_t n_seg;           // ntrk of the state index  itrk + ntrk * ipix
};

__global__ void __launch_bounds__(256) current_mc_kernel(McArgs Mc) {
  const CurArgs& A = Mc.c;
  const LdsimConsts* c = A.c;
  const int64_t pair = blockIdx.x;
  if (pair >= A.n_pairs) return;
  const int tid = threadIdx.x;
  int64_t r, ipix, pID;
  if (A.pair_val) {
    const int32_t v = A.pair_val[pair];
    r = v / A.P;
    ipix = v - r * A.P;
    pID = (int64_t)((A.pair_key[pair] >> 4) & 0xFFFFFFFFull);
  } else {
    r = pair / A.P;
    ipix = pair - r * A.P;
    pID = A.pixels[pair];
  }
  const int64_t seg = A.seg_begin + r;
  float* out = A.out + pair * (int64_t)A.T;
  int T = A.T;
  if (A.tmax_batch) T = min(T, A.tmax_batch[A.s.batch[seg] - A.batch0]);

  __shared__ double g[20];
  __shared__ int gi[4];
  if (tid == 0) {
    gi[0] = 0;
    const SegStore& s = A.s;
    int64_t px, py, pplane;
    id2pixel(c, pID, px, py, pplane);
    const int32_t tplane = s.pixel_plane[seg];
    const int64_t bplane = pplane < 0 ? pplane + c->n_tpc : pplane;
    if (px >= 0 && py >= 0 && bplane >= 0 && bplane < c->n_tpc && tplane >= 0 && tplane < c->n_tpc) {
      const double(*pb)[2] = c->tpc_borders[bplane];
      const double x_p = px * c->pixel_pitch + pb[0][0] + c->pixel_pitch / 2;
      const double y_p = py * c->pixel_pitch + pb[1][0] + c->pixel_pitch / 2;
      double st[3], en[3];
      const double xs = s.f[LDSIM_X_START][seg], ys = s.f[LDSIM_Y_START][seg], zs = s.f[LDSIM_Z_START][seg];
      const double xe = s.f[LDSIM_X_END][seg], ye = s.f[LDSIM_Y_END][seg], ze = s.f[LDSIM_Z_END][seg];
      if (zs < ze) { st[0] = xs; st[1] = ys; st[2] = zs; en[0] = xe; en[1] = ye; en[2] = ze; }
      else { en[0] = xs; en[1] = ys; en[2] = zs; st[0] = xe; st[1] = ye; st[2] = ze; }
      const double t_start = py_round((s.f[LDSIM_T_START][seg] - s.f[LDSIM_T0_START][seg] - c->time_padding) / c->time_sampling) *
                             c->time_sampling;
      const double sx = en[0] - st[0], sy = en[1] - st[1], sz = en[2] - st[2];
      const double length = sqrt(sx * sx + sy * sy + sz * sz);
      const double dirx = sx / length, diry = sy / length, dirz = sz / length;
      const double impact = sqrt((double)A.ni * A.ni + (double)A.nj * A.nj) * c->response_bin_size;
      // overlapping_segment (detsim.py:220-256)
      const double dx = x_p - st[0], dy = y_p - st[1];
      double vx = en[0] - st[0], vy = en[1] - st[1];
      const double l = sqrt(vx * vx + vy * vy);
      vx /= l; vy /= l;
      const double sp = (dx * vx + dy * vy) / l;
      const double rx = dx - vx * sp * l, ry = dy - vy * sp * l;
      const double rr = sqrt(rx * rx + ry * ry);
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
      const double ux = ne[0] - ns[0], uy = ne[1] - ns[1], uz = ne[2] - ns[2];
      const double sublen = sqrt(ux * ux + uy * uy + uz * uz);
      if (sublen > 0 && length > 0 && sublen < 1e6) {   // == 0: return (detsim.py:315-316); NaN geometry: no signal
        const double nstep_f = fmax(py_round(sublen / c->min_step_size), 1.0);
        if (nstep_f < 2.0e9) {
          const int nstep = (int)nstep_f;
          g[0] = x_p; g[1] = y_p; g[2] = ns[0]; g[3] = ns[1]; g[4] = ns[2];
          g[5] = dirx; g[6] = diry; g[7] = dirz;
          g[8] = sublen / nstep;                                        // step
          g[9] = s.f[LDSIM_N_ELECTRONS][seg] * (sublen / length) / ((double)nstep * c->mc_sample_multiplier);   // charge
          g[10] = s.f[LDSIM_TRAN_DIFF][seg]; g[11] = s.f[LDSIM_LONG_DIFF][seg];
          g[12] = t_start; g[13] = c->tpc_borders[tplane][2][0];
          gi[1] = nstep;
          gi[0] = 1;
        }
      }
    }
  }
  __syncthreads();
  if (!gi[0]) {
    for (int it = tid; it < A.T; it += 256) out[it] = 0.f;
    return;
  }
  const double x_p = g[0], y_p = g[1], sx0 = g[2], sy0 = g[3], sz0 = g[4], dirx = g[5], diry = g[6], dirz = g[7];
  const double step = g[8], charge = g[9], sT = g[10], sL = g[11], t_start = g[12], z_anode = g[13];
  const int nstep = gi[1], mult = c->mc_sample_multiplier;
  const double dt = c->time_sampling, dtr = c->response_sampling, TW = c->time_window, bin = c->response_bin_size;
  const RngState base = Mc.states[r + Mc.n_seg * ipix];
  for (int it = tid; it < A.T; it += 256) {
    const double time_tick = t_start + it * dt;
    if (it >= T || time_tick < 0) {      // beyond this batch's max_length / detsim.py:297-298: the signal stays 0
      out[it] = 0.f;
      continue;
    }
    RngState rs;
    mc_stream_words(base.s0, base.s1, (uint32_t)it, rs.s0, rs.s1);
    double total = 0;
    for (int istep = 0; istep < nstep; istep++)
      for (int m = 0; m < mult; m++) {
        double x = sx0 + step * (istep + 0.5) * dirx;
        double y = sy0 + step * (istep + 0.5) * diry;
        double z = sz0 + step * (istep + 0.5) * dirz;
        z += (double)rng_normal_f32(rs) * sL;
        const double t0 = fabs(z - z_anode) / c->v_drift - TW;
        if (!(t0 < time_tick && time_tick < t0 + TW)) continue;
        x += (double)rng_normal_f32(rs) * sT;
        y += (double)rng_normal_f32(rs) * sT;
        const double xd = fabs(x_p - x), yd = fabs(y_p - y);
        if (xd > bin * A.ni) continue;
        if (yd > bin * A.nj) continue;
        const int64_t i = (int64_t)py_round(xd / bin - 0.5), j = (int64_t)py_round(yd / bin - 0.5);
        const int64_t k = (int64_t)py_round((time_tick - t0) / dtr);
        if (i >= 0 && i < A.ni && j >= 0 && j < A.nj && k >= 0 && k < A.nk) total += charge * A.resp[(i * A.nj + j) * (int64_t)A.nk + k];
      }
    out[it] = (float)total;
  }
}

// the (pair, tick) stream: SplitMix64 finaliser of the table state's words offset by the tick (never the all-zero state)
__host__ __device__ inline void mc_stream_words(uint64_t b0, uint64_t b1, uint32_t it, uint64_t& s0, uint64_t& s1) {
  uint64_t z0 = b0 + 0x9E3779B97F4A7C15ULL * (uint64_t)(it + 1u);
  uint64_t z1 = b1 ^ (0xD1B54A32D192ED03ULL * (uint64_t)(it + 1u));
  z0 = (z0 ^ (z0 >> 30)) * 0xBF58476D1CE4E5B9ULL; z0 = (z0 ^ (z0 >> 27)) * 0x94D049BB133111EBULL; z0 ^= z0 >> 31;
  z1 = (z1 ^ (z1 >> 30)) * 0xBF58476D1CE4E5B9ULL; z1 = (z1 ^ (z1 >> 27)) * 0x94D049BB133111EBULL; z1 ^= z1 >> 31;
  if ((z0 | z1) == 0) z0 = 1;
  s0 = z0;
  s1 = z1;
}

__global__ void __launch_bounds__(256) rng_step_kernel(RngState* states, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  RngState st = states[i];
  (void)rng_next(st);
  states[i] = st;
}

// signals for args.n_pairs pairs; state index = (relative segment) + n_seg * ipix.  Uses (and steps once) states [0, n_seg * P).
extern "C++" int current_mc_launch(ldsim_ctx* ctx, const CurArgs& args, int64_t n_seg) {
  if (args.n_pairs == 0) return 0;
  const LdsimConsts& h = ctx->h_consts;
  if (!(h.min_step_size > 0) || h.mc_sample_multiplier < 1) {
    ldsim_set_error("tracks_current_mc needs MIN_STEP_SIZE > 0 and MC_SAMPLE_MULTIPLIER >= 1");
    return LDSIM_EINVAL;
  }
  if (args.n_pairs > 0x7fffffffLL) {
    ldsim_set_error("too many pairs for one launch");
    return LDSIM_EINVAL;
  }
  const int64_t n_states = n_seg * (int64_t)args.P;
  int rc = rng_ensure_states(ctx, n_states);
  if (rc) return rc;
  McArgs M;
  M.c = args;
  M.states = (const RngState*)ctx->d_rng.p;
  M.n_seg = n_seg;
  hipLaunchKernelGGL(current_mc_kernel, dim3((unsigned)args.n_pairs), dim3(256), 0, ctx->stream, M);
  HIPCHK(hipGetLastError());
  hipLaunchKernelGGL(rng_step_kernel, dim3((unsigned)((n_states + 255) / 256)), dim3(256), 0, ctx->stream,
                     (RngState*)ctx->d_rng.p, n_states);
  HIPCHK(hipGetLastError());
  return 0;
}
