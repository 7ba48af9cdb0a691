// kernels_light.hip -- a17 lightLUT.calculate_light_incidence (larndsim/lightLUT.py:15-136) and
// a18 light_sim.sum_light_signals (larndsim/light_sim.py:58-129).
// The LUT is held as SoA planes (vis, t0, t0_avg, time_dist) instead of the reference's structured array.
#include "ldsim_dev.h"

__device__ __forceinline__ void get_voxel(const LdsimConsts* c, double x, double y, double z, int itpc, int nx, int ny,
                                          int nz, int& i, int& j, int& k) {
  const double(*b)[2] = c->tpc_borders[itpc];
  bool is_even = b[2][1] > b[2][0];
  double x_min = b[0][0] - 2e-2, x_max = b[0][1] + 2e-2, y_min = b[1][0] - 2e-2, y_max = b[1][1] + 2e-2;
  double z_min = b[2][0] - 2e-2, z_max = b[2][1] + 2e-2;
  i = is_even ? (int)((x - x_min) / (x_max - x_min) * nx) : (int)((x_max - x) / (x_max - x_min) * nx);
  j = (int)((y_max - y) / (y_max - y_min) * ny);
  k = (int)((z - z_min) / (z_max - z_min) * nz);
  i = min(nx - 1, max(0, i));
  j = min(ny - 1, max(0, j));
  k = min(nz - 1, max(0, k));
}

// one thread per (segment, output channel): the N x n_op x 8 B output write dominates, keep it coalesced
__global__ void __launch_bounds__(256) light_incidence_kernel(SegStore s, const LdsimConsts* __restrict__ c,
                                                              const float* __restrict__ vis, const float* __restrict__ t0lut,
                                                              int nx, int ny, int nz, int ndet,
                                                              const double* __restrict__ eff, const int32_t* __restrict__ ch2tpc,
                                                              int n_out, float* __restrict__ nph, float* __restrict__ t0det,
                                                              int32_t* __restrict__ voxel) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= s.n * n_out) return;
  int64_t it = idx / n_out;
  int o = (int)(idx % n_out);
  int itpc = s.pixel_plane[it];
  if (itpc == c->default_plane_index || itpc < 0 || itpc >= c->n_tpc) return;
  int i, j, k;
  get_voxel(c, s.f[LDSIM_X][it], s.f[LDSIM_Y][it], s.f[LDSIM_Z][it], itpc, nx, ny, nz, i, j, k);
  if (o == 0) {
    voxel[it * 3 + 0] = i;
    voxel[it * 3 + 1] = j;
    voxel[it * 3 + 2] = k;
  }
  int imod = itpc / 2;
  int channel_offset = (n_out < c->n_op_channel) ? n_out * imod : 0;
  int op = o + channel_offset, li = o % ndet;
  int64_t vb = (((int64_t)i * ny + j) * nz + k) * ndet + li;
  double v = (double)vis[vb] * (ch2tpc[op] == itpc ? 1 : 0);
  nph[idx] = (float)(eff[op] * v * s.f[LDSIM_N_PHOTONS][it]);
  if (c->light_trig_mode == 0) {
    const double ns = 1.0, mus = 1e-6 * 1e9;
    t0det[idx] = (float)(((double)t0lut[vb] * ns + s.f[LDSIM_T0][it] * mus) / mus);
  }
}

// literal (detector, tick) thread like the reference: f32 accumulator updated in sorted_indices order
__global__ void __launch_bounds__(64) sum_light_signals_kernel(SegStore s, const LdsimConsts* __restrict__ c,
                                                               const int32_t* __restrict__ voxel,
                                                               const int64_t* __restrict__ track_id,
                                                               const float* __restrict__ nph, int n_inc,
                                                               const int32_t* __restrict__ op_channel, int n_det,
                                                               const float* __restrict__ t0_avg,
                                                               const float* __restrict__ time_dist, int ny, int nz,
                                                               int ndet_lut, int nprof, double start_time,
                                                               const int32_t* __restrict__ sorted_idx, int64_t n_ticks,
                                                               float* __restrict__ out, int64_t* __restrict__ true_id,
                                                               double* __restrict__ true_ph, int max_truth) {
  int idet = blockIdx.x;
  int64_t itick = (int64_t)blockIdx.y * blockDim.x + threadIdx.x;
  if (idet >= n_det || itick >= n_ticks) return;
  const double ns = 1.0, mus = 1e-6 * 1e9, tick = c->light_tick_size;
  const int64_t n = s.n;
  double st = itick * tick + start_time, en = st + tick;
  int opch = op_channel[idet];
  int idet_lut = opch % ndet_lut;
  float acc = out[(int64_t)idet * n_ticks + itick];
  for (int64_t q = 0; q < n; q++) {
    int64_t itrk = sorted_idx[(int64_t)idet * n + q];
    float ph = nph[itrk * n_inc + opch];
    if (!(ph > 0)) continue;
    double track_time = s.f[LDSIM_T0][itrk];
    double track_end = track_time + nprof * ns / mus;
    if (track_end < st || track_time > en) continue;
    const int32_t* vx = voxel + itrk * 3;
    int64_t lb = ((((int64_t)vx[0] * ny + vx[1]) * nz + vx[2]) * ndet_lut + idet_lut);
    if (c->enable_lut_smearing) {
      const float* prof = time_dist + lb * nprof;
      for (int ip = 0; ip < nprof; ip++) {
        double pt = track_time + ip * ns / mus;
        if (pt < en && pt > st) {
          double photons = (double)ph * (double)prof[ip] / tick;
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
      double pt = track_time + (double)t0_avg[lb] * ns / mus;
      if (pt < en && pt > st) {
        double photons = (double)ph / tick;
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
  out[(int64_t)idet * n_ticks + itick] = acc;
}


// Scatter form of the photon sum for the (default) case without truth slots: one workgroup per (detector, tick tile)
// keeps the tile in LDS as f64 and every thread walks its share of the segments, adding each contribution to the one
// tick whose (start, end) window the reference's strict inequalities select (candidates it-1, it, it+1 are tested with
// the reference's own expressions).  Work is O(n_det * tiles * S + contributions) instead of O(n_det * n_ticks * S).
// The accumulation order differs from the reference's sorted loop, so the f32 result can differ in the last bits.
#define LTILE 8192
__global__ void __launch_bounds__(256) sum_light_scatter_kernel(SegStore s, const LdsimConsts* __restrict__ c,
                                                                const int32_t* __restrict__ voxel,
                                                                const float* __restrict__ nph, int n_inc,
                                                                const int32_t* __restrict__ op_channel, int n_det,
                                                                const float* __restrict__ t0_avg,
                                                                const float* __restrict__ time_dist, int ny, int nz,
                                                                int ndet_lut, int nprof, double start_time,
                                                                int64_t n_ticks, float* __restrict__ out) {
  __shared__ double acc[LTILE];
  const int idet = blockIdx.x;
  const int64_t tile0 = (int64_t)blockIdx.y * LTILE;
  const int tlen = (int)min((int64_t)LTILE, n_ticks - tile0);
  if (idet >= n_det || tlen <= 0) return;
  const double ns = 1.0, mus = 1e-6 * 1e9, tick = c->light_tick_size;
  for (int i = threadIdx.x; i < tlen; i += 256) acc[i] = 0;
  __syncthreads();
  const int opch = op_channel[idet];
  const int idet_lut = opch % ndet_lut;
  auto deposit = [&](double pt, double photons) {
    double f = floor((pt - start_time) / tick);
    if (!(f > -2.0 && f < (double)n_ticks + 1.0)) return;
    int64_t it0 = (int64_t)f;
    for (int64_t it = it0 - 1; it <= it0 + 1; it++) {
      if (it < tile0 || it >= tile0 + tlen) continue;
      double st = it * tick + start_time, en = st + tick;
      if (pt < en && pt > st) atomicAdd(&acc[it - tile0], photons);
    }
  };
  for (int64_t itrk = threadIdx.x; itrk < s.n; itrk += 256) {
    float ph = nph[itrk * n_inc + opch];
    if (!(ph > 0)) continue;
    double track_time = s.f[LDSIM_T0][itrk];
    const int32_t* vx = voxel + itrk * 3;
    int64_t lb = ((((int64_t)vx[0] * ny + vx[1]) * nz + vx[2]) * ndet_lut + idet_lut);
    if (c->enable_lut_smearing) {
      const float* prof = time_dist + lb * nprof;
      for (int ip = 0; ip < nprof; ip++) {
        float p = prof[ip];
        if (p != 0.f) deposit(track_time + ip * ns / mus, (double)ph * (double)p / tick);
      }
    } else {
      deposit(track_time + (double)t0_avg[lb] * ns / mus, (double)ph / tick);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < tlen; i += 256) {
    int64_t o = (int64_t)idet * n_ticks + tile0 + i;
    out[o] = (float)((double)out[o] + acc[i]);
  }
}

extern "C++" {
int light_launch_incidence(ldsim_ctx* ctx, int n_out, float* nph, float* t0det, int32_t* voxel) {
  int64_t total = ctx->seg.n * n_out;
  if (total == 0) return 0;
  hipLaunchKernelGGL(light_incidence_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, ctx->seg,
                     ctx->d_consts, ctx->d_lut_vis, ctx->d_lut_t0, ctx->lut_nx, ctx->lut_ny, ctx->lut_nz, ctx->lut_ndet,
                     ctx->d_eff, ctx->d_ch2tpc, n_out, nph, t0det, voxel);
  HIPCHK(hipGetLastError());
  return 0;
}
int light_launch_sum(ldsim_ctx* ctx, const int32_t* voxel, const int64_t* track_id, const float* nph, int n_inc,
                     const int32_t* op_channel, int n_det, const int32_t* sorted_idx, double start_time, int64_t n_ticks,
                     float* out, int64_t* true_id, double* true_ph, int max_truth) {
  if (n_det == 0 || n_ticks == 0) return 0;
  if (max_truth == 0) {
    hipLaunchKernelGGL(sum_light_scatter_kernel, dim3(n_det, (unsigned)((n_ticks + LTILE - 1) / LTILE)), dim3(256), 0,
                       ctx->stream, ctx->seg, ctx->d_consts, voxel, nph, n_inc, op_channel, n_det, ctx->d_lut_t0avg,
                       ctx->d_lut_td, ctx->lut_ny, ctx->lut_nz, ctx->lut_ndet, ctx->lut_nprof, start_time, n_ticks, out);
    HIPCHK(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(sum_light_signals_kernel, dim3(n_det, (unsigned)((n_ticks + 63) / 64)), dim3(64), 0, ctx->stream,
                     ctx->seg, ctx->d_consts, voxel, track_id, nph, n_inc, op_channel, n_det, ctx->d_lut_t0avg,
                     ctx->d_lut_td, ctx->lut_ny, ctx->lut_nz, ctx->lut_ndet, ctx->lut_nprof, start_time, sorted_idx,
                     n_ticks, out, true_id, true_ph, max_truth);
  HIPCHK(hipGetLastError());
  return 0;
}
}
