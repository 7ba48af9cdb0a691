// kernels_light.hip -- a17 lightLUT.calculate_light_incidence (larndsim/lightLUT.py:15-136) and
// a18 light_sim.sum_light_signals (larndsim/light_sim.py:58-129), for host-buffer stage calls and for the
// device-resident light leg (ldsim_dev_light_incidence / ldsim_dev_sum_light).
// The LUT is held as SoA planes (vis, t0, t0_avg, time_dist) instead of the reference's structured array.
//
// a17: a workgroup owns a tile of consecutive segments; 1 thread per segment finds the voxel (lightLUT.py:15-63) once,
//      then all threads stream the tile's [segment][channel] outputs in one coalesced run (the write dominates:
//      8 or 4 bytes per (segment, channel)).
// a18: the reference gives every (detector, tick) a thread that walks ALL segments.  Here the work is proportional to
//      the photons that actually arrive:
//      * without truth slots (MAX_MC_TRUTH_IDS = 0): sum_light_scatter_kernel -- a workgroup per (detector, tick tile)
//        keeps the tile in LDS as f64 and its threads deposit the batch's contributions;
//      * with truth slots: the (detector, segment) pairs are walked in the reference's visiting order -- detector after detector,
//        inside one its segments by descending photons (light_sim.py:86, cli/simulate_pixels.py:1141-1144) -- and each emits
//        its contributions, profile bin after profile bin, as records {32-bit key (detector, tick) | 64-bit payload (segment,
//        the f4 product of light_sim.py:100)}; a STABLE radix sort over the key's 20-odd bits (three passes) puts each
//        (detector, tick) cell's records together without disturbing the order they were emitted in, and a wave per block of
//        records replays them: same f4 accumulation order, same first-come truth slots (light_sim.py:103-110,119-127).
//        [Rounds 2-3 keyed the records (detector, tick, rank, bin) in 64 bits -- eight passes over 12-byte pairs -- and fetched
//        photons and segment of a sorted record through its index: 5.2 ms per 2x2 batch of 2.4e7 records.]
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

#define LI_SEGS 32     // segments per workgroup tile
// Outputs are [n][n_out] relative to seg0.  `fill` = 1: segments outside every TPC get zeros (resident arrays are not
// pre-filled by a caller); 0: they are left untouched like the reference (lightLUT.py:86-88) for caller-owned arrays.
__global__ void __launch_bounds__(256) light_incidence_kernel(SegStore s, const LdsimConsts* __restrict__ c, int64_t seg0,
                                                              int64_t n, const float* __restrict__ vis,
                                                              const float* __restrict__ t0lut, int nx, int ny, int nz,
                                                              int ndet, const double* __restrict__ eff,
                                                              const int32_t* __restrict__ ch2tpc, int n_out,
                                                              float* __restrict__ nph, float* __restrict__ t0det,
                                                              int32_t* __restrict__ voxel, int fill) {
  __shared__ int64_t s_vb[LI_SEGS];
  __shared__ int s_tpc[LI_SEGS];
  __shared__ double s_np[LI_SEGS], s_t0[LI_SEGS];
  const int64_t r0 = (int64_t)blockIdx.x * LI_SEGS;
  const int nt = (int)min((int64_t)LI_SEGS, n - r0);
  if (nt <= 0) return;
  if (threadIdx.x < nt) {
    const int64_t r = r0 + threadIdx.x, it = seg0 + r;
    int itpc = s.pixel_plane[it];
    if (itpc == c->default_plane_index || itpc < 0 || itpc >= c->n_tpc) {
      itpc = -1;
      if (fill) { voxel[r * 3 + 0] = 0; voxel[r * 3 + 1] = 0; voxel[r * 3 + 2] = 0; }
    } else {
      int i, j, k;
      get_voxel(c, s.f[LDSIM_X][it], s.f[LDSIM_Y][it], s.f[LDSIM_Z][it], itpc, nx, ny, nz, i, j, k);
      voxel[r * 3 + 0] = i;
      voxel[r * 3 + 1] = j;
      voxel[r * 3 + 2] = k;
      s_vb[threadIdx.x] = (((int64_t)i * ny + j) * nz + k) * ndet;
      s_np[threadIdx.x] = s.f[LDSIM_N_PHOTONS][it];
      s_t0[threadIdx.x] = s.f[LDSIM_T0][it];
    }
    s_tpc[threadIdx.x] = itpc;
  }
  __syncthreads();
  const bool trig0 = c->light_trig_mode == 0;
  const double ns = 1.0, mus = 1e-6 * 1e9;
  // flat walk over the tile's [segment][channel] block: consecutive threads write consecutive addresses
  int sl = (int)(threadIdx.x / n_out), o = (int)(threadIdx.x % n_out);
  const int dsl = 256 / n_out, d_o = 256 % n_out;
  while (sl < nt) {
    const int64_t idx = (r0 + sl) * (int64_t)n_out + o;
    const int itpc = s_tpc[sl];
    if (itpc >= 0) {
      const int channel_offset = (n_out < c->n_op_channel) ? n_out * (itpc / 2) : 0;
      const int op = o + channel_offset;
      const int64_t vb = s_vb[sl] + (o % ndet);
      const double v = (double)vis[vb] * (ch2tpc[op] == itpc ? 1 : 0);
      nph[idx] = (float)(eff[op] * v * s_np[sl]);
      if (trig0) t0det[idx] = (float)(((double)t0lut[vb] * ns + s_t0[sl] * mus) / mus);
    } else if (fill) {
      nph[idx] = 0.f;
      if (trig0) t0det[idx] = 0.f;
    }
    sl += dsl;
    o += d_o;
    if (o >= n_out) { o -= n_out; sl++; }
  }
}

// The same stage with four channels per lane (n_out and the LUT's detector count multiples of 4): the output is a dense
// [segment][channel] f4 array of which only the segment's own TPC's channels are non-zero, so the kernel is a streaming
// write and should run at HBM write speed.  One float4 store per lane and (segment, channel quad); FIXED (n_out ==
// n_op_channel: every segment reads the same slice of ch2tpc) keeps a lane's ch2tpc quads in registers across the tile's 32
// segments; efficiency and visibility are only read for quads that hold a channel of the segment's TPC.  A channel of
// another TPC gets (float)(eff * (vis * 0) * n_photons) in the reference = (float)(0.0 * n_photons) for a finite eff >= 0
// and vis >= 0 (the launcher checks the efficiencies and otherwise takes the one-channel kernel): identical bits.
#define LI_QPT 4       // channel quads per lane and pass: 256 lanes x 4 quads x 4 = 4096 channels per pass
template <bool FIXED>
__global__ void __launch_bounds__(256) light_incidence4_kernel(SegStore s, const LdsimConsts* __restrict__ c, int64_t seg0,
                                                               int64_t n, const float* __restrict__ vis,
                                                               const float* __restrict__ t0lut, int nx, int ny, int nz,
                                                               int ndet, const double* __restrict__ eff,
                                                               const int32_t* __restrict__ ch2tpc, int n_out,
                                                               float* __restrict__ nph, float* __restrict__ t0det,
                                                               int32_t* __restrict__ voxel, int fill) {
  __shared__ int64_t s_vb[LI_SEGS];
  __shared__ int s_tpc[LI_SEGS];
  __shared__ double s_np[LI_SEGS], s_t0[LI_SEGS];
  const int64_t r0 = (int64_t)blockIdx.x * LI_SEGS;
  const int nt = (int)min((int64_t)LI_SEGS, n - r0);
  if (nt <= 0) return;
  if (threadIdx.x < nt) {
    const int64_t r = r0 + threadIdx.x, it = seg0 + r;
    int itpc = s.pixel_plane[it];
    if (itpc == c->default_plane_index || itpc < 0 || itpc >= c->n_tpc) {
      itpc = -1;
      if (fill) { voxel[r * 3 + 0] = 0; voxel[r * 3 + 1] = 0; voxel[r * 3 + 2] = 0; }
    } else {
      int i, j, k;
      get_voxel(c, s.f[LDSIM_X][it], s.f[LDSIM_Y][it], s.f[LDSIM_Z][it], itpc, nx, ny, nz, i, j, k);
      voxel[r * 3 + 0] = i;
      voxel[r * 3 + 1] = j;
      voxel[r * 3 + 2] = k;
      s_vb[threadIdx.x] = (((int64_t)i * ny + j) * nz + k) * ndet;
      s_np[threadIdx.x] = s.f[LDSIM_N_PHOTONS][it];
      s_t0[threadIdx.x] = s.f[LDSIM_T0][it];
    }
    s_tpc[threadIdx.x] = itpc;
  }
  __syncthreads();
  const bool trig0 = c->light_trig_mode == 0;
  const double ns = 1.0, mus = 1e-6 * 1e9;
  const int nquad = n_out >> 2;
  for (int q0 = 0; q0 < nquad; q0 += 256 * LI_QPT) {
    int4 ct0 = make_int4(-1, -1, -1, -1), ct1 = ct0, ct2 = ct0, ct3 = ct0;
    const int qa = q0 + (int)threadIdx.x, qb = qa + 256, qc = qa + 512, qd = qa + 768;
    if (FIXED) {
      if (qa < nquad) ct0 = *(const int4*)(ch2tpc + 4 * qa);
      if (qb < nquad) ct1 = *(const int4*)(ch2tpc + 4 * qb);
      if (qc < nquad) ct2 = *(const int4*)(ch2tpc + 4 * qc);
      if (qd < nquad) ct3 = *(const int4*)(ch2tpc + 4 * qd);
    }
    for (int sl = 0; sl < nt; sl++) {
      const int itpc = s_tpc[sl];
      if (itpc < 0 && !fill) continue;
      float* rowp = nph + (r0 + sl) * (int64_t)n_out;
      float* rowt = t0det + (r0 + sl) * (int64_t)n_out;
      const int64_t vb0 = itpc >= 0 ? s_vb[sl] : 0;
      const double npho = itpc >= 0 ? s_np[sl] : 0.0, t0s = itpc >= 0 ? s_t0[sl] : 0.0;
      const int coff = FIXED ? 0 : n_out * (max(itpc, 0) / 2);
      const float zv = itpc >= 0 ? (float)(0.0 * npho) : 0.f;
      auto quad = [&](int q, int4 cq) {
        if (q >= nquad) return;
        float4 o4 = make_float4(zv, zv, zv, zv);
        if (itpc >= 0) {
          const int op = 4 * q + coff;
          if (!FIXED) cq = *(const int4*)(ch2tpc + op);
          const bool m0 = cq.x == itpc, m1 = cq.y == itpc, m2 = cq.z == itpc, m3 = cq.w == itpc;
          const int64_t vb = vb0 + (4 * q) % ndet;      // ndet % 4 == 0: the quad's LUT entries are consecutive, 16-byte aligned
          if (m0 | m1 | m2 | m3) {
            const float4 v4 = *(const float4*)(vis + vb);
            const double2 e0 = *(const double2*)(eff + op), e1 = *(const double2*)(eff + op + 2);
            o4.x = (float)(e0.x * ((double)v4.x * (m0 ? 1 : 0)) * npho);
            o4.y = (float)(e0.y * ((double)v4.y * (m1 ? 1 : 0)) * npho);
            o4.z = (float)(e1.x * ((double)v4.z * (m2 ? 1 : 0)) * npho);
            o4.w = (float)(e1.y * ((double)v4.w * (m3 ? 1 : 0)) * npho);
          }
          if (trig0) {
            const float4 l4 = *(const float4*)(t0lut + vb);
            float4 t4;
            t4.x = (float)(((double)l4.x * ns + t0s * mus) / mus);
            t4.y = (float)(((double)l4.y * ns + t0s * mus) / mus);
            t4.z = (float)(((double)l4.z * ns + t0s * mus) / mus);
            t4.w = (float)(((double)l4.w * ns + t0s * mus) / mus);
            *(float4*)(rowt + 4 * q) = t4;
          }
        } else if (trig0) {
          *(float4*)(rowt + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        *(float4*)(rowp + 4 * q) = o4;
      };
      quad(qa, ct0);
      quad(qb, ct1);
      quad(qc, ct2);
      quad(qd, ct3);
    }
  }
}

// min / max of t0_det over the entries with n_photons_det > 0 (light_sim.get_nticks, light_sim.py:34-39), as ordered ints
__device__ __forceinline__ int f2ord(float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__global__ void __launch_bounds__(256) light_t0_range_kernel(const float* __restrict__ nph, const float* __restrict__ t0det,
                                                             int64_t total, int* __restrict__ res) {
  int lo = 0x7fffffff, hi = (int)0x80000000, any = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
    if (nph[i] > 0) {
      const int v = f2ord(t0det[i]);
      lo = min(lo, v);
      hi = max(hi, v);
      any = 1;
    }
  for (int off = 32; off > 0; off >>= 1) {
    lo = min(lo, __shfl_xor(lo, off));
    hi = max(hi, __shfl_xor(hi, off));
    any |= __shfl_xor(any, off);
  }
  if ((threadIdx.x & 63) == 0 && any) {
    atomicMin(&res[0], lo);
    atomicMax(&res[1], hi);
    atomicOr(&res[2], 1);
  }
}

struct LightSum {
  SegStore s;
  const LdsimConsts* c;
  int64_t seg0, n;                // resident range; per-segment arrays below are relative to seg0
  const int32_t* voxel;           // [n][3]
  const float* nph;               // [n][n_inc]
  int n_inc;
  const int32_t* op_channel;      // [n_det]
  int n_det;
  const float* t0_avg;
  const float* time_dist;
  int ny, nz, ndet_lut, nprof;
  double start_time;
  int64_t n_ticks;
};

// every photon deposit of (segment r, detector row idet), in the reference's order (profile bin ascending);
// f(tick, profile bin, photons).  The tick is the one whose open window (start, end) holds the arrival time, tested
// with the reference's own expressions on the three candidates around floor((t - start) / tick) (light_sim.py:81-82,100,117)
template <class F>
__device__ __forceinline__ void light_deposits_ph(const LightSum& L, int64_t r, int opch, float ph, F f) {
  const LdsimConsts* c = L.c;
  if (!(ph > 0)) return;
  const double ns = 1.0, mus = 1e-6 * 1e9, tick = c->light_tick_size;
  const double track_time = L.s.f[LDSIM_T0][L.seg0 + r];
  const double track_end = track_time + L.nprof * ns / mus;
  const int32_t* vx = L.voxel + r * 3;
  const int64_t lb = ((((int64_t)vx[0] * L.ny + vx[1]) * L.nz + vx[2]) * L.ndet_lut + (opch % L.ndet_lut));
  auto at = [&](double pt, int ip, double photons) {
    const double fl = floor((pt - L.start_time) / tick);
    if (!(fl > -2.0 && fl < (double)L.n_ticks + 1.0)) return;
    const int64_t it0 = (int64_t)fl;
    for (int64_t it = it0 - 1; it <= it0 + 1; it++) {
      if (it < 0 || it >= L.n_ticks) continue;
      const double st = it * tick + L.start_time, en = st + tick;
      if (track_end < st || track_time > en) continue;
      if (pt < en && pt > st) f(it, ip, photons);
    }
  };
  if (c->enable_lut_smearing) {
    const float* prof = L.time_dist + lb * L.nprof;
    // photons = n_photons_det * time_profile[iprof] / LIGHT_TICK_SIZE: Numba types the f4 * f4 product f4 (light_sim.py:100)
    for (int ip = 0; ip < L.nprof; ip++) {
      const float pp = ph * prof[ip];
      at(track_time + ip * ns / mus, ip, (double)pp / tick);
    }
  } else {
    at(track_time + (double)L.t0_avg[lb] * ns / mus, 0, (double)ph / tick);
  }
}
// the same deposits with the f4 product `n_photons_det * time_profile[ip]` (or n_photons_det alone without LUT smearing) instead
// of photons = (double)product / LIGHT_TICK_SIZE: what a record carries in 32 bits
template <class F>
__device__ __forceinline__ void light_deposits_f4(const LightSum& L, int64_t r, int idet, F f) {
  const LdsimConsts* c = L.c;
  const int opch = L.op_channel[idet];
  const float ph = L.nph[r * L.n_inc + opch];
  if (c->enable_lut_smearing) {
    const int32_t* vx = L.voxel + r * 3;
    const int64_t lb = ((((int64_t)vx[0] * L.ny + vx[1]) * L.nz + vx[2]) * L.ndet_lut + (opch % L.ndet_lut));
    const float* prof = L.time_dist + lb * L.nprof;
    light_deposits_ph(L, r, opch, ph, [&](int64_t it, int ip, double) { f(it, ip, ph * prof[ip]); });
  } else {
    light_deposits_ph(L, r, opch, ph, [&](int64_t it, int ip, double) { f(it, ip, ph); });
  }
}
template <class F>
__device__ __forceinline__ void light_deposits(const LightSum& L, int64_t r, int idet, F f) {
  const int opch = L.op_channel[idet];
  light_deposits_ph(L, r, opch, L.nph[r * L.n_inc + opch], f);
}

// ---- without truth slots: scatter into an LDS tick tile ------------------------------------------------------------------
// The accumulation order differs from the reference's sorted loop (f64 tile, one f4 rounding at the end), so the f4
// result can differ from the reference's in the last bits.
#define LTILE 8192
__global__ void __launch_bounds__(256) sum_light_scatter_kernel(LightSum L, float* __restrict__ out) {
  __shared__ double acc[LTILE];
  const int idet = blockIdx.x;
  const int64_t tile0 = (int64_t)blockIdx.y * LTILE;
  const int tlen = (int)min((int64_t)LTILE, L.n_ticks - tile0);
  if (idet >= L.n_det || tlen <= 0) return;
  for (int i = threadIdx.x; i < tlen; i += 256) acc[i] = 0;
  __syncthreads();
  int touched = 0;
  for (int64_t r = threadIdx.x; r < L.n; r += 256)
    light_deposits(L, r, idet, [&](int64_t it, int, double photons) {
      if (it >= tile0 && it < tile0 + tlen && photons != 0.0) {
        atomicAdd(&acc[it - tile0], photons);
        touched = 1;
      }
    });
  if (!__syncthreads_or(touched)) return;      // nothing arrived in this tile: the output keeps what it holds
  for (int i = threadIdx.x; i < tlen; i += 256) {
    const int64_t o = (int64_t)idet * L.n_ticks + tile0 + i;
    if (acc[i] != 0.0) out[o] = (float)((double)out[o] + acc[i]);
  }
}

// ---- the same sum over a list of the (detector, tick tile) cells that receive light -----------------------------------------------------
// A batch of the driver's loop lights the channels of its own TPCs only (ndlar: 96 of 3360 rows), and the dense
// [n_det][n_ticks] array is 148 MB there: a grid over every (detector, tile) and a memset of the whole array per batch cost
// 50 us of which 2 are work.  light_active_kernel marks the tiles some deposit falls into (reads of the incidence rows
// coalesced along the channels, 16 segments in flight per lane) and appends each newly marked one to a list;
// sum_light_list_kernel is sum_light_scatter_kernel over that list (same accumulation: f64 LDS tile, one f4 rounding);
// light_clear_list_kernel zeroes the listed tiles again before the next batch's sum (ldsim_dev_sum_light).
#define LACT_SEGS 16
__device__ __forceinline__ void light_clear_entries(const unsigned* __restrict__ count, const int32_t* __restrict__ list, int ntile,
                                                    int64_t n_ticks, float* __restrict__ out, unsigned long long* __restrict__ dmask,
                                                    int first, int stride) {
  const int cnt = (int)*count;
  for (int e = first; e < cnt; e += stride) {
    const int code = list[e];
    const int idet = code / ntile;
    const int64_t tile0 = (int64_t)(code - idet * ntile) * LIGHT_TILE;
    const int tlen = (int)min((int64_t)LIGHT_TILE, n_ticks - tile0);
    float* o = out + (int64_t)idet * n_ticks + tile0;
    for (int i = threadIdx.x; i < tlen; i += 256) o[i] = 0.f;
    if (threadIdx.x == 0) dmask[idet] = 0ull;
  }
}

// rows blockIdx.y < ny_active: 64 detectors x 4 x LACT_SEGS segments per block, the tiles their deposits fall into marked and
// listed; the rows past them: the previous sum's listed tiles back to zero (they touch the array and the other half of the
// buffer only)
__global__ void __launch_bounds__(256) light_active_kernel(LightSum L, LightAct A, int ny_active, float* __restrict__ out) {
  if ((int)blockIdx.y >= ny_active) {
    if (A.clear)
      light_clear_entries(A.p_count, A.p_list, A.p_ntile, A.p_nticks, out, A.p_dmask,
                          ((int)blockIdx.y - ny_active) * (int)gridDim.x + (int)blockIdx.x, ((int)gridDim.y - ny_active) * (int)gridDim.x);
    return;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int idet = blockIdx.x * 64 + lane;
  const int64_t r0 = ((int64_t)blockIdx.y * 4 + wv) * LACT_SEGS;
  if (idet >= L.n_det || r0 >= L.n) return;
  const int opch = L.op_channel[idet];
  float ph[LACT_SEGS];
#pragma unroll
  for (int k = 0; k < LACT_SEGS; k++) ph[k] = (r0 + k < L.n) ? L.nph[(r0 + k) * L.n_inc + opch] : 0.f;
  unsigned long long mask = 0;
#pragma unroll
  for (int k = 0; k < LACT_SEGS; k++)
    light_deposits_ph(L, r0 + k, opch, ph[k], [&](int64_t it, int, double photons) {
      if (photons != 0.0) mask |= 1ull << (int)(it / LIGHT_TILE);
    });
  if (!mask) return;
  const unsigned long long old = atomicOr(&A.dmask[idet], mask);
  for (unsigned long long nb = mask & ~old; nb; nb &= nb - 1)
    A.list[atomicAdd(A.count, 1u)] = idet * A.ntile + (__ffsll((long long)nb) - 1);
}

__global__ void __launch_bounds__(256) sum_light_list_kernel(LightSum L, LightAct A, float* __restrict__ out) {
  __shared__ double acc[LIGHT_TILE];
  if (A.clear && blockIdx.x == 0 && threadIdx.x == 0) *A.p_count = 0u;      // (light_active_kernel's clearing rows are done with it)
  const int cnt = (int)*A.count, ntile = A.ntile;
  for (int e = blockIdx.x; e < cnt; e += gridDim.x) {
    const int code = A.list[e];
    const int idet = code / ntile;
    const int64_t tile0 = (int64_t)(code - idet * ntile) * LIGHT_TILE;
    const int tlen = (int)min((int64_t)LIGHT_TILE, L.n_ticks - tile0);
    for (int i = threadIdx.x; i < tlen; i += 256) acc[i] = 0;
    __syncthreads();
    for (int64_t r = threadIdx.x; r < L.n; r += 256)
      light_deposits(L, r, idet, [&](int64_t it, int, double photons) {
        if (it >= tile0 && it < tile0 + tlen && photons != 0.0) atomicAdd(&acc[it - tile0], photons);
      });
    __syncthreads();
    for (int i = threadIdx.x; i < tlen; i += 256) {
      const int64_t o = (int64_t)idet * L.n_ticks + tile0 + i;
      if (acc[i] != 0.0) out[o] = (float)((double)out[o] + acc[i]);
    }
    __syncthreads();
  }
}

// the same clearing on its own (a sum over an empty range launches nothing else)
__global__ void __launch_bounds__(256) light_clear_list_kernel(LightAct A, float* __restrict__ out) {
  light_clear_entries(A.p_count, A.p_list, A.p_ntile, A.p_nticks, out, A.p_dmask, (int)blockIdx.x, (int)gridDim.x);
}

// ---- with truth slots: records sorted into the reference's visiting order --------------------------------------------------------
#define LK_RANK_BITS 20
// resident path: order of cli/simulate_pixels.py:1141-1144, np.argsort(n_photons_det)[::-1] per detector: descending
// photons; equal photons (the reference's sort is not stable, their order is unpinned) by descending segment index
__global__ void light_order_keys_kernel(LightSum L, unsigned long long* __restrict__ keys, int32_t* __restrict__ vals) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (int64_t)L.n_det * L.n) return;
  const int64_t idet = p / L.n, r = p - idet * L.n;
  const float ph = L.nph[r * L.n_inc + L.op_channel[idet]];
  const unsigned int bits = ph > 0 ? ~(unsigned int)__float_as_int(ph) : 0xFFFFFFFFu;
  keys[p] = ((unsigned long long)idet << 52) | ((unsigned long long)bits << LK_RANK_BITS) |
            (unsigned long long)((~(unsigned int)r) & ((1u << LK_RANK_BITS) - 1));
  vals[p] = (int32_t)r;
}
// ---- round 4: records emitted in visiting order, 32-bit keys, the payload travelling with the key ---------------------------------
// order[idet][q] = the segment visited q-th for detector idet (the first sort's values, or the caller's sorted_indices); p runs
// over (idet, q), so the scan of the counts lays the records out detector after detector, segment after segment in visiting order
__global__ void light_count_ordered_kernel(LightSum L, const int32_t* __restrict__ order, int32_t* __restrict__ count) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= L.n * L.n_det) return;
  const int idet = (int)(p / L.n);
  const int64_t r = order[p];
  int cnt = 0;
  if (r >= 0 && r < L.n) light_deposits(L, r, idet, [&](int64_t, int, double) { cnt++; });
  count[p] = cnt;
}
__global__ void light_fill_ordered_kernel(LightSum L, const int32_t* __restrict__ order, const int32_t* __restrict__ offs, int tick_bits,
                                          unsigned* __restrict__ keys, unsigned long long* __restrict__ vals) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= L.n * L.n_det) return;
  const int idet = (int)(p / L.n);
  const int64_t r = order[p];
  if (r < 0 || r >= L.n) return;
  int64_t w = offs[p];
  light_deposits_f4(L, r, idet, [&](int64_t it, int, float pp) {
    keys[w] = ((unsigned)idet << tick_bits) | (unsigned)it;
    vals[w] = ((unsigned long long)(unsigned)r << 32) | (unsigned long long)(unsigned)__float_as_int(pp);
    w++;
  });
}

// ---- round 4, resident sum: only the (detector, segment) pairs that carry photons are ordered, and a wave emits a pair's records ----
// A segment lights the detectors of its own TPC: 1 in 8 (2x2) to 1 in 70 (ndlar) of the n_det x n pairs hold photons.  The pairs
// that do are compacted (any order: the sort that follows orders them, their keys are distinct) ...
#define LIGHT_SPARE 2                 // record slots of a pair beyond its profile bins (a bin at a window edge can deposit into two ticks: ~1e-10 per bin)
#define LIGHT_NONE 0xFFFFFFFFu        // key of an unused record slot: sorts behind every (detector, tick) cell (those use <= 31 bits)
__global__ void __launch_bounds__(256) light_active_pairs_kernel(LightSum L, unsigned long long* __restrict__ keys,
                                                                int32_t* __restrict__ vals, unsigned* __restrict__ count) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  bool on = false;
  unsigned long long key = 0;
  if (p < (int64_t)L.n_det * L.n) {
    const int64_t idet = p / L.n, r = p - idet * L.n;
    const float ph = L.nph[r * L.n_inc + L.op_channel[idet]];
    on = ph > 0;
    key = ((unsigned long long)idet << 52) | ((unsigned long long)(~(unsigned int)__float_as_int(ph)) << LK_RANK_BITS) |
          (unsigned long long)((~(unsigned int)r) & ((1u << LK_RANK_BITS) - 1));
  }
  const unsigned long long m = __ballot(on);
  if (!m) return;
  unsigned base = 0;
  if (lane == 0) base = atomicAdd(count, (unsigned)__popcll(m));
  base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
  if (on) {
    const unsigned w = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
    keys[w] = key;
    vals[w] = (int32_t)p;
  }
}
// ... and each, in visiting order q, gets `cap` record slots from q * cap on, filled by a wave: lane ip takes profile bin ip, the
// bins that deposit (light_deposits_ph: the tick whose open window holds the arrival time -- at a window edge the reference's
// expressions can admit two ticks, both are emitted, tick ascending) are ranked by ballot and written side by side; the slots
// left over carry LIGHT_NONE.  No count pass, no scan, no read-back of the record count, and the stores are coalesced (a thread per
// pair wrote its ~100 records alone: 0.63 ms per 2x2 batch).  cap = bins + LIGHT_SPARE; a pair that would need more (three window-edge
// coincidences in one pair: ~1e-24) raises *overflow and the sum fails loudly.
__global__ void __launch_bounds__(256) light_emit_wave_kernel(LightSum L, const int32_t* __restrict__ order, int64_t n_act, int cap,
                                                             int tick_bits, unsigned* __restrict__ keys,
                                                             unsigned long long* __restrict__ vals, unsigned* __restrict__ overflow) {
  const int lane = threadIdx.x & 63;
  const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= n_act) return;
  const LdsimConsts* c = L.c;
  const int64_t p = order[q];
  const int idet = (int)(p / L.n);
  const int64_t r = p - (int64_t)idet * L.n;
  const int opch = L.op_channel[idet];
  const float ph = L.nph[r * L.n_inc + opch];
  const double ns = 1.0, mus = 1e-6 * 1e9, tick = c->light_tick_size;
  const double track_time = L.s.f[LDSIM_T0][L.seg0 + r];
  const double track_end = track_time + L.nprof * ns / mus;
  const int32_t* vx = L.voxel + r * 3;
  const int64_t lb = ((((int64_t)vx[0] * L.ny + vx[1]) * L.nz + vx[2]) * L.ndet_lut + (opch % L.ndet_lut));
  const bool smear = c->enable_lut_smearing != 0;
  const int nbin = smear ? L.nprof : 1;
  const float* prof = L.time_dist + lb * L.nprof;
  const int64_t base = q * (int64_t)cap;
  const unsigned long long lt = (1ull << lane) - 1ull;
  int n_out = 0;
  for (int ip0 = 0; ip0 < nbin; ip0 += 64) {
    const int ip = ip0 + lane;
    bool e[3] = {false, false, false};
    int64_t it0 = 0;
    float pp = 0.f;
    if (ip < nbin) {
      const double pt = smear ? track_time + ip * ns / mus : track_time + (double)L.t0_avg[lb] * ns / mus;
      pp = smear ? ph * prof[ip] : ph;              // (the f4 product of light_sim.py:100)
      const double fl = floor((pt - L.start_time) / tick);
      if (fl > -2.0 && fl < (double)L.n_ticks + 1.0) {
        it0 = (int64_t)fl;
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const int64_t it = it0 - 1 + k;
          if (it < 0 || it >= L.n_ticks) continue;
          const double st = it * tick + L.start_time, en = st + tick;
          if (track_end < st || track_time > en) continue;
          e[k] = pt < en && pt > st;
        }
      }
    }
    const unsigned long long m0 = __ballot(e[0]), m1 = __ballot(e[1]), m2 = __ballot(e[2]);
    int w = n_out + __popcll(m0 & lt) + __popcll(m1 & lt) + __popcll(m2 & lt);
#pragma unroll
    for (int k = 0; k < 3; k++) {
      if (e[k]) {
        if (w < cap) {
          keys[base + w] = ((unsigned)idet << tick_bits) | (unsigned)(it0 - 1 + k);
          vals[base + w] = ((unsigned long long)(unsigned)r << 32) | (unsigned long long)(unsigned)__float_as_int(pp);
        }
        w++;
      }
    }
    n_out += __popcll(m0) + __popcll(m1) + __popcll(m2);
  }
  if (n_out > cap) {
    if (lane == 0) atomicOr(overflow, 1u);
    n_out = cap;
  }
  for (int k = n_out + lane; k < cap; k += 64) {
    keys[base + k] = LIGHT_NONE;
    vals[base + k] = 0ull;
  }
}

// one thread per (detector, tick) cell that received something: replay its records in order (light_sim.py:101-110,118-127)
__global__ void light_replay_kernel(const unsigned* __restrict__ keys, const unsigned long long* __restrict__ vals, int64_t n_rec,
                                    int tick_bits, double tick_size, const int64_t* __restrict__ track_id, int64_t n_ticks,
                                    double truth_threshold, float* __restrict__ out, int64_t* __restrict__ true_id,
                                    double* __restrict__ true_ph, int max_truth) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rec) return;
  const unsigned cell = keys[i];
  if (i > 0 && keys[i - 1] == cell) return;
  const int64_t idet = (int64_t)(cell >> tick_bits), it = (int64_t)(cell & ((1u << tick_bits) - 1));
  const int64_t o = idet * n_ticks + it;
  float acc = out[o];
  for (int64_t j = i; j < n_rec && keys[j] == cell; j++) {
    const unsigned long long v = vals[j];
    const double photons = (double)__int_as_float((int)(unsigned)v) / tick_size;
    acc = (float)((double)acc + photons);
    if (photons > truth_threshold) {
      const int64_t id = track_id[(int64_t)(v >> 32)];
      for (int k = 0; k < max_truth; k++) {
        int64_t* tid = &true_id[o * max_truth + k];
        if (*tid == -1 || *tid == id) {
          *tid = id;
          true_ph[o * max_truth + k] += photons;
          break;
        }
      }
    }
  }
  out[o] = acc;
}


// The cells the last truth-slot photon sum wrote (the heads of its sorted records, still in light_tmp[4]) back to their initial
// values: the next sum on the same buffers starts from arrays that are clean everywhere else (ldsim_dev_sum_light) -- at 50 truth
// slots and 50 000 ticks clearing 15 GB per batch was half the photon sum's time, the cells written are 3 % of them.
__global__ void __launch_bounds__(256) light_reset_cells_kernel(const unsigned* __restrict__ keys, int64_t n_rec, int tick_bits,
                                                               int64_t n_ticks, int max_truth, float* __restrict__ out,
                                                               int64_t* __restrict__ true_id, double* __restrict__ true_ph) {
  // a workgroup per 256 records: the heads among them (first record of a cell) are collected, then each wave clears whole cells,
  // lane k slot k
  __shared__ unsigned s_cell[256];
  __shared__ int s_n;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t i = (int64_t)blockIdx.x * 256 + tid;
  if (tid == 0) s_n = 0;
  __syncthreads();
  unsigned cell = 0;
  bool head = false;
  if (i < n_rec) {
    cell = keys[i];
    head = (i == 0 || keys[i - 1] != cell) && cell != 0xFFFFFFFFu;      // (an unused record slot of the compact form: no cell)
  }
  const unsigned long long m = __ballot(head);
  int base = 0;
  if (lane == 0 && m) base = atomicAdd(&s_n, __popcll(m));
  base = __builtin_amdgcn_readfirstlane(base);
  if (head) s_cell[base + __popcll(m & ((1ull << lane) - 1ull))] = cell;
  __syncthreads();
  const int nh = s_n;
  for (int h = wv; h < nh; h += 4) {
    const unsigned c = s_cell[h];
    const int64_t o = (int64_t)(c >> tick_bits) * n_ticks + (int64_t)(c & ((1u << tick_bits) - 1));
    if (lane == 0) out[o] = 0.f;
    for (int k = lane; k < max_truth; k += 64) {
      true_id[o * max_truth + k] = -1;
      true_ph[o * max_truth + k] = 0.0;
    }
  }
}

// The same replay with a wave per block of RW_BLOCK sorted records (max_truth <= 64): the wave owns the (detector, tick) cells
// whose first record lies in its block and walks each to its end.  The records are fetched 64 at a time, one per lane (key and
// payload side by side, then the track id of the payload's segment), and replayed in order from registers; lane k keeps truth
// slot k of the cell, so "the first slot that is empty or holds this id" (light_sim.py:120-126) is one ballot and the row is
// read and written once, coalesced.  One thread per cell walked its run alone with a global-memory slot search per record:
// a busy tick of a busy detector (thousands of records) set the kernel's time.
#define RW_BLOCK 256
// Round 4, second form: 2.4e7 records per 2x2 batch make this kernel a matter of instructions per record (the walk is sequential by
// the reference's definition: an f4 sum rounded after every addend, slots given out first come first served).  Record by record
// through v_readlane it cost ~35 instructions each (1.65 ms per batch, 37 % of the 2x2 + light step).  Now a cell's records are
// taken 64 at a time, one per lane: which slot a record's track holds is looked up by all lanes at once against the slots filled
// so far (ids mirrored in LDS), tracks seen for the first time get the next slots in order of first appearance (one ballot round
// per NEW track: at most max_truth per cell), and only the two chains that must be sequential stay in the per-record loop --
// the f4 sum and "slot s += photons" on the lane that owns slot s -- fed from LDS broadcasts, 9 instructions per record.
// A run with a track id of -1 above the threshold (an id that looks like an empty slot: it is overwritten by the next track)
// or a sum that accumulates into the caller's arrays (fresh = 0: slots may be anything) is walked record by record as before.
__global__ void __launch_bounds__(64) light_replay_wave_kernel(const unsigned* __restrict__ keys,
                                                               const unsigned long long* __restrict__ vals, int64_t n_rec,
                                                               int tick_bits, double tick_size,
                                                               const int64_t* __restrict__ track_id, int64_t n_ticks,
                                                               double truth_threshold, float* __restrict__ out,
                                                               int64_t* __restrict__ true_id, double* __restrict__ true_ph,
                                                               int max_truth, int fresh) {
  const int lane = threadIdx.x;
  const int64_t b0 = (int64_t)blockIdx.x * RW_BLOCK, b1 = min(b0 + RW_BLOCK, n_rec);
  __shared__ double s_pt[64];
  __shared__ long long s_sid[64];
  __shared__ unsigned char s_slot[64];
  auto rl64 = [](unsigned long long v, int l) {
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) |
           (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)v, l);
  };
  bool open = false;                 // a cell of this wave is being replayed
  int64_t o = 0;
  float acc = 0.f;
  int64_t slot_id = -1;              // lane k: truth slot k of the open cell
  double slot_ph = 0.0;
  int n_filled = 0;                  // slots 0 .. n_filled - 1 hold a track (fresh sums: slots fill from the front)
  auto close_cell = [&]() {
    if (lane == 0) out[o] = acc;
    if (lane < max_truth) {
      true_id[o * max_truth + lane] = slot_id;
      true_ph[o * max_truth + lane] = slot_ph;
    }
  };
  unsigned prev_cell = b0 > 0 ? keys[b0 - 1] : 0xFFFFFFFFu;       // (cells use at most 28 bits)
  for (int64_t c0 = b0; c0 < n_rec; c0 += 64) {
    if (c0 >= b1 && !open) break;                 // past the block and nothing left to finish
    const int64_t j = c0 + lane;
    unsigned cell = 0xFFFFFFFFu;
    double ph = 0.0;
    int64_t id = -1;
    if (j < n_rec) {
      cell = keys[j];
      const unsigned long long v = vals[j];
      ph = (double)__int_as_float((int)(unsigned)v) / tick_size;        // photons of light_sim.py:100
      id = track_id[(int64_t)(v >> 32)];
    }
    const int nv = (int)min((int64_t)64, n_rec - c0);
    bool done = false;
    int ta = 0;
    while (ta < nv && !done) {
      // ---- the run [ta, tb) of records of one cell ----------------------------------------------------------------------------------------
      const unsigned ct = (unsigned)__builtin_amdgcn_readlane((int)cell, ta);
      const unsigned long long differ = __ballot(cell != ct) & ~((2ull << ta) - 1ull);      // lanes past ta of another cell
      const int tb = differ ? __ffsll((long long)differ) - 1 : 64;         // (lanes >= nv carry the sentinel cell)
      const bool head = ct != prev_cell;
      prev_cell = ct;
      if (head) {
        if (open) { close_cell(); open = false; }
        if (c0 + ta >= b1 || ct == 0xFFFFFFFFu) { done = true; break; }          // the next block's cell, or the unused slots at the end
        open = true;
        // fresh (the resident sum, whose arrays the call initialises): the cell starts from 0 photons and empty slots, nothing is
        // read back; otherwise (the host-array entry accumulates into what the caller passed, like the reference's +=) the rows are read
        o = (int64_t)(ct >> tick_bits) * n_ticks + (int64_t)(ct & ((1u << tick_bits) - 1));
        acc = fresh ? 0.f : out[o];
        slot_id = lane < max_truth ? (fresh ? -1 : true_id[o * max_truth + lane]) : -2;
        slot_ph = lane < max_truth && !fresh ? true_ph[o * max_truth + lane] : 0.0;
        n_filled = 0;
      }
      if (!open) { ta = tb; continue; }                      // the tail of a cell that began in an earlier block
      const bool in_run = lane >= ta && lane < tb;
      const bool valid = in_run && ph > truth_threshold;
      if (!fresh || __ballot(valid && id == -1)) {
        // ---- record by record (light_sim.py:101-110,118-127 as written) ----------------------------------------------------------------------
        for (int t = ta; t < tb; t++) {
          const double pt = __longlong_as_double((long long)rl64((unsigned long long)__double_as_longlong(ph), t));
          acc = (float)((double)acc + pt);
          if (pt > truth_threshold) {
            const int64_t it = (int64_t)rl64((unsigned long long)id, t);
            const unsigned long long m = __ballot(lane < max_truth && (slot_id == -1 || slot_id == it));
            if (m && lane == __ffsll((long long)m) - 1) {
              slot_id = it;
              slot_ph += pt;
            }
          }
        }
        n_filled = __popcll(__ballot(lane < max_truth && slot_id != -1));
        s_sid[lane] = slot_id;
        ta = tb;
        continue;
      }
      // ---- slot of each record: the ones filled so far, then new tracks in order of first appearance ---------------------------------------
      int slot = 0xFF;
      for (int k = 0; k < n_filled; k++)
        if (s_sid[k] == id) slot = k;
      slot = valid ? slot : 0xFF;
      unsigned long long fresh_ids = __ballot(valid && slot == 0xFF);
      while (fresh_ids) {
        const int t1 = __ffsll((long long)fresh_ids) - 1;
        const int64_t idn = (int64_t)rl64((unsigned long long)id, t1);
        const bool same = valid && slot == 0xFF && id == idn;
        if (n_filled < max_truth) {
          if (same) slot = n_filled;
          if (lane == n_filled) slot_id = idn;
          if (lane == 0) s_sid[n_filled] = idn;
          n_filled++;
          fresh_ids &= ~__ballot(same);
        } else {
          fresh_ids = 0;                      // every slot holds another track: these records leave no trace in the slots
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (in_run) {
        s_pt[lane] = ph;
        s_slot[lane] = (unsigned char)slot;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // ---- the two sequential chains ---------------------------------------------------------------------------------------------------------
      int t = ta;
      for (; t + 8 <= tb; t += 8) {
        double p[8];
        int sl[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { p[u] = s_pt[t + u]; sl[u] = s_slot[t + u]; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          acc = (float)((double)acc + p[u]);
          slot_ph += sl[u] == lane ? p[u] : 0.0;
        }
      }
      for (; t < tb; t++) {
        const double p = s_pt[t];
        acc = (float)((double)acc + p);
        slot_ph += s_slot[t] == lane ? p : 0.0;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      ta = tb;
    }
    if (done) break;
  }
  if (open) close_cell();
}

extern "C++" {
int light_launch_reset_cells(ldsim_ctx* ctx, int64_t n_rec, int64_t n_ticks, int max_truth, float* out, int64_t* true_id,
                              double* true_ph) {
  if (n_rec <= 0) return 0;
  hipLaunchKernelGGL(light_reset_cells_kernel, dim3((unsigned)((n_rec + 255) / 256)), dim3(256), 0, ctx->stream,
                     (const unsigned*)ctx->light_tmp[4].p, n_rec, ctx->light_lazy_tick_bits, n_ticks, max_truth, out, true_id, true_ph);
  HIPCHK(hipGetLastError());
  return 0;
}
// the record-slot overflow flag of the last compact photon sum (light_emit_wave_kernel); the caller has synchronised the stream
int light_check_emit_overflow(ldsim_ctx* ctx) {
  if (!ctx->light_emit_flag) return 0;
  unsigned f = 0;
  HIPCHK(hipMemcpy(&f, ctx->light_emit_flag, 4, hipMemcpyDeviceToHost));
  ctx->light_emit_flag = nullptr;
  if (f) {
    HIPCHK(hipMemset(ctx->light_flag_dev, 0, 4));
    ldsim_set_error("photon sum with truth slots: a (detector, segment) pair deposited into more ticks than its record slots hold "
                    "(profile bins + 2); the arrays of that sum are incomplete");
    return LDSIM_ESTATE;
  }
  return 0;
}
int light_launch_clear_list(ldsim_ctx* ctx, const LightAct* act, float* out) {
  hipLaunchKernelGGL(light_clear_list_kernel, dim3(256), dim3(256), 0, ctx->stream, *act, out);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemsetAsync(act->p_count, 0, 4, ctx->stream));
  return 0;
}

int sort_pairs(ldsim_ctx*, unsigned long long*, unsigned long long*, int32_t*, int32_t*, int64_t);
int sort_pairs_bits(ldsim_ctx*, unsigned long long*, unsigned long long*, int32_t*, int32_t*, int64_t, int, int);
int sort_pairs_u32_u64(ldsim_ctx*, unsigned*, unsigned*, unsigned long long*, unsigned long long*, int64_t, int);
int sort_exclusive_scan_i32(ldsim_ctx*, const int32_t*, int32_t*, int64_t);
static inline int nblk(int64_t n, int b) { return (int)((n + b - 1) / b); }

int light_launch_incidence(ldsim_ctx* ctx, int64_t seg0, int64_t n, int n_out, float* nph, float* t0det, int32_t* voxel,
                           int fill) {
  if (n == 0 || n_out == 0) return 0;
  if (n_out % 4 == 0 && ctx->lut_ndet % 4 == 0 && ctx->light_eff_plain && !ctx->light_incidence_scalar) {
    const dim3 grid((unsigned)((n + LI_SEGS - 1) / LI_SEGS));
    if (n_out < ctx->h_consts.n_op_channel)
      hipLaunchKernelGGL(light_incidence4_kernel<false>, grid, dim3(256), 0, ctx->stream, ctx->seg, ctx->d_consts, seg0, n,
                         ctx->d_lut_vis, ctx->d_lut_t0, ctx->lut_nx, ctx->lut_ny, ctx->lut_nz, ctx->lut_ndet, ctx->d_eff,
                         ctx->d_ch2tpc, n_out, nph, t0det, voxel, fill);
    else
      hipLaunchKernelGGL(light_incidence4_kernel<true>, grid, dim3(256), 0, ctx->stream, ctx->seg, ctx->d_consts, seg0, n,
                         ctx->d_lut_vis, ctx->d_lut_t0, ctx->lut_nx, ctx->lut_ny, ctx->lut_nz, ctx->lut_ndet, ctx->d_eff,
                         ctx->d_ch2tpc, n_out, nph, t0det, voxel, fill);
    HIPCHK(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(light_incidence_kernel, dim3((unsigned)((n + LI_SEGS - 1) / LI_SEGS)), dim3(256), 0, ctx->stream,
                     ctx->seg, ctx->d_consts, seg0, n, ctx->d_lut_vis, ctx->d_lut_t0, ctx->lut_nx, ctx->lut_ny,
                     ctx->lut_nz, ctx->lut_ndet, ctx->d_eff, ctx->d_ch2tpc, n_out, nph, t0det, voxel, fill);
  HIPCHK(hipGetLastError());
  return 0;
}

int light_launch_t0_range(ldsim_ctx* ctx, const float* nph, const float* t0det, int64_t total, int* d_res) {
  const int init[3] = {0x7fffffff, (int)0x80000000, 0};
  HIPCHK(hipMemcpyAsync(d_res, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
  if (total > 0) {
    hipLaunchKernelGGL(light_t0_range_kernel, dim3((unsigned)min((int64_t)2048, (total + 255) / 256)), dim3(256), 0,
                       ctx->stream, nph, t0det, total, d_res);
    HIPCHK(hipGetLastError());
  }
  return 0;
}

// Photon sum over resident segments [seg0, seg0 + n).  voxel / nph / track_id are relative to seg0.  sorted_idx (device,
// [n_det][n]) = the caller's visiting order, or NULL = descending photons per detector.  Uses ctx->light_tmp[].
int light_launch_sum(ldsim_ctx* ctx, int64_t seg0, int64_t n, const int32_t* voxel, const int64_t* track_id,
                     const float* nph, int n_inc, const int32_t* op_channel, int n_det, const int32_t* sorted_idx,
                     double start_time, int64_t n_ticks, float* out, int64_t* true_id, double* true_ph, int max_truth,
                     int64_t* n_rec_out /* truth path: records left sorted in light_tmp[4] (their cells are what was written) */,
                     const LightAct* act /* no truth slots: sum over a device-built list of the lit (detector, tile) cells */) {
  if (n_rec_out) *n_rec_out = 0;
  if (n_det == 0 || n_ticks == 0 || n == 0) {
    if (max_truth == 0 && act && act->clear) return light_launch_clear_list(ctx, act, out);
    return 0;
  }
  LightSum L;
  L.s = ctx->seg; L.c = ctx->d_consts; L.seg0 = seg0; L.n = n; L.voxel = voxel; L.nph = nph; L.n_inc = n_inc;
  L.op_channel = op_channel; L.n_det = n_det; L.t0_avg = ctx->d_lut_t0avg; L.time_dist = ctx->d_lut_td;
  L.ny = ctx->lut_ny; L.nz = ctx->lut_nz; L.ndet_lut = ctx->lut_ndet; L.nprof = ctx->lut_nprof;
  L.start_time = start_time; L.n_ticks = n_ticks;
  hipStream_t st = ctx->stream;
  if (max_truth == 0 && act) {
    const unsigned gx = (unsigned)nblk(n_det, 64), gy = (unsigned)nblk(n, 4 * LACT_SEGS);
    const unsigned gclear = act->clear ? (255u + gx) / gx : 0u;
    hipLaunchKernelGGL(light_active_kernel, dim3(gx, gy + gclear), dim3(256), 0, st, L, *act, (int)gy, out);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(sum_light_list_kernel, dim3(512), dim3(256), 0, st, L, *act, out);
    HIPCHK(hipGetLastError());
    return 0;
  }
  if (max_truth == 0) {
    hipLaunchKernelGGL(sum_light_scatter_kernel, dim3(n_det, (unsigned)((n_ticks + LTILE - 1) / LTILE)), dim3(256), 0, st,
                       L, out);
    HIPCHK(hipGetLastError());
    return 0;
  }
  auto bits_for = [](int64_t v) { int b = 1; while ((1ll << b) < v) b++; return b; };
  const int tick_bits = bits_for(n_ticks), det_bits = bits_for(n_det);
  if (tick_bits + det_bits > 32 || n > (1 << LK_RANK_BITS) || (int64_t)n_det * n >= 0x7fffffffLL) {
    ldsim_set_error("photon sum with truth slots: (detector, tick) must fit 32 bits and a call 2^20 segments (got %d detectors, "
                    "%lld ticks, %lld segments)", n_det, (long long)n_ticks, (long long)n);
    return LDSIM_EINVAL;
  }
  const int64_t np = (int64_t)n_det * n;
  DevBuf* T = ctx->light_tmp;
  int rc;
  const double tick_size = ctx->h_consts.light_tick_size;
  const int tick_bits_c = bits_for(n_ticks + 1);        // compact form: the all-ones tick is no tick, it marks the unused record slots
  if (!sorted_idx && max_truth <= 64 && tick_bits_c + det_bits <= 31 && n_rec_out) {
    const int tick_bits = tick_bits_c;
    // ---- resident sum, compact form (kernels above): pairs with photons -> visiting order -> records by wave -> cells -> replay ----
    const int cap = (ctx->h_consts.enable_lut_smearing ? ctx->lut_nprof : 1) + LIGHT_SPARE;
    if ((rc = ldsim_ensure_buf(ctx, &T[1], 64))) return rc;                        // [0] pairs with photons
    if (!ctx->light_flag_dev) {                                                       // record slot overflow: sticky until read
      HIPCHK(hipMalloc((void**)&ctx->light_flag_dev, 8));
      HIPCHK(hipMemsetAsync(ctx->light_flag_dev, 0, 8, st));
    }
    if ((rc = ldsim_ensure_buf(ctx, &T[3], (size_t)np * 8))) return rc;
    if ((rc = ldsim_ensure_buf(ctx, &T[0], (size_t)np * 8))) return rc;
    if ((rc = ldsim_ensure_buf(ctx, &T[5], (size_t)np * 4))) return rc;
    if ((rc = ldsim_ensure_buf(ctx, &T[6], (size_t)np * 4))) return rc;
    unsigned* d_cnt = (unsigned*)T[1].p;
    HIPCHK(hipMemsetAsync(d_cnt, 0, 4, st));
    hipLaunchKernelGGL(light_active_pairs_kernel, dim3(nblk(np, 256)), dim3(256), 0, st, L, (unsigned long long*)T[3].p, (int32_t*)T[5].p,
                       d_cnt);
    HIPCHK(hipGetLastError());
    unsigned n_act_u = 0;
    HIPCHK(hipMemcpyAsync(&n_act_u, d_cnt, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if ((rc = light_check_emit_overflow(ctx))) return rc;            // (of the previous compact sum, if nobody looked since)
    const int64_t n_act = n_act_u, n_slots = n_act * cap;
    if (n_act == 0) return 0;
    if (n_slots >= 0x7fffffffLL) {
      ldsim_set_error("photon sum with truth slots: %lld (detector, segment) pairs with photons x %d record slots exceed 2^31; sum "
                      "fewer segments per call", (long long)n_act, cap);
      return LDSIM_EINVAL;
    }
    if ((rc = sort_pairs_bits(ctx, (unsigned long long*)T[3].p, (unsigned long long*)T[0].p, (int32_t*)T[5].p, (int32_t*)T[6].p, n_act, 0,
                              52 + det_bits)))
      return rc;
    // records: T[3] keys in, T[4] keys sorted (kept for the next sum's lazy reset), T[7] / T[8] payloads in / sorted
    if ((rc = ldsim_ensure_buf(ctx, &T[3], (size_t)n_slots * 4))) return rc;
    if ((rc = ldsim_ensure_buf(ctx, &T[4], (size_t)n_slots * 4))) return rc;
    if ((rc = ldsim_ensure_buf(ctx, &T[7], (size_t)n_slots * 8))) return rc;
    if ((rc = ldsim_ensure_buf(ctx, &T[8], (size_t)n_slots * 8))) return rc;
    unsigned *k0 = (unsigned*)T[3].p, *k1 = (unsigned*)T[4].p;
    unsigned long long *v0 = (unsigned long long*)T[7].p, *v1 = (unsigned long long*)T[8].p;
    hipLaunchKernelGGL(light_emit_wave_kernel, dim3(nblk(n_act, 4)), dim3(256), 0, st, L, (const int32_t*)T[6].p, n_act, cap, tick_bits,
                       k0, v0, ctx->light_flag_dev);
    HIPCHK(hipGetLastError());
    // The records leave the emit kernel detector after detector: a STABLE sort on the tick bits alone puts a (detector, tick) cell's
    // records side by side -- tick-major, detectors ascending inside a tick -- in their order of emission, in two radix passes for up
    // to 65 535 ticks instead of the three or four over (detector, tick); the unused slots (tick bits all ones) gather at the end.
    if ((rc = sort_pairs_u32_u64(ctx, k0, k1, v0, v1, n_slots, tick_bits))) return rc;
    ctx->light_lazy_tick_bits = tick_bits;
    hipLaunchKernelGGL(light_replay_wave_kernel, dim3(nblk(n_slots, RW_BLOCK)), dim3(64), 0, st, k1, v1, n_slots, tick_bits, tick_size,
                       track_id, n_ticks, ctx->h_consts.mc_truth_threshold, out, true_id, true_ph, max_truth, 1);
    HIPCHK(hipGetLastError());
    *n_rec_out = n_slots;
    ctx->light_emit_flag = ctx->light_flag_dev;  // (read at the next synchronising call: light_check_emit_overflow)
    return 0;
  }
  if ((rc = ldsim_ensure_buf(ctx, &T[1], (size_t)np * 4 + 16))) return rc;     // count, then offsets in T[2]
  if ((rc = ldsim_ensure_buf(ctx, &T[2], (size_t)np * 4 + 16))) return rc;
  int32_t* d_count = (int32_t*)T[1].p;
  int32_t* d_offs = (int32_t*)T[2].p;
  const int32_t* d_order = sorted_idx;            // [n_det][n]: the segment visited q-th for detector idet
  if (!sorted_idx) {
    // descending photons per detector (cli/simulate_pixels.py:1141-1144): one sort of the (detector, segment) pairs over the key bits
    // that are used -- detector | inverted photon bits | inverted segment index
    if ((rc = ldsim_ensure_buf(ctx, &T[3], (size_t)np * 8))) return rc;
    if ((rc = ldsim_ensure_buf(ctx, &T[0], (size_t)np * 8))) return rc;
    if ((rc = ldsim_ensure_buf(ctx, &T[5], (size_t)np * 4))) return rc;
    if ((rc = ldsim_ensure_buf(ctx, &T[6], (size_t)np * 4))) return rc;
    hipLaunchKernelGGL(light_order_keys_kernel, dim3(nblk(np, 256)), dim3(256), 0, st, L, (unsigned long long*)T[3].p,
                       (int32_t*)T[5].p);
    HIPCHK(hipGetLastError());
    if ((rc = sort_pairs_bits(ctx, (unsigned long long*)T[3].p, (unsigned long long*)T[0].p, (int32_t*)T[5].p, (int32_t*)T[6].p, np, 0,
                              52 + det_bits)))
      return rc;
    d_order = (const int32_t*)T[6].p;
  }
  hipLaunchKernelGGL(light_count_ordered_kernel, dim3(nblk(np, 256)), dim3(256), 0, st, L, d_order, d_count);
  HIPCHK(hipGetLastError());
  if ((rc = sort_exclusive_scan_i32(ctx, d_count, d_offs, np))) return rc;
  int32_t last_off = 0, last_cnt = 0;
  HIPCHK(hipMemcpyAsync(&last_off, d_offs + (np - 1), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&last_cnt, d_count + (np - 1), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  const int64_t n_rec = (int64_t)last_off + last_cnt;
  if (n_rec == 0) return 0;
  if (n_rec_out) *n_rec_out = n_rec;
  // records: T[3] keys in, T[4] keys sorted (kept for the next sum's lazy reset), T[7] / T[8] payloads in / sorted
  if ((rc = ldsim_ensure_buf(ctx, &T[3], (size_t)n_rec * 4))) return rc;
  if ((rc = ldsim_ensure_buf(ctx, &T[4], (size_t)n_rec * 4))) return rc;
  if ((rc = ldsim_ensure_buf(ctx, &T[7], (size_t)n_rec * 8))) return rc;
  if ((rc = ldsim_ensure_buf(ctx, &T[8], (size_t)n_rec * 8))) return rc;
  unsigned *k0 = (unsigned*)T[3].p, *k1 = (unsigned*)T[4].p;
  unsigned long long *v0 = (unsigned long long*)T[7].p, *v1 = (unsigned long long*)T[8].p;
  hipLaunchKernelGGL(light_fill_ordered_kernel, dim3(nblk(np, 256)), dim3(256), 0, st, L, d_order, d_offs, tick_bits, k0, v0);
  HIPCHK(hipGetLastError());
  if ((rc = sort_pairs_u32_u64(ctx, k0, k1, v0, v1, n_rec, tick_bits + det_bits))) return rc;
  ctx->light_lazy_tick_bits = tick_bits;
  if (max_truth <= 64)
    hipLaunchKernelGGL(light_replay_wave_kernel, dim3(nblk(n_rec, RW_BLOCK)), dim3(64), 0, st, k1, v1, n_rec, tick_bits, tick_size,
                       track_id, n_ticks, ctx->h_consts.mc_truth_threshold, out, true_id, true_ph, max_truth,
                       n_rec_out != nullptr /* the resident sum: arrays initialised by the call */);
  else
    hipLaunchKernelGGL(light_replay_kernel, dim3(nblk(n_rec, 256)), dim3(256), 0, st, k1, v1, n_rec, tick_bits, tick_size, track_id,
                       n_ticks, ctx->h_consts.mc_truth_threshold, out, true_id, true_ph, max_truth);
  HIPCHK(hipGetLastError());
  return 0;
}
}
