// kernels_seg.hip -- per-segment stages: record unpack/repack, quench, drift, pixel walk, time intervals.
// All of these are HBM-bound (184 B of compulsory traffic per segment); one thread per segment over
// SoA columns so every load/store is a coalesced 8-byte stream.
#include "ldsim_dev.h"

// ---- AoS <-> SoA ------------------------------------------------------------------------------------------
__device__ __forceinline__ double load_field(const unsigned char* rec, int off, int code) {
  if (off < 0) return 0.0;
  const unsigned char* p = rec + off;
  switch (code) {
    case LDSIM_F4: return (double)*(const float*)p;
    case LDSIM_F8: return *(const double*)p;
    case LDSIM_I4: return (double)*(const int32_t*)p;
    case LDSIM_U4: return (double)*(const uint32_t*)p;
    case LDSIM_I8: return (double)*(const int64_t*)p;
    case LDSIM_U8: return (double)*(const uint64_t*)p;
  }
  return 0.0;
}

__device__ __forceinline__ void store_field(unsigned char* rec, int off, int code, double v) {
  if (off < 0) return;
  unsigned char* p = rec + off;
  switch (code) {
    case LDSIM_F4: *(float*)p = (float)v; break;
    case LDSIM_F8: *(double*)p = v; break;
    case LDSIM_I4: *(int32_t*)p = (int32_t)v; break;
    case LDSIM_U4: *(uint32_t*)p = (uint32_t)v; break;
    case LDSIM_I8: *(int64_t*)p = (int64_t)v; break;
    case LDSIM_U8: *(uint64_t*)p = (uint64_t)v; break;
  }
}

__global__ void __launch_bounds__(256) unpack_kernel(const unsigned char* __restrict__ raw, LdsimTrackLayout lay,
                                                     SegStore s, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned char* rec = raw + i * (int64_t)lay.itemsize;
#pragma unroll
  for (int f = 0; f < LDSIM_NFIELDS - 1; f++) s.f[f][i] = load_field(rec, lay.offset[f], lay.dtype[f]);
  s.pixel_plane[i] = (int32_t)load_field(rec, lay.offset[LDSIM_PIXEL_PLANE], lay.dtype[LDSIM_PIXEL_PLANE]);
}

// write back the fields quench/drift mutate (quenching.py:43-44, drifting.py:41-58)
__global__ void __launch_bounds__(256) repack_kernel(unsigned char* __restrict__ raw, LdsimTrackLayout lay, SegStore s,
                                                     int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned char* rec = raw + i * (int64_t)lay.itemsize;
  const int fields[] = {LDSIM_N_ELECTRONS, LDSIM_N_PHOTONS, LDSIM_LONG_DIFF, LDSIM_TRAN_DIFF,
                        LDSIM_T,           LDSIM_T_START,   LDSIM_T_END};
#pragma unroll
  for (int k = 0; k < 7; k++) store_field(rec, lay.offset[fields[k]], lay.dtype[fields[k]], s.f[fields[k]][i]);
  store_field(rec, lay.offset[LDSIM_PIXEL_PLANE], lay.dtype[LDSIM_PIXEL_PLANE], (double)s.pixel_plane[i]);
}

// ---- a2 quench (quenching.py:11-44) + a3 drift (drifting.py:11-58) -------------------------------------------
// do_quench / do_drift select the stage(s); the chain runs both in one pass over the columns.
__global__ void __launch_bounds__(256) quench_drift_kernel(SegStore s, const LdsimConsts* __restrict__ c, int mode,
                                                           int do_quench, int do_drift, int* __restrict__ err) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= s.n) return;
  double n_e = s.f[LDSIM_N_ELECTRONS][i];
  if (do_quench) {
    double dEdx = s.f[LDSIM_DEDX][i], dE = s.f[LDSIM_DE][i];
    double recomb = 0;
    if (mode == 1) {  // BOX
      double csi = c->box_beta * dEdx / (c->e_field * c->lar_density);
      double r = log(c->box_alpha + csi) / csi;
      recomb = (r > 0) ? r : 0;  // Python max(0, r): NaN -> 0
    } else if (mode == 2) {  // BIRKS
      recomb = c->birks_ab / (1 + c->birks_kb * dEdx / (c->e_field * c->lar_density));
    } else {
      *err = 1;
      return;
    }
    if (isnan(recomb)) {
      *err = 2;
    } else {
      n_e = narrow_store(recomb * dE / c->w_ion, s.store_code[LDSIM_N_ELECTRONS]);
      s.f[LDSIM_N_ELECTRONS][i] = n_e;
      s.f[LDSIM_N_PHOTONS][i] =
          narrow_store((dE / c->w_ph - n_e) * c->scint_prescale, s.store_code[LDSIM_N_PHOTONS]);
    }
  }
  if (do_drift) {
    double x = s.f[LDSIM_X][i], y = s.f[LDSIM_Y][i], z = s.f[LDSIM_Z][i];
    int32_t plane = c->default_plane_index;
    for (int ip = 0; ip < c->n_tpc; ip++) {
      const double(*p)[2] = c->tpc_borders[ip];
      double zlo = fmin(p[2][1] - 2e-2, p[2][0] - 2e-2), zhi = fmax(p[2][1] + 2e-2, p[2][0] + 2e-2);
      if (p[0][0] - 2e-2 <= x && x <= p[0][1] + 2e-2 && p[1][0] - 2e-2 <= y && y <= p[1][1] + 2e-2 && zlo <= z &&
          z <= zhi) {
        plane = ip;
        break;
      }
    }
    s.pixel_plane[i] = plane;
    if (plane != c->default_plane_index) {
      double z_anode = c->tpc_borders[plane][2][0];
      double zs = s.f[LDSIM_Z_START][i], ze = s.f[LDSIM_Z_END][i], t0 = s.f[LDSIM_T0][i];
      double drift_distance = fabs(z - z_anode);
      double drift_start = fabs(fmin(zs, ze) - z_anode);
      double drift_end = fabs(fmax(zs, ze) - z_anode);
      double drift_time = drift_distance / c->v_drift;
      double lifetime_red = exp(-drift_time / c->electron_lifetime);
      s.f[LDSIM_N_ELECTRONS][i] = narrow_store(n_e * lifetime_red, s.store_code[LDSIM_N_ELECTRONS]);
      s.f[LDSIM_LONG_DIFF][i] = narrow_store(sqrt(drift_time * 2 * c->long_diff), s.store_code[LDSIM_LONG_DIFF]);
      s.f[LDSIM_TRAN_DIFF][i] = narrow_store(sqrt(drift_time * 2 * c->tran_diff), s.store_code[LDSIM_TRAN_DIFF]);
      s.f[LDSIM_T][i] = narrow_store(s.f[LDSIM_T][i] + (drift_time + t0), s.store_code[LDSIM_T]);
      s.f[LDSIM_T_START][i] = narrow_store(
          s.f[LDSIM_T_START][i] + (fmin(drift_start, drift_end) / c->v_drift + t0), s.store_code[LDSIM_T_START]);
      s.f[LDSIM_T_END][i] = narrow_store(s.f[LDSIM_T_END][i] + (fmax(drift_start, drift_end) / c->v_drift + t0),
                                         s.store_code[LDSIM_T_END]);
    }
  }
}

// ---- pixel walk (pixels_from_track.py:43-65, 111-199) ----------------------------------------------------------
struct Walk {
  int64_t x0, y0, x1, y1, plane;
};

__device__ __forceinline__ bool walk_init(const SegStore& s, const LdsimConsts* c, int64_t i, Walk& w) {
  int32_t plane = s.pixel_plane[i];
  if (plane < 0 || plane >= c->n_tpc) return false;  // reference indexes TPC_BORDERS out of bounds here
  const double(*b)[2] = c->tpc_borders[plane];
  w.x0 = (int64_t)py_floordiv(s.f[LDSIM_X_START][i] - b[0][0], c->pixel_pitch);
  w.y0 = (int64_t)py_floordiv(s.f[LDSIM_Y_START][i] - b[1][0], c->pixel_pitch);
  w.x1 = (int64_t)py_floordiv(s.f[LDSIM_X_END][i] - b[0][0], c->pixel_pitch);
  w.y1 = (int64_t)py_floordiv(s.f[LDSIM_Y_END][i] - b[1][0], c->pixel_pitch);
  w.plane = plane;
  return true;
}

// Bresenham without diagonal moves; returns the number of in-range cells, writes ids at walk positions
__device__ __forceinline__ int64_t walk_pixels(const LdsimConsts* c, Walk w, int32_t* out, int64_t cap) {
  int64_t x0 = w.x0, y0 = w.y0;
  int64_t dx = llabs(w.x1 - x0), sx = x0 < w.x1 ? 1 : -1;
  int64_t dy = -llabs(w.y1 - y0), sy = y0 < w.y1 ? 1 : -1;
  int64_t err = dx + dy, n = 0, i = 0;
  if (pix_in_range(c, x0, y0, w.plane)) {
    if (out && i < cap) out[i] = (int32_t)pixel2id(c, x0, y0, w.plane);
    n++;
  }
  while (x0 != w.x1 || y0 != w.y1) {
    i++;
    int64_t e2 = 2 * err;
    if (e2 - dy > dx - e2) {
      err += dy;
      x0 += sx;
    } else {
      err += dx;
      y0 += sy;
    }
    if (pix_in_range(c, x0, y0, w.plane)) {
      if (out && i < cap) out[i] = (int32_t)pixel2id(c, x0, y0, w.plane);
      n++;
    }
  }
  return n;
}

// a5 max_pixels: atomic max of the in-range walk length; also the batch's max tran_diff (for max_radius,
// cli/simulate_pixels.py:918) as order-preserving float bits.
__global__ void __launch_bounds__(256) max_pixels_kernel(SegStore s, const LdsimConsts* __restrict__ c, int64_t begin,
                                                         int64_t end, int32_t* __restrict__ n_max,
                                                         unsigned long long* __restrict__ max_tran_bits) {
  int64_t i = begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int32_t k = 0;
  double td = 0;
  if (i < end) {
    Walk w;
    if (walk_init(s, c, i, w)) k = (int32_t)walk_pixels(c, w, nullptr, 0);
    td = s.f[LDSIM_TRAN_DIFF][i];
    if (!(td > 0)) td = 0;
  }
  // wave-64 reduction, then one atomic per wave
  for (int off = 32; off > 0; off >>= 1) {
    k = max(k, __shfl_down(k, off));
    td = fmax(td, __shfl_down(td, off));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(n_max, k);
    atomicMax(max_tran_bits, (unsigned long long)__double_as_longlong(td));
  }
}

__device__ __forceinline__ int32_t ring_code(int x_r, int y_r) {  // pixels_from_track.py:246-269
  int dx = abs(x_r), dy = abs(y_r), dmax = max(dx, dy), dmin = min(dx, dy), dsum = dmax + dmin;
  if (dsum > 4) return -1;
  if (dsum <= 1) return dsum;
  if (dsum == 2) return dmax == 1 ? 2 : 3;
  if (dsum == 3) return dmax == 2 ? 4 : 5;
  return dmax == 2 ? 6 : (dmax == 3 ? 7 : 8);
}

// a6 get_pixels: active pixel ids (walk order, -1 gaps), first-seen-unique neighbour union, ring codes.
// Outputs must be pre-filled with -1 (the reference's cp.full(..., -1)).
__global__ void __launch_bounds__(128) get_pixels_kernel(SegStore s, const LdsimConsts* __restrict__ c, int64_t begin,
                                                         int64_t end, int radius, int32_t* __restrict__ active,
                                                         int max_active, int32_t* __restrict__ neigh,
                                                         int32_t* __restrict__ nrad, int P,
                                                         double* __restrict__ n_list,
                                                         const int32_t* __restrict__ radius_b, int32_t batch0) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t i = begin + r;
  if (i >= end) return;
  // the reference derives max_radius per batch from max(tran_diff) (cli/simulate_pixels.py:918)
  if (radius_b) {
    int32_t b = s.batch[i];
    radius = b >= 0 ? radius_b[b - batch0] : 0;
  }
  int32_t* act = active + r * max_active;
  int32_t* ng = neigh + r * P;
  int32_t* nr = nrad + r * P;
  Walk w;
  if (!walk_init(s, c, i, w)) {
    if (n_list) n_list[r] = 0;
    return;
  }
  walk_pixels(c, w, act, max_active);
  int count = 0;
  for (int p = 0; p < max_active; p++) {
    int32_t a = act[p];
    if (a == -1) continue;
    int64_t ax, ay, pl;
    id2pixel(c, a, ax, ay, pl);
    for (int x_r = -radius; x_r <= radius; x_r++)
      for (int y_r = -radius; y_r <= radius; y_r++) {
        int64_t nx = ax + x_r, ny = ay + y_r;
        if (!pix_in_range(c, nx, ny, pl)) continue;
        int32_t np_ = (int32_t)pixel2id(c, nx, ny, pl);
        bool uniq = true;
        for (int q = 0; q < count; q++)  // entries >= count are still -1 and np_ >= 0
          if (ng[q] == np_) {
            uniq = false;
            break;
          }
        if (uniq && count < P) {
          ng[count] = np_;
          nr[count] = ring_code(x_r, y_r);
          count++;
        }
      }
  }
  if (n_list) n_list[r] = (double)count;
}

// a8 time_intervals (detsim.py:18-40); max_length per launch via atomic max
__global__ void __launch_bounds__(256) time_intervals_kernel(SegStore s, const LdsimConsts* __restrict__ c,
                                                             int64_t begin, int64_t end, double* __restrict__ starts,
                                                             int32_t* __restrict__ tmax) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t i = begin + r;
  int32_t k = 0;
  if (i < end) {
    double t_end = py_round((s.f[LDSIM_T_END][i] + 1) / c->time_sampling) * c->time_sampling;
    double t_start = py_round((s.f[LDSIM_T_START][i] - c->time_padding) / c->time_sampling) * c->time_sampling;
    starts[r] = t_start;
    double len = ceil((t_end - t_start) / c->time_sampling);
    k = (len > 0 && len < 2.0e9) ? (int32_t)len : 0;
  }
  for (int off = 32; off > 0; off >>= 1) k = max(k, __shfl_down(k, off));
  if ((threadIdx.x & 63) == 0 && k > 0) atomicMax(tmax, k);
}

// ---- host launchers ----------------------------------------------------------------------------------------------
static inline int nblk(int64_t n, int b) { return (int)((n + b - 1) / b); }

extern "C++" {
int seg_launch_unpack(ldsim_ctx* ctx, const LdsimTrackLayout* lay, int64_t n) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(unpack_kernel, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream,
                     (const unsigned char*)ctx->raw.p, *lay, ctx->seg, n);
  HIPCHK(hipGetLastError());
  return 0;
}
int seg_launch_repack(ldsim_ctx* ctx, const LdsimTrackLayout* lay, int64_t n) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(repack_kernel, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, (unsigned char*)ctx->raw.p, *lay,
                     ctx->seg, n);
  HIPCHK(hipGetLastError());
  return 0;
}
int seg_launch_quench_drift(ldsim_ctx* ctx, int mode, int do_q, int do_d, int* d_err) {
  if (ctx->seg.n == 0) return 0;
  hipLaunchKernelGGL(quench_drift_kernel, dim3(nblk(ctx->seg.n, 256)), dim3(256), 0, ctx->stream, ctx->seg,
                     ctx->d_consts, mode, do_q, do_d, d_err);
  HIPCHK(hipGetLastError());
  return 0;
}
int seg_launch_max_pixels(ldsim_ctx* ctx, int64_t b, int64_t e, int32_t* d_nmax, unsigned long long* d_tranbits) {
  if (e <= b) return 0;
  hipLaunchKernelGGL(max_pixels_kernel, dim3(nblk(e - b, 256)), dim3(256), 0, ctx->stream, ctx->seg, ctx->d_consts, b,
                     e, d_nmax, d_tranbits);
  HIPCHK(hipGetLastError());
  return 0;
}
int seg_launch_get_pixels(ldsim_ctx* ctx, int64_t b, int64_t e, int radius, int32_t* active, int max_active,
                          int32_t* neigh, int32_t* nrad, int P, double* n_list, const int32_t* radius_b,
                          int32_t batch0) {
  if (e <= b) return 0;
  hipLaunchKernelGGL(get_pixels_kernel, dim3(nblk(e - b, 128)), dim3(128), 0, ctx->stream, ctx->seg, ctx->d_consts, b,
                     e, radius, active, max_active, neigh, nrad, P, n_list, radius_b, batch0);
  HIPCHK(hipGetLastError());
  return 0;
}
int seg_launch_time_intervals(ldsim_ctx* ctx, int64_t b, int64_t e, double* starts, int32_t* tmax) {
  if (e <= b) return 0;
  hipLaunchKernelGGL(time_intervals_kernel, dim3(nblk(e - b, 256)), dim3(256), 0, ctx->stream, ctx->seg,
                     ctx->d_consts, b, e, starts, tmax);
  HIPCHK(hipGetLastError());
  return 0;
}
}
