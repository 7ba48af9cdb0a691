// light_wvfm.hip -- second half of the light chain: from the photo-electron rate per (channel, tick) to the
// digitised waveforms the reference writes as light_wvfm.
//   light_sim.calc_stat_fluctuations        larndsim/light_sim.py:186-238   light_stat_fluct_kernel
//   light_sim.get_triggers                  :339-443                       light_trig_mask_kernel + the host scan below
//   light_sim.gen_light_detector_noise      :445-478                       light_noise_spec_kernel + light_idft_kernel
//   light_sim.sim_triggers / digitize_signal :480-619                      light_sample_kernel + light_digitize_kernel
// All arithmetic that decides an integer (a Poisson count, a threshold crossing, a rounded ADC value) is f64 in the
// reference's operation order; the random numbers come from the restated numba generator (rng.h, third-party, unpinned) or,
// for the noise phases -- cupy's global generator in the reference, which nothing can pin -- from a counter hash.
//
// The padded copies sim_triggers makes of the signal and of the two truth arrays (each [n_det][n_ticks][MAX_MC_TRUTH_IDS],
// GBs for 2x2) are not made: the kernels translate padded tick / sorted row indices back to the caller's arrays, and the
// signal (+ noise) is evaluated only at the ticks the digitiser reads.
#include <algorithm>
#include <cmath>
#include <vector>

#include "ldsim_dev.h"
#include "rng.h"

extern "C++" int rng_ensure_states(ldsim_ctx* ctx, int64_t n);      // kernels_rng.hip

#define LW_CK(x)             \
  do {                       \
    int rc_ = (x);           \
    if (rc_) return rc_;     \
  } while (0)
#define LW_NEED(cond, msg)         \
  do {                             \
    if (!(cond)) {                 \
      ldsim_set_error("%s", msg);  \
      return LDSIM_EINVAL;         \
    }                              \
  } while (0)

namespace {
struct LTmp {                       // temporary device buffer
  void* p = nullptr;
  ~LTmp() {
    if (p) (void)hipFree(p);
  }
  int alloc(size_t bytes) {
    HIPCHK(hipMalloc(&p, bytes ? bytes : 8));
    return 0;
  }
  template <class T>
  T* as() { return (T*)p; }
};
}  // namespace

// ---- calc_stat_fluctuations ----------------------------------------------------------------------------------------------
// xoroshiro128p_poisson_int32 (:186-216): inversion with one float32 uniform below a mean of 30, else a normal truncated at 0
__device__ inline int poisson_int32(double mean, RngState& st) {
  if (mean <= 0) return 0;
  if (mean < 30) {
    const double u = (double)rng_uniform_f32(st);
    int x = 0;
    double p = exp(-mean), s = p, prev_s = s;
    while (u > s) {
      x += 1;
      p = p * mean / x;
      prev_s = s;
      s = s + p;
      if (s == prev_s) break;
    }
    return x;
  }
  const double v = (double)rng_normal_f32(st) * sqrt(mean) + mean;
  const long long iv = (long long)v;
  return iv > 0 ? (int)iv : 0;
}

// element e = idet*ntick + itick draws from states[e] (:236), one thread each
__global__ void __launch_bounds__(256) light_stat_fluct_kernel(const float* __restrict__ inc, float* __restrict__ out,
                                                               RngState* __restrict__ states, int64_t n, double tick) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const float x = inc[e];
  if (x > 0) {
    RngState st = states[e];
    out[e] = (float)(1. / tick * (double)poisson_int32((double)x * tick, st));
    states[e] = st;
  } else {
    out[e] = 0.f;
  }
}

extern "C++" int light_launch_stat_fluct(ldsim_ctx* ctx, const float* inc, float* out, int64_t n) {
  if (n == 0) return 0;
  LW_CK(rng_ensure_states(ctx, n));
  hipLaunchKernelGGL(light_stat_fluct_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, inc, out,
                     (RngState*)ctx->d_rng.p, n, ctx->h_consts.light_tick_size);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---- get_triggers ------------------------------------------------------------------------------------------------------------
// numpy's pairwise summation of a contiguous run (what .mean(axis=-1) of the f8 blocks does, :363), n <= 128
__host__ __device__ inline double np_pairwise_sum128(const double* a, int n) {
  if (n < 8) {
    double res = 0.;
    for (int i = 0; i < n; i++) res += a[i];
    return res;
  }
  double r[8];
  for (int j = 0; j < 8; j++) r[j] = a[j];
  int i;
  for (i = 8; i < n - (n % 8); i += 8)
    for (int j = 0; j < 8; j++) r[j] += a[i + j];
  double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; i++) res += a[i];
  return res;
}
static double np_pairwise_sum(const double* a, int64_t n) {
  if (n <= 128) return np_pairwise_sum128(a, (int)n);
  int64_t n2 = n / 2;
  n2 -= n2 % 8;
  return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

#define LW_MAX_SF 128
// one thread per (group, block of `sf` ticks): f4 sum of the group's rows in row order (:358), zero padding, f8 block mean,
// `< threshold` (:368); a block below threshold marks its ticks for every module one of the group's rows belongs to
__global__ void __launch_bounds__(64) light_trig_mask_kernel(const float* __restrict__ sig, int n_det, int64_t nt, int per,
                                                             int sf, int64_t n_blocks, const double* __restrict__ thr,
                                                             const int32_t* __restrict__ row_module,
                                                             uint8_t* __restrict__ above /* [n_mod][nt] */) {
  const int64_t blk = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const int g = blockIdx.y;
  if (blk >= n_blocks) return;
  double a[LW_MAX_SF];
  const int64_t t0 = blk * sf;
  if (t0 >= nt) return;                                     // the all-padding block past the end marks nothing
  for (int i = 0; i < sf; i++) {
    const int64_t t = t0 + i;
    if (t < nt) {
      float acc = sig[(int64_t)(g * per) * nt + t];
      for (int r = 1; r < per; r++) acc = acc + sig[(int64_t)(g * per + r) * nt + t];
      a[i] = (double)acc;
    } else {
      a[i] = 0.0;
    }
  }
  const double mean = np_pairwise_sum128(a, sf) / (double)sf;
  if (!(mean < thr[g])) return;
  int prev = -1;
  for (int r = 0; r < per; r++) {
    const int m = row_module[g * per + r];
    if (m < 0 || m == prev) continue;
    prev = m;
    for (int i = 0; i < sf && t0 + i < nt; i++) above[(int64_t)m * nt + t0 + i] = 1;
  }
}

// the threshold loop of :386-411 over one module's mask, bookkeeping kept as the reference has it: `hot` is re-sliced by
// the ABSOLUTE index of the trigger just found plus the digitisation window
static void trigger_scan(const uint8_t* hot, int64_t nt, int64_t digit_ticks, int mod, std::vector<int64_t>& idx,
                         std::vector<int32_t>& mods) {
  int64_t start = 0, last = 0;                               // hot[start:] is the current view
  while (start < nt) {
    int64_t rel = -1;
    for (int64_t i = start; i < nt; i++)
      if (hot[i]) { rel = i - start; break; }
    if (rel < 0) break;
    const int64_t nxt = rel + last;
    idx.push_back(nxt);
    mods.push_back(mod);
    start += nxt + digit_ticks;
    last = nxt + digit_ticks;
  }
}

// d_signal: [n_det][n_ticks] f4 on the device
static int light_triggers_run(ldsim_ctx* ctx, const float* d_signal, int32_t n_det, int64_t n_ticks,
                              const double* group_threshold, int32_t n_grp, const int32_t* row_module, int32_t n_mod,
                              int64_t* trig_idx, int32_t* trig_mod, int64_t capacity, int64_t* n_trig) {
  const LdsimConsts& h = ctx->h_consts;
  const int per = h.op_channel_per_trig;
  LW_NEED(per > 0 && n_grp * per == n_det, "n_det must be n_grp * OP_CHANNEL_PER_TRIG");
  LW_NEED(h.light_tick_size > 0 && h.light_digit_sample_spacing > 0, "light constants not set");
  const int64_t sf = llrint(h.light_digit_sample_spacing / h.light_tick_size);        // Python round(): half to even
  LW_NEED(sf >= 1 && sf <= LW_MAX_SF, "LIGHT_DIGIT_SAMPLE_SPACING / LIGHT_TICK_SIZE must round to 1..128");
  for (int i = 0; i < n_det; i++) LW_NEED(row_module[i] >= -1 && row_module[i] < n_mod, "row_module out of range");
  *n_trig = 0;
  if (n_ticks == 0 || n_mod == 0) return 0;
  const int64_t padding = sf - n_ticks % sf;
  const int64_t n_blocks = (n_ticks + padding) / sf;
  const int64_t digit_ticks = (int64_t)ceil((h.light_trig_window[1] + h.light_trig_window[0]) / h.light_tick_size);
  LTmp d_thr, d_rm, d_above;
  LW_CK(d_thr.alloc((size_t)n_grp * 8));
  LW_CK(d_rm.alloc((size_t)n_det * 4));
  LW_CK(d_above.alloc((size_t)n_mod * n_ticks));
  hipStream_t st = ctx->stream;
  HIPCHK(hipMemcpyAsync(d_thr.p, group_threshold, (size_t)n_grp * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_rm.p, row_module, (size_t)n_det * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemsetAsync(d_above.p, 0, (size_t)n_mod * n_ticks, st));
  hipLaunchKernelGGL(light_trig_mask_kernel, dim3((unsigned)((n_blocks + 63) / 64), (unsigned)n_grp), dim3(64), 0, st,
                     d_signal, n_det, n_ticks, per, (int)sf, n_blocks, d_thr.as<double>(), d_rm.as<int32_t>(),
                     d_above.as<uint8_t>());
  HIPCHK(hipGetLastError());
  std::vector<uint8_t> above((size_t)n_mod * n_ticks);
  HIPCHK(hipMemcpyAsync(above.data(), d_above.p, above.size(), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  std::vector<int64_t> idx;
  std::vector<int32_t> mods;
  for (int m = 0; m < n_mod; m++) trigger_scan(above.data() + (size_t)m * n_ticks, n_ticks, digit_ticks, m, idx, mods);
  *n_trig = (int64_t)idx.size();
  if ((int64_t)idx.size() > capacity) {
    ldsim_set_error("%lld triggers, the output arrays hold %lld", (long long)idx.size(), (long long)capacity);
    return LDSIM_ENOSPC;
  }
  for (size_t i = 0; i < idx.size(); i++) {
    trig_idx[i] = idx[i];
    trig_mod[i] = mods[i];
  }
  return 0;
}

// ---- gen_light_detector_noise ---------------------------------------------------------------------------------------------
__host__ __device__ inline uint64_t lw_hash(uint64_t x) {      // SplitMix64 finaliser
  x += 0x9E3779B97F4A7C15ULL;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
  return x ^ (x >> 31);
}

// spectrum row r, frequency bin k: np.interp of the channel's noise spectrum at the FFT frequency (j, dx, dxp and `exact`
// prepared on the host), times the power-rescaling factor (:456-461), times exp(2j*pi*u) (:465)
__global__ void __launch_bounds__(256) light_noise_spec_kernel(const double* __restrict__ tab, int nbins,
                                                               const int32_t* __restrict__ chan, int R, int m,
                                                               const int32_t* __restrict__ jidx, const double* __restrict__ dx,
                                                               const double* __restrict__ dxp, double factor,
                                                               const double* __restrict__ phases /* [R][m] or NULL */,
                                                               uint64_t seed, double2* __restrict__ spec) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int r = blockIdx.y;
  if (k >= m) return;
  const double* fp = tab + (int64_t)chan[r] * nbins;
  const int j = jidx[k];
  double v = 0.0;
  if (j >= 0) {
    if (dxp[k] == 0.0) {
      v = fp[j];                                            // x == xp[j], or the last point
    } else {
      const double slope = (fp[j + 1] - fp[j]) / dxp[k];
      v = slope * dx[k] + fp[j];
      if (isnan(v)) {
        v = slope * (dx[k] - dxp[k]) + fp[j + 1];
        if (isnan(v) && fp[j] == fp[j + 1]) v = fp[j];
      }
    }
  }
  v *= factor;
  const double u = phases ? phases[(int64_t)r * m + k]
                          : (double)(lw_hash(seed ^ lw_hash(((uint64_t)chan[r] << 32) | (uint32_t)k)) >> 11) * (1.0 / 9007199254740992.0);
  double s, c;
  sincos((2.0 * 3.14159265358979323846) * u, &s, &c);
  spec[(int64_t)r * m + k] = make_double2(v * c, v * s);
}

// numpy.fft.irfft(spec, axis=-1)[tick] for the listed ticks: N = 2(m-1) points,
//   x[n] = (Re X0 + (-1)^n Re X_{N/2} + 2 sum_{k=1}^{N/2-1} (Re X_k cos(2 pi k n / N) - Im X_k sin(2 pi k n / N))) / N,
// then np.round and the digitiser's LSB (:470).  One thread per (row, tick): the spectrum is staged through LDS in chunks
// (all lanes read the same X_k), the twiddle is advanced by rotation and re-seeded exactly at every chunk start.
#define IDFT_CHUNK 512
__global__ void __launch_bounds__(256) light_idft_kernel(const double2* __restrict__ spec, int m, const int32_t* __restrict__ ticks,
                                                         int nn, double lsb, double* __restrict__ out /* [R][nn] */) {
  __shared__ double2 s_x[IDFT_CHUNK];
  const int r = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int64_t N = 2 * (int64_t)(m - 1);
  const double2* X = spec + (int64_t)r * m;
  const int64_t n = i < nn ? ticks[i] : 0;
  const bool live = i < nn && n < N;                        // an odd request is padded with one zero sample (:472-474)
  double sn, cs;
  sincospi(2.0 * (double)n / (double)N, &sn, &cs);          // e^{i theta}, theta = 2 pi n / N
  double acc = 0.0;
  for (int k0 = 1; k0 < m - 1; k0 += IDFT_CHUNK) {
    const int nk = min(IDFT_CHUNK, m - 1 - k0);
    __syncthreads();
    for (int k = threadIdx.x; k < nk; k += 256) s_x[k] = X[k0 + k];
    __syncthreads();
    if (!live) continue;
    double wr, wi;
    sincospi(2.0 * (double)((n * (int64_t)k0) % N) / (double)N, &wi, &wr);
    for (int k = 0; k < nk; k++) {
      const double2 x = s_x[k];
      acc = fma(x.x, wr, acc);
      acc = fma(-x.y, wi, acc);
      const double t = fma(wr, cs, -(wi * sn));
      wi = fma(wr, sn, wi * cs);
      wr = t;
    }
  }
  if (i >= nn) return;
  double v = 0.0;
  if (live) {
    v = 2.0 * acc + X[0].x;
    if (m > 1) v += (n & 1) ? -X[m - 1].x : X[m - 1].x;
    v = rint(v / (double)N) * lsb;
  }
  out[(int64_t)r * nn + i] = v;
}

// Noise for rows with optical channels chan[R] at the listed ticks of an n_samples-long waveform -> d_out [R][nn].
// h_tab: [n_ch][nbins] spectrum table on the host (uploaded here); h_phases: [R][n_samples/2+1] or NULL (hash of seed).
static int light_noise_run(ldsim_ctx* ctx, const double* h_tab, int32_t n_ch, int32_t nbins, const std::vector<int32_t>& chan,
                           int64_t n_samples, const std::vector<int32_t>& ticks, const double* h_phases, uint64_t seed,
                           double* d_out) {
  const LdsimConsts& h = ctx->h_consts;
  const int R = (int)chan.size(), nn = (int)ticks.size();
  if (R == 0 || nn == 0) return 0;
  LW_NEED(n_samples >= 2, "gen_light_detector_noise needs at least 2 samples (the reference divides by an empty mean below that)");
  LW_NEED(nbins >= 2 && n_samples < (1LL << 30), "noise spectrum needs >= 2 bins");
  for (int c : chan) LW_NEED(c >= 0 && c < n_ch, "optical channel outside the noise spectrum table");
  const int m = (int)(n_samples / 2 + 1);
  // np.fft.rfftfreq(n, d): arange(n//2 + 1) * (1.0 / (n * d))
  const int64_t n_noise = 2 * (int64_t)(nbins - 1);
  const double val_n = 1.0 / ((double)n_noise * h.light_det_noise_sample_spacing);
  const double val_d = 1.0 / ((double)n_samples * h.light_tick_size);
  std::vector<double> xp((size_t)nbins), xd((size_t)m), tmp;
  for (int i = 0; i < nbins; i++) xp[(size_t)i] = (double)i * val_n;
  for (int k = 0; k < m; k++) xd[(size_t)k] = (double)k * val_d;
  auto mean_diff = [&](const std::vector<double>& f) {
    tmp.resize(f.size() - 1);
    for (size_t i = 0; i + 1 < f.size(); i++) tmp[i] = f[i + 1] - f[i];
    return np_pairwise_sum(tmp.data(), (int64_t)tmp.size()) / (double)tmp.size();
  };
  const double bin_size = mean_diff(xd);
  const double factor = sqrt(mean_diff(xp) / bin_size) * h.light_digit_sample_spacing / h.light_tick_size;
  std::vector<int32_t> jidx((size_t)m);
  std::vector<double> dx((size_t)m), dxp((size_t)m);
  for (int k = 0; k < m; k++) {
    const double x = xd[(size_t)k];
    if (x > xp.back() || x < xp.front()) {                  // left = right = 0 (:459)
      jidx[(size_t)k] = -1;
      continue;
    }
    int j = (int)(std::upper_bound(xp.begin(), xp.end(), x) - xp.begin()) - 1;       // xp[j] <= x < xp[j+1]
    jidx[(size_t)k] = j;
    if (j == nbins - 1 || xp[(size_t)j] == x) {
      dxp[(size_t)k] = 0.0;
    } else {
      dx[(size_t)k] = x - xp[(size_t)j];
      dxp[(size_t)k] = xp[(size_t)j + 1] - xp[(size_t)j];
    }
  }
  LTmp d_tab, d_chan, d_j, d_dx, d_dxp, d_ph, d_spec, d_ticks;
  LW_CK(d_tab.alloc((size_t)n_ch * nbins * 8));
  LW_CK(d_chan.alloc((size_t)R * 4));
  LW_CK(d_j.alloc((size_t)m * 4));
  LW_CK(d_dx.alloc((size_t)m * 8));
  LW_CK(d_dxp.alloc((size_t)m * 8));
  LW_CK(d_spec.alloc((size_t)R * m * 16));
  LW_CK(d_ticks.alloc((size_t)nn * 4));
  hipStream_t st = ctx->stream;
  HIPCHK(hipMemcpyAsync(d_tab.p, h_tab, (size_t)n_ch * nbins * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_chan.p, chan.data(), (size_t)R * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_j.p, jidx.data(), (size_t)m * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_dx.p, dx.data(), (size_t)m * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_dxp.p, dxp.data(), (size_t)m * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_ticks.p, ticks.data(), (size_t)nn * 4, hipMemcpyHostToDevice, st));
  if (h_phases) {
    LW_CK(d_ph.alloc((size_t)R * m * 8));
    HIPCHK(hipMemcpyAsync(d_ph.p, h_phases, (size_t)R * m * 8, hipMemcpyHostToDevice, st));
  }
  hipLaunchKernelGGL(light_noise_spec_kernel, dim3((unsigned)((m + 255) / 256), (unsigned)R), dim3(256), 0, st,
                     d_tab.as<double>(), nbins, d_chan.as<int32_t>(), R, m, d_j.as<int32_t>(), d_dx.as<double>(),
                     d_dxp.as<double>(), factor, h_phases ? d_ph.as<double>() : (const double*)nullptr, seed,
                     d_spec.as<double2>());
  HIPCHK(hipGetLastError());
  const double lsb = ldexp(1.0, 16 - h.light_nbit);
  hipLaunchKernelGGL(light_idft_kernel, dim3((unsigned)((nn + 255) / 256), (unsigned)R), dim3(256), 0, st,
                     d_spec.as<double2>(), m, d_ticks.as<int32_t>(), nn, lsb, d_out);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st));                         // the temporaries go out of scope
  return 0;
}

// ---- sim_triggers / digitize_signal ------------------------------------------------------------------------------------------
struct DigitArgs {
  // padded / sorted signal evaluated at the needed ticks
  const double* sig_s;        // [Rp][nn]
  int Rp, nn;
  // per sample: interp branch (0: value 0, 1: sig[slot0], 2: lerp) and operands
  const int32_t* mode;
  const int32_t* slot0;
  const int32_t* slot1;
  const double* frac;
  const int64_t* itick0;      // floor / ceil of the sample tick in the PADDED frame
  const int64_t* itick1;
  // triggers
  const int32_t* trig_row;    // [ntrig][ndm] row of the padded signal each trigger channel reads (first match, else last row)
  const int32_t* trig_op;     // [ntrig][ndm] optical channel ids
  int ntrig, ndm, ns;
  // truth of the caller's (unpadded, unsorted) arrays
  const int64_t* tid;         // [R][T][Mt]
  const double* tph;
  const int32_t* src_row;     // [Rp] caller's row of a padded row, -1 for an added (missing) channel
  int64_t T, n0, Tp;
  int Mt;
  double truth_threshold, lsb;
  int sig_f32diff;            // Numba typing of interp's v1 - v0 on an f4 signal
};

__device__ inline int64_t lw_tid(const DigitArgs& A, int prow, int64_t ptick, int j) {
  const int src = A.src_row[prow];
  const int64_t t = ptick - A.n0;
  return (src >= 0 && t >= 0 && t < A.T) ? A.tid[((int64_t)src * A.T + t) * A.Mt + j] : -1;
}
__device__ inline double lw_tph(const DigitArgs& A, int prow, int64_t ptick, int j) {
  const int src = A.src_row[prow];
  const int64_t t = ptick - A.n0;
  return (src >= 0 && t >= 0 && t < A.T) ? A.tph[((int64_t)src * A.T + t) * A.Mt + j] : 0.0;
}

// one thread per (trigger, channel of the trigger, sample), :492-543
__global__ void __launch_bounds__(256) light_digitize_kernel(DigitArgs A, double* __restrict__ digit,
                                                             int64_t* __restrict__ dtid, double* __restrict__ dtph) {
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= (int64_t)A.ntrig * A.ndm * A.ns) return;
  const int is = (int)(o % A.ns);
  const int64_t tc = o / A.ns;
  const int s = A.trig_row[tc];
  double v = 0.0;
  const int md = A.mode[is];
  if (md == 1) {
    v = A.sig_s[(int64_t)s * A.nn + A.slot0[is]];
  } else if (md == 2) {
    const double v0 = A.sig_s[(int64_t)s * A.nn + A.slot0[is]], v1 = A.sig_s[(int64_t)s * A.nn + A.slot1[is]];
    const double d = A.sig_f32diff ? (double)((float)v1 - (float)v0) : v1 - v0;
    v = v0 + d * A.frac[is];
  }
  digit[o] = rint(v / A.lsb) * A.lsb;                       // :617
  if (A.Mt == 0) return;
  const int64_t it0 = A.itick0[is], it1 = A.itick1[is];
  if (it0 < 0 || it0 >= A.Tp) return;
  const int idet = A.trig_op[tc];
  const double fr = A.frac[is];
  int itrue = 0;
  for (int j = 0; j < A.Mt; j++) {
    if (itrue >= A.Mt) break;
    const int64_t id0 = lw_tid(A, s, it0, j);
    if (id0 == -1) break;
    double photons0 = 0, photons1 = 0;
    int64_t* slot = &dtid[o * A.Mt + itrue];
    if (id0 == *slot || *slot == -1) {
      *slot = id0;
      itrue += 1;
      // :520 reads row `idet` (the optical channel id) of the photons array, not the matched row
      photons0 = (idet >= 0 && idet < A.Rp) ? lw_tph(A, idet, it0, j) : 0.0;
      if (fabs(photons0) < A.truth_threshold) continue;
      if (it1 < A.Tp) {
        if (id0 == lw_tid(A, s, it1, j)) {
          photons1 = lw_tph(A, s, it1, j);
        } else {
          for (int k = 0; k < A.Mt; k++)
            if (id0 == lw_tid(A, s, it1, k)) {
              photons1 = lw_tph(A, s, it1, k);
              break;
            }
        }
      }
    }
    const int last = itrue - 1 < 0 ? A.Mt - 1 : itrue - 1;
    if (dtid[o * A.Mt + last] != -1)                        // interp(sample_tick - itick0, (photons0, photons1), 0, 0)
      dtph[o * A.Mt + last] = fr == 0.0 ? photons0 : photons0 + (photons1 - photons0) * fr;
  }
}

// padded/sorted signal at the needed ticks: the caller's f4 value (0 in the padding and in added rows) plus the noise; an
// array that was never padded is still f4 when the reference adds the noise, so the sum rounds to f4 (:592)
__global__ void __launch_bounds__(256) light_sample_kernel(const float* __restrict__ sig, int64_t T, int64_t n0,
                                                           const int32_t* __restrict__ src_row, const int32_t* __restrict__ ticks,
                                                           int nn, int Rp, const double* __restrict__ noise /* [Rp][nn] or NULL */,
                                                           int is_f4, double* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int r = blockIdx.y;
  if (i >= nn) return;
  const int src = src_row[r];
  const int64_t t = (int64_t)ticks[i] - n0;
  const float x = (src >= 0 && t >= 0 && t < T) ? sig[(int64_t)src * T + t] : 0.f;
  const double nz = noise ? noise[(int64_t)r * nn + i] : 0.0;
  double v = (double)x + nz;
  if (is_f4 && src >= 0) v = (double)(float)v;
  out[(int64_t)r * nn + i] = v;
}

// light_sim.sim_triggers: d_signal [R][T] f4, d_tid / d_tph [R][T][Mt] on the device; everything else on the host.
// Outputs d_digit [ntrig][ndm][ns] f64, d_dtid / d_dtph [..][Mt] on the device (initialised here).
static int sim_triggers_run(ldsim_ctx* ctx, const float* d_signal, const int32_t* sig_op, int32_t R, int64_t T,
                            const int64_t* d_tid, const double* d_tph, int32_t Mt, const int64_t* trig_idx, int32_t ntrig,
                            const int32_t* trig_op, int32_t ndm, int32_t ns, const double* noise_tab, int32_t n_ch,
                            int32_t nbins, const double* ph_signal, const double* ph_missing, double* d_digit,
                            int64_t* d_dtid, double* d_dtph) {
  const LdsimConsts& h = ctx->h_consts;
  hipStream_t st = ctx->stream;
  const size_t n_out = (size_t)ntrig * ndm * ns;
  if (n_out == 0) return 0;
  HIPCHK(hipMemsetAsync(d_digit, 0, n_out * 8, st));
  if (Mt) {
    HIPCHK(hipMemsetAsync(d_dtid, 0xFF, n_out * Mt * 8, st));
    HIPCHK(hipMemsetAsync(d_dtph, 0, n_out * Mt * 8, st));
  }
  LW_NEED(h.light_tick_size > 0 && h.light_digit_sample_spacing > 0, "light constants not set");
  // padding (:566-587)
  int64_t tmin = trig_idx[0], tmax = trig_idx[0];
  for (int i = 1; i < ntrig; i++) {
    tmin = std::min(tmin, trig_idx[i]);
    tmax = std::max(tmax, trig_idx[i]);
  }
  const int64_t pre = (int64_t)ceil(h.light_trig_window[0] / h.light_tick_size);
  const int64_t post = (int64_t)ceil(h.light_trig_window[1] / h.light_tick_size);
  int64_t n0 = 0, Tp = T;
  bool is_f4 = true;
  if (tmin - pre < 0) {
    n0 = pre - tmin;
    Tp += n0;
    is_f4 = false;
  }
  if (post + tmax + n0 > Tp) {
    Tp += post + tmax + n0 - Tp;
    is_f4 = false;
  }
  LW_NEED(Tp < (1LL << 30), "padded waveform too long");
  // channels of the triggers the signal does not hold, appended and everything sorted by channel (:594-609)
  std::vector<int32_t> chan(sig_op, sig_op + R), src_row((size_t)R);
  for (int i = 0; i < R; i++) src_row[(size_t)i] = i;
  std::vector<int32_t> missing;
  {
    std::vector<int32_t> have(chan);
    std::sort(have.begin(), have.end());
    for (size_t i = 0; i < (size_t)ntrig * ndm; i++)
      if (!std::binary_search(have.begin(), have.end(), trig_op[i])) missing.push_back(trig_op[i]);
    std::sort(missing.begin(), missing.end());
    missing.erase(std::unique(missing.begin(), missing.end()), missing.end());
  }
  const int n_missing = (int)missing.size();
  const int Rp = R + n_missing;
  if (Rp == 0) return 0;
  std::vector<int32_t> first_call_row((size_t)Rp, -1);      // row of the first / second noise call a padded row takes
  for (int i = 0; i < R; i++) first_call_row[(size_t)i] = i;
  if (n_missing) {
    for (int i = 0; i < n_missing; i++) {
      chan.push_back(missing[(size_t)i]);
      src_row.push_back(-1);
    }
    std::vector<int32_t> order((size_t)Rp);
    for (int i = 0; i < Rp; i++) order[(size_t)i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return chan[(size_t)a] < chan[(size_t)b]; });
    std::vector<int32_t> c2((size_t)Rp), s2((size_t)Rp), f2((size_t)Rp);
    for (int i = 0; i < Rp; i++) {
      c2[(size_t)i] = chan[(size_t)order[(size_t)i]];
      s2[(size_t)i] = src_row[(size_t)order[(size_t)i]];
      f2[(size_t)i] = order[(size_t)i];                     // < R: row of the first noise call, else R + row of the second
    }
    chan.swap(c2);
    src_row.swap(s2);
    first_call_row.swap(f2);
  }
  // the row each trigger channel reads (:494-498: first match; the loop variable stays on the last row when none matches)
  std::vector<int32_t> trig_row((size_t)ntrig * ndm);
  for (size_t i = 0; i < trig_row.size(); i++) {
    int s = Rp - 1;
    for (int r = 0; r < Rp; r++)
      if (chan[(size_t)r] == trig_op[i]) { s = r; break; }
    trig_row[i] = s;
  }
  // the ticks the digitiser reads (:491), with interp's branches resolved per sample (:255-271)
  std::vector<int32_t> mode((size_t)ns), slot0((size_t)ns, 0), slot1((size_t)ns, 0), ticks;
  std::vector<double> frac((size_t)ns, 0.0);
  std::vector<int64_t> it0((size_t)ns), it1((size_t)ns);
  auto slot_of = [&](int64_t t) {
    auto it = std::lower_bound(ticks.begin(), ticks.end(), (int32_t)t);
    return (int32_t)(it - ticks.begin());
  };
  for (int pass = 0; pass < 2; pass++) {
    for (int i = 0; i < ns; i++) {
      const double stick = (double)i * h.light_digit_sample_spacing / h.light_tick_size;
      const int64_t i0 = (int64_t)floor(stick);
      it0[(size_t)i] = i0;
      it1[(size_t)i] = (int64_t)ceil(stick);
      frac[(size_t)i] = stick - (double)i0;
      int md;
      if (i0 < 0 || i0 > Tp - 1) md = 0;
      else if ((double)i0 == stick) md = 1;
      else if (i0 > Tp - 2) md = 0;
      else md = 2;
      mode[(size_t)i] = md;
      if (pass == 0) {
        if (md >= 1) ticks.push_back((int32_t)i0);
        if (md == 2) ticks.push_back((int32_t)i0 + 1);
      } else {
        if (md >= 1) slot0[(size_t)i] = slot_of(i0);
        if (md == 2) slot1[(size_t)i] = slot_of(i0 + 1);
      }
    }
    if (pass == 0) {
      std::sort(ticks.begin(), ticks.end());
      ticks.erase(std::unique(ticks.begin(), ticks.end()), ticks.end());
    }
  }
  const int nn = (int)ticks.size();
  // noise of the two gen_light_detector_noise calls (:592, :597), rows in padded order
  LTmp d_noise;
  bool any_noise = false;
  if (noise_tab)
    for (int r = 0; r < Rp && !any_noise; r++) {
      LW_NEED(chan[(size_t)r] >= 0 && chan[(size_t)r] < n_ch, "optical channel outside the noise spectrum table");
      for (int b = 0; b < nbins; b++)
        if (noise_tab[(size_t)chan[(size_t)r] * nbins + b] != 0.0) { any_noise = true; break; }
    }
  if (any_noise && nn) {
    const int m = (int)(Tp / 2 + 1);
    std::vector<double> ph;
    if (ph_signal || ph_missing) {
      LW_NEED(ph_signal && (n_missing == 0 || ph_missing), "phases of both noise calls are needed");
      ph.resize((size_t)Rp * m);
      for (int r = 0; r < Rp; r++) {
        const int f = first_call_row[(size_t)r];
        const double* srcp = f < R ? ph_signal + (size_t)f * m : ph_missing + (size_t)(f - R) * m;
        std::copy(srcp, srcp + m, ph.begin() + (size_t)r * m);
      }
    } else if (!ctx->rng_seeded) {
      ldsim_set_error("the noise spectrum is non-zero but no random state exists: call ldsim_rng_seed first");
      return LDSIM_ESTATE;
    }
    LW_CK(d_noise.alloc((size_t)Rp * nn * 8));
    const uint64_t seed = lw_hash(ctx->rng_seed ^ lw_hash(0x6c69676874ULL + ctx->light_noise_calls++));
    LW_CK(light_noise_run(ctx, noise_tab, n_ch, nbins, chan, Tp, ticks, ph.empty() ? nullptr : ph.data(), seed,
                          d_noise.as<double>()));
  }
  // device copies of the small host tables
  LTmp d_src, d_ticks, d_sig, d_mode, d_s0, d_s1, d_frac, d_it0, d_it1, d_trow, d_top;
  LW_CK(d_src.alloc((size_t)Rp * 4));
  LW_CK(d_ticks.alloc((size_t)(nn ? nn : 1) * 4));
  LW_CK(d_sig.alloc((size_t)Rp * (nn ? nn : 1) * 8));
  LW_CK(d_mode.alloc((size_t)ns * 4));
  LW_CK(d_s0.alloc((size_t)ns * 4));
  LW_CK(d_s1.alloc((size_t)ns * 4));
  LW_CK(d_frac.alloc((size_t)ns * 8));
  LW_CK(d_it0.alloc((size_t)ns * 8));
  LW_CK(d_it1.alloc((size_t)ns * 8));
  LW_CK(d_trow.alloc(trig_row.size() * 4));
  LW_CK(d_top.alloc(trig_row.size() * 4));
  HIPCHK(hipMemcpyAsync(d_src.p, src_row.data(), (size_t)Rp * 4, hipMemcpyHostToDevice, st));
  if (nn) HIPCHK(hipMemcpyAsync(d_ticks.p, ticks.data(), (size_t)nn * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_mode.p, mode.data(), (size_t)ns * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_s0.p, slot0.data(), (size_t)ns * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_s1.p, slot1.data(), (size_t)ns * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_frac.p, frac.data(), (size_t)ns * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_it0.p, it0.data(), (size_t)ns * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_it1.p, it1.data(), (size_t)ns * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_trow.p, trig_row.data(), trig_row.size() * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_top.p, trig_op, trig_row.size() * 4, hipMemcpyHostToDevice, st));
  if (nn) {
    hipLaunchKernelGGL(light_sample_kernel, dim3((unsigned)((nn + 255) / 256), (unsigned)Rp), dim3(256), 0, st, d_signal, T,
                       n0, d_src.as<int32_t>(), d_ticks.as<int32_t>(), nn, Rp,
                       any_noise ? d_noise.as<double>() : (const double*)nullptr, (int)is_f4, d_sig.as<double>());
    HIPCHK(hipGetLastError());
  }
  DigitArgs A;
  A.sig_s = d_sig.as<double>(); A.Rp = Rp; A.nn = nn;
  A.mode = d_mode.as<int32_t>(); A.slot0 = d_s0.as<int32_t>(); A.slot1 = d_s1.as<int32_t>(); A.frac = d_frac.as<double>();
  A.itick0 = d_it0.as<int64_t>(); A.itick1 = d_it1.as<int64_t>();
  A.trig_row = d_trow.as<int32_t>(); A.trig_op = d_top.as<int32_t>();
  A.ntrig = ntrig; A.ndm = ndm; A.ns = ns;
  A.tid = d_tid; A.tph = d_tph; A.src_row = d_src.as<int32_t>();
  A.T = T; A.n0 = n0; A.Tp = Tp; A.Mt = Mt;
  A.truth_threshold = h.mc_truth_threshold;
  A.lsb = ldexp(1.0, 16 - h.light_nbit);
  A.sig_f32diff = is_f4 && ctx->numba_f32;
  hipLaunchKernelGGL(light_digitize_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, st, A, d_digit, d_dtid, d_dtph);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}

// ======================================================================================================================
// C ABI, host-buffer forms (one per reference function) and the device-resident forms working on the last
// ldsim_dev_sum_light / ldsim_dev_light_response result
// ======================================================================================================================
extern "C" int ldsim_stat_fluctuations(ldsim_ctx* ctx, const float* light_sample_inc, int32_t n_det, int32_t n_ticks,
                                       float* light_sample_inc_disc) {
  LDSIM_ENTER(ctx);
  LW_NEED(ctx && light_sample_inc && light_sample_inc_disc && n_det >= 0 && n_ticks >= 0, "bad argument");
  LW_NEED(ctx->h_consts.light_tick_size > 0, "light constants not set");
  HIPCHK(hipSetDevice(ctx->device));
  const size_t n = (size_t)n_det * n_ticks;
  if (n == 0) return 0;
  LTmp din, dout;
  LW_CK(din.alloc(n * 4));
  LW_CK(dout.alloc(n * 4));
  HIPCHK(hipMemcpyAsync(din.p, light_sample_inc, n * 4, hipMemcpyHostToDevice, ctx->stream));
  LW_CK(light_launch_stat_fluct(ctx, din.as<float>(), dout.as<float>(), (int64_t)n));
  HIPCHK(hipMemcpyAsync(light_sample_inc_disc, dout.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

extern "C" int ldsim_light_triggers(ldsim_ctx* ctx, const float* signal, int32_t n_det, int32_t n_ticks,
                                    const double* group_threshold, int32_t n_grp, const int32_t* row_module, int32_t n_mod,
                                    int64_t* trigger_idx, int32_t* trigger_module, int64_t capacity, int64_t* n_trig) {
  LDSIM_ENTER(ctx);
  LW_NEED(ctx && group_threshold && row_module && n_trig && n_det >= 0 && n_ticks >= 0 && n_grp >= 0 && n_mod >= 0 &&
              capacity >= 0 && (capacity == 0 || (trigger_idx && trigger_module)), "bad argument");
  HIPCHK(hipSetDevice(ctx->device));
  LTmp dsig;
  const float* d_signal;
  if (signal) {
    LW_CK(dsig.alloc((size_t)n_det * n_ticks * 4));
    HIPCHK(hipMemcpyAsync(dsig.p, signal, (size_t)n_det * n_ticks * 4, hipMemcpyHostToDevice, ctx->stream));
    d_signal = dsig.as<float>();
  } else {
    LW_NEED(ctx->light_resp_valid && ctx->light_sum_ndet == n_det && ctx->light_sum_nticks == n_ticks,
            "no resident detector response of that shape (ldsim_dev_light_response)");
    d_signal = (const float*)ctx->light_resp.p;
  }
  return light_triggers_run(ctx, d_signal, n_det, n_ticks, group_threshold, n_grp, row_module, n_mod, trigger_idx,
                            trigger_module, capacity, n_trig);
}

extern "C" int ldsim_light_detector_noise(ldsim_ctx* ctx, int32_t n_rows, int32_t n_samples, const double* spectrum,
                                          int32_t nbins, const double* phases, double* noise) {
  LDSIM_ENTER(ctx);
  LW_NEED(ctx && spectrum && noise && n_rows >= 0 && n_samples >= 0 && nbins >= 0, "bad argument");
  HIPCHK(hipSetDevice(ctx->device));
  if (n_rows == 0 || n_samples == 0) return 0;
  if (!phases && !ctx->rng_seeded) {
    ldsim_set_error("gen_light_detector_noise draws random phases: call ldsim_rng_seed first");
    return LDSIM_ESTATE;
  }
  std::vector<int32_t> chan((size_t)n_rows), ticks((size_t)n_samples);
  for (int i = 0; i < n_rows; i++) chan[(size_t)i] = i;
  for (int i = 0; i < n_samples; i++) ticks[(size_t)i] = i;
  LTmp dout;
  LW_CK(dout.alloc((size_t)n_rows * n_samples * 8));
  const uint64_t seed = lw_hash(ctx->rng_seed ^ lw_hash(0x6c69676874ULL + ctx->light_noise_calls++));
  LW_CK(light_noise_run(ctx, spectrum, n_rows, nbins, chan, n_samples, ticks, phases, seed, dout.as<double>()));
  HIPCHK(hipMemcpy(noise, dout.p, (size_t)n_rows * n_samples * 8, hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int ldsim_sim_triggers(ldsim_ctx* ctx, const float* signal, const int32_t* signal_op_channel_idx, int32_t n_det,
                                  int32_t n_ticks, const int64_t* signal_true_track_id, const double* signal_true_photons,
                                  int32_t max_truth, const int64_t* trigger_idx, int32_t n_trig,
                                  const int32_t* trigger_op_channel_idx, int32_t n_det_trig, int32_t digit_samples,
                                  const double* light_det_noise, int32_t n_noise_channels, int32_t n_noise_bins,
                                  const double* phases_signal, const double* phases_missing, double* digit_signal,
                                  int64_t* digit_true_track_id, double* digit_true_photons) {
  LDSIM_ENTER(ctx);
  LW_NEED(ctx && n_det >= 0 && n_ticks >= 0 && max_truth >= 0 && n_trig >= 0 && n_det_trig >= 0 && digit_samples >= 0,
          "bad argument");
  LW_NEED(n_det == 0 || signal_op_channel_idx, "signal_op_channel_idx missing");
  LW_NEED(n_trig == 0 || (trigger_idx && trigger_op_channel_idx && digit_signal), "trigger arrays missing");
  HIPCHK(hipSetDevice(ctx->device));
  const size_t n_out = (size_t)n_trig * n_det_trig * digit_samples;
  if (n_out == 0) return 0;
  const size_t bo = (size_t)n_det * n_ticks;
  hipStream_t st = ctx->stream;
  LTmp dsig, dtid, dtph, dd, ddt, ddp;
  const float* d_signal;
  const int64_t* d_tid = nullptr;
  const double* d_tph = nullptr;
  if (signal) {
    LW_NEED(max_truth == 0 || (signal_true_track_id && signal_true_photons), "truth arrays missing");
    LW_CK(dsig.alloc(bo * 4));
    HIPCHK(hipMemcpyAsync(dsig.p, signal, bo * 4, hipMemcpyHostToDevice, st));
    d_signal = dsig.as<float>();
    if (max_truth) {
      LW_CK(dtid.alloc(bo * max_truth * 8));
      LW_CK(dtph.alloc(bo * max_truth * 8));
      HIPCHK(hipMemcpyAsync(dtid.p, signal_true_track_id, bo * max_truth * 8, hipMemcpyHostToDevice, st));
      HIPCHK(hipMemcpyAsync(dtph.p, signal_true_photons, bo * max_truth * 8, hipMemcpyHostToDevice, st));
      d_tid = dtid.as<int64_t>();
      d_tph = dtph.as<double>();
    }
  } else {                                                  // the resident detector response
    LW_NEED(ctx->light_resp_valid && ctx->light_sum_ndet == n_det && ctx->light_sum_nticks == n_ticks &&
                ctx->light_sum_truth == max_truth,
            "no resident detector response of that shape (ldsim_dev_light_response)");
    d_signal = (const float*)ctx->light_resp.p;
    d_tid = (const int64_t*)ctx->light_resp_tid.p;
    d_tph = (const double*)ctx->light_resp_tph.p;
  }
  LW_NEED(max_truth == 0 || (digit_true_track_id && digit_true_photons), "output truth arrays missing");
  LW_CK(dd.alloc(n_out * 8));
  if (max_truth) {
    LW_CK(ddt.alloc(n_out * max_truth * 8));
    LW_CK(ddp.alloc(n_out * max_truth * 8));
  }
  LW_CK(sim_triggers_run(ctx, d_signal, signal_op_channel_idx, n_det, n_ticks, d_tid, d_tph, max_truth, trigger_idx, n_trig,
                         trigger_op_channel_idx, n_det_trig, digit_samples, light_det_noise, n_noise_channels, n_noise_bins,
                         phases_signal, phases_missing, dd.as<double>(), ddt.as<int64_t>(), ddp.as<double>()));
  HIPCHK(hipMemcpyAsync(digit_signal, dd.p, n_out * 8, hipMemcpyDeviceToHost, st));
  if (max_truth) {
    HIPCHK(hipMemcpyAsync(digit_true_track_id, ddt.p, n_out * max_truth * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(digit_true_photons, ddp.p, n_out * max_truth * 8, hipMemcpyDeviceToHost, st));
  }
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}
