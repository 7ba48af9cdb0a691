// kernels_fee.hip -- a13-a16: per-pixel summation, LArPix self-trigger / ADC scan, digitisation.
// Reference: larndsim/detsim.py:468-607 (sum_pixel_signals, get_track_pixel_map2),
//            larndsim/fee.py:499-655 (digitize, get_adc_values).
//
// Chain form: one workgroup per unique (batch, pixel).  The pairs of a pixel are contiguous in the
// sorted pair list in exactly the slot order get_track_pixel_map2 produces (ring distance, then
// segment index), so the pixel's waveform is a register-owned sum over <= 50 compact f32 waveforms
// (no atomics, no [U][N_t][50] slab), the trigger scan is a wave-64 prefix scan + ballot over
// 64-tick chunks, and the per-hit backtracking fractions are wave reductions over the hit spans.
#include <algorithm>
#include "ldsim_args.h"

#define FEE_THREADS 256
#define FEE_SPAN 512     // ticks of LDS of the one-wave instantiation of pixel_adc_kernel
#define NT_MAX 4096   // max len(TIME_TICKS) held in LDS
#define A_MAX 64      // max MAX_ADC_VALUES
#define M_MAX 64      // max MAX_TRACKS_PER_PIXEL handled by the chain kernel


// Wave-64 scans with DPP (row shifts inside the 16-lane rows, then row_bcast:15 / :31 across rows: the GFX9 controls gfx950
// keeps) instead of shuffles through the LDS crossbar, whose latency the trigger scan paid six times per 64-tick chunk.
// Lanes without a source read 0 (bound_ctrl), rows masked out keep the `old` operand 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_or_zero(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_incl_scan(double v, int lane) {
  v += dpp_or_zero<0x111, 0xF>(v);       // row_shr:1
  v += dpp_or_zero<0x112, 0xF>(v);       // row_shr:2
  v += dpp_or_zero<0x114, 0xF>(v);       // row_shr:4
  v += dpp_or_zero<0x118, 0xF>(v);       // row_shr:8   -> inclusive scan inside every row of 16
  v += dpp_or_zero<0x142, 0xA>(v);       // row_bcast:15 into rows 1 and 3: + total of the row before
  v += dpp_or_zero<0x143, 0xC>(v);       // row_bcast:31 into rows 2 and 3: + total of lanes 0..31
  return v;
}
// value of lane l (wave-uniform index) in every lane
__device__ __forceinline__ double lane_bcast(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wave_sum(double v) { return lane_bcast(wave_incl_scan(v, 0), 63); }

// q(ic): buffer-convolved charge of tick ic (fee.py:566-579), taps limited by last_reset and N_t.  S holds the ticks
// [s_lo, s_hi) of the pixel's waveform (element 0 = tick s_lo); it is zero outside them.
__device__ __forceinline__ double conv_q(const double* S, int NT, int ic, int last_reset, int ntap, const double* wtap,
                                         double dt, bool has_rt, int s_lo, int s_hi) {
  double q = 0;
  if (has_rt) {
    int cs = ic - ntap;
    if (cs < last_reset) cs = last_reset;
    if (cs < s_lo) cs = s_lo;
    int ce = ic + 1 < NT ? ic + 1 : NT;
    if (ce > s_hi) ce = s_hi;
    for (int jc = cs; jc < ce; jc++) q += S[jc - s_lo] * dt * wtap[ic - jc];
  } else if (ic < NT && ic >= s_lo && ic < s_hi) {
    q = S[ic - s_lo] * dt;
  }
  return q;
}

template <class K>
__device__ __forceinline__ double digitize_one(const K* c, double q, double gain) {
  const double mV = 1e-3 * (1e-6 * 1.0);
  double v = q * gain + c->v_pedestal * mV - c->v_cm * mV;
  v = v > 0 ? v : 0;
  v = rint(v * c->adc_counts / (c->v_ref * mV - c->v_cm * mV));
  double top = c->adc_counts - 1;
  return v < top ? v : top;
}

// ---- the scan itself, shared by the chain and the dense (materialising) kernels -------------------------------
// Runs on wave 0.  S = summed pixel waveform in LDS.  Writes hits into LDS arrays; returns n_hits.
struct HitRec {
  int lr, b;      // span [lr, b] of ticks whose charge entered the hit
  double q;       // adc_list value: integrated charge + the noise terms
  double tq;      // true_q: integrated charge alone, normalises the backtracking fractions (fee.py:633-635)
  double tick;    // adc_ticks_list value
};

// z: this pixel's normal draws in stream order (fee_noise_kernel), or NULL when every noise charge is 0.  The reference
// draws one normal before the loop (reset noise, fee.py:557), two at every pass of the loop (uncorrelated + discriminator
// noise, :583-584, also on busy ticks), two after an integration (:616-617) and one at every reset (:621,649); *n_draws
// returns how many were consumed.  t_stop = time_ticks[-1] of linspace(0, t_stop, NT + 1) (cli/simulate_pixels.py:1072).
// [t_first, t_end): the ticks where S can be non-zero.  Without noise and with a positive threshold the scan's state does not
// change on ticks whose charge is zero while the ADC is idle and the sum below threshold, so it starts at t_first and stops
// once past t_end + ntap with the ADC idle (with noise every tick draws its normals and may trigger: all ticks are walked).
template <class K>
__device__ int adc_scan(const K* c, const double* S, int NT, double t_stop, double thr, double time_padding,
                        int lane, HitRec* hits /* LDS, [A] */, const double* wtap, int ntap, const float* __restrict__ z,
                        int* n_draws, int t_first = 0, int t_end = 1 << 30, int s_lo = 0, int s_hi = 1 << 30) {
  const double dt = c->time_sampling;
  const bool has_rt = c->buffer_risetime > 0;
  const int A = c->max_adc_values;
  const int interval = (int)py_round((3 * c->clock_cycle + c->adc_hold_delay * c->clock_cycle) / dt);
  const int reset_ticks = (int)py_round(c->reset_cycles * c->clock_cycle / dt);
  const int busy_ticks = (int)py_round(c->adc_busy_delay * c->clock_cycle / dt);
  const int n_time_ticks = NT + 1;
  const double tstep = t_stop / (double)NT;
  const double s_reset = c->reset_noise_charge, s_unc = c->uncorrelated_noise_charge, s_disc = c->discriminator_noise;
  int ic = 0, iadc = 0, adc_busy = 0, last_reset = 0, cur = 0;
  double q_sum = 0, true_q = 0;
  if (z) q_sum = (double)z[cur++] * s_reset;
  const bool skip_idle = !z && thr > 0;
  if (skip_idle && t_first > 0) ic = t_first < NT ? t_first : NT;
  const int t_quiet = t_end + ntap;            // from here on q(ic) = 0
  while ((ic < NT || adc_busy > 0) && iadc < A) {
    // one chunk of 64 consecutive ticks
    int my_ic = ic + lane;
    int busy_before = adc_busy - lane;           // busy value seen at the top of this lane's iteration
    bool live = (my_ic < NT) || (busy_before > 0);
    // liveness must be contiguous from lane 0: once a lane is dead the loop has ended
    unsigned long long live_mask = __ballot(live);
    int n_live = (live_mask == ~0ull) ? 64 : __ffsll((long long)~live_mask) - 1;
    double q = (lane < n_live) ? conv_q(S, NT, my_ic, last_reset, ntap, wtap, dt, has_rt, s_lo, s_hi) : 0.0;
    const double incl = wave_incl_scan(q, lane);
    double qs = q_sum + incl;
    double q_noise = 0.0, disc_noise = 0.0;
    if (z && lane < n_live) {
      q_noise = (double)z[cur + 2 * lane] * s_unc;
      disc_noise = (double)z[cur + 2 * lane + 1] * s_disc;
    }
    int busy_after = busy_before > 0 ? busy_before - 1 : 0;
    bool trig = (lane < n_live) && (qs + q_noise >= thr + disc_noise) && (busy_after == 0);
    unsigned long long tm = __ballot(trig);
    if (tm == 0) {
      if (z) cur += 2 * n_live;
      if (n_live < 64) break;
      q_sum = lane_bcast(qs, 63);
      true_q += lane_bcast(incl, 63);
      ic += 64;
      adc_busy = adc_busy > 64 ? adc_busy - 64 : 0;
      if (skip_idle && adc_busy == 0 && ic >= t_quiet) break;      // (lane 63 was idle and below threshold: nothing can follow)
      continue;
    }
    int f = __ffsll((long long)tm) - 1;
    q_sum = lane_bcast(qs, f);
    true_q += lane_bcast(incl, f);
    if (z) cur += 2 * (f + 1);
    int ict = ic + f;
    int integrate_end = ict + interval;
    // integrate the next `interval` ticks (fee.py:590-614)
    double qi = 0;
    for (int base = ict + 1; base <= integrate_end; base += 64) {
      int t = base + lane;
      double v = (t <= integrate_end) ? conv_q(S, NT, t, last_reset, ntap, wtap, dt, has_rt, s_lo, s_hi) : 0.0;
      qi += wave_sum(v);
    }
    q_sum += qi;
    true_q += qi;
    ic = integrate_end + 1;
    double adc = q_sum, disc2 = 0.0;
    if (z) {
      adc = q_sum + (double)z[cur] * s_unc;
      disc2 = (double)z[cur + 1] * s_disc;
      cur += 2;
    }
    if (adc < thr + disc2) {  // fee.py:619-628
      ic += reset_ticks;
      q_sum = z ? (double)z[cur++] * s_reset : 0.0;
      true_q = 0;
      last_reset = ic;
      adc_busy = 0;
      continue;
    }
    if (lane == 0) {
      int crossing = ic < n_time_ticks - 1 ? ic : n_time_ticks - 1;
      int post = ic - crossing > 0 ? ic - crossing : 0;
      double tt = (crossing == NT) ? t_stop : crossing * tstep;
      hits[iadc].lr = last_reset;
      hits[iadc].b = integrate_end;
      hits[iadc].q = adc;
      hits[iadc].tq = true_q;
      hits[iadc].tick = tt + time_padding - 2 + post;
    }
    ic += reset_ticks;
    last_reset = ic;
    adc_busy = busy_ticks;
    q_sum = z ? (double)z[cur++] * s_reset : 0.0;
    true_q = 0;
    iadc++;
  }
  *n_draws = cur;
  return iadc;
}

// ---- chain kernel ----------------------------------------------------------------------------------------------------
// Two instantiations.  THREADS = 256, the whole tick axis in LDS (16-26 KB: eight pixels per CU), for every pixel when noise is on
// (the scan then walks every tick) and for pixels whose slots' windows span more than FEE_SPAN ticks.  THREADS = 64 -- a wave per
// pixel, FEE_SPAN ticks of LDS starting at the pixel's first written tick -- for the others, over a list (fee_span_kernel): the
// kernel is a chain of dependent trips to memory per pixel (pair range -> keys -> slots' starts and windows -> rows), so what
// counts is pixels in flight, and a CU holds 2048 threads: 8 pixels of 256 threads, 19 of 64 at 8.4 KB each.
template <int THREADS>
__device__ __forceinline__ void pixel_adc_body(const FeeArgs& F, const int64_t u, const int32_t* __restrict__ span /* [U][2] or NULL */,
                                               const int s_cap) {
  const FeeK* c = &F.k;          // (kernel arguments: scalar registers, no loads)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int NT = c->n_time_ticks;
  const int A = c->max_adc_values, M = c->max_tracks_per_pixel;
  const double dt = c->time_sampling, rt = c->buffer_risetime;

  extern __shared__ double S[];          // [n_time_ticks]: sized at launch, so LDS (not VGPRs) stops at 8 pixels per CU
  __shared__ HitRec hits[A_MAX];
  __shared__ int s_start[M_MAX], s_w0[M_MAX], s_w1[M_MAX];
  __shared__ int64_t s_row[M_MAX];
  __shared__ double wtap[64], G[64];
  __shared__ int s_nh, s_trange[2];

  const int64_t p0 = F.uoff[u], p1 = F.uoff[u + 1];
  // slots: pairs whose ring code is valid (key low nibble != 15), at most M (detsim.py:582-607).  Keys are sorted, the
  // invalid-distance pairs (nibble 15) sit at the end of the group: their count by one pass of parallel loads (a bisection
  // was four dependent round trips to memory per pixel)
  __shared__ int s_nvalid;
  if (tid == 0) s_nvalid = 0;
  __syncthreads();
  {
    int cnt = 0;
    for (int64_t p = p0 + tid; p < p1; p += THREADS) cnt += (F.pair_key[p] & 15ull) != 15ull;
    if (cnt) atomicAdd(&s_nvalid, cnt);
  }
  __syncthreads();
  const int n_valid = s_nvalid;
  const int n_slots = n_valid < M ? n_valid : M;
  const bool overflow = (p1 - p0) > n_slots;
  const int ubatch = F.ubatch[u];
  const int bfirst = F.batch_first[ubatch - F.batch0];
  if (tid < n_slots) {
    int32_t v = F.pair_val[p0 + tid];
    int r = v / F.P;
    s_start[tid] = (int)py_round(F.track_starts[r] / dt);   // detsim.py:506
    s_row[tid] = p0 + tid;
    s_w0[tid] = F.win ? F.win[2 * (p0 + tid)] : 0;         // the ticks of the row tracks_current wrote
    s_w1[tid] = F.win ? F.win[2 * (p0 + tid) + 1] : F.T;
    F.tpm[u * M + tid] = r - bfirst;
  } else if (tid < M) {
    F.tpm[u * M + tid] = -1;
  }
  const int ntap = rt > 0 ? (int)ceil(10 * rt / dt) : 0;  // floor(ic - 10*rt/dt) == ic - ceil(10*rt/dt)
  if (tid <= ntap && rt > 0) {
    // w(d) = exp(-d*dt/rt) * (1 - exp(-dt/rt))   (fee.py:569)
    wtap[tid] = exp((-tid) * dt / rt) * (1 - exp(-dt / rt));
  }
  __syncthreads();
  if (tid == 0) {
    // G[n] = dt * sum_{d=0..n} w(d): weight of a tick n ticks before the end of a hit span
    double acc = 0;
    for (int d = 0; d <= ntap; d++) {
      acc += (rt > 0 ? wtap[d] : 1.0) * dt;
      G[d] = acc;
    }
  }
  // ---- summed waveform: each thread owns ticks tid, tid+256, ... and adds the slots' rows in slot order, each over the ticks
  // its window puts on the pixel's time axis (detsim.py:516-520) ------------------------------------------------------------
  // the ticks S holds: all of them, or the pixel's window from the set-up pass (every slot's ticks lie inside it)
  const int s_lo = span ? span[2 * u] : 0;
  const int s_hi = span ? min(s_lo + s_cap, NT) : NT;
  for (int t = tid; t < s_hi - s_lo; t += THREADS) S[t] = 0;
  if (tid == 0) {          // the ticks the slots' windows cover
    int t_lo = NT, t_hi = 0;
    for (int k = 0; k < n_slots; k++) {
      const int lo = max(s_start[k] + s_w0[k], 0), hi = min(s_start[k] + s_w1[k], NT);
      if (hi > lo) { t_lo = min(t_lo, lo); t_hi = max(t_hi, hi); }
    }
    s_trange[0] = t_lo;
    s_trange[1] = t_hi;
  }
  for (int k = 0; k < n_slots && !(F.debug & 0x10000); k++) {
    const int st = s_start[k];
    const int lo = max(st + s_w0[k], 0), hi = min(st + s_w1[k], NT);
    const float* wf = F.waves + s_row[k] * (int64_t)F.T - st;
    for (int t = lo + ((tid - lo) & (THREADS - 1)); t < hi; t += THREADS) S[t - s_lo] += (double)wf[t];
  }
  __syncthreads();
  // ---- trigger scan on wave 0 ----------------------------------------------------------------------------------------
  if (wv == 0) {
    const double thr = F.thr_table ? F.thr_table[F.upix[u]] : F.threshold;
    int nd = 0;
    int nh = (F.debug & 0x20000) ? 0 : adc_scan(c, S, NT, 1 * c->time_interval1, thr, F.time_padding, lane, hits, wtap, ntap,
                      F.noise_z ? F.noise_z + u * (int64_t)F.noise_nd : nullptr, &nd, s_trange[0], s_trange[1], s_lo, s_hi);
    if (lane == 0) {
      s_nh = nh;
      if (F.n_draws) F.n_draws[u] = nd;
    }
  }
  __syncthreads();
  const int nh = s_nh;
  const double gain = F.gain_table ? F.gain_table[F.upix[u]] : c->gain * (1e-3 * (1e-6 * 1.0)) / 1.0;   // GAIN * mV / e
  for (int h = tid; h < A; h += THREADS) {
    double q = h < nh ? hits[h].q : 0.0;
    F.adc_list[u * A + h] = q;
    F.adc_ticks[u * A + h] = h < nh ? hits[h].tick : 0.0;
    F.adc_digit[u * A + h] = digitize_one(c, q, gain);
  }
  if (tid == 0) {
    F.hit_count[u] = nh;
    if (overflow) stat_add(F.counters, 2, 1ull);
    if (nh) stat_add(F.counters, 3, (unsigned long long)nh);
  }
  // ---- backtracking fractions (fee.py:572-573, 633-635): sum_jc sig_k[jc]*G[min(ntap, b-jc)] / true_q ------------
  if (F.fractions && !(F.debug & 0x40000)) {
    double* fr = F.fractions + u * (int64_t)A * M;       // zero on entry (one memset of the whole array by the launcher)
    for (int k = wv; k < n_slots; k += THREADS / 64) {      // (a wave per slot)
      const float* wf = F.waves + s_row[k] * (int64_t)F.T;
      const int st = s_start[k];
      for (int h = 0; h < nh; h++) {
        int lr = hits[h].lr, b = hits[h].b;
        int hi = b < NT - 1 ? b : NT - 1;
        // (the ticks of the hit span the slot's window covers: a first hit's span starts at tick 0, a thousand ticks before
        // any window; same terms in the same lane-strided order)
        const int j0 = max(lr, st + s_w0[k]), j1 = min(hi, st + s_w1[k] - 1);
        double acc = 0;
        for (int jc = j0 + ((lr + lane - j0) & 63); jc <= j1; jc += 64) {
          int it = jc - st;
          int d = b - jc;
          acc += (double)wf[it] * (rt > 0 ? G[d < ntap ? d : ntap] : dt);
        }
        acc = wave_sum(acc);
        if (lane == 0) fr[h * M + k] = hits[h].tq > 0 ? acc / hits[h].tq : acc;
      }
    }
  }
}

// one workgroup per unique pixel
template <int THREADS>
__global__ void __launch_bounds__(THREADS) pixel_adc_kernel(FeeArgs F) {
  if ((int64_t)blockIdx.x < F.U) pixel_adc_body<THREADS>(F, (int64_t)blockIdx.x, nullptr, 0);
}
// a fixed grid walks a device-built list of pixels (its length stays on the device: no host round trip to size the launch)
template <int THREADS>
__global__ void __launch_bounds__(THREADS) pixel_adc_list_kernel(FeeArgs F, const int32_t* __restrict__ list,
                                                                const unsigned long long* __restrict__ count,
                                                                const int32_t* __restrict__ span, int s_cap) {
  const int64_t n = (int64_t)*count;
  for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
    pixel_adc_body<THREADS>(F, (int64_t)list[i], span, s_cap);
    __syncthreads();          // (the next pixel reuses the workgroup's LDS)
  }
}

// zero every entry of fractions [U][A][M] that pixel_adc_kernel did not write: rows past the pixel's last hit, slots past its last track
__global__ void __launch_bounds__(256) fee_clear_fractions_kernel(int64_t U, const int32_t* __restrict__ hit_count,
                                                                 const int64_t* __restrict__ tpm, int A, int M, double* __restrict__ fr) {
  const int64_t u = blockIdx.x;
  if (u >= U) return;
  __shared__ int s_ns;
  if (threadIdx.x == 0) s_ns = 0;
  __syncthreads();
  if ((int)threadIdx.x < M && tpm[u * M + threadIdx.x] != -1) atomicAdd(&s_ns, 1);      // (filled slots come first)
  __syncthreads();
  const int nh = hit_count[u], ns = s_ns;
  double* row = fr + u * (int64_t)A * M;
  for (int h = 0; h < A; h++)
    for (int k = threadIdx.x; k < M; k += 256)
      if (!(h < nh && k < ns)) row[h * M + k] = 0.0;
}

extern "C++" int fee_clear_unwritten_fractions(ldsim_ctx* ctx, int64_t U, const int32_t* hit_count, const int64_t* tpm, double* fr) {
  if (U == 0) return 0;
  const LdsimConsts& h = ctx->h_consts;
  hipLaunchKernelGGL(fee_clear_fractions_kernel, dim3((unsigned)U), dim3(256), 0, ctx->stream, U, hit_count, tpm,
                     h.max_adc_values, h.max_tracks_per_pixel, fr);
  HIPCHK(hipGetLastError());
  return 0;
}

// The ticks a pixel's slots write, [t_lo, t_hi), from the same expressions as pixel_adc_kernel; the pixels whose window fits
// FEE_SPAN ticks go to list 0, the others to list 1 (one atomic per wave and list).
__global__ void __launch_bounds__(256) fee_span_kernel(FeeArgs F, int32_t* __restrict__ span, int32_t* __restrict__ lists /* [2][U] */,
                                                       unsigned long long* __restrict__ counts /* [2] */) {
  const FeeK* c = &F.k;
  const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  int cls = -1;
  if (u < F.U) {
    const int NT = c->n_time_ticks, M = c->max_tracks_per_pixel;
    const double dt = c->time_sampling;
    const int64_t p0 = F.uoff[u], p1 = F.uoff[u + 1];
    int n_slots = 0;
    for (int64_t p = p0; p < p1 && n_slots < M; p++) n_slots += (F.pair_key[p] & 15ull) != 15ull;      // (valid ones come first)
    int t_lo = NT, t_hi = 0;
    for (int k = 0; k < n_slots; k++) {
      const int r = F.pair_val[p0 + k] / F.P;
      const int st = (int)py_round(F.track_starts[r] / dt);
      const int w0 = F.win ? F.win[2 * (p0 + k)] : 0, w1 = F.win ? F.win[2 * (p0 + k) + 1] : F.T;
      const int lo = max(st + w0, 0), hi = min(st + w1, NT);
      if (hi > lo) { t_lo = min(t_lo, lo); t_hi = max(t_hi, hi); }
    }
    if (t_hi <= t_lo) { t_lo = 0; t_hi = 0; }
    span[2 * u] = t_lo;
    span[2 * u + 1] = t_hi;
    cls = (t_hi - t_lo <= FEE_SPAN) ? 0 : 1;
  }
  for (int cc = 0; cc < 2; cc++) {
    const unsigned long long m = __ballot(cls == cc);
    if (!m) continue;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(&counts[cc], (unsigned long long)__popcll(m));
    base = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((int)base);
    if (cls == cc) lists[(int64_t)cc * F.U + (int64_t)base + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)u;
  }
}

extern "C++" int fee_launch_chain(ldsim_ctx* ctx, const FeeArgs& F) {
  if (F.U == 0) return 0;
  const LdsimConsts& h = ctx->h_consts;
  if (h.n_time_ticks > NT_MAX || h.max_adc_values > A_MAX || h.max_tracks_per_pixel > M_MAX ||
      (h.buffer_risetime > 0 && 10 * h.buffer_risetime / h.time_sampling > 62)) {
    ldsim_set_error("FEE constants exceed the kernel's static tiles");
    return LDSIM_EINVAL;
  }
  // (the kernel writes the (hit, slot) entries that exist; everything else of `fractions` reads 0 like the reference's array once
  // fee_clear_unwritten_fractions has run: the dense downloads call it -- clearing 12 KB per pixel in every launch cost 0.45 ms per
  // 100 k segments, and the compact download reads the written entries only)
  const size_t full = (size_t)((h.n_time_ticks + 1) & ~1) * 8;
  const bool skip_idle = !F.noise_z && ((F.thr_table != nullptr) || F.threshold > 0);
  if (!skip_idle || !F.win || ctx->fee_one_class || h.n_time_ticks <= FEE_SPAN) {
    // (noise: every tick is walked; a threshold table may hold non-positive entries -- the scan decides per pixel, so the table
    // case keeps the windows only when the constant path would; complete rows: no windows to go by)
    hipLaunchKernelGGL(pixel_adc_kernel<FEE_THREADS>, dim3((unsigned)F.U), dim3(FEE_THREADS), full, ctx->stream, F);
    HIPCHK(hipGetLastError());
    return 0;
  }
  int rc;
  if ((rc = ldsim_ensure(ctx, SB_SPAN, (size_t)F.U * 16 + 64))) return rc;      // span [U][2] | lists [2][U] | counts [2]
  int32_t* d_span = (int32_t*)ctx->scratch[SB_SPAN].p;
  int32_t* d_lists = d_span + 2 * F.U;
  unsigned long long* d_counts = (unsigned long long*)((char*)ctx->scratch[SB_SPAN].p + (((size_t)F.U * 16 + 15) & ~(size_t)15));
  HIPCHK(hipMemsetAsync(d_counts, 0, 16, ctx->stream));
  hipLaunchKernelGGL(fee_span_kernel, dim3((unsigned)((F.U + 255) / 256)), dim3(256), 0, ctx->stream, F, d_span, d_lists, d_counts);
  HIPCHK(hipGetLastError());
  // both launches over the whole grid: a workgroup past its list's count leaves at once (the counts stay on the device: no
  // host round trip); the one-wave launch first
  const unsigned g_small = (unsigned)std::min<int64_t>(F.U, 256 * 24), g_big = (unsigned)std::min<int64_t>(F.U, 256 * 8);
  hipLaunchKernelGGL(pixel_adc_list_kernel<64>, dim3(g_small), dim3(64), (size_t)FEE_SPAN * 8, ctx->stream, F, d_lists,
                     d_counts, d_span, FEE_SPAN);
  HIPCHK(hipGetLastError());
  hipLaunchKernelGGL(pixel_adc_list_kernel<FEE_THREADS>, dim3(g_big), dim3(FEE_THREADS), full, ctx->stream, F,
                     d_lists + F.U, d_counts + 1, (const int32_t*)nullptr, 0);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---- materialising forms (parity API) ----------------------------------------------------------------------------------
// get_track_pixel_map2, literal (detsim.py:564-607): one thread per unique pixel
__global__ void track_pixel_map_kernel(int64_t* map, const int32_t* unique_pix, int64_t U, const int32_t* pixels,
                                       const int32_t* dist, int64_t S, int P, int max_distance, int M) {
  int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= U) return;
  int32_t upix = unique_pix[u];
  int64_t* row = map + u * M;
  for (int target = 0; target < max_distance; target++)
    for (int64_t itrk = 0; itrk < S; itrk++)
      for (int ipix = 0; ipix < P; ipix++) {
        if (upix != pixels[itrk * P + ipix]) continue;
        if (dist[itrk * P + ipix] == target) {
          int imap = 0;
          while (imap < M) {
            if (row[imap] == itrk) { imap = -1; break; }
            if (row[imap] == -1) break;
            imap++;
          }
          if (imap >= 0 && imap < M) row[imap] = itrk;
        }
        break;
      }
}

// sum_pixel_signals, literal incl. f64 atomics (detsim.py:468-527): thread per (itrk, ipix, itick)
__global__ void sum_pixel_signals_kernel(double* pixels_signals, const float* signals, const double* track_starts,
                                         const int64_t* pim, const int64_t* tpm, double* pts, double* overflow,
                                         int64_t S, int P, int T, int NT, int M, double dt) {
  int64_t pairi = blockIdx.x;
  int64_t itrk = pairi / P;
  int64_t pidx = pim[pairi];
  if (pidx < 0) return;
  int start_tick = (int)py_round(track_starts[itrk] / dt);
  int counter = -99;
  for (int k = 0; k < M; k++)
    if (itrk == tpm[pidx * M + k]) { counter = k; break; }
  if (counter < 0) {
    if (threadIdx.x == 0) overflow[pidx] = 1;
    return;
  }
  for (int itick = threadIdx.x; itick < T; itick += blockDim.x) {
    int itime = start_tick + itick;
    if (itime < NT && itime > -1) {
      double v = signals[pairi * T + itick];
      atomicAdd(&pixels_signals[pidx * NT + itime], v);
      if (pts) atomicAdd(&pts[(pidx * NT + itime) * M + counter], v);
    }
  }
}

// get_adc_values on dense arrays (fee.py:517-655): workgroup per pixel, same scan as the chain
__global__ void __launch_bounds__(FEE_THREADS) adc_dense_kernel(const LdsimConsts* c, const double* pixels_signals,
                                                                const double* pts, int64_t U, int NT, int M,
                                                                const double* thresholds, double time_padding,
                                                                double t_stop, const float* noise_z, int noise_nd,
                                                                int32_t* n_draws, double* adc_list, double* adc_ticks,
                                                                double* fractions) {
  const int64_t u = blockIdx.x;
  if (u >= U) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int A = c->max_adc_values;
  const double dt = c->time_sampling, rt = c->buffer_risetime;
  __shared__ double S[NT_MAX];
  __shared__ HitRec hits[A_MAX];
  __shared__ double wtap[64], G[64];
  __shared__ int s_nh;
  const int ntap = rt > 0 ? (int)ceil(10 * rt / dt) : 0;  // floor(ic - 10*rt/dt) == ic - ceil(10*rt/dt)
  if (tid <= ntap && rt > 0) wtap[tid] = exp((-tid) * dt / rt) * (1 - exp(-dt / rt));
  for (int t = tid; t < NT; t += FEE_THREADS) S[t] = pixels_signals[u * NT + t];
  __syncthreads();
  if (tid == 0) {
    double acc = 0;
    for (int d = 0; d <= ntap; d++) {
      acc += (rt > 0 ? wtap[d] : 1.0) * dt;
      G[d] = acc;
    }
  }
  if (wv == 0) {
    int nd = 0;
    int nh = adc_scan(c, S, NT, t_stop, thresholds[u], time_padding, lane, hits, wtap, ntap,
                      noise_z ? noise_z + u * (int64_t)noise_nd : nullptr, &nd);
    if (lane == 0) {
      s_nh = nh;
      if (n_draws) n_draws[u] = nd;
    }
  }
  __syncthreads();
  const int nh = s_nh;
  for (int h = tid; h < A; h += FEE_THREADS) {
    adc_list[u * A + h] = h < nh ? hits[h].q : 0.0;
    adc_ticks[u * A + h] = h < nh ? hits[h].tick : 0.0;
  }
  if (fractions && pts) {
    double* fr = fractions + u * (int64_t)A * M;
    for (int i = tid; i < A * M; i += FEE_THREADS) fr[i] = 0;
    __syncthreads();
    for (int k = wv; k < M; k += FEE_THREADS / 64)
      for (int h = 0; h < nh; h++) {
        int lr = hits[h].lr, b = hits[h].b;
        int hi = b < NT - 1 ? b : NT - 1;
        double acc = 0;
        for (int jc = lr + lane; jc <= hi; jc += 64) {
          int d = b - jc;
          acc += pts[(u * NT + jc) * M + k] * (rt > 0 ? G[d < ntap ? d : ntap] : dt);
        }
        acc = wave_sum(acc);
        if (lane == 0) fr[h * M + k] = hits[h].tq > 0 ? acc / hits[h].tq : acc;
      }
  }
}

__global__ void digitize_kernel(const LdsimConsts* c, const double* q, const double* gain_list, double* out, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double gain = gain_list ? gain_list[i] : c->gain * (1e-3 * (1e-6 * 1.0)) / 1.0;
  out[i] = digitize_one(c, q[i], gain);
}

extern "C++" {
int fee_launch_track_pixel_map(ldsim_ctx* ctx, int64_t* map, const int32_t* upix, int64_t U, const int32_t* pixels,
                               const int32_t* dist, int64_t S, int P, int max_distance, int M) {
  if (U == 0) return 0;
  hipLaunchKernelGGL(track_pixel_map_kernel, dim3((unsigned)((U + 63) / 64)), dim3(64), 0, ctx->stream, map, upix, U,
                     pixels, dist, S, P, max_distance, M);
  HIPCHK(hipGetLastError());
  return 0;
}
int fee_launch_sum_pixel_signals(ldsim_ctx* ctx, double* ps, const float* signals, const double* starts,
                                 const int64_t* pim, const int64_t* tpm, double* pts, double* ovf, int64_t S, int P,
                                 int T, int NT, int M) {
  if (S * P == 0) return 0;
  hipLaunchKernelGGL(sum_pixel_signals_kernel, dim3((unsigned)(S * P)), dim3(256), 0, ctx->stream, ps, signals, starts,
                     pim, tpm, pts, ovf, S, P, T, NT, M, ctx->h_consts.time_sampling);
  HIPCHK(hipGetLastError());
  return 0;
}
int fee_launch_adc_dense(ldsim_ctx* ctx, const double* ps, const double* pts, int64_t U, int NT, int M,
                         const double* thr, double time_padding, double t_stop, const float* noise_z, int noise_nd,
                         int32_t* n_draws, double* adc, double* ticks, double* frac) {
  if (U == 0) return 0;
  const LdsimConsts& h = ctx->h_consts;
  if (NT > NT_MAX || h.max_adc_values > A_MAX || (h.buffer_risetime > 0 && 10 * h.buffer_risetime / h.time_sampling > 62)) {
    ldsim_set_error("FEE constants exceed the kernel's static tiles");
    return LDSIM_EINVAL;
  }
  hipLaunchKernelGGL(adc_dense_kernel, dim3((unsigned)U), dim3(FEE_THREADS), 0, ctx->stream, ctx->d_consts, ps, pts, U,
                     NT, M, thr, time_padding, t_stop, noise_z, noise_nd, n_draws, adc, ticks, frac);
  HIPCHK(hipGetLastError());
  return 0;
}
int fee_launch_digitize(ldsim_ctx* ctx, const double* q, const double* gain, double* out, int64_t n) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(digitize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_consts, q,
                     gain, out, n);
  HIPCHK(hipGetLastError());
  return 0;
}
}
