// kernels_light_response.hip -- light waveform response:
//   light_sim.calc_scintillation_effect        larndsim/light_sim.py:148-184  (weights: scintillation_model :131-146)
//   light_sim.calc_light_detector_response     larndsim/light_sim.py:303-337  (weights: sipm_response_model :274-300)
// Both are causal convolutions along the tick axis of one detector row:  out[d][i] += sum_{j = max(i-C,0)}^{i} w(i-j) x[d][j],
// C = ceil((LIGHT_WINDOW[1] - LIGHT_WINDOW[0]) / LIGHT_TICK_SIZE), the output an f4 array the reference updates term by
// term -- the f4 store after every term, in ascending j, is part of the result and is kept.  w depends on i-j only, so the
// host tabulates it once per call with the reference's expressions (ldsim_abi.hip) and the kernels never call exp/sin.
//
// One thread per (row, tick), a workgroup = 256 consecutive ticks of one row.  The j range of the workgroup is walked in
// chunks of JCHUNK: the chunk of x and the weights it can meet are staged in LDS; inside a chunk all lanes read the same
// x[j] (broadcast) and consecutive weights (conflict-free).  Truth slots (optional) follow the reference literally; a
// per-tick bound on the slots' photons (light_truth_max_kernel) lets a pair (i, j) skip its slot walk when none can pass.
#include "ldsim_dev.h"
#include "wave_ops.h"

#define LR_THREADS 256
#define JCHUNK 1024

template <bool RESPONSE>
__global__ void __launch_bounds__(LR_THREADS) light_conv_kernel(
    const float* __restrict__ inc, const int64_t* __restrict__ tid, const double* __restrict__ tph, int D, int T, int Mt,
    const double* __restrict__ weights /* [C+1] */, int C, const double* __restrict__ gain /* [D] or NULL */,
    double truth_threshold, float* __restrict__ out, int64_t* __restrict__ out_tid /* [D][Mt][T]: slot-major working copy */,
    double* __restrict__ out_tph /* [D][Mt][T] */,
    const double* __restrict__ truth_max /* [D][T]: light_truth_max_kernel, or NULL when Mt == 0 */,
    const int64_t* __restrict__ tid_sm /* RESPONSE: the input ids slot-major, [D][Mt][T] (read at the OUTPUT tick) */,
    const double* __restrict__ inc_f64 /* [D][T]: `inc` widened (light_widen_kernel), or NULL */) {
  __shared__ float s_x[JCHUNK];
  __shared__ double s_w[JCHUNK + LR_THREADS];
  // largest photon count (RESPONSE: magnitude) among the filled truth slots of input tick j, -1 when it has none: a pair
  // (i, j) whose weight times this stays below the threshold cannot pass the per-slot test (rounding is monotone), so its
  // walk over the <= Mt slots -- Mt global reads per pair -- is skipped as a whole
  __shared__ double s_tmax[JCHUNK];
  __shared__ int64_t s_sid[LR_THREADS / 64][64];       // a wave's copy of the truth slots of the input tick it is walking
  __shared__ double s_sph[LR_THREADS / 64][64];
  const int d = blockIdx.y;
  const int i0 = blockIdx.x * LR_THREADS;
  const int i = i0 + threadIdx.x;
  const bool live = i < T;
  const int i_last = min(i0 + LR_THREADS, T) - 1;          // last tick of this workgroup
  const int j_begin = max(i0 - C, 0);                       // first j any of its ticks can reach
  const float* x = inc + (int64_t)d * T;
  float acc = live ? out[(int64_t)d * T + i] : 0.f;
  const double g = RESPONSE ? gain[d] : 1.0;
  const int my_j0 = max(i - C, 0);
  // the output's truth rows live slot-major while the kernel works ([detector][slot][tick]: the wave's 64 ticks of one slot are
  // one 512-byte run; in the reference's [detector][tick][slot] every lane's row is 8 Mt bytes from its neighbour's and each
  // access of a term its own cache line -- the stage ran at the rate the L2 serves those lines)
  const int64_t orow = (int64_t)d * Mt * T + i;
  auto O = [&](int b) { return orow + (int64_t)b * T; };
  // RESPONSE: the reference's slot search compares ids of the INPUT row at the OUTPUT tick (light_sim.py:331-335), which
  // does not change while this thread works: with f = index of that row's first -1 (Mt if none) and distinct ids in front
  // of it, "first b with tid[dst+b] == tid[dst+a] or -1" is a itself for a < f and f otherwise -- no search.  Rows with a
  // repeated id keep the literal search.
  // Scintillation stage: the output row fills from the front (a term takes the first slot that holds its id or is empty).
  // The thread keeps the number of filled slots and a 256-bit signature of the ids stored: an id whose bit is clear is not
  // in the row, so it goes straight to slot `filled` (or nowhere when the row is full) without the search; only ids whose
  // bit is set search.  A row that arrives with a hole in front of a filled slot keeps the literal search.
  int filled = 0;
  bool prefix_ok = false;
  unsigned long long sig0 = 0, sig1 = 0, sig2 = 0, sig3 = 0;
  auto sig_bit = [](int64_t id, unsigned long long& word_sel) {
    unsigned long long h = (unsigned long long)id * 0x9E3779B97F4A7C15ull;
    word_sel = (h >> 62);
    return 1ull << ((h >> 56) & 63ull);
  };
  if (!RESPONSE && Mt > 0 && live) {
    prefix_ok = true;
    bool seen_empty = false;
    for (int b = 0; b < Mt; b++) {
      const int64_t id = out_tid[O(b)];
      if (id == -1) { seen_empty = true; continue; }
      if (seen_empty) { prefix_ok = false; break; }
      filled = b + 1;
      unsigned long long ws;
      const unsigned long long bit = sig_bit(id, ws);
      if (ws == 0) sig0 |= bit; else if (ws == 1) sig1 |= bit; else if (ws == 2) sig2 |= bit; else sig3 |= bit;
    }
  }
  int row_f = Mt;
  bool row_unique = false;
  if (RESPONSE && Mt > 0 && live) {
    for (int b = 0; b < Mt; b++)
      if (tid_sm[O(b)] == -1) { row_f = b; break; }
    row_unique = true;
    for (int b = 1; b < row_f && row_unique; b++) {
      const int64_t idb = tid_sm[O(b)];
      for (int e = 0; e < b; e++)
        if (tid_sm[O(e)] == idb) { row_unique = false; break; }
    }
  }

  for (int jc = j_begin; jc <= i_last; jc += JCHUNK) {
    const int nj = min(JCHUNK, i_last - jc + 1);
    // weights this chunk can meet: n = i - j in [i0 - (jc + nj - 1), i_last - jc]
    const int n_lo = max(i0 - (jc + nj - 1), 0), n_hi = min(i_last - jc, C);
    __syncthreads();
    for (int k = threadIdx.x; k < nj; k += LR_THREADS) {
      s_x[k] = x[jc + k];
      // (without truth slots the bound's place holds the samples as doubles: the unrolled blocks below read them from there)
      s_tmax[k] = Mt > 0 ? truth_max[(int64_t)d * T + jc + k] : (double)x[jc + k];
    }
    // (SiPM stage without truth slots: the weights staged as LIGHT_GAIN[row] * w, the product the reference forms first, :320)
    const bool premul = RESPONSE && Mt <= 0;
    for (int k = threadIdx.x; k <= n_hi - n_lo; k += LR_THREADS) s_w[k] = premul ? g * weights[n_lo + k] : weights[n_lo + k];
    __syncthreads();
    // One j loop for the whole wave (a thread's own range [i - C, i] is a predicate): the input tick's sample and bound are
    // broadcast LDS reads, a zero sample or a tick without truth skips the wave as a whole, and when some lane's product can
    // pass the threshold the wave stages the tick's <= 64 truth slots in LDS with one coalesced read -- the walk over them
    // (the bulk of the truth leg: one dependent L2 round trip per slot and pair before) then runs on LDS broadcasts.
    // The chunk is walked 64 input ticks at a time: lane l fetches sample and bound of tick jb + l once, the loop over the 64 ticks
    // takes them from there with v_readlane (three LDS reads per term were what the plain convolution ran at: the LDS, shared by
    // the CU's four SIMDs, was its limit), and which ticks are zero / without truth is two ballots.
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int jb = jc; jb <= jc + nj - 1; jb += 64) {
      const int nb64 = min(64, jc + nj - jb);
      const float x_l = lane < nb64 ? s_x[jb - jc + lane] : 0.f;
      const double t_l = (Mt > 0 && lane < nb64) ? s_tmax[jb - jc + lane] : -1.0;
      const unsigned long long m_nz = RESPONSE ? ~0ull : __ballot(x_l != 0.f);
      const unsigned long long m_truth = __ballot(t_l >= 0.0);
      unsigned long long todo = (nb64 == 64 ? ~0ull : ((1ull << nb64) - 1ull)) & m_nz;
      // a block inside every lane's own range [i - C, i] (most blocks) and without truth: the bare convolution, no predicate
      const int w_i0 = i0 + (wv << 6);
      const bool inside = w_i0 + 63 < T && jb >= max(w_i0 + 63 - C, 0) && jb + nb64 - 1 <= w_i0;
      if (inside && !m_truth) {
        const double* wl = s_w + (i - jb) - n_lo;
        if (todo == ~0ull) {
          // all 64 ticks of the block (the SiPM stage, and lit stretches of the scintillation stage): lane numbers and weight
          // offsets are constants -- no bit scan, no address arithmetic per term; what is left is the sample, the weight and the
          // three dependent operations of the f4 sum (5-6 instructions per term instead of ~12)
          if (Mt <= 0) {
            // the sample arrives as a double -- no v_cvt_f64_f32 per term (the conversions are the slow instructions of the term: with
            // it gone the SiPM stage of a 2x2 batch runs 15.6 -> 11.8 ms).  SiPM stage (every block is of this kind): lane t holds
            // tick jb + t's sample widened once, two v_readlane per term; from LDS by a broadcast read instead the stage is bound by
            // the LDS (13.5 ms: two 8-byte reads per term and lane).  Scintillation stage (few such blocks): the broadcast read,
            // 1.82 against 1.91 ms.
            const double* xd = s_tmax + (jb - jc);
            const double xd_l = (double)x_l;
            if (inc_f64) {
              // (the widened samples from memory at a wave-uniform address: scalar loads, eight doubles per request straight into
              // scalar registers -- no vector instruction and no LDS read for the sample)
              const double* xg = inc_f64 + (int64_t)d * T + jb;
#pragma unroll
              for (int t8 = 0; t8 < 64; t8 += 8) {
                double w8[8], x8[8];
#pragma unroll
                for (int u = 0; u < 8; u++) w8[u] = wl[-(t8 + u)];
#pragma unroll
                for (int u = 0; u < 8; u++) x8[u] = xg[t8 + u];
#pragma unroll
                for (int u = 0; u < 8; u++) acc = (float)((double)acc + w8[u] * x8[u]);
              }
              continue;
            }
#pragma unroll
            for (int t8 = 0; t8 < 64; t8 += 8) {
              double w8[8], x8[8];
#pragma unroll
              for (int u = 0; u < 8; u++) w8[u] = wl[-(t8 + u)];
#pragma unroll
              for (int u = 0; u < 8; u++) x8[u] = RESPONSE ? wave_lane_f64(xd_l, t8 + u) : xd[t8 + u];
#pragma unroll
              for (int u = 0; u < 8; u++) acc = (float)((double)acc + w8[u] * x8[u]);
            }
            continue;
          }
#pragma unroll
          for (int t8 = 0; t8 < 64; t8 += 8) {
            double w8[8];
#pragma unroll
            for (int u = 0; u < 8; u++) w8[u] = wl[-(t8 + u)];
#pragma unroll
            for (int u = 0; u < 8; u++) {
              const float xv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x_l), t8 + u));
              if (RESPONSE && !premul) acc = (float)((double)acc + g * w8[u] * (double)xv);
              else acc = (float)((double)acc + w8[u] * (double)xv);
            }
          }
          continue;
        }
        for (; todo; todo &= todo - 1) {
          const int t = __ffsll((long long)todo) - 1;
          const float xv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x_l), t));
          const double w = wl[-t];
          if (RESPONSE && !premul) acc = (float)((double)acc + g * w * (double)xv);
          else acc = (float)((double)acc + w * (double)xv);
        }
        continue;
      }
    for (; todo; todo &= todo - 1) {
      const int t = __ffsll((long long)todo) - 1;
      const int j = jb + t;
      const float xv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x_l), t));
      const bool mine = live && j >= my_j0 && j <= i;
      const double w = mine ? s_w[(i - j) - n_lo] : 0.0;
      if (mine) {
        if (RESPONSE && !premul) acc = (float)((double)acc + g * w * (double)xv);          // :320  LIGHT_GAIN[idet] * tick_weight * x
        else acc = (float)((double)acc + w * (double)xv);                       // :169 (and :320 with the gain already in w)
      }
      if (!((m_truth >> t) & 1ull)) continue;
      const double bound = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(t_l), t), __builtin_amdgcn_readlane(__double2loint(t_l), t));
      const bool walk = mine && !(RESPONSE ? (fabs(w) * bound < truth_threshold) : (w >= 0.0 && w * bound < truth_threshold));
      if (!__ballot(walk)) continue;
      const int64_t src = ((int64_t)d * T + j) * Mt;
      const bool staged = Mt <= 64;
      // the slots some lane's product can pass on: lane a holds slot a while it stages it, so with the largest weight among the
      // walking lanes this is one ballot -- the walk visits those slots only, in ascending order, instead of all filled ones
      unsigned long long cand = ~0ull;
      if (staged) {
        __builtin_amdgcn_wave_barrier();
        int64_t my_id = -1;
        double my_ph = 0.0;
        if (lane < Mt) {
          my_id = tid[src + lane];
          my_ph = tph[src + lane];
          s_sid[wv][lane] = my_id;
          s_sph[wv][lane] = my_ph;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const unsigned long long empty = __ballot(lane < Mt && my_id == -1);
        const int nfill = empty ? __ffsll((long long)empty) - 1 : Mt;          // (the walk stops at the first empty slot)
        const double wmax = wave_max_f64(walk ? fabs(w) : 0.0);
        if (RESPONSE) cand = __ballot(lane < nfill && !(wmax * fabs(my_ph) < truth_threshold));
        else if (__ballot(walk && w < 0.0)) cand = nfill >= 64 ? ~0ull : ((1ull << nfill) - 1ull);
        else cand = __ballot(lane < nfill && !(my_ph >= 0.0 ? wmax * my_ph < truth_threshold : truth_threshold > 0.0));
      }
      if (walk) {
        for (int a0 = 0; a0 < Mt; a0++) {
          int a = a0;
          if (staged) {
            if (!cand) break;
            a = __ffsll((long long)cand) - 1;
            cand &= cand - 1;
          }
          const int64_t ida_src = staged ? s_sid[wv][a] : tid[src + a];
          if (ida_src == -1) break;
          const double ph = staged ? s_sph[wv][a] : tph[src + a];
          if (RESPONSE ? (fabs(w * ph) < truth_threshold) : (w * ph < truth_threshold)) continue;
          if (RESPONSE && row_unique) {
            const int b = a < row_f ? a : row_f;
            if (b < Mt) {
              out_tid[O(b)] = tid_sm[O(a)];
              out_tph[O(b)] += w * ph;
            }
            continue;
          }
          if (!RESPONSE && prefix_ok) {
            const int64_t id = ida_src;
            unsigned long long ws;
            const unsigned long long bit = sig_bit(id, ws);
            const unsigned long long word = ws == 0 ? sig0 : (ws == 1 ? sig1 : (ws == 2 ? sig2 : sig3));
            if (id != -1 && !(word & bit)) {             // certainly not stored yet
              if (filled < Mt) {
                out_tid[O(filled)] = id;
                out_tph[O(filled)] += w * ph;
                filled++;
                if (ws == 0) sig0 |= bit; else if (ws == 1) sig1 |= bit; else if (ws == 2) sig2 |= bit; else sig3 |= bit;
              }
              continue;
            }
            for (int b = 0; b < Mt; b++) {              // maybe stored: the literal search (an insert extends the prefix)
              const int64_t cur = out_tid[O(b)];
              if (cur == id || cur == -1) {
                out_tid[O(b)] = id;
                out_tph[O(b)] += w * ph;
                if (cur == -1 && id != -1) {
                  filled = b + 1;
                  if (ws == 0) sig0 |= bit; else if (ws == 1) sig1 |= bit; else if (ws == 2) sig2 |= bit; else sig3 |= bit;
                }
                break;
              }
            }
            continue;
          }
          for (int b = 0; b < Mt; b++) {
            if (RESPONSE) {
              // :331-335 literally: the slot test reads the INPUT ids at [idet, itick], not the output's
              const int64_t idb = tid_sm[O(b)], ida = tid_sm[O(a)];
              if (idb == ida || idb == -1) {
                out_tid[O(b)] = ida;
                out_tph[O(b)] += w * ph;
                break;
              }
            } else {
              const int64_t id = ida_src;
              if (out_tid[O(b)] == id || out_tid[O(b)] == -1) {           // :180-183
                out_tid[O(b)] = id;
                out_tph[O(b)] += w * ph;
                break;
              }
            }
          }
        }
      }
    }
    }
  }
  if (live) out[(int64_t)d * T + i] = acc;
}

// truth_max[d][j] = max over the filled slots a of tph[d][j][a] (RESPONSE: |tph|), -1 when slot 0 is empty
__global__ void light_truth_max_kernel(const int64_t* __restrict__ tid, const double* __restrict__ tph, int64_t n, int Mt,
                                       int response, double* __restrict__ truth_max) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  double m = -1.0;
  bool any = false;
  for (int a = 0; a < Mt; a++) {
    if (tid[e * Mt + a] == -1) break;
    const double ph = response ? fabs(tph[e * Mt + a]) : tph[e * Mt + a];
    m = any ? fmax(m, ph) : ph;
    any = true;
  }
  // (scintillation mode compares signed products: a tick whose largest entry is negative still has filled slots; 0 keeps
  // it in play -- the per-slot test decides)
  truth_max[e] = any ? (response ? m : fmax(m, 0.0)) : -1.0;
}

// ---- the truth slots on their own: a wave's 64 output rows in LDS (round 4) ----------------------------------------------------
// The 2x2 batch of tools/light_wvfm_profile.py accepts ~10^9 (output tick, input tick, slot) terms per stage; in light_conv_kernel
// every one is a read-modify-write of the output slot through L2 (and a search of the output row through L2 in the scintillation
// stage), and a 64-tick block of input ticks that holds any truth takes the predicated path for the plain sum as well.  Here the
// plain sum is left to light_conv_kernel (called without truth: its unrolled blocks) and one wave owns the truth rows of 64
// consecutive output ticks of a detector, in LDS as [slot][tick] (stride LT_S words: lane = tick reads and adds without a bank
// conflict whatever the slot, and the transposed load / store of the reference's [tick][slot] rows is conflict-free too):
//   * the input ticks are screened 64 at a time -- lane l holds tick jb + l: its bound (light_truth_max_kernel) times the largest
//     weight any of the wave's ticks can meet it with (light_env_kernel) against the threshold, one ballot per block;
//   * a tick that passes is walked like the reference walks it (ascending slots, light_sim.py:171-184 / :323-337), the accepted
//     term added with ds_add_f64 -- the adds of one lane to one address retire in program order, the sums are the reference's;
//   * SiPM stage: the reference searches the INPUT row at the OUTPUT tick (:331-335), fixed while the kernel runs: slot min(a, f)
//     for a row of distinct ids with its first -1 at f (other rows: the literal search);
//   * scintillation stage: the output row fills from the front; per lane a 128-entry table (id hash -> slot, one byte) finds an
//     id or proves it new in one or two LDS reads -- the id is the same for the whole wave, so are the hash and the probe sequence.
// Rows are loaded at the wave's first accepted tick and only touched entries are written back.
#define LT_S 65
#define LT_TAB 128
#define LT_NONE 0xFFu

// env[m + 63], m = (first output tick of a wave) - (input tick) in [-63, C]: the largest weight (SiPM stage: magnitude) any of the
// wave's 64 ticks meets that input tick with -- n = i - j in [m, m + 63], inside [0, C]; +inf where the per-term test cannot be
// bounded (a negative weight in the scintillation stage's signed test, a weight that is not finite)
__global__ void light_env_kernel(const double* __restrict__ weights, int C, int response, double* __restrict__ env) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > C + 63) return;
  const int m = k - 63;
  double e = 0.0;
  bool open = false;
  for (int n = max(m, 0); n <= min(m + 63, C); n++) {
    const double w = weights[n];
    if (!(fabs(w) <= 1.7e308) || (!response && !(w >= 0.0))) open = true;
    e = fmax(e, response ? fabs(w) : w);
  }
  env[k] = open ? __builtin_inf() : e;
}

// env2[q] = max of env over the 64 offsets a block of 64 input ticks, q blocks in front of the wave's own, is met with
__global__ void light_env2_kernel(const double* __restrict__ env, int C, double* __restrict__ env2) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (64 * q > C + 63) return;
  double e = 0.0;
  for (int k = 64 * q; k <= min(64 * q + 63, C + 63); k++) e = fmax(e, env[k]);
  env2[q] = e;
}

// bound[d][j]: light_truth_max_kernel's bound of input tick j, -1 also where the scintillation stage skips the tick (a zero sample,
// light_sim.py:166); block_bound[d][j / 64]: the largest of a block of 64 ticks (-1: none of them takes part) -- one wave per block
__global__ void __launch_bounds__(64) light_truth_bound_kernel(const float* __restrict__ inc, const int64_t* __restrict__ tid,
                                                               const double* __restrict__ tph, int T, int Mt, int response,
                                                               double* __restrict__ bound, double* __restrict__ block_bound) {
  const int d = blockIdx.y, j = blockIdx.x * 64 + threadIdx.x;
  double m = -1.0;
  if (j < T && (response || inc[(int64_t)d * T + j] != 0.f)) {
    const int64_t e = (int64_t)d * T + j;
    bool any = false;
    for (int a = 0; a < Mt; a++) {
      if (tid[e * Mt + a] == -1) break;
      const double ph = response ? fabs(tph[e * Mt + a]) : tph[e * Mt + a];
      m = any ? fmax(m, ph) : ph;
      any = true;
    }
    m = any ? (response ? m : fmax(m, 0.0)) : -1.0;
  }
  if (j < T) bound[(int64_t)d * T + j] = m;
  const double bm = wave_max_f64(m);
  if (threadIdx.x == 0) block_bound[(int64_t)d * gridDim.x + blockIdx.x] = bm;
}

__device__ __forceinline__ int64_t wave_lane_i64(int64_t v, int l) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(unsigned long long)v, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v >> 32), l);
  return (int64_t)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

template <bool RESPONSE>
__global__ void __launch_bounds__(64) light_truth_lds_kernel(
    const int64_t* __restrict__ tid, const double* __restrict__ tph, int T, int Mt, const double* __restrict__ weights, int C,
    const double* __restrict__ env /* [C + 64] */, double truth_threshold, int64_t* __restrict__ out_tid,
    double* __restrict__ out_tph /* [D][T][Mt], the reference's layout */, const double* __restrict__ truth_max /* [D][T] */,
    const double* __restrict__ block_max /* [D][ceil(T / 64)] */, const double* __restrict__ env2) {
  extern __shared__ double s_lt[];
  double* s_acc = s_lt;                                        // [Mt][LT_S]: the output photons
  // scintillation: [Mt][LT_S] output ids, 32 bits each (ids 0 .. 2^32 - 2; all ones = -1 = empty) -- 48 instead of 61 KB per wave at 50
  // slots, three waves per CU instead of two.  A wave that meets an id outside that range (none in the reference's use: segment / track
  // indices) writes its rows back and goes on in memory, literally (`in_memory` below).
  unsigned* s_id = (unsigned*)(s_acc + Mt * LT_S);
  double* s_w = RESPONSE ? s_acc + Mt * LT_S : (double*)(s_id + ((Mt * LT_S + 1) & ~1));        // [128]: the weights the wave's ticks meet a block of input ticks with
  unsigned char* s_byte = (unsigned char*)(s_w + 128);         // SiPM: [64 ticks][64] last input slot added to an output slot; scintillation: [LT_TAB][64 ticks] id hash -> slot
  const int lane = threadIdx.x;
  const int d = blockIdx.y, i0 = blockIdx.x * 64, i = i0 + lane;
  const bool live = i < T;
  const int nrow = min(64, T - i0) * Mt;
  const int64_t rbase = ((int64_t)d * T + i0) * Mt;           // the wave's rows: nrow contiguous words
  const double* tmax = truth_max + (int64_t)d * T;
  bool loaded = false, changed = false;
  int filled = 0;                  // scintillation: slots in front of the row's first -1
  bool literal = false;            // scintillation, wave-uniform: some row arrived with a hole in front of a filled slot
  bool in_memory = false;          // scintillation, wave-uniform: an id that does not fit 32 bits was met -- the rows are walked in memory
  auto narrow = [](int64_t id) { return (unsigned long long)(id + 1) <= 0xFFFFFFFFull; };          // -1 .. 2^32 - 2
  auto wide_id = [](unsigned u) { return u == 0xFFFFFFFFu ? (int64_t)-1 : (int64_t)u; };
  int row_f = Mt;                  // SiPM: first -1 of the input row at the output tick
  bool regular = true;             // SiPM: distinct ids in front of it, -1 behind
  unsigned long long m_reg = 0;    // SiPM, wave-uniform: the ticks with such a row

  auto id_hash = [](int64_t id) { return (int)(((unsigned long long)id * 0x9E3779B97F4A7C15ull) >> 57); };
  auto load_rows = [&]() {
    // ids first (SiPM: into the accumulators' place), the per-row state from them, then the photons
    int64_t* ids = (int64_t*)s_acc;
    const int64_t* gid = RESPONSE ? tid : out_tid;
    bool all_narrow = true;
    for (int e0 = lane; e0 < 64 * Mt; e0 += 512) {          // (eight words per lane in flight)
      int64_t v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = e0 + 64 * u < nrow ? gid[rbase + e0 + 64 * u] : (int64_t)-1;
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int e = e0 + 64 * u, t = e / Mt, b = e - t * Mt;
        if (e < 64 * Mt) {
          if (RESPONSE) ids[b * LT_S + t] = v[u];
          else { s_id[b * LT_S + t] = (unsigned)v[u]; all_narrow = all_narrow && narrow(v[u]); }
        }
      }
    }
    if (!RESPONSE && __ballot(!all_narrow)) {          // (nothing is in LDS that memory does not hold)
      in_memory = true;
      return;
    }
    const int nbyte = (RESPONSE ? 64 : LT_TAB) * 64;
    for (int e = lane * 4; e < nbyte; e += 256) *(unsigned*)(s_byte + e) = 0xFFFFFFFFu;
    wave_lds_sync();
    if (RESPONSE) {
      row_f = Mt;
      for (int b = 0; b < Mt; b++)
        if (ids[b * LT_S + lane] == -1) { row_f = b; break; }
      regular = true;
      for (int b = row_f + 1; b < Mt; b++)
        if (ids[b * LT_S + lane] != -1) { regular = false; break; }
      for (int b = 1; b < row_f && regular; b++) {
        const int64_t idb = ids[b * LT_S + lane];
        for (int e = 0; e < b; e++)
          if (ids[e * LT_S + lane] == idb) { regular = false; break; }
      }
      m_reg = __ballot(regular);
      wave_lds_sync();
    } else {
      bool seen_empty = false, clean = true;
      filled = 0;
      for (int b = 0; b < Mt; b++) {
        if (s_id[b * LT_S + lane] == 0xFFFFFFFFu) seen_empty = true;
        else if (seen_empty) clean = false;
        else filled = b + 1;
      }
      literal = __ballot(live && !clean) != 0ull;
      if (!literal) {
        for (int b = 0; b < filled; b++) {              // the ids the row arrives with (a repeated one keeps its first slot)
          const unsigned id32 = s_id[b * LT_S + lane];
          const int h0 = id_hash((int64_t)id32);
          for (int k = 0; k < LT_TAB; k++) {
            const int idx = (h0 + k) & (LT_TAB - 1);
            const unsigned p = s_byte[idx * 64 + lane];
            if (p == LT_NONE) { s_byte[idx * 64 + lane] = (unsigned char)b; break; }
            if (s_id[(int)p * LT_S + lane] == id32) break;
          }
        }
      }
    }
    for (int e0 = lane; e0 < 64 * Mt; e0 += 512) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = e0 + 64 * u < nrow ? out_tph[rbase + e0 + 64 * u] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int e = e0 + 64 * u, t = e / Mt, b = e - t * Mt;
        if (e < 64 * Mt) s_acc[b * LT_S + t] = v[u];
      }
    }
    wave_lds_sync();
  };

  // write back what was touched, in the reference's layout (neighbouring lanes: neighbouring words)
  auto store_rows = [&]() {
    if (!loaded || in_memory || !__ballot(changed)) return;
    unsigned char* s_fill = s_byte;          // scintillation: slots to write per tick (in the id table's place: it is not read again; the weights' are)
    wave_lds_sync();
    if (!RESPONSE) s_fill[lane] = (unsigned char)(changed ? (literal ? Mt : filled) : 0);
    wave_lds_sync();
    for (int e = lane; e < nrow; e += 64) {
      const int t = e / Mt, b = e - t * Mt;
      if (RESPONSE) {
        const unsigned la = s_byte[t * 64 + b];
        if (la != LT_NONE) {
          out_tid[rbase + e] = tid[rbase + (int64_t)t * Mt + (int)la];
          out_tph[rbase + e] = s_acc[b * LT_S + t];
        }
      } else if (b < (int)s_fill[t]) {
        out_tid[rbase + e] = wide_id(s_id[b * LT_S + t]);
        out_tph[rbase + e] = s_acc[b * LT_S + t];
      }
    }
  };
  // an accepted term of the scintillation stage on the rows in memory, literally (:180-183) -- after an id that does not fit the rows
  // in LDS; loads past the L1 (the wave's own stores, of other lanes, may sit behind a stale line)
  auto term_in_memory = [&](int64_t id, double v) {
    unsigned long long* orow = (unsigned long long*)out_tid + rbase + (int64_t)lane * Mt;
    unsigned long long* prow = (unsigned long long*)out_tph + rbase + (int64_t)lane * Mt;
    for (int b = 0; b < Mt; b++) {
      const int64_t cur = (int64_t)__hip_atomic_load(orow + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (cur == id || cur == -1) {
        const unsigned long long pb = __hip_atomic_load(prow + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double sum = __longlong_as_double((long long)pb) + v;
        __hip_atomic_store(orow + b, (unsigned long long)id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(prow + b, (unsigned long long)__double_as_longlong(sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  };
  // Screening in two levels: 64 blocks of 64 input ticks per trip to memory (lane = block: its largest bound times the largest
  // weight the wave meets it with), then the ticks of a block that passed (lane = tick) -- fetched, with the weights the wave meets
  // the block with, one passing block ahead of the one being walked; the truth slots of a tick that passed one tick ahead.
  const int j_lo = max(i0 - C, 0), j_hi = min(i0 + 63, T - 1);
  const int B_lo = j_lo >> 6, B_hi = j_hi >> 6;
  const double* bmax = block_max + (int64_t)d * ((T + 63) >> 6);
  for (int Bg = B_lo; Bg <= B_hi; Bg += 64) {
    unsigned long long m1;
    {
      const int B_l = Bg + lane;
      const bool bv = B_l <= B_hi;
      const double b_l = bv ? bmax[B_l] : -1.0;
      const double e2 = bv ? env2[(i0 >> 6) - B_l] : 0.0;
      m1 = __ballot(b_l >= 0.0 && !(e2 * b_l < truth_threshold));
    }
    if (!m1) continue;
    double tn, en, wan, wbn;
    auto fetch_block = [&](int B) {
      const int j_l = 64 * B + lane;
      const bool jv = j_l >= j_lo && j_l <= j_hi;
      tn = jv ? tmax[j_l] : -1.0;
      en = jv ? env[i0 - j_l + 63] : 0.0;
      const int n1 = i0 - 64 * B - 63 + lane, n2 = n1 + 64;
      wan = (n1 >= 0 && n1 <= C) ? weights[n1] : 0.0;
      wbn = (n2 >= 0 && n2 <= C) ? weights[n2] : 0.0;
    };
    int Bn = Bg + __ffsll((long long)m1) - 1;
    m1 &= m1 - 1;
    fetch_block(Bn);
    for (bool more_b = true; more_b;) {
      const int jb = 64 * Bn;
      const double t_l = tn, e_l = en, wa = wan, wb = wbn;
      more_b = m1 != 0ull;
      if (more_b) {
        Bn = Bg + __ffsll((long long)m1) - 1;
        m1 &= m1 - 1;
        fetch_block(Bn);
      }
      unsigned long long mc = __ballot(t_l >= 0.0 && !(e_l * t_l < truth_threshold));
      if (!mc) continue;
      wave_lds_sync();
      s_w[lane] = wa;
      s_w[64 + lane] = wb;
      wave_lds_sync();
      int64_t id_n = -1;
      double ph_n = 0.0;
      auto fetch_row = [&](int jj) {
        const int64_t src = ((int64_t)d * T + jj) * Mt;
        id_n = -1;
        ph_n = 0.0;
        if (lane < Mt) { id_n = tid[src + lane]; ph_n = tph[src + lane]; }
      };
      int tnx = __ffsll((long long)mc) - 1;
      mc &= mc - 1;
      fetch_row(jb + tnx);
      for (bool more_t = true; more_t;) {
        const int t = tnx;
        const int64_t my_id = id_n;
        const double my_ph = ph_n;
        more_t = mc != 0ull;
        if (more_t) {
          tnx = __ffsll((long long)mc) - 1;
          mc &= mc - 1;
          fetch_row(jb + tnx);
        }
        const int j = jb + t;
        const int n = i - j;
        const bool mine = live && n >= 0 && n <= C;
        const double w = mine ? s_w[lane + 63 - t] : 0.0;
        const double bound = wave_lane_f64(t_l, t);
        const bool walk = mine && !(RESPONSE ? (fabs(w) * bound < truth_threshold) : (w >= 0.0 && w * bound < truth_threshold));
        if (!__ballot(walk)) continue;
        if (!loaded) { load_rows(); loaded = true; }
        // the input tick's slots: lane a holds slot a
        const unsigned long long empty = __ballot(lane < Mt && my_id == -1);
        const int nfill = empty ? __ffsll((long long)empty) - 1 : Mt;          // (the walk stops at the first empty slot)
        if (!RESPONSE && !in_memory && __ballot(lane < nfill && !narrow(my_id))) {
          store_rows();
          __threadfence();
          in_memory = true;
        }
        if (RESPONSE) {
          // SiPM stage, the walk turned round: lane a holds slot a of the input tick and the loop runs over the wave's output ticks
          // that can pass -- a term reaches ~13 of the wave's 64 ticks but most of the tick's slots, and the target slot needs no
          // search: min(a, f) for a row of distinct ids with its first -1 at f.  Terms of different slots below f meet different
          // words; those at or above f all meet slot f and are added one by one, in ascending a like the reference's loop.
          changed = true;            // (what was touched is in s_byte)
          for (unsigned long long mw = __ballot(walk); mw; mw &= mw - 1) {
            const int ti = __ffsll((long long)mw) - 1;
            const double w_i = wave_lane_f64(w, ti);
            const int rf = __builtin_amdgcn_readlane(row_f, ti);
            const double v = w_i * my_ph;
            const bool pass = lane < nfill && !(fabs(v) < truth_threshold);
            unsigned long long serial;
            if ((m_reg >> ti) & 1ull) {
              if (pass && lane < rf) {
                atomicAdd(&s_acc[lane * LT_S + ti], v);
                s_byte[ti * 64 + lane] = (unsigned char)lane;
              }
              serial = rf < Mt ? __ballot(pass && lane >= rf) : 0ull;
              for (; serial; serial &= serial - 1) {
                const int a = __ffsll((long long)serial) - 1;
                if (lane == a) {
                  atomicAdd(&s_acc[rf * LT_S + ti], v);
                  s_byte[ti * 64 + rf] = (unsigned char)a;
                }
              }
            } else {
              // :331-335 literally: the slot test reads the INPUT ids at [idet, itick] -- lane e holds that row's slot e
              const int64_t idb = lane < Mt ? tid[rbase + (int64_t)ti * Mt + lane] : (int64_t)0;
              for (serial = __ballot(pass); serial; serial &= serial - 1) {
                const int a = __ffsll((long long)serial) - 1;
                const int64_t ida = wave_lane_i64(idb, a);
                const unsigned long long hit = __ballot(lane < Mt && (idb == ida || idb == -1));
                if (hit && lane == a) {
                  const int b = __ffsll((long long)hit) - 1;
                  atomicAdd(&s_acc[b * LT_S + ti], v);
                  s_byte[ti * 64 + b] = (unsigned char)a;
                }
              }
            }
          }
          continue;
        }
        const double wmax = wave_max_f64(walk ? fabs(w) : 0.0);
        changed = changed || walk;          // (a superset of the rows that change: what is written back for the others is what they hold)
        // the slots some lane's product can pass on, in ascending order
        unsigned long long cand;
        if (__ballot(walk && w < 0.0)) cand = nfill >= 64 ? ~0ull : ((1ull << nfill) - 1ull);
        else cand = __ballot(lane < nfill && !(my_ph >= 0.0 ? wmax * my_ph < truth_threshold : truth_threshold > 0.0));
        for (; cand; cand &= cand - 1) {
          const int a = __ffsll((long long)cand) - 1;
          const double ph = wave_lane_f64(my_ph, a);
          const double v = w * ph;
          const bool pass = walk && !(RESPONSE ? (fabs(v) < truth_threshold) : (v < truth_threshold));
          if (!__ballot(pass)) continue;
          if (!RESPONSE) {
            const int64_t id = wave_lane_i64(my_id, a);
            if (in_memory) {
              if (pass) term_in_memory(id, v);
            } else if (literal) {
              if (pass) {
                const unsigned id32 = (unsigned)id;
                for (int b = 0; b < Mt; b++) {                                  // :180-183
                  const unsigned cur = s_id[b * LT_S + lane];
                  if (cur == id32 || cur == 0xFFFFFFFFu) {
                    s_id[b * LT_S + lane] = id32;
                    atomicAdd(&s_acc[b * LT_S + lane], v);
                    changed = true;
                    break;
                  }
                }
              }
            } else {
              // the first entry of the probe sequence without a branch (both reads unconditional, from valid addresses): it holds
              // the id (two terms of three) or is empty (the rest); an entry with another id goes on along the sequence below
              const int h0 = id_hash(id);
              const unsigned p0 = s_byte[h0 * 64 + lane];
              const bool has = p0 != LT_NONE;
              const unsigned id32 = (unsigned)id;
              const unsigned cur = s_id[(has ? (int)p0 : 0) * LT_S + lane];
              const bool fresh = pass && !has && filled < Mt;
              bool need = pass && has && cur != id32;
              const int pos = has ? (int)p0 : filled;
              if (fresh) {                               // not in the row: its first empty slot, if it has one
                s_id[pos * LT_S + lane] = id32;
                s_byte[h0 * 64 + lane] = (unsigned char)pos;
                filled++;
              }
              if (fresh || (pass && has && cur == id32)) atomicAdd(&s_acc[pos * LT_S + lane], v);
              for (int k = 1; k < LT_TAB && __ballot(need); k++) {
                const int idx = (h0 + k) & (LT_TAB - 1);
                if (need) {
                  const unsigned p = s_byte[idx * 64 + lane];
                  if (p == LT_NONE) {
                    if (filled < Mt) {
                      s_id[filled * LT_S + lane] = id32;
                      s_byte[idx * 64 + lane] = (unsigned char)filled;
                      atomicAdd(&s_acc[filled * LT_S + lane], v);
                      filled++;
                      changed = true;
                    }
                    need = false;
                  } else if (s_id[(int)p * LT_S + lane] == id32) {
                    atomicAdd(&s_acc[(int)p * LT_S + lane], v);
                    changed = true;
                    need = false;
                  }
                }
              }
            }
          }
        }
      }
    }
  }
  store_rows();
}

// [detector][tick][slot] <-> [detector][slot][tick] of an 8-byte array, TR_TICKS ticks of one detector per workgroup through LDS
// (both sides in runs of at least 512 bytes)
#define TR_TICKS 64
template <bool TO_SLOT_MAJOR>
__global__ void __launch_bounds__(256) light_truth_transpose_kernel(const unsigned long long* __restrict__ src,
                                                                    unsigned long long* __restrict__ dst, int T, int Mt) {
  extern __shared__ unsigned long long s_t[];        // [TR_TICKS][Mt + 1]
  const int d = blockIdx.y, t0 = blockIdx.x * TR_TICKS, nt = min(TR_TICKS, T - t0);
  const int64_t row_major = ((int64_t)d * T + t0) * Mt;          // nt * Mt contiguous words
  const int64_t slot_major = (int64_t)d * Mt * T + t0;           // slot b: nt words at + b * T
  const int n = nt * Mt;
  if (TO_SLOT_MAJOR) {
    for (int e = threadIdx.x; e < n; e += 256) {
      const int t = e / Mt, b = e - t * Mt;
      s_t[t * (Mt + 1) + b] = src[row_major + e];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < Mt * TR_TICKS; e += 256) {
      const int b = e / TR_TICKS, t = e % TR_TICKS;
      if (t < nt) dst[slot_major + (int64_t)b * T + t] = s_t[t * (Mt + 1) + b];
    }
  } else {
    for (int e = threadIdx.x; e < Mt * TR_TICKS; e += 256) {
      const int b = e / TR_TICKS, t = e % TR_TICKS;
      if (t < nt) s_t[t * (Mt + 1) + b] = src[slot_major + (int64_t)b * T + t];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < n; e += 256) {
      const int t = e / Mt, b = e - t * Mt;
      dst[row_major + e] = s_t[t * (Mt + 1) + b];
    }
  }
}

__global__ void light_widen_kernel(const float* __restrict__ x, int64_t n, double* __restrict__ xd) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) xd[e] = (double)x[e];
}

// the plain sum of a stage (no truth slots) on `st`
extern "C++" int light_response_plain(ldsim_ctx* ctx, hipStream_t st, bool response, const float* inc, int D, int T, const double* weights,
                                      int C, const double* gain, float* out) {
  if (D <= 0 || T <= 0) return 0;
  dim3 grid((unsigned)((T + LR_THREADS - 1) / LR_THREADS), (unsigned)D), block(LR_THREADS);
  const double thr = ctx->h_consts.mc_truth_threshold;
  const double* xd = nullptr;
  if (response) {          // (the SiPM stage's blocks of 64 input ticks all take part: the samples once more as doubles)
    const int64_t n = (int64_t)D * T;
    int rc = ldsim_ensure_buf(ctx, &ctx->light_xd, (size_t)n * 8 + 64);
    if (rc) return rc;
    hipLaunchKernelGGL(light_widen_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, inc, n, (double*)ctx->light_xd.p);
    xd = (const double*)ctx->light_xd.p;
  }
  if (response)
    hipLaunchKernelGGL(light_conv_kernel<true>, grid, block, 0, st, inc, (const int64_t*)nullptr, (const double*)nullptr, D, T, 0, weights,
                       C, gain, thr, out, (int64_t*)nullptr, (double*)nullptr, (const double*)nullptr, (const int64_t*)nullptr, xd);
  else
    hipLaunchKernelGGL(light_conv_kernel<false>, grid, block, 0, st, inc, (const int64_t*)nullptr, (const double*)nullptr, D, T, 0, weights,
                       C, gain, thr, out, (int64_t*)nullptr, (double*)nullptr, (const double*)nullptr, (const int64_t*)nullptr, xd);
  HIPCHK(hipGetLastError());
  return 0;
}

// can the truth slots of a stage go by light_truth_lds_kernel?
extern "C++" bool light_truth_in_lds(const ldsim_ctx* ctx, int Mt) { return Mt > 0 && Mt <= 64 && ctx->light_truth_lds; }

// the truth slots of a stage (light_truth_in_lds) on `st`: bounds, weight envelopes, light_truth_lds_kernel.  The scintillation and the
// SiPM stage of one call share the bound and envelope buffers: both on the same stream.
extern "C++" int light_response_truth(ldsim_ctx* ctx, hipStream_t st, bool response, const float* inc, const int64_t* tid, const double* tph,
                                      int D, int T, int Mt, const double* weights, int C, int64_t* out_tid, double* out_tph) {
  if (D <= 0 || T <= 0) return 0;
  const double thr = ctx->h_consts.mc_truth_threshold;
  const int nblk = (T + 63) / 64, nq = (C + 63) / 64 + 1;
  const int64_t n = (int64_t)D * T;
  int rc = ldsim_ensure_buf(ctx, &ctx->light_tmax, (size_t)(n + (int64_t)D * nblk) * 8);
  if (rc) return rc;
  if ((rc = ldsim_ensure_buf(ctx, &ctx->light_env, (size_t)(C + 64 + nq) * 8))) return rc;
  double* tmax = (double*)ctx->light_tmax.p;
  double* env = (double*)ctx->light_env.p;
  double* env2 = env + C + 64;
  hipLaunchKernelGGL(light_truth_bound_kernel, dim3((unsigned)nblk, (unsigned)D), dim3(64), 0, st, inc, tid, tph, T, Mt, response ? 1 : 0,
                     tmax, tmax + n);
  hipLaunchKernelGGL(light_env_kernel, dim3((unsigned)((C + 64 + 255) / 256)), dim3(256), 0, st, weights, C, response ? 1 : 0, env);
  hipLaunchKernelGGL(light_env2_kernel, dim3((unsigned)((nq + 63) / 64)), dim3(64), 0, st, (const double*)env, C, env2);
  HIPCHK(hipGetLastError());
  const size_t lds = response ? ((size_t)Mt * LT_S + 128) * 8 + 64 * 64
                              : ((size_t)Mt * LT_S + 128) * 8 + (((size_t)Mt * LT_S + 1) & ~(size_t)1) * 4 + (size_t)LT_TAB * 64;
  const double* bmax = tmax + n;
  dim3 tg((unsigned)nblk, (unsigned)D);
  if (response) {
    if (lds > 65536) HIPCHK(hipFuncSetAttribute((const void*)light_truth_lds_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(light_truth_lds_kernel<true>, tg, dim3(64), lds, st, tid, tph, T, Mt, weights, C, (const double*)env, thr, out_tid,
                       out_tph, (const double*)tmax, bmax, (const double*)env2);
  } else {
    if (lds > 65536) HIPCHK(hipFuncSetAttribute((const void*)light_truth_lds_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(light_truth_lds_kernel<false>, tg, dim3(64), lds, st, tid, tph, T, Mt, weights, C, (const double*)env, thr, out_tid,
                       out_tph, (const double*)tmax, bmax, (const double*)env2);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C++" int light_response_launch(ldsim_ctx* ctx, bool response, const float* inc, const int64_t* tid,
                                       const double* tph, int D, int T, int Mt, const double* weights, int C,
                                       const double* gain, float* out, int64_t* out_tid, double* out_tph) {
  if (D <= 0 || T <= 0) return 0;
  if (light_truth_in_lds(ctx, Mt)) {
    // the plain sum by light_conv_kernel without truth, the truth slots by light_truth_lds_kernel (rows in LDS: no slot-major copies)
    int rc = light_response_plain(ctx, ctx->stream, response, inc, D, T, weights, C, gain, out);
    if (rc) return rc;
    return light_response_truth(ctx, ctx->stream, response, inc, tid, tph, D, T, Mt, weights, C, out_tid, out_tph);
  }
  dim3 grid((unsigned)((T + LR_THREADS - 1) / LR_THREADS), (unsigned)D), block(LR_THREADS);
  const double thr = ctx->h_consts.mc_truth_threshold;
  double* tmax = nullptr;
  if (Mt > 0) {
    const int64_t n = (int64_t)D * T;
    int rc = ldsim_ensure_buf(ctx, &ctx->light_tmax, (size_t)n * 8);
    if (rc) return rc;
    tmax = (double*)ctx->light_tmax.p;
    hipLaunchKernelGGL(light_truth_max_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, tid, tph, n, Mt,
                       response ? 1 : 0, tmax);
    HIPCHK(hipGetLastError());
  }
  // (more than 64 slots, or "light_truth_lds" 0:) the output's truth rows to slot-major, through light_conv_kernel, and back (2 x 2
  // passes over [D][T][Mt]: a few ms at 2.5 GB each)
  int64_t* w_tid = nullptr;
  int64_t* in_sm = nullptr;
  double* w_tph = nullptr;
  if (Mt > 0) {
    const size_t bt = (size_t)D * T * Mt;
    int rc = ldsim_ensure_buf(ctx, &ctx->light_wtid, bt * 8 + 16);
    if (rc) return rc;
    if ((rc = ldsim_ensure_buf(ctx, &ctx->light_wtph, bt * 8 + 16))) return rc;
    w_tid = (int64_t*)ctx->light_wtid.p;
    w_tph = (double*)ctx->light_wtph.p;
    dim3 tg((unsigned)((T + TR_TICKS - 1) / TR_TICKS), (unsigned)D);
    hipLaunchKernelGGL(light_truth_transpose_kernel<true>, tg, dim3(256), (size_t)TR_TICKS * (Mt + 1) * 8, ctx->stream,
                       (const unsigned long long*)out_tid, (unsigned long long*)w_tid, T, Mt);
    hipLaunchKernelGGL(light_truth_transpose_kernel<true>, tg, dim3(256), (size_t)TR_TICKS * (Mt + 1) * 8, ctx->stream,
                       (const unsigned long long*)out_tph, (unsigned long long*)w_tph, T, Mt);
    if (response) {        // the SiPM stage compares ids of the INPUT row at the output tick (light_sim.py:331-335): slot-major too
      if ((rc = ldsim_ensure_buf(ctx, &ctx->light_wtid2, bt * 8 + 16))) return rc;
      in_sm = (int64_t*)ctx->light_wtid2.p;
      hipLaunchKernelGGL(light_truth_transpose_kernel<true>, tg, dim3(256), (size_t)TR_TICKS * (Mt + 1) * 8, ctx->stream,
                         (const unsigned long long*)tid, (unsigned long long*)in_sm, T, Mt);
    }
    HIPCHK(hipGetLastError());
  }
  if (response)
    hipLaunchKernelGGL(light_conv_kernel<true>, grid, block, 0, ctx->stream, inc, tid, tph, D, T, Mt, weights, C, gain, thr,
                       out, w_tid, w_tph, tmax, in_sm, (const double*)nullptr);
  else
    hipLaunchKernelGGL(light_conv_kernel<false>, grid, block, 0, ctx->stream, inc, tid, tph, D, T, Mt, weights, C, gain,
                       thr, out, w_tid, w_tph, tmax, (const int64_t*)nullptr, (const double*)nullptr);
  HIPCHK(hipGetLastError());
  if (Mt > 0) {
    dim3 tg((unsigned)((T + TR_TICKS - 1) / TR_TICKS), (unsigned)D);
    hipLaunchKernelGGL(light_truth_transpose_kernel<false>, tg, dim3(256), (size_t)TR_TICKS * (Mt + 1) * 8, ctx->stream,
                       (const unsigned long long*)w_tid, (unsigned long long*)out_tid, T, Mt);
    hipLaunchKernelGGL(light_truth_transpose_kernel<false>, tg, dim3(256), (size_t)TR_TICKS * (Mt + 1) * 8, ctx->stream,
                       (const unsigned long long*)w_tph, (unsigned long long*)out_tph, T, Mt);
    HIPCHK(hipGetLastError());
  }
  return 0;
}
