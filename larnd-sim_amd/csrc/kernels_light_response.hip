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
    const int64_t* __restrict__ tid_sm /* RESPONSE: the input ids slot-major, [D][Mt][T] (read at the OUTPUT tick) */) {
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
      if (Mt > 0) s_tmax[k] = truth_max[(int64_t)d * T + jc + k];
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

extern "C++" int light_response_launch(ldsim_ctx* ctx, bool response, const float* inc, const int64_t* tid,
                                       const double* tph, int D, int T, int Mt, const double* weights, int C,
                                       const double* gain, float* out, int64_t* out_tid, double* out_tph) {
  if (D <= 0 || T <= 0) return 0;
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
  // the output's truth rows: to slot-major, through the kernel, and back (2 x 2 passes over [D][T][Mt]: a few ms at 2.5 GB each)
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
                       out, w_tid, w_tph, tmax, in_sm);
  else
    hipLaunchKernelGGL(light_conv_kernel<false>, grid, block, 0, ctx->stream, inc, tid, tph, D, T, Mt, weights, C, gain,
                       thr, out, w_tid, w_tph, tmax, (const int64_t*)nullptr);
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
