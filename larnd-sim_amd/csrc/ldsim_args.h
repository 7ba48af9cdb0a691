// ldsim_args.h -- kernel argument blocks shared by the kernel translation units and the chain glue.
#pragma once
#include "ldsim_dev.h"

// Statistics of a launch: 16 u64 at `counters`, then STAT_STRIPES copies of the block that the kernels add to -- one
// workgroup, one stripe (blockIdx mod STAT_STRIPES).  A single address took 1.6 M device-scope atomics per launch from
// gcorr_kernel alone and cost it 2.7 ms of 11 (profiles/r03_phases_gform.log); the host sums the stripes after the download.
// Values read on the device or by a direct copy (sort: [4], pool cursor: [7]) stay in the flat block.
#define STAT_STRIPES 64
#define STAT_WORDS (16 * (1 + STAT_STRIPES))
#define STAT_BYTES (8 * STAT_WORDS)
#define MISC_BYTES (256 + STAT_BYTES + 256)
#ifdef __HIPCC__
__device__ __forceinline__ void stat_add(unsigned long long* counters, int i, unsigned long long v) {
  atomicAdd(&counters[16 * (1 + (blockIdx.x & (STAT_STRIPES - 1))) + i], v);
}
#endif
static inline void stat_sum(const unsigned long long* raw, unsigned long long* sum16) {
  for (int i = 0; i < 16; i++) {
    sum16[i] = raw[i];
    for (int s = 0; s < STAT_STRIPES; s++) sum16[i] += raw[16 * (1 + s) + i];
  }
}

struct CurArgs {
  SegStore s;
  const LdsimConsts* c;
  const double* resp;
  int32_t ni, nj, nk;
  int32_t k_first, k_last;       // response support (exact zeros outside), or [0, nk-1]
  const int32_t* pair_val;       // sorted pair list: value = r*P + ipix   (chain mode) or NULL (dense mode)
  const unsigned long long* pair_key;  // key carrying the pixel id            (chain mode)
  const int32_t* pixels;         // dense mode: pixels[S][P]
  int64_t seg_begin;             // first resident segment of this call (r is relative to it)
  int32_t P;
  int64_t n_pairs;
  float* out;                    // [n_pairs][T]
  int32_t T;                     // row stride == max ticks
  const int32_t* tmax_batch;     // per-batch max_length (chain) or NULL -> T
  int32_t batch0;
  double prune_log;
  double tail_log;        // split path: samples below exp(-tail_log) of the peak density are evaluated in f32 (0 = off)
  int32_t debug_phases;
  int32_t numba_f32;      // 1: the sub-expressions Numba types f32 for f4 record fields are evaluated in float (oracle: o_set_numba_f32)
  int32_t split_max_items;   // validation knob: pairs with more items than this take the monolithic kernel (0 = capacity)
  unsigned long long* counters;  // [0] ambiguous-rounding slices, [5] DFMA lanes, [6] pairs sent to the fallback
  int32_t* win;                  // [n_pairs][2] or NULL.  Set: a kernel that knows a pair's response-visible tick window writes it here
                                 // and leaves the ticks outside it unwritten (the chain's pixel sum reads the window only); NULL: every
                                 // row is written in full (zeros outside the window)
  const int32_t* only_flagged;   // if set: run only pairs whose only_flagged[pair*flag_stride + 7] != 0
  int32_t flag_stride;
  const int32_t* flag_list;      // if set: the flagged pairs as a list of *flag_count entries; a fixed grid walks it (no workgroup per pair
  const unsigned long long* flag_count;   //   that only finds its flag unset)
};

// The constants the FEE kernels use, by value (kernel arguments live in scalar registers): read through the pointer to the
// constants block in HBM every field costs a trip to L2, taken one after the other where the code first needs them -- a dozen
// dependent round trips per pixel in a kernel that is nothing but such chains.
struct FeeK {
  int32_t n_time_ticks, max_adc_values, max_tracks_per_pixel, pad;
  double time_sampling, buffer_risetime, clock_cycle, adc_hold_delay, reset_cycles, adc_busy_delay;
  double reset_noise_charge, uncorrelated_noise_charge, discriminator_noise;
  double v_pedestal, v_cm, v_ref, adc_counts, gain, time_interval1;
};
#define FEEK_FROM(h) FeeK{(h).n_time_ticks, (h).max_adc_values, (h).max_tracks_per_pixel, 0, (h).time_sampling, (h).buffer_risetime, \
                          (h).clock_cycle, (double)(h).adc_hold_delay, (double)(h).reset_cycles, (double)(h).adc_busy_delay,         \
                          (h).reset_noise_charge, (h).uncorrelated_noise_charge, (h).discriminator_noise, (h).v_pedestal, (h).v_cm, \
                          (h).v_ref, (double)(h).adc_counts, (h).gain, (h).time_interval[1]}

struct FeeArgs {
  const LdsimConsts* c;
  FeeK k;
  // unique pixels
  int64_t U;
  const int32_t* upix;
  const int32_t* ubatch;
  const int64_t* uoff;        // [U+1] offsets into the sorted pair list
  // sorted pairs
  const int32_t* pair_val;    // r*P + ipix
  const unsigned long long* pair_key;
  int32_t P;
  const double* track_starts; // [n_seg] relative index r
  const float* waves;         // [n_pairs][T]
  const int32_t* win;         // [n_pairs][2]: the ticks of a row that were written, or NULL = all
  int32_t T;
  const int32_t* batch_first; // [n_batches] first relative segment index of each batch
  int32_t batch0;
  double threshold;
  const double* thr_table;    // [n_pixel_ids] per-pixel thresholds (cli/simulate_pixels.py:1079-1084) or NULL -> threshold
  const double* gain_table;   // [n_pixel_ids] per-pixel gains (:1097-1100) or NULL -> GAIN * mV / e
  double time_padding;
  // outputs
  double* adc_list;           // [U][A]
  double* adc_ticks;          // [U][A]
  double* adc_digit;          // [U][A]
  int64_t* tpm;               // [U][M]
  double* fractions;          // [U][A][M] or NULL
  unsigned long long* counters;  // [2] overflow pixels, [3] hits
  int32_t* hit_count;         // [U]
  // FEE noise (fee.py:557,583-584,616-617,621,649): normals drawn ahead by fee_noise_kernel, or NULL = all noise charges 0
  const float* noise_z;       // [U][noise_nd]
  int32_t noise_nd;
  int32_t* n_draws;           // [U] normals the scan consumed
  int32_t debug;              // timing tools (debug_phases bits 0x10000 / 0x20000 / 0x40000: no waveform sum / scan / fractions)
};

int current_launch(ldsim_ctx* ctx, const CurArgs& args);
int current_mc_launch(ldsim_ctx* ctx, const CurArgs& args, int64_t n_seg);
int rng_ensure_states(ldsim_ctx* ctx, int64_t n);
int rng_fee_draws_per_pixel(const LdsimConsts& h, int NT);
int rng_launch_fee_noise(ldsim_ctx* ctx, int64_t U, int nd, float* z);
int rng_launch_advance(ldsim_ctx* ctx, int64_t U, const int32_t* n_draws);
int fee_launch_chain(ldsim_ctx* ctx, const FeeArgs& F);
int split_launch_weights(ldsim_ctx* ctx, const CurArgs& args, void* items, void* hdr, void* corr, double* wbuf,
                         unsigned long long wbuf_cap, unsigned long long* cursor);
int split_launch_mac(ldsim_ctx* ctx, const CurArgs& args, void* items, void* hdr, void* corr, double* wbuf,
                     unsigned long long wbuf_cap, unsigned long long* cursor);
struct SplitArgs;
int qweights_launch(ldsim_ctx* ctx, const SplitArgs& S, int M, void* params);
int qpair_setup_launch(ldsim_ctx* ctx, const SplitArgs& S, int M, void* params, void* ginfo, void* maps);
size_t qpair_params_bytes(int64_t n_pairs);
int split_sizes(const ldsim_ctx* ctx, const CurArgs& args, size_t* item_bytes, size_t* hdr_bytes, size_t* corr_bytes);
