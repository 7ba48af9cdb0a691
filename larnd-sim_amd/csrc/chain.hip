// chain.hip -- host orchestration of the device-resident charge chain (a5..a16) on the ctx stream.
// Mirrors the body of the reference's batch loop (cli/simulate_pixels.py:917-1105) for MANY batches at
// once: every (event, TPC-group, sub-batch) batch keeps its own unique-pixel set, max_length and
// segment numbering because the batch id is the leading key of the pair sort.
#include <stdio.h>
#include <math.h>

#include <vector>

#include "ldsim_args.h"
#include <utility>

int seg_launch_max_pixels(ldsim_ctx*, int64_t, int64_t, int32_t*, unsigned long long*);
int seg_launch_get_pixels(ldsim_ctx*, int64_t, int64_t, int, int32_t*, int, int32_t*, int32_t*, int, double*, const int32_t*,
                          int32_t);
int sort_make_keys(ldsim_ctx*, const int32_t*, const int32_t*, int64_t, int32_t, int, int64_t, unsigned long long*,
                   int32_t*, unsigned long long*);
int sort_pairs(ldsim_ctx*, unsigned long long*, unsigned long long*, int32_t*, int32_t*, int64_t);
int sort_pairs_bits(ldsim_ctx*, unsigned long long*, unsigned long long*, int32_t*, int32_t*, int64_t, int, int);
int sort_compact_valid(ldsim_ctx*, const unsigned long long*, int64_t, unsigned long long*, int32_t*, unsigned int*);
int sort_exclusive_scan_i32(ldsim_ctx*, const int32_t*, int32_t*, int64_t);
int sort_heads(ldsim_ctx*, const unsigned long long*, int64_t, int32_t*);
int sort_fill_unique(ldsim_ctx*, const unsigned long long*, const int32_t*, const int32_t*, int64_t, int32_t, int32_t*,
                     int32_t*, int64_t*, int64_t);
int sort_batch_first(ldsim_ctx*, int64_t, int64_t, int32_t, int32_t*);
int sort_tmax_batch(ldsim_ctx*, int64_t, int64_t, int32_t, double*, int32_t*, unsigned long long*);
int gform_launch(ldsim_ctx* ctx, const CurArgs& a, unsigned long long* counters, int32_t** flags_out,
                 const int32_t** flag_list, const unsigned long long** flag_count);
int sort_compact_hits(ldsim_ctx*, const int32_t*, const int32_t*, const int32_t*, const int32_t*, const double*,
                      const double*, int, int64_t, int32_t*);

#define CK(x)            \
  do {                   \
    int rc_ = (x);       \
    if (rc_) return rc_; \
  } while (0)

static void fill_cur_common(ldsim_ctx* ctx, CurArgs& a) {
  a.s = ctx->seg;
  a.c = ctx->d_consts;
  a.resp = ctx->d_resp;
  a.ni = ctx->ni; a.nj = ctx->nj; a.nk = ctx->nk;
  if (ctx->trim_response && ctx->resp_k_last >= ctx->resp_k_first) {
    a.k_first = ctx->resp_k_first;
    a.k_last = ctx->resp_k_last;
  } else {
    a.k_first = 0;
    a.k_last = ctx->nk - 1;
  }
  a.prune_log = ctx->prune_log;
  a.tail_log = ctx->tail_log;
  a.debug_phases = ctx->debug_phases;
  a.numba_f32 = ctx->numba_f32;
  a.split_max_items = ctx->split_max_items;
}

// a9-a12 for the pairs of `a` (sorted pair list of the chain, or the dense [S][P] grid of the stage call) by the configured
// kernels: tracks_current_mc, the split path (weights stage + correlation, overflowed pairs recomputed by the monolithic
// kernel) or the monolithic kernel for everything.  counters = the 16 u64 of the misc block, zeroed by the caller.
// a.win set: *win_used tells whether the kernels that ran wrote the windows (else every row is complete and a.win is unset).
static int run_tracks_current(ldsim_ctx* ctx, CurArgs& a, int64_t n_seg, unsigned long long* counters, bool* split_timed,
                              bool* win_used = nullptr) {
  hipStream_t st = ctx->stream;
  const int64_t n_valid = a.n_pairs;
  *split_timed = false;
  size_t ib = 0, hb = 0, cb = 0;
  int32_t* win = a.win;
  a.win = nullptr;
  if (win_used) *win_used = false;
  if (ctx->mc_current) return current_mc_launch(ctx, a, n_seg);   // the driver's call site (cli/simulate_pixels.py:1016)
  // The node-separable form pays per response tick of the staged support (nodes x cells x ticks on the matrix pipe), the
  // shifted-window kernels per 512-tick tile and (weight, tick) pair.  Until round 4 a table with full support (no exact zeros
  // over ~2000 ticks) was 20 % faster in the latter and "gform_max_support" = 768 ticks handed it over; with the 1e-7 node rule
  // (one batch of <= 16 nodes for most pairs) the matrix form is ahead there too -- module0 / 2x2 +18 %, ndlar +33 %
  // (profiles/r04_dense_handover.log) -- and the default no longer hands over.  weights_mode 1 / 0 force the shifted-window path,
  // a finite gform_max_support (ticks of TIME_SAMPLING) restores the hand-over.
  const int M_ratio = (int)llround(ctx->h_consts.time_sampling / ctx->h_consts.response_sampling);
  const int support_ticks = (a.k_last - a.k_first + 1) / (M_ratio > 0 ? M_ratio : 1);
  const bool use_gform = ctx->weights_mode == 2 && support_ticks <= ctx->gform_max_support;
  if (ctx->split_kernels && use_gform && n_valid > 0) {
    // node-separable form (gform.h): tables, then the correlation on the matrix pipe; no weight pool, no repeat launches
    int32_t* flags = nullptr;
    a.win = win;
    int rc = gform_launch(ctx, a, counters, &flags, &a.flag_list, &a.flag_count);
    a.win = nullptr;                       // (the monolithic kernel writes its rows in full)
    if (rc < 0) return rc;
    if (rc == 0) {
      *split_timed = true;
      if (win_used) *win_used = win != nullptr;
      a.only_flagged = flags - 7;          // current_kernel reads only_flagged[pair * flag_stride + 7]
      a.flag_stride = 1;
      return current_launch(ctx, a);
    }
    a.flag_list = nullptr;
    a.flag_count = nullptr;
  }
  if (!(ctx->split_kernels && n_valid > 0 && split_sizes(ctx, a, &ib, &hb, &cb) > 0)) return current_launch(ctx, a);
  CK(ldsim_ensure(ctx, SB_ITEMS, (size_t)n_valid * ib));
  CK(ldsim_ensure(ctx, SB_HDR, (size_t)n_valid * hb));
  CK(ldsim_ensure(ctx, SB_CORR, (size_t)n_valid * cb));
  // The weights go to one pool shared by all pairs of the launch (bump allocator, counters[7] = doubles requested).
  // Which pairs lose when it runs dry depends on scheduling order, and a pair recomputed by the monolithic kernel
  // agrees only to rounding -- so a launch that exhausted the pool is never used: the pool is grown to the demand
  // (a lower bound then: an exhausted pair stops requesting) and weights_kernel runs again.  The size per pair is
  // remembered, so this happens on the first launches of a new detector configuration only.
  double per_pair = fmax((double)ctx->wbuf_doubles_per_pair, ctx->wbuf_learned);
  unsigned long long wcap = 0, demand = 0;
  bool fits = false;
  for (int attempt = 0; attempt < 6 && !fits; attempt++) {
    wcap = (unsigned long long)ceil(per_pair * (double)n_valid);
    CK(ldsim_ensure(ctx, SB_WBUF, (size_t)wcap * 8));
    if (attempt > 0) {
      HIPCHK(hipMemsetAsync(&counters[7], 0, 8, st));                      // pool cursor
      HIPCHK(hipMemsetAsync(&counters[16], 0, STAT_BYTES - 128, st));     // the kernels' stripes (nothing earlier adds to them)
    }
    int rc = split_launch_weights(ctx, a, ctx->scratch[SB_ITEMS].p, ctx->scratch[SB_HDR].p, ctx->scratch[SB_CORR].p,
                                  (double*)ctx->scratch[SB_WBUF].p, wcap, &counters[7]);
    if (rc > 0) { ldsim_set_error("split path refused a configuration split_sizes accepted"); return LDSIM_ESTATE; }
    if (rc < 0) return rc;
    HIPCHK(hipMemcpyAsync(&demand, &counters[7], 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    fits = demand <= wcap;
    if (!fits) per_pair = 1.5 * (double)(demand > wcap ? demand : wcap) / (double)n_valid;
  }
  ctx->wbuf_learned = fmax(ctx->wbuf_learned, 1.25 * (double)demand / (double)n_valid);
  HIPCHK(hipEventRecord(ctx->ev[5], st));
  int rc = split_launch_mac(ctx, a, ctx->scratch[SB_ITEMS].p, ctx->scratch[SB_HDR].p, ctx->scratch[SB_CORR].p,
                            (double*)ctx->scratch[SB_WBUF].p, wcap, &counters[7]);
  if (rc > 0) { ldsim_set_error("split path refused a configuration split_sizes accepted"); return LDSIM_ESTATE; }
  if (rc < 0) return rc;
  HIPCHK(hipEventRecord(ctx->ev[6], st));
  *split_timed = true;
  // pairs beyond the per-pair item / correction / run capacities (and, after 6 attempts, a pool that still does not
  // fit): recomputed by the monolithic kernel
  a.only_flagged = (const int32_t*)ctx->scratch[SB_HDR].p;
  a.flag_stride = (int32_t)(hb / 4);
  return current_launch(ctx, a);
}

static void stats_from_counters(LdsimChainStats& s, const unsigned long long* h_cnt) {
  s.n_ambiguous = (int32_t)h_cnt[0];
  s.n_samples = (int64_t)h_cnt[1];
  s.n_dfma = (int64_t)h_cnt[5];
  s.n_fallback = (int64_t)h_cnt[6];
  s.n_wbuf = (int64_t)h_cnt[7];
  s.n_dfma_useful = (int64_t)h_cnt[8];
}

// materialising tracks_current: dense [S][P] pixel array, signals [S][P][T].  Runs the kernels the options select, like the
// chain: the per-tick parity tests reach the default (split) kernels through this call; ldsim_tracks_current_stats tells
// which kernels carried the pairs.
int chain_tracks_current(ldsim_ctx* ctx, const int32_t* d_pixels, int P, float* d_signals, int T, int mc) {
  CK(ldsim_ensure(ctx, SB_MISC, MISC_BYTES));
  unsigned long long* counters = (unsigned long long*)((char*)ctx->scratch[SB_MISC].p + 256);
  HIPCHK(hipMemsetAsync(counters, 0, STAT_BYTES, ctx->stream));
  CurArgs a{};
  fill_cur_common(ctx, a);
  a.pair_val = nullptr; a.pair_key = nullptr;
  a.pixels = d_pixels;
  a.seg_begin = 0;
  a.P = P;
  a.n_pairs = ctx->seg.n * (int64_t)P;
  a.out = d_signals;
  a.T = T;
  a.tmax_batch = nullptr;
  a.batch0 = 0;
  a.counters = counters;
  a.only_flagged = nullptr;
  a.flag_stride = 0;
  ctx->stage_stats = LdsimChainStats{};
  ctx->stage_stats.n_segments = ctx->seg.n;
  ctx->stage_stats.n_pairs = a.n_pairs;
  ctx->stage_stats.max_neigh = P;
  ctx->stage_stats.max_length = T;
  if (mc) return current_mc_launch(ctx, a, ctx->seg.n);
  bool split_timed = false;
  const int keep_mc = ctx->mc_current;
  ctx->mc_current = 0;                 // the stage call names its kernel itself
  int rc = run_tracks_current(ctx, a, ctx->seg.n, counters, &split_timed);
  ctx->mc_current = keep_mc;
  if (rc) return rc;
  unsigned long long h_cnt[16], h_raw[STAT_WORDS];
  HIPCHK(hipMemcpyAsync(h_raw, counters, STAT_BYTES, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  stat_sum(h_raw, h_cnt);
  stats_from_counters(ctx->stage_stats, h_cnt);
  return 0;
}

int chain_run(ldsim_ctx* ctx, int64_t seg_begin, int64_t seg_end, int want_fractions) {
  const LdsimConsts& h = ctx->h_consts;
  const int64_t n = seg_end - seg_begin;
  // results of the launch before this one may still be on their way to the host (ldsim_chain_download_async): this launch
  // writes the other set of output buffers; a copy that reads the set about to be written (two launches old) is waited for
  ctx->out_gen++;
  if (ctx->async_out) {
    if (ctx->copy_pending && ctx->pending_gen <= ctx->out_gen - 2) {
      HIPCHK(hipStreamSynchronize(ctx->copy_stream));
      ctx->copy_pending = 0;
    }
    static const int out_slots[7] = {SB_UPIX, SB_UBATCH, SB_ADC, SB_TICKS, SB_DIGIT, SB_TPM, SB_FRAC};
    for (int k = 0; k < 7; k++) std::swap(ctx->scratch[out_slots[k]], ctx->out_alt[k]);
  }
  ctx->stats = LdsimChainStats{};
  ctx->stats.n_segments = n;
  ctx->chain_U = 0;
  ctx->chain_hits = 0;
  ctx->want_fractions = want_fractions;
  ctx->ms_current = ctx->ms_adc = ctx->ms_total = 0;
  // per-pixel threshold / gain tables are dense over one pixel geometry: refuse another one before any work is launched
  const bool tables_fit = ctx->pix_table_n == (int64_t)h.n_pixels[0] * h.n_pixels[1] * h.n_tpc;
  if (!tables_fit && (ctx->d_pix_thr || ctx->d_pix_gain)) {
    ldsim_set_error("pixel threshold/gain tables were set for another pixel geometry: set them again after set_consts");
    return LDSIM_ESTATE;
  }
  if (n == 0) return 0;
  const int A = h.max_adc_values, M = h.max_tracks_per_pixel;
  hipStream_t st = ctx->stream;
  HIPCHK(hipEventRecord(ctx->ev[0], st));

  // ---- misc block: [0] err, [8] nmax i32, [16] tran bits u64, [256..] counters u64[16] ----------------------------
  CK(ldsim_ensure(ctx, SB_MISC, MISC_BYTES));
  char* misc = (char*)ctx->scratch[SB_MISC].p;
  unsigned long long* counters = (unsigned long long*)(misc + 256);
  HIPCHK(hipMemsetAsync(misc, 0, 256 + STAT_BYTES, st));

  // batch id range of this call: the non-negative ids are non-decreasing (checked at upload against the host copy kept
  // in the ctx), so the first and the last one are the range; negative = not simulated
  int32_t b_first = 0, b_last = 0;
  {
    if ((int64_t)ctx->h_batch.size() < seg_end) {
      ldsim_set_error("resident batch ids missing: ldsim_segments_upload first");
      return LDSIM_ESTATE;
    }
    const int32_t* hb = ctx->h_batch.data() + seg_begin;
    int64_t i0 = 0, i1 = n - 1;
    while (i0 < n && hb[i0] < 0) i0++;
    if (i0 == n) return 0;
    while (hb[i1] < 0) i1--;
    b_first = hb[i0];
    b_last = hb[i1];
    if (b_last < b_first) {
      ldsim_set_error("batch ids of the resident segments are not non-decreasing");
      return LDSIM_EINVAL;
    }
  }
  const int32_t batch0 = b_first;
  const int64_t n_batches = (int64_t)b_last - b_first + 1;
  ctx->stats.n_batches = n_batches;
  if (n_batches >= (1 << 27)) {
    ldsim_set_error("too many batches in one chain call");
    return LDSIM_EINVAL;
  }

  // ---- a8 time_intervals per batch, and the per-batch max tran_diff that sets max_radius -------------------------------
  CK(ldsim_ensure(ctx, SB_STARTS, (size_t)n * 8));
  CK(ldsim_ensure(ctx, SB_NLIST, (size_t)n_batches * 24 + 64));   // [nb] tmax i32 | [nb] first i32 | [nb] tran u64 | [nb] radius i32
  int32_t* d_tmax_b = (int32_t*)ctx->scratch[SB_NLIST].p;
  int32_t* d_first_b = d_tmax_b + n_batches;
  unsigned long long* d_tran_b = (unsigned long long*)((char*)ctx->scratch[SB_NLIST].p + ((n_batches * 8 + 15) / 16) * 16);
  int32_t* d_radius_b = (int32_t*)(d_tran_b + n_batches);
  HIPCHK(hipMemsetAsync(ctx->scratch[SB_NLIST].p, 0, (size_t)n_batches * 24 + 64, st));
  double* d_starts = (double*)ctx->scratch[SB_STARTS].p;
  CK(sort_tmax_batch(ctx, seg_begin, n, batch0, d_starts, d_tmax_b, d_tran_b));
  CK(sort_batch_first(ctx, seg_begin, n, batch0, d_first_b));

  // ---- a5 max_pixels (cli/simulate_pixels.py:918-928) ---------------------------------------------------------------------
  CK(seg_launch_max_pixels(ctx, seg_begin, seg_end, (int32_t*)(misc + 8), (unsigned long long*)(misc + 16)));
  int32_t nmax = 0;
  std::vector<unsigned long long> h_tran(n_batches);
  std::vector<int32_t> h_radius(n_batches);
  HIPCHK(hipMemcpyAsync(&nmax, misc + 8, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(h_tran.data(), d_tran_b, n_batches * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  int radius = 0;
  for (int64_t b = 0; b < n_batches; b++) {
    double mt;
    memcpy(&mt, &h_tran[b], 8);
    h_radius[b] = (int32_t)ceil(mt * 5 / h.pixel_pitch);      // max_radius of the batch
    radius = h_radius[b] > radius ? h_radius[b] : radius;
  }
  HIPCHK(hipMemcpyAsync(d_radius_b, h_radius.data(), n_batches * 4, hipMemcpyHostToDevice, st));
  const int P = (2 * radius + 1) * nmax + (1 + 2 * radius) * radius * 2;
  ctx->stats.max_active = nmax;
  ctx->stats.max_neigh = P;
  if (nmax == 0 || P == 0) return 0;

  // ---- a6 get_pixels -----------------------------------------------------------------------------------------------------
  const int64_t n_entries = n * P;
  if (n_entries >= 0x7fffffffLL) {
    ldsim_set_error("chain call too large: %lld (segment, pixel) slots; split the segment range", (long long)n_entries);
    return LDSIM_EINVAL;
  }
  CK(ldsim_ensure(ctx, SB_ACTIVE, (size_t)n * nmax * 4));
  CK(ldsim_ensure(ctx, SB_NEIGH, (size_t)n_entries * 4));
  CK(ldsim_ensure(ctx, SB_NRAD, (size_t)n_entries * 4));
  HIPCHK(hipMemsetAsync(ctx->scratch[SB_ACTIVE].p, 0xFF, (size_t)n * nmax * 4, st));
  HIPCHK(hipMemsetAsync(ctx->scratch[SB_NEIGH].p, 0xFF, (size_t)n_entries * 4, st));
  HIPCHK(hipMemsetAsync(ctx->scratch[SB_NRAD].p, 0xFF, (size_t)n_entries * 4, st));
  int32_t* d_neigh = (int32_t*)ctx->scratch[SB_NEIGH].p;
  int32_t* d_nrad = (int32_t*)ctx->scratch[SB_NRAD].p;
  CK(seg_launch_get_pixels(ctx, seg_begin, seg_end, radius, (int32_t*)ctx->scratch[SB_ACTIVE].p, nmax, d_neigh, d_nrad,
                           P, nullptr, d_radius_b, batch0));

  // ---- a7 unique pixels: stable radix sort of (batch, pixel, ring code) --------------------------------------------------------
  CK(ldsim_ensure(ctx, SB_KEYS, (size_t)n_entries * 8));
  CK(ldsim_ensure(ctx, SB_KEYS2, (size_t)n_entries * 8));
  CK(ldsim_ensure(ctx, SB_VALS, (size_t)n_entries * 4));
  CK(ldsim_ensure(ctx, SB_VALS2, (size_t)n_entries * 4));
  unsigned long long* d_keys = (unsigned long long*)ctx->scratch[SB_KEYS].p;
  unsigned long long* d_keys2 = (unsigned long long*)ctx->scratch[SB_KEYS2].p;
  int32_t* d_vals = (int32_t*)ctx->scratch[SB_VALS].p;
  int32_t* d_vals2 = (int32_t*)ctx->scratch[SB_VALS2].p;
  CK(sort_make_keys(ctx, d_neigh, d_nrad, seg_begin, batch0, P, n_entries, d_keys, d_vals, counters));
  // The empty slots (key ~0: three quarters of the entries of the module0 bench) are dropped before the sort -- a stable compaction, so
  // equal keys keep the order of their entries -- and the count comes back with the per-batch tick counts, before the sort instead
  // of after it.
  CK(sort_compact_valid(ctx, d_keys, n_entries, d_keys2, d_vals2, (unsigned int*)(misc + 64)));      // (its count: a word of its own; the launch uses counters[4])
  unsigned long long n_valid_ull = 0;
  std::vector<int32_t> h_tmax(n_batches);
  HIPCHK(hipMemcpyAsync(&n_valid_ull, &counters[4], 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(h_tmax.data(), d_tmax_b, n_batches * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  {
    // key = batch << 36 | pixel << 4 | ring code: the sort stops above the launch's batch count (20 batches: 41 bits, six 8-bit passes
    // instead of eight)
    int bb = 0;
    while ((1ll << bb) < n_batches) bb++;
    CK(sort_pairs_bits(ctx, d_keys2, d_keys, d_vals2, d_vals, (int64_t)n_valid_ull, 0, 36 + bb));
    std::swap(d_keys, d_keys2);         // (d_keys2 / d_vals2: the sorted list, as below)
    std::swap(d_vals, d_vals2);
  }
  const int64_t n_valid = (int64_t)n_valid_ull;
  ctx->stats.n_pairs = n_valid;
  int32_t T = 0;
  for (auto v : h_tmax) T = v > T ? v : T;
  ctx->stats.max_length = T;
  if (n_valid == 0 || T == 0) return 0;

  CK(ldsim_ensure(ctx, SB_HEADS, (size_t)n_valid * 4 + 16));
  CK(ldsim_ensure(ctx, SB_UOFF, (size_t)n_valid * 4 + 16));   // uidx (exclusive scan of heads)
  int32_t* d_heads = (int32_t*)ctx->scratch[SB_HEADS].p;
  int32_t* d_uidx = (int32_t*)ctx->scratch[SB_UOFF].p;
  CK(sort_heads(ctx, d_keys2, n_valid, d_heads));
  CK(sort_exclusive_scan_i32(ctx, d_heads, d_uidx, n_valid));
  int32_t last_idx = 0, last_head = 0;
  HIPCHK(hipMemcpyAsync(&last_idx, d_uidx + (n_valid - 1), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&last_head, d_heads + (n_valid - 1), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  const int64_t U = (int64_t)last_idx + last_head;
  ctx->stats.n_unique = U;
  ctx->chain_U = U;

  CK(ldsim_ensure(ctx, SB_UPIX, (size_t)U * 4));
  CK(ldsim_ensure(ctx, SB_UBATCH, (size_t)U * 4));
  CK(ldsim_ensure(ctx, SB_PAIRSEG, (size_t)(U + 1) * 8));     // uoff
  int32_t* d_upix = (int32_t*)ctx->scratch[SB_UPIX].p;
  int32_t* d_ubatch = (int32_t*)ctx->scratch[SB_UBATCH].p;
  int64_t* d_uoff = (int64_t*)ctx->scratch[SB_PAIRSEG].p;
  CK(sort_fill_unique(ctx, d_keys2, d_heads, d_uidx, n_valid, batch0, d_upix, d_ubatch, d_uoff, U));

  // ---- a9-a12 induced current per sorted pair -------------------------------------------------------------------------------------
  CK(ldsim_ensure(ctx, SB_WAVES, (size_t)n_valid * T * 4));
  float* d_waves = (float*)ctx->scratch[SB_WAVES].p;
  CurArgs a{};
  fill_cur_common(ctx, a);
  a.pair_val = d_vals2;
  a.pair_key = d_keys2;
  a.pixels = nullptr;
  a.seg_begin = seg_begin;
  a.P = P;
  a.n_pairs = n_valid;
  a.out = d_waves;
  a.T = T;
  a.tmax_batch = d_tmax_b;
  a.batch0 = batch0;
  a.counters = counters;
  a.only_flagged = nullptr;
  a.flag_stride = 0;
  CK(ldsim_ensure(ctx, SB_WIN, (size_t)n_valid * 8 + 8));
  a.win = (int32_t*)ctx->scratch[SB_WIN].p;
  HIPCHK(hipEventRecord(ctx->ev[1], st));
  bool split_timed = false, win_used = false;
  CK(run_tracks_current(ctx, a, n, counters, &split_timed, &win_used));
  HIPCHK(hipEventRecord(ctx->ev[2], st));

  // ---- a13-a16 per-pixel sum, trigger scan, digitise ---------------------------------------------------------------------------------
  CK(ldsim_ensure(ctx, SB_ADC, (size_t)U * A * 8));
  CK(ldsim_ensure(ctx, SB_TICKS, (size_t)U * A * 8));
  CK(ldsim_ensure(ctx, SB_DIGIT, (size_t)U * A * 8));
  CK(ldsim_ensure(ctx, SB_TPM, (size_t)U * M * 8));
  CK(ldsim_ensure(ctx, SB_PAIRPIX, (size_t)U * 8 + 16));      // hit_count i32 [U], hit_off i32 [U]
  if (want_fractions) CK(ldsim_ensure(ctx, SB_FRAC, (size_t)U * A * M * 8));
  int32_t* d_hitcnt = (int32_t*)ctx->scratch[SB_PAIRPIX].p;
  int32_t* d_hitoff = d_hitcnt + U;
  FeeArgs F{};
  F.c = ctx->d_consts;
  F.k = FEEK_FROM(h);
  F.U = U;
  F.upix = d_upix; F.ubatch = d_ubatch; F.uoff = d_uoff;
  F.pair_val = d_vals2; F.pair_key = d_keys2;
  F.P = P;
  F.track_starts = d_starts;
  F.waves = d_waves;
  F.win = win_used ? (const int32_t*)ctx->scratch[SB_WIN].p : nullptr;
  F.T = T;
  F.batch_first = d_first_b;
  F.batch0 = batch0;
  F.threshold = h.discrimination_threshold * 1.0;   // DISCRIMINATION_THRESHOLD * units.e
  F.thr_table = tables_fit ? ctx->d_pix_thr : nullptr;
  F.gain_table = tables_fit ? ctx->d_pix_gain : nullptr;
  F.time_padding = 0.0;                             // cli/simulate_pixels.py:1092
  F.adc_list = (double*)ctx->scratch[SB_ADC].p;
  F.adc_ticks = (double*)ctx->scratch[SB_TICKS].p;
  F.adc_digit = (double*)ctx->scratch[SB_DIGIT].p;
  F.tpm = (int64_t*)ctx->scratch[SB_TPM].p;
  F.fractions = want_fractions ? (double*)ctx->scratch[SB_FRAC].p : nullptr;
  F.counters = counters;
  F.hit_count = d_hitcnt;
  F.debug = ctx->debug_phases;
  // FEE noise (fee.py:557,583-584,616-617,621,649): row u of this launch draws from state u of the numba-style table
  const bool noisy = h.reset_noise_charge != 0 || h.uncorrelated_noise_charge != 0 || h.discriminator_noise != 0;
  if (noisy) {
    CK(rng_ensure_states(ctx, U));
    const int nd = rng_fee_draws_per_pixel(h, h.n_time_ticks);
    CK(ldsim_ensure(ctx, SB_NOISE, (size_t)U * nd * 4));
    CK(ldsim_ensure(ctx, SB_NDRAWS, (size_t)U * 4 + 4));
    CK(rng_launch_fee_noise(ctx, U, nd, (float*)ctx->scratch[SB_NOISE].p));
    F.noise_z = (const float*)ctx->scratch[SB_NOISE].p;
    F.noise_nd = nd;
    F.n_draws = (int32_t*)ctx->scratch[SB_NDRAWS].p;
  }
  CK(fee_launch_chain(ctx, F));
  if (noisy) CK(rng_launch_advance(ctx, U, F.n_draws));
  HIPCHK(hipEventRecord(ctx->ev[3], st));

  // ---- compact hit rows (payload of the multi-GPU all-gather) -----------------------------------------------------------------------------
  // (the rows are sized by their bound, U x MAX_ADC_VALUES -- 140 MB per 100 k segments -- so that the hit count comes back with
  // the launch's last synchronisation instead of one of its own)
  CK(sort_exclusive_scan_i32(ctx, d_hitcnt, d_hitoff, U));
  CK(ldsim_ensure(ctx, SB_HITS, (size_t)U * A * 24 + 24));
  CK(sort_compact_hits(ctx, d_upix, d_ubatch, d_hitcnt, d_hitoff, F.adc_digit, F.adc_ticks, A, U,
                       (int32_t*)ctx->scratch[SB_HITS].p));
  unsigned long long h_cnt[16], h_raw[STAT_WORDS];
  HIPCHK(hipMemcpyAsync(h_raw, counters, STAT_BYTES, hipMemcpyDeviceToHost, st));
  HIPCHK(hipEventRecord(ctx->ev[4], st));
  HIPCHK(hipStreamSynchronize(st));
  stat_sum(h_raw, h_cnt);
  if (ctx->debug_gform & 128)       // timing tools: cycle stamps of gcorr_kernel's waves (kernels_gcorr.hip)
    fprintf(stderr, "gcorr stamps (shader cycles, summed over waves): info %llu stage %llu G %llu P %llu edges %llu tail %llu life %llu\n",
            h_cnt[9], h_cnt[10], h_cnt[11], h_cnt[12], h_cnt[13], h_cnt[14], h_cnt[15]);
  if (ctx->debug_gform & 2048)      // the same stripes, written by gtables_wave_kernel's waves instead (kernels_gtables.hip)
    fprintf(stderr, "gtables stamps (shader cycles, summed over waves): loads %llu maps %llu batch_prologue %llu tables_xy %llu tables_z %llu cells %llu life %llu\n",
            h_cnt[9], h_cnt[10], h_cnt[11], h_cnt[12], h_cnt[13], h_cnt[14], h_cnt[15]);
  ctx->stats.n_overflow = (int64_t)h_cnt[2];
  ctx->chain_hits = (int64_t)h_cnt[3];
  stats_from_counters(ctx->stats, h_cnt);
  ctx->n_fallback = (int64_t)h_cnt[6];
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, ctx->ev[1], ctx->ev[2])); ctx->ms_current = ms;
  HIPCHK(hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3])); ctx->ms_adc = ms;
  ctx->ms_weights = ctx->ms_mac = 0;
  ctx->ms_fallback = ctx->ms_current;
  if (split_timed) {
    HIPCHK(hipEventElapsedTime(&ms, ctx->ev[1], ctx->ev[5])); ctx->ms_weights = ms;
    HIPCHK(hipEventElapsedTime(&ms, ctx->ev[5], ctx->ev[6])); ctx->ms_mac = ms;
    HIPCHK(hipEventElapsedTime(&ms, ctx->ev[6], ctx->ev[2])); ctx->ms_fallback = ms;
  }
  HIPCHK(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[4])); ctx->ms_total = ms;
  return 0;
}
