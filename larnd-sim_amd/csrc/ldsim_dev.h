// ldsim_dev.h -- shared host/device declarations for libldsim_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <atomic>
#include <exception>
#include <thread>
#include <vector>

#include "../../include/ldsim.h"

#define LDSIM_WAVE 64

// ---- segment store: struct of arrays in HBM, one f64 column per hot-path field -------------------
// Values are held "as stored": after a kernel mutates a field the column holds the value narrowed
// through the record's storage dtype (f4 rounding / u4 truncation), exactly what the next reference
// kernel would read back from the record.
struct SegStore {
  double* f[LDSIM_NFIELDS - 1];  // all float-like fields (index = enum ldsim_field), n_electrons included
  int32_t* pixel_plane;
  int32_t* batch;                // (event, TPC-group, sub-batch) id; <0 = not simulated
  int32_t store_code[LDSIM_NFIELDS];
  int64_t n;
  int64_t cap;
};

// ---- Python / Numba scalar semantics on device -----------------------------------------------------
__device__ __forceinline__ double py_round(double x) { return rint(x); }  // half-to-even (v_rndne_f64)

// Python float floor division (CPython float_divmod algorithm, which Numba reproduces)
__device__ __forceinline__ double py_floordiv(double vx, double wx) {
  double mod = fmod(vx, wx);
  double div = (vx - mod) / wx;
  if (mod != 0.0) {
    if ((wx < 0) != (mod < 0)) {
      mod += wx;
      div -= 1.0;
    }
  }
  double fd;
  if (div != 0.0) {
    fd = floor(div);
    if (div - fd > 0.5) fd += 1.0;
  } else {
    fd = copysign(0.0, vx / wx);
  }
  return fd;
}

__device__ __forceinline__ int64_t ifloordiv(int64_t a, int64_t b) {
  int64_t q = a / b;
  if ((a % b != 0) && ((a < 0) != (b < 0))) q -= 1;
  return q;
}
__device__ __forceinline__ int64_t ifloormod(int64_t a, int64_t b) { return a - ifloordiv(a, b) * b; }

__device__ __forceinline__ double narrow_store(double v, int code) {
  switch (code) {
    case LDSIM_F4: return (double)(float)v;
    case LDSIM_I4: return (double)(int32_t)v;
    case LDSIM_U4: return (double)(uint32_t)v;
    case LDSIM_I8: return (double)(int64_t)v;
    case LDSIM_U8: return (double)(uint64_t)v;
    default: return v;
  }
}

__device__ __forceinline__ int64_t pixel2id(const LdsimConsts* c, int64_t px, int64_t py, int64_t plane) {
  return px + c->n_pixels[0] * (py + c->n_pixels[1] * plane);
}
__device__ __forceinline__ void id2pixel(const LdsimConsts* c, int64_t pid, int64_t& px, int64_t& py, int64_t& plane) {
  px = ifloormod(pid, c->n_pixels[0]);
  py = ifloormod(ifloordiv(pid, c->n_pixels[0]), c->n_pixels[1]);
  plane = ifloordiv(pid, (int64_t)c->n_pixels[0] * c->n_pixels[1]);
}
__device__ __forceinline__ bool pix_in_range(const LdsimConsts* c, int64_t x, int64_t y, int64_t plane) {
  return 0 <= x && x < c->n_pixels[0] && 0 <= y && y < c->n_pixels[1] && 0 <= plane && plane < c->n_tpc;
}

// ---- host-side context --------------------------------------------------------------------------------
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

// photon sum without truth slots over a device-built list of the lit (detector, tick tile) cells (kernels_light.hip)
#define LIGHT_TILE 2048      // ticks of a tile (16 KB of f64 sums in LDS: the sum kernel finds room beside the charge chain's workgroups)
struct LightAct {
  unsigned long long* dmask;    // [n_det] tiles of a detector row already listed
  unsigned* count;              // entries in `list`
  int32_t* list;                // idet * ntile + tile
  int ntile;
  // the sum before this one into the same array (the other half of the buffer): its listed tiles are zeroed again, its masks
  // and its count reset, by the kernels of this sum (clear = 0: nothing to undo, the array was cleared whole)
  int clear, p_ntile;
  long long p_nticks;
  unsigned long long* p_dmask;
  unsigned* p_count;
  const int32_t* p_list;
};

struct ldsim_ctx {
  // The thread inside an entry point of this ctx (CtxEnter below): a ctx is thread-compatible, not thread-safe -- a second
  // thread entering while a call is in progress gets LDSIM_ESTATE before it touches any buffer (two threads in one ctx once
  // freed the same scratch buffer twice under running kernels: GPU memory access fault, gpurun_out/r03_h25.log).
  std::atomic<std::thread::id> owner{};
  int owner_depth = 0;          // entry points call each other (ldsim_ctx_create -> ldsim_set_consts ...): re-entrant for the owner
  int device = 0;
  hipStream_t stream = nullptr;
  LdsimConsts h_consts;
  LdsimConsts* d_consts = nullptr;
  // response table
  double* d_resp = nullptr;
  int32_t ni = 0, nj = 0, nk = 0;
  int32_t resp_k_first = 0, resp_k_last = -1;  // support of the table over all cells (exact zeros outside)
  // per-pixel discrimination thresholds / gains of the fused chain: dense tables indexed by pixel id (NULL = constant)
  double* d_pix_thr = nullptr;
  double* d_pix_gain = nullptr;
  int64_t pix_table_n = 0;      // n_pixels[0] * n_pixels[1] * n_tpc the tables were built for
  // light
  double* d_eff = nullptr;
  int32_t* d_ch2tpc = nullptr;
  int32_t n_light_ch = 0;
  DevBuf resp_pad;                           // zero-padded copy of the response rows for mac_shift_kernel
  int32_t resp_pad_lo = 0, resp_pad_hi = -2; // staged range it was built for (-2: not built)
  double quad_n0 = 3.4, quad_slope = 1.38;   // Gauss-Legendre node rule N = ceil(n0 + slope * r): 1e-7 of the peak weight (option quad_accuracy_log10, tools/quad_sweep.py; 6 = 3.0 + 1.3 r is 3.7 % faster and fails the 2e-8 charge tests against the closed form)
  // overlapped download of the chain's results (ldsim_chain_download_async): a second stream copies launch k's per-pixel
  // arrays to the host while launch k + 1 computes into the other set of output buffers
  DevBuf light_wtid, light_wtph, light_wtid2;            // slot-major working copies of a response stage's output truth rows
  DevBuf light_xd;                           // the SiPM stage's input samples widened to doubles (light_widen_kernel)
  DevBuf light_env;                          // per offset of a wave's first tick from an input tick: the largest weight its 64 ticks meet it with (light_env_kernel)
  int light_truth_lds = 1;                   // truth slots of the light response stages by light_truth_lds_kernel (0: inside light_conv_kernel, slot-major copies)
  DevBuf light_tmax;                         // per (detector, tick) bound on the truth slots' photons (light_truth_max_kernel)
  hipStream_t copy_stream = nullptr;
  DevBuf out_alt[7];                         // the other set of SB_UPIX, SB_UBATCH, SB_ADC, SB_TICKS, SB_DIGIT, SB_TPM, SB_FRAC
  int async_out = 0;                         // alternate the output buffers from launch to launch (set by the first async download)
  int copy_pending = 0;
  int64_t out_gen = 0, pending_gen = 0;      // chain launches so far; the launch whose results the pending copy reads
  int light_eff_plain = 0;                   // every OP_CHANNEL_EFFICIENCY finite and >= 0
  int light_incidence_scalar = 0;            // 1 = the one-channel-per-lane light_incidence_kernel (A/B checks)
  int mac_mode = 1;                          // M = 1 correlation: 1 = mac_shift_kernel (DPP window), 0 = mac_kernel<1> (LDS rows)
  float *d_lut_vis = nullptr, *d_lut_t0 = nullptr, *d_lut_t0avg = nullptr, *d_lut_td = nullptr;
  int32_t lut_nx = 0, lut_ny = 0, lut_nz = 0, lut_ndet = 0, lut_nprof = 0;
  // options
  double prune_log = 23.0;                   // exp(-23) = 1e-10 of the pair's peak weight, the accuracy the node rule is fitted for
  double tail_log = 14.0;
  int trim_response = 1;
  double trim_response_log = 23.0;           // response ticks below exp(-v) of the table's largest entry are not read (0: exact zeros only)
  std::vector<double> h_resp_kmax;           // largest |entry| of every response tick over all cells (host)
  int debug_phases = 15;
  int debug_lds_b1_kb = 0;                   // timing tools: LDS budget of gcorr_kernel's second class in KB (0 = 32)
  int debug_lds_pad_kb = 0;                  // timing tools: KB taken off the LDS budget of gcorr_kernel's small class
  long long frac_clean_gen = -1;             // out_gen of the output set whose dense fractions array has been completed with zeros
  int fee_one_class = 0;                     // option: 1 = pixel_adc_kernel with 256 threads and the whole tick axis for every pixel (A/B checks)
  int gform_chunks = 1;                      // tables and correlation of the node-separable form in this many pair ranges, the tables of range c + 1 on a second stream beside the correlation of range c (1: one after the other)
  hipStream_t tab_stream = nullptr;          // that second stream, and its per-range events
  hipEvent_t tab_ev[34] = {};
  int gform_wave_tables = 1;                 // 1: gtables_wave_kernel (a wave per pair) for the pairs that fit it, 0: gtables_kernel for all
  int debug_gform = 0;                       // timing tools: parts of gtables_kernel / gcorr_kernel switched off (tools/gform_phases.py)
  int split_kernels = 1;            // 1: weights_kernel + mac_kernel (default), 0: monolithic current_kernel
  int wbuf_doubles_per_pair = 6144; // initial average budget of the split path's weight pool, doubles per pair
  int split_max_items = 0;          // validation knob, see CurArgs
  int weights_mode = 2;             // split path: 2 = node-separable form (gtables_kernel + gcorr_kernel, gform.h), 1 = qweights_kernel (Gauss-Legendre along the segment) + mac kernel, 0 = weights_kernel (per-sample closed form) + mac kernel
  int gform_max_support = 1000000000;   // staged response support (ticks) up to which weights_mode 2 runs the node-separable form (round 4: every table -- with the 1e-7 node rule
                                        // the matrix form is ahead on full-support tables too, profiles/r04_dense_handover.log; 768 until then: wider tables took the shifted-window kernels)
  int mc_current = 0;               // 1: the chain's induced currents come from current_mc_kernel (tracks_current_mc) instead of tracks_current
  int numba_f32 = 0;                // 1: evaluate the sub-expressions Numba types f32 for f4 record fields in float
  double* d_glx = nullptr;          // Gauss-Legendre nodes / weights on [-1, 1] for every N <= gl_nmax, rule N at N(N-1)/2
  double* d_glw = nullptr;
  int gl_nmax = 0;
  double wbuf_learned = 0;          // high-water demand per pair seen so far (x1.25): later calls size the pool with it
  int64_t n_fallback = 0;   // bit0: weights phase, bit1: correlation phase (timing experiments only)
  // resident segments
  SegStore seg{};
  DevBuf seg_block;
  DevBuf raw;          // AoS staging (H2D/D2H)
  LdsimTrackLayout seg_layout{};
  DevBuf scratch[40];  // named scratch buffers, grown on demand
  std::vector<int32_t> h_batch;   // host copy of the resident segments' batch ids (validated non-decreasing at upload)
  int seg_owner = 0;   // who filled the segment store last: 1 = ldsim_segments_upload (resident chain), 2 = a host-buffer stage call
  // device-resident light leg (ldsim_dev_light_incidence / ldsim_dev_sum_light)
  DevBuf light_nph, light_t0, light_vox;       // [n][light_n_out] f32, f32 (trigger mode 0 only), [n][3] i32 over all resident segments
  int32_t light_n_out = 0;
  int64_t light_n = -1;                        // resident segment count the incidence was computed for (-1 = not computed)
  DevBuf light_out, light_tid, light_tph, light_opc, light_trk;   // last photon sum: [n_det][n_ticks] f32, truth ids / photons
  int32_t light_sum_ndet = 0, light_sum_nticks = 0, light_sum_truth = 0;
  DevBuf light_tmp[9];
  // lazy clear of the resident photon-sum arrays (ldsim_dev_sum_light): valid = the arrays are clean over `light_clean_cells`
  // (detector, tick) cells except where the records still sorted in light_tmp[4] fell
  int light_lazy_valid = 0, light_lazy_mt = 0, light_lazy_tick_bits = 16;
  long long light_lazy_nrec = 0, light_lazy_nticks = 0;
  unsigned* light_flag_dev = nullptr;    // 8 bytes of device memory for that flag (sticky until read)
  unsigned* light_emit_flag = nullptr;   // device word of the last compact photon sum: a pair ran out of record slots (checked at the next synchronising light call)
  size_t light_clean_cells = 0, light_lazy_cap[3] = {0, 0, 0};
  void *light_lazy_out = nullptr, *light_lazy_tid = nullptr, *light_lazy_tph = nullptr;
  // the same for a sum without truth slots: clean except the tiles in light_act's list (LightAct; geometry of that sum below)
  DevBuf light_act;                            // two halves (this sum's, the previous one's) of: [cap] u64 masks, count u32, list i32
  int light_nt_valid = 0, light_nt_ntile = 0, light_nt_ndet_cap = 0, light_nt_half = 0, light_nt_list_cap = 0;
  long long light_nt_nticks = 0;
  void* light_nt_out = nullptr;
  size_t light_nt_cap = 0;
  int32_t h_opc_n_out = 0;
  int light_sum_no_list = 0;                   // option: 1 = the grid over every (detector, tile) and a full clear (A/B checks)
  std::vector<int32_t> h_opc;                  // host copy of light_opc's contents (validated once, re-sent only when it changes)
  // option light_sum_async: the sums without truth slots run on a stream of their own, beside whatever the ctx's stream carries
  // next (the charge chain).  light_pending: the last such sum has not been joined into ctx->stream yet (light_join, ldsim_abi.hip)
  hipStream_t light_stream = nullptr;
  hipEvent_t ev_light_in = nullptr, ev_light_done = nullptr;
  int light_async = 0, light_pending = 0;
  int light_sum_timed = 1;                     // 0: evl[2..3] of the last sum not yet read into ms_light_sum
  // resident waveform stages on the last photon sum (ldsim_dev_light_response): scintillation profile (+ truth), Poisson
  // fluctuated rate, detector response (+ truth), all [light_sum_ndet][light_sum_nticks]
  DevBuf light_scint, light_scint_tid, light_scint_tph, light_disc, light_resp, light_resp_tid, light_resp_tph, light_w[2],
      light_gain;
  int light_resp_valid = 0;
  uint64_t light_noise_calls = 0;
  double ms_light_resp[3] = {0, 0, 0};
  double ms_light_inc = 0, ms_light_sum = 0;
  hipEvent_t evl[4] = {nullptr, nullptr, nullptr, nullptr};
  // random streams (kernels_rng.hip): numba-style table of xoroshiro128p states, grown on demand
  DevBuf d_rng;
  int64_t rng_n = 0;
  uint64_t rng_seed = 0, rng_last_init[2] = {0, 0};
  int rng_seeded = 0;
  // multi-GPU exchange (comm.hip): RCCL communicator, rows accumulated over the chain calls of a pass, gathered rows
  void* comm = nullptr;
  int comm_rank = 0, comm_world = 0;
  DevBuf comm_tmp, hits_acc, hits_all;
  int64_t hits_acc_rows = 0;
  // chain results
  LdsimChainStats stats{};
  LdsimChainStats stage_stats{};   // counters of the last ldsim_tracks_current stage call (ldsim_tracks_current_stats)
  int64_t chain_U = 0, chain_hits = 0;
  int64_t cpt_n[4] = {0, 0, 0, 0};   // compact result of the last ldsim_chain_compact_build: hit pixels, hits, track entries, fraction entries
  int64_t cpt_gen = -1;              // chain launch it was built for
  int want_fractions = 0;
  hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  double ms_current = 0, ms_adc = 0, ms_total = 0;
  double ms_weights = 0, ms_mac = 0, ms_fallback = 0;   // split path: per-kernel share of ms_current
};

// scratch slots
enum {
  SB_ACTIVE = 0, SB_NEIGH, SB_NRAD, SB_NLIST, SB_STARTS, SB_MISC, SB_KEYS, SB_KEYS2, SB_VALS, SB_VALS2,
  SB_SORTTMP, SB_PAIRSEG, SB_PAIRPIX, SB_HEADS, SB_UOFF, SB_UPIX, SB_UBATCH, SB_WAVES, SB_ADC, SB_TICKS,
  SB_DIGIT, SB_TPM, SB_FRAC, SB_HITS, SB_ITEMS, SB_HDR, SB_CORR, SB_WBUF, SB_NOISE, SB_NDRAWS, SB_PPAR, SB_CPT, SB_CPO, SB_WIN, SB_SPAN, SB_GMAPS
};

void ldsim_set_error(const char* fmt, ...);
int ldsim_ensure(ldsim_ctx* ctx, int slot, size_t bytes);
int ldsim_ensure_buf(ldsim_ctx* ctx, DevBuf* b, size_t bytes);

// Claims the ctx for the calling thread for the duration of an extern "C" entry point.
struct CtxEnter {
  ldsim_ctx* c;
  bool ok;
  explicit CtxEnter(ldsim_ctx* ctx) : c(ctx), ok(true) {
    if (!c) return;                                  // (the entry point's own argument check reports a null ctx)
    const std::thread::id me = std::this_thread::get_id();
    if (c->owner.load(std::memory_order_acquire) == me) {
      c->owner_depth++;
      return;
    }
    std::thread::id none{};
    ok = c->owner.compare_exchange_strong(none, me, std::memory_order_acq_rel);
    if (ok) c->owner_depth = 1;
  }
  ~CtxEnter() {
    if (c && ok && --c->owner_depth == 0) c->owner.store(std::thread::id{}, std::memory_order_release);
  }
  CtxEnter(const CtxEnter&) = delete;
  CtxEnter& operator=(const CtxEnter&) = delete;
};
#define LDSIM_ENTER(ctx)                                                                                      \
  CtxEnter enter_(ctx);                                                                                       \
  if (!enter_.ok) {                                                                                           \
    ldsim_set_error("ctx in use by another thread (a ctx is thread-compatible, not thread-safe: one per thread)"); \
    return LDSIM_ESTATE;                                                                                      \
  }

#define HIPCHK(expr)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) {                                                               \
      ldsim_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return LDSIM_EHIP;                                                                  \
    }                                                                                     \
  } while (0)
