// ldsim_abi.hip -- the C-ABI of libldsim_hip.so (include/ldsim.h): context, host-buffer stage API,
// device-resident chain orchestration.  No compute happens on the host: every entry point only moves
// buffers and launches the HIP kernels in kernels_*.hip / sort.hip.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <vector>

#include "ldsim_args.h"

// ---- launchers defined in the kernel translation units ----------------------------------------------------
struct CurArgs;
struct FeeArgs;
int seg_launch_unpack(ldsim_ctx*, const LdsimTrackLayout*, int64_t);
int seg_launch_repack(ldsim_ctx*, const LdsimTrackLayout*, int64_t);
int seg_launch_quench_drift(ldsim_ctx*, int, int, int, int*);
int seg_launch_max_pixels(ldsim_ctx*, int64_t, int64_t, int32_t*, unsigned long long*);
int seg_launch_get_pixels(ldsim_ctx*, int64_t, int64_t, int, int32_t*, int, int32_t*, int32_t*, int, double*, const int32_t*,
                          int32_t);
int seg_launch_time_intervals(ldsim_ctx*, int64_t, int64_t, double*, int32_t*);
int fee_launch_track_pixel_map(ldsim_ctx*, int64_t*, const int32_t*, int64_t, const int32_t*, const int32_t*, int64_t,
                               int, int, int);
int fee_launch_sum_pixel_signals(ldsim_ctx*, double*, const float*, const double*, const int64_t*, const int64_t*,
                                 double*, double*, int64_t, int, int, int, int);
int fee_launch_adc_dense(ldsim_ctx*, const double*, const double*, int64_t, int, int, const double*, double, double,
                         const float*, int, int32_t*, double*, double*, double*);
int rng_ensure_states(ldsim_ctx* ctx, int64_t n);
int rng_fee_draws_per_pixel(const LdsimConsts& h, int NT);
int rng_launch_fee_noise(ldsim_ctx* ctx, int64_t U, int nd, float* z);
int rng_launch_advance(ldsim_ctx* ctx, int64_t U, const int32_t* n_draws);
int fee_launch_digitize(ldsim_ctx*, const double*, const double*, double*, int64_t);
int light_launch_incidence(ldsim_ctx*, int64_t, int64_t, int, float*, float*, int32_t*, int);
int light_launch_t0_range(ldsim_ctx*, const float*, const float*, int64_t, int*);
int light_launch_sum(ldsim_ctx*, int64_t, int64_t, const int32_t*, const int64_t*, const float*, int, const int32_t*, int,
                     const int32_t*, double, int64_t, float*, int64_t*, double*, int, int64_t*, const LightAct* = nullptr);
int light_launch_reset_cells(ldsim_ctx*, int64_t, int64_t, int, float*, int64_t*, double*);
int light_check_emit_overflow(ldsim_ctx*);
int sort_make_keys(ldsim_ctx*, const int32_t*, const int32_t*, int64_t, int32_t, int, int64_t, unsigned long long*,
                   int32_t*, unsigned long long*);
int sort_pairs(ldsim_ctx*, unsigned long long*, unsigned long long*, int32_t*, int32_t*, int64_t);
int sort_exclusive_scan_i32(ldsim_ctx*, const int32_t*, int32_t*, int64_t);
int sort_heads(ldsim_ctx*, const unsigned long long*, int64_t, int32_t*);
int sort_fill_unique(ldsim_ctx*, const unsigned long long*, const int32_t*, const int32_t*, int64_t, int32_t, int32_t*,
                     int32_t*, int64_t*, int64_t);
int sort_batch_first(ldsim_ctx*, int64_t, int64_t, int32_t, int32_t*);
int sort_compact_hits(ldsim_ctx*, const int32_t*, const int32_t*, const int32_t*, const int32_t*, const double*,
                      const double*, int, int64_t, int32_t*);
// chain glue implemented in chain.hip (needs the kernel argument structs)
int chain_run(ldsim_ctx* ctx, int64_t seg_begin, int64_t seg_end, int want_fractions);
extern "C++" int fee_clear_unwritten_fractions(ldsim_ctx* ctx, int64_t U, const int32_t* hit_count, const int64_t* tpm, double* fr);
// the dense fractions array of the current output set, complete (entries the FEE kernel did not write: zero)
static int fractions_complete(ldsim_ctx* ctx) {
  if (ctx->frac_clean_gen == ctx->out_gen) return 0;
  int rc = fee_clear_unwritten_fractions(ctx, ctx->chain_U, (const int32_t*)ctx->scratch[SB_PAIRPIX].p,
                                         (const int64_t*)ctx->scratch[SB_TPM].p, (double*)ctx->scratch[SB_FRAC].p);
  if (rc) return rc;
  ctx->frac_clean_gen = ctx->out_gen;
  return 0;
}
int chain_tracks_current(ldsim_ctx* ctx, const int32_t* d_pixels, int P, float* d_signals, int T, int mc);

// ---- errors ---------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void ldsim_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* ldsim_last_error(void) { return g_err; }
extern "C" int ldsim_abi_version(void) { return LDSIM_ABI_VERSION; }

extern "C" int ldsim_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int ldsim_ensure_buf(ldsim_ctx* ctx, DevBuf* b, size_t bytes) {
  (void)ctx;
  if (bytes <= b->bytes && b->p) return 0;
  if (b->p) HIPCHK(hipFree(b->p));
  b->p = nullptr;
  b->bytes = 0;
  size_t want = bytes + bytes / 8 + 256;
  HIPCHK(hipMalloc(&b->p, want));
  b->bytes = want;
  return 0;
}
int ldsim_ensure(ldsim_ctx* ctx, int slot, size_t bytes) { return ldsim_ensure_buf(ctx, &ctx->scratch[slot], bytes); }

// temporary device buffer for the host-buffer API
struct Tmp {
  void* p = nullptr;
  ~Tmp() {
    if (p) (void)hipFree(p);
  }
  int alloc(size_t bytes) {
    HIPCHK(hipMalloc(&p, bytes ? bytes : 8));
    return 0;
  }
  template <class T>
  T* as() { return (T*)p; }
};

#define CK(x)                \
  do {                       \
    int rc_ = (x);           \
    if (rc_) return rc_;     \
  } while (0)
#define NEED(cond, msg)            \
  do {                             \
    if (!(cond)) {                 \
      ldsim_set_error("%s", msg);  \
      return LDSIM_EINVAL;         \
    }                              \
  } while (0)

// Photon sums issued on the light stream (option light_sum_async) read the segment store, the incidence arrays, the LUT and the
// constants, and write the resident photon-sum array: every entry point that writes one of the former or reads the latter makes
// the ctx's stream wait for the last of them first.
static int light_join(ldsim_ctx* ctx) {
  if (ctx->light_pending) {
    HIPCHK(hipEventRecord(ctx->ev_light_done, ctx->light_stream));
    HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_light_done, 0));
    ctx->light_pending = 0;
  }
  return 0;
}

extern "C" int ldsim_host_alloc(void** p, size_t bytes) {
  NEED(p, "null argument");
  *p = nullptr;
  HIPCHK(hipHostMalloc(p, bytes ? bytes : 8, hipHostMallocDefault));
  return 0;
}
extern "C" int ldsim_host_free(void* p) {
  if (p) HIPCHK(hipHostFree(p));
  return 0;
}

// Gauss-Legendre nodes and weights of every rule N = 1..nmax (Newton on P_N from the Chebyshev guess, long double):
// rule N occupies entries [N(N-1)/2, N(N-1)/2 + N), nodes ascending.  Used by qweights_kernel (kernels_qweights.hip).
static int make_gl_tables(ldsim_ctx* ctx, int nmax) {
  const size_t total = (size_t)nmax * (nmax + 1) / 2;
  std::vector<double> x(total), w(total);
  const long double PI = 3.14159265358979323846264338327950288L;
  for (int n = 1; n <= nmax; n++) {
    const size_t off = (size_t)n * (n - 1) / 2;
    for (int k = 0; k < (n + 1) / 2; k++) {
      long double z = cosl(PI * (k + 0.75L) / (n + 0.5L)), pp = 1;
      for (int it = 0; it < 100; it++) {
        long double p1 = 1, p2 = 0;
        for (int j = 1; j <= n; j++) {
          long double p3 = p2;
          p2 = p1;
          p1 = ((2 * j - 1) * z * p2 - (j - 1) * p3) / j;
        }
        pp = n * (z * p1 - p2) / (z * z - 1);
        long double z1 = z;
        z = z1 - p1 / pp;
        if (fabsl(z - z1) < 1e-19L) break;
      }
      const long double wt = 2 / ((1 - z * z) * pp * pp);
      x[off + k] = (double)-z;
      x[off + n - 1 - k] = (double)z;
      w[off + k] = w[off + n - 1 - k] = (double)wt;
    }
  }
  HIPCHK(hipMalloc((void**)&ctx->d_glx, total * sizeof(double)));
  HIPCHK(hipMalloc((void**)&ctx->d_glw, total * sizeof(double)));
  HIPCHK(hipMemcpy(ctx->d_glx, x.data(), total * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(ctx->d_glw, w.data(), total * sizeof(double), hipMemcpyHostToDevice));
  ctx->gl_nmax = nmax;
  return 0;
}

// ---- context ------------------------------------------------------------------------------------------------------
static int ctx_init(ldsim_ctx* ctx, const LdsimConsts* consts) {
  HIPCHK(hipStreamCreate(&ctx->stream));
  HIPCHK(hipMalloc((void**)&ctx->d_consts, sizeof(LdsimConsts)));
  for (int i = 0; i < 8; i++) HIPCHK(hipEventCreate(&ctx->ev[i]));
  for (int i = 0; i < 4; i++) HIPCHK(hipEventCreate(&ctx->evl[i]));
  CK(make_gl_tables(ctx, 256));
  return ldsim_set_consts(ctx, consts);
}

extern "C" int ldsim_ctx_create(int device, const LdsimConsts* consts, ldsim_ctx** out) {
  NEED(consts && out, "null argument");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    ldsim_set_error("no HIP device visible: libldsim_hip.so has no CPU fallback");
    return LDSIM_ENODEV;
  }
  NEED(device >= 0 && device < n, "device index out of range");
  *out = nullptr;
  HIPCHK(hipSetDevice(device));
  ldsim_ctx* ctx = new (std::nothrow) ldsim_ctx();
  NEED(ctx, "out of host memory");
  ctx->device = device;
  const int rc = ctx_init(ctx, consts);
  if (rc) {                                  // nothing half-built is handed out or left behind
    (void)ldsim_ctx_destroy(ctx);
    return rc;
  }
  *out = ctx;
  return 0;
}

extern "C" int ldsim_set_consts(ldsim_ctx* ctx, const LdsimConsts* consts) {
  LDSIM_ENTER(ctx);
  NEED(ctx && consts, "null argument");
  CK(light_join(ctx));
  NEED(consts->n_tpc >= 0 && consts->n_tpc <= LDSIM_MAX_TPC, "n_tpc out of range");
  // another configuration has another demand on the split path's weight pool (ndlar needs ~4x module0's): relearn it
  // instead of keeping the high-water mark of everything this process has ever run
  if (memcmp(&ctx->h_consts, consts, sizeof(LdsimConsts)) != 0) ctx->wbuf_learned = 0;
  ctx->h_consts = *consts;
  HIPCHK(hipMemcpyAsync(ctx->d_consts, &ctx->h_consts, sizeof(LdsimConsts), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

extern "C" int ldsim_ctx_destroy(ldsim_ctx* ctx) {
  if (!ctx) return 0;
  {
    // a ctx another thread is inside is not torn down under it; the claim is kept until the object is gone
    const std::thread::id me = std::this_thread::get_id();
    std::thread::id none{};
    if (ctx->owner.load(std::memory_order_acquire) != me && !ctx->owner.compare_exchange_strong(none, me, std::memory_order_acq_rel)) {
      ldsim_set_error("ctx in use by another thread: not destroyed");
      return LDSIM_ESTATE;
    }
  }
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  void* ptrs[] = {ctx->d_consts, ctx->d_resp,      ctx->d_eff,    ctx->d_ch2tpc, ctx->d_lut_vis, ctx->d_lut_t0,
                  ctx->d_lut_t0avg, ctx->d_lut_td, ctx->seg_block.p, ctx->raw.p,  ctx->d_pix_thr, ctx->d_pix_gain,
                  ctx->d_glx, ctx->d_glw, ctx->light_flag_dev};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (ctx->copy_stream) {
    (void)hipStreamSynchronize(ctx->copy_stream);
    (void)hipStreamDestroy(ctx->copy_stream);
  }
  if (ctx->light_stream) {
    (void)hipStreamSynchronize(ctx->light_stream);
    (void)hipStreamDestroy(ctx->light_stream);
  }
  if (ctx->tab_stream) {
    (void)hipStreamSynchronize(ctx->tab_stream);
    (void)hipStreamDestroy(ctx->tab_stream);
  }
  for (auto& e : ctx->tab_ev)
    if (e) (void)hipEventDestroy(e);
  if (ctx->ev_light_in) (void)hipEventDestroy(ctx->ev_light_in);
  if (ctx->ev_light_done) (void)hipEventDestroy(ctx->ev_light_done);
  for (auto& b : ctx->scratch)
    if (b.p) (void)hipFree(b.p);
  for (auto& b : ctx->out_alt)
    if (b.p) (void)hipFree(b.p);
  for (int i = 0; i < 8; i++)
    if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
  for (int i = 0; i < 4; i++)
    if (ctx->evl[i]) (void)hipEventDestroy(ctx->evl[i]);
  for (DevBuf* b : {&ctx->light_nph, &ctx->light_t0, &ctx->light_vox, &ctx->light_out, &ctx->light_tid, &ctx->light_tph,
                    &ctx->light_opc, &ctx->light_trk, &ctx->light_scint, &ctx->light_scint_tid, &ctx->light_scint_tph,
                    &ctx->light_disc, &ctx->light_resp, &ctx->light_resp_tid, &ctx->light_resp_tph, &ctx->light_w[0],
                    &ctx->light_w[1], &ctx->light_gain, &ctx->resp_pad, &ctx->light_tmax, &ctx->light_env, &ctx->light_xd, &ctx->light_wtid, &ctx->light_wtph, &ctx->light_wtid2})
    if (b->p) (void)hipFree(b->p);
  for (auto& b : ctx->light_tmp)
    if (b.p) (void)hipFree(b.p);
  (void)ldsim_comm_destroy(ctx);
  if (ctx->d_rng.p) (void)hipFree(ctx->d_rng.p);
  for (DevBuf* b : {&ctx->comm_tmp, &ctx->hits_acc, &ctx->hits_all})
    if (b->p) (void)hipFree(b->p);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return 0;
}

extern "C" int ldsim_synchronize(ldsim_ctx* ctx) {
  LDSIM_ENTER(ctx);
  NEED(ctx, "null ctx");
  CK(light_join(ctx));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

static void resp_support_update(ldsim_ctx* ctx);

extern "C" int ldsim_set_option(ldsim_ctx* ctx, const char* name, double value) {
  LDSIM_ENTER(ctx);
  NEED(ctx && name, "null argument");
  if (!strcmp(name, "prune_log")) ctx->prune_log = value;
  else if (!strcmp(name, "tail_log")) ctx->tail_log = value;
  else if (!strcmp(name, "trim_response")) ctx->trim_response = value != 0;
  else if (!strcmp(name, "trim_response_log")) {
    if (!(value >= 0)) { ldsim_set_error("trim_response_log must be >= 0"); return LDSIM_EINVAL; }
    ctx->trim_response_log = value;
    if (!ctx->h_resp_kmax.empty()) resp_support_update(ctx);
  }
  else if (!strcmp(name, "debug_phases")) ctx->debug_phases = (int)value;
  else if (!strcmp(name, "debug_lds_b1_kb")) ctx->debug_lds_b1_kb = value >= 12 && value <= 64 ? (int)value : 0;
  else if (!strcmp(name, "debug_lds_pad_kb")) ctx->debug_lds_pad_kb = value >= -40 && value <= 16 ? (int)value : 0;
  else if (!strcmp(name, "debug_gform")) ctx->debug_gform = (int)value;
  else if (!strcmp(name, "gform_wave_tables")) ctx->gform_wave_tables = value != 0;
  else if (!strcmp(name, "gform_chunks")) {
    if (!(value >= 1 && value <= 32)) { ldsim_set_error("gform_chunks must be in [1, 32]"); return LDSIM_EINVAL; }
    ctx->gform_chunks = (int)value;
  }
  else if (!strcmp(name, "fee_one_class")) ctx->fee_one_class = value != 0;
  else if (!strcmp(name, "split_kernels")) ctx->split_kernels = value != 0;
  else if (!strcmp(name, "wbuf_doubles_per_pair")) { ctx->wbuf_doubles_per_pair = (int)value; ctx->wbuf_learned = 0; }
  else if (!strcmp(name, "split_max_items")) ctx->split_max_items = (int)value;
  else if (!strcmp(name, "weights_mode")) {
    if (!(value == 0 || value == 1 || value == 2)) { ldsim_set_error("weights_mode must be 0, 1 or 2"); return LDSIM_EINVAL; }
    ctx->weights_mode = (int)value;
    ctx->wbuf_learned = 0;
  }
  else if (!strcmp(name, "gform_max_support")) {
    if (!(value >= 0)) { ldsim_set_error("gform_max_support must be >= 0"); return LDSIM_EINVAL; }
    ctx->gform_max_support = value > 1e9 ? 1000000000 : (int)value;
  }
  else if (!strcmp(name, "light_truth_lds")) ctx->light_truth_lds = value != 0;
  else if (!strcmp(name, "light_incidence_scalar")) ctx->light_incidence_scalar = value != 0;
  else if (!strcmp(name, "light_sum_no_list")) ctx->light_sum_no_list = value != 0;
  else if (!strcmp(name, "light_sum_async")) {
    CK(light_join(ctx));
    ctx->light_async = value != 0;
  }
  else if (!strcmp(name, "mac_mode")) {
    if (!(value == 0 || value == 1)) { ldsim_set_error("mac_mode must be 0 or 1"); return LDSIM_EINVAL; }
    ctx->mac_mode = (int)value;
  }
  else if (!strcmp(name, "numba_f32")) ctx->numba_f32 = value != 0;
  else if (!strcmp(name, "mc_current")) ctx->mc_current = value != 0;
  else if (!strcmp(name, "quad_accuracy_log10")) {
    // relative quadrature error of the tables / weights stage as a power of ten of the peak weight (fits of tools/quad_nodes.py,
    // upper envelopes: N = ceil(n0 + slope r) nodes for a segment r Gaussian widths long)
    static const struct { int acc; double n0, slope; } rules[] = {
      {5, 2.8, 1.2}, {6, 3.0, 1.3}, {7, 3.4, 1.38}, {8, 3.8, 1.46}, {9, 4.4, 1.54}, {10, 4.8, 1.6}, {12, 6.0, 1.9}};
    bool ok = false;
    for (const auto& r : rules)
      if (value == r.acc) { ctx->quad_n0 = r.n0; ctx->quad_slope = r.slope; ok = true; }
    if (!ok) { ldsim_set_error("quad_accuracy_log10 must be 5, 6, 7, 8, 9, 10 or 12"); return LDSIM_EINVAL; }
  }
  else if (!strcmp(name, "quad_max_nodes")) {
    if (!(value >= 8 && value <= 256)) { ldsim_set_error("quad_max_nodes must be in [8, 256]"); return LDSIM_EINVAL; }
    ctx->gl_nmax = (int)value;
  }
  else {
    ldsim_set_error("unknown option %s", name);
    return LDSIM_EINVAL;
  }
  return 0;
}

// dense table over all pixel ids of the current geometry: the reference's CudaDict lookup with its default
static int set_pixel_table(ldsim_ctx* ctx, double** slot, const int32_t* keys, const double* values, int64_t n,
                           double default_value) {
  NEED(ctx && n >= 0 && (n == 0 || (keys && values)), "bad pixel table");
  HIPCHK(hipSetDevice(ctx->device));
  const LdsimConsts& h = ctx->h_consts;
  const int64_t n_ids = (int64_t)h.n_pixels[0] * h.n_pixels[1] * h.n_tpc;
  NEED(n_ids > 0, "pixel geometry not set");
  if (ctx->pix_table_n != n_ids) {        // tables of another geometry cannot be mixed with this one
    if (ctx->d_pix_thr) (void)hipFree(ctx->d_pix_thr);
    if (ctx->d_pix_gain) (void)hipFree(ctx->d_pix_gain);
    ctx->d_pix_thr = ctx->d_pix_gain = nullptr;
    ctx->pix_table_n = n_ids;
  }
  std::vector<double> tab((size_t)n_ids, default_value);
  for (int64_t i = 0; i < n; i++)
    if (keys[i] >= 0 && keys[i] < n_ids) tab[(size_t)keys[i]] = values[i];   // other keys can never be looked up
  if (!*slot) HIPCHK(hipMalloc((void**)slot, (size_t)n_ids * sizeof(double)));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(*slot, tab.data(), (size_t)n_ids * sizeof(double), hipMemcpyHostToDevice));
  return 0;
}

extern "C" int ldsim_set_pixel_thresholds(ldsim_ctx* ctx, const int32_t* keys, const double* values, int64_t n,
                                          double default_value) {
  LDSIM_ENTER(ctx);
  NEED(ctx, "null ctx");
  return set_pixel_table(ctx, &ctx->d_pix_thr, keys, values, n, default_value);
}

extern "C" int ldsim_set_pixel_gains(ldsim_ctx* ctx, const int32_t* keys, const double* values, int64_t n,
                                     double default_value) {
  LDSIM_ENTER(ctx);
  NEED(ctx, "null ctx");
  return set_pixel_table(ctx, &ctx->d_pix_gain, keys, values, n, default_value);
}

extern "C" int ldsim_clear_pixel_tables(ldsim_ctx* ctx) {
  LDSIM_ENTER(ctx);
  NEED(ctx, "null ctx");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (ctx->d_pix_thr) (void)hipFree(ctx->d_pix_thr);
  if (ctx->d_pix_gain) (void)hipFree(ctx->d_pix_gain);
  ctx->d_pix_thr = ctx->d_pix_gain = nullptr;
  ctx->pix_table_n = 0;
  return 0;
}

// Ticks of the response table that are read: [first, last] of the entries above the trim threshold -- exp(-trim_response_log)
// of the table's largest entry (default exp(-23) = 1e-10, the level the weights are pruned at; 0 = exact zeros only).  The
// ticks outside contribute less than that fraction of a waveform's peak, below the f4 resolution of the stored currents.
static void resp_support_update(ldsim_ctx* ctx) {
  const int nk = (int)ctx->h_resp_kmax.size();
  double vmax = 0;
  for (double v : ctx->h_resp_kmax) vmax = v > vmax ? v : vmax;
  const double thr = ctx->trim_response_log > 0 ? exp(-ctx->trim_response_log) * vmax : 0.0;
  int first = 0, last = nk - 1;
  while (first < nk && !(ctx->h_resp_kmax[first] > thr)) first++;
  while (last >= first && !(ctx->h_resp_kmax[last] > thr)) last--;
  if (last < first) { first = 0; last = -1; }
  ctx->resp_k_first = first;
  ctx->resp_k_last = last;
}

extern "C" int ldsim_set_response(ldsim_ctx* ctx, const double* response, int32_t ni, int32_t nj, int32_t nk) {
  LDSIM_ENTER(ctx);
  NEED(ctx && response && ni > 0 && nj > 0 && nk > 0, "bad response table");
  HIPCHK(hipSetDevice(ctx->device));
  if (ctx->d_resp) HIPCHK(hipFree(ctx->d_resp));
  size_t bytes = (size_t)ni * nj * nk * sizeof(double);
  HIPCHK(hipMalloc((void**)&ctx->d_resp, bytes));
  HIPCHK(hipMemcpy(ctx->d_resp, response, bytes, hipMemcpyHostToDevice));
  ctx->ni = ni; ctx->nj = nj; ctx->nk = nk;
  // largest |entry| of every tick over all cells: the support of the table along k follows from it for any trim threshold
  ctx->h_resp_kmax.assign((size_t)nk, 0.0);
  for (int64_t c = 0; c < (int64_t)ni * nj; c++) {
    const double* row = response + c * nk;
    for (int k = 0; k < nk; k++) {
      const double v = fabs(row[k]);
      if (v > ctx->h_resp_kmax[k]) ctx->h_resp_kmax[k] = v;      // (NaN entries never raise a maximum: such ticks count as empty)
    }
  }
  resp_support_update(ctx);
  ctx->resp_pad_hi = -2;                     // the padded copy of mac_shift_kernel is rebuilt on the next launch
  return 0;
}

extern "C" int ldsim_set_light_channels(ldsim_ctx* ctx, const double* eff, const int32_t* ch2tpc, int32_t n) {
  LDSIM_ENTER(ctx);
  NEED(ctx && n >= 0, "bad light channels");
  CK(light_join(ctx));
  if (ctx->d_eff) HIPCHK(hipFree(ctx->d_eff));
  if (ctx->d_ch2tpc) HIPCHK(hipFree(ctx->d_ch2tpc));
  ctx->d_eff = nullptr; ctx->d_ch2tpc = nullptr; ctx->n_light_ch = n;
  if (n == 0) return 0;
  HIPCHK(hipMalloc((void**)&ctx->d_eff, n * sizeof(double)));
  HIPCHK(hipMalloc((void**)&ctx->d_ch2tpc, n * sizeof(int32_t)));
  HIPCHK(hipMemcpy(ctx->d_eff, eff, n * sizeof(double), hipMemcpyHostToDevice));
  ctx->light_eff_plain = 1;             // finite, non-negative efficiencies: what light_incidence4_kernel's zero fill assumes
  for (int32_t i = 0; i < n; i++)
    if (!(eff[i] >= 0.0 && eff[i] <= 1.7e308)) ctx->light_eff_plain = 0;
  HIPCHK(hipMemcpy(ctx->d_ch2tpc, ch2tpc, n * sizeof(int32_t), hipMemcpyHostToDevice));
  return 0;
}

extern "C" int ldsim_set_light_lut(ldsim_ctx* ctx, const float* vis, const float* t0, const float* t0_avg,
                                   const float* time_dist, int32_t nx, int32_t ny, int32_t nz, int32_t ndet,
                                   int32_t nprof) {
  LDSIM_ENTER(ctx);
  NEED(ctx && vis && t0 && t0_avg && time_dist, "null LUT plane");
  CK(light_join(ctx));
  size_t nv = (size_t)nx * ny * nz * ndet;
  float** dst[] = {&ctx->d_lut_vis, &ctx->d_lut_t0, &ctx->d_lut_t0avg, &ctx->d_lut_td};
  const float* src[] = {vis, t0, t0_avg, time_dist};
  size_t cnt[] = {nv, nv, nv, nv * nprof};
  for (int i = 0; i < 4; i++) {
    if (*dst[i]) HIPCHK(hipFree(*dst[i]));
    HIPCHK(hipMalloc((void**)dst[i], cnt[i] * sizeof(float)));
    HIPCHK(hipMemcpy(*dst[i], src[i], cnt[i] * sizeof(float), hipMemcpyHostToDevice));
  }
  ctx->lut_nx = nx; ctx->lut_ny = ny; ctx->lut_nz = nz; ctx->lut_ndet = ndet; ctx->lut_nprof = nprof;
  return 0;
}

// The segment store is shared by the device-resident chain and the host-buffer stage calls (which upload their own
// records into it).  A chain entry point that finds the store filled by a stage call refuses to run on those records.
#define NEED_RESIDENT(ctx)                                                                                         \
  do {                                                                                                             \
    if ((ctx)->seg_owner != 1) {                                                                                   \
      ldsim_set_error("%s", (ctx)->seg_owner == 2                                                                  \
                                ? "a host-buffer stage call replaced the resident segments: ldsim_segments_upload again" \
                                : "no resident segments: call ldsim_segments_upload first");                         \
      return LDSIM_ESTATE;                                                                                         \
    }                                                                                                              \
  } while (0)

// ---- resident segments ----------------------------------------------------------------------------------------------
static int seg_reserve(ldsim_ctx* ctx, int64_t n) {
  if (n > ctx->seg.cap) {
    int64_t cap = n + n / 8 + 64;
    size_t bytes = (size_t)cap * ((LDSIM_NFIELDS - 1) * sizeof(double) + 2 * sizeof(int32_t));
    CK(ldsim_ensure_buf(ctx, &ctx->seg_block, bytes));
    char* p = (char*)ctx->seg_block.p;
    for (int f = 0; f < LDSIM_NFIELDS - 1; f++) {
      ctx->seg.f[f] = (double*)p;
      p += cap * sizeof(double);
    }
    ctx->seg.pixel_plane = (int32_t*)p;
    p += cap * sizeof(int32_t);
    ctx->seg.batch = (int32_t*)p;
    ctx->seg.cap = cap;
  }
  return 0;
}

// batch_id_is_resident: the call comes from ldsim_segments_upload (the store then belongs to the device-resident chain);
// every host-buffer stage call passes false and takes the store over
static int upload_tracks(ldsim_ctx* ctx, const void* tracks, int64_t n, const LdsimTrackLayout* lay,
                         const int32_t* batch_id, bool batch_id_is_resident = false) {
  NEED(ctx && lay && n >= 0 && (tracks || n == 0), "bad tracks argument");
  NEED(lay->itemsize > 0, "bad layout");
  HIPCHK(hipSetDevice(ctx->device));
  CK(light_join(ctx));
  CK(seg_reserve(ctx, n));
  ctx->seg.n = n;
  ctx->seg_owner = batch_id_is_resident ? 1 : 2;
  ctx->light_n = -1;
  ctx->seg_layout = *lay;
  for (int f = 0; f < LDSIM_NFIELDS; f++) ctx->seg.store_code[f] = lay->offset[f] >= 0 ? lay->dtype[f] : LDSIM_F8;
  if (n == 0) return 0;
  size_t bytes = (size_t)n * lay->itemsize;
  CK(ldsim_ensure_buf(ctx, &ctx->raw, bytes));
  HIPCHK(hipMemcpyAsync(ctx->raw.p, tracks, bytes, hipMemcpyHostToDevice, ctx->stream));
  CK(seg_launch_unpack(ctx, lay, n));
  if (batch_id)
    HIPCHK(hipMemcpyAsync(ctx->seg.batch, batch_id, n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  else
    HIPCHK(hipMemsetAsync(ctx->seg.batch, 0, n * sizeof(int32_t), ctx->stream));
  return 0;
}

static int download_tracks(ldsim_ctx* ctx, void* tracks, int64_t n, const LdsimTrackLayout* lay) {
  NEED(n == ctx->seg.n, "record count differs from the resident segment count");
  if (n == 0) return 0;
  size_t bytes = (size_t)n * lay->itemsize;
  NEED(ctx->raw.bytes >= bytes, "staging buffer missing");
  // the staging buffer still holds the uploaded bytes of every untouched field
  CK(seg_launch_repack(ctx, lay, n));
  HIPCHK(hipMemcpyAsync(tracks, ctx->raw.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

extern "C" int ldsim_segments_upload(ldsim_ctx* ctx, const void* tracks, int64_t n, const LdsimTrackLayout* layout,
                                     const int32_t* batch_id) {
  LDSIM_ENTER(ctx);
  NEED(ctx, "null ctx");
  // validated into a local copy: the ctx keeps describing the segments that are resident until the new ones really are
  std::vector<int32_t> ids;
  try {
    ids.assign((size_t)(n > 0 ? n : 0), 0);
  } catch (const std::exception&) {
    ldsim_set_error("out of host memory for %lld batch ids", (long long)n);
    return LDSIM_EINVAL;
  }
  if (batch_id) {
    int32_t last = -1;       // last non-negative id seen: ids < 0 (not simulated) may sit anywhere in between
    for (int64_t i = 0; i < n; i++) {
      if (batch_id[i] >= 0) {
        if (batch_id[i] < last) {
          ldsim_set_error("batch ids must be non-decreasing (segment %lld has %d after %d)", (long long)i, batch_id[i], last);
          return LDSIM_EINVAL;
        }
        last = batch_id[i];
      }
      ids[(size_t)i] = batch_id[i];
    }
  }
  {
    const int rc = upload_tracks(ctx, tracks, n, layout, batch_id, true);
    if (rc) {                 // a failed upload may have overwritten part of the store: nothing is resident any more
      ctx->seg_owner = 0;
      ctx->h_batch.clear();
      return rc;
    }
  }
  ctx->h_batch.swap(ids);
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

extern "C" int ldsim_segments_download(ldsim_ctx* ctx, void* tracks, int64_t n, const LdsimTrackLayout* layout) {
  LDSIM_ENTER(ctx);
  NEED(ctx && tracks && layout, "null argument");
  NEED_RESIDENT(ctx);
  return download_tracks(ctx, tracks, n, layout);
}

extern "C" int ldsim_segments_reset(ldsim_ctx* ctx) {
  LDSIM_ENTER(ctx);
  NEED(ctx, "null ctx");
  NEED_RESIDENT(ctx);
  NEED(ctx->seg.n == 0 || ctx->raw.p, "no uploaded records");
  HIPCHK(hipSetDevice(ctx->device));
  CK(light_join(ctx));
  ctx->light_n = -1;
  return seg_launch_unpack(ctx, &ctx->seg_layout, ctx->seg.n);
}

static int run_quench_drift(ldsim_ctx* ctx, int mode, int do_q, int do_d) {
  CK(ldsim_ensure(ctx, SB_MISC, MISC_BYTES));
  int* d_err = (int*)ctx->scratch[SB_MISC].p;
  HIPCHK(hipMemsetAsync(d_err, 0, sizeof(int), ctx->stream));
  CK(seg_launch_quench_drift(ctx, mode, do_q, do_d, d_err));
  int h_err = 0;
  HIPCHK(hipMemcpyAsync(&h_err, d_err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (h_err == 1) {
    ldsim_set_error("Invalid recombination mode: must be 'physics.BOX' or 'physics.BIRKS'");
    return LDSIM_EINVAL;
  }
  if (h_err == 2) {
    ldsim_set_error("Invalid recombination value");
    return LDSIM_EINVAL;
  }
  return 0;
}

extern "C" int ldsim_dev_quench_drift(ldsim_ctx* ctx, int32_t mode) {
  LDSIM_ENTER(ctx);
  NEED(ctx, "null ctx");
  NEED_RESIDENT(ctx);
  CK(light_join(ctx));
  return run_quench_drift(ctx, mode, 1, 1);
}

// ---- (1) stage-by-stage host-buffer API ---------------------------------------------------------------------------
extern "C" int ldsim_quench(ldsim_ctx* ctx, void* tracks, int64_t n, const LdsimTrackLayout* layout, int32_t mode) {
  LDSIM_ENTER(ctx);
  CK(upload_tracks(ctx, tracks, n, layout, nullptr));
  CK(run_quench_drift(ctx, mode, 1, 0));
  return download_tracks(ctx, tracks, n, layout);
}

extern "C" int ldsim_drift(ldsim_ctx* ctx, void* tracks, int64_t n, const LdsimTrackLayout* layout) {
  LDSIM_ENTER(ctx);
  CK(upload_tracks(ctx, tracks, n, layout, nullptr));
  CK(run_quench_drift(ctx, 2, 0, 1));
  return download_tracks(ctx, tracks, n, layout);
}

extern "C" int ldsim_max_pixels(ldsim_ctx* ctx, const void* tracks, int64_t n, const LdsimTrackLayout* layout,
                                int64_t* n_max_pixels) {
  LDSIM_ENTER(ctx);
  NEED(n_max_pixels, "null output");
  CK(upload_tracks(ctx, tracks, n, layout, nullptr));
  CK(ldsim_ensure(ctx, SB_MISC, MISC_BYTES));
  char* m = (char*)ctx->scratch[SB_MISC].p;
  HIPCHK(hipMemsetAsync(m, 0, 64, ctx->stream));
  CK(seg_launch_max_pixels(ctx, 0, n, (int32_t*)(m + 8), (unsigned long long*)(m + 16)));
  int32_t h = 0;
  HIPCHK(hipMemcpyAsync(&h, m + 8, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (h > *n_max_pixels) *n_max_pixels = h;  // cuda.atomic.max into the caller's array
  return 0;
}

extern "C" int ldsim_get_pixels(ldsim_ctx* ctx, const void* tracks, int64_t n, const LdsimTrackLayout* layout,
                                int32_t radius, int32_t* active, int32_t max_active, int32_t* neigh, int32_t* nrad,
                                int32_t P, double* n_list) {
  LDSIM_ENTER(ctx);
  NEED(active && neigh && nrad && max_active >= 0 && P >= 0 && radius >= 0, "bad get_pixels arguments");
  CK(upload_tracks(ctx, tracks, n, layout, nullptr));
  size_t ba = (size_t)n * max_active * 4, bn = (size_t)n * P * 4;
  CK(ldsim_ensure(ctx, SB_ACTIVE, ba));
  CK(ldsim_ensure(ctx, SB_NEIGH, bn));
  CK(ldsim_ensure(ctx, SB_NRAD, bn));
  CK(ldsim_ensure(ctx, SB_NLIST, (size_t)n * 8));
  HIPCHK(hipMemsetAsync(ctx->scratch[SB_ACTIVE].p, 0xFF, ba, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->scratch[SB_NEIGH].p, 0xFF, bn, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->scratch[SB_NRAD].p, 0xFF, bn, ctx->stream));
  CK(seg_launch_get_pixels(ctx, 0, n, radius, (int32_t*)ctx->scratch[SB_ACTIVE].p, max_active,
                           (int32_t*)ctx->scratch[SB_NEIGH].p, (int32_t*)ctx->scratch[SB_NRAD].p, P,
                           (double*)ctx->scratch[SB_NLIST].p, nullptr, 0));
  if (ba) HIPCHK(hipMemcpyAsync(active, ctx->scratch[SB_ACTIVE].p, ba, hipMemcpyDeviceToHost, ctx->stream));
  if (bn) HIPCHK(hipMemcpyAsync(neigh, ctx->scratch[SB_NEIGH].p, bn, hipMemcpyDeviceToHost, ctx->stream));
  if (bn) HIPCHK(hipMemcpyAsync(nrad, ctx->scratch[SB_NRAD].p, bn, hipMemcpyDeviceToHost, ctx->stream));
  if (n_list && n) HIPCHK(hipMemcpyAsync(n_list, ctx->scratch[SB_NLIST].p, n * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

extern "C" int ldsim_time_intervals(ldsim_ctx* ctx, const void* tracks, int64_t n, const LdsimTrackLayout* layout,
                                    double* track_starts, int64_t* time_max) {
  LDSIM_ENTER(ctx);
  NEED(track_starts && time_max, "null output");
  CK(upload_tracks(ctx, tracks, n, layout, nullptr));
  CK(ldsim_ensure(ctx, SB_STARTS, (size_t)n * 8 + 8));
  CK(ldsim_ensure(ctx, SB_MISC, MISC_BYTES));
  char* m = (char*)ctx->scratch[SB_MISC].p;
  HIPCHK(hipMemsetAsync(m, 0, 64, ctx->stream));
  CK(seg_launch_time_intervals(ctx, 0, n, (double*)ctx->scratch[SB_STARTS].p, (int32_t*)(m + 24)));
  int32_t h = 0;
  if (n) HIPCHK(hipMemcpyAsync(track_starts, ctx->scratch[SB_STARTS].p, n * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(&h, m + 24, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (h > *time_max) *time_max = h;
  return 0;
}

extern "C" int ldsim_tracks_current(ldsim_ctx* ctx, const void* tracks, int64_t n, const LdsimTrackLayout* layout,
                                    const int32_t* pixels, int32_t P, float* signals, int32_t T) {
  LDSIM_ENTER(ctx);
  NEED(pixels && signals && P >= 0 && T >= 0, "bad tracks_current arguments");
  NEED(ctx && ctx->d_resp, "no response table set (ldsim_set_response)");
  CK(upload_tracks(ctx, tracks, n, layout, nullptr));
  size_t bp = (size_t)n * P * 4, bs = (size_t)n * P * T * 4;
  if (bs == 0) return 0;
  CK(ldsim_ensure(ctx, SB_NEIGH, bp));
  CK(ldsim_ensure(ctx, SB_WAVES, bs));
  HIPCHK(hipMemcpyAsync(ctx->scratch[SB_NEIGH].p, pixels, bp, hipMemcpyHostToDevice, ctx->stream));
  CK(chain_tracks_current(ctx, (const int32_t*)ctx->scratch[SB_NEIGH].p, P, (float*)ctx->scratch[SB_WAVES].p, T, 0));
  HIPCHK(hipMemcpyAsync(signals, ctx->scratch[SB_WAVES].p, bs, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

extern "C" int ldsim_tracks_current_stats(ldsim_ctx* ctx, LdsimChainStats* stats) {
  LDSIM_ENTER(ctx);
  NEED(ctx && stats, "null argument");
  *stats = ctx->stage_stats;
  return 0;
}

extern "C" int ldsim_tracks_current_mc(ldsim_ctx* ctx, const void* tracks, int64_t n, const LdsimTrackLayout* layout,
                                       const int32_t* pixels, int32_t P, float* signals, int32_t T) {
  LDSIM_ENTER(ctx);
  NEED(pixels && signals && P >= 0 && T >= 0, "bad tracks_current_mc arguments");
  NEED(ctx && ctx->d_resp, "no response table set (ldsim_set_response)");
  CK(upload_tracks(ctx, tracks, n, layout, nullptr));
  size_t bp = (size_t)n * P * 4, bs = (size_t)n * P * T * 4;
  if (bs == 0) return 0;
  CK(ldsim_ensure(ctx, SB_NEIGH, bp));
  CK(ldsim_ensure(ctx, SB_WAVES, bs));
  HIPCHK(hipMemcpyAsync(ctx->scratch[SB_NEIGH].p, pixels, bp, hipMemcpyHostToDevice, ctx->stream));
  CK(chain_tracks_current(ctx, (const int32_t*)ctx->scratch[SB_NEIGH].p, P, (float*)ctx->scratch[SB_WAVES].p, T, 1));
  HIPCHK(hipMemcpyAsync(signals, ctx->scratch[SB_WAVES].p, bs, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

extern "C" int ldsim_track_pixel_map(ldsim_ctx* ctx, const int32_t* unique_pix, int64_t U, const int32_t* pixels,
                                     const int32_t* distances, int64_t n, int32_t P, int32_t max_distance,
                                     int64_t* track_pixel_map, int32_t M) {
  LDSIM_ENTER(ctx);
  NEED(ctx && track_pixel_map && (unique_pix || U == 0), "bad track_pixel_map arguments");
  HIPCHK(hipSetDevice(ctx->device));
  Tmp du, dp, dd, dm;
  CK(du.alloc(U * 4)); CK(dp.alloc((size_t)n * P * 4)); CK(dd.alloc((size_t)n * P * 4)); CK(dm.alloc((size_t)U * M * 8));
  if (U) HIPCHK(hipMemcpyAsync(du.p, unique_pix, U * 4, hipMemcpyHostToDevice, ctx->stream));
  if (n * P) {
    HIPCHK(hipMemcpyAsync(dp.p, pixels, (size_t)n * P * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(dd.p, distances, (size_t)n * P * 4, hipMemcpyHostToDevice, ctx->stream));
  }
  if (U * M) HIPCHK(hipMemcpyAsync(dm.p, track_pixel_map, (size_t)U * M * 8, hipMemcpyHostToDevice, ctx->stream));
  CK(fee_launch_track_pixel_map(ctx, dm.as<int64_t>(), du.as<int32_t>(), U, dp.as<int32_t>(), dd.as<int32_t>(), n, P,
                                max_distance, M));
  if (U * M) HIPCHK(hipMemcpyAsync(track_pixel_map, dm.p, (size_t)U * M * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

extern "C" int ldsim_sum_pixel_signals(ldsim_ctx* ctx, const float* signals, int64_t n, int32_t P, int32_t T,
                                       const double* track_starts, const int64_t* pim, const int64_t* tpm, int32_t M,
                                       int64_t U, double* pixels_signals, double* pts, double* overflow) {
  LDSIM_ENTER(ctx);
  NEED(ctx && signals && track_starts && pim && tpm && pixels_signals && overflow, "null argument");
  HIPCHK(hipSetDevice(ctx->device));
  const int NT = ctx->h_consts.n_time_ticks;
  Tmp ds, dst, dpim, dtpm, dps, dpts, dov;
  size_t bs = (size_t)n * P * T * 4;
  CK(ds.alloc(bs)); CK(dst.alloc(n * 8)); CK(dpim.alloc((size_t)n * P * 8)); CK(dtpm.alloc((size_t)U * M * 8));
  CK(dps.alloc((size_t)U * NT * 8)); CK(dov.alloc(U * 8));
  if (pts) CK(dpts.alloc((size_t)U * NT * M * 8));
  HIPCHK(hipMemcpyAsync(ds.p, signals, bs, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(dst.p, track_starts, n * 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(dpim.p, pim, (size_t)n * P * 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(dtpm.p, tpm, (size_t)U * M * 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemsetAsync(dps.p, 0, (size_t)U * NT * 8, ctx->stream));
  HIPCHK(hipMemsetAsync(dov.p, 0, U * 8, ctx->stream));
  if (pts) HIPCHK(hipMemsetAsync(dpts.p, 0, (size_t)U * NT * M * 8, ctx->stream));
  CK(fee_launch_sum_pixel_signals(ctx, dps.as<double>(), ds.as<float>(), dst.as<double>(), dpim.as<int64_t>(),
                                  dtpm.as<int64_t>(), pts ? dpts.as<double>() : nullptr, dov.as<double>(), n, P, T, NT,
                                  M));
  HIPCHK(hipMemcpyAsync(pixels_signals, dps.p, (size_t)U * NT * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(overflow, dov.p, U * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (pts) HIPCHK(hipMemcpyAsync(pts, dpts.p, (size_t)U * NT * M * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

extern "C" int ldsim_get_adc_values(ldsim_ctx* ctx, const double* ps, const double* pts, int64_t U, int32_t NT,
                                    int32_t M, const double* time_ticks, int32_t n_time_ticks, double time_padding,
                                    const double* thresholds, double* adc_list, double* adc_ticks, double* fractions) {
  LDSIM_ENTER(ctx);
  NEED(ctx && ps && thresholds && adc_list && adc_ticks, "null argument");
  NEED(n_time_ticks == NT + 1 && time_ticks, "time_ticks must be linspace(0, stop, N_t+1)");
  HIPCHK(hipSetDevice(ctx->device));
  const LdsimConsts& h = ctx->h_consts;
  const bool noisy = h.reset_noise_charge != 0 || h.uncorrelated_noise_charge != 0 || h.discriminator_noise != 0;
  const int A = h.max_adc_values;
  Tmp dps, dpts, dthr, dadc, dtk, dfr, dz, dnd;
  CK(dps.alloc((size_t)U * NT * 8)); CK(dthr.alloc(U * 8)); CK(dadc.alloc((size_t)U * A * 8)); CK(dtk.alloc((size_t)U * A * 8));
  if (pts) CK(dpts.alloc((size_t)U * NT * M * 8));
  if (fractions) CK(dfr.alloc((size_t)U * A * M * 8));
  int nd = 0;
  if (noisy) {      // rng_states[ip] of the reference's call (fee.py:557): state ip of the table, advanced in place
    CK(rng_ensure_states(ctx, U));
    nd = rng_fee_draws_per_pixel(h, NT);
    CK(dz.alloc((size_t)U * nd * 4)); CK(dnd.alloc((size_t)U * 4 + 4));
    CK(rng_launch_fee_noise(ctx, U, nd, dz.as<float>()));
  }
  HIPCHK(hipMemcpyAsync(dps.p, ps, (size_t)U * NT * 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(dthr.p, thresholds, U * 8, hipMemcpyHostToDevice, ctx->stream));
  if (pts) HIPCHK(hipMemcpyAsync(dpts.p, pts, (size_t)U * NT * M * 8, hipMemcpyHostToDevice, ctx->stream));
  if (fractions) HIPCHK(hipMemsetAsync(dfr.p, 0, (size_t)U * A * M * 8, ctx->stream));
  // the kernel regenerates linspace(0, stop, N_t+1) from the caller's stop value (cli/simulate_pixels.py:1072)
  CK(fee_launch_adc_dense(ctx, dps.as<double>(), pts ? dpts.as<double>() : nullptr, U, NT, M, dthr.as<double>(),
                          time_padding, time_ticks[NT], noisy ? dz.as<float>() : nullptr, nd,
                          noisy ? dnd.as<int32_t>() : nullptr, dadc.as<double>(), dtk.as<double>(),
                          (fractions && pts) ? dfr.as<double>() : nullptr));
  if (noisy) CK(rng_launch_advance(ctx, U, dnd.as<int32_t>()));
  HIPCHK(hipMemcpyAsync(adc_list, dadc.p, (size_t)U * A * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(adc_ticks, dtk.p, (size_t)U * A * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (fractions) HIPCHK(hipMemcpyAsync(fractions, dfr.p, (size_t)U * A * M * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

extern "C" int ldsim_digitize(ldsim_ctx* ctx, const double* integral, int64_t n, const double* gain, double* adcs) {
  LDSIM_ENTER(ctx);
  NEED(ctx && integral && adcs, "null argument");
  HIPCHK(hipSetDevice(ctx->device));
  Tmp di, dg, dout;
  CK(di.alloc(n * 8)); CK(dout.alloc(n * 8));
  if (gain) CK(dg.alloc(n * 8));
  if (n) HIPCHK(hipMemcpyAsync(di.p, integral, n * 8, hipMemcpyHostToDevice, ctx->stream));
  if (gain && n) HIPCHK(hipMemcpyAsync(dg.p, gain, n * 8, hipMemcpyHostToDevice, ctx->stream));
  CK(fee_launch_digitize(ctx, di.as<double>(), gain ? dg.as<double>() : nullptr, dout.as<double>(), n));
  if (n) HIPCHK(hipMemcpyAsync(adcs, dout.p, n * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

extern "C" int ldsim_light_incidence(ldsim_ctx* ctx, const void* tracks, int64_t n, const LdsimTrackLayout* layout,
                                     int32_t n_out, float* nph, float* t0det, int32_t* voxel) {
  LDSIM_ENTER(ctx);
  NEED(ctx && nph && t0det && voxel && n_out >= 0, "null argument");
  NEED(ctx->d_lut_vis && ctx->d_eff, "light LUT / channel tables not set");
  NEED(n_out <= ctx->n_light_ch, "more output channels than light channels configured");
  CK(upload_tracks(ctx, tracks, n, layout, nullptr));
  Tmp dn, dt, dv;
  size_t bc = (size_t)n * n_out * 4;
  CK(dn.alloc(bc)); CK(dt.alloc(bc)); CK(dv.alloc((size_t)n * 12));
  HIPCHK(hipMemsetAsync(dn.p, 0, bc, ctx->stream));
  HIPCHK(hipMemsetAsync(dt.p, 0, bc, ctx->stream));
  HIPCHK(hipMemsetAsync(dv.p, 0, (size_t)n * 12, ctx->stream));
  CK(light_launch_incidence(ctx, 0, n, n_out, dn.as<float>(), dt.as<float>(), dv.as<int32_t>(), 0));
  if (bc) {
    HIPCHK(hipMemcpyAsync(nph, dn.p, bc, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(t0det, dt.p, bc, hipMemcpyDeviceToHost, ctx->stream));
  }
  if (n) HIPCHK(hipMemcpyAsync(voxel, dv.p, (size_t)n * 12, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

extern "C" int ldsim_sum_light_signals(ldsim_ctx* ctx, const void* tracks, int64_t n, const LdsimTrackLayout* layout,
                                       const int32_t* voxel, const int64_t* track_id, const float* nph, int32_t n_inc,
                                       const int32_t* op_channel, int32_t n_det, const int32_t* sorted_indices,
                                       double start_time, int32_t n_ticks, float* out, int64_t* true_id,
                                       double* true_ph, int32_t max_truth) {
  LDSIM_ENTER(ctx);
  NEED(ctx && voxel && track_id && nph && op_channel && sorted_indices && out, "null argument");
  NEED(ctx->d_lut_t0avg, "light LUT not set");
  NEED(max_truth == 0 || (true_id && true_ph), "truth arrays missing");
  CK(upload_tracks(ctx, tracks, n, layout, nullptr));
  Tmp dv, dti, dn, dop, dsi, dout, dtid, dtph;
  size_t bo = (size_t)n_det * n_ticks;
  CK(dv.alloc((size_t)n * 12)); CK(dti.alloc(n * 8)); CK(dn.alloc((size_t)n * n_inc * 4)); CK(dop.alloc(n_det * 4));
  CK(dsi.alloc((size_t)n_det * n * 4)); CK(dout.alloc(bo * 4)); CK(dtid.alloc(bo * max_truth * 8 + 8));
  CK(dtph.alloc(bo * max_truth * 8 + 8));
  if (n) {
    HIPCHK(hipMemcpyAsync(dv.p, voxel, (size_t)n * 12, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(dti.p, track_id, n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(dn.p, nph, (size_t)n * n_inc * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(dsi.p, sorted_indices, (size_t)n_det * n * 4, hipMemcpyHostToDevice, ctx->stream));
  }
  HIPCHK(hipMemcpyAsync(dop.p, op_channel, n_det * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(dout.p, out, bo * 4, hipMemcpyHostToDevice, ctx->stream));   // accumulates into the caller's array
  if (max_truth) {
    HIPCHK(hipMemcpyAsync(dtid.p, true_id, bo * max_truth * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(dtph.p, true_ph, bo * max_truth * 8, hipMemcpyHostToDevice, ctx->stream));
  }
  CK(light_launch_sum(ctx, 0, n, dv.as<int32_t>(), dti.as<int64_t>(), dn.as<float>(), n_inc, dop.as<int32_t>(), n_det,
                      dsi.as<int32_t>(), start_time, n_ticks, dout.as<float>(), dtid.as<int64_t>(), dtph.as<double>(),
                      max_truth, nullptr));
  ctx->light_lazy_valid = 0;          // (light_tmp[] no longer holds the resident sum's records)
  HIPCHK(hipMemcpyAsync(out, dout.p, bo * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (max_truth) {
    HIPCHK(hipMemcpyAsync(true_id, dtid.p, bo * max_truth * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(true_ph, dtph.p, bo * max_truth * 8, hipMemcpyDeviceToHost, ctx->stream));
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

// ---- device-resident light leg ------------------------------------------------------------------------------------------------
// a17 over all resident segments (cli/simulate_pixels.py:795-797, after quench + drift): n_photons_det / t0_det / voxel stay
// in HBM for the per-batch photon sums.
extern "C" int ldsim_dev_light_incidence(ldsim_ctx* ctx, int32_t n_out) {
  LDSIM_ENTER(ctx);
  NEED(ctx && n_out > 0, "bad argument");
  NEED_RESIDENT(ctx);
  NEED(ctx->d_lut_vis && ctx->d_eff, "light LUT / channel tables not set");
  NEED(n_out <= ctx->n_light_ch, "more output channels than light channels configured");
  HIPCHK(hipSetDevice(ctx->device));
  CK(light_join(ctx));
  const int64_t n = ctx->seg.n;
  const size_t bc = (size_t)n * n_out * 4;
  const bool trig0 = ctx->h_consts.light_trig_mode == 0;
  CK(ldsim_ensure_buf(ctx, &ctx->light_nph, bc));
  if (trig0) CK(ldsim_ensure_buf(ctx, &ctx->light_t0, bc));
  CK(ldsim_ensure_buf(ctx, &ctx->light_vox, (size_t)n * 12));
  HIPCHK(hipEventRecord(ctx->evl[0], ctx->stream));
  CK(light_launch_incidence(ctx, 0, n, n_out, (float*)ctx->light_nph.p, (float*)ctx->light_t0.p, (int32_t*)ctx->light_vox.p,
                            1));
  HIPCHK(hipEventRecord(ctx->evl[1], ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, ctx->evl[0], ctx->evl[1]));
  ctx->ms_light_inc = ms;
  ctx->light_n_out = n_out;
  ctx->light_n = n;
  return 0;
}

#define NEED_LIGHT_INC(ctx)                                                                                      \
  do {                                                                                                           \
    NEED_RESIDENT(ctx);                                                                                          \
    if ((ctx)->light_n != (ctx)->seg.n) {                                                                        \
      ldsim_set_error("light incidence of the resident segments not computed: ldsim_dev_light_incidence first"); \
      return LDSIM_ESTATE;                                                                                       \
    }                                                                                                            \
  } while (0)

extern "C" int ldsim_dev_light_incidence_download(ldsim_ctx* ctx, int64_t seg_begin, int64_t seg_end, float* nph,
                                                  float* t0det, int32_t* voxel) {
  LDSIM_ENTER(ctx);
  NEED(ctx, "null ctx");
  NEED_LIGHT_INC(ctx);
  NEED(seg_begin >= 0 && seg_end >= seg_begin && seg_end <= ctx->seg.n, "segment range outside the resident store");
  const int64_t n = seg_end - seg_begin;
  const size_t row = (size_t)ctx->light_n_out * 4;
  if (n == 0) return 0;
  if (nph) HIPCHK(hipMemcpyAsync(nph, (char*)ctx->light_nph.p + seg_begin * row, n * row, hipMemcpyDeviceToHost, ctx->stream));
  if (t0det) {
    NEED(ctx->h_consts.light_trig_mode == 0, "t0_det is only computed in trigger mode 0 (lightLUT.py:126-131)");
    HIPCHK(hipMemcpyAsync(t0det, (char*)ctx->light_t0.p + seg_begin * row, n * row, hipMemcpyDeviceToHost, ctx->stream));
  }
  if (voxel) HIPCHK(hipMemcpyAsync(voxel, (char*)ctx->light_vox.p + seg_begin * 12, n * 12, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

// light_sim.get_nticks needs min / max of t0_det over the entries with photons (light_sim.py:34-39); the host finishes
// the expression so that its scalar typing stays the caller's
extern "C" int ldsim_dev_light_t0_range(ldsim_ctx* ctx, int64_t seg_begin, int64_t seg_end, float* t0_min, float* t0_max,
                                        int32_t* any) {
  LDSIM_ENTER(ctx);
  NEED(ctx && t0_min && t0_max && any, "null argument");
  NEED_LIGHT_INC(ctx);
  NEED(seg_begin >= 0 && seg_end >= seg_begin && seg_end <= ctx->seg.n, "segment range outside the resident store");
  NEED(ctx->h_consts.light_trig_mode == 0, "t0_det is only computed in trigger mode 0");
  CK(ldsim_ensure_buf(ctx, &ctx->light_tmp[0], 64));
  const int64_t total = (seg_end - seg_begin) * ctx->light_n_out;
  const size_t off = (size_t)seg_begin * ctx->light_n_out;
  CK(light_launch_t0_range(ctx, (const float*)ctx->light_nph.p + off, (const float*)ctx->light_t0.p + off, total,
                           (int*)ctx->light_tmp[0].p));
  int res[3];
  HIPCHK(hipMemcpyAsync(res, ctx->light_tmp[0].p, sizeof(res), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  auto ord2f = [](int v) { int i = v >= 0 ? v : v ^ 0x7fffffff; float f; memcpy(&f, &i, 4); return f; };
  *any = res[2];
  *t0_min = res[2] ? ord2f(res[0]) : 0.f;
  *t0_max = res[2] ? ord2f(res[1]) : 0.f;
  return 0;
}

static int light_sum_time(ldsim_ctx* ctx) {
  if (ctx->light_sum_timed) return 0;
  HIPCHK(hipEventSynchronize(ctx->evl[3]));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, ctx->evl[2], ctx->evl[3]));
  ctx->ms_light_sum = ms;
  ctx->light_sum_timed = 1;
  return 0;
}

// a18 for the resident segments [seg_begin, seg_end) = one batch of the reference's loop (cli/simulate_pixels.py:1120-1153):
// light_sample_inc [n_det][n_ticks] f4 (+ truth slots) is zero / -1 initialised here and stays in HBM.
extern "C" int ldsim_dev_sum_light(ldsim_ctx* ctx, int64_t seg_begin, int64_t seg_end, const int32_t* op_channel,
                                   int32_t n_det, const int64_t* segment_track_id, int32_t max_truth, double start_time,
                                   int32_t n_ticks) {
  LDSIM_ENTER(ctx);
  NEED(ctx && op_channel && n_det > 0 && n_ticks >= 0 && max_truth >= 0, "bad argument");
  NEED_LIGHT_INC(ctx);
  NEED(ctx->d_lut_t0avg, "light LUT not set");
  NEED(seg_begin >= 0 && seg_end >= seg_begin && seg_end <= ctx->seg.n, "segment range outside the resident store");
  // (the driver passes the same channel list batch after batch: checked and sent when it changes)
  const bool same_opc = ctx->light_opc.p && (int64_t)ctx->h_opc.size() == n_det && ctx->h_opc_n_out == ctx->light_n_out &&
                        memcmp(ctx->h_opc.data(), op_channel, (size_t)n_det * 4) == 0;
  if (!same_opc)
    for (int i = 0; i < n_det; i++) NEED(op_channel[i] >= 0 && op_channel[i] < ctx->light_n_out, "op_channel outside the incidence array");
  HIPCHK(hipSetDevice(ctx->device));
  const int64_t n = seg_end - seg_begin;
  const size_t bo = (size_t)n_det * n_ticks;
  // Without truth slots the sum runs over a device-built list of the lit (detector, tick tile) cells (LightAct); the array
  // differs from zero in those tiles only, and the next such sum into the same buffer clears them instead of everything
  // (ndlar: 96 of 3360 rows are lit per batch, the whole array is 148 MB).
  const int ntile = (int)((n_ticks + LIGHT_TILE - 1) / LIGHT_TILE);
  const bool use_list = max_truth == 0 && n_ticks > 0 && ntile <= 64 && (int64_t)n_det * ntile < 0x7fffffffLL && !ctx->light_sum_no_list;
  // Option light_sum_async: such a sum goes to the light stream, ordered after everything the ctx's stream holds so far and
  // beside what it gets next (the charge chain of the same segments); light_join brings it back.  Everything below, helpers
  // included, launches on ctx->stream: it points at the light stream until the call returns.
  const bool on_light = use_list && ctx->light_async;
  struct StreamScope {
    ldsim_ctx* c;
    hipStream_t keep;
    bool on_light;
    ~StreamScope() {
      if (on_light) c->light_pending = 1;
      c->stream = keep;
    }
  };
  if (on_light) {
    if (!ctx->light_stream) {
      int prio_lo = 0, prio_hi = 0;          // (the small sums ahead of the charge chain's grids when both have workgroups to place)
      HIPCHK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
      HIPCHK(hipStreamCreateWithPriority(&ctx->light_stream, hipStreamNonBlocking, prio_hi));
    }
    if (!ctx->ev_light_in) HIPCHK(hipEventCreateWithFlags(&ctx->ev_light_in, hipEventDisableTiming));
    if (!ctx->ev_light_done) HIPCHK(hipEventCreateWithFlags(&ctx->ev_light_done, hipEventDisableTiming));
    // (the first sum since the last join waits for what the ctx's stream holds; while sums are pending, whatever writes their
    // inputs there joins first -- light_join -- so the following sums have nothing new to wait for)
    if (!ctx->light_pending) {
      HIPCHK(hipEventRecord(ctx->ev_light_in, ctx->stream));
      HIPCHK(hipStreamWaitEvent(ctx->light_stream, ctx->ev_light_in, 0));
    }
  } else {
    CK(light_join(ctx));
  }
  StreamScope scope{ctx, ctx->stream, on_light};
  if (on_light) ctx->stream = ctx->light_stream;
  hipStream_t st = ctx->stream;
  CK(ldsim_ensure_buf(ctx, &ctx->light_out, bo * 4 + 16));
  if (!same_opc) {
    ctx->h_opc.clear();
    CK(ldsim_ensure_buf(ctx, &ctx->light_opc, (size_t)n_det * 4));
    HIPCHK(hipMemcpyAsync(ctx->light_opc.p, op_channel, (size_t)n_det * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));              // (pageable source: the copy must have left it before the call returns)
    try {
      ctx->h_opc.assign(op_channel, op_channel + n_det);
    } catch (const std::exception&) {      // (nothing may throw across the C ABI; without the copy the list is sent again next time)
      ctx->h_opc.clear();
    }
    ctx->h_opc_n_out = ctx->light_n_out;
  }
  HIPCHK(hipEventRecord(ctx->evl[2], st));
  if (max_truth) {
    CK(ldsim_ensure_buf(ctx, &ctx->light_tid, bo * max_truth * 8 + 16));
    CK(ldsim_ensure_buf(ctx, &ctx->light_tph, bo * max_truth * 8 + 16));
    CK(ldsim_ensure_buf(ctx, &ctx->light_trk, (size_t)(n > 0 ? n : 1) * 8));
  }
  // The arrays start at 0 / -1.  After a truth-slot sum they differ from that only in the cells its records fell into (their sorted
  // keys are still in light_tmp[4]): when the same buffers are large enough for this batch those cells are reset instead of clearing
  // everything -- [n_det][n_ticks][50] i8 + f8 is 15 GB at 50 000 ticks, the cells written a few per cent of it.
  const bool lazy = max_truth > 0 && ctx->light_lazy_valid && ctx->light_lazy_mt == max_truth && ctx->light_lazy_out == ctx->light_out.p &&
                    ctx->light_lazy_tid == ctx->light_tid.p && ctx->light_lazy_tph == ctx->light_tph.p &&
                    ctx->light_lazy_cap[0] == ctx->light_out.bytes && ctx->light_lazy_cap[1] == ctx->light_tid.bytes &&
                    ctx->light_lazy_cap[2] == ctx->light_tph.bytes &&      // (a buffer that grew was reallocated: contents undefined)
                    ctx->light_clean_cells >= bo;
  LightAct act{};
  bool act_fresh = false;
  if (use_list) {
    // two halves of one buffer: [cap] u64 masks | count (16 bytes) | [list_cap] i32 -- this sum's and the previous one's
    if (n_det > ctx->light_nt_ndet_cap || (int64_t)n_det * ntile > ctx->light_nt_list_cap || !ctx->light_act.p) {
      const int cap_det = n_det > ctx->light_nt_ndet_cap ? n_det : ctx->light_nt_ndet_cap;
      const int64_t want_list = (int64_t)cap_det * (ntile > 8 ? ntile : 8);
      const int cap_list = (int)(want_list > ctx->light_nt_list_cap ? want_list : ctx->light_nt_list_cap);
      const size_t half = (((size_t)cap_det * 8 + 16 + (size_t)cap_list * 4 + 255) / 256) * 256;
      ctx->light_nt_ndet_cap = ctx->light_nt_list_cap = 0;
      ctx->light_nt_valid = 0;
      if (ctx->light_act.p) { HIPCHK(hipFree(ctx->light_act.p)); ctx->light_act.p = nullptr; ctx->light_act.bytes = 0; }
      CK(ldsim_ensure_buf(ctx, &ctx->light_act, 2 * half));
      ctx->light_nt_ndet_cap = cap_det;
      ctx->light_nt_list_cap = cap_list;
      act_fresh = true;
    }
    const size_t half = (((size_t)ctx->light_nt_ndet_cap * 8 + 16 + (size_t)ctx->light_nt_list_cap * 4 + 255) / 256) * 256;
    auto fill = [&](int h, unsigned long long** m, unsigned** c, int32_t** l) {
      char* base = (char*)ctx->light_act.p + (size_t)h * half;
      *m = (unsigned long long*)base;
      *c = (unsigned*)(base + (size_t)ctx->light_nt_ndet_cap * 8);
      *l = (int32_t*)(base + (size_t)ctx->light_nt_ndet_cap * 8 + 16);
    };
    const bool nt_lazy0 = !act_fresh && ctx->light_nt_valid && ctx->light_nt_out == ctx->light_out.p &&
                          ctx->light_nt_cap == ctx->light_out.bytes && ctx->light_clean_cells >= bo;
    const int cur = nt_lazy0 ? 1 - ctx->light_nt_half : 0;
    int32_t* pl = nullptr;
    fill(cur, &act.dmask, &act.count, &act.list);
    fill(1 - cur, &act.p_dmask, &act.p_count, &pl);
    act.p_list = pl;
    act.ntile = ntile;
    act.clear = nt_lazy0 ? 1 : 0;
    act.p_ntile = ctx->light_nt_ntile;
    act.p_nticks = ctx->light_nt_nticks;
    ctx->light_nt_half = cur;
  }
  const bool nt_lazy = use_list && act.clear;
  ctx->light_nt_valid = 0;
  // (nt_lazy: the previous sum's tiles are cleared by this sum's first kernel, beside its own marking pass)
  if (lazy) {
    CK(light_launch_reset_cells(ctx, ctx->light_lazy_nrec, ctx->light_lazy_nticks, max_truth, (float*)ctx->light_out.p,
                                (int64_t*)ctx->light_tid.p, (double*)ctx->light_tph.p));
  } else if (!nt_lazy) {
    if (use_list) {                        // masks and counts of both halves
      const size_t half = (((size_t)ctx->light_nt_ndet_cap * 8 + 16 + (size_t)ctx->light_nt_list_cap * 4 + 255) / 256) * 256;
      HIPCHK(hipMemsetAsync(ctx->light_act.p, 0, (size_t)ctx->light_nt_ndet_cap * 8 + 16, st));
      HIPCHK(hipMemsetAsync((char*)ctx->light_act.p + half, 0, (size_t)ctx->light_nt_ndet_cap * 8 + 16, st));
    }
    HIPCHK(hipMemsetAsync(ctx->light_out.p, 0, bo * 4, st));
    if (max_truth) {
      HIPCHK(hipMemsetAsync(ctx->light_tid.p, 0xFF, bo * max_truth * 8, st));     // -1
      HIPCHK(hipMemsetAsync(ctx->light_tph.p, 0, bo * max_truth * 8, st));
    }
    ctx->light_clean_cells = bo;
  }
  ctx->light_lazy_valid = 0;
  if (max_truth && n) {
    NEED(segment_track_id, "segment_track_id is needed for the truth slots");
    HIPCHK(hipMemcpyAsync(ctx->light_trk.p, segment_track_id, (size_t)n * 8, hipMemcpyHostToDevice, st));
  }
  int64_t n_rec = 0;
  CK(light_launch_sum(ctx, seg_begin, n, (const int32_t*)ctx->light_vox.p + seg_begin * 3, (const int64_t*)ctx->light_trk.p,
                      (const float*)ctx->light_nph.p + (size_t)seg_begin * ctx->light_n_out, ctx->light_n_out,
                      (const int32_t*)ctx->light_opc.p, n_det, nullptr, start_time, n_ticks, (float*)ctx->light_out.p,
                      (int64_t*)ctx->light_tid.p, (double*)ctx->light_tph.p, max_truth, &n_rec, use_list ? &act : nullptr));
  if (use_list) {                        // what the next such call has to undo
    ctx->light_nt_valid = 1;
    ctx->light_nt_ntile = ntile;
    ctx->light_nt_nticks = n_ticks;
    ctx->light_nt_out = ctx->light_out.p;
    ctx->light_nt_cap = ctx->light_out.bytes;
  }
  if (max_truth) {                       // what the next call has to undo
    ctx->light_lazy_valid = 1;
    ctx->light_lazy_mt = max_truth;
    ctx->light_lazy_nrec = n_rec;
    ctx->light_lazy_nticks = n_ticks;
    ctx->light_lazy_out = ctx->light_out.p; ctx->light_lazy_tid = ctx->light_tid.p; ctx->light_lazy_tph = ctx->light_tph.p;
    ctx->light_lazy_cap[0] = ctx->light_out.bytes; ctx->light_lazy_cap[1] = ctx->light_tid.bytes; ctx->light_lazy_cap[2] = ctx->light_tph.bytes;
  }
  HIPCHK(hipEventRecord(ctx->evl[3], st));
  // (the list-driven sum has nothing to bring back: the call returns with its kernels in flight, like a chain launch's stages;
  // ldsim_light_kernel_ms waits for the events when it is asked)
  ctx->light_sum_timed = 0;
  if (!use_list) CK(light_sum_time(ctx));
  ctx->light_sum_ndet = n_det; ctx->light_sum_nticks = n_ticks; ctx->light_sum_truth = max_truth;
  ctx->light_resp_valid = 0;
  return 0;
}

extern "C" int ldsim_dev_light_download(ldsim_ctx* ctx, float* light_sample_inc, int64_t* true_track_id,
                                        double* true_photons) {
  LDSIM_ENTER(ctx);
  NEED(ctx, "null ctx");
  CK(light_join(ctx));
  const size_t bo = (size_t)ctx->light_sum_ndet * ctx->light_sum_nticks;
  if (bo == 0) return 0;
  if (light_sample_inc) HIPCHK(hipMemcpyAsync(light_sample_inc, ctx->light_out.p, bo * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (ctx->light_sum_truth) {
    if (true_track_id)
      HIPCHK(hipMemcpyAsync(true_track_id, ctx->light_tid.p, bo * ctx->light_sum_truth * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (true_photons)
      HIPCHK(hipMemcpyAsync(true_photons, ctx->light_tph.p, bo * ctx->light_sum_truth * 8, hipMemcpyDeviceToHost, ctx->stream));
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  CK(light_check_emit_overflow(ctx));
  return 0;
}

extern "C" int ldsim_light_kernel_ms(ldsim_ctx* ctx, double* incidence_ms, double* sum_ms) {
  LDSIM_ENTER(ctx);
  NEED(ctx, "null ctx");
  if (incidence_ms) *incidence_ms = ctx->ms_light_inc;
  if (sum_ms) {
    CK(light_sum_time(ctx));
    *sum_ms = ctx->ms_light_sum;
  }
  return 0;
}

// ---- light waveform response: weight tables on the host, with the reference's expressions -----------------------------
int light_response_launch(ldsim_ctx* ctx, bool response, const float* inc, const int64_t* tid, const double* tph, int D,
                          int T, int Mt, const double* weights, int C, const double* gain, float* out, int64_t* out_tid,
                          double* out_tph);

static int64_t conv_ticks(const LdsimConsts& h) {   // light_sim.py:160 / :315
  return (int64_t)ceil((h.light_window[1] - h.light_window[0]) / h.light_tick_size);
}

// light_sim.scintillation_model(time_tick), :131-146
static double scintillation_model(int64_t n, const LdsimConsts& h) {
  const double tick = h.light_tick_size;
  double p1 = h.singlet_fraction * exp(-(double)n * tick / h.tau_s) * (1 - exp(-tick / h.tau_s));
  double p3 = (1 - h.singlet_fraction) * exp(-(double)n * tick / h.tau_t) * (1 - exp(-tick / h.tau_t));
  return (p1 + p3) * (n >= 0 ? 1.0 : 0.0);
}

// light_sim.interp(idx, arr, low, high), :241-271
static double interp_model(double idx, const double* arr, int64_t len, double low, double high) {
  int64_t i0 = (int64_t)floor(idx);
  if (i0 < 0) return low;
  if (i0 > len - 1) return high;
  if ((double)i0 == idx) return arr[i0];
  if (i0 > len - 2) return high;
  return arr[i0] + (arr[i0 + 1] - arr[i0]) * (idx - (double)i0);
}

// light_sim.sipm_response_model(idet, time_tick), :274-300 (idet is unused there as well)
static double sipm_response_model(int64_t n, const double* impulse, int64_t n_impulse, const LdsimConsts& h) {
  if (h.sipm_response_model == 0) {
    double t = (double)n * h.light_tick_size;
    double v = (t >= 0 ? 1.0 : 0.0) * exp(-t / h.light_response_time) * sin(t / h.light_oscillation_period);
    v /= h.light_oscillation_period * (h.light_response_time * h.light_response_time);
    v *= h.light_oscillation_period * h.light_oscillation_period + h.light_response_time * h.light_response_time;
    return v * h.light_tick_size;
  }
  double v = interp_model((double)n * h.light_tick_size / h.impulse_tick_size, impulse, n_impulse, 0, 0);
  v /= h.impulse_tick_size / h.light_tick_size;
  return v;
}

static int light_response_stage(ldsim_ctx* ctx, bool response, const float* inc, const int64_t* tid, const double* tph,
                                int32_t n_det, int32_t n_ticks, int32_t max_truth, const std::vector<double>& weights,
                                const double* gain, float* out, int64_t* out_tid, double* out_tph) {
  HIPCHK(hipSetDevice(ctx->device));
  const size_t bo = (size_t)n_det * n_ticks, bt = bo * (size_t)max_truth;
  Tmp dinc, dtid, dtph, dw, dg, dout, dotid, dotph;
  CK(dinc.alloc(bo * 4 + 8)); CK(dout.alloc(bo * 4 + 8)); CK(dw.alloc(weights.size() * 8));
  CK(dtid.alloc(bt * 8 + 8)); CK(dtph.alloc(bt * 8 + 8)); CK(dotid.alloc(bt * 8 + 8)); CK(dotph.alloc(bt * 8 + 8));
  CK(dg.alloc((size_t)n_det * 8 + 8));
  hipStream_t st = ctx->stream;
  HIPCHK(hipMemcpyAsync(dinc.p, inc, bo * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(dout.p, out, bo * 4, hipMemcpyHostToDevice, st));          // accumulates into the caller's array
  HIPCHK(hipMemcpyAsync(dw.p, weights.data(), weights.size() * 8, hipMemcpyHostToDevice, st));
  if (gain) HIPCHK(hipMemcpyAsync(dg.p, gain, (size_t)n_det * 8, hipMemcpyHostToDevice, st));
  if (max_truth) {
    HIPCHK(hipMemcpyAsync(dtid.p, tid, bt * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(dtph.p, tph, bt * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(dotid.p, out_tid, bt * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(dotph.p, out_tph, bt * 8, hipMemcpyHostToDevice, st));
  }
  CK(light_response_launch(ctx, response, dinc.as<float>(), dtid.as<int64_t>(), dtph.as<double>(), n_det, n_ticks,
                           max_truth, dw.as<double>(), (int)weights.size() - 1, dg.as<double>(), dout.as<float>(),
                           dotid.as<int64_t>(), dotph.as<double>()));
  HIPCHK(hipMemcpyAsync(out, dout.p, bo * 4, hipMemcpyDeviceToHost, st));
  if (max_truth) {
    HIPCHK(hipMemcpyAsync(out_tid, dotid.p, bt * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(out_tph, dotph.p, bt * 8, hipMemcpyDeviceToHost, st));
  }
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}

extern "C" int ldsim_scintillation_effect(ldsim_ctx* ctx, const float* light_sample_inc, const int64_t* true_track_id,
                                          const double* true_photons, int32_t n_det, int32_t n_ticks, int32_t max_truth,
                                          float* scint, int64_t* scint_true_track_id, double* scint_true_photons) {
  LDSIM_ENTER(ctx);
  NEED(ctx && light_sample_inc && scint && n_det >= 0 && n_ticks >= 0 && max_truth >= 0, "bad argument");
  NEED(max_truth == 0 || (true_track_id && true_photons && scint_true_track_id && scint_true_photons),
       "truth arrays missing");
  const LdsimConsts& h = ctx->h_consts;
  NEED(h.light_tick_size > 0 && h.tau_s > 0 && h.tau_t > 0, "light constants not set");
  const int64_t C = conv_ticks(h);
  NEED(C >= 0 && C < (1 << 24), "LIGHT_WINDOW / LIGHT_TICK_SIZE out of range");
  std::vector<double> w((size_t)C + 1);
  for (int64_t n = 0; n <= C; n++) w[(size_t)n] = scintillation_model(n, h);
  return light_response_stage(ctx, false, light_sample_inc, true_track_id, true_photons, n_det, n_ticks, max_truth, w,
                              nullptr, scint, scint_true_track_id, scint_true_photons);
}

extern "C" int ldsim_light_detector_response(ldsim_ctx* ctx, const float* light_sample_inc, const int64_t* true_track_id,
                                             const double* true_photons, int32_t n_det, int32_t n_ticks,
                                             int32_t max_truth, const double* light_gain, const double* impulse_model,
                                             int32_t n_impulse, float* response, int64_t* response_true_track_id,
                                             double* response_true_photons) {
  LDSIM_ENTER(ctx);
  NEED(ctx && light_sample_inc && response && light_gain && n_det >= 0 && n_ticks >= 0 && max_truth >= 0, "bad argument");
  NEED(max_truth == 0 || (true_track_id && true_photons && response_true_track_id && response_true_photons),
       "truth arrays missing");
  const LdsimConsts& h = ctx->h_consts;
  NEED(h.light_tick_size > 0, "light constants not set");
  NEED(h.sipm_response_model == 0 || (h.sipm_response_model == 1 && impulse_model && n_impulse > 0 && h.impulse_tick_size > 0),
       "SIPM_RESPONSE_MODEL 1 needs IMPULSE_MODEL and IMPULSE_TICK_SIZE");
  const int64_t C = conv_ticks(h);
  NEED(C >= 0 && C < (1 << 24), "LIGHT_WINDOW / LIGHT_TICK_SIZE out of range");
  std::vector<double> w((size_t)C + 1);
  for (int64_t n = 0; n <= C; n++) w[(size_t)n] = sipm_response_model(n, impulse_model, n_impulse, h);
  return light_response_stage(ctx, true, light_sample_inc, true_track_id, true_photons, n_det, n_ticks, max_truth, w,
                              light_gain, response, response_true_track_id, response_true_photons);
}

// light_sim.calc_scintillation_effect -> calc_stat_fluctuations -> calc_light_detector_response on the resident photon sum
// (cli/simulate_pixels.py:1159-1180), everything staying in HBM
int light_launch_stat_fluct(ldsim_ctx* ctx, const float* inc, float* out, int64_t n);      // light_wvfm.hip

extern "C" int ldsim_dev_light_response(ldsim_ctx* ctx, const double* light_gain, const double* impulse_model,
                                        int32_t n_impulse, int32_t fluctuate) {
  LDSIM_ENTER(ctx);
  NEED(ctx && light_gain, "bad argument");
  NEED(ctx->light_sum_ndet > 0, "no resident photon sum (ldsim_dev_sum_light)");
  CK(light_join(ctx));
  const LdsimConsts& h = ctx->h_consts;
  NEED(h.light_tick_size > 0 && h.tau_s > 0 && h.tau_t > 0, "light constants not set");
  NEED(h.sipm_response_model == 0 || (h.sipm_response_model == 1 && impulse_model && n_impulse > 0 && h.impulse_tick_size > 0),
       "SIPM_RESPONSE_MODEL 1 needs IMPULSE_MODEL and IMPULSE_TICK_SIZE");
  const int64_t C = conv_ticks(h);
  NEED(C >= 0 && C < (1 << 24), "LIGHT_WINDOW / LIGHT_TICK_SIZE out of range");
  HIPCHK(hipSetDevice(ctx->device));
  const int D = ctx->light_sum_ndet, T = ctx->light_sum_nticks, Mt = ctx->light_sum_truth;
  const size_t bo = (size_t)D * T, bt = bo * (size_t)Mt;
  ctx->light_resp_valid = 0;
  std::vector<double> w0((size_t)C + 1), w1((size_t)C + 1);
  for (int64_t n = 0; n <= C; n++) {
    w0[(size_t)n] = scintillation_model(n, h);
    w1[(size_t)n] = sipm_response_model(n, impulse_model, n_impulse, h);
  }
  hipStream_t st = ctx->stream;
  CK(ldsim_ensure_buf(ctx, &ctx->light_w[0], w0.size() * 8));
  CK(ldsim_ensure_buf(ctx, &ctx->light_w[1], w1.size() * 8));
  CK(ldsim_ensure_buf(ctx, &ctx->light_gain, (size_t)D * 8));
  CK(ldsim_ensure_buf(ctx, &ctx->light_scint, bo * 4 + 16));
  CK(ldsim_ensure_buf(ctx, &ctx->light_disc, bo * 4 + 16));
  CK(ldsim_ensure_buf(ctx, &ctx->light_resp, bo * 4 + 16));
  HIPCHK(hipMemcpyAsync(ctx->light_w[0].p, w0.data(), w0.size() * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(ctx->light_w[1].p, w1.data(), w1.size() * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(ctx->light_gain.p, light_gain, (size_t)D * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemsetAsync(ctx->light_scint.p, 0, bo * 4, st));
  HIPCHK(hipMemsetAsync(ctx->light_resp.p, 0, bo * 4, st));
  if (Mt) {
    CK(ldsim_ensure_buf(ctx, &ctx->light_scint_tid, bt * 8 + 16));
    CK(ldsim_ensure_buf(ctx, &ctx->light_scint_tph, bt * 8 + 16));
    CK(ldsim_ensure_buf(ctx, &ctx->light_resp_tid, bt * 8 + 16));
    CK(ldsim_ensure_buf(ctx, &ctx->light_resp_tph, bt * 8 + 16));
    HIPCHK(hipMemsetAsync(ctx->light_scint_tid.p, 0xFF, bt * 8, st));
    HIPCHK(hipMemsetAsync(ctx->light_scint_tph.p, 0, bt * 8, st));
    HIPCHK(hipMemsetAsync(ctx->light_resp_tid.p, 0xFF, bt * 8, st));
    HIPCHK(hipMemsetAsync(ctx->light_resp_tph.p, 0, bt * 8, st));
  }
  if (bo == 0) {
    ctx->light_resp_valid = 1;
    return 0;
  }
  hipEvent_t ev[4];
  for (auto& e : ev) HIPCHK(hipEventCreate(&e));
  HIPCHK(hipEventRecord(ev[0], st));
  CK(light_response_launch(ctx, false, (const float*)ctx->light_out.p, (const int64_t*)ctx->light_tid.p,
                           (const double*)ctx->light_tph.p, D, T, Mt, (const double*)ctx->light_w[0].p, (int)C, nullptr,
                           (float*)ctx->light_scint.p, (int64_t*)ctx->light_scint_tid.p, (double*)ctx->light_scint_tph.p));
  HIPCHK(hipEventRecord(ev[1], st));
  const float* disc = (const float*)ctx->light_scint.p;
  if (fluctuate) {
    CK(light_launch_stat_fluct(ctx, (const float*)ctx->light_scint.p, (float*)ctx->light_disc.p, (int64_t)bo));
    disc = (const float*)ctx->light_disc.p;
  }
  HIPCHK(hipEventRecord(ev[2], st));
  CK(light_response_launch(ctx, true, disc, (const int64_t*)ctx->light_scint_tid.p, (const double*)ctx->light_scint_tph.p, D,
                           T, Mt, (const double*)ctx->light_w[1].p, (int)C, (const double*)ctx->light_gain.p,
                           (float*)ctx->light_resp.p, (int64_t*)ctx->light_resp_tid.p, (double*)ctx->light_resp_tph.p));
  HIPCHK(hipEventRecord(ev[3], st));
  HIPCHK(hipStreamSynchronize(st));
  for (int i = 0; i < 3; i++) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
    ctx->ms_light_resp[i] = ms;
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  ctx->light_resp_valid = 1;
  return 0;
}

extern "C" int ldsim_dev_light_response_download(ldsim_ctx* ctx, float* scint, float* disc, float* response,
                                                 int64_t* response_true_track_id, double* response_true_photons) {
  LDSIM_ENTER(ctx);
  NEED(ctx, "null ctx");
  NEED(ctx->light_resp_valid, "no resident detector response (ldsim_dev_light_response)");
  const size_t bo = (size_t)ctx->light_sum_ndet * ctx->light_sum_nticks, bt = bo * (size_t)ctx->light_sum_truth;
  if (bo == 0) return 0;
  hipStream_t st = ctx->stream;
  if (scint) HIPCHK(hipMemcpyAsync(scint, ctx->light_scint.p, bo * 4, hipMemcpyDeviceToHost, st));
  if (disc) HIPCHK(hipMemcpyAsync(disc, ctx->light_disc.p, bo * 4, hipMemcpyDeviceToHost, st));
  if (response) HIPCHK(hipMemcpyAsync(response, ctx->light_resp.p, bo * 4, hipMemcpyDeviceToHost, st));
  if (bt && response_true_track_id) HIPCHK(hipMemcpyAsync(response_true_track_id, ctx->light_resp_tid.p, bt * 8, hipMemcpyDeviceToHost, st));
  if (bt && response_true_photons) HIPCHK(hipMemcpyAsync(response_true_photons, ctx->light_resp_tph.p, bt * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}

extern "C" int ldsim_light_response_ms(ldsim_ctx* ctx, double* scint_ms, double* fluct_ms, double* response_ms) {
  LDSIM_ENTER(ctx);
  NEED(ctx, "null ctx");
  if (scint_ms) *scint_ms = ctx->ms_light_resp[0];
  if (fluct_ms) *fluct_ms = ctx->ms_light_resp[1];
  if (response_ms) *response_ms = ctx->ms_light_resp[2];
  return 0;
}

// ---- (2) chain -------------------------------------------------------------------------------------------------------
extern "C" int ldsim_charge_chain(ldsim_ctx* ctx, int64_t seg_begin, int64_t seg_end, int32_t want_fractions,
                                  LdsimChainStats* stats) {
  LDSIM_ENTER(ctx);
  NEED(ctx, "null ctx");
  NEED(ctx->d_resp, "no response table set (ldsim_set_response)");
  NEED_RESIDENT(ctx);
  NEED(seg_begin >= 0 && seg_end >= seg_begin && seg_end <= ctx->seg.n, "segment range outside the resident store");
  HIPCHK(hipSetDevice(ctx->device));
  int rc;
  try {
    rc = chain_run(ctx, seg_begin, seg_end, want_fractions);
  } catch (const std::exception& e) {      // host allocations of the orchestration: nothing may throw across the C ABI
    ldsim_set_error("chain_run: %s", e.what());
    rc = LDSIM_EINVAL;
  }
  if (stats) *stats = ctx->stats;
  return rc;
}

extern "C" int ldsim_chain_download(ldsim_ctx* ctx, int64_t capacity, int32_t* unique_pix, int32_t* batch,
                                    double* adc_list, double* adc_ticks, double* adc_digit, int64_t* tpm,
                                    double* fractions) {
  LDSIM_ENTER(ctx);
  NEED(ctx, "null ctx");
  const int64_t U = ctx->chain_U;
  if (capacity < U) {
    ldsim_set_error("capacity %lld < required %lld rows", (long long)capacity, (long long)U);
    return LDSIM_ENOSPC;
  }
  if (U == 0) return 0;
  const int A = ctx->h_consts.max_adc_values, M = ctx->h_consts.max_tracks_per_pixel;
  HIPCHK(hipSetDevice(ctx->device));
  if (unique_pix) HIPCHK(hipMemcpyAsync(unique_pix, ctx->scratch[SB_UPIX].p, U * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (batch) HIPCHK(hipMemcpyAsync(batch, ctx->scratch[SB_UBATCH].p, U * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (adc_list) HIPCHK(hipMemcpyAsync(adc_list, ctx->scratch[SB_ADC].p, (size_t)U * A * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (adc_ticks) HIPCHK(hipMemcpyAsync(adc_ticks, ctx->scratch[SB_TICKS].p, (size_t)U * A * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (adc_digit) HIPCHK(hipMemcpyAsync(adc_digit, ctx->scratch[SB_DIGIT].p, (size_t)U * A * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (tpm) HIPCHK(hipMemcpyAsync(tpm, ctx->scratch[SB_TPM].p, (size_t)U * M * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (fractions) {
    NEED(ctx->want_fractions, "fractions were not requested in the last ldsim_charge_chain call");
    CK(fractions_complete(ctx));
    HIPCHK(hipMemcpyAsync(fractions, ctx->scratch[SB_FRAC].p, (size_t)U * A * M * 8, hipMemcpyDeviceToHost, ctx->stream));
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

// The same copies on a second stream, returning at once: the next ldsim_charge_chain computes (into the other set of output
// buffers) while these rows cross PCIe.  Host buffers should be page-locked (ldsim_host_alloc) or the copy is not
// asynchronous; they hold the rows after ldsim_chain_download_wait.  One download is in flight at a time.
extern "C" int ldsim_chain_download_async(ldsim_ctx* ctx, int64_t capacity, int32_t* unique_pix, int32_t* batch,
                                          double* adc_list, double* adc_ticks, double* adc_digit, int64_t* tpm,
                                          double* fractions) {
  LDSIM_ENTER(ctx);
  NEED(ctx, "null ctx");
  const int64_t U = ctx->chain_U;
  if (capacity < U) {
    ldsim_set_error("capacity %lld < required %lld rows", (long long)capacity, (long long)U);
    return LDSIM_ENOSPC;
  }
  HIPCHK(hipSetDevice(ctx->device));
  if (!ctx->copy_stream) HIPCHK(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
  if (ctx->copy_pending) {
    HIPCHK(hipStreamSynchronize(ctx->copy_stream));
    ctx->copy_pending = 0;
  }
  ctx->async_out = 1;
  if (U == 0) return 0;
  if (fractions) NEED(ctx->want_fractions, "fractions were not requested in the last ldsim_charge_chain call");
  const int A = ctx->h_consts.max_adc_values, M = ctx->h_consts.max_tracks_per_pixel;
  hipStream_t cs = ctx->copy_stream;      // (ldsim_charge_chain returns with its stream drained: the rows are complete)
  if (unique_pix) HIPCHK(hipMemcpyAsync(unique_pix, ctx->scratch[SB_UPIX].p, U * 4, hipMemcpyDeviceToHost, cs));
  if (batch) HIPCHK(hipMemcpyAsync(batch, ctx->scratch[SB_UBATCH].p, U * 4, hipMemcpyDeviceToHost, cs));
  if (adc_list) HIPCHK(hipMemcpyAsync(adc_list, ctx->scratch[SB_ADC].p, (size_t)U * A * 8, hipMemcpyDeviceToHost, cs));
  if (adc_ticks) HIPCHK(hipMemcpyAsync(adc_ticks, ctx->scratch[SB_TICKS].p, (size_t)U * A * 8, hipMemcpyDeviceToHost, cs));
  if (adc_digit) HIPCHK(hipMemcpyAsync(adc_digit, ctx->scratch[SB_DIGIT].p, (size_t)U * A * 8, hipMemcpyDeviceToHost, cs));
  if (tpm) HIPCHK(hipMemcpyAsync(tpm, ctx->scratch[SB_TPM].p, (size_t)U * M * 8, hipMemcpyDeviceToHost, cs));
  if (fractions) {
    CK(fractions_complete(ctx));                        // (on the compute stream: drained before the copy stream reads the rows)
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpyAsync(fractions, ctx->scratch[SB_FRAC].p, (size_t)U * A * M * 8, hipMemcpyDeviceToHost, cs));
  }
  ctx->copy_pending = 1;
  ctx->pending_gen = ctx->out_gen;
  return 0;
}

extern "C" int ldsim_chain_download_wait(ldsim_ctx* ctx) {
  LDSIM_ENTER(ctx);
  NEED(ctx, "null ctx");
  if (ctx->copy_pending) {
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->copy_stream));
    ctx->copy_pending = 0;
  }
  return 0;
}

extern "C" int ldsim_chain_compact_hits(ldsim_ctx* ctx, void** dev_rows, int64_t* n_rows, int32_t* row_bytes) {
  LDSIM_ENTER(ctx);
  NEED(ctx && dev_rows && n_rows && row_bytes, "null argument");
  *dev_rows = ctx->scratch[SB_HITS].p;
  *n_rows = ctx->chain_hits;
  *row_bytes = 24;
  return 0;
}

extern "C" int ldsim_chain_kernel_ms(ldsim_ctx* ctx, double* current_ms, double* adc_ms, double* total_ms) {
  LDSIM_ENTER(ctx);
  NEED(ctx, "null ctx");
  if (current_ms) *current_ms = ctx->ms_current;
  if (adc_ms) *adc_ms = ctx->ms_adc;
  if (total_ms) *total_ms = ctx->ms_total;
  return 0;
}

extern "C" int ldsim_chain_kernel_ms_detail(ldsim_ctx* ctx, double* weights_ms, double* mac_ms, double* fallback_ms) {
  LDSIM_ENTER(ctx);
  NEED(ctx, "null ctx");
  if (weights_ms) *weights_ms = ctx->ms_weights;
  if (mac_ms) *mac_ms = ctx->ms_mac;
  if (fallback_ms) *fallback_ms = ctx->ms_fallback;
  return 0;
}
