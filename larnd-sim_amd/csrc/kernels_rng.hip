// kernels_rng.hip -- random streams of the noisy stages.
//
// The reference keeps one array of xoroshiro128p states (cli/simulate_pixels.py:396, create_xoroshiro128p_states(262144,
// seed)) and get_adc_values advances rng_states[ip] for the batch's ip-th unique pixel, a data-dependent number of normal
// draws per pixel (2 per tick, 3 more per trigger; fee.py:557,583-584,616-617,621,649).  A pixel's stream is strictly
// serial, so instead of letting one lane of a 256-thread workgroup draw 4000 numbers while the rest wait, the draws are
// made ahead of the trigger scan with ONE LANE PER PIXEL ROW (64 independent streams per wave):
//   fee_noise_kernel     row u draws its next `nd` normals into z[u][0..nd) (an upper bound of what the scan can consume)
//   pixel_adc_kernel     consumes z[u][cursor..] exactly where the reference draws; reports the count
//   rng_advance_kernel   advances state[u] by the draws actually consumed (2 words per normal), as the reference's in-place
//                        update of rng_states[ip] leaves it
// Row u of a chain launch uses state u of the table (for a launch holding one batch that is the reference's rng_states[ip]).
#include "ldsim_dev.h"
#include "rng.h"

__global__ void __launch_bounds__(256) fee_noise_kernel(const RngState* __restrict__ states, int64_t U, int nd,
                                                        float* __restrict__ z) {
  const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (u >= U) return;
  RngState st = states[u];
  float4* row = (float4*)(z + u * (int64_t)nd);      // nd is a multiple of 4 and the table is 16-byte aligned
  for (int k = 0; k < nd / 4; k++) {
    float4 v;
    v.x = rng_normal_f32(st);
    v.y = rng_normal_f32(st);
    v.z = rng_normal_f32(st);
    v.w = rng_normal_f32(st);
    row[k] = v;
  }
}

__global__ void __launch_bounds__(256) rng_advance_kernel(RngState* __restrict__ states, int64_t U,
                                                          const int32_t* __restrict__ n_draws) {
  const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (u >= U) return;
  RngState st = states[u];
  const int n = 2 * n_draws[u];
  for (int i = 0; i < n; i++) (void)rng_next(st);
  states[u] = st;
}

extern "C++" {
static void rng_jump_host(RngState& st) {
  static const uint64_t JUMP[2] = {0xbeac0467eba5facbULL, 0xd86b048b86aa9922ULL};
  uint64_t s0 = 0, s1 = 0;
  for (int i = 0; i < 2; i++)
    for (int b = 0; b < 64; b++) {
      if (JUMP[i] & (1ULL << b)) { s0 ^= st.s0; s1 ^= st.s1; }
      (void)rng_next(st);
    }
  st.s0 = s0;
  st.s1 = s1;
}

static uint64_t splitmix_first(uint64_t seed) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

// Append states so that the table holds exactly `n`: a fresh create_xoroshiro128p_states(n - rng_n, seed) chain when
// `fresh_seed` is given (maybe_create_rng_states, cli/simulate_pixels.py:92-104), else the continuation of the last chain
// (state i = state i-1 jumped 2^64 steps).  States [0, rng_n) keep whatever the kernels advanced them to.  The allocation
// grows geometrically, the logical length ctx->rng_n is exact.
static int rng_append(ldsim_ctx* ctx, int64_t n, const uint64_t* fresh_seed) {
  if (n <= ctx->rng_n) return 0;
  std::vector<RngState> fresh((size_t)(n - ctx->rng_n));
  RngState cur;
  cur.s0 = ctx->rng_last_init[0];
  cur.s1 = ctx->rng_last_init[1];
  for (size_t i = 0; i < fresh.size(); i++) {
    if (i == 0 && (fresh_seed || ctx->rng_n == 0)) cur.s0 = cur.s1 = splitmix_first(fresh_seed ? *fresh_seed : ctx->rng_seed);
    else rng_jump_host(cur);
    fresh[i] = cur;
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if ((size_t)n * sizeof(RngState) > ctx->d_rng.bytes) {
    const int64_t cap = n + n / 4 + 1024;
    DevBuf nb;
    int rc = ldsim_ensure_buf(ctx, &nb, (size_t)cap * sizeof(RngState));
    if (rc) return rc;
    if (ctx->rng_n) HIPCHK(hipMemcpy(nb.p, ctx->d_rng.p, (size_t)ctx->rng_n * sizeof(RngState), hipMemcpyDeviceToDevice));
    if (ctx->d_rng.p) (void)hipFree(ctx->d_rng.p);
    ctx->d_rng = nb;
  }
  HIPCHK(hipMemcpy((char*)ctx->d_rng.p + (size_t)ctx->rng_n * sizeof(RngState), fresh.data(), fresh.size() * sizeof(RngState),
                   hipMemcpyHostToDevice));
  ctx->rng_n = n;
  ctx->rng_last_init[0] = cur.s0;
  ctx->rng_last_init[1] = cur.s1;
  return 0;
}

int rng_ensure_states(ldsim_ctx* ctx, int64_t n) {
  if (!ctx->rng_seeded) {
    ldsim_set_error("this stage draws random numbers but no random state exists: call ldsim_rng_seed first");
    return LDSIM_ESTATE;
  }
  return rng_append(ctx, n, nullptr);
}

// upper bound of the normals one pixel's scan can consume (fee.py:557-655): the first reset draw, 2 per loop pass -- every
// tick, plus one busy period past the end of the window for every trigger (a noise trigger in the pass where adc_busy reaches 0
// starts another one, up to MAX_ADC_VALUES of them) --, 3 per trigger; rounded up to a multiple of 4
int rng_fee_draws_per_pixel(const LdsimConsts& h, int NT) {
  const int busy = (int)llround(h.adc_busy_delay * h.clock_cycle / h.time_sampling);
  const int nd = 1 + 2 * (NT + h.max_adc_values * (busy + 1)) + 3 * h.max_adc_values;
  return (nd + 3) & ~3;
}

int rng_launch_fee_noise(ldsim_ctx* ctx, int64_t U, int nd, float* z) {
  if (U == 0) return 0;
  hipLaunchKernelGGL(fee_noise_kernel, dim3((unsigned)((U + 255) / 256)), dim3(256), 0, ctx->stream,
                     (const RngState*)ctx->d_rng.p, U, nd, z);
  HIPCHK(hipGetLastError());
  return 0;
}
int rng_launch_advance(ldsim_ctx* ctx, int64_t U, const int32_t* n_draws) {
  if (U == 0) return 0;
  hipLaunchKernelGGL(rng_advance_kernel, dim3((unsigned)((U + 255) / 256)), dim3(256), 0, ctx->stream,
                     (RngState*)ctx->d_rng.p, U, n_draws);
  HIPCHK(hipGetLastError());
  return 0;
}
}

// numba.cuda.random.create_xoroshiro128p_states(n, seed) (cli/simulate_pixels.py:92-104,396)
extern "C" int ldsim_rng_seed(ldsim_ctx* ctx, uint64_t seed, int64_t n_states) {
  LDSIM_ENTER(ctx);
  if (!ctx || n_states < 0) {
    ldsim_set_error("bad argument");
    return LDSIM_EINVAL;
  }
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (ctx->d_rng.p) (void)hipFree(ctx->d_rng.p);
  ctx->d_rng = DevBuf{};
  ctx->rng_n = 0;
  ctx->rng_seed = seed;
  ctx->rng_seeded = 1;
  ctx->light_noise_calls = 0;
  ctx->rng_last_init[0] = ctx->rng_last_init[1] = 0;
  try {
    return n_states ? rng_ensure_states(ctx, n_states) : 0;
  } catch (const std::exception& e) {        // (the host copy of the fresh states: nothing throws across the C ABI)
    ldsim_set_error("ldsim_rng_seed: %s", e.what());
    return LDSIM_EINVAL;
  }
}

// maybe_create_rng_states(n, seed, rng_states) (cli/simulate_pixels.py:92-104): no table yet -> create n states from `seed`;
// a shorter table -> append create_xoroshiro128p_states(n - len, seed); long enough -> untouched
extern "C" int ldsim_rng_extend(ldsim_ctx* ctx, int64_t n_states, uint64_t seed) {
  LDSIM_ENTER(ctx);
  if (!ctx || n_states < 0) {
    ldsim_set_error("bad argument");
    return LDSIM_EINVAL;
  }
  if (!ctx->rng_seeded) return ldsim_rng_seed(ctx, seed, n_states);
  HIPCHK(hipSetDevice(ctx->device));
  try {
    return rng_append(ctx, n_states, &seed);
  } catch (const std::exception& e) {
    ldsim_set_error("ldsim_rng_extend: %s", e.what());
    return LDSIM_EINVAL;
  }
}

extern "C" int64_t ldsim_rng_count(ldsim_ctx* ctx) { return ctx && ctx->rng_seeded ? ctx->rng_n : -1; }

// forget the table: noisy calls are refused again until the next ldsim_rng_seed
extern "C" int ldsim_rng_clear(ldsim_ctx* ctx) {
  LDSIM_ENTER(ctx);
  if (!ctx) return 0;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (ctx->d_rng.p) (void)hipFree(ctx->d_rng.p);
  ctx->d_rng = DevBuf{};
  ctx->rng_n = 0;
  ctx->rng_seeded = 0;
  return 0;
}

// states [0, n) as they stand (s0, s1 pairs), for tests and for a driver that wants to checkpoint them
extern "C" int ldsim_rng_states_download(ldsim_ctx* ctx, uint64_t* states, int64_t n) {
  LDSIM_ENTER(ctx);
  if (!ctx || !states || n < 0 || n > ctx->rng_n) {
    ldsim_set_error("bad argument (the table holds %lld states)", ctx ? (long long)ctx->rng_n : 0LL);
    return LDSIM_EINVAL;
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (n) HIPCHK(hipMemcpy(states, ctx->d_rng.p, (size_t)n * sizeof(RngState), hipMemcpyDeviceToHost));
  return 0;
}
