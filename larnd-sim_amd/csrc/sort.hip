// sort.hip -- device-wide sort / scan steps of the chain (rocPRIM), and the small glue kernels around them.
// a7 (cli/simulate_pixels.py:952-957,1019-1026): unique pixel set + index map, here as a stable radix sort
// of (batch, pixel, ring-code) keys whose value is the pair's position in segment order.
#include <string.h>
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "ldsim_dev.h"
#include "wave_ops.h"

// key = batch(24) | pixel(32) | ringcode(4, 15 = invalid) ; invalid pairs get ~0 and sort last
// (n_entries < 2^31, checked by the chain: 32-bit index arithmetic; the valid count goes to its counter once per workgroup -- one
// atomic per wave on the same address had made this kernel 0.34 ms per 100 k segments)
__global__ void __launch_bounds__(256) make_keys_kernel(const int32_t* __restrict__ neigh, const int32_t* __restrict__ nrad,
                                 const int32_t* __restrict__ batch, int64_t seg_begin, int32_t batch0, int P,
                                 int64_t n_entries, unsigned long long* __restrict__ keys, int32_t* __restrict__ vals,
                                 unsigned long long* __restrict__ counters) {
  __shared__ int s_cnt[4];
  int valid = 0;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)n_entries; i += gridDim.x * 256u) {
    int32_t pix = neigh[i];
    const unsigned r = i / (unsigned)P;
    int32_t b = batch[seg_begin + r];
    unsigned long long key = ~0ull;
    if (pix >= 0 && b >= 0) {
      int32_t d = nrad[i];
      unsigned long long dn = (d < 0 || d > 14) ? 15ull : (unsigned long long)d;
      key = ((unsigned long long)(uint32_t)(b - batch0) << 36) | ((unsigned long long)(uint32_t)pix << 4) | dn;
      valid++;
    }
    keys[i] = key;
    vals[i] = (int32_t)i;
  }
  const int wsum = wave_lane_i32(wave_scan_i32(valid, 0, [](int a, int b) { return a + b; }), 63);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = wsum;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int c = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    if (c) atomicAdd(&counters[4], (unsigned long long)c);
  }
}

__global__ void heads_kernel(const unsigned long long* __restrict__ keys, int64_t n_valid, int32_t* __restrict__ heads) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_valid) return;
  heads[i] = (i == 0 || (keys[i] >> 4) != (keys[i - 1] >> 4)) ? 1 : 0;
}

__global__ void fill_unique_kernel(const unsigned long long* __restrict__ keys, const int32_t* __restrict__ heads,
                                   const int32_t* __restrict__ uidx, int64_t n_valid, int32_t batch0,
                                   int32_t* __restrict__ upix, int32_t* __restrict__ ubatch, int64_t* __restrict__ uoff,
                                   int64_t U) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) uoff[U] = n_valid;
  if (i >= n_valid || !heads[i]) return;
  int32_t u = uidx[i];
  upix[u] = (int32_t)((keys[i] >> 4) & 0xFFFFFFFFull);
  ubatch[u] = (int32_t)(keys[i] >> 36) + batch0;
  uoff[u] = i;
}

// first relative segment index of every batch in [seg_begin, seg_end)
__global__ void batch_first_kernel(const int32_t* __restrict__ batch, int64_t seg_begin, int64_t n, int32_t batch0,
                                   int32_t* __restrict__ first) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  int32_t b = batch[seg_begin + r];
  if (b < 0) return;
  if (r == 0 || batch[seg_begin + r - 1] != b) first[b - batch0] = (int32_t)r;
}

__global__ void tmax_batch_kernel(SegStore s, const LdsimConsts* __restrict__ c, int64_t begin, int64_t n, int32_t batch0,
                                  double* __restrict__ starts, int32_t* __restrict__ tmax_b,
                                  unsigned long long* __restrict__ tran_b) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  int32_t b = -1, len_i = 0;
  double td = 0;
  if (r < n) {
    const int64_t i = begin + r;
    double t_end = py_round((s.f[LDSIM_T_END][i] + 1) / c->time_sampling) * c->time_sampling;
    double t_start = py_round((s.f[LDSIM_T_START][i] - c->time_padding) / c->time_sampling) * c->time_sampling;
    starts[r] = t_start;
    b = s.batch[i];
    if (b >= 0) {
      double len = ceil((t_end - t_start) / c->time_sampling);
      if (len > 0 && len < 2.0e9) len_i = (int32_t)len;
      td = s.f[LDSIM_TRAN_DIFF][i];
      if (!(td > 0)) td = 0;
    }
  }
  // segments are sorted by batch, so nearly every wave holds one batch: one atomic per wave instead of one per segment on
  // the few per-batch cells (50 k contended atomics cost 0.7 ms per launch)
  const unsigned long long act = __ballot(b >= 0);
  if (act == 0) return;
  const int first = __ffsll((long long)act) - 1;
  const int bref = __builtin_amdgcn_readlane(b, first);
  if (__ballot(b >= 0 && b != bref) == 0) {
    const int m = wave_max_i32(len_i);
    const double tm = wave_max_f64(td);
    if (lane == first) {
      if (m > 0) atomicMax(&tmax_b[bref - batch0], m);
      if (tran_b && tm > 0) atomicMax(&tran_b[bref - batch0], (unsigned long long)__double_as_longlong(tm));
    }
    return;
  }
  if (b < 0) return;
  if (len_i > 0) atomicMax(&tmax_b[b - batch0], len_i);
  if (tran_b && td > 0) atomicMax(&tran_b[b - batch0], (unsigned long long)__double_as_longlong(td));
}

// compact hit rows for the multi-GPU all-gather: {batch i32, pixel i32, adc i32, tick f64-bits hi/lo}
__global__ void compact_hits_kernel(const int32_t* __restrict__ upix, const int32_t* __restrict__ ubatch,
                                    const int32_t* __restrict__ hit_count, const int32_t* __restrict__ hit_off,
                                    const double* __restrict__ digit, const double* __restrict__ ticks, int A, int64_t U,
                                    int32_t* __restrict__ rows) {
  int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= U) return;
  int n = hit_count[u];
  int64_t o = hit_off[u];
  for (int h = 0; h < n; h++) {
    int32_t* r = rows + (o + h) * 6;
    r[0] = ubatch[u];
    r[1] = upix[u];
    r[2] = (int32_t)digit[u * A + h];
    r[3] = h;
    double t = ticks[u * A + h];
    *(double*)(r + 4) = t;
  }
}

extern "C++" {
static inline int nblk(int64_t n, int b) { return (int)((n + b - 1) / b); }

int sort_make_keys(ldsim_ctx* ctx, const int32_t* neigh, const int32_t* nrad, int64_t seg_begin, int32_t batch0, int P,
                   int64_t n_entries, unsigned long long* keys, int32_t* vals, unsigned long long* counters) {
  if (n_entries == 0) return 0;
  const int64_t nb = nblk(n_entries, 256);
  hipLaunchKernelGGL(make_keys_kernel, dim3((unsigned)(nb < 8192 ? nb : 8192)), dim3(256), 0, ctx->stream, neigh, nrad,
                     ctx->seg.batch, seg_begin, batch0, P, n_entries, keys, vals, counters);
  HIPCHK(hipGetLastError());
  return 0;
}

int sort_pairs(ldsim_ctx* ctx, unsigned long long* keys_in, unsigned long long* keys_out, int32_t* vals_in,
               int32_t* vals_out, int64_t n) {
  if (n == 0) return 0;
  size_t tmp = 0;
  HIPCHK(rocprim::radix_sort_pairs(nullptr, tmp, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0, 64, ctx->stream));
  int rc = ldsim_ensure(ctx, SB_SORTTMP, tmp);
  if (rc) return rc;
  HIPCHK(rocprim::radix_sort_pairs(ctx->scratch[SB_SORTTMP].p, tmp, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0,
                                   64, ctx->stream));
  return 0;
}

// The valid entries of make_keys_kernel's arrays (key != ~0) in their order of appearance: keys -> keys_out, their indices ->
// vals_out, their number -> *d_count (a device word).  Two stable rocPRIM selects; the pair-list sort then runs over the valid
// entries only (a quarter of the (segment, neighbour slot) entries of the module0 bench) and equal keys keep the entry order they
// had among all entries.
namespace {
struct KeyValid {
  __device__ bool operator()(const unsigned long long& k) const { return k != ~0ull; }
};
struct KeyFlag {
  __device__ unsigned char operator()(const unsigned long long& k) const { return k != ~0ull ? 1 : 0; }
};
}  // namespace
int sort_compact_valid(ldsim_ctx* ctx, const unsigned long long* keys_in, int64_t n, unsigned long long* keys_out, int32_t* vals_out,
                       unsigned int* d_count) {
  if (n == 0) return 0;
  auto flags = rocprim::make_transform_iterator(keys_in, KeyFlag());
  rocprim::counting_iterator<int32_t> idx(0);
  size_t t1 = 0, t2 = 0;
  HIPCHK(rocprim::select(nullptr, t1, keys_in, keys_out, d_count, (size_t)n, KeyValid(), ctx->stream));
  HIPCHK(rocprim::select(nullptr, t2, idx, flags, vals_out, d_count, (size_t)n, ctx->stream));
  int rc = ldsim_ensure(ctx, SB_SORTTMP, t1 > t2 ? t1 : t2);
  if (rc) return rc;
  t1 = t2 = ctx->scratch[SB_SORTTMP].bytes;
  HIPCHK(rocprim::select(ctx->scratch[SB_SORTTMP].p, t1, keys_in, keys_out, d_count, (size_t)n, KeyValid(), ctx->stream));
  HIPCHK(rocprim::select(ctx->scratch[SB_SORTTMP].p, t2, idx, flags, vals_out, d_count, (size_t)n, ctx->stream));
  return 0;
}

// the same over the key bits [begin_bit, end_bit) only (a pass per 8 bits)
int sort_pairs_bits(ldsim_ctx* ctx, unsigned long long* keys_in, unsigned long long* keys_out, int32_t* vals_in, int32_t* vals_out,
                    int64_t n, int begin_bit, int end_bit) {
  if (n == 0) return 0;
  if (end_bit > 64) end_bit = 64;
  size_t tmp = 0;
  HIPCHK(rocprim::radix_sort_pairs(nullptr, tmp, keys_in, keys_out, vals_in, vals_out, (size_t)n, begin_bit, end_bit, ctx->stream));
  int rc = ldsim_ensure(ctx, SB_SORTTMP, tmp);
  if (rc) return rc;
  HIPCHK(rocprim::radix_sort_pairs(ctx->scratch[SB_SORTTMP].p, tmp, keys_in, keys_out, vals_in, vals_out, (size_t)n, begin_bit,
                                   end_bit, ctx->stream));
  return 0;
}

// stable sort of 32-bit keys (their low `bits` bits) with 64-bit payloads: the photon-sum records (kernels_light.hip)
int sort_pairs_u32_u64(ldsim_ctx* ctx, unsigned* keys_in, unsigned* keys_out, unsigned long long* vals_in, unsigned long long* vals_out,
                       int64_t n, int bits) {
  if (n == 0) return 0;
  if (bits > 32) bits = 32;
  size_t tmp = 0;
  HIPCHK(rocprim::radix_sort_pairs(nullptr, tmp, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0, bits, ctx->stream));
  int rc = ldsim_ensure(ctx, SB_SORTTMP, tmp);
  if (rc) return rc;
  HIPCHK(rocprim::radix_sort_pairs(ctx->scratch[SB_SORTTMP].p, tmp, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0, bits,
                                   ctx->stream));
  return 0;
}

int sort_exclusive_scan_i32(ldsim_ctx* ctx, const int32_t* in, int32_t* out, int64_t n) {
  if (n == 0) return 0;
  size_t tmp = 0;
  HIPCHK(rocprim::exclusive_scan(nullptr, tmp, in, out, (int32_t)0, (size_t)n, rocprim::plus<int32_t>(), ctx->stream));
  int rc = ldsim_ensure(ctx, SB_SORTTMP, tmp);
  if (rc) return rc;
  HIPCHK(rocprim::exclusive_scan(ctx->scratch[SB_SORTTMP].p, tmp, in, out, (int32_t)0, (size_t)n,
                                 rocprim::plus<int32_t>(), ctx->stream));
  return 0;
}

int sort_exclusive_scan_u64(ldsim_ctx* ctx, const unsigned long long* in, unsigned long long* out, int64_t n) {
  if (n == 0) return 0;
  size_t tmp = 0;
  HIPCHK(rocprim::exclusive_scan(nullptr, tmp, in, out, 0ull, (size_t)n, rocprim::plus<unsigned long long>(), ctx->stream));
  int rc = ldsim_ensure(ctx, SB_SORTTMP, tmp);
  if (rc) return rc;
  HIPCHK(rocprim::exclusive_scan(ctx->scratch[SB_SORTTMP].p, tmp, in, out, 0ull, (size_t)n,
                                 rocprim::plus<unsigned long long>(), ctx->stream));
  return 0;
}

int sort_heads(ldsim_ctx* ctx, const unsigned long long* keys, int64_t n_valid, int32_t* heads) {
  if (n_valid == 0) return 0;
  hipLaunchKernelGGL(heads_kernel, dim3(nblk(n_valid, 256)), dim3(256), 0, ctx->stream, keys, n_valid, heads);
  HIPCHK(hipGetLastError());
  return 0;
}

int sort_fill_unique(ldsim_ctx* ctx, const unsigned long long* keys, const int32_t* heads, const int32_t* uidx,
                     int64_t n_valid, int32_t batch0, int32_t* upix, int32_t* ubatch, int64_t* uoff, int64_t U) {
  hipLaunchKernelGGL(fill_unique_kernel, dim3(nblk(n_valid > 0 ? n_valid : 1, 256)), dim3(256), 0, ctx->stream, keys,
                     heads, uidx, n_valid, batch0, upix, ubatch, uoff, U);
  HIPCHK(hipGetLastError());
  return 0;
}

int sort_batch_first(ldsim_ctx* ctx, int64_t seg_begin, int64_t n, int32_t batch0, int32_t* first) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(batch_first_kernel, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, ctx->seg.batch, seg_begin, n,
                     batch0, first);
  HIPCHK(hipGetLastError());
  return 0;
}

int sort_tmax_batch(ldsim_ctx* ctx, int64_t seg_begin, int64_t n, int32_t batch0, double* starts, int32_t* tmax_b,
                    unsigned long long* tran_b) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(tmax_batch_kernel, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, ctx->seg, ctx->d_consts,
                     seg_begin, n, batch0, starts, tmax_b, tran_b);
  HIPCHK(hipGetLastError());
  return 0;
}

int sort_compact_hits(ldsim_ctx* ctx, const int32_t* upix, const int32_t* ubatch, const int32_t* hit_count,
                      const int32_t* hit_off, const double* digit, const double* ticks, int A, int64_t U, int32_t* rows) {
  if (U == 0) return 0;
  hipLaunchKernelGGL(compact_hits_kernel, dim3(nblk(U, 256)), dim3(256), 0, ctx->stream, upix, ubatch, hit_count,
                     hit_off, digit, ticks, A, U, rows);
  HIPCHK(hipGetLastError());
  return 0;
}
}
