// rng.h -- numba.cuda.random's xoroshiro128p generator and float32 Box-Muller normal, as Numba documents them
// (third-party: module `numba`, not under the reference tree; restated, see oracle/ldsim_oracle.c for the statement of
// what is and is not pinned).  Call sites in the reference: fee.py:557,583-584,616-617,621,649; detsim.py:331,336-337.
#pragma once
#include <stdint.h>

struct RngState {
  uint64_t s0, s1;
};

__host__ __device__ inline uint64_t rng_rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
__host__ __device__ inline uint64_t rng_next(RngState& st) {
  uint64_t s0 = st.s0, s1 = st.s1;
  const uint64_t result = s0 + s1;
  s1 ^= s0;
  st.s0 = rng_rotl(s0, 55) ^ s1 ^ (s1 << 14);
  st.s1 = rng_rotl(s1, 36);
  return result;
}
// uint64_to_unit_float32: float32((x >> 11) * 2^-53)  (may round up to 1.0f, like Numba's)
__host__ __device__ inline float rng_uniform_f32(RngState& st) {
  return (float)((double)(rng_next(st) >> 11) * (1.0 / 9007199254740992.0));
}
// xoroshiro128p_normal_float32: z0 of Box-Muller in float32; the second value is discarded
__device__ inline float rng_normal_f32(RngState& st) {
  const float u1 = rng_uniform_f32(st), u2 = rng_uniform_f32(st);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}
