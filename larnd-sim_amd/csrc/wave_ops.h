// wave_ops.h -- wave-64 scans and reductions with DPP (row shifts inside the 16-lane rows, then row_bcast:15 / :31 across
// the rows: GFX9 controls that gfx950 keeps) instead of shuffles through the LDS crossbar (ds_bpermute), whose ~100-cycle
// latency a serial phase pays at every step.  A lane without a source keeps `ident` (bound_ctrl off, old = ident).
#pragma once
#include <hip/hip_runtime.h>

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int wdpp_i32(int ident, int v) {
  return __builtin_amdgcn_update_dpp(ident, v, CTRL, ROW_MASK, 0xF, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double wdpp_f64(double ident, double v) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(ident), __double2loint(v), CTRL, ROW_MASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(ident), __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
  return __hiloint2double(hi, lo);
}

// inclusive scan over the 64 lanes: lane l ends with op(v[0], .., v[l]); Op is associative, `ident` its identity
template <class Op>
__device__ __forceinline__ int wave_scan_i32(int v, int ident, Op op) {
  v = op(v, wdpp_i32<0x111, 0xF>(ident, v));
  v = op(v, wdpp_i32<0x112, 0xF>(ident, v));
  v = op(v, wdpp_i32<0x114, 0xF>(ident, v));
  v = op(v, wdpp_i32<0x118, 0xF>(ident, v));
  v = op(v, wdpp_i32<0x142, 0xA>(ident, v));
  v = op(v, wdpp_i32<0x143, 0xC>(ident, v));
  return v;
}
template <class Op>
__device__ __forceinline__ double wave_scan_f64(double v, double ident, Op op) {
  v = op(v, wdpp_f64<0x111, 0xF>(ident, v));
  v = op(v, wdpp_f64<0x112, 0xF>(ident, v));
  v = op(v, wdpp_f64<0x114, 0xF>(ident, v));
  v = op(v, wdpp_f64<0x118, 0xF>(ident, v));
  v = op(v, wdpp_f64<0x142, 0xA>(ident, v));
  v = op(v, wdpp_f64<0x143, 0xC>(ident, v));
  return v;
}
__device__ __forceinline__ int wave_lane_i32(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ double wave_lane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// reductions: the result in every lane (wave-uniform)
__device__ __forceinline__ int wave_min_i32(int v) {
  return wave_lane_i32(wave_scan_i32(v, 0x7fffffff, [](int a, int b) { return a < b ? a : b; }), 63);
}
__device__ __forceinline__ int wave_max_i32(int v) {
  return wave_lane_i32(wave_scan_i32(v, (int)0x80000000, [](int a, int b) { return a > b ? a : b; }), 63);
}
__device__ __forceinline__ double wave_min_f64(double v) {
  return wave_lane_f64(wave_scan_f64(v, 1e300, [](double a, double b) { return fmin(a, b); }), 63);
}
__device__ __forceinline__ double wave_max_f64(double v) {
  return wave_lane_f64(wave_scan_f64(v, -1e300, [](double a, double b) { return fmax(a, b); }), 63);
}
__device__ __forceinline__ double wave_add_f64(double v) {
  return wave_lane_f64(wave_scan_f64(v, 0.0, [](double a, double b) { return a + b; }), 63);
}
