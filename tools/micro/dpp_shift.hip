// micro-test: whole-wave shift by one lane with DPP (wave_shl:1, GFX9 only) on gfx950, 64-bit payload, "old" kept in lane 63
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ inline double wave_shl1(double old, double src) {
  int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), 0x130, 0xF, 0xF, false);
  int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), 0x130, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__global__ void k(double* out) {
  const int lane = threadIdx.x;
  double v = 100.0 + lane, old = -1.0 - lane;
  out[lane] = wave_shl1(old, v);
}
int main() {
  double* d; double h[64];
  hipMalloc(&d, 64 * 8);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, 64 * 8, hipMemcpyDeviceToHost);
  for (int i = 0; i < 64; i++) printf("%g ", h[i]);
  printf("\n");
  return 0;
}
