// f64_pipes.hip -- can the FP64 matrix pipe (v_mfma_f64_16x16x4_f64) and the FP64 vector pipe (v_fma_f64) of one SIMD run
// side by side?  Three kernels on 256-thread workgroups, 4 per CU: all waves DFMA; all waves MFMA; waves 0-1 MFMA and waves
// 2-3 DFMA.  Prints TFLOP/s of each.  hipcc --offload-arch=gfx950 -O3 tools/micro/f64_pipes.hip -o /tmp/f64_pipes
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double run_valu(double a, double b, int iters) {
  double acc[16];
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = a + i;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc[i] = fma(acc[i], a, b);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += acc[i];
  return s;
}
__device__ __forceinline__ double run_mfma(double a, double b, int iters) {
  d4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; i++) acc[i] = (d4){a, b, a, b};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  return s;
}
// mode 0: all VALU (64 DFMA per iter per wave = 64*64 FMA); 1: all MFMA (4 MFMA per iter = 4096 FMA); 2: waves 0,1 MFMA, 2,3 VALU
__global__ void __launch_bounds__(256, 4) k(int mode, int iters, double a, double b, double* out) {
  const int wv = threadIdx.x >> 6;
  double s;
  if (mode == 0) s = run_valu(a, b, iters);
  else if (mode == 1) s = run_mfma(a, b, iters);
  else if (mode == 3) s = (wv < 2) ? run_mfma(a, b, iters) : 0.0;     // half the waves MFMA, the others idle
  else if (mode == 4) s = (wv < 2) ? 0.0 : run_valu(a, b, iters);     // half the waves VALU, the others idle
  else s = (wv < 2) ? run_mfma(a, b, iters) : run_valu(a, b, iters);
  if (s == 12345.678) out[0] = s;
}
int main() {
  double* d;
  hipMalloc(&d, 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, blocks = 256 * 4;
  const char* names[] = {"all waves DFMA", "all waves MFMA f64 16x16x4", "2 waves MFMA + 2 waves DFMA", "2 waves MFMA only", "2 waves DFMA only"};
  for (int rep = 0; rep < 2; rep++)
    for (int mode = 0; mode < 5; mode++) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, mode, iters, 1.0000001, 1e-9, d);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double fma_per_wave_iter = 4096.0;   // 64 DFMA x 64 lanes == 4 MFMA x 1024
      double waves = (mode == 3 || mode == 4) ? 2 : 4;
      double flops = 2.0 * fma_per_wave_iter * iters * waves * blocks;
      if (rep) printf("%-32s %8.3f ms  %7.2f TFLOP/s\n", names[mode], ms, flops / (ms * 1e-3) / 1e12);
    }
  return 0;
}
