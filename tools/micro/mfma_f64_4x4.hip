// mfma_f64_4x4.hip -- v_mfma_f64_4x4x4_4b_f64 on gfx950: (1) its rate beside v_mfma_f64_16x16x4_f64 (256 vs 1024 FMAs per
// instruction: is the small shape 4 passes, i.e. the same FMAs per cycle on this part?), (2) which lane holds which element
// of A, B and D.  Why: the node-separable correlation pads the quadrature nodes of a pair to 16 MFMA rows (25 rows issued for
// 15 nodes on average); four-row blocks would issue 17.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_f64_4x4.hip -o tools/micro/bin/mfma_f64_4x4
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256, 2) rate(int iters, double a, double b, double* out) {
  double s = 0;
  if (MODE == 0) {
    d4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; i++) acc[i] = (d4){a, b, a, b};
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    double acc[16];
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = a + i;
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; i++) s += acc[i];
  }
  if (s == 12345.678) out[0] = s;
}

__global__ void layout(const double* A, const double* B, double* D) {
  const int l = threadIdx.x;
  double acc = 0;
  acc = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], acc, 0, 0, 0);      // raw: lane l's A, B element in, lane l's D out
  D[l] = acc;
}

int main() {
  double* d;
  hipMalloc(&d, 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, blocks = 256 * 2;
  for (int rep = 0; rep < 2; rep++)
    for (int mode = 0; mode < 2; mode++) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(blocks), dim3(256), 0, 0, iters, 1.0000001, 1e-9, d);
      else hipLaunchKernelGGL(rate<1>, dim3(blocks), dim3(256), 0, 0, iters, 1.0000001, 1e-9, d);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double fma = (mode == 0 ? 4 * 1024.0 : 16 * 256.0) * iters * 4 * blocks;
      if (rep) printf("%-28s %8.3f ms  %7.2f TFLOP/s  (%.1f cycles per instruction per SIMD at 2.4 GHz, 2 waves per SIMD)\n",
                      mode == 0 ? "v_mfma_f64_16x16x4" : "v_mfma_f64_4x4x4 (4 blocks)", ms, 2 * fma / (ms * 1e-3) / 1e12,
                      ms * 1e-3 * 2.4e9 / (iters * (mode == 0 ? 4.0 : 16.0) * 2));
    }
  // layout: one-hot probes.  A one-hot in lane la, B one-hot in lane lb: which lanes of D light up?
  double hA[64], hB[64], hD[64], *dA, *dB, *dD;
  hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 512);
  printf("pairs (lane of A, lane of B) -> lanes of D that receive A*B:\n");
  for (int la = 0; la < 64; la++) {
    int first = 1;
    for (int lb = 0; lb < 64; lb++) {
      for (int i = 0; i < 64; i++) { hA[i] = i == la ? 3.0 : 0.0; hB[i] = i == lb ? 5.0 : 0.0; }
      hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
      hipMemcpy(hD, dD, 512, hipMemcpyDeviceToHost);
      for (int i = 0; i < 64; i++)
        if (hD[i] != 0.0) {
          if (first) printf("A lane %2d:", la);
          first = 0;
          printf("  B%2d->D%2d", lb, i);
        }
    }
    if (!first) printf("\n");
  }
  return 0;
}
