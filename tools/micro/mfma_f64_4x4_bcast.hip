// mfma_f64_4x4_bcast.hip -- does v_mfma_f64_4x4x4_4b_f64 honour CBSZ / ABID on gfx950?  With cbsz = 2 all four 4x4x4 blocks should
// take their A operand from block `abid`: then the A register of the 16-node product (lane 16 k + node) feeds a 4-node x 16-tick
// product per node block without any re-load -- the node-separable correlation could issue ceil(nodes / 4) x 17 cycles per cell
// group instead of 64.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_f64_4x4_bcast.hip -o tools/micro/bin/mfma_f64_4x4_bcast
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>

template <int ABID>
__global__ void probe(const double* A, const double* B, double* D) {
  const int l = threadIdx.x;
  D[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 2, ABID, 0);
}

int main() {
  double hA[64], hB[64], hD[64], *dA, *dB, *dD;
  hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 512);
  for (int i = 0; i < 64; i++) { hA[i] = 1.0 + 0.37 * i + 0.01 * i * i; hB[i] = 2.0 - 0.11 * i + 0.003 * i * i; }
  hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
  for (int abid = 0; abid < 4; abid++) {
    if (abid == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    if (abid == 1) hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    if (abid == 2) hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    if (abid == 3) hipLaunchKernelGGL(probe<3>, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 512, hipMemcpyDeviceToHost);
    double worst_b = 0, worst_n = 0;      // against "A of block abid for every block" and against "no broadcast"
    for (int i = 0; i < 4; i++)
      for (int b = 0; b < 4; b++)
        for (int j = 0; j < 4; j++) {
          double eb = 0, en = 0;
          for (int k = 0; k < 4; k++) {
            eb = fma(hA[16 * k + 4 * abid + i], hB[16 * k + 4 * b + j], eb);
            en = fma(hA[16 * k + 4 * b + i], hB[16 * k + 4 * b + j], en);
          }
          const double got = hD[16 * i + 4 * b + j];
          worst_b = fmax(worst_b, fabs(got - eb) / fabs(eb));
          worst_n = fmax(worst_n, fabs(got - en) / fabs(en));
        }
    printf("cbsz 2 abid %d: worst relative difference to the broadcast form %.2e, to the plain form %.2e\n", abid, worst_b, worst_n);
  }
  return 0;
}
