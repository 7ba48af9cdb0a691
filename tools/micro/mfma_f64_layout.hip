// mfma_f64_layout.hip -- which lane holds which element of A, B and D for v_mfma_f64_16x16x4_f64 on gfx950?
// Hypothesis (CDNA3 ISA, "16x16x4 f64"): A[i][k] in lane 16k + i, B[k][j] in lane 16k + j, D[4(l / 16) + r][l % 16] in register
// r of lane l.  The kernel multiplies A[i][k] = 100 i + k by B[k][j] = (j == jsel) (one-hot column) and by B = k-selector
// matrices, and the host checks D against the plain triple loop.  hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_f64_layout.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k(const double* A /*16x4*/, const double* B /*4x16*/, double* D /*16x16*/) {
  const int l = threadIdx.x;
  const double a = A[(l % 16) * 4 + (l / 16)];      // A[i = l % 16][k = l / 16]
  const double b = B[(l / 16) * 16 + (l % 16)];     // B[k = l / 16][j = l % 16]
  d4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; r++) D[l * 4 + r] = acc[r];                               // raw: register r of lane l
}

int main() {
  double hA[64], hB[64], hD[256], ref[256];
  for (int i = 0; i < 16; i++) for (int kk = 0; kk < 4; kk++) hA[i * 4 + kk] = sin(1.0 + 3.1 * i + 0.7 * kk);
  for (int kk = 0; kk < 4; kk++) for (int j = 0; j < 16; j++) hB[kk * 16 + j] = cos(0.3 + 1.3 * kk + 2.9 * j);
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
    double s = 0;
    for (int kk = 0; kk < 4; kk++) s = fma(hA[i * 4 + kk], hB[kk * 16 + j], s);
    ref[i * 16 + j] = s;
  }
  double *dA, *dB, *dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  // find the element every (lane, register) holds: the reference values are all different
  int bad = 0, rule_a = 0, rule_b = 0;
  for (int l = 0; l < 64; l++)
    for (int r = 0; r < 4; r++) {
      int found = -1;
      for (int e = 0; e < 256; e++)
        if (fabs(hD[l * 4 + r] - ref[e]) < 1e-13) found = found < 0 ? e : -2;
      if (found < 0) { bad++; continue; }
      const int i = found / 16, j = found % 16;
      rule_a += (i == 4 * (l / 16) + r && j == l % 16);
      rule_b += (i == 4 * r + l / 16 && j == l % 16);
      if (l < 2 || l == 16 || l == 63) printf("lane %2d reg %d holds D[%2d][%2d]\n", l, r, i, j);
    }
  printf("v_mfma_f64_16x16x4_f64, A[i][k] in lane 16k+i, B[k][j] in lane 16k+j: unmatched %d; D[4(l/16)+r][l%%16]: %d of 256; "
         "D[4r+l/16][l%%16]: %d of 256\n", bad, rule_a, rule_b);
  double worst = (rule_a == 256 || rule_b == 256) && !bad ? 0 : 1;
  return worst < 1e-12 ? 0 : 1;
}
