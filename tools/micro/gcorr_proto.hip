// gcorr_proto.hip -- what the node-separable correlation costs before it is built: per (segment, pixel)
// pair   G[n][k] = sum_cells XY[n][cell] R[cell][k]   as v_mfma_f64_16x16x4 (16 quadrature nodes = the M rows, response rows
// streamed from L2 as the B operand, XY formed from LDS tables), then   P[s][k] = sum_n Z[n][s] G[n][k]   by a second MFMA whose
// B operand IS the first one's accumulator (register r of lane l holds G[4r + l/16][l % 16] = B[kk = l/16][j = l%16] of k-step
// r), and the diagonal sum out[k - s] += P[s][k] with ds_add_f64 into a wave-private tick array.
// Synthetic sizes of the module0 bench set: 192 cells, 304 response ticks, 32 shifts per pair.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/gcorr_proto.hip -o /tmp/gcorr_proto && /tmp/gcorr_proto
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

#define NKP 4510          // padded response row
#define NCELLS 2025
#define NC 192            // cells per pair
#define KT 19             // 16-tick k tiles per pair (304 ticks)
#define ST 2              // 16-shift tiles
#define NCOL 14

template <int PF, int WPB>
__global__ void __launch_bounds__(256, WPB) gproto(const double* __restrict__ resp, const double* __restrict__ tabs,
                                                   float* __restrict__ out, int n_pairs, int variant, int ngrp) {
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pair = blockIdx.x;
  __shared__ double s_X[16][NCOL + 1], s_Y[16][NCOL + 1];
  __shared__ unsigned short s_cell[NC], s_col[NC], s_j[NC];
  __shared__ double s_out[4][KT * 16 + 64];
  // tables of this pair (kernel 1 would have written them): 16 x 14 X, 16 x 14 Y, cell list
  const double* tb = tabs + (size_t)(pair & 1023) * 1024;
  for (int i = tid; i < 16 * NCOL; i += 256) {
    s_X[i / NCOL][i % NCOL] = tb[i];
    s_Y[i / NCOL][i % NCOL] = tb[256 + i];
  }
  unsigned h = pair * 2654435761u;
  // cells of a pair: a 14 x 14 patch of the (i, j) plane with i, j <= 20 -- what a pixel and its first neighbours reach
  // (|x_p - x| <= 1.5 pitch + 4 sigma_T = 20 response bins); ticks: the ~300 of the table's support plus the shift range
  const int i0 = h % 7, j0 = (h >> 8) % 7;
  const int kbase = 1500 + (h >> 20) % 64;
  if (tid < NC) {
    const int col = tid / NCOL, j = tid % NCOL;
    s_col[tid] = col < NCOL ? col : NCOL - 1; s_j[tid] = j;
    s_cell[tid] = (i0 + (col < NCOL ? col : NCOL - 1)) * 45 + j0 + j;
  }
  for (int i = tid; i < 4 * (KT * 16 + 64); i += 256) (&s_out[0][0])[i] = 0;
  __syncthreads();
  const int kk = lane >> 4, jj = lane & 15;
  // Z as the A operand of the second product: lane 16 kk + i holds Z[node 4 q + kk][shift 16 st + i]
  double zA[ST][4];
#pragma unroll
  for (int st = 0; st < ST; st++)
#pragma unroll
    for (int q = 0; q < 4; q++) zA[st][q] = tb[512 + (4 * q + kk) * 32 + 16 * st + jj];
  double* ow = s_out[wv];
  for (int kt = wv; kt < KT; kt += 4) {
    const int k0 = kbase + 16 * kt;
    d4 acc = {0, 0, 0, 0};
    const double* rp = resp + k0 + jj;
    double b[PF];
#pragma unroll
    for (int u = 0; u < PF; u++) b[u] = rp[(size_t)s_cell[4 * u + kk] * NKP];
#pragma unroll 1
    for (int g0 = 0; g0 < ngrp; g0 += PF) {
      double bn[PF];
      const int g1 = g0 + PF;
#pragma unroll
      for (int u = 0; u < PF; u++) bn[u] = (g1 + u < ngrp) ? rp[(size_t)s_cell[4 * (g1 + u) + kk] * NKP] : 0.0;
#pragma unroll
      for (int u = 0; u < PF; u++) {
        const int c = 4 * (g0 + u) + kk;
        const double a = s_X[jj][s_col[c]] * s_Y[jj][s_j[c]];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[u], acc, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < PF; u++) b[u] = bn[u];
    }
    if (variant & 1) {
      // second product + diagonal sum
#pragma unroll
      for (int st = 0; st < ST; st++) {
        d4 p = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; q++) p = __builtin_amdgcn_mfma_f64_16x16x4f64(zA[st][q], acc[q], p, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int s = 16 * st + 4 * r + kk;               // P[s][k0 + jj]
          atomicAdd(&ow[16 * kt + jj - s + 48], p[r]);
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; r++) ow[16 * kt + jj + 4 * r + kk] += acc[r];
    }
  }
  __syncthreads();
  for (int i = tid; i < KT * 16; i += 256)
    out[(size_t)pair * (KT * 16) + i] = (float)(s_out[0][i + 48] + s_out[1][i + 48] + s_out[2][i + 48] + s_out[3][i + 48]);
}

// Variant 2: the A operand (XY of the lane's node and cell) and the row offsets are tabulated once per pair in LDS, a wave owns
// 32-tick tiles: one 16-byte load per lane (ticks k0 + 2j, k0 + 2j + 1 of the lane's cell) feeds two MFMAs (even / odd columns).
typedef double d2 __attribute__((ext_vector_type(2)));
template <int PF, int WPB>
__global__ void __launch_bounds__(256, WPB) gproto2(const double* __restrict__ resp, const double* __restrict__ tabs,
                                                    float* __restrict__ out, int n_pairs, int variant, int ngrp) {
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pair = blockIdx.x;
  __shared__ double s_X[16][NCOL + 1], s_Y[16][NCOL + 1];
  __shared__ double s_A[NC / 4][64];
  __shared__ int s_off[NC];
  __shared__ double s_out[4][KT * 16 + 96];
  const double* tb = tabs + (size_t)(pair & 1023) * 1024;
  for (int i = tid; i < 16 * NCOL; i += 256) {
    s_X[i / NCOL][i % NCOL] = tb[i];
    s_Y[i / NCOL][i % NCOL] = tb[256 + i];
  }
  unsigned h = pair * 2654435761u;
  const int i0 = h % 7, j0 = (h >> 8) % 7;
  const int kbase = 1500 + 2 * ((h >> 20) % 32);
  __syncthreads();
  for (int e = tid; e < NC * 16; e += 256) {          // A[g][16 kk + i] = X[i][col(4g + kk)] Y[i][j(4g + kk)]
    const int c = e >> 4, i = e & 15;
    const int col = min(c / NCOL, NCOL - 1), j = c % NCOL;
    s_A[c >> 2][16 * (c & 3) + i] = s_X[i][col] * s_Y[i][j];
    if (i == 0) s_off[c] = ((i0 + col) * 45 + j0 + j) * NKP;
  }
  for (int i = tid; i < 4 * (KT * 16 + 96); i += 256) (&s_out[0][0])[i] = 0;
  __syncthreads();
  const int kk = lane >> 4, jj = lane & 15;
  double zA[ST][4];
#pragma unroll
  for (int st = 0; st < ST; st++)
#pragma unroll
    for (int q = 0; q < 4; q++) zA[st][q] = tb[512 + (4 * q + kk) * 32 + 16 * st + jj];
  double* ow = s_out[wv];
  const int nt32 = (KT + 1) / 2;
  for (int kt = wv; kt < nt32; kt += 4) {
    const int k0 = kbase + 32 * kt;
    d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    const double* rp = resp + k0 + 2 * jj;
    d2 b[PF];
#pragma unroll
    for (int u = 0; u < PF; u++) b[u] = *(const d2*)(rp + s_off[4 * u + kk]);
#pragma unroll 1
    for (int g0 = 0; g0 < ngrp; g0 += PF) {
      d2 bn[PF];
      const int g1 = g0 + PF;
#pragma unroll
      for (int u = 0; u < PF; u++) bn[u] = (g1 + u < ngrp) ? *(const d2*)(rp + s_off[4 * (g1 + u) + kk]) : (d2){0.0, 0.0};
#pragma unroll
      for (int u = 0; u < PF; u++) {
        const double a = s_A[g0 + u][lane];
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[u].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[u].y, acc1, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < PF; u++) b[u] = bn[u];
    }
#pragma unroll
    for (int st = 0; st < ST; st++) {
      d4 p0 = {0, 0, 0, 0}, p1 = {0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < 4; q++) {
        p0 = __builtin_amdgcn_mfma_f64_16x16x4f64(zA[st][q], acc0[q], p0, 0, 0, 0);
        p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(zA[st][q], acc1[q], p1, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int s = 16 * st + 4 * r + kk;                 // P[s][k0 + 2 jj (+ 1)]
        atomicAdd(&ow[32 * kt + 2 * jj - s + 48], p0[r]);
        atomicAdd(&ow[32 * kt + 2 * jj + 1 - s + 48], p1[r]);
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < KT * 16; i += 256)
    out[(size_t)pair * (KT * 16) + i] = (float)(s_out[0][i + 48] + s_out[1][i + 48] + s_out[2][i + 48] + s_out[3][i + 48]);
}

// ---- issue rate of the whole-wave DPP move against v_fma_f64 (DESIGN section 4's "issue limit" claim as a number) --------------
__device__ __forceinline__ int dpp_shl1(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, 0x130, 0xF, 0xF, false); }
template <int NFMA, int NDPP>
__global__ void __launch_bounds__(256, 5) issue_mix(int iters, double a, double b, double* out) {
  double acc[8];
  int m[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { acc[i] = a + i + threadIdx.x; m[i] = threadIdx.x + i; }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < NFMA / 8; r++)
#pragma unroll
      for (int i = 0; i < 8; i++) acc[i] = fma(acc[i], a, b);
#pragma unroll
    for (int r = 0; r < NDPP / 8; r++)
#pragma unroll
      for (int i = 0; i < 8; i++) m[i] = dpp_shl1(m[i], m[(i + 1) & 7]);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s += acc[i] + m[i];
  if (s == 12345.678) out[0] = s;
}

int main() {
  const int n_pairs = 131072;
  double *resp, *tabs; float* out;
  hipMalloc(&resp, (size_t)NCELLS * NKP * 8);
  hipMalloc(&tabs, 1024 * 1024 * 8);
  hipMalloc(&out, (size_t)n_pairs * KT * 16 * 4);
  std::vector<double> h((size_t)NCELLS * NKP);
  for (auto& v : h) v = rand() / (double)RAND_MAX;
  hipMemcpy(resp, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(tabs, h.data(), 1024 * 1024 * 8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](auto launch, const char* name, double fma) {
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s %8.3f ms  %7.2f TFLOP/s  (%.2f ms per 550k pairs)\n", name, ms, 2 * fma / (ms * 1e-3) / 1e12, ms * 550000.0 / n_pairs);
  };
  const double fma_g = (double)n_pairs * 16.0 * NC * KT * 16, fma_p = (double)n_pairs * KT * ST * 4 * 1024.0;
  time([&] { hipLaunchKernelGGL((gproto<8, 4>), dim3(n_pairs), dim3(256), 0, 0, resp, tabs, out, n_pairs, 0, NC / 4); }, "G only, prefetch 8, 4 WG/CU", fma_g);
  time([&] { hipLaunchKernelGGL((gproto<8, 4>), dim3(n_pairs), dim3(256), 0, 0, resp, tabs, out, n_pairs, 1, NC / 4); }, "G + P + diagonal ds_add, prefetch 8, 4 WG/CU", fma_g + fma_p);
  time([&] { hipLaunchKernelGGL((gproto<4, 4>), dim3(n_pairs), dim3(256), 0, 0, resp, tabs, out, n_pairs, 1, NC / 4); }, "G + P + diagonal ds_add, prefetch 4, 4 WG/CU", fma_g + fma_p);
  time([&] { hipLaunchKernelGGL((gproto<12, 4>), dim3(n_pairs), dim3(256), 0, 0, resp, tabs, out, n_pairs, 1, NC / 4); }, "G + P + diagonal ds_add, prefetch 12, 4 WG/CU", fma_g + fma_p);
  time([&] { hipLaunchKernelGGL((gproto<8, 6>), dim3(n_pairs), dim3(256), 0, 0, resp, tabs, out, n_pairs, 1, NC / 4); }, "G + P + diagonal ds_add, prefetch 8, 6 WG/CU", fma_g + fma_p);
  time([&] { hipLaunchKernelGGL((gproto<8, 8>), dim3(n_pairs), dim3(256), 0, 0, resp, tabs, out, n_pairs, 1, NC / 4); }, "G + P + diagonal ds_add, prefetch 8, 8 WG/CU", fma_g + fma_p);
  const double fma_g2 = (double)n_pairs * 16.0 * NC * ((KT + 1) / 2) * 32, fma_p2 = (double)n_pairs * ((KT + 1) / 2) * 2 * ST * 4 * 1024.0;
  time([&] { hipLaunchKernelGGL((gproto2<4, 3>), dim3(n_pairs), dim3(256), 0, 0, resp, tabs, out, n_pairs, 1, NC / 4); }, "v2: A table, 32-tick tiles, prefetch 4, 3 WG/CU", fma_g2 + fma_p2);
  time([&] { hipLaunchKernelGGL((gproto2<8, 3>), dim3(n_pairs), dim3(256), 0, 0, resp, tabs, out, n_pairs, 1, NC / 4); }, "v2: A table, 32-tick tiles, prefetch 8, 3 WG/CU", fma_g2 + fma_p2);
  time([&] { hipLaunchKernelGGL((gproto2<4, 4>), dim3(n_pairs), dim3(256), 0, 0, resp, tabs, out, n_pairs, 1, NC / 4); }, "v2: A table, 32-tick tiles, prefetch 4, 4 WG/CU", fma_g2 + fma_p2);
  time([&] { hipLaunchKernelGGL((gproto2<8, 4>), dim3(n_pairs), dim3(256), 0, 0, resp, tabs, out, n_pairs, 1, NC / 4); }, "v2: A table, 32-tick tiles, prefetch 8, 4 WG/CU", fma_g2 + fma_p2);
  time([&] { hipLaunchKernelGGL((gproto2<6, 4>), dim3(n_pairs), dim3(256), 0, 0, resp, tabs, out, n_pairs, 1, NC / 4); }, "v2: A table, 32-tick tiles, prefetch 6, 4 WG/CU", fma_g2 + fma_p2);
  // issue mix: wave-instructions per cycle and SIMD = blocks*4 waves*(NFMA+NDPP)*iters / (ms * clock * 1024 SIMDs)
  double* d; hipMalloc(&d, 8);
  const int iters = 4000, blocks = 256 * 5;
  auto mix = [&](auto launch, const char* name, int nf, int nd) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst = (double)blocks * 4 * iters, cyc = ms * 1e-3 * 2.4e9 * 1024;
    printf("%-44s %8.3f ms  cycles per wave-instruction and SIMD at 2.4 GHz: %.2f (%d fma + %d dpp per iteration)\n", name, ms,
           cyc / (inst * (nf + nd)), nf, nd);
  };
  mix([&] { hipLaunchKernelGGL((issue_mix<64, 0>), dim3(blocks), dim3(256), 0, 0, iters, 1.0000001, 1e-9, d); }, "64 v_fma_f64", 64, 0);
  mix([&] { hipLaunchKernelGGL((issue_mix<0, 64>), dim3(blocks), dim3(256), 0, 0, iters, 1.0000001, 1e-9, d); }, "64 v_mov_b32 dpp wave_shl:1", 0, 64);
  mix([&] { hipLaunchKernelGGL((issue_mix<64, 16>), dim3(blocks), dim3(256), 0, 0, iters, 1.0000001, 1e-9, d); }, "64 v_fma_f64 + 16 dpp (a mac_shift block)", 64, 16);
  mix([&] { hipLaunchKernelGGL((issue_mix<64, 32>), dim3(blocks), dim3(256), 0, 0, iters, 1.0000001, 1e-9, d); }, "64 v_fma_f64 + 32 dpp", 64, 32);
  return 0;
}
