// wg_launch.hip -- what does a workgroup cost before it computes anything?  gcorr_kernel with its tile loop and table staging
// switched off still takes 1.25 ms per 549 k pairs (profiles/r03_phases_gcorr.log): is that the dispatcher's rate for
// two-wave workgroups with 13 KB of LDS, or the chain of dependent loads (flag / GInfo -> record -> LDS -> barrier)?
//   a) empty workgroups (threads, dynamic LDS as given), one per pair
//   b) the same with one 64-byte record loaded per workgroup and one word stored
//   c) b + a second, dependent load (address from the first)
//   d) a persistent grid (waves resident once, each walks pairs with the next pair's record requested one pair ahead)
// hipcc --offload-arch=gfx950 -O3 tools/micro/wg_launch.hip -o /tmp/wg_launch && /tmp/wg_launch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

struct Rec { int v[14]; unsigned long long off; };      // 64 bytes, like GInfo

template <int T>
__global__ void __launch_bounds__(T) k_empty(int* out) {
  extern __shared__ double s[];
  if (threadIdx.x == 0 && out == (int*)1) s[0] = 1;       // (keeps the LDS allocation)
}

template <int T, int DEP>
__global__ void __launch_bounds__(T) k_load(const Rec* __restrict__ rec, const double* __restrict__ pool, int* __restrict__ out, int n) {
  extern __shared__ double s[];
  const int p = blockIdx.x;
  if (p >= n) return;
  const Rec r = rec[p];
  double acc = 0;
  if (DEP >= 1) {
    const double* q = pool + r.off;
    acc = q[threadIdx.x];
    if (DEP >= 2) {                       // through LDS and a barrier, like the staging of the tables
      s[threadIdx.x] = acc;
      __syncthreads();
      acc = s[(threadIdx.x + 17) % T];
    }
  }
  if (threadIdx.x == 0) out[p] = r.v[0] + (int)acc;
}

// persistent: each wave walks pairs w, w + W, ...; the record of the pair after this one is requested before this one is used
template <int DEP>
__global__ void __launch_bounds__(64) k_persist(const Rec* __restrict__ rec, const double* __restrict__ pool, int* __restrict__ out, int n) {
  const int W = gridDim.x;
  int p = blockIdx.x;
  if (p >= n) return;
  Rec r = rec[p];
  double a = DEP ? pool[r.off + threadIdx.x] : 0.0;
  while (p < n) {
    const int pn = p + W;
    Rec rn = r;
    double an = 0;
    if (pn < n) {
      rn = rec[pn];
      if (DEP) an = pool[rn.off + threadIdx.x];
    }
    if (threadIdx.x == 0) out[p] = r.v[0] + (int)a;
    r = rn; a = an; p = pn;
  }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <class F>
static float timeit(F f) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < 5; i++) f();
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / 5;
}

int main() {
  const int n = 548598;
  const size_t per = 1900;                     // doubles per record (15 KB)
  std::vector<Rec> h(n);
  for (int i = 0; i < n; i++) { h[i].v[0] = i; h[i].off = (unsigned long long)i * per; }
  Rec* d_rec; double* d_pool; int* d_out;
  CK(hipMalloc(&d_rec, n * sizeof(Rec)));
  CK(hipMalloc(&d_pool, (size_t)n * per * 8 + 4096));
  CK(hipMalloc(&d_out, n * 4));
  CK(hipMemcpy(d_rec, h.data(), n * sizeof(Rec), hipMemcpyHostToDevice));
  CK(hipMemset(d_pool, 0, (size_t)n * per * 8 + 4096));
  for (int lds : {0, 6656, 13312, 26624}) {
    printf("LDS %5d B: empty 64 thr %.3f ms, 128 thr %.3f ms, 256 thr %.3f ms\n", lds,
           timeit([&] { hipLaunchKernelGGL(k_empty<64>, dim3(n), dim3(64), lds, 0, d_out); }),
           timeit([&] { hipLaunchKernelGGL(k_empty<128>, dim3(n), dim3(128), lds, 0, d_out); }),
           timeit([&] { hipLaunchKernelGGL(k_empty<256>, dim3(n), dim3(256), lds, 0, d_out); }));
  }
  for (int lds : {6656, 13312}) {
    printf("LDS %5d B, 128 thr: record only %.3f ms, + dependent load %.3f ms, + LDS and barrier %.3f ms\n", lds,
           timeit([&] { hipLaunchKernelGGL((k_load<128, 0>), dim3(n), dim3(128), lds, 0, d_rec, d_pool, d_out, n); }),
           timeit([&] { hipLaunchKernelGGL((k_load<128, 1>), dim3(n), dim3(128), lds, 0, d_rec, d_pool, d_out, n); }),
           timeit([&] { hipLaunchKernelGGL((k_load<128, 2>), dim3(n), dim3(128), lds, 0, d_rec, d_pool, d_out, n); }));
    printf("LDS %5d B,  64 thr: record only %.3f ms, + dependent load %.3f ms, + LDS and barrier %.3f ms\n", lds,
           timeit([&] { hipLaunchKernelGGL((k_load<64, 0>), dim3(n), dim3(64), lds, 0, d_rec, d_pool, d_out, n); }),
           timeit([&] { hipLaunchKernelGGL((k_load<64, 1>), dim3(n), dim3(64), lds, 0, d_rec, d_pool, d_out, n); }),
           timeit([&] { hipLaunchKernelGGL((k_load<64, 2>), dim3(n), dim3(64), lds, 0, d_rec, d_pool, d_out, n); }));
  }
  for (int wpc : {4, 8, 12, 16}) {
    const int W = 256 * wpc;
    printf("persistent, %2d waves per CU: record only %.3f ms, + dependent load %.3f ms\n", wpc,
           timeit([&] { hipLaunchKernelGGL(k_persist<0>, dim3(W), dim3(64), 0, 0, d_rec, d_pool, d_out, n); }),
           timeit([&] { hipLaunchKernelGGL(k_persist<1>, dim3(W), dim3(64), 0, 0, d_rec, d_pool, d_out, n); }));
  }
  return 0;
}
