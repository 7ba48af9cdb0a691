// qpair.h -- per-pair record of the quadrature weight kernels (pair_setup_kernel in kernels_qsetup.hip writes it, one thread
// per pair; qweights_kernel reads it) and small device helpers they share.
#pragma once
#include "split_common.h"

struct PairParams {
  int32_t status;            // 0 = nothing to emit, 1 = compute, 2 = too many nodes: monolithic kernel
  int32_t NQ, iz_lo, iz_hi, it0, T, it_w0, it_w1;
  double x_p, y_p, x_start, y_start, x_step, y_step, sgnx, sgny, sT, sx, sy;
  double z_start_int, z_step, z_anode, t_start, sz;
  double uxr, uyr, uzr, i2T, i2L, kappa, s_lo, qlen, wscale, thr;
};

// the record's doubles as the wave keeps them in LDS (read where a phase needs them: short live ranges, no SGPR spills)
enum { PP_X_P = 0, PP_Y_P, PP_X_START, PP_Y_START, PP_X_STEP, PP_Y_STEP, PP_SGNX, PP_SGNY, PP_ST, PP_SX, PP_SY,
       PP_Z_START_INT, PP_Z_STEP, PP_Z_ANODE, PP_T_START, PP_SZ,
       PP_UXR, PP_UYR, PP_UZR, PP_I2T, PP_I2L, PP_KAPPA, PP_S_LO, PP_QLEN, PP_WSCALE, PP_THR, PP_COUNT };
static_assert(sizeof(PairParams) == 32 + 8 * PP_COUNT, "PairParams layout");

__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// exp(x) for x <= 0: k = rint(x log2 e), r = x - k ln2 (two-term), exp(r) by its Taylor polynomial to r^13 (|r| <= 0.347:
// truncation 4e-18), scaled by 2^k.  Below -708 the result is 0 (the reference's exp underflows to subnormals there: < 1e-307).
__device__ __forceinline__ double exp_neg(double x) {
  if (x < -708.0) return 0.0;
  const double k = rint(x * 1.4426950408889634074);
  double r = fma(-k, 6.93147180369123816490e-01, x);
  r = fma(-k, 1.90821492927058770002e-10, r);
  double p = 1.0 / 6227020800.0;
  p = fma(p, r, 1.0 / 479001600.0);
  p = fma(p, r, 1.0 / 39916800.0);
  p = fma(p, r, 1.0 / 3628800.0);
  p = fma(p, r, 1.0 / 362880.0);
  p = fma(p, r, 1.0 / 40320.0);
  p = fma(p, r, 1.0 / 5040.0);
  p = fma(p, r, 1.0 / 720.0);
  p = fma(p, r, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)k);
}

// the same value without a branch (x < -708 selects 0 at the end): several of these in a row interleave in the schedule
__device__ __forceinline__ double exp_neg_sel(double x) {
  const double xc = x < -708.0 ? -708.0 : x;
  const double k = rint(xc * 1.4426950408889634074);
  double r = fma(-k, 6.93147180369123816490e-01, xc);
  r = fma(-k, 1.90821492927058770002e-10, r);
  double p = 1.0 / 6227020800.0;
  p = fma(p, r, 1.0 / 479001600.0);
  p = fma(p, r, 1.0 / 39916800.0);
  p = fma(p, r, 1.0 / 3628800.0);
  p = fma(p, r, 1.0 / 362880.0);
  p = fma(p, r, 1.0 / 40320.0);
  p = fma(p, r, 1.0 / 5040.0);
  p = fma(p, r, 1.0 / 720.0);
  p = fma(p, r, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  const double v = ldexp(p, (int)k);
  return x < -708.0 ? 0.0 : v;
}

// response shift of slice iz (k - M * it for the tick in the middle of the slice's window), as weights_kernel / qweights_kernel
template <int M>
__device__ __forceinline__ int slice_shift_of(const LdsimConsts* c, double z_start_int, double z_step, double z_anode,
                                              double t_start, int iz, double& z, double& t0, bool& ambiguous) {
  const double dt = c->time_sampling, dtr = c->response_sampling, TW = c->time_window;
  z = z_start_int + iz * z_step;
  t0 = fabs(z - z_anode) / c->v_drift - TW;
  int it_ref = (int)((t0 + 0.5 * TW - t_start) / dt);
  if (it_ref < 0) it_ref = 0;
  const double tt = t_start + it_ref * dt;
  const double val = (tt - t0) / dtr;
  const double kr = py_round(val);
  ambiguous = fabs(val - kr) > 0.5 - 1e-7;
  return (int)kr - M * it_ref;
}

__device__ __forceinline__ void edge_ks(const LdsimConsts* c, const CurArgs& A, int* edge_k, int& k_stage_lo, int& k_stage_hi) {
  const double V = c->time_window / c->response_sampling;
  edge_k[0] = 0; edge_k[1] = -1; edge_k[2] = -1;
  int ka = (int)floor(V - 0.5 - 1e-6);
  if ((double)ka + 0.5 >= V - 1e-6) ka--;
  const int kn = (int)ceil(V + 0.5 + 1e-6);
  int k_top = kn - 1;
  int ne = 1;
  for (int k = ka + 1; k <= k_top && ne < NEDGE; k++) edge_k[ne++] = k;
  if (k_top - ka > NEDGE - 1) k_top = ka + NEDGE - 1;
  k_stage_hi = min(min(k_top, A.nk - 1), A.k_last);
  k_stage_lo = max(0, A.k_first);
}

