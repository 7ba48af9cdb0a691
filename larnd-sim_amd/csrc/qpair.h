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

// exp(x) for x <= 0 with a 64-entry table: x = (64 e + j) ln2 / 64 + r, |r| <= ln2 / 128, exp(x) = 2^e * 2^(j/64) * exp(r), exp(r) by its
// Taylor polynomial to r^4 (truncation |r|^5 / 120 <= 3.9e-14 relative: the tables it fills carry a quadrature error of 1e-7 of
// their peak, and 1e-13 is what round 3's degree-5 form was held to; one multiply-add fewer per Gaussian).  Against exp_neg's
// 13-term polynomial: nine multiply-adds fewer for one read of an LDS table that 64 lanes hit without a bank conflict (entry j
// lies in banks 2j, 2j + 1).  gtables_wave_kernel spends 58 % of its VALU instructions in Gaussians.  `tab` = g_exp2_64 staged in LDS.
__device__ const double g_exp2_64[64] = {
  0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
  0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,
  0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
  0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
  0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
  0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
  0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
  0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
  0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
  0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
  0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,
  0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
  0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
  0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
  0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
  0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0,
};
__device__ __forceinline__ double exp_neg_tab(double x, const double* __restrict__ tab) {
  const double xc = x < -708.0 ? -708.0 : x;
  const double k = rint(xc * 0x1.71547652b82fep+6);             // 64 / ln 2
  double r = fma(-k, 0x1.62e4200000000p-7, xc);                 // ln2 / 64: high part (k times it is exact), low part
  r = fma(-k, 0x1.fdf473de6af28p-28, r);
  const int ki = (int)k;
  double p = 1.0 / 24.0;
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  const double v = ldexp(tab[ki & 63] * p, ki >> 6);
  return x < -708.0 ? 0.0 : v;
}

// MC of them side by side: the index parts first, so that the MC table reads are in flight together and the polynomials run under
// them (one after the other, as the compiler schedules separate calls, a wave waited on LDS MC times per member).  No clamp at -708:
// the argument is bounded below once (one v_max instead of a compare and four selects per value) and ldexp underflows to 0 by itself
// -- values in (-745, -708) come out as subnormals instead of 0.
template <int MC>
__device__ __forceinline__ void exp_neg_tab_n(const double (&x)[4], double (&v)[4], const double* __restrict__ tab) {
  double k[4], t[4], xc[4];
  int ki[4];
#pragma unroll
  for (int m = 0; m < 4; m++) {
    if (m < MC) {
      xc[m] = fmax(x[m], -2000.0);
      k[m] = rint(xc[m] * 0x1.71547652b82fep+6);
      ki[m] = (int)k[m];
      t[m] = tab[ki[m] & 63];
    }
  }
#pragma unroll
  for (int m = 0; m < 4; m++) {
    if (m < MC) {
      double r = fma(-k[m], 0x1.62e4200000000p-7, xc[m]);
      r = fma(-k[m], 0x1.fdf473de6af28p-28, r);
      double p = 1.0 / 24.0;
      p = fma(p, r, 1.0 / 6.0);
      p = fma(p, r, 0.5);
      p = fma(p, r, 1.0);
      p = fma(p, r, 1.0);
      v[m] = ldexp(t[m] * p, ki[m] >> 6);
    }
  }
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

__host__ __device__ __forceinline__ void edge_ks(const LdsimConsts* c, const CurArgs& A, int* edge_k, int& k_stage_lo, int& k_stage_hi) {
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

