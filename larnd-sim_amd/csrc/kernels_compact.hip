// kernels_compact.hip -- compact form of a chain launch's results for the trip across PCIe.
//
// ldsim_chain_download moves the reference's dense per-pixel arrays -- adc_list / adc_ticks_list / digitised ADC [U][30],
// track_pixel_map [U][50], current_fractions [U][30][50] f64 = 13 KB per unique pixel, 2.5 GB per 50k-segment launch -- of
// which the exporter (fee.export_to_hdf5, fee.py:143-344) reads the pixels that hold a hit, the slots up to the first ADC at
// the pedestal, and the fractions of the track slots the pixel really has.  Here exactly that is gathered in HBM first:
//   per hit pixel   row index in the dense arrays, pixel id, batch, number of hits, number of track slots -- and the same row
//                   (0 hits, 0 slots) for the first unique pixel of every batch even when it holds no hit: the driver counts
//                   batches per export (sim.WRITE_BATCH_SIZE) from what it is handed, like the reference, which appends every
//                   simulated batch (cli/simulate_pixels.py:1207-1214), and the exporter keeps its clock state in a batch's row 0
//   per track slot  the segment index (track_pixel_map entry)
//   per hit         the 24-byte row of the multi-GPU exchange {batch, pixel, ADC code, slot, tick} + the integrated charge
//   per hit x slot  the backtracking fraction
// a few MB per launch; larndsim_amd.chain.expand_compact rebuilds the dense rows of the hit pixels on the host (tested equal
// to the dense download).
#include "ldsim_args.h"

int sort_exclusive_scan_i32(ldsim_ctx*, const int32_t*, int32_t*, int64_t);

__global__ void __launch_bounds__(256) compact_count_kernel(const int32_t* __restrict__ hit_count, const int64_t* __restrict__ tpm,
                                                            const int32_t* __restrict__ ubatch,
                                                            int M, int64_t U, int32_t* __restrict__ c_hp, int32_t* __restrict__ c_trk,
                                                            int32_t* __restrict__ c_frac) {
  const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (u >= U) return;
  const int nh = hit_count[u];
  int nt = 0;
  if (nh > 0)
    for (int m = 0; m < M; m++) nt += tpm[u * M + m] >= 0;       // (filled from slot 0 on, -1 behind)
  c_hp[u] = nh > 0 || u == 0 || ubatch[u - 1] != ubatch[u];      // (a batch's first row always travels)
  c_trk[u] = nt;
  c_frac[u] = nh * nt;
}

__global__ void __launch_bounds__(256) compact_fill_kernel(const int32_t* __restrict__ hit_count, const int32_t* __restrict__ hit_off,
                                                           const int32_t* __restrict__ o_hp, const int32_t* __restrict__ o_trk,
                                                           const int32_t* __restrict__ o_frac, const int32_t* __restrict__ c_trk,
                                                           const int32_t* __restrict__ upix, const int32_t* __restrict__ ubatch,
                                                           const int64_t* __restrict__ tpm, const double* __restrict__ adc_list,
                                                           const double* __restrict__ fractions, int A, int M, int64_t U,
                                                           int32_t* __restrict__ hp_rows /* [n_hp][5] */, int64_t* __restrict__ trk_seg,
                                                           double* __restrict__ hit_charge, double* __restrict__ frac_val) {
  const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (u >= U) return;
  const int nh = hit_count[u] > 0 ? hit_count[u] : 0;
  const bool first = u == 0 || ubatch[u - 1] != ubatch[u];
  if (nh <= 0 && !first) return;
  const int nt = c_trk[u];
  int32_t* r = hp_rows + (int64_t)o_hp[u] * 5;
  // (bit 8 of the last word: the pixel is the first row of its batch in the dense arrays -- the exporter's clock-rollover
  // bookkeeping treats row 0 of what it is handed specially, fee.py:164-183,267-277)
  r[0] = (int32_t)u; r[1] = upix[u]; r[2] = ubatch[u]; r[3] = nh; r[4] = nt | (first ? 256 : 0);
  for (int m = 0; m < nt; m++) trk_seg[o_trk[u] + m] = tpm[u * M + m];
  for (int h = 0; h < nh; h++) {
    hit_charge[hit_off[u] + h] = adc_list[u * A + h];
    if (fractions)
      for (int m = 0; m < nt; m++) frac_val[o_frac[u] + h * nt + m] = fractions[((int64_t)u * A + h) * M + m];
  }
}

// gathers the compact form of the last chain launch in HBM; sizes[4] = hit pixels, hits, track entries, fraction entries
extern "C" int ldsim_chain_compact_build(ldsim_ctx* ctx, int64_t* sizes) {
  LDSIM_ENTER(ctx);
  if (!ctx || !sizes) { ldsim_set_error("null argument"); return LDSIM_EINVAL; }
  HIPCHK(hipSetDevice(ctx->device));
  const int64_t U = ctx->chain_U;
  const int A = ctx->h_consts.max_adc_values, M = ctx->h_consts.max_tracks_per_pixel;
  for (int k = 0; k < 4; k++) ctx->cpt_n[k] = 0;
  ctx->cpt_gen = ctx->out_gen;
  if (U == 0) { for (int k = 0; k < 4; k++) sizes[k] = 0; return 0; }
  hipStream_t st = ctx->stream;
  int rc;
  if ((rc = ldsim_ensure(ctx, SB_CPT, (size_t)(6 * U + 16) * 4))) return rc;
  int32_t* c_hp = (int32_t*)ctx->scratch[SB_CPT].p;
  int32_t *c_trk = c_hp + U, *c_frac = c_trk + U, *o_hp = c_frac + U, *o_trk = o_hp + U, *o_frac = o_trk + U;
  const int32_t* d_hitcnt = (const int32_t*)ctx->scratch[SB_PAIRPIX].p;
  const int32_t* d_hitoff = d_hitcnt + U;
  const unsigned g0 = (unsigned)((U + 255) / 256);
  hipLaunchKernelGGL(compact_count_kernel, dim3(g0), dim3(256), 0, st, d_hitcnt, (const int64_t*)ctx->scratch[SB_TPM].p,
                     (const int32_t*)ctx->scratch[SB_UBATCH].p, M, U, c_hp, c_trk, c_frac);
  HIPCHK(hipGetLastError());
  if ((rc = sort_exclusive_scan_i32(ctx, c_hp, o_hp, U))) return rc;
  if ((rc = sort_exclusive_scan_i32(ctx, c_trk, o_trk, U))) return rc;
  if ((rc = sort_exclusive_scan_i32(ctx, c_frac, o_frac, U))) return rc;
  int32_t last_c[3], last_o[3];
  HIPCHK(hipMemcpyAsync(&last_c[0], c_hp + (U - 1), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&last_c[1], c_trk + (U - 1), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&last_c[2], c_frac + (U - 1), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&last_o[0], o_hp + (U - 1), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&last_o[1], o_trk + (U - 1), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&last_o[2], o_frac + (U - 1), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  const int64_t n_hp = (int64_t)last_c[0] + last_o[0], n_trk = (int64_t)last_c[1] + last_o[1];
  const int64_t n_frac = ctx->want_fractions ? (int64_t)last_c[2] + last_o[2] : 0, n_hits = ctx->chain_hits;
  // one block: [n_hp][5] i32 | trk_seg i64 | hit_charge f64 | frac_val f64   (each part 8-byte aligned)
  const size_t b_hp = ((size_t)n_hp * 20 + 7) & ~(size_t)7, b_trk = (size_t)n_trk * 8, b_chg = (size_t)n_hits * 8, b_frac = (size_t)n_frac * 8;
  if ((rc = ldsim_ensure(ctx, SB_CPO, b_hp + b_trk + b_chg + b_frac + 64))) return rc;
  char* base = (char*)ctx->scratch[SB_CPO].p;
  hipLaunchKernelGGL(compact_fill_kernel, dim3(g0), dim3(256), 0, st, d_hitcnt, d_hitoff, o_hp, o_trk, o_frac, c_trk,
                     (const int32_t*)ctx->scratch[SB_UPIX].p, (const int32_t*)ctx->scratch[SB_UBATCH].p,
                     (const int64_t*)ctx->scratch[SB_TPM].p, (const double*)ctx->scratch[SB_ADC].p,
                     ctx->want_fractions ? (const double*)ctx->scratch[SB_FRAC].p : nullptr, A, M, U, (int32_t*)base,
                     (int64_t*)(base + b_hp), (double*)(base + b_hp + b_trk), (double*)(base + b_hp + b_trk + b_chg));
  HIPCHK(hipGetLastError());
  ctx->cpt_n[0] = n_hp; ctx->cpt_n[1] = n_hits; ctx->cpt_n[2] = n_trk; ctx->cpt_n[3] = n_frac;
  for (int k = 0; k < 4; k++) sizes[k] = ctx->cpt_n[k];
  return 0;
}

// the compact arrays to host buffers sized from ldsim_chain_compact_build's sizes; any pointer may be NULL
extern "C" int ldsim_chain_compact_download(ldsim_ctx* ctx, int32_t* hit_pixels /* [n_hp][5] */, int64_t* track_segments,
                                            void* hit_rows /* [n_hits] 24-byte rows */, double* hit_charge, double* fractions) {
  LDSIM_ENTER(ctx);
  if (!ctx) { ldsim_set_error("null ctx"); return LDSIM_EINVAL; }
  if (ctx->cpt_gen != ctx->out_gen) {
    ldsim_set_error("ldsim_chain_compact_build has not run for the last chain launch");
    return LDSIM_ESTATE;
  }
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int64_t n_hp = ctx->cpt_n[0], n_hits = ctx->cpt_n[1], n_trk = ctx->cpt_n[2], n_frac = ctx->cpt_n[3];
  const size_t b_hp = ((size_t)n_hp * 20 + 7) & ~(size_t)7, b_trk = (size_t)n_trk * 8, b_chg = (size_t)n_hits * 8;
  const char* base = (const char*)ctx->scratch[SB_CPO].p;
  if (hit_pixels && n_hp) HIPCHK(hipMemcpyAsync(hit_pixels, base, (size_t)n_hp * 20, hipMemcpyDeviceToHost, st));
  if (track_segments && n_trk) HIPCHK(hipMemcpyAsync(track_segments, base + b_hp, b_trk, hipMemcpyDeviceToHost, st));
  if (hit_rows && n_hits) HIPCHK(hipMemcpyAsync(hit_rows, ctx->scratch[SB_HITS].p, (size_t)n_hits * 24, hipMemcpyDeviceToHost, st));
  if (hit_charge && n_hits) HIPCHK(hipMemcpyAsync(hit_charge, base + b_hp + b_trk, b_chg, hipMemcpyDeviceToHost, st));
  if (fractions && n_frac) HIPCHK(hipMemcpyAsync(fractions, base + b_hp + b_trk + b_chg, (size_t)n_frac * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}
