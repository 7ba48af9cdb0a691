// kernels_macshift.hip -- mac_shift_kernel: the correlation stage of the split path at M = 1 (TIME_SAMPLING ==
// RESPONSE_SAMPLING) without staging response rows in LDS.
//
// mac_kernel<1> (kernels_split.hip) stages every item's row segment in wave-private LDS and lets each lane read its
// sliding window from there: 10 ds_write + 8 ds_read per lane and item plus 16 ds_read per 8-shift block.  Counted against
// the 128 B/clk of a CU's LDS that is as much time as the DFMAs of the four SIMDs take -- the kernel sat at 50 % of the
// DFMA peak because LDS bandwidth was the co-limiter, not issue.  Here a lane loads the 8 row elements under its own 8
// ticks straight from global memory (one coalesced 4 KB read per wave and item), and the window slides by passing the
// 8-element chunks from lane to lane with whole-wave DPP shifts (v_mov_b32 wave_shl:1, a GFX9 control gfx950 still has):
//     block b needs  chunk_b(L) ++ chunk_{b+1}(L),   chunk_c(L) = R[kb + 8(L + c) .. +7] = chunk_{c-1}(L + 1)
// so chunk_{b+2} = shift(chunk_{b+1}); lane 63 takes its next chunk from a 64-element tail every lane loaded one element
// of.  The first chunk of an item is prefetched into one of two register sets that swap roles from item to item (no copies).  LDS now only carries that tail (broadcast reads); the item's weights are wave-uniform and come through the
// scalar cache straight into the FMAs' SGPR operand.  Per block: 64 v_fma_f64 + 16 DPP moves.  Arithmetic and summation order are those of mac_kernel<1>: results are bit-identical (tested).
#include "split_common.h"

__device__ __forceinline__ double wave_shl1(double old, double src) {
  // lane L <- lane L+1; lane 63 keeps `old`
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), 0x130, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), 0x130, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

typedef const double __attribute__((address_space(4)))* wconst_ptr;   // constant address space: uniform reads go through s_load

__global__ void __launch_bounds__(CUR_THREADS, 5) mac_shift_kernel(SplitArgs S) {
  const CurArgs& A = S.c;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform, and known to be: item walk and weights stay scalar
  const int64_t pair = blockIdx.x;
  if (pair >= A.n_pairs) return;
  const int32_t* hdr = S.hdr + pair * HDR_INTS;
  if (hdr[7]) return;                               // overflowed: the monolithic kernel writes this pair
  float* out = A.out + pair * (int64_t)A.T;
  const int n_items = hdr[0], n_corr = hdr[1], it0 = hdr[2], T = hdr[3], it_w0 = hdr[4], it_w1 = hdr[5];
  if (n_items <= 0 || it_w1 <= it_w0) {
    for (int it = tid; it < A.T; it += CUR_THREADS) out[it] = 0.f;
    return;
  }
  __shared__ double s_out[TILE_TICKS];
  __shared__ double s_tail[NWAVE][64 + 8];          // wave-private: row elements past the tile, for lane 63
  __shared__ Item s_items[ItemCap<1>::value];
  const Item* __restrict__ gitems = S.items + pair * (int64_t)ItemCap<1>::value;
  for (int i = tid; i < n_items; i += CUR_THREADS) s_items[i] = gitems[i];
  __syncthreads();
  unsigned long long n_blocks = 0;

  for (int sup0 = it_w0; sup0 < it_w1; sup0 += TILE_TICKS) {
    const int wlen = min(it_w1 - sup0, TILE_TICKS);
    const int ntt = (wlen + WTILE - 1) / WTILE;
    int my_tile, share_rank, nshare;
    if (ntt >= 3) { my_tile = wv; share_rank = 0; nshare = 1; }
    else if (ntt == 2) { my_tile = wv >> 1; share_rank = wv & 1; nshare = 2; }
    else { my_tile = 0; share_rank = wv; nshare = 4; }
    const bool tile_live = my_tile < ntt;
    const int tb = sup0 + my_tile * WTILE;
    double acc[TPL];
#pragma unroll
    for (int j = 0; j < TPL; j++) acc[j] = 0;

    if (tile_live && !(A.debug_phases & 0x2000)) {            // 0x2000 (timing tools): prologue and epilogue only
      double* tl = s_tail[wv];
      // item descriptors are wave-uniform: kept in SGPRs, so the row / weight base addresses are scalar too
      auto descriptor = [&](int li) {
        Item d;
        d.cell_nblk = __builtin_amdgcn_readfirstlane(s_items[li].cell_nblk);
        d.sbase = __builtin_amdgcn_readfirstlane(s_items[li].sbase);
        d.woff_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_items[li].woff_lo);
        d.woff_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_items[li].woff_hi);
        return d;
      };
      // the lane's first chunk (row elements 8L .. 8L+7) and its element of the tail past the tile, from the zero-padded
      // rows: no range checks (an item whose window would leave the padding does not exist: the weights stage only emits
      // shifts that meet the response range, and the launcher checks the pad against the tile)
      auto fetch = [&](const Item& itx, double (&ca)[8], double& ct, double& touch) {
        const int nblk = (itx.cell_nblk >> 16) & 0xFF;
        // (0x4000, timing tools: every item reads cell 0's row -- what the kernel costs when the rows come from L1/L2)
        const int cell = (A.debug_phases & 0x4000) ? 0 : (itx.cell_nblk & 0xFFFF);
        int kb = tb + itx.sbase;
        // a window that starts outside these bounds lies wholly in the zero padding: clamping keeps it there and in memory
        kb = max(-RESP_PAD, min(kb, S.nkp - RESP_PAD - (WTILE + 64)));
        const double* src = S.resp_pad + (int64_t)cell * S.nkp + (RESP_PAD + kb);
#pragma unroll
        for (int q = 0; q < 8; q++) ca[q] = src[8 * lane + q];
        ct = (lane < nblk * 8) ? src[WTILE + lane] : 0.0;
        // The item's weights stream from HBM (the pool is many GB per launch): a scalar load that misses every cache takes
        // longer than a block of FMAs and cannot be waited on selectively.  One vector load per lane an item ahead pulls
        // the <= 512 bytes into L2, where the scalar loads of the blocks then hit.
        const unsigned long long wo = ((unsigned long long)itx.woff_hi << 32) | (unsigned long long)itx.woff_lo;
        touch = (lane < nblk * 8) ? S.wbuf[wo + lane] : 0.0;
      };
      // one 8-shift block on the window lo ++ hi, weights w[0..8) read through the scalar cache (wave-uniform address)
      auto fmas = [&](double (&lo)[8], double (&hi)[8], wconst_ptr w) {
#pragma unroll
        for (int du = 0; du < 8; du++) {
          const double av = w[du];
#pragma unroll
          for (int j = 0; j < TPL; j++) acc[j] = fma(av, (j + du < 8) ? lo[j + du] : hi[j + du - 8], acc[j]);
        }
      };
      // dst (a dead chunk's registers) becomes the chunk after hi: shifted in from the next lane, lane 63's from the tail.
      // Three register sets take the roles lo / hi / dst in rotation, so no chunk is ever copied.
      auto shift = [&](double (&dst)[8], double (&hi)[8], const double* tnext) {
#pragma unroll
        for (int q = 0; q < 8; q++) dst[q] = wave_shl1(tnext[q], hi[q]);
      };
      Item d_cur{}, d_next{};
      // two register sets: the item being correlated and the one being fetched swap roles every item (no copies)
      double a0[8], a1[8], t0 = 0, t1 = 0, h0 = 0, h1 = 0;
      auto run_item = [&](double (&a)[8], double& tcur, double& hcur, double (&na)[8], double& tnxt, double& hnxt, int li) {
        const int nblk = (d_cur.cell_nblk >> 16) & 0xFF;
        acc[0] = fma(hcur, 0.0, acc[0]);                       // keeps the touch load alive; adds +0.0
        const unsigned long long wo = ((unsigned long long)d_cur.woff_hi << 32) | (unsigned long long)d_cur.woff_lo;
        wconst_ptr w = (wconst_ptr)(S.wbuf + wo);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        tl[lane] = tcur;                                       // tl[l] = row element 512 + l
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (li + nshare < n_items) fetch(d_next, na, tnxt, hnxt);
        d_cur = d_next;
        if (li + 2 * nshare < n_items) d_next = descriptor(li + 2 * nshare);
        double y[8], z[8];
#pragma unroll
        for (int q = 0; q < 8; q++) y[q] = wave_shl1(tl[q], a[q]);
        // The <= 8 blocks of an item as straight-line code with forward exits instead of a loop: a loop's back edge made the
        // compiler copy every shifted chunk back into the registers the loop entered with (8 v_mov_b64 per block).
        // chunk c of lane 63 = tl[8(c-1) ..]; before block b the sets hold chunks b and b+1.
        if (!(A.debug_phases & 0x1000)) {                      // 0x1000 (timing tools): everything but the correlation blocks
          fmas(a, y, w);
          if (nblk > 1) {
            shift(z, y, tl + 8);  fmas(y, z, w + 8);
            if (nblk > 2) {
              shift(a, z, tl + 16);  fmas(z, a, w + 16);
              if (nblk > 3) {
                shift(y, a, tl + 24);  fmas(a, y, w + 24);
                if (nblk > 4) {
                  shift(z, y, tl + 32);  fmas(y, z, w + 32);
                  if (nblk > 5) {
                    shift(a, z, tl + 40);  fmas(z, a, w + 40);
                    if (nblk > 6) {
                      shift(y, a, tl + 48);  fmas(a, y, w + 48);
                      if (nblk > 7) {
                        shift(z, y, tl + 56);  fmas(y, z, w + 56);
                      }
                    }
                  }
                }
              }
            }
          }
        }
        n_blocks += nblk;
      };
      if (share_rank < n_items) {
        d_cur = descriptor(share_rank);
        fetch(d_cur, a0, t0, h0);
      }
      if (share_rank + nshare < n_items) d_next = descriptor(share_rank + nshare);
      for (int li = share_rank; li < n_items; li += 2 * nshare) {
        run_item(a0, t0, h0, a1, t1, h1, li);
        if (li + nshare < n_items) run_item(a1, t1, h1, a0, t0, h0, li + nshare);
      }
    }
    // ---- combine waves sharing a tile, window-edge corrections, mask, f32 store (as mac_kernel) -------------------------
    for (int rnk = 0; rnk < nshare; rnk++) {
      __syncthreads();
      if (tile_live && share_rank == rnk) {
#pragma unroll
        for (int j = 0; j < TPL; j++) {
          const int idx = my_tile * WTILE + TPL * lane + j;
          s_out[idx] = (rnk == 0) ? acc[j] : s_out[idx] + acc[j];
        }
      }
    }
    __syncthreads();
    {
      const Corr* cr = S.corr + pair * CMAX;
      for (int k = tid; k < n_corr; k += CUR_THREADS) {
        int i = cr[k].tick - sup0;
        if (i >= 0 && i < wlen) atomicAdd(&s_out[i], -cr[k].val);
      }
    }
    __syncthreads();
    for (int i = tid; i < wlen; i += CUR_THREADS) {
      int it = sup0 + i;
      if (it < A.T) out[it] = (it >= it0 && it < T) ? (float)s_out[i] : 0.f;
    }
    __syncthreads();
  }
  for (int it = tid; it < A.T; it += CUR_THREADS)
    if (it < it_w0 || it >= it_w1) out[it] = 0.f;
  if (lane == 0 && n_blocks) stat_add(A.counters, 5, n_blocks * 64ull * 64ull);
}

// ---- M = 2 (TIME_SAMPLING = 2 RESPONSE_SAMPLING, ndlar): tick t of lane L reads R[kb + 16L + 2j + u] -------------------------
// Chunks of 8 row elements c_k(L) = R[kb + 16L + 8k ..]; block b needs c_b ++ c_{b+1} ++ c_{b+2} and the next chunk is
// c_{b+3}(L) = c_{b+1}(L + 1): the same whole-wave shift, four register sets in rotation.  The first two chunks of an item
// are loaded (the lane's own 16 elements, one coalesced 8 KB read per wave), c_2 is the first shift.  Item descriptors are
// read through the scalar cache straight from HBM (the list is up to 32 KB per pair at M = 2, too much LDS).
typedef const Item __attribute__((address_space(4)))* iconst_ptr;

__global__ void __launch_bounds__(CUR_THREADS, 4) mac_shift2_kernel(SplitArgs S) {
  const CurArgs& A = S.c;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t pair = blockIdx.x;
  if (pair >= A.n_pairs) return;
  const int32_t* hdr = S.hdr + pair * HDR_INTS;
  if (hdr[7]) return;                               // overflowed: the monolithic kernel writes this pair
  float* out = A.out + pair * (int64_t)A.T;
  const int n_items = hdr[0], n_corr = hdr[1], it0 = hdr[2], T = hdr[3], it_w0 = hdr[4], it_w1 = hdr[5];
  if (n_items <= 0 || it_w1 <= it_w0) {
    for (int it = tid; it < A.T; it += CUR_THREADS) out[it] = 0.f;
    return;
  }
  __shared__ double s_out[TILE_TICKS];
  __shared__ double s_tail[NWAVE][64 + 8];          // wave-private: row elements past the tile, for lane 63
  iconst_ptr gitems = (iconst_ptr)(S.items + pair * (int64_t)ItemCap<2>::value);
  unsigned long long n_blocks = 0;

  for (int sup0 = it_w0; sup0 < it_w1; sup0 += TILE_TICKS) {
    const int wlen = min(it_w1 - sup0, TILE_TICKS);
    const int ntt = (wlen + WTILE - 1) / WTILE;
    int my_tile, share_rank, nshare;
    if (ntt >= 3) { my_tile = wv; share_rank = 0; nshare = 1; }
    else if (ntt == 2) { my_tile = wv >> 1; share_rank = wv & 1; nshare = 2; }
    else { my_tile = 0; share_rank = wv; nshare = 4; }
    const bool tile_live = my_tile < ntt;
    const int tb = sup0 + my_tile * WTILE;
    double acc[TPL];
#pragma unroll
    for (int j = 0; j < TPL; j++) acc[j] = 0;

    if (tile_live) {
      double* tl = s_tail[wv];
      auto descriptor = [&](int li) {
        Item d;
        d.cell_nblk = gitems[li].cell_nblk;
        d.sbase = gitems[li].sbase;
        d.woff_lo = gitems[li].woff_lo;
        d.woff_hi = gitems[li].woff_hi;
        return d;
      };
      auto fetch = [&](const Item& itx, double (&c0)[8], double (&c1)[8], double& ct) {
        const int nblk = (itx.cell_nblk >> 16) & 0xFF;
        int kb = 2 * tb + itx.sbase;
        kb = max(-RESP_PAD, min(kb, S.nkp - RESP_PAD - (2 * WTILE + 64)));       // see mac_shift_kernel
        const int cell = (A.debug_phases & 0x4000) ? 0 : (itx.cell_nblk & 0xFFFF);       // 0x4000: timing tools
        const double* src = S.resp_pad + (int64_t)cell * S.nkp + (RESP_PAD + kb);
#pragma unroll
        for (int q = 0; q < 8; q++) c0[q] = src[16 * lane + q];
#pragma unroll
        for (int q = 0; q < 8; q++) c1[q] = src[16 * lane + 8 + q];
        ct = (lane < nblk * 8) ? src[2 * WTILE + lane] : 0.0;
      };
      // one 8-shift block on the window lo ++ mid ++ hi (tick j, shift u reads element 2j + u); with `more`, dst becomes
      // the chunk after hi = the next lane's chunk `mid`
      auto block = [&](double (&lo)[8], double (&mid)[8], double (&hi)[8], double (&dst)[8], wconst_ptr w, const double* tnext,
                       bool more) {
#pragma unroll
        for (int du = 0; du < 8; du++) {
          const double av = w[du];
#pragma unroll
          for (int j = 0; j < TPL; j++) {
            const int e = 2 * j + du;
            acc[j] = fma(av, e < 8 ? lo[e] : (e < 16 ? mid[e - 8] : hi[e - 16]), acc[j]);
          }
        }
        if (more) {
#pragma unroll
          for (int q = 0; q < 8; q++) dst[q] = wave_shl1(tnext[q], mid[q]);
        }
      };
      Item d_cur{}, d_next{};
      // (no touch load of the weights here: every item is walked by all four waves, one per tile, so three of the four
      // scalar reads already hit L2 -- measured 13 % slower with it)
      double p0[8], p1[8], n0[8], n1[8], t0 = 0, t1 = 0;
      auto run_item = [&](double (&a)[8], double (&b)[8], double& tcur, double (&na)[8], double (&nb)[8], double& tnxt, int li) {
        const int nblk = (d_cur.cell_nblk >> 16) & 0xFF;
        const unsigned long long wo = ((unsigned long long)d_cur.woff_hi << 32) | (unsigned long long)d_cur.woff_lo;
        wconst_ptr w = (wconst_ptr)(S.wbuf + wo);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        tl[lane] = tcur;                                       // tl[l] = row element 1024 + l; chunk k of lane 63 = tl[8(k-2) ..]
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (li + nshare < n_items) fetch(d_next, na, nb, tnxt);
        d_cur = d_next;
        if (li + 2 * nshare < n_items) d_next = descriptor(li + 2 * nshare);
        double c[8], d[8];
#pragma unroll
        for (int q = 0; q < 8; q++) c[q] = wave_shl1(tl[q], a[q]);          // chunk 2 = the next lane's chunk 0
        // straight-line blocks with forward exits (see mac_shift_kernel); before block k the sets hold chunks k, k+1, k+2 and
        // block k shifts in chunk k + 3 = tl[8(k+1) ..] for lane 63
        block(a, b, c, d, w, tl + 8, nblk > 1);
        if (nblk > 1) {
          block(b, c, d, a, w + 8, tl + 16, nblk > 2);
          if (nblk > 2) {
            block(c, d, a, b, w + 16, tl + 24, nblk > 3);
            if (nblk > 3) {
              block(d, a, b, c, w + 24, tl + 32, nblk > 4);
              if (nblk > 4) {
                block(a, b, c, d, w + 32, tl + 40, nblk > 5);
                if (nblk > 5) {
                  block(b, c, d, a, w + 40, tl + 48, nblk > 6);
                  if (nblk > 6) {
                    block(c, d, a, b, w + 48, tl + 56, nblk > 7);
                    if (nblk > 7) block(d, a, b, c, w + 56, tl, false);
                  }
                }
              }
            }
          }
        }
        n_blocks += nblk;
      };
      if (share_rank < n_items) {
        d_cur = descriptor(share_rank);
        fetch(d_cur, p0, p1, t0);
      }
      if (share_rank + nshare < n_items) d_next = descriptor(share_rank + nshare);
      for (int li = share_rank; li < n_items; li += 2 * nshare) {
        run_item(p0, p1, t0, n0, n1, t1, li);
        if (li + nshare < n_items) run_item(n0, n1, t1, p0, p1, t0, li + nshare);
      }
    }
    // ---- combine waves sharing a tile, window-edge corrections, mask, f32 store (as mac_kernel) -------------------------
    for (int rnk = 0; rnk < nshare; rnk++) {
      __syncthreads();
      if (tile_live && share_rank == rnk) {
#pragma unroll
        for (int j = 0; j < TPL; j++) {
          const int idx = my_tile * WTILE + TPL * lane + j;
          s_out[idx] = (rnk == 0) ? acc[j] : s_out[idx] + acc[j];
        }
      }
    }
    __syncthreads();
    {
      const Corr* cr = S.corr + pair * CMAX;
      for (int k = tid; k < n_corr; k += CUR_THREADS) {
        int i = cr[k].tick - sup0;
        if (i >= 0 && i < wlen) atomicAdd(&s_out[i], -cr[k].val);
      }
    }
    __syncthreads();
    for (int i = tid; i < wlen; i += CUR_THREADS) {
      int it = sup0 + i;
      if (it < A.T) out[it] = (it >= it0 && it < T) ? (float)s_out[i] : 0.f;
    }
    __syncthreads();
  }
  for (int it = tid; it < A.T; it += CUR_THREADS)
    if (it < it_w0 || it >= it_w1) out[it] = 0.f;
  if (lane == 0 && n_blocks) stat_add(A.counters, 5, n_blocks * 64ull * 64ull);
}

// rows of the response table with RESP_PAD zeros in front and behind, and zeros where mac_kernel's staging would put them
// (outside [k_lo, k_hi]: the part of the table's support a tick of the window can meet)
__global__ void __launch_bounds__(256) pad_response_kernel(const double* __restrict__ resp, int64_t n_cells, int nk, int nkp,
                                                           int k_lo, int k_hi, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_cells * nkp) return;
  const int64_t c = i / nkp;
  const int k = (int)(i - c * nkp) - RESP_PAD;
  out[i] = (k >= k_lo && k <= k_hi && k >= 0 && k < nk) ? resp[c * nk + k] : 0.0;
}

// The zero-padded copy of the response rows (mac_shift kernels, gcorr_kernel): rebuilt when the table or the staged range --
// the part of the table's support a tick of a window can meet, incl. the partially valid edges (mac_kernel's staging,
// kernels_split.hip) -- changes.
extern "C++" int resp_pad_ensure(ldsim_ctx* ctx, const CurArgs& A, int* k_lo_out, int* k_hi_out, int* nkp_out) {
  const LdsimConsts& h = ctx->h_consts;
  const int k_lo = A.k_first > 0 ? A.k_first : 0;
  int k_hi;
  {
    const double V = h.time_window / h.response_sampling;
    int kn = (int)ceil(V + 0.5 + 1e-6);
    int ka = (int)floor(V - 0.5 - 1e-6);
    if ((double)ka + 0.5 >= V - 1e-6) ka--;
    int k_top = kn - 1;
    if (k_top - ka > NEDGE - 1) k_top = ka + NEDGE - 1;
    k_hi = k_top < A.nk - 1 ? k_top : A.nk - 1;
    k_hi = k_hi < A.k_last ? k_hi : A.k_last;
  }
  const int nkp = A.nk + 2 * RESP_PAD;
  const int64_t n_cells = (int64_t)A.ni * A.nj;
  if (ctx->resp_pad_hi == -2 || ctx->resp_pad_lo != k_lo || ctx->resp_pad_hi != k_hi ||
      ctx->resp_pad.bytes < (size_t)n_cells * nkp * 8) {
    int rc = ldsim_ensure_buf(ctx, &ctx->resp_pad, (size_t)n_cells * nkp * 8);
    if (rc) return rc;
    hipLaunchKernelGGL(pad_response_kernel, dim3((unsigned)((n_cells * nkp + 255) / 256)), dim3(256), 0, ctx->stream, A.resp,
                       n_cells, A.nk, nkp, k_lo, k_hi, (double*)ctx->resp_pad.p);
    HIPCHK(hipGetLastError());
    ctx->resp_pad_lo = k_lo;
    ctx->resp_pad_hi = k_hi;
  }
  *k_lo_out = k_lo; *k_hi_out = k_hi; *nkp_out = nkp;
  return 0;
}

extern "C++" int mac_shift_launch(ldsim_ctx* ctx, SplitArgs S, int M) {
  const CurArgs& A = S.c;
  int k_lo, k_hi, nkp;
  int rc = resp_pad_ensure(ctx, A, &k_lo, &k_hi, &nkp);
  if (rc) return rc;
  S.resp_pad = (const double*)ctx->resp_pad.p;
  S.nkp = nkp;
  S.k_lo = k_lo;
  S.k_hi = k_hi;
  if (M == 2) hipLaunchKernelGGL(mac_shift2_kernel, dim3((unsigned)S.c.n_pairs), dim3(CUR_THREADS), 0, ctx->stream, S);
  else hipLaunchKernelGGL(mac_shift_kernel, dim3((unsigned)S.c.n_pairs), dim3(CUR_THREADS), 0, ctx->stream, S);
  HIPCHK(hipGetLastError());
  return 0;
}
