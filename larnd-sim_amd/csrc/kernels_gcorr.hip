// kernels_gcorr.hip -- a9-a12, node-separable form ("weights_mode" 2, gform.h): the correlation stage, and the host glue of
// the whole form.
//
// gcorr_kernel: one 128-thread workgroup per (segment, pixel) pair, on the tables gtables_kernel wrote.  Per 16-tick tile of
// response indices k a wave forms
//     G[n][k] = sum_cells X[n][col(cell)] Y[n][j(cell)] R[cell][k]                     (v_mfma_f64_16x16x4: nodes x cells x ticks)
// with the 16 quadrature nodes of the batch as the M rows, four cells per instruction as the contraction and the response
// row segments R[cell][k0 .. k0 + 15] streamed from L2 as the B operand (one 8-byte load per lane and instruction, from the
// zero-padded copy of the table: no range checks); then
//     P[u][k] = sum_n Z[n][u] G[n][k]                                                  (the same instruction: shifts x nodes x ticks)
// whose B operand IS the accumulator of the first product -- register r of lane l holds G[4r + l/16][k0 + l%16], which is
// B[kk = l/16][j = l%16] of contraction step r -- so G never leaves the registers; and the diagonal sum
//     out[(k - u_min - u) / M] += P[u][k]
// as ds_add_f64 into a tick array private to the wave (the four arrays are summed in wave order afterwards: results do not
// depend on timing).  A window edge where some slices are invalid (detsim.py:418-428) subtracts the same product with the
// Zi table at the one response index it concerns.  No DPP shifts, no 512-tick tile padding, no weight pool: the matrix pipe
// carries 16 x cells x ticks FMAs per pair where the shifted-window kernels issue (cells x shifts) x 512.
#include "gform.h"
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
#define GPF 4            // cell groups whose response loads are in flight ahead of the products of a wave (x 2 tiles)
static_assert(G_CELLPAD % (4 * GPF) == 0 && GPF % 4 == 0, "the cell list is padded to whole prefetch rounds");
#define GSTG 6           // 16-byte registers per lane that carry the next image while the current one finishes (6 KB per wave)
#define GZR 12           // Z entries per lane the P step keeps in registers (M = 1: up to 48 shifts x 16 nodes), fetched before the G loop

// One work item of the correlation: a pair that has tables, with everything the kernel needs to size and find its record --
// written in launch order by gwork_kernel after the tables stage (GInfo.emask and the fallback flag are the tables stage's).
enum { GW_PAIR = 0, GW_NCOL, GW_NJ, GW_UMIN, GW_NU, GW_EBOUND, GW_NB, GW_NQ, GW_IT0, GW_T, GW_ITW0, GW_ITW1, GW_EMASK, GW_LIVE,
       GW_OFF_LO, GW_OFF_HI, GW_WORDS };
struct GWork {
  int32_t v[GW_WORDS];
};
static_assert(sizeof(GWork) == 64, "GWork layout");

// LDS of a wave (doubles): its tick array | the column G_n[edge_k] | the image of one node batch (gform.h)
__host__ __device__ __forceinline__ int g_arena_doubles(int TT, int img_cap_d) { return TT + G_NODES + img_cap_d; }

// gcorr_kernel -- persistent: a wave is resident once and walks the work items w, w + W, ... of its launch; it owns a pair
// completely (every tick tile pair, the P steps, the edge columns, the store), so there is no workgroup barrier, no second tick
// array to add and no idle wave when a pair has an odd number of tile pairs.  What a pair waits for before its first product --
// its work item, its record -- is requested while the previous pair is still being multiplied: the work item one pair ahead
// (scalar loads), the first GSTG KB of the next image into registers right after the last G loop of the current image (whose
// tables are dead from there on; the P step and the store run on the accumulators and the tick array), written to LDS when the
// current pair is done.  [Round 3's kernel, a two-wave workgroup per pair, spent 2.0 of its 6.4 ms per 50 k segments in
// GInfo -> record -> LDS -> barrier chains with twelve pairs in flight per CU: DESIGN.md section 4.]
template <int M>
__global__ void __launch_bounds__(64, 3) gcorr_kernel(GArgs GA, int TT, int img_cap_d, const GWork* __restrict__ work, unsigned n_work) {
  const CurArgs& A = GA.c;
  const LdsimConsts* c = A.c;
  const int lane = threadIdx.x;
  const int kk = lane >> 4, jj = lane & 15;
  const unsigned W = gridDim.x;
  extern __shared__ double s_dyn[];
  double* const ow = s_dyn;                         // [TT]
  double* const s_gs = ow + TT;                     // [16]: the column G_n[edge_k]
  double* const s_img = s_gs + G_NODES;             // the image: counts | info | X | Y
  const int nkp = GA.nkp;
  (void)nkp;
  int edge_k[NEDGE], k_stage_lo, k_stage_hi;
  edge_ks(c, A, edge_k, k_stage_lo, k_stage_hi);
  unsigned long long n_mfma = 0, n_useful = 0;

  // ---- staging of an image: global -> registers (early) -> LDS (late); what does not fit the registers follows synchronously ---
  // (six named registers, not an array: the compiler kept an array in scratch memory)
  static_assert(GSTG == 6, "staging registers are spelled out");
  double2 stg0, stg1, stg2, stg3, stg4, stg5;
#define GSTG_EACH(X) X(0) X(1) X(2) X(3) X(4) X(5)
  auto stage_load = [&](const double* src, int doubles) {
    const double2* s2 = (const double2*)src;
    const int n2 = doubles >> 1;
#define GSTG_LD(r) { const int i = r * 64 + lane; stg##r = s2[i < n2 ? i : n2 - 1]; }      /* (what lies past the image is never written to LDS) */
    GSTG_EACH(GSTG_LD)
#undef GSTG_LD
  };
  auto stage_write = [&](const double* src, int doubles) {
    double2* d2 = (double2*)s_img;
    const double2* s2 = (const double2*)src;
    const int n2 = doubles >> 1;
    wsync();                                   // (every read of the image this one replaces has been issued)
#define GSTG_ST(r) { const int i = r * 64 + lane; if (i < n2) d2[i] = stg##r; }
    GSTG_EACH(GSTG_ST)
#undef GSTG_ST
    for (int i = GSTG * 64 + lane; i < n2; i += 64) d2[i] = s2[i];
    wsync();
  };
  // a work item is wave-uniform: its sixteen words are kept in scalar registers
  auto load_work = [&](unsigned i, int (&w)[GW_WORDS]) {
    const int4* src = (const int4*)(work + i);
#pragma unroll
    for (int k = 0; k < GW_WORDS / 4; k++) {
      const int4 q = src[k];
      w[4 * k] = __builtin_amdgcn_readfirstlane(q.x);
      w[4 * k + 1] = __builtin_amdgcn_readfirstlane(q.y);
      w[4 * k + 2] = __builtin_amdgcn_readfirstlane(q.z);
      w[4 * k + 3] = __builtin_amdgcn_readfirstlane(q.w);
    }
  };
  auto work_off = [](const int (&w)[GW_WORDS]) {
    return ((unsigned long long)(unsigned)w[GW_OFF_HI] << 32) | (unsigned long long)(unsigned)w[GW_OFF_LO];
  };
  unsigned item = blockIdx.x;
  if (item >= n_work) return;
  int wc[GW_WORDS], wn[GW_WORDS];
  load_work(item, wc);
  {
    const GImg I0 = g_img(wc[GW_NCOL], wc[GW_NJ]);
    if (wc[GW_LIVE]) {
      stage_load(GA.rec + work_off(wc), I0.doubles);
      stage_write(GA.rec + work_off(wc), I0.doubles);
    }
  }
  for (; item < n_work; item += W) {
    const bool has_next = item + W < n_work;
    load_work(has_next ? item + W : item, wn);
    const bool next_live = has_next && wn[GW_LIVE];
    const GImg In = g_img(wn[GW_NCOL], wn[GW_NJ]);
    const double* next_src = GA.rec + work_off(wn);       // the next pair's first image
    if (!wc[GW_LIVE]) {           // (flagged by the tables stage: the monolithic kernel writes this pair)
      if (next_live) {
        stage_load(next_src, In.doubles);
        stage_write(next_src, In.doubles);
      }
#pragma unroll
      for (int k = 0; k < GW_WORDS; k++) wc[k] = wn[k];
      continue;
    }
    const int pair = wc[GW_PAIR], ncol = wc[GW_NCOL], NJ = wc[GW_NJ], u_min = wc[GW_UMIN], NU = wc[GW_NU], ebound = wc[GW_EBOUND];
    const int NB = wc[GW_NB], NQ = wc[GW_NQ], emask = wc[GW_EMASK], it0 = wc[GW_IT0], T = wc[GW_T], it_w0 = wc[GW_ITW0], it_w1 = wc[GW_ITW1];
    const GImg I = g_img(ncol, NJ);
    const double* __restrict__ rec = GA.rec + work_off(wc);
    float* out = A.out + (int64_t)pair * (int64_t)A.T;
    const int NUr = g_nur(NU), NU16 = NUr >> 4;
    const int xs = I.xs, ys = I.ys;
    const unsigned long long batch_d = g_batch_doubles(ncol, NJ, NU, ebound);
    const int32_t* s_cnt = (const int32_t*)s_img;
    const unsigned* s_inf32 = (const unsigned*)(s_img + G_CELL0 / 2);
    const char* xl = (const char*)(s_img + I.x_d + jj * xs);
    const char* yl = (const char*)(s_img + I.y_d + jj * ys);
    bool next_staged = false;      // the next pair's image sits in the staging registers
    int resident = 0;              // node batch whose image the LDS holds

    for (int sup0 = it_w0; sup0 < it_w1; sup0 += TT) {
      const int wlen = min(it_w1 - sup0, TT);
      const bool last_sup = sup0 + TT >= it_w1;
      for (int i = lane; i < TT; i += 64) ow[i] = 0;
      // response indices this tick tile can meet, inside the staged range (zeros outside it)
      const int kA = max(M * sup0 + u_min, GA.k_lo), kB = min(M * (sup0 + wlen - 1) + u_min + NU - 1, GA.k_hi);
      const int n32 = kB >= kA ? (kB - kA) / 32 + 1 : 0;
      for (int b = 0; b < NB && n32 > 0; b++) {
        const double* brec = rec + G_HDR / 2 + (unsigned long long)b * batch_d;
        const int rows = g_rows(NQ, b);                // node rows the record keeps of this batch's Z tables
        const double* gZ = brec + I.doubles;
        if (resident != b) {                           // (a pair of several batches or tick tiles: its other images come in here)
          wsync();
          stage_load(brec, I.doubles);
          stage_write(brec, I.doubles);
          resident = b;
        }
        const int ncell = s_cnt[0];
        const int ngrp = ncell >> 2;                   // a multiple of GPF (the tables stage pads the list with weightless cells)
        if (lane == 0)
          n_useful += (unsigned long long)min(G_NODES, NQ - b * G_NODES) *
                      ((unsigned long long)s_cnt[1] * (unsigned long long)(kB - kA + 1) + (unsigned long long)NU * (unsigned long long)wlen);
        const int nq4 = rows >> 2;                     // node groups of four the batch has (its last batch: often one)
        const bool last_img = last_sup && b == NB - 1; // after this image's last G loop the tables in LDS are dead
        // Z of the lane's shifts and nodes for the P step, M = 1: fetched before the G loop, used after it
        const bool z_regs = false;   // (tried: +76 VGPRs as the compiler schedules it)
        double zr[GZR] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        auto z_fetch = [&]() {
#pragma unroll
          for (int st = 0; st < GZR / 4; st++)
#pragma unroll
            for (int q = 0; q < 4; q++)
              zr[4 * st + q] = (st < NU16 && q < nq4) ? gZ[(4 * q + kk) * NUr + 16 * st + jj] : 0.0;
        };
        // P[u][k] = sum_n Z[n][u] G[n][k]; out[(k - u_min - u) / M] += P[u][k] -- for both tiles of the wave at once: one A
        // operand (Z of the lane's shift and node) feeds two independent accumulation chains.
        auto pstep2 = [&](const d4& g0acc, const d4& g1acc, int k0) {
          if constexpr (M == 2) {
            // M = 2: out[(k - u_min - u) / 2] takes P[u][k] only where k - u_min - u is even.  The wave's two tiles hold the even and
            // the odd response indices of its 32 (below): each meets the shifts of one parity only, so the products run over
            // 16 shifts of that parity at a time -- none is computed to be dropped.
            const int par0 = (k0 - u_min) & 1, par1 = par0 ^ 1;
            const int NV16 = (((NU + 1) >> 1) + 15) >> 4;
            for (int st = 0; st < NV16; st++) {
              const int ue = 2 * (16 * st + jj);
              const bool in0 = ue + par0 < NUr, in1 = ue + par1 < NUr;
              auto zpair = [&](int q, double& a0, double& a1) {
                const int n = 4 * q + kk;
                a0 = in0 ? gZ[n * NUr + ue + par0] : 0.0;
                a1 = in1 ? gZ[n * NUr + ue + par1] : 0.0;
              };
              d4 p0 = {0, 0, 0, 0}, p1 = {0, 0, 0, 0};
              double a0[4], a1[4];
#pragma unroll
              for (int q = 0; q < 4; q++) {
                a0[q] = a1[q] = 0.0;
                if (q < nq4) zpair(q, a0[q], a1[q]);
              }
#pragma unroll
              for (int q = 0; q < 4; q++) {
                if (q < nq4) {
                  p0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[q], g0acc[q], p0, 0, 0, 0);
                  p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[q], g1acc[q], p1, 0, 0, 0);
                }
              }
#pragma unroll
              for (int r = 0; r < 4; r++) {
                const int v = 16 * st + 4 * r + kk;
                const int u0 = 2 * v + par0, u1 = 2 * v + par1;
                const int idx0 = ((k0 + 2 * jj - (u_min + u0)) >> 1) - sup0;          // (even numerators by construction)
                const int idx1 = ((k0 + 2 * jj + 1 - (u_min + u1)) >> 1) - sup0;
                if (u0 < NU && idx0 >= 0 && idx0 < wlen) atomicAdd(&ow[idx0], p0[r]);
                if (u1 < NU && idx1 >= 0 && idx1 < wlen) atomicAdd(&ow[idx1], p1[r]);
              }
            }
            n_mfma += 2 * nq4 * NV16;
            return;
          }
          for (int st = 0; st < NU16; st++) {
            double za[4];
            if (z_regs) {
#pragma unroll
              for (int q = 0; q < 4; q++) za[q] = st == 0 ? zr[q] : (st == 1 ? zr[4 + q] : zr[8 + q]);
            } else {
#pragma unroll
              for (int q = 0; q < 4; q++) za[q] = q < nq4 ? gZ[(4 * q + kk) * NUr + 16 * st + jj] : 0.0;
            }
            d4 p0 = {0, 0, 0, 0}, p1 = {0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < 4; q++) {
              if (q < nq4) {
                p0 = __builtin_amdgcn_mfma_f64_16x16x4f64(za[q], g0acc[q], p0, 0, 0, 0);
                p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(za[q], g1acc[q], p1, 0, 0, 0);
              }
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const int u = 16 * st + 4 * r + kk;
              const int idx = k0 + jj - (u_min + u) - sup0;
              const int idx1 = idx + 16;
              if (u < NU && idx >= 0 && idx < wlen) atomicAdd(&ow[idx], p0[r]);
              if (u < NU && idx1 >= 0 && idx1 < wlen) atomicAdd(&ow[idx1], p1[r]);
            }
          }
          n_mfma += 2 * nq4 * NU16;
        };
        // The wave takes the pairs of adjacent 16-tick tiles one after the other (the second tile may lie past the range: its
        // products meet zeros or ticks outside the window and are dropped): one A operand (X Y of the lane's node and cell) feeds
        // both products.  Two LDS words per (group, lane) name the cell: its response row for the B operands, its X column and Y
        // row for the A operand (a padding cell: a listed cell's row, valid memory, and the zero column of X).  A software
        // pipeline over the groups t of a tile pair, one slot per group: the products of group t; the 2 response loads of
        // group t + GPF into the registers just consumed; the row word of group t + GPF + 1; X and Y of group t + 1; the column
        // word of group t + 2 (past the last group the indices wrap: valid, unused).  Every load is issued one slot (LDS) or GPF
        // (L2) before its use, no branch, and sched_barrier keeps the compiler from sinking them back to their uses.
        const bool run_tiles = ngrp > 0 && !(A.debug_phases & 0x100000);
        auto wrap = [&](int g) {            // (g < 3 ngrp)
          g = g >= ngrp ? g - ngrp : g;
          return g >= ngrp ? g - ngrp : g;
        };
        auto row_word = [&](int g) { return s_inf32[2 * (4 * wrap(g) + kk)]; };
        auto col_word = [&](int g) { return s_inf32[2 * (4 * wrap(g) + kk) + 1]; };
        if (z_regs && run_tiles) z_fetch();
        for (int kt = 0; kt < n32; kt++) {
          const int k0 = kA + 32 * kt;
          d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
          if (run_tiles) {
            // (M = 2: tile 0 holds the even, tile 1 the odd indices of the 32 -- lane jj loads k0 + 2 jj and k0 + 2 jj + 1, adjacent)
            const double* rpl = GA.resp_pad + RESP_PAD + k0 + (M == 2 ? 2 * jj : jj);
            constexpr int B1 = M == 2 ? 1 : 16;
            double b0[GPF], b1[GPF], xv[2], yv[2];
            unsigned lo[2], hi[2];
#pragma unroll
            for (int u = 0; u < GPF; u++) {
              const double* q = rpl + row_word(u);
              b0[u] = q[0];
              b1[u] = q[B1];
            }
            lo[0] = row_word(GPF);
            hi[0] = col_word(0);
            hi[1] = col_word(1);
            xv[0] = *(const double*)(xl + (hi[0] & 0xFFFFu));
            yv[0] = *(const double*)(yl + (hi[0] >> 16));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
            for (int g0 = 0; g0 < ngrp; g0 += GPF) {
#pragma unroll
              for (int u = 0; u < GPF; u++) {
                const double a = xv[u & 1] * yv[u & 1];
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0[u], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1[u], acc1, 0, 0, 0);
                const double* q = rpl + lo[u & 1];
                b0[u] = q[0];
                b1[u] = q[B1];
                lo[(u + 1) & 1] = row_word(g0 + u + GPF + 1);
                xv[(u + 1) & 1] = *(const double*)(xl + (hi[(u + 1) & 1] & 0xFFFFu));
                yv[(u + 1) & 1] = *(const double*)(yl + (hi[(u + 1) & 1] >> 16));
                hi[u & 1] = col_word(g0 + u + 2);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
            n_mfma += 2 * ngrp;
          }
          if (last_img && kt == n32 - 1 && next_live && !(GA.dbg & 4)) {      // the tables are dead: the next pair's image sets out
            stage_load(next_src, In.doubles);
            next_staged = true;
          }
          if (run_tiles && !(A.debug_phases & 0x200000)) {
            pstep2(acc0, acc1, k0);
            // window edges: the share of the slices that are not valid at response index edge_k[e] comes off the tick it maps to,
            // sum_n Zi_e[n][u] G_n[edge_k[e]] per shift u.  The column of G sits in four lanes of this wave's accumulators
            // (register r of lane 16 q + col = G[4 r + q][k0 + col]): through LDS to all lanes, one shift per lane.
            if (emask && !(GA.dbg & 2)) {
              int et = 0;
              for (int e = 0; e < NEDGE; e++) {
                if (!(ebound & (1 << e))) continue;
                const double* gZi = gZ + (unsigned long long)rows * NUr * (unsigned long long)(1 + et);
                et++;
                const int ke = edge_k[e];
                if (!(emask & (1 << e)) || ke < k0 || ke >= k0 + 32) continue;
                const bool upper = M == 2 ? ((ke - k0) & 1) != 0 : ke >= k0 + 16;
                if (jj == (M == 2 ? (ke - k0) >> 1 : ((ke - k0) & 15))) {
#pragma unroll
                  for (int r = 0; r < 4; r++) s_gs[4 * r + kk] = upper ? acc1[r] : acc0[r];
                }
                wsync();
                for (int u = lane; u < NU; u += 64) {
                  double cv = 0;
#pragma unroll 1
                  for (int n0 = 0; n0 < rows; n0 += 4) {          // (four loads in flight)
                    double zv[4];
#pragma unroll
                    for (int n = 0; n < 4; n++) zv[n] = gZi[(n0 + n) * NUr + u];
#pragma unroll
                    for (int n = 0; n < 4; n++) cv = fma(zv[n], s_gs[n0 + n], cv);
                  }
                  const int num = ke - (u_min + u);
                  const int idx = (M == 1 ? num : (num >> 1)) - sup0;
                  if ((M == 1 || (num & 1) == 0) && idx >= 0 && idx < wlen) atomicAdd(&ow[idx], -cv);
                }
                wsync();
              }
            }
          }
        }
      }
      wsync();
      for (int i = lane; i < wlen; i += 64) {
        const int it = sup0 + i;
        if (it < A.T && !(GA.dbg & 8)) out[it] = (it >= it0 && it < T) ? (float)ow[i] : 0.f;
      }
      wsync();
    }
    if (A.win) {
      if (lane == 0) { A.win[2 * pair] = min(it_w0, A.T); A.win[2 * pair + 1] = min(it_w1, A.T); }
    } else if (!(A.debug_phases & 0x10000000)) {    // (timing tools)
      for (int it = lane; it < A.T; it += 64)
        if (it < it_w0 || it >= it_w1) out[it] = 0.f;
    }
    // the next pair's first image: from the staging registers (or, when this pair had no tile to multiply, straight away)
    if (next_live) {
      if (!next_staged) stage_load(next_src, In.doubles);
      stage_write(next_src, In.doubles);
    }
#pragma unroll
    for (int k = 0; k < GW_WORDS; k++) wc[k] = wn[k];
  }
  if (GA.dbg & 1) return;
  if (lane == 0 && n_mfma) stat_add(A.counters, 5, n_mfma * 1024ull);
  if (lane == 0 && n_useful) stat_add(A.counters, 8, n_useful);
}

// What the persistent kernel does not touch: pairs without tables (nothing to emit) and pairs the tables stage handed to the
// monolithic kernel (flagged: it writes them in full).
__global__ void __launch_bounds__(256) gcorr_idle_kernel(GArgs GA) {
  const CurArgs& A = GA.c;
  if (A.win) {
    const int64_t pair = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (pair >= A.n_pairs) return;
    if (GA.flags[pair]) { A.win[2 * pair] = 0; A.win[2 * pair + 1] = A.T; }
    else if (GA.gi[pair].status != 1) { A.win[2 * pair] = 0; A.win[2 * pair + 1] = 0; }
    return;
  }
  // complete rows: four pairs per workgroup, a wave each
  const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= A.n_pairs || GA.flags[pair] || GA.gi[pair].status == 1) return;
  float* out = A.out + pair * (int64_t)A.T;
  for (int it = threadIdx.x & 63; it < A.T; it += 64) out[it] = 0.f;
}

// launch class of a pair by the size of its image: 0 = fits the staging registers' 6 KB (sixteen waves per CU), 1 = 14 KB,
// 2 = the largest image the caps allow
__host__ __device__ __forceinline__ int g_img_class(int ncol, int NJ, int cap0_d, int cap1_d) {
  const int d = g_img(ncol, NJ).doubles;
  return d <= cap0_d ? 0 : (d <= cap1_d ? 1 : 2);
}

// the pairs with tables, by launch class: lists[c][..] (one atomic per wave and class)
__global__ void __launch_bounds__(256) gclass_list_kernel(const GInfo* __restrict__ gi, int64_t n, int cap0_d, int cap1_d,
                                                          int32_t* __restrict__ lists /* [3][n] */,
                                                          unsigned long long* __restrict__ counts /* [3] */) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  int cls = -1;
  if (i < n && gi[i].status == 1) cls = g_img_class(gi[i].ncol, gi[i].NJ, cap0_d, cap1_d);
  for (int c = 0; c < 3; c++) {
    const unsigned long long m = __ballot(cls == c);
    if (!m) continue;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(&counts[c], (unsigned long long)__popcll(m));
    base = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((int)base);
    if (cls == c) lists[(int64_t)c * n + (int64_t)base + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)i;
  }
}

// work items of a class in ascending pair order of ... the list's order (after the tables stage: emask and the flag are its)
__global__ void __launch_bounds__(256) gwork_kernel(const GInfo* __restrict__ gi, const int32_t* __restrict__ flags,
                                                    const int32_t* __restrict__ list, int64_t n_list, GWork* __restrict__ work) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_list) return;
  const int32_t pair = list[i];
  const GInfo g = gi[pair];
  GWork w;
  w.v[GW_PAIR] = pair; w.v[GW_NCOL] = g.ncol; w.v[GW_NJ] = g.NJ; w.v[GW_UMIN] = g.u_min; w.v[GW_NU] = g.NU; w.v[GW_EBOUND] = g.edge_bound;
  w.v[GW_NB] = g.NB; w.v[GW_NQ] = g.NQ; w.v[GW_IT0] = g.it0; w.v[GW_T] = g.T; w.v[GW_ITW0] = g.it_w0; w.v[GW_ITW1] = g.it_w1;
  w.v[GW_EMASK] = g.emask;
  w.v[GW_LIVE] = flags[pair] ? 0 : 1;
  w.v[GW_OFF_LO] = (int32_t)(unsigned)(g.off & 0xFFFFFFFFull);
  w.v[GW_OFF_HI] = (int32_t)(unsigned)(g.off >> 32);
  work[i] = w;
}

// ---- record offsets: exclusive scan of the sizes pair_setup_kernel wrote ------------------------------------------------------------
__global__ void __launch_bounds__(256) gsize_gather_kernel(const GInfo* __restrict__ gi, int64_t n, unsigned long long* __restrict__ sz) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) sz[i] = gi[i].size;
}
__global__ void __launch_bounds__(256) goff_scatter_kernel(GInfo* __restrict__ gi, int64_t n, const unsigned long long* __restrict__ off,
                                                           unsigned long long* __restrict__ total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    gi[i].off = off[i];
    if (i == n - 1) *total = off[i] + gi[i].size;
  }
}

int sort_exclusive_scan_u64(ldsim_ctx*, const unsigned long long*, unsigned long long*, int64_t);
extern "C++" int gtables_launch(ldsim_ctx* ctx, const GArgs& GA, int M, const int32_t* wg_list, int64_t n_wg, const int32_t* w2_list,
                                int64_t n_w2);
extern "C++" int gtables_list_launch(ldsim_ctx* ctx, const GArgs& GA, int32_t* wg_list, unsigned long long* wg_count, int32_t* w2_list,
                                     unsigned long long* w2_count);
extern "C++" int resp_pad_ensure(ldsim_ctx* ctx, const CurArgs& A, int* k_lo, int* k_hi, int* nkp);

// M of the form for these constants, 0 = configuration not covered (caller uses the monolithic kernel)
extern "C++" int gform_M(const ldsim_ctx* ctx, const CurArgs& args) {
  const LdsimConsts& h = ctx->h_consts;
  const double ratio = h.time_sampling / h.response_sampling;
  const int M = (int)llround(ratio);
  if (M < 1 || M > 2 || fabs(ratio - M) > 1e-9 || h.sampled_points > NS_MAX || args.nj > NJ_MAX || args.ni > 64 ||
      args.ni * args.nj > 65535 || args.n_pairs > 0x7fffffffLL)
    return 0;
  return M;
}

// tables + correlation of all pairs of `a`; *flags = device array [n_pairs] of the pairs left to the monolithic kernel.
// Returns 0 = done, 1 = configuration not covered, < 0 = error.  One host sync (the size of the record pool).
// the pairs flagged for the monolithic kernel, as a list (none in most launches: a workgroup per pair to look would cost 0.5 ms)
__global__ void __launch_bounds__(256) gflag_list_kernel(const int32_t* __restrict__ flags, int64_t n, int32_t* __restrict__ list,
                                                        unsigned long long* __restrict__ count) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n && flags[i]) list[atomicAdd(count, 1ull)] = (int32_t)i;
}

extern "C++" int gform_launch(ldsim_ctx* ctx, const CurArgs& a, unsigned long long* counters, int32_t** flags_out,
                              const int32_t** flag_list, const unsigned long long** flag_count) {
  const int M = gform_M(ctx, a);
  if (!M) return 1;
  const int64_t n = a.n_pairs;
  hipStream_t st = ctx->stream;
  int rc;
  if ((rc = ldsim_ensure(ctx, SB_PPAR, qpair_params_bytes(n)))) return rc;
  if ((rc = ldsim_ensure(ctx, SB_HDR, (size_t)n * sizeof(GInfo)))) return rc;
  if ((rc = ldsim_ensure(ctx, SB_ITEMS, (size_t)(n + 16) * 4))) return rc;                    // flags
  if ((rc = ldsim_ensure(ctx, SB_CORR, (size_t)(2 * n + 8) * 8 + (size_t)(6 * n + 2) * 4))) return rc;      // sizes | offsets | total, 3 class counts, n_wg, n_flagged, n_w2 | 3 class lists | wg list | flagged list | wide-wave list
  if ((rc = ldsim_ensure(ctx, SB_GWORK, (size_t)(n + 1) * sizeof(GWork)))) return rc;
  SplitArgs S{};
  S.c = a;
  GInfo* gi = (GInfo*)ctx->scratch[SB_HDR].p;
  if ((rc = qpair_setup_launch(ctx, S, M, ctx->scratch[SB_PPAR].p, gi))) return rc;
  unsigned long long* d_sz = (unsigned long long*)ctx->scratch[SB_CORR].p;
  unsigned long long* d_off = d_sz + n;
  unsigned long long* d_total = d_off + n;         // [0] pool doubles, [1..3] class counts, [4] n_wg, [5] n_flagged, [6] n_w2
  const unsigned g0 = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(gsize_gather_kernel, dim3(g0), dim3(256), 0, st, gi, n, d_sz);
  if ((rc = sort_exclusive_scan_u64(ctx, d_sz, d_off, n))) return rc;
  hipLaunchKernelGGL(goff_scatter_kernel, dim3(g0), dim3(256), 0, st, gi, n, d_off, d_total);
  HIPCHK(hipGetLastError());
  GArgs GA{};
  if ((rc = resp_pad_ensure(ctx, a, &GA.k_lo, &GA.k_hi, &GA.nkp))) return rc;
  // ticks per tile of the correlation: the pairs' windows are as long as the staged response support (over M) plus their
  // shift range; one tile where that fits 128 or 256 ticks, 512-tick tiles for a table with full support
  const int support = (GA.k_hi - GA.k_lo + 1) / M + 64;
  const int TT = support <= 128 ? 128 : (support <= 256 ? 256 : 512);
  // Launches of the correlation by the size of a pair's image (g_img_class): what fits the staging registers (6 KB: sixteen
  // waves per CU at the kernel's 128 VGPRs) in the first, 14 KB images in the second, the largest the caps allow in the third.
  const int cap_d[3] = {GSTG * 128, ctx->debug_lds_b1_kb > 0 ? ctx->debug_lds_b1_kb * 128 : 1792, g_img(G_NCOL, NJ_MAX).doubles};
  int32_t* d_cls = (int32_t*)(d_total + 8);        // [3][n]
  int32_t* d_wg = d_cls + 3 * n;                   // the pairs the tables stage gives to its workgroup kernel
  HIPCHK(hipMemsetAsync(d_total + 1, 0, 56, st));
  hipLaunchKernelGGL(gclass_list_kernel, dim3(g0), dim3(256), 0, st, gi, n, cap_d[0], cap_d[1], d_cls, d_total + 1);
  HIPCHK(hipGetLastError());
  GA.gi = gi;
  GA.dbg = (ctx->debug_gform & ~64) | (ctx->gform_wave_tables ? 0 : 64);
  GA.c.n_pairs = n;
  int32_t* d_w2 = d_wg + 2 * n;                    // (d_wg + n: the flagged list)
  if ((rc = gtables_list_launch(ctx, GA, d_wg, d_total + 4, d_w2, d_total + 6))) return rc;
  unsigned long long h_tot[7] = {0, 0, 0, 0, 0, 0, 0};
  HIPCHK(hipMemcpyAsync(h_tot, d_total, 56, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  const unsigned long long total = h_tot[0], n_wg = h_tot[4], n_w2 = h_tot[6];
  const unsigned long long n_cls[3] = {h_tot[1], h_tot[2], h_tot[3]};
  if (getenv("LDSIM_DEBUG_GFORM")) {      // class sizes of the launch, and why pairs miss the wave kernel
    std::vector<GInfo> h((size_t)n);
    HIPCHK(hipMemcpy(h.data(), gi, (size_t)n * sizeof(GInfo), hipMemcpyDeviceToHost));
    long n1 = 0, nu = 0, xy = 0, sl = 0, nb2 = 0;
    for (const GInfo& g : h) {
      if (g.status != 1) continue;
      n1++;
      nb2 += g.NB > 1;
      if (g.wave_ok) continue;
      if (g.NU > 2 * G_NUCAP) nu++;
      else if (g.ncol + g.NJ > 80) xy++;
      else sl++;
    }
    fprintf(stderr, "gform: %ld pairs, %ld with tables (%ld in 2+ node batches), workgroup tables kernel %llu (NU > 256: %ld, X | Y bins > 80: %ld, "
            "slices > 64: %ld), wide wave tables kernel %llu, correlation launches with images up to %d / %d / %d KB: %llu / %llu / %llu pairs, pool %.2f GB\n", (long)n, n1, nb2, n_wg, nu, xy, sl,
            n_w2, cap_d[0] >> 7, cap_d[1] >> 7, (cap_d[2] + 127) >> 7, n_cls[0], n_cls[1], n_cls[2], total * 8e-9);
  }
  if ((rc = ldsim_ensure(ctx, SB_WBUF, (size_t)(total + 16) * 8))) return rc;
  HIPCHK(hipMemsetAsync(&counters[7], 0, 8, st));
  GA.c = a;
  GA.pp = (const PairParams*)ctx->scratch[SB_PPAR].p;
  GA.gi = gi;
  GA.rec = (double*)ctx->scratch[SB_WBUF].p;
  GA.flags = (int32_t*)ctx->scratch[SB_ITEMS].p;
  GA.glx = ctx->d_glx;
  GA.glw = ctx->d_glw;
  GA.resp_pad = (const double*)ctx->resp_pad.p;
  if ((rc = gtables_launch(ctx, GA, M, d_wg, (int64_t)n_wg, d_w2, (int64_t)n_w2))) return rc;
  HIPCHK(hipEventRecord(ctx->ev[5], st));
  if (ctx->n_cu <= 0) {
    int v = 0;
    HIPCHK(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, ctx->device));
    ctx->n_cu = v > 0 ? v : 256;
  }
  GWork* d_work = (GWork*)ctx->scratch[SB_GWORK].p;
  for (int cls = 0; cls < 3; cls++) {
    if (n_cls[cls] == 0) continue;
    const int32_t* list = d_cls + (int64_t)cls * n;
    hipLaunchKernelGGL(gwork_kernel, dim3((unsigned)((n_cls[cls] + 255) / 256)), dim3(256), 0, st, gi, GA.flags, list, (int64_t)n_cls[cls], d_work);
    HIPCHK(hipGetLastError());
    d_work += n_cls[cls];
  }
  d_work = (GWork*)ctx->scratch[SB_GWORK].p;
  for (int cls = 0; cls < 3; cls++) {
    if (n_cls[cls] == 0) continue;
    const size_t dyn = (size_t)g_arena_doubles(TT, cap_d[cls]) * 8;
    // resident waves: what the kernel's registers and this arena allow per CU (the occupancy the runtime computes), never more than
    // there are work items
    int per_cu = 0;
    if (M == 1) HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gcorr_kernel<1>, 64, dyn));
    else HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gcorr_kernel<2>, 64, dyn));
    if (ctx->debug_lds_pad_kb > 0) per_cu = std::min(per_cu, ctx->debug_lds_pad_kb);      // timing tools: waves per CU
    if (per_cu < 1) per_cu = 1;
    const unsigned long long grid = std::min<unsigned long long>(n_cls[cls], (unsigned long long)per_cu * (unsigned long long)ctx->n_cu);
    if (M == 1) hipLaunchKernelGGL(gcorr_kernel<1>, dim3((unsigned)grid), dim3(64), dyn, st, GA, TT, cap_d[cls], d_work, (unsigned)n_cls[cls]);
    else hipLaunchKernelGGL(gcorr_kernel<2>, dim3((unsigned)grid), dim3(64), dyn, st, GA, TT, cap_d[cls], d_work, (unsigned)n_cls[cls]);
    HIPCHK(hipGetLastError());
    d_work += n_cls[cls];
  }
  hipLaunchKernelGGL(gcorr_idle_kernel, dim3(a.win ? g0 : (unsigned)((n + 3) / 4)), dim3(256), 0, st, GA);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(ctx->ev[6], st));
  // the pool's size in the statistics slot of the split paths (doubles)
  HIPCHK(hipMemcpyAsync(&counters[7], d_total, 8, hipMemcpyDeviceToDevice, st));
  *flags_out = GA.flags;
  hipLaunchKernelGGL(gflag_list_kernel, dim3(g0), dim3(256), 0, st, GA.flags, n, d_wg + n, d_total + 5);
  HIPCHK(hipGetLastError());
  *flag_list = d_wg + n;
  *flag_count = d_total + 5;
  return 0;
}
