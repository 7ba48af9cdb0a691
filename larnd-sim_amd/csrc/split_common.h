// split_common.h -- records handed from the weights kernels (kernels_split.hip: per-sample weights_kernel,
// kernels_qweights.hip: quadrature qweights_kernel) to mac_kernel.
#pragma once
#include "current_common.h"

// items per pair: 512 at M = 1 (mac_kernel<1> keeps the list in LDS); at M = 2 a 64-shift chunk covers half as many
// slices (twice the runs, twice the items) and ndlar's pairs are heavier still -- mac_kernel<2> reads the list from HBM,
// so the capacity only costs scratch memory (32 KB per pair)
template <int M> struct ItemCap { static constexpr int value = M == 1 ? 512 : 2048; };
#define CMAX 192        // edge corrections per pair
#define RUNS_MAX 8      // sorted runs (slice chunks) per pair
#define HDR_INTS 24     // n_items, n_corr, it0, T, it_w0, it_w1, nruns, flags, run_start[9]
#define W_ARENA 3072
#define W_CELLS 512

struct Item {
  int32_t cell_nblk;   // cell | nblk << 16
  int32_t sbase;       // response shift of weight[0]:  k = M*it + sbase + u
  uint32_t woff_lo, woff_hi;
};
struct Corr {
  int32_t tick, pad;
  double val;
};

struct SplitArgs {
  CurArgs c;
  Item* items;            // [n_pairs][ItemCap<M>::value]
  int32_t* hdr;           // [n_pairs][HDR_INTS]
  Corr* corr;             // [n_pairs][CMAX]
  double* wbuf;           // weight arena
  unsigned long long wbuf_cap;   // doubles
  unsigned long long* cursor;    // bump allocator (doubles)
  // response rows with RESP_PAD zeros on either side and zeros outside the staged range [k_lo, k_hi] (mac_shift_kernel):
  // element k of cell c sits at resp_pad[c * nkp + RESP_PAD + k], so a row window needs no range checks
  const double* resp_pad;
  int32_t nkp, k_lo, k_hi;
};
#define RESP_PAD 1280       // >= M * WTILE + 64 + slack for M <= 2

