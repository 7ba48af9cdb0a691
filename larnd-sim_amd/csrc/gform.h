// gform.h -- the node-separable form of a9-a12 (option "weights_mode" 2, default): records handed from gtables_kernel
// (kernels_gtables.hip) to gcorr_kernel (kernels_gcorr.hip).
//
// The quadrature along the segment (kernels_qweights.hip) makes the binned weights a sum over nodes of separable terms,
//      A[i][j][shift] = sum_n X_n[i] Y_n[j] Z_n[shift],
// so the waveform of a (segment, pixel) pair is
//      out[t] = sum_n sum_s Z_n[s] G_n[M t + s],        G_n[k] = sum_(i,j) X_n[i] Y_n[j] R[i][j][k]
// -- a [16 nodes] x [cells] x [response ticks] matrix product followed by a 16-row Toeplitz sum -- and the weights A are never
// formed: no weight pool, no item lists.  gtables_kernel writes X, Y, Z (a few KB per pair) and the list of response cells
// that can carry weight; gcorr_kernel runs the two products on the f64 matrix pipe (v_mfma_f64_16x16x4).
#pragma once
#include "qpair.h"

#define G_NODES 16          // quadrature nodes per batch = rows of the matrix product
#define G_NCOL 40           // distinct response columns i of a pair (<= SAMPLED_POINTS)
#define G_XS 41             // row strides of the LDS tables (odd: the 16 node rows fall on distinct banks)
#define G_YS 49
#define G_NUCAP 128         // response shifts of a pair whose Z table the kernels hold in LDS at once (more: in parts / from L2)
#define G_ZS (G_NUCAP + 16) // == 16 mod 32: the four 16-shift runs of an A operand read conflict-free
#define G_CELLCAP 2048      // cells of a batch held in LDS as 16-bit (col, j) codes: every pair fits (40 columns x 48 rows)
#define G_CELLPAD 16        // the cell list is padded to whole prefetch rounds of gcorr_kernel (4 cells x GPF groups; 32 until round 4: 11 % of the G products were padding)
// The maps of a pair that the wave tables kernel takes (wave_ok != 0), written by pair_setup_kernel -- one THREAD per pair there, where
// the same index arithmetic and a counting sort cost a twentieth of the ballots a whole wave spent on them (round 4: the maps were 30 %
// of gtables_wave_kernel's cycle stamps).  G_MAPB bytes per pair, read back with one 8-byte load per lane:
//   XPOS  [40]  position of x sample s in the member list ordered by (response column, s); 0xFF: outside the table
//   YPOS  [40]  the same for the y samples, ordered by (row j, s)
//   COLI  [40]  response column i of the pair's k-th distinct column     COLSTART [41]  first member of column k, then the count
//   JSTART[49]  first member of row jmin + k, then the count
//   ZSH   [64]  response shift of slice iz_lo + k, minus u_min           ZINV [64]  window edges the slice is invalid at (bits)
//   AMB   [1]   slices whose shift was a rounding tie (statistics)
// The timing switches of tools/gform_phases.py (debug_gform / debug_phases bits read inside gcorr_kernel and gtables_wave_kernel) and their cycle stamps are
// compiled in with -DLDSIM_GCORR_DEBUG only (make DEBUG_GCORR=1): every switch is a loop-invariant condition the compiler keeps in
// two scalar registers, the kernel has more of those than registers, and a spilled one costs vector instructions -- which on
// this part do not overlap with the f64 matrix instructions (SQ_VALU_MFMA_COEXEC_CYCLES = 0: their times add up to the kernel's).
#ifdef LDSIM_GCORR_DEBUG
#define GDBG(bit) ((GA.dbg & (bit)) != 0)
#define GPHASE(bit) ((A.debug_phases & (bit)) != 0)
#else
#define GDBG(bit) false
#define GPHASE(bit) false
#endif
#define G_MAP_NS 40
#define G_MAPB 384
#define G_MAP_XPOS 0
#define G_MAP_YPOS 40
#define G_MAP_COLI 80
#define G_MAP_COLSTART 120
#define G_MAP_JSTART 168
#define G_MAP_ZSH 224
#define G_MAP_ZINV 288
#define G_MAP_AMB 352
#define G_HDR 0             // header ints of a record (none: what gcorr_kernel needs first sits in GInfo, one load away)
#define G_CELL0 4           // ints before the first cell entry of a batch (counts; four, so that every table starts 16-byte aligned)

// per pair, written by pair_setup_kernel next to PairParams: what sizes the record before any table exists
struct GInfo {
  int32_t ncol, NJ, jmin, u_min, NU, edge_bound, NB, status;   // edge_bound: superset of the edges that need a Z table
  int32_t NQ, it0, T, it_w0, it_w1;                             // copies of the PairParams fields gcorr_kernel needs
  int32_t emask;                                                // written by gtables_kernel: the edges some slice is invalid at
  int32_t wave_ok;                                              // gtables_wave_kernel takes the pair (one chunk of slices): 1 = <= 128 shifts, X | Y bins <= 54,
                                                                // 2 = <= 256 / <= 80 (its wide instantiation, over a list); 0 = gtables_kernel
  int32_t pad;
  unsigned long long off;                                       // record offset in doubles (exclusive scan of `size`)
  unsigned long long size;                                      // record size in doubles
};

// record layout in doubles from `off`:
//   per node batch b:
//     cells     2 + padded / 2              int count padded to a multiple of G_CELLPAD with dummies, int count of real cells,
//                                           two unused, then u32 row | col << 16 | j << 24 per cell
//     X         rows * ncol                 X[n][col]; rows = the batch's nodes rounded up to 4 (16 for all batches but the last:
//                                           59 % of the survey workload's pairs end in a batch of 1 .. 4 nodes)
//     Y         rows * NJ                   Y[n][j]
//     Z         rows * NUr                  Z[n][u]; NUr = NU rounded up to 16
//     Zi        popcount(edge_bound) * rows * NUr   per possible window edge: Z over the slices that are invalid at that edge only
//                                           (gcorr_kernel takes their share of the one tick the edge maps to back: one more
//                                           16-node product against the column G_n[edge_k] it holds anyway)
__host__ __device__ __forceinline__ int g_nur(int NU) { return (NU + 15) & ~15; }
__host__ __device__ __forceinline__ int g_popc3(int m) { return (m & 1) + ((m >> 1) & 1) + ((m >> 2) & 1); }
__host__ __device__ __forceinline__ unsigned long long g_cells_doubles(int ncol, int NJ) {
  return G_CELL0 / 2 + (unsigned long long)((ncol * NJ + G_CELLPAD - 1) & ~(G_CELLPAD - 1)) / 2;
}
// node rows batch b of a rule of NQ nodes stores
__host__ __device__ __forceinline__ int g_rows(int NQ, int b) {
  const int nb = NQ - G_NODES * b;
  return nb >= G_NODES ? G_NODES : ((nb + 3) & ~3);
}
__host__ __device__ __forceinline__ unsigned long long g_batch_doubles(int ncol, int NJ, int NU, int edge_bound, int rows = G_NODES) {
  return g_cells_doubles(ncol, NJ) + (unsigned long long)rows * (ncol + NJ + (1 + g_popc3(edge_bound)) * g_nur(NU));
}
// (batch b starts b full batches into the record: only the last one is short)
__host__ __device__ __forceinline__ unsigned long long g_record_doubles(int NB, int NQ, int ncol, int NJ, int NU, int edge_bound) {
  return G_HDR / 2 + (unsigned long long)(NB - 1) * g_batch_doubles(ncol, NJ, NU, edge_bound) +
         g_batch_doubles(ncol, NJ, NU, edge_bound, g_rows(NQ, NB - 1));
}

struct GArgs {
  CurArgs c;
  const PairParams* pp;
  GInfo* gi;
  const unsigned char* maps;    // [n_pairs][G_MAPB], pairs with wave_ok != 0
  double* rec;                   // record pool
  int32_t* flags;                // [n_pairs] 1 = the monolithic kernel recomputes this pair
  const double* resp_pad;        // response rows with RESP_PAD zeros either side, zeros outside the staged range
  int32_t nkp, k_lo, k_hi;
  const double* glx;
  const double* glw;
  int32_t dbg;                   // timing tools (option debug_gform)
  // edge_ks() of the launch's constants, computed on the host: a kernel that reads them from the constants block in HBM puts one
  // more memory round trip in front of every pair (gcorr_kernel, gtables_wave_kernel)
  int32_t edge_k[NEDGE], k_stage_lo, k_stage_hi;
};
