// crc32.hip -- host code: CRC-32 (reflected 0xEDB88320, what ZIP members carry) of a member that exists only as a list of
// chunks (the .npy header and the per-launch pieces of a dataset), on several host threads (hostpool.h).  The output writer of the driver
// (cli/simulate_pixels.py OutputFile: the stand-in for the reference's h5py datasets, cli/simulate_pixels.py:1240-1301 there)
// spent 0.95 of 3.25 s per 10^6 segments in Python's zlib.crc32 over ~1 GB of uncompressed datasets.
//
// Per piece: slicing-by-8 (8 table look-ups per 8 input bytes).  Pieces are joined with the GF(2) "append len zero bytes"
// operator (x^(8 len) mod P by repeated squaring of the one-zero-bit matrix) -- the published zlib crc32_combine construction.
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/ldsim.h"
#include "hostpool.h"

namespace {

struct Tables {
  uint32_t t[8][256];
  Tables() {
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t c = i;
      for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      t[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; i++)
      for (int s = 1; s < 8; s++) t[s][i] = t[0][t[s - 1][i] & 255] ^ (t[s - 1][i] >> 8);
  }
};
const Tables T;

// running CRC over [p, p + n): `crc` is the finished CRC of what came before (0 for nothing)
uint32_t crc_bytes(uint32_t crc, const uint8_t* p, uint64_t n) {
  uint32_t c = ~crc;
  while (n && ((uintptr_t)p & 7)) { c = T.t[0][(c ^ *p++) & 255] ^ (c >> 8); n--; }
  while (n >= 8) {
    uint64_t w;
    memcpy(&w, p, 8);
    w ^= c;                                   // little-endian host (x86-64)
    c = T.t[7][w & 255] ^ T.t[6][(w >> 8) & 255] ^ T.t[5][(w >> 16) & 255] ^ T.t[4][(w >> 24) & 255] ^
        T.t[3][(w >> 32) & 255] ^ T.t[2][(w >> 40) & 255] ^ T.t[1][(w >> 48) & 255] ^ T.t[0][w >> 56];
    p += 8; n -= 8;
  }
  while (n--) c = T.t[0][(c ^ *p++) & 255] ^ (c >> 8);
  return ~c;
}

uint32_t mat_times(const uint32_t* m, uint32_t v) {
  uint32_t s = 0;
  for (; v; v >>= 1, m++) if (v & 1) s ^= *m;
  return s;
}
void mat_square(uint32_t* sq, const uint32_t* m) { for (int i = 0; i < 32; i++) sq[i] = mat_times(m, m[i]); }

// CRC of A|B from crc(A), crc(B) and len(B)
uint32_t crc_join(uint32_t a, uint32_t b, uint64_t len_b) {
  if (!len_b) return a;
  uint32_t even[32], odd[32];
  odd[0] = 0xEDB88320u;                       // one zero BIT appended
  for (int i = 1; i < 32; i++) odd[i] = 1u << (i - 1);
  mat_square(even, odd);                      // two bits
  mat_square(odd, even);                      // four bits
  for (;;) {
    mat_square(even, odd);                    // first pass: one zero byte
    if (len_b & 1) a = mat_times(even, a);
    if (!(len_b >>= 1)) break;
    mat_square(odd, even);
    if (len_b & 1) a = mat_times(odd, a);
    if (!(len_b >>= 1)) break;
  }
  return a ^ b;
}

}  // namespace

extern "C" uint32_t ldsim_crc32_parts(const void* const* parts, const uint64_t* sizes, int64_t n_parts, int32_t n_threads) {
  struct Piece { const uint8_t* p; uint64_t n; uint32_t crc; };
  const uint64_t PIECE = 4u << 20;
  std::vector<Piece> pieces;
  for (int64_t i = 0; i < n_parts; i++)
    for (uint64_t o = 0; o < sizes[i]; o += PIECE)
      pieces.push_back({(const uint8_t*)parts[i] + o, std::min(PIECE, sizes[i] - o), 0u});
  const int nt = std::max(1, std::min({n_threads > 0 ? n_threads : HostPool::size(), HostPool::size(), (int)pieces.size()}));
  HostPool::run(nt, [&](int t) { for (size_t k = t; k < pieces.size(); k += nt) pieces[k].crc = crc_bytes(0u, pieces[k].p, pieces[k].n); });
  uint32_t crc = 0;
  for (const Piece& q : pieces) crc = crc_join(crc, q.crc, q.n);
  return crc;
}
