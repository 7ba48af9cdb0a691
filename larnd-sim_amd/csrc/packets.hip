// packets.hip -- host code: the hit loop of fee.export_to_hdf5 (larndsim/fee.py:143-344) on the COMPACT rows of a chain launch
// (kernels_compact.hip): LArPix packets and mc_packets_assn rows without a dense [pixel][30][50] fraction array anywhere.
//
// larndsim_amd/packets.py: build_packets does the same with array operations on the dense per-pixel arrays (its bytes are what
// the reference's loop writes: tests/golden/packets_*.npz, tests/packets_loop.py); the driver spent 1.7 of 5.7 s per 10^6
// segments there and 0.6 s rebuilding the dense rows to feed it.  Here the loop runs as a loop: per hit the clock rollover
// (:164-183), the "event changed" packets (:187-230), the "timestamp changed" packet (:267-277), the data packet with its parity,
// and the association row -- fractions in descending order, the ASSOCIATION_COUNT_TO_STORE largest kept, trajectories summed
// per id (:284-344).  What is per row and array-shaped (pixel -> io_group / io_channel / chip / channel, event start times)
// arrives precomputed from packets.py's own code.  No GPU, no ctx: plain C behind the C-ABI.
//
// Order of equal fractions: the reference sorts with np.argsort, whose order of equal keys depends on numpy's build (its SIMD
// sort kernels are not stable) -- unpinned by construction.  Here equal fractions keep descending slot order (what flipping a
// stable ascending sort gives).  Only real track slots with bit-equal fractions are concerned: the unused slots of a pixel all
// carry fraction 0 and track id -1, so their order among themselves cannot be seen in the output.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/ldsim.h"
#include "hostpool.h"

void ldsim_set_error(const char* fmt, ...);

#pragma pack(push, 1)
struct PacketRow {       // larpix.format.hdf5format 2.4, dataset 'packets' (numpy packed layout, larndsim_amd/packets.py packets_dtype)
  uint8_t io_group, io_channel, chip_id, packet_type, downstream_marker, parity, valid_parity, channel_id;
  uint64_t timestamp;
  uint8_t dataword, trigger_type, local_fifo, shared_fifo, register_address, register_data, direction, local_fifo_events;
  uint16_t shared_fifo_events;
  uint32_t counter;
  uint8_t fifo_diagnostics_enabled, first_packet;
  uint32_t receipt_timestamp;
};
#pragma pack(pop)
static_assert(sizeof(PacketRow) == 36, "packets dtype layout");

extern "C" int32_t ldsim_packets_row_bytes(void) { return (int32_t)sizeof(PacketRow); }
extern "C" int32_t ldsim_packets_assn_row_bytes(int32_t n_keep) { return 8 + 32 * n_keep; }

namespace {
enum { P_DATA = 0, P_TIMESTAMP = 4, P_SYNC = 6, P_TRIGGER = 7 };

inline void other_row(PacketRow* r, int ptype, int io_group, uint64_t timestamp, int trigger_type) {
  memset(r, 0, sizeof(*r));
  r->io_group = (uint8_t)io_group;
  r->packet_type = (uint8_t)ptype;
  r->valid_parity = 1;
  r->timestamp = timestamp;
  r->trigger_type = (uint8_t)trigger_type;
}
// assn row: event_ids (1,) i8 | segment_ids (n,) i8 | fraction (n,) f8 | file_traj_ids (n,) i8 | fraction_traj (n,) f8
inline void empty_assn(char* row, int n_keep) {
  int64_t* ev = (int64_t*)row;
  int64_t* seg = ev + 1;
  double* fr = (double*)(seg + n_keep);
  int64_t* tid = (int64_t*)(fr + n_keep);
  double* tfr = (double*)(tid + n_keep);
  ev[0] = -1;
  for (int k = 0; k < n_keep; k++) { seg[k] = -1; fr[k] = 0.0; tid[k] = -1; tfr[k] = 0.0; }
}

struct Ent { double f; int slot; };
struct TEnt { int64_t id; double f; int pos; };
struct Todo { int64_t out, i, h; };

// association row of data hit h of row i (fee.py:284-344)
void fill_assn(const LdsimPacketsIn* in, int64_t i, int64_t h, char* row, Ent* ent, TEnt* tent) {
  const int n_keep = in->n_keep, MT = in->max_tracks;
  const int64_t h0 = in->row_hit0[i], t0 = in->row_trk0[i];
  const int nt = (int)(in->row_trk0[i + 1] - t0);
  const int64_t event = in->row_event[i];
  empty_assn(row, n_keep);
  int64_t* ev = (int64_t*)row;
  int64_t* seg = ev + 1;
  double* fr = (double*)(seg + n_keep);
  int64_t* tid = (int64_t*)(fr + n_keep);
  double* tfr = (double*)(tid + n_keep);
  ev[0] = event;
  // all max_tracks slots in descending fraction order (equal fractions: descending slot); unused slots: fraction 0, id -1
  const double* f = in->hit_frac + in->row_frac0[i] + (int64_t)(h - h0) * nt;
  // Only the nt filled slots are sorted; the MT - nt unused ones (slots MT-1 .. nt, all 0) sit as one run behind the last
  // positive fraction: a filled slot with fraction 0 has the lower slot number and follows them.
  for (int k = 0; k < nt; k++) { ent[k].f = f[k]; ent[k].slot = k; }
  std::sort(ent, ent + nt, [](const Ent& a, const Ent& b) { return a.f > b.f || (a.f == b.f && a.slot > b.slot); });
  int n_pos = 0;
  while (n_pos < nt && ent[n_pos].f > 0.0) n_pos++;
  const int w = n_keep < MT ? n_keep : MT, n_unused = MT - nt;
  for (int k = 0; k < w; k++) {
    const int e = k < n_pos ? k : (k < n_pos + n_unused ? -1 : k - n_unused);
    if (e >= 0) { seg[k] = in->trk_segment[t0 + ent[e].slot]; fr[k] = ent[e].f; }      // (else: -1 and 0 from empty_assn)
  }
  // trajectories: ids ascending, each with the sum of its slots' fractions taken in fraction order (left to right), stored f4
  int ntr = 0;
  for (int k = 0; k < nt; k++) {
    const int64_t id = in->trk_traj[t0 + ent[k].slot];
    if (id > -1) { tent[ntr].id = id; tent[ntr].f = ent[k].f; tent[ntr].pos = ntr; ntr++; }
  }
  std::sort(tent, tent + ntr, [](const TEnt& a, const TEnt& b) { return a.id < b.id || (a.id == b.id && a.pos < b.pos); });
  int ng = 0;
  for (int k = 0; k < ntr && ng < n_keep;) {
    double sum = tent[k].f;
    int e = k + 1;
    while (e < ntr && tent[e].id == tent[k].id) { sum += tent[e].f; e++; }
    tid[ng] = (int64_t)(int32_t)tent[k].id;            // (the reference's id array is int32)
    tfr[ng] = (double)(float)sum;
    ng++;
    k = e;
  }
}
}  // namespace

extern "C" int64_t ldsim_packets_build(const LdsimPacketsIn* in, void* packets_out, void* assn_out, int64_t capacity) {
  if (!in || !packets_out || !assn_out || in->n_rows < 0 || in->n_keep <= 0 || in->n_keep > 64 || in->max_tracks <= 0 ||
      in->max_tracks > 256) {
    ldsim_set_error("ldsim_packets_build: bad argument");
    return LDSIM_EINVAL;
  }
  const int n_keep = in->n_keep, MT = in->max_tracks;
  const int64_t CRP = in->clock_reset_period;
  const double cc = in->clock_cycle;
  const size_t assn_bytes = (size_t)(8 + 32 * n_keep);
  PacketRow* pk = (PacketRow*)packets_out;
  char* as = (char*)assn_out;
  int64_t n_out = 0;
  auto need = [&](int64_t k) { return n_out + k <= capacity; };
  int64_t off = 0, off_row0_final = 0;
  bool have_prev_event = false, have_prev_tick = false;
  int64_t prev_event = -1, prev_tick = -1;
  std::vector<Todo> todo;
  for (int64_t i = 0; i < in->n_rows; i++) {
    const int64_t h0 = in->row_hit0[i], h1 = in->row_hit0[i + 1];
    const int64_t t0 = in->row_trk0[i];
    const int nt = (int)(in->row_trk0[i + 1] - t0);
    if (nt > MT) { ldsim_set_error("ldsim_packets_build: a row holds more track slots than max_tracks"); return LDSIM_EINVAL; }
    const bool row0 = in->first_row_is_row0 && i == 0;
    const int64_t base = in->row_base[i], event = in->row_event[i];
    const bool passes = in->row_ok[i] != 0;
    for (int64_t h = h0; h < h1; h++) {
      // ---- clock rollover (fee.py:164-183): `off` reset periods have been taken off this and every later row -----------------------
      const double t_cc = in->hit_tick[h] / cc;
      int64_t event_t0, tick;
      for (;;) {
        event_t0 = base - off * CRP;
        tick = (int64_t)floor(t_cc + (double)event_t0);
        if (event_t0 > CRP - 1 || tick > CRP - 1) off++; else break;
      }
      if (row0) off_row0_final = off;
      // (row 0 of event_start_time_list as the timestamp packets see it: rollovers reach it only while row 0 is processed)
      const int64_t row0_value = in->base0 - (row0 ? off : off_row0_final) * CRP;
      // Python's % on these (possibly negative) integers: result has the sign of the divisor
      auto pymod = [](int64_t a, int64_t m) { int64_t r = a % m; return r < 0 ? r + m : r; };
      const int64_t event_t0_m = pymod(event_t0, CRP), tick_m = pymod(tick, CRP);
      // ---- new event: timestamp + sync per io_group, the event's light triggers (fee.py:187-230) ---------------------------------
      if (in->light_trig_mode != 1) {
        const bool changed = have_prev_event ? event != prev_event : event != -1;
        have_prev_event = true;
        prev_event = event;
        if (changed) {
          if (!need(2 * in->n_io_groups)) goto full;
          for (int g = 0; g < in->n_io_groups; g++) {
            other_row(&pk[n_out], P_TIMESTAMP, in->io_groups[g], (uint64_t)in->row_ts_s[i], 0);
            empty_assn(as + (size_t)n_out * assn_bytes, n_keep);
            n_out++;
            other_row(&pk[n_out], P_SYNC, in->io_groups[g], (uint64_t)tick_m, 'S');
            empty_assn(as + (size_t)n_out * assn_bytes, n_keep);
            n_out++;
          }
          for (int64_t q = 0; q < in->n_trig; q++) {
            if (in->trig_event[q] != event) continue;
            const int64_t t_trig = pymod((int64_t)floor(in->trig_time[q] / cc + (double)event_t0_m), CRP);
            if (in->light_trig_mode == 0) {
              const int64_t mod = in->trig_module[q];
              bool found = false;
              for (int m = 0; m < in->n_modules; m++) {
                if (in->module_ids[m] != mod) continue;
                found = true;
                for (int g = in->module_group0[m]; g < in->module_group0[m + 1]; g++) {
                  if (!need(1)) goto full;
                  other_row(&pk[n_out], P_TRIGGER, in->module_groups[g], (uint64_t)t_trig, 2);
                  empty_assn(as + (size_t)n_out * assn_bytes, n_keep);
                  n_out++;
                }
              }
              if (!found) { ldsim_set_error("ldsim_packets_build: light trigger on a module without io groups"); return LDSIM_EINVAL; }
            }
          }
        }
      }
      if (!passes) continue;
      // ---- timestamp changed (fee.py:267-277) ---------------------------------------------------------------------------------------
      {
        const bool changed = have_prev_tick ? tick_m != prev_tick : tick_m != -1;
        have_prev_tick = true;
        prev_tick = tick_m;
        if (changed) {
          if (!need(1)) goto full;
          const double ts = floor((double)row0_value * cc * in->mus / in->s);
          other_row(&pk[n_out], P_TIMESTAMP, in->row_io_group[i], (uint64_t)(int64_t)ts, 0);
          empty_assn(as + (size_t)n_out * assn_bytes, n_keep);
          n_out++;
        }
      }
      // ---- the data packet ------------------------------------------------------------------------------------------------------------
      if (!need(1)) goto full;
      {
        PacketRow* r = &pk[n_out];
        memset(r, 0, sizeof(*r));
        const int64_t dataword = (int64_t)in->hit_adc[h];
        const int chip = in->row_chip[i], channel = in->row_channel[i];
        r->io_group = (uint8_t)in->row_io_group[i];
        r->io_channel = (uint8_t)in->row_io_channel[i];
        r->chip_id = (uint8_t)chip;
        r->packet_type = P_DATA;
        r->valid_parity = 1;
        r->channel_id = (uint8_t)channel;
        r->timestamp = (uint64_t)tick_m;
        r->dataword = (uint8_t)dataword;
        r->first_packet = 1;
        r->receipt_timestamp = (uint32_t)tick_m;
        const uint64_t word = (((uint64_t)chip & 0xFFull) << 2) | (((uint64_t)channel & 0x3Full) << 10) |
                              (((uint64_t)tick_m & 0x7FFFFFFFull) << 16) | (1ull << 47) | (((uint64_t)dataword & 0xFFull) << 48);
        r->parity = (uint8_t)(1 - (__builtin_popcountll(word) & 1));
        todo.push_back({n_out, i, h});         // the association row: second pass, on host threads
      }
      n_out++;
    }
  }
  {
    // ---- association rows (fee.py:284-344): independent per data packet, on the parked host threads (hostpool.h) ------------------
    const int n_parts = HostPool::parts_for(todo.size(), 1024);
    HostPool::run(n_parts, [&](int t) {
      std::vector<Ent> ent((size_t)MT);
      std::vector<TEnt> tent((size_t)MT);
      const size_t lo = todo.size() * (size_t)t / n_parts, hi = todo.size() * (size_t)(t + 1) / n_parts;
      for (size_t q = lo; q < hi; q++) fill_assn(in, todo[q].i, todo[q].h, as + (size_t)todo[q].out * assn_bytes, ent.data(), tent.data());
    });
  }
  return n_out;
full:
  ldsim_set_error("ldsim_packets_build: output capacity %lld exhausted", (long long)capacity);
  return LDSIM_EINVAL;
}
