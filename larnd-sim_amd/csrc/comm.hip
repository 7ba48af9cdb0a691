// comm.hip -- the one exchange step of the batch-sharded path, directly on RCCL over xGMI: one process per GPU,
// every rank simulates its own (event, TPC-group) batches with no data-path collective, then the compact hit rows
// {batch, pixel, adc, slot, tick} (24 B) are reassembled on every rank: an all-gather of the row counts followed by an
// all-gather-v of the rows (one ncclBroadcast per rank inside a group call: no padding to the largest shard).
// The communicator is bootstrapped from an ncclUniqueId the host side hands to every rank (larndsim_amd/comm.py).
#include <rccl/rccl.h>

#include "ldsim_dev.h"

#define NCCLCHK(expr)                                                                                  \
  do {                                                                                                 \
    ncclResult_t r_ = (expr);                                                                          \
    if (r_ != ncclSuccess) {                                                                           \
      ldsim_set_error("%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__, __LINE__);     \
      return LDSIM_EHIP;                                                                               \
    }                                                                                                  \
  } while (0)
#define NEEDC(cond, msg)           \
  do {                             \
    if (!(cond)) {                 \
      ldsim_set_error("%s", msg);  \
      return LDSIM_EINVAL;         \
    }                              \
  } while (0)

static_assert(sizeof(ncclUniqueId) == LDSIM_COMM_ID_BYTES, "LDSIM_COMM_ID_BYTES must be sizeof(ncclUniqueId)");

extern "C" int ldsim_comm_unique_id(void* id) {
  NEEDC(id, "null id");
  ncclUniqueId u;
  NCCLCHK(ncclGetUniqueId(&u));
  memcpy(id, &u, sizeof(u));
  return 0;
}

extern "C" int ldsim_comm_init(ldsim_ctx* ctx, const void* id, int32_t rank, int32_t world) {
  LDSIM_ENTER(ctx);
  NEEDC(ctx && id && world >= 1 && rank >= 0 && rank < world, "bad communicator arguments");
  NEEDC(!ctx->comm, "communicator already initialised");
  HIPCHK(hipSetDevice(ctx->device));
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  ncclComm_t comm = nullptr;
  NCCLCHK(ncclCommInitRank(&comm, world, u, rank));
  ctx->comm = (void*)comm;
  ctx->comm_rank = rank;
  ctx->comm_world = world;
  return 0;
}

// ranks in the communicator as RCCL reports them (ncclCommCount / ncclCommUserRank): lets a caller check that every rank of the
// launch really joined, instead of trusting the environment it was started with
extern "C" int ldsim_comm_count(ldsim_ctx* ctx, int32_t* n_ranks, int32_t* rank) {
  LDSIM_ENTER(ctx);
  NEEDC(ctx && ctx->comm && n_ranks, "no communicator / null argument");
  int n = 0, r = 0;
  NCCLCHK(ncclCommCount((ncclComm_t)ctx->comm, &n));
  NCCLCHK(ncclCommUserRank((ncclComm_t)ctx->comm, &r));
  *n_ranks = n;
  if (rank) *rank = r;
  return 0;
}

extern "C" int ldsim_comm_destroy(ldsim_ctx* ctx) {
  LDSIM_ENTER(ctx);
  if (!ctx || !ctx->comm) return 0;
  (void)hipStreamSynchronize(ctx->stream);
  NCCLCHK(ncclCommDestroy((ncclComm_t)ctx->comm));
  ctx->comm = nullptr;
  ctx->comm_world = 0;
  return 0;
}

// value (host, in/out) reduced over the ranks: op 0 = sum, 1 = max.  Doubles as the barrier (every rank leaves after all entered).
extern "C" int ldsim_comm_allreduce_f64(ldsim_ctx* ctx, double* value, int32_t op) {
  LDSIM_ENTER(ctx);
  NEEDC(ctx && value && ctx->comm, "no communicator");
  HIPCHK(hipSetDevice(ctx->device));
  int rc = ldsim_ensure_buf(ctx, &ctx->comm_tmp, 64 + 8 * (size_t)ctx->comm_world);
  if (rc) return rc;
  double* d = (double*)ctx->comm_tmp.p;
  HIPCHK(hipMemcpyAsync(d, value, 8, hipMemcpyHostToDevice, ctx->stream));
  NCCLCHK(ncclAllReduce(d, d, 1, ncclDouble, op == 1 ? ncclMax : ncclSum, (ncclComm_t)ctx->comm, ctx->stream));
  HIPCHK(hipMemcpyAsync(value, d, 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

// Keep the compact hit rows of the chain calls of one pass: reset != 0 starts a new pass, then the last chain call's rows
// are appended (device-to-device, on the ctx stream, before the next chain call reuses its buffer).
extern "C" int ldsim_hits_accumulate(ldsim_ctx* ctx, int32_t reset) {
  LDSIM_ENTER(ctx);
  NEEDC(ctx, "null ctx");
  HIPCHK(hipSetDevice(ctx->device));
  if (reset) ctx->hits_acc_rows = 0;
  const int64_t n = ctx->chain_hits;
  if (n == 0) return 0;
  const size_t need = (size_t)(ctx->hits_acc_rows + n) * 24;
  if (need > ctx->hits_acc.bytes) {
    DevBuf nb;
    int rc = ldsim_ensure_buf(ctx, &nb, need * 2);
    if (rc) return rc;
    if (ctx->hits_acc_rows)
      HIPCHK(hipMemcpyAsync(nb.p, ctx->hits_acc.p, (size_t)ctx->hits_acc_rows * 24, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->hits_acc.p) (void)hipFree(ctx->hits_acc.p);
    ctx->hits_acc = nb;
  }
  HIPCHK(hipMemcpyAsync((char*)ctx->hits_acc.p + (size_t)ctx->hits_acc_rows * 24, ctx->scratch[SB_HITS].p, (size_t)n * 24,
                        hipMemcpyDeviceToDevice, ctx->stream));
  ctx->hits_acc_rows += n;
  return 0;
}

// All-gather-v of the accumulated rows.  counts[world] (host, may be NULL) receives every rank's row count; *gathered is a
// device pointer owned by the ctx (valid until the next call) holding the rows of rank 0, 1, .. back to back.
extern "C" int ldsim_comm_allgather_hits(ldsim_ctx* ctx, void** gathered, int64_t* total_rows, int64_t* counts) {
  LDSIM_ENTER(ctx);
  NEEDC(ctx && gathered && total_rows && ctx->comm, "no communicator / null argument");
  HIPCHK(hipSetDevice(ctx->device));
  const int W = ctx->comm_world;
  ncclComm_t comm = (ncclComm_t)ctx->comm;
  int rc = ldsim_ensure_buf(ctx, &ctx->comm_tmp, 64 + 8 * (size_t)W);
  if (rc) return rc;
  int64_t* d_mine = (int64_t*)ctx->comm_tmp.p;
  int64_t* d_all = (int64_t*)((char*)ctx->comm_tmp.p + 64);
  const int64_t mine = ctx->hits_acc_rows;
  HIPCHK(hipMemcpyAsync(d_mine, &mine, 8, hipMemcpyHostToDevice, ctx->stream));
  NCCLCHK(ncclAllGather(d_mine, d_all, 1, ncclInt64, comm, ctx->stream));
  std::vector<int64_t> h((size_t)W);
  HIPCHK(hipMemcpyAsync(h.data(), d_all, 8 * (size_t)W, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  int64_t total = 0;
  for (int r = 0; r < W; r++) {
    NEEDC(h[r] >= 0, "negative row count received");
    total += h[r];
  }
  if ((rc = ldsim_ensure_buf(ctx, &ctx->hits_all, (size_t)(total > 0 ? total : 1) * 24))) return rc;
  NCCLCHK(ncclGroupStart());
  int64_t off = 0;
  for (int r = 0; r < W; r++) {
    if (h[r] > 0) {
      char* dst = (char*)ctx->hits_all.p + (size_t)off * 24;
      const void* src = (r == ctx->comm_rank) ? ctx->hits_acc.p : (const void*)dst;
      NCCLCHK(ncclBroadcast(src, dst, (size_t)h[r] * 24, ncclChar, r, comm, ctx->stream));
    }
    off += h[r];
  }
  NCCLCHK(ncclGroupEnd());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (counts) memcpy(counts, h.data(), 8 * (size_t)W);
  *gathered = ctx->hits_all.p;
  *total_rows = total;
  return 0;
}

// rows [0, n) of the gathered buffer to host (tests / the driver's output writer)
extern "C" int ldsim_comm_gathered_download(ldsim_ctx* ctx, void* rows, int64_t n) {
  LDSIM_ENTER(ctx);
  NEEDC(ctx && (rows || n == 0), "null argument");
  if (n == 0) return 0;
  NEEDC((size_t)n * 24 <= ctx->hits_all.bytes, "more rows requested than gathered");
  HIPCHK(hipMemcpy(rows, ctx->hits_all.p, (size_t)n * 24, hipMemcpyDeviceToHost));
  return 0;
}
