// shard.hip -- candidate batches sharded over the GPUs of one node, for a C++ host (no Python, no torch).
//
// TBV registers / verifies every loop-closure candidate independently (tbv_slam/src/tbv_slam/loopclosure.cpp:658-721;
// each candidate builds its own n_scan_normal_reg at :56), so a batch shards trivially: contiguous blocks of
// ceil(n / world) candidates per rank, no exchange on the data path, ONE all_gather of fixed-size result records at the
// end (rank order = candidate order).  The collective itself belongs to the host -- it owns the communicator -- and is
// handed in as a callback; cfear_rccl_allgather below is the ready-made one over an ncclComm_t (RCCL over xGMI on
// MI355X), resolved from librccl.so at run time so that the library has no link-time dependency on it.
#include <dlfcn.h>

#include <cstring>

#include "common.hpp"

extern "C" int cfear_shard_range(int32_t n, int32_t world, int32_t rank, int32_t* lo, int32_t* hi, int32_t* per_rank) {
  if (n < 0 || world < 1 || rank < 0 || rank >= world || !lo || !hi) return CFEAR_ERR_INVALID_ARGUMENT;
  const int per = (n + world - 1) / world;
  *lo = std::min(rank * per, n);
  *hi = std::min(*lo + per, n);
  if (per_rank) *per_rank = per;
  return CFEAR_OK;
}

// ---- the exchange step ------------------------------------------------------------------------------------------------
// Every rank contributes ceil(n / world) records (its block, zero-padded) PLUS a trailer {int32 rank status, int32 records};
// all[] receives the n real records in candidate order.  gather(user, send, recv, bytes): recv = the concatenation of every
// rank's `bytes` bytes, rank order.  A rank whose own sub-batch failed STILL enters the collective -- its peers are already
// on their way into it and would wait for ever -- with its status in the trailer; after the exchange every rank returns
// the first failed rank's status, so the whole job sees one verdict.
namespace {
struct ShardTrailer { int32_t status, n_records; };

int first_rank_status(const char* recv, size_t bytes, int world) {
  for (int r = 0; r < world; r++) {
    ShardTrailer t;
    memcpy(&t, recv + (size_t)r * bytes + (bytes - sizeof(ShardTrailer)), sizeof(t));
    if (t.status != CFEAR_OK) return t.status;
  }
  return CFEAR_OK;
}

void unpack_blocks(const char* recv, size_t bytes, int32_t n_total, int32_t record_bytes, int32_t world, void* all) {
  for (int r = 0; r < world; r++) {                          // drop every rank's padding and trailer
    int32_t l2, h2;
    cfear_shard_range(n_total, world, r, &l2, &h2, nullptr);
    if (h2 > l2) memcpy((char*)all + (size_t)l2 * record_bytes, recv + (size_t)r * bytes, (size_t)(h2 - l2) * record_bytes);
  }
}

// local_status: what this rank's own work returned (CFEAR_OK or an error; on error `local` may be null / stale)
int gather_records_status(const void* local, int local_status, int32_t n_total, int32_t record_bytes, int32_t world, int32_t rank,
                          cfear_allgather_fn gather, void* user, void* all) {
  int32_t lo, hi, per;
  int rc = cfear_shard_range(n_total, world, rank, &lo, &hi, &per);
  if (rc != CFEAR_OK) return rc;
  if (!gather) {                                             // no collective: a single rank
    if (world != 1) return CFEAR_ERR_INVALID_ARGUMENT;
    if (local_status != CFEAR_OK) return local_status;
    if (n_total > 0) memcpy(all, local, (size_t)n_total * record_bytes);
    return CFEAR_OK;
  }
  const size_t bytes = (size_t)per * record_bytes + sizeof(ShardTrailer);
  std::vector<char> send(bytes, 0), recv(bytes * (size_t)world);
  if (hi > lo && local_status == CFEAR_OK && local) memcpy(send.data(), local, (size_t)(hi - lo) * record_bytes);
  const ShardTrailer t{local_status, hi - lo};
  memcpy(send.data() + bytes - sizeof(t), &t, sizeof(t));
  rc = gather(user, send.data(), recv.data(), bytes);
  if (rc != 0) return rc < 0 ? rc : CFEAR_ERR_HIP;           // a cfear status is handed back as it is; anything else (a callback's own positive code) is not one
  unpack_blocks(recv.data(), bytes, n_total, record_bytes, world, all);
  return first_rank_status(recv.data(), bytes, world);
}
}  // namespace

extern "C" int cfear_gather_records(const void* local, int32_t n_total, int32_t record_bytes, int32_t world, int32_t rank,
                                    cfear_allgather_fn gather, void* user, void* all) {
  if (!all || record_bytes <= 0 || (n_total > 0 && !local && world > 0 && rank >= 0 && rank < world && n_total > rank * ((n_total + world - 1) / world)))
    return CFEAR_ERR_INVALID_ARGUMENT;
  return gather_records_status(local, CFEAR_OK, n_total, record_bytes, world, rank, gather, user, all);
}

// ---- ready-made callback over an ncclComm_t (RCCL) ----------------------------------------------------------------
namespace {
typedef int (*nccl_allgather_t)(const void*, void*, size_t, int /*ncclDataType_t*/, void* /*ncclComm_t*/, hipStream_t);
nccl_allgather_t resolve_allgather() {
  static nccl_allgather_t fn = []() -> nccl_allgather_t {
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    return h ? (nccl_allgather_t)dlsym(h, "ncclAllGather") : nullptr;
  }();
  return fn;
}
}  // namespace

// device buffers, enqueued on the communicator's context stream, NOT synchronised
extern "C" int cfear_rccl_allgather_device(void* user, const void* d_send, void* d_recv_all, size_t bytes) {
  cfear_rccl_comm* c = (cfear_rccl_comm*)user;
  if (!c || !c->ctx || !c->nccl_comm || c->world < 1) return CFEAR_ERR_INVALID_ARGUMENT;
  nccl_allgather_t ag = resolve_allgather();
  if (!ag) return cfear_set_error(c->ctx, CFEAR_ERR_HIP, "librccl.so / ncclAllGather not found");
  const int st = ag(d_send, d_recv_all, bytes, 0 /* ncclInt8 */, c->nccl_comm, c->ctx->stream);
  if (st != 0) return cfear_set_error(c->ctx, CFEAR_ERR_HIP, "ncclAllGather failed (%d)", st);
  return 0;
}

// host buffers (the generic callback signature): staged through the context's workspace
extern "C" int cfear_rccl_allgather(void* user, const void* send, void* recv, size_t bytes) {
  cfear_rccl_comm* c = (cfear_rccl_comm*)user;
  if (!c || !c->ctx || !c->nccl_comm || c->world < 1) return CFEAR_ERR_INVALID_ARGUMENT;
  cfear_ctx* ctx = c->ctx;
  char* ws = (char*)cfear_workspace(ctx, 10, bytes * ((size_t)c->world + 1));
  if (!ws) return cfear_set_error(ctx, CFEAR_ERR_HIP, "workspace allocation failed");
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(ws, send, bytes, hipMemcpyHostToDevice, ctx->stream));
  const int rc = cfear_rccl_allgather_device(user, ws, ws + bytes, bytes);
  if (rc != 0) return rc;
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(recv, ws + bytes, bytes * (size_t)c->world, hipMemcpyDeviceToHost, ctx->stream));
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

// ---- sharded callers ---------------------------------------------------------------------------------------------------
extern "C" int cfear_register_batch_sharded(cfear_ctx* ctx, const cfear_reg_job* jobs, int32_t n_jobs, const cfear_reg_params* par,
                                            int32_t rank, int32_t world, cfear_allgather_fn gather, void* user,
                                            cfear_reg_result* results) {
  if (!ctx) return CFEAR_ERR_INVALID_ARGUMENT;
  if ((!jobs && n_jobs > 0) || !par || !results) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "null argument");
  int32_t lo, hi, per;
  int rc = cfear_shard_range(n_jobs, world, rank, &lo, &hi, &per);
  if (rc != CFEAR_OK) return cfear_set_error(ctx, rc, "bad rank %d / world %d", rank, world);
  if (!gather && world != 1) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "world %d needs a gather callback", world);
  if (gather == cfear_rccl_allgather && user && ((cfear_rccl_comm*)user)->ctx == ctx) {
    // The ready-made RCCL callback: the registration kernel writes this rank's records straight into the send buffer on
    // the device, ncclAllGather moves them over xGMI, ONE device-to-host copy returns all n records (the generic route
    // would read the block back, upload it again and synchronise twice for 36 KiB).
    const size_t rbytes = (size_t)per * sizeof(cfear_reg_result), bytes = rbytes + sizeof(ShardTrailer);
    char* ws = (char*)cfear_workspace(ctx, 10, bytes * ((size_t)world + 1) + 256);
    if (!ws) return cfear_set_error(ctx, CFEAR_ERR_HIP, "workspace allocation failed");
    std::vector<char> host(bytes * (size_t)world);
    char* d_send = ws;
    char* d_recv = ws + (bytes + 255) / 256 * 256;
    int local_rc = CFEAR_OK;
    if (hipMemsetAsync(d_send, 0, bytes, ctx->stream) != hipSuccess) local_rc = CFEAR_ERR_HIP;
    if (local_rc == CFEAR_OK && hi > lo)
      local_rc = cfear_register_batch_device(ctx, jobs + lo, hi - lo, par, (cfear_reg_result*)d_send, nullptr);
    // the trailer travels with the block (a failed block is zeros + the status); written by the device so that no host
    // buffer has to outlive this call
    if (local_rc != CFEAR_OK) (void)hipMemsetAsync(d_send, 0, rbytes, ctx->stream);
    // From here to the collective nothing returns: a rank that left before ncclAllGather would leave its peers in it for
    // ever.  A trailer write that fails is reported AFTER the collective (the peers then see whatever the buffer held --
    // zeros from the memset above, i.e. status OK with the block's zero records; this rank returns the error).
    bool trailer_ok = hipMemsetD32Async((hipDeviceptr_t)(d_send + rbytes), local_rc, 1, ctx->stream) == hipSuccess;
    trailer_ok = hipMemsetD32Async((hipDeviceptr_t)(d_send + rbytes + 4), hi - lo, 1, ctx->stream) == hipSuccess && trailer_ok;
    rc = cfear_rccl_allgather_device(user, d_send, d_recv, bytes);
    if (rc != 0) return rc;
    if (!trailer_ok) { (void)hipStreamSynchronize(ctx->stream); return cfear_set_error(ctx, CFEAR_ERR_HIP, "status trailer could not be written"); }
    CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(host.data(), d_recv, bytes * (size_t)world, hipMemcpyDeviceToHost, ctx->stream));
    CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    unpack_blocks(host.data(), bytes, n_jobs, (int32_t)sizeof(cfear_reg_result), world, results);
    rc = first_rank_status(host.data(), bytes, world);
    return rc == CFEAR_OK ? rc : (local_rc != CFEAR_OK ? local_rc : cfear_set_error(ctx, rc, "a peer rank's registrations failed (%d)", rc));
  }
  std::vector<cfear_reg_result> local((size_t)std::max(hi - lo, 1));
  int local_rc = CFEAR_OK;
  if (hi > lo) local_rc = cfear_register_batch(ctx, jobs + lo, hi - lo, par, local.data());
  rc = gather_records_status(local.data(), local_rc, n_jobs, (int32_t)sizeof(cfear_reg_result), world, rank, gather, user, results);
  if (rc == CFEAR_OK || rc == local_rc) return rc;            // (this rank's own error text is already set)
  return cfear_set_error(ctx, rc, "result all_gather failed or a peer rank's registrations failed (%d)", rc);
}

extern "C" int cfear_verify_loop_candidates_sharded(cfear_ctx* ctx, const cfear_verify_job* jobs, int32_t n_jobs,
                                                    const cfear_verify_params* par, int32_t rank, int32_t world,
                                                    cfear_allgather_fn gather, void* user, cfear_verify_result* results) {
  if (!ctx) return CFEAR_ERR_INVALID_ARGUMENT;
  if ((!jobs && n_jobs > 0) || !par || !results) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "null argument");
  int32_t lo, hi;
  int rc = cfear_shard_range(n_jobs, world, rank, &lo, &hi, nullptr);
  if (rc != CFEAR_OK) return cfear_set_error(ctx, rc, "bad rank %d / world %d", rank, world);
  if (!gather && world != 1) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "world %d needs a gather callback", world);
  std::vector<cfear_verify_result> local((size_t)std::max(hi - lo, 1));
  int local_rc = CFEAR_OK;
  if (hi > lo) local_rc = cfear_verify_loop_candidates(ctx, jobs + lo, hi - lo, par, local.data());
  rc = gather_records_status(local.data(), local_rc, n_jobs, (int32_t)sizeof(cfear_verify_result), world, rank, gather, user, results);
  if (rc != CFEAR_OK) return rc == local_rc ? rc : cfear_set_error(ctx, rc, "result all_gather failed or a peer rank's verification failed (%d)", rc);
  if (world > 1) {
    // ApplyConstratins (loopclosure.cpp:261-274) over the WHOLE list: a query's candidates may straddle a rank boundary,
    // so the selection each rank made inside its block is redone -- per query sort by probability (larger first, earlier
    // candidate first on ties), accept above the threshold, every candidate or only the best
    std::vector<int> order((size_t)n_jobs);
    for (int i = 0; i < n_jobs; i++) order[(size_t)i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
      if (jobs[a].group != jobs[b].group) return jobs[a].group < jobs[b].group;
      return results[a].probability > results[b].probability;
    });
    int rank_in_group = 0;
    for (int k = 0; k < n_jobs; k++) {
      const int i = order[(size_t)k];
      rank_in_group = (k > 0 && jobs[order[(size_t)k - 1]].group == jobs[i].group) ? rank_in_group + 1 : 0;
      results[i].rank = rank_in_group;
      results[i].accepted = ((rank_in_group == 0) || par->all_candidates) && results[i].probability > par->model_threshold ? 1 : 0;
    }
  }
  return CFEAR_OK;
}
