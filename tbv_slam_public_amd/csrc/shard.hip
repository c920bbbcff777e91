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

// Every rank contributes ceil(n / world) records (its block, zero-padded); all[] receives the n real records in
// candidate order.  gather(user, send, recv, bytes): recv = the concatenation of every rank's `bytes` bytes, rank order.
extern "C" int cfear_gather_records(const void* local, int32_t n_total, int32_t record_bytes, int32_t world, int32_t rank,
                                    cfear_allgather_fn gather, void* user, void* all) {
  int32_t lo, hi, per;
  if (!all || record_bytes <= 0 || (n_total > 0 && !local && world > 0 && rank >= 0 && rank < world && n_total > rank * ((n_total + world - 1) / world)))
    return CFEAR_ERR_INVALID_ARGUMENT;
  int rc = cfear_shard_range(n_total, world, rank, &lo, &hi, &per);
  if (rc != CFEAR_OK) return rc;
  if (world == 1) { if (n_total > 0) memcpy(all, local, (size_t)n_total * record_bytes); return CFEAR_OK; }
  if (!gather) return CFEAR_ERR_INVALID_ARGUMENT;
  const size_t bytes = (size_t)per * record_bytes;
  std::vector<char> send(bytes, 0), recv(bytes * (size_t)world);
  if (hi > lo) memcpy(send.data(), local, (size_t)(hi - lo) * record_bytes);
  rc = gather(user, send.data(), recv.data(), bytes);
  if (rc != 0) return rc < 0 ? rc : CFEAR_ERR_HIP;
  for (int r = 0; r < world; r++) {                          // drop every rank's padding
    int32_t l2, h2;
    cfear_shard_range(n_total, world, r, &l2, &h2, nullptr);
    if (h2 > l2) memcpy((char*)all + (size_t)l2 * record_bytes, recv.data() + (size_t)r * bytes, (size_t)(h2 - l2) * record_bytes);
  }
  return CFEAR_OK;
}

extern "C" int cfear_register_batch_sharded(cfear_ctx* ctx, const cfear_reg_job* jobs, int32_t n_jobs, const cfear_reg_params* par,
                                            int32_t rank, int32_t world, cfear_allgather_fn gather, void* user,
                                            cfear_reg_result* results) {
  if (!ctx) return CFEAR_ERR_INVALID_ARGUMENT;
  if ((!jobs && n_jobs > 0) || !par || !results) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "null argument");
  int32_t lo, hi;
  int rc = cfear_shard_range(n_jobs, world, rank, &lo, &hi, nullptr);
  if (rc != CFEAR_OK) return cfear_set_error(ctx, rc, "bad rank %d / world %d", rank, world);
  std::vector<cfear_reg_result> local((size_t)std::max(hi - lo, 1));
  if (hi > lo) {
    rc = cfear_register_batch(ctx, jobs + lo, hi - lo, par, local.data());
    if (rc != CFEAR_OK) return rc;
  }
  rc = cfear_gather_records(local.data(), n_jobs, (int32_t)sizeof(cfear_reg_result), world, rank, gather, user, results);
  return rc == CFEAR_OK ? rc : cfear_set_error(ctx, rc, "result all_gather failed (%d)", rc);
}

extern "C" int cfear_verify_loop_candidates_sharded(cfear_ctx* ctx, const cfear_verify_job* jobs, int32_t n_jobs,
                                                    const cfear_verify_params* par, int32_t rank, int32_t world,
                                                    cfear_allgather_fn gather, void* user, cfear_verify_result* results) {
  if (!ctx) return CFEAR_ERR_INVALID_ARGUMENT;
  if ((!jobs && n_jobs > 0) || !par || !results) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "null argument");
  int32_t lo, hi;
  int rc = cfear_shard_range(n_jobs, world, rank, &lo, &hi, nullptr);
  if (rc != CFEAR_OK) return cfear_set_error(ctx, rc, "bad rank %d / world %d", rank, world);
  std::vector<cfear_verify_result> local((size_t)std::max(hi - lo, 1));
  if (hi > lo) {
    rc = cfear_verify_loop_candidates(ctx, jobs + lo, hi - lo, par, local.data());
    if (rc != CFEAR_OK) return rc;
  }
  rc = cfear_gather_records(local.data(), n_jobs, (int32_t)sizeof(cfear_verify_result), world, rank, gather, user, results);
  if (rc != CFEAR_OK) return cfear_set_error(ctx, rc, "result all_gather failed (%d)", rc);
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

extern "C" int cfear_rccl_allgather(void* user, const void* send, void* recv, size_t bytes) {
  cfear_rccl_comm* c = (cfear_rccl_comm*)user;
  if (!c || !c->ctx || !c->nccl_comm || c->world < 1) return CFEAR_ERR_INVALID_ARGUMENT;
  cfear_ctx* ctx = c->ctx;
  nccl_allgather_t ag = resolve_allgather();
  if (!ag) return cfear_set_error(ctx, CFEAR_ERR_HIP, "librccl.so / ncclAllGather not found");
  char* ws = (char*)cfear_workspace(ctx, 10, bytes * ((size_t)c->world + 1));
  if (!ws) return cfear_set_error(ctx, CFEAR_ERR_HIP, "workspace allocation failed");
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(ws, send, bytes, hipMemcpyHostToDevice, ctx->stream));
  const int st = ag(ws, ws + bytes, bytes, 0 /* ncclInt8 */, c->nccl_comm, ctx->stream);
  if (st != 0) return cfear_set_error(ctx, CFEAR_ERR_HIP, "ncclAllGather failed (%d)", st);
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(recv, ws + bytes, bytes * (size_t)c->world, hipMemcpyDeviceToHost, ctx->stream));
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}
