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
struct NcclId { char b[128]; };                                   // ncclUniqueId: 128 opaque bytes, passed by value
typedef int (*nccl_get_id_t)(NcclId*);
typedef int (*nccl_init_rank_t)(void** /*ncclComm_t*/, int, NcclId, int);
typedef int (*nccl_destroy_t)(void*);
struct Rccl {
  nccl_allgather_t allgather = nullptr;
  nccl_get_id_t get_id = nullptr;
  nccl_init_rank_t init_rank = nullptr;
  nccl_destroy_t destroy = nullptr;
};
const Rccl& rccl() {
  static Rccl r = []() {
    Rccl q;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (h) {
      q.allgather = (nccl_allgather_t)dlsym(h, "ncclAllGather");
      q.get_id = (nccl_get_id_t)dlsym(h, "ncclGetUniqueId");
      q.init_rank = (nccl_init_rank_t)dlsym(h, "ncclCommInitRank");
      q.destroy = (nccl_destroy_t)dlsym(h, "ncclCommDestroy");
    }
    return q;
  }();
  return r;
}
nccl_allgather_t resolve_allgather() { return rccl().allgather; }

int allgather_on(const cfear_rccl_comm* c, hipStream_t stream, const void* d_send, void* d_recv_all, size_t bytes) {
  nccl_allgather_t ag = resolve_allgather();
  if (!ag) return cfear_set_error(c->ctx, CFEAR_ERR_HIP, "librccl.so / ncclAllGather not found");
  const int st = ag(d_send, d_recv_all, bytes, 0 /* ncclInt8 */, c->nccl_comm, stream);
  if (st != 0) return cfear_set_error(c->ctx, CFEAR_ERR_HIP, "ncclAllGather failed (%d)", st);
  return 0;
}
}  // namespace

// A communicator of the library's own making, for hosts that have none: rank 0 calls cfear_rccl_unique_id and hands the 128
// bytes to its peers by whatever means it has (MPI, a socket, torch.distributed); every rank then calls cfear_rccl_comm_init.
extern "C" int cfear_rccl_unique_id(char id128[128]) {
  if (!id128) return CFEAR_ERR_INVALID_ARGUMENT;
  if (!rccl().get_id) return CFEAR_ERR_HIP;
  NcclId id;
  if (rccl().get_id(&id) != 0) return CFEAR_ERR_HIP;
  memcpy(id128, id.b, 128);
  return CFEAR_OK;
}

extern "C" int cfear_rccl_comm_init(cfear_ctx* ctx, const char id128[128], int32_t world, int32_t rank, cfear_rccl_comm* out) {
  if (!ctx) return CFEAR_ERR_INVALID_ARGUMENT;
  if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "bad rank %d / world %d", rank, world);
  if (!rccl().init_rank) return cfear_set_error(ctx, CFEAR_ERR_HIP, "librccl.so / ncclCommInitRank not found");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  NcclId id;
  memcpy(id.b, id128, 128);
  void* comm = nullptr;
  const int st = rccl().init_rank(&comm, world, id, rank);
  if (st != 0 || !comm) return cfear_set_error(ctx, CFEAR_ERR_HIP, "ncclCommInitRank failed (%d)", st);
  out->ctx = ctx; out->nccl_comm = comm; out->world = world; out->pad = 0;
  return CFEAR_OK;
}

extern "C" int cfear_rccl_comm_destroy(cfear_rccl_comm* c) {
  if (!c || !c->nccl_comm) return CFEAR_OK;
  if (c->ctx) { (void)hipSetDevice(c->ctx->device); (void)hipStreamSynchronize(c->ctx->stream); }
  if (rccl().destroy) (void)rccl().destroy(c->nccl_comm);
  c->nccl_comm = nullptr;
  return CFEAR_OK;
}

// device buffers, enqueued on the communicator's context stream, NOT synchronised
extern "C" int cfear_rccl_allgather_device(void* user, const void* d_send, void* d_recv_all, size_t bytes) {
  cfear_rccl_comm* c = (cfear_rccl_comm*)user;
  if (!c || !c->ctx || !c->nccl_comm || c->world < 1) return CFEAR_ERR_INVALID_ARGUMENT;
  return allgather_on(c, c->ctx->stream, d_send, d_recv_all, bytes);
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

// ---- pipelined candidate steps ----------------------------------------------------------------------------------------
// A loop-closure thread hands over candidate batches one after another (loopclosure.cpp:658-721 walks them as the odometry
// produces nodes).  One step of one rank is a chain  upload -> expand -> matcher -> all_gather -> read-back  that occupies the
// GPU for the matcher only; a rank's block of an 8-way sharded batch (512 candidates) is a quarter of a chip's worth of
// workgroups and ~0.13 ms of kernel, so whatever sits around the kernel decides the rate.  The pipe keeps `depth` steps in
// flight: the matcher of step k + 1 runs on the context's stream while the collective and the device-to-host copy of step k
// run on the pipe's exchange stream, and the expand kernel of step k + 1 on a preparation stream beside them; submit never waits,
// collect waits on ONE event.  Optionally a slot's matcher launches are captured once into a hipGraph and replayed while the
// block's size, geometry and parameters stay the same.
struct cfear_candidate_pipe {
  cfear_ctx* ctx = nullptr;
  const cfear_scan_table* table = nullptr;
  cfear_rccl_comm comm{};                 // nccl_comm == nullptr: no collective (world 1)
  int rank = 0, world = 1, depth = 2, max_total = 0, per_cap = 0;
  hipStream_t xstream = nullptr;          // exchange stream: all_gather + read-back
  hipStream_t pstream = nullptr;          // preparation stream: step k + 1's expand kernel runs beside step k's matcher
  int use_graph = 0, timing = 0;
  int64_t next_ticket = 0;
  double exchange_ms = 0.0;               // CFEAR_PIPE_TIMING: sum over collected steps of (all_gather + read-back) on the exchange stream
  int64_t collected = 0;
  struct Slot {
    cfear_candidate* h_cands = nullptr;   // pinned [per_cap]
    char* h_recv = nullptr;               // pinned [world][per_cap * 72 + 8]
    char* d_jobs = nullptr;               // device [per_cap] job records (reg_job_stride(2) bytes each)
    char* d_send = nullptr;               // device [per_cap * 72 + 8]
    char* d_recv = nullptr;               // device [world][...]
    hipEvent_t prepared = nullptr, computed = nullptr, xbegin = nullptr, done = nullptr;
    int64_t ticket = -1;                  // the step in this slot (-1: free)
    int n_total = 0, status = CFEAR_OK;
    hipGraphExec_t exec = nullptr;        // the captured compute chain ...
    int g_n = -1, g_per = -1;             // ... of a block of g_n candidates (g_per slots per rank) with parameters g_par,
    cfear_reg_params g_par{};             //     whose nodes hold the context's workspaces as they were at capture
    const void* g_ws7 = nullptr;
    int g_pairs_cap = 0, g_hint = 0;
  };
  std::vector<Slot> slots;
  size_t block_bytes() const { return (size_t)per_cap * sizeof(cfear_reg_result) + sizeof(ShardTrailer); }
};


extern "C" int cfear_candidate_pipe_destroy(cfear_candidate_pipe* p) {
  if (!p) return CFEAR_OK;
  (void)hipSetDevice(p->ctx->device);
  (void)hipStreamSynchronize(p->ctx->stream);
  if (p->xstream) (void)hipStreamSynchronize(p->xstream);
  if (p->pstream) (void)hipStreamSynchronize(p->pstream);
  for (auto& s : p->slots) {
    if (s.exec) (void)hipGraphExecDestroy(s.exec);
    if (s.h_cands) (void)hipHostFree(s.h_cands);
    if (s.h_recv) (void)hipHostFree(s.h_recv);
    if (s.d_send) (void)hipFree(s.d_send);
    if (s.d_jobs) (void)hipFree(s.d_jobs);
    if (s.prepared) (void)hipEventDestroy(s.prepared);
    if (s.d_recv) (void)hipFree(s.d_recv);
    if (s.computed) (void)hipEventDestroy(s.computed);
    if (s.xbegin) (void)hipEventDestroy(s.xbegin);
    if (s.done) (void)hipEventDestroy(s.done);
  }
  if (p->xstream) (void)hipStreamDestroy(p->xstream);
  if (p->pstream) (void)hipStreamDestroy(p->pstream);
  delete p;
  return CFEAR_OK;
}

extern "C" int cfear_candidate_pipe_create(cfear_ctx* ctx, const cfear_scan_table* table, int32_t max_candidates, int32_t rank,
                                           int32_t world, const cfear_rccl_comm* comm, int32_t depth, int32_t flags,
                                           cfear_candidate_pipe** out) {
  if (!ctx) return CFEAR_ERR_INVALID_ARGUMENT;
  if (!table || !out || max_candidates < 1 || depth < 1 || depth > 16 || world < 1 || rank < 0 || rank >= world)
    return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "bad pipe geometry (n %d, depth %d, rank %d / world %d)", max_candidates, depth, rank, world);
  if (world > 1 && (!comm || !comm->nccl_comm || comm->world != world))
    return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "world %d needs a cfear_rccl_comm of that size", world);
  *out = nullptr;
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  cfear_candidate_pipe* p = new cfear_candidate_pipe();
  p->ctx = ctx; p->table = table; p->rank = rank; p->world = world; p->depth = depth; p->max_total = max_candidates;
  p->per_cap = (max_candidates + world - 1) / world;
  p->use_graph = (flags & CFEAR_PIPE_GRAPH) ? 1 : 0;
  p->timing = (flags & CFEAR_PIPE_TIMING) ? 1 : 0;
  if (comm && comm->nccl_comm) { p->comm = *comm; p->comm.ctx = ctx; }
  p->slots.resize((size_t)depth);
  const size_t bb = (p->block_bytes() + 255) / 256 * 256;
  bool ok = hipStreamCreateWithFlags(&p->xstream, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipStreamCreateWithFlags(&p->pstream, hipStreamNonBlocking) == hipSuccess;
  for (auto& s : p->slots) {
    ok = ok && hipHostMalloc((void**)&s.h_cands, (size_t)p->per_cap * sizeof(cfear_candidate), hipHostMallocDefault) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&s.h_recv, bb * (size_t)world, hipHostMallocDefault) == hipSuccess;
    ok = ok && hipMalloc((void**)&s.d_send, bb) == hipSuccess;
    ok = ok && hipMalloc((void**)&s.d_jobs, (size_t)p->per_cap * reg_job_stride(2) + 256) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&s.prepared, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipMalloc((void**)&s.d_recv, bb * (size_t)world) == hipSuccess;
    ok = ok && hipMemsetAsync(s.d_send, 0, bb, ctx->stream) == hipSuccess;        // padding slots: zero once, never written
    ok = ok && hipEventCreateWithFlags(&s.computed, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&s.done, p->timing ? hipEventDefault : hipEventDisableTiming) == hipSuccess;
    if (p->timing) ok = ok && hipEventCreate(&s.xbegin) == hipSuccess;
  }
  ok = ok && hipStreamSynchronize(ctx->stream) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    cfear_candidate_pipe_destroy(p);
    return cfear_set_error(ctx, CFEAR_ERR_HIP, "candidate pipe: buffer / stream creation failed");
  }
  *out = p;
  return CFEAR_OK;
}

extern "C" int cfear_candidate_pipe_submit(cfear_candidate_pipe* p, const cfear_candidate* cands, int32_t n_total,
                                           const cfear_reg_params* par, int64_t* ticket) {
  if (!p) return CFEAR_ERR_INVALID_ARGUMENT;
  cfear_ctx* ctx = p->ctx;
  if (!ticket || !par || n_total < 0 || n_total > p->max_total || (n_total > 0 && !cands))
    return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "candidate pipe: bad batch (n %d of at most %d)", n_total, p->max_total);
  cfear_candidate_pipe::Slot& s = p->slots[(size_t)(p->next_ticket % p->depth)];
  if (s.ticket >= 0) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "candidate pipe: %d steps in flight, collect ticket %lld first", p->depth, (long long)s.ticket);
  int rc = cfear_check_reg_params(ctx, par);
  if (rc != CFEAR_OK) return rc;
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int32_t lo, hi, per;
  cfear_shard_range(n_total, p->world, p->rank, &lo, &hi, &per);
  const int n = hi - lo;
  const size_t rbytes = (size_t)per * sizeof(cfear_reg_result), bytes = rbytes + sizeof(ShardTrailer);
  int32_t* d_trailer = (int32_t*)(s.d_send + rbytes);
  // From here on nothing returns before the collective: peers are on their way into it (see gather_records_status).
  int local_rc = CFEAR_OK;
  if (n > 0) {
    // expand on the preparation stream: it needs nothing of the steps in flight (its job buffer is the slot's own), so step
    // k + 1's expand runs beside step k's matcher and the context's stream carries matcher after matcher
    // (with nothing in flight there is no matcher to run beside: the expand goes straight onto the compute stream, one event hop less)
    bool idle = true;
    for (auto& o : p->slots) idle = idle && o.ticket < 0;
    CandGeometry geom;
    local_rc = cfear_candidates_expand(ctx, idle ? ctx->stream : p->pstream, p->table, cands + lo, n, par, s.h_cands, s.d_jobs, d_trailer, CFEAR_OK, &geom);
    if (!idle && local_rc == CFEAR_OK && (hipEventRecord(s.prepared, p->pstream) != hipSuccess || hipStreamWaitEvent(ctx->stream, s.prepared, 0) != hipSuccess))
      local_rc = cfear_set_error(ctx, CFEAR_ERR_HIP, "candidate pipe: event between the preparation and the compute stream failed");
    const int hint_bits = (geom.hint.small_pairs ? 1 : 0) | (geom.hint.big_pass ? 2 : 0) | (geom.hint.whole_cu ? 4 : 0);
    const bool graph = p->use_graph && ctx->profile == 0;         // (per-kernel events do not go into a capture)
    const bool replay = graph && s.exec && s.g_n == n && memcmp(&s.g_par, par, sizeof(*par)) == 0 && s.g_ws7 == ctx->ws[7].p &&
                        s.g_pairs_cap == geom.pairs_cap && s.g_hint == hint_bits;
    if (local_rc != CFEAR_OK) {
    } else if (replay) {
      if (hipGraphLaunch(s.exec, ctx->stream) != hipSuccess) local_rc = cfear_set_error(ctx, CFEAR_ERR_HIP, "hipGraphLaunch failed");
    } else if (graph && s.g_n != -2) {
      // first step of this shape in this slot: run the matcher once directly (workspaces and LDS attributes settle outside a
      // capture), then capture the same launches for the steps to come
      local_rc = cfear_candidates_match(ctx, s.d_jobs, n, par, &geom, (cfear_reg_result*)s.d_send);
      if (local_rc == CFEAR_OK && hipStreamSynchronize(ctx->stream) == hipSuccess) {
        if (s.exec) { (void)hipGraphExecDestroy(s.exec); s.exec = nullptr; }
        hipGraph_t g = nullptr;
        bool cap = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
        int crc = cap ? cfear_candidates_match(ctx, s.d_jobs, n, par, &geom, (cfear_reg_result*)s.d_send) : CFEAR_ERR_HIP;
        if (cap && hipStreamEndCapture(ctx->stream, &g) != hipSuccess) crc = CFEAR_ERR_HIP;
        if (crc == CFEAR_OK && g && hipGraphInstantiate(&s.exec, g, nullptr, nullptr, 0) == hipSuccess) {
          s.g_n = n; s.g_per = per; s.g_par = *par; s.g_ws7 = ctx->ws[7].p; s.g_pairs_cap = geom.pairs_cap; s.g_hint = hint_bits;
        } else {
          (void)hipGetLastError();
          s.exec = nullptr; s.g_n = -2;                           // capture is not available here: direct launches from now on
        }
        if (g) (void)hipGraphDestroy(g);
        // (the direct run above already produced this step's records)
      }
    } else {
      local_rc = cfear_candidates_match(ctx, s.d_jobs, n, par, &geom, (cfear_reg_result*)s.d_send);
    }
  }
  if (n == 0 || local_rc != CFEAR_OK) {                           // an empty or failed block: zeros + the status
    (void)hipMemsetAsync(s.d_send, 0, rbytes, ctx->stream);
    (void)hipMemsetD32Async((hipDeviceptr_t)d_trailer, local_rc, 1, ctx->stream);
    (void)hipMemsetD32Async((hipDeviceptr_t)(d_trailer + 1), 0, 1, ctx->stream);
  }
  bool ok = hipEventRecord(s.computed, ctx->stream) == hipSuccess;
  ok = ok && hipStreamWaitEvent(p->xstream, s.computed, 0) == hipSuccess;
  if (p->timing) ok = ok && hipEventRecord(s.xbegin, p->xstream) == hipSuccess;
  const char* d_all = s.d_send;
  if (p->comm.nccl_comm) {
    rc = allgather_on(&p->comm, p->xstream, s.d_send, s.d_recv, bytes);
    if (rc != 0) return rc;
    d_all = s.d_recv;
  }
  ok = ok && hipMemcpyAsync(s.h_recv, d_all, bytes * (size_t)p->world, hipMemcpyDeviceToHost, p->xstream) == hipSuccess;
  ok = ok && hipEventRecord(s.done, p->xstream) == hipSuccess;
  // the next step's kernels write other slots' buffers, but a step that comes back to THIS slot must find its exchange over:
  // collect() waits for `done` before it frees the slot, so no device-side edge is needed
  if (!ok) { (void)hipGetLastError(); (void)hipStreamSynchronize(ctx->stream); (void)hipStreamSynchronize(p->xstream);
             return cfear_set_error(ctx, CFEAR_ERR_HIP, "candidate pipe: enqueue failed"); }
  s.ticket = p->next_ticket; s.n_total = n_total; s.status = local_rc;
  *ticket = p->next_ticket++;
  return CFEAR_OK;
}

extern "C" int cfear_candidate_pipe_collect(cfear_candidate_pipe* p, int64_t ticket, cfear_reg_result* results) {
  if (!p) return CFEAR_ERR_INVALID_ARGUMENT;
  cfear_ctx* ctx = p->ctx;
  if (!results || ticket < 0) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "null argument");
  cfear_candidate_pipe::Slot& s = p->slots[(size_t)(ticket % p->depth)];
  if (s.ticket != ticket) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "candidate pipe: ticket %lld is not in flight", (long long)ticket);
  CFEAR_HIP_CHECK(ctx, hipEventSynchronize(s.done));
  const int per = (s.n_total + p->world - 1) / p->world;
  const size_t bytes = (size_t)per * sizeof(cfear_reg_result) + sizeof(ShardTrailer);
  if (p->timing) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s.xbegin, s.done) == hipSuccess) p->exchange_ms += ms;
  }
  p->collected++;
  unpack_blocks(s.h_recv, bytes, s.n_total, (int32_t)sizeof(cfear_reg_result), p->world, results);
  const int rc = first_rank_status(s.h_recv, bytes, p->world);
  const int mine = s.status;
  s.ticket = -1;
  if (rc == CFEAR_OK) {
    // every rank's trailer carries the number of records its kernel wrote: together they must be the batch (a collective
    // that did not deliver a peer's block leaves that block's trailer at whatever the receive buffer held)
    int64_t got = 0;
    for (int r = 0; r < p->world; r++) {
      ShardTrailer t;
      memcpy(&t, s.h_recv + (size_t)r * bytes + (bytes - sizeof(ShardTrailer)), sizeof(t));
      got += t.n_records;
    }
    if (got != s.n_total) return cfear_set_error(ctx, CFEAR_ERR_HIP, "candidate pipe: the exchange returned %lld of %d records", (long long)got, s.n_total);
    return CFEAR_OK;
  }
  return mine != CFEAR_OK ? mine : cfear_set_error(ctx, rc, "a peer rank's registrations failed (%d)", rc);
}

extern "C" int cfear_candidate_pipe_stats(const cfear_candidate_pipe* p, double* exchange_ms_sum, int64_t* steps_collected, int32_t* graph_slots) {
  if (!p) return CFEAR_ERR_INVALID_ARGUMENT;
  if (exchange_ms_sum) *exchange_ms_sum = p->exchange_ms;
  if (steps_collected) *steps_collected = p->collected;
  if (graph_slots) { int g = 0; for (auto& s : p->slots) g += s.exec != nullptr; *graph_slots = g; }
  return CFEAR_OK;
}
