// Multi-GPU search: the index sharded by node-id range, one rank per GPU.
//
// Replaces MultiIndexSearcher::search (src/core/search.rs:211-237) / the product's cross-index merge
// (src/indexer/service.rs:775-801) with one sub-index per rank: shard search on the index's lanes,
// ONE all-gather of the packed per-rank answer records (RCCL over xGMI) on a side stream behind a
// device event of the search, then merge_topk over the gathered records on the same stream.
// RCCL is loaded on first use (dlopen), so single-GPU users of the library never touch it.
#include "common.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>

namespace {

// ---- RCCL, bound at first use: the copy already mapped into the process (a host that brought its
// own, e.g. PyTorch's bundled one) or the ROCm installation's ----
struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string error;
};

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* env = getenv("ISL_RCCL_LIB");
    const char* resident[] = {"librccl.so", "librccl.so.1"};
    const char* load[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    if (env && *env) r.handle = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
    for (const char* n : resident)
      if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char* n : load)
      if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!r.handle) {
      const char* e = dlerror();
      r.error = std::string("librccl.so could not be loaded: ") + (e ? e : "not found");
      return;
    }
    auto sym = [&](const char* name) -> void* {
      void* p = dlsym(r.handle, name);
      if (!p && r.error.empty()) r.error = std::string("librccl.so lacks ") + name;
      return p;
    };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
  });
  return &r;
}

isl_status rccl_ready(Rccl*& out) {
  out = rccl();
  if (!out->error.empty()) return isl::fail(ISL_ERR_DEVICE, "%s", out->error.c_str());
  return ISL_OK;
}

#define ISL_NCCL(r, expr)                                                                          \
  do {                                                                                             \
    ncclResult_t _n = (expr);                                                                      \
    if (_n != ncclSuccess)                                                                         \
      return ::isl::fail(ISL_ERR_DEVICE, "%s failed: %s", #expr, (r)->GetErrorString(_n));         \
  } while (0)

static_assert(ISL_SHARD_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

struct Slot {
  bool busy = false;
  uint64_t handle = 0, token = 0;
  uint64_t nq = 0, k = 0;
  uint8_t* rec = nullptr;    // this rank's record, written in place by the search kernels
  uint8_t* gath = nullptr;   // [world][B]
  uint64_t* ids = nullptr;   // merged answers
  float* dist = nullptr;
  uint32_t* src = nullptr;
  uint32_t* cnt = nullptr;
  uint8_t* h_rec = nullptr;  // host transport: pinned mirrors
  uint8_t* h_gath = nullptr;
  hipEvent_t done = nullptr;  // behind the merge of the batch in this slot
  bool recorded = false;      // ... once the submit got that far
};

}  // namespace

struct isl_shard_group {
  int32_t device = 0, world = 1, rank = 0;
  ncclComm_t comm = nullptr;
  isl_shard_allgather_fn host_fn = nullptr;
  void* host_user = nullptr;
};

struct isl_sharded_searcher {
  const isl_index* idx = nullptr;
  isl_shard_group* grp = nullptr;
  int32_t device = 0, world = 1, rank = 0;
  uint64_t n_total = 0;
  std::vector<uint64_t> id_base;
  uint64_t* d_base = nullptr;
  uint32_t* d_flags = nullptr;
  hipStream_t side = nullptr;
  int32_t depth = 1;
  std::vector<Slot> slots;
  uint64_t cap_nq = 0, cap_k = 0;
  uint64_t next_slot = 0, next_handle = 1;
  // staging of the host-buffer entry point
  float* d_q = nullptr;
  uint64_t d_q_bytes = 0;
  std::mutex mu;
};

namespace {

void free_slot(Slot& s) {
  void* dev[] = {s.rec, s.gath, s.ids, s.dist, s.src, s.cnt};
  for (void* p : dev)
    if (p) (void)hipFree(p);
  if (s.h_rec) (void)hipHostFree(s.h_rec);
  if (s.h_gath) (void)hipHostFree(s.h_gath);
  if (s.done) (void)hipEventDestroy(s.done);
  s = Slot{};
}

isl_status size_slots(isl_sharded_searcher* s, uint64_t nq, uint64_t k) {
  if (nq <= s->cap_nq && k <= s->cap_k && !s->slots.empty()) return ISL_OK;
  for (const Slot& sl : s->slots)
    if (sl.busy)
      return isl::fail(ISL_ERR_SEARCH, "Search error: a larger batch than isl_sharded_prepare sized the buffers for, "
                       "while batches are in flight");
  const uint64_t cnq = std::max(nq, s->cap_nq), ck = std::max<uint64_t>(std::max(k, s->cap_k), 1);
  for (Slot& sl : s->slots) free_slot(sl);
  s->slots.assign((size_t)s->depth, Slot{});
  s->cap_nq = s->cap_k = 0;
  const uint64_t B = isl_shard_record_bytes(cnq, ck);
  for (Slot& sl : s->slots) {
    ISL_HIP(hipMalloc(&sl.rec, B));
    ISL_HIP(hipMalloc(&sl.gath, B * (uint64_t)s->world));
    ISL_HIP(hipMalloc(&sl.ids, cnq * ck * 8));
    ISL_HIP(hipMalloc(&sl.dist, cnq * ck * 4));
    ISL_HIP(hipMalloc(&sl.src, cnq * ck * 4));
    ISL_HIP(hipMalloc(&sl.cnt, cnq * 4));
    ISL_HIP(hipMemset(sl.rec, 0, B));
    if (s->grp && s->grp->host_fn) {
      ISL_HIP(hipHostMalloc(&sl.h_rec, B));
      ISL_HIP(hipHostMalloc(&sl.h_gath, B * (uint64_t)s->world));
    }
    ISL_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
  }
  s->cap_nq = cnq;
  s->cap_k = ck;
  return ISL_OK;
}

}  // namespace

extern "C" {

isl_status isl_shard_unique_id(uint8_t id[ISL_SHARD_UNIQUE_ID_BYTES]) {
  if (!id) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "id is NULL");
  Rccl* r = nullptr;
  ISL_TRY(rccl_ready(r));
  ncclUniqueId u;
  ISL_NCCL(r, r->GetUniqueId(&u));
  memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
  return ISL_OK;
}

isl_status isl_shard_group_create(int32_t device, int32_t world, int32_t rank,
                                  const uint8_t id[ISL_SHARD_UNIQUE_ID_BYTES], isl_shard_group** out) {
  if (!out || !id) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL argument");
  *out = nullptr;
  if (world < 1 || world > 64 || rank < 0 || rank >= world)
    return isl::fail(ISL_ERR_INVALID_ARGUMENT, "rank %d of %d ranks (at most 64)", rank, world);
  ISL_TRY(isl::use_device(device));
  Rccl* r = nullptr;
  ISL_TRY(rccl_ready(r));
  ncclUniqueId u;
  memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
  ncclComm_t comm = nullptr;
  ISL_NCCL(r, r->CommInitRank(&comm, world, u, rank));
  auto* g = new isl_shard_group;
  g->device = device;
  g->world = world;
  g->rank = rank;
  g->comm = comm;
  *out = g;
  return ISL_OK;
}

isl_status isl_shard_group_create_host(int32_t device, int32_t world, int32_t rank,
                                       isl_shard_allgather_fn allgather, void* user, isl_shard_group** out) {
  if (!out || !allgather) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL argument");
  *out = nullptr;
  if (world < 1 || world > 64 || rank < 0 || rank >= world)
    return isl::fail(ISL_ERR_INVALID_ARGUMENT, "rank %d of %d ranks (at most 64)", rank, world);
  ISL_TRY(isl::use_device(device));
  auto* g = new isl_shard_group;
  g->device = device;
  g->world = world;
  g->rank = rank;
  g->host_fn = allgather;
  g->host_user = user;
  *out = g;
  return ISL_OK;
}

isl_status isl_shard_group_info(const isl_shard_group* grp, int32_t* world, int32_t* rank, int32_t* comm_ranks,
                                int32_t* is_rccl) {
  if (!grp) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "group is NULL");
  if (world) *world = grp->world;
  if (rank) *rank = grp->rank;
  if (is_rccl) *is_rccl = grp->comm ? 1 : 0;
  if (comm_ranks) {
    *comm_ranks = grp->world;
    if (grp->comm) {
      Rccl* r = nullptr;
      ISL_TRY(rccl_ready(r));
      int n = 0;
      ISL_NCCL(r, r->CommCount(grp->comm, &n));
      *comm_ranks = n;
    }
  }
  return ISL_OK;
}

void isl_shard_group_free(isl_shard_group* grp) {
  if (!grp) return;
  if (grp->comm) {
    (void)hipSetDevice(grp->device);
    Rccl* r = rccl();
    if (r->CommDestroy) (void)r->CommDestroy(grp->comm);
  }
  delete grp;
}

isl_status isl_sharded_searcher_new(const isl_index* shard, isl_shard_group* grp, uint64_t n_total,
                                    const uint64_t* id_base, int32_t depth, isl_sharded_searcher** out) {
  if (!out || !shard) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL argument");
  *out = nullptr;
  if (depth < 1 || depth > isl::kSearchLanes)
    return isl::fail(ISL_ERR_INVALID_ARGUMENT, "depth must be 1..%d", isl::kSearchLanes);
  if (shard->device < 0) return isl::fail(ISL_ERR_DEVICE, "the shard index is not resident on a device (isl_index_upload)");
  if (grp && grp->device != shard->device)
    return isl::fail(ISL_ERR_INVALID_ARGUMENT, "group and shard index live on different devices");
  ISL_TRY(isl::use_device(shard->device));
  auto* s = new isl_sharded_searcher;
  s->idx = shard;
  s->grp = grp;
  s->device = shard->device;
  s->world = grp ? grp->world : 1;
  s->rank = grp ? grp->rank : 0;
  s->n_total = n_total;
  s->depth = depth;
  s->id_base.resize((size_t)s->world);
  for (int r = 0; r < s->world; ++r)
    s->id_base[(size_t)r] = id_base ? id_base[r] : n_total * (uint64_t)r / (uint64_t)s->world;
  auto bail = [&](isl_status st) { isl_sharded_searcher_free(s); return st; };
  if (hipMalloc(&s->d_base, (size_t)s->world * 8) != hipSuccess || hipMalloc(&s->d_flags, 4) != hipSuccess ||
      hipMemcpy(s->d_base, s->id_base.data(), (size_t)s->world * 8, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemset(s->d_flags, 0, 4) != hipSuccess ||
      hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking) != hipSuccess)
    return bail(isl::fail(ISL_ERR_DEVICE, "device set-up of the sharded searcher failed: %s",
                          hipGetErrorString(hipGetLastError())));
  *out = s;
  return ISL_OK;
}

void isl_sharded_searcher_free(isl_sharded_searcher* s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  if (s->side) (void)hipStreamSynchronize(s->side);
  for (Slot& sl : s->slots) {
    // a batch nobody asked the result of: its search still holds a lane of the index
    if (sl.busy && sl.token) (void)isl_search_wait(s->idx, sl.token);
    free_slot(sl);
  }
  if (s->d_base) (void)hipFree(s->d_base);
  if (s->d_flags) (void)hipFree(s->d_flags);
  if (s->d_q) (void)hipFree(s->d_q);
  if (s->side) (void)hipStreamDestroy(s->side);
  delete s;
}

isl_status isl_sharded_prepare(isl_sharded_searcher* s, uint64_t max_nq, uint64_t max_k, uint64_t max_ef) {
  if (!s) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "searcher is NULL");
  if (max_nq == 0) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "max_nq out of range");
  ISL_TRY(isl::use_device(s->device));
  std::lock_guard<std::mutex> lock(s->mu);
  ISL_TRY(size_slots(s, max_nq, max_k));
  if (s->idx->num_nodes)
    ISL_TRY(isl_index_prepare(const_cast<isl_index*>(s->idx), max_nq, max_ef, max_k, s->depth));
  if (s->grp && s->grp->comm) {
    // the communicator's first collective sets up its channels: do it here, not in the first batch
    Rccl* r = nullptr;
    ISL_TRY(rccl_ready(r));
    Slot& sl = s->slots[0];
    ISL_NCCL(r, r->AllGather(sl.rec, sl.gath, 16, ncclUint8, s->grp->comm, s->side));
    ISL_HIP(hipStreamSynchronize(s->side));
  }
  return ISL_OK;
}

isl_status isl_sharded_submit(isl_sharded_searcher* s, const float* d_queries, uint64_t nq, uint64_t d,
                              uint64_t k, uint64_t ef, void* stream, uint64_t* handle) {
  if (!s || !handle) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL argument");
  *handle = 0;
  if (nq == 0 || k == 0) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "nq and k must be positive");
  ISL_TRY(isl::use_device(s->device));
  std::lock_guard<std::mutex> lock(s->mu);
  ISL_TRY(size_slots(s, nq, k));
  // slots go round in submission order: a completed batch's answers stay untouched until `depth`
  // further batches have been submitted
  Slot& sl = s->slots[(size_t)(s->next_slot % (uint64_t)s->depth)];
  if (sl.busy)
    return isl::fail(ISL_ERR_SEARCH, "Search error: %d sharded batches already in flight; isl_sharded_result one first",
                     s->depth);
  const uint64_t B = isl_shard_record_bytes(nq, k);
  uint64_t tok = 0;
  ISL_TRY(isl_search_batch_device_async(s->idx, d_queries, nq, d, k, ef, (uint64_t*)sl.rec, (float*)(sl.rec + nq * k * 8),
                                        (uint32_t*)(sl.rec + nq * k * 12), stream, &tok));
  // From here on the batch exists: every path below leaves the slot busy with the token in it, so
  // that isl_sharded_result (or _free) completes the search and releases its lane.
  sl.busy = true;
  sl.token = tok;
  sl.nq = nq;
  sl.k = k;
  sl.handle = s->next_handle++;
  sl.recorded = false;
  s->next_slot++;
  *handle = sl.handle;
  if (tok) {
    ISL_TRY(isl_search_stream_wait(s->idx, tok, s->side));
  } else {
    // an empty shard answered at once (counts zeroed on the NULL stream, leann.rs:875-877)
    ISL_HIP(hipStreamSynchronize(nullptr));
  }
  if (s->grp && s->grp->comm) {  // (also with a single rank: the communicator the caller set up is used)
    Rccl* r = nullptr;
    ISL_TRY(rccl_ready(r));
    ISL_NCCL(r, r->AllGather(sl.rec, sl.gath, B, ncclUint8, s->grp->comm, s->side));
  } else if (s->world == 1) {
    ISL_HIP(hipMemcpyAsync(sl.gath, sl.rec, B, hipMemcpyDeviceToDevice, s->side));
  } else {
    ISL_HIP(hipMemcpyAsync(sl.h_rec, sl.rec, B, hipMemcpyDeviceToHost, s->side));
    ISL_HIP(hipStreamSynchronize(s->side));
    const int32_t rc = s->grp->host_fn(s->grp->host_user, sl.h_rec, sl.h_gath, B);
    if (rc != 0) return isl::fail(ISL_ERR_IO, "IO error: the host all-gather callback returned %d", rc);
    ISL_HIP(hipMemcpyAsync(sl.gath, sl.h_gath, B * (uint64_t)s->world, hipMemcpyHostToDevice, s->side));
  }
  ISL_TRY(isl_merge_topk_packed_async((uint64_t)s->world, nq, k, sl.gath, B, s->d_base, k, sl.ids, sl.dist, sl.src, sl.cnt,
                                      s->d_flags, s->device, s->side));
  ISL_HIP(hipEventRecord(sl.done, s->side));
  sl.recorded = true;
  return ISL_OK;
}

isl_status isl_sharded_result(isl_sharded_searcher* s, uint64_t handle, const uint64_t** d_ids,
                              const float** d_dist, const uint32_t** d_src, const uint32_t** d_count,
                              isl_search_stats* stats) {
  if (!s) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "searcher is NULL");
  ISL_TRY(isl::use_device(s->device));
  Slot* sl = nullptr;
  {
    std::lock_guard<std::mutex> lock(s->mu);
    for (Slot& c : s->slots)
      if (c.busy && c.handle == handle) { sl = &c; break; }
  }
  if (!sl || handle == 0) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "unknown or already completed sharded batch");
  if (stats) *stats = isl_search_stats{};
  // per-query failures of this rank's shard surface here; the exchange and the merge of the batch are
  // waited for in either case -- the slot must not go back while they still read its buffers
  const isl_status st = sl->token ? isl_search_wait_stats(s->idx, sl->token, stats) : ISL_OK;
  const isl::ErrorRecord keep = isl::last_error();
  // (the batch's own event: the side stream also carries the exchanges of the batches behind it)
  const hipError_t e = sl->recorded ? hipEventSynchronize(sl->done) : hipStreamSynchronize(s->side);
  {
    std::lock_guard<std::mutex> lock(s->mu);
    sl->busy = false;
    sl->token = 0;
  }
  if (st != ISL_OK) { isl::last_error() = keep; return st; }
  if (e != hipSuccess) return isl::fail(ISL_ERR_DEVICE, "shard exchange failed: %s", hipGetErrorString(e));
  if (d_ids) *d_ids = sl->ids;
  if (d_dist) *d_dist = sl->dist;
  if (d_src) *d_src = sl->src;
  if (d_count) *d_count = sl->cnt;
  return ISL_OK;
}

isl_status isl_sharded_flags(isl_sharded_searcher* s, uint32_t* flags) {
  if (!s || !flags) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL argument");
  ISL_TRY(isl::use_device(s->device));
  ISL_HIP(hipStreamSynchronize(s->side));
  ISL_HIP(hipMemcpy(flags, s->d_flags, 4, hipMemcpyDeviceToHost));
  return ISL_OK;
}

isl_status isl_sharded_search_batch(isl_sharded_searcher* s, const float* queries, uint64_t nq, uint64_t d,
                                    uint64_t k, uint64_t ef, uint64_t* out_ids, float* out_dist,
                                    uint32_t* out_src, uint32_t* out_count) {
  if (!s) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "searcher is NULL");
  if (nq == 0) return ISL_OK;
  if (!queries || !out_count || (k && (!out_ids || !out_dist))) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL buffer");
  if (k == 0) { memset(out_count, 0, nq * 4); return ISL_OK; }
  ISL_TRY(isl::use_device(s->device));
  {
    std::lock_guard<std::mutex> lock(s->mu);
    const uint64_t bytes = nq * d * 4;
    if (s->d_q_bytes < bytes) {
      if (s->d_q) (void)hipFree(s->d_q);
      s->d_q = nullptr;
      s->d_q_bytes = 0;
      ISL_HIP(hipMalloc(&s->d_q, bytes));
      s->d_q_bytes = bytes;
    }
  }
  ISL_HIP(hipMemcpy(s->d_q, queries, nq * d * 4, hipMemcpyHostToDevice));
  uint64_t h = 0;
  const isl_status sub = isl_sharded_submit(s, s->d_q, nq, d, k, ef, nullptr, &h);
  const isl::ErrorRecord keep = isl::last_error();
  const uint64_t *ids = nullptr; const float* dist = nullptr; const uint32_t *src = nullptr, *cnt = nullptr;
  if (h) {
    const isl_status st = isl_sharded_result(s, h, &ids, &dist, &src, &cnt, nullptr);
    if (sub == ISL_OK) ISL_TRY(st);
  }
  if (sub != ISL_OK) { isl::last_error() = keep; return sub; }
  ISL_HIP(hipMemcpy(out_ids, ids, nq * k * 8, hipMemcpyDeviceToHost));
  ISL_HIP(hipMemcpy(out_dist, dist, nq * k * 4, hipMemcpyDeviceToHost));
  if (out_src) ISL_HIP(hipMemcpy(out_src, src, nq * k * 4, hipMemcpyDeviceToHost));
  ISL_HIP(hipMemcpy(out_count, cnt, nq * 4, hipMemcpyDeviceToHost));
  uint32_t flags = 0;
  ISL_HIP(hipMemcpy(&flags, s->d_flags, 4, hipMemcpyDeviceToHost));
  if (flags & 1u) return isl::fail(ISL_ERR_SEARCH, "Search error: NaN score in merge (the reference panics here)");
  return ISL_OK;
}

}  // extern "C"
