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
#include <chrono>

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
  bool zombie = false;        // the submit failed on this rank: nobody will ask for the result, the slot
                              // goes back once its exchange has completed (reclaimed by a later submit)
  bool lost = false;          // its exchange timed out: the collective may still touch the buffers, never reused
  uint64_t handle = 0, token = 0;
  uint64_t nq = 0, k = 0;
  uint64_t released_at = 0;   // free slots are reused least-recently-released first
  uint8_t* rec = nullptr;    // this rank's record, written in place by the search kernels
  uint8_t* gath = nullptr;   // [world][B]
  uint64_t* ids = nullptr;   // merged answers
  float* dist = nullptr;
  uint32_t* src = nullptr;
  uint32_t* cnt = nullptr;
  uint8_t* h_rec = nullptr;  // host transport: pinned mirrors
  uint8_t* h_gath = nullptr;
  uint32_t* d_flags = nullptr;  // this batch's merge flags (one word of the searcher's array)
  uint32_t* h_flags = nullptr;  // ... and their pinned mirror, written on the side stream behind the merge
  hipEvent_t done = nullptr;  // behind the merge of the batch in this slot
  bool recorded = false;      // ... once the submit got that far
  isl_status local_st = ISL_OK;  // why this rank poisoned its record (ISL_OK: it did not)
  isl::ErrorRecord local_err;
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
  uint32_t* d_flags = nullptr;   // [depth] one word per slot, cleared on the side stream before the slot's merge
  uint32_t* h_flags = nullptr;   // [depth] pinned mirrors
  uint32_t flags_seen = 0;       // OR of the flags of every completed batch (isl_sharded_flags)
  uint8_t* d_warm = nullptr;     // [16 + 16 * world] the 16-byte status exchange of isl_sharded_prepare
  uint8_t* h_warm = nullptr;     // pinned, same size
  hipStream_t side = nullptr;
  int32_t depth = 1;
  std::vector<Slot> slots;
  uint64_t cap_nq = 0, cap_k = 0;
  uint64_t release_clock = 1, next_handle = 1;
  // staging of the host-buffer entry point (one such call at a time: host_mu spans the whole call)
  float* d_q = nullptr;
  uint64_t d_q_bytes = 0;
  std::mutex mu;
  std::mutex host_mu;
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

// a slot is free for a submit when nothing of a batch can still touch it
bool slot_in_use(const Slot& sl) { return sl.busy || sl.zombie || sl.lost; }

// Slots whose submit failed locally (nobody asks for their result) go back once their exchange has
// completed; under s->mu.
void reclaim_zombies(isl_sharded_searcher* s) {
  for (Slot& sl : s->slots) {
    if (!sl.zombie) continue;
    if (sl.recorded && hipEventQuery(sl.done) != hipSuccess) { (void)hipGetLastError(); continue; }
    if (sl.token) { (void)isl_search_wait(s->idx, sl.token); sl.token = 0; }
    sl.zombie = false;
    sl.released_at = s->release_clock++;
  }
}

isl_status size_slots(isl_sharded_searcher* s, uint64_t nq, uint64_t k) {
  if (nq <= s->cap_nq && k <= s->cap_k && !s->slots.empty()) return ISL_OK;
  reclaim_zombies(s);
  for (const Slot& sl : s->slots)
    if (slot_in_use(sl))
      return isl::fail(ISL_ERR_SEARCH, "Search error: a larger batch than isl_sharded_prepare sized the buffers for, "
                       "while batches are in flight");
  const uint64_t cnq = std::max(nq, s->cap_nq), ck = std::max<uint64_t>(std::max(k, s->cap_k), 1);
  for (Slot& sl : s->slots) free_slot(sl);
  s->slots.assign((size_t)s->depth, Slot{});
  s->cap_nq = s->cap_k = 0;
  const uint64_t B = isl_shard_record_bytes(cnq, ck);
  size_t i = 0;
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
    sl.d_flags = s->d_flags + i;
    sl.h_flags = s->h_flags + i;
    sl.released_at = s->release_clock++;
    ++i;
  }
  s->cap_nq = cnq;
  s->cap_k = ck;
  return ISL_OK;
}

// how long a wait on an exchange may take before the batch is declared lost (a peer that never arrives)
uint64_t exchange_timeout_ms() {
  static const uint64_t ms = [] {
    const char* e = getenv("ISL_SHARD_TIMEOUT_MS");
    const long long v = e ? atoll(e) : 60000;
    return (uint64_t)(v > 0 ? v : 60000);
  }();
  return ms;
}

// hipEventSynchronize with a deadline: 0 = completed, 1 = timed out, -1 = a HIP error (in *err)
int wait_event_bounded(hipEvent_t ev, hipError_t* err) {
  const auto t0 = std::chrono::steady_clock::now();
  uint32_t spins = 0;
  for (;;) {
    const hipError_t e = hipEventQuery(ev);
    if (e == hipSuccess) return 0;
    if (e != hipErrorNotReady) { *err = e; (void)hipGetLastError(); return -1; }
    (void)hipGetLastError();
    if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
    if ((spins & 255u) == 0 &&
        (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() >
            exchange_timeout_ms())
      return 1;
  }
}
int wait_stream_bounded(hipStream_t st, hipError_t* err) {
  const auto t0 = std::chrono::steady_clock::now();
  uint32_t spins = 0;
  for (;;) {
    const hipError_t e = hipStreamQuery(st);
    if (e == hipSuccess) return 0;
    if (e != hipErrorNotReady) { *err = e; (void)hipGetLastError(); return -1; }
    (void)hipGetLastError();
    if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
    if ((spins & 255u) == 0 &&
        (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() >
            exchange_timeout_ms())
      return 1;
  }
}

// the batch's merge flags -> their pinned mirror (a kernel, like the search path's publish: no runtime copy)
__global__ void publish_flags_kernel(const uint32_t* __restrict__ d_flags, uint32_t* __restrict__ h_flags) {
  *h_flags = *d_flags;
}
// every count of a record = ISL_SHARD_POISON_COUNT: this rank has no answer for the batch
__global__ void poison_record_kernel(uint32_t* __restrict__ counts, uint32_t nq) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nq) counts[i] = ISL_SHARD_POISON_COUNT;
}

// One all-gather of `bytes` from every rank over the searcher's transport, device buffers, on the side
// stream (RCCL / single rank: enqueued; host transport: synchronous, through the pinned mirrors).
isl_status exchange(isl_sharded_searcher* s, const uint8_t* d_send, uint8_t* d_recv, uint8_t* h_send, uint8_t* h_recv,
                    uint64_t bytes) {
  if (s->grp && s->grp->comm) {  // (also with a single rank: the communicator the caller set up is used)
    Rccl* r = nullptr;
    ISL_TRY(rccl_ready(r));
    ISL_NCCL(r, r->AllGather(d_send, d_recv, bytes, ncclUint8, s->grp->comm, s->side));
  } else if (s->world == 1 || !s->grp) {
    ISL_HIP(hipMemcpyAsync(d_recv, d_send, bytes, hipMemcpyDeviceToDevice, s->side));
  } else {
    ISL_HIP(hipMemcpyAsync(h_send, d_send, bytes, hipMemcpyDeviceToHost, s->side));
    ISL_HIP(hipStreamSynchronize(s->side));
    const int32_t rc = s->grp->host_fn(s->grp->host_user, h_send, h_recv, bytes);
    if (rc != 0) return isl::fail(ISL_ERR_IO, "IO error: the host all-gather callback returned %d", rc);
    ISL_HIP(hipMemcpyAsync(d_recv, h_recv, bytes * (uint64_t)s->world, hipMemcpyHostToDevice, s->side));
  }
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
  const size_t warm = 16 + 16 * (size_t)s->world;
  if (hipMalloc(&s->d_base, (size_t)s->world * 8) != hipSuccess || hipMalloc(&s->d_flags, (size_t)depth * 4) != hipSuccess ||
      hipHostMalloc(&s->h_flags, (size_t)depth * 4) != hipSuccess ||
      hipMalloc(&s->d_warm, warm) != hipSuccess || hipHostMalloc(&s->h_warm, warm) != hipSuccess ||
      hipMemcpy(s->d_base, s->id_base.data(), (size_t)s->world * 8, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemset(s->d_flags, 0, (size_t)depth * 4) != hipSuccess || hipMemset(s->d_warm, 0, warm) != hipSuccess ||
      hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking) != hipSuccess)
    return bail(isl::fail(ISL_ERR_DEVICE, "device set-up of the sharded searcher failed: %s",
                          hipGetErrorString(hipGetLastError())));
  memset(s->h_flags, 0, (size_t)depth * 4);
  *out = s;
  return ISL_OK;
}

void isl_sharded_searcher_free(isl_sharded_searcher* s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  if (s->side) {  // bounded: a peer that never entered a collective must not hang the teardown either
    hipError_t e = hipSuccess;
    if (wait_stream_bounded(s->side, &e) == 1)
      fprintf(stderr, "[isl] sharded searcher freed with an exchange still pending after %llu ms\n",
              (unsigned long long)exchange_timeout_ms());
  }
  for (Slot& sl : s->slots) {
    // a batch nobody asked the result of: its search still holds a lane of the index
    if ((sl.busy || sl.zombie) && sl.token) (void)isl_search_wait(s->idx, sl.token);
    if (!sl.lost) free_slot(sl);  // (a lost slot's buffers may still be written by a late collective: leaked on purpose)
  }
  if (s->d_base) (void)hipFree(s->d_base);
  if (s->d_flags) (void)hipFree(s->d_flags);
  if (s->h_flags) (void)hipHostFree(s->h_flags);
  if (s->d_warm) (void)hipFree(s->d_warm);
  if (s->h_warm) (void)hipHostFree(s->h_warm);
  if (s->d_q) (void)hipFree(s->d_q);
  if (s->side) (void)hipStreamDestroy(s->side);
  delete s;
}

isl_status isl_sharded_prepare(isl_sharded_searcher* s, uint64_t max_nq, uint64_t max_k, uint64_t max_ef) {
  if (!s) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "searcher is NULL");
  if (max_nq == 0) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "max_nq out of range");
  ISL_TRY(isl::use_device(s->device));
  std::lock_guard<std::mutex> lock(s->mu);
  // this rank's own set-up; whatever it returns, the collective below is entered
  isl_status local = size_slots(s, max_nq, max_k);
  if (local == ISL_OK && s->idx->num_nodes)
    local = isl_index_prepare(const_cast<isl_index*>(s->idx), max_nq, max_ef, max_k, s->depth);
  const isl::ErrorRecord keep = isl::last_error();
  if (!s->grp) return local;
  // 16 bytes per rank, byte 0 = "my set-up failed".  With RCCL this is also the communicator's first
  // collective, which sets up its channels: made here, not in the first batch.
  uint8_t mine[16] = {(uint8_t)(local != ISL_OK ? 1 : 0)};
  isl_status ex = ISL_OK;
  if (hipMemcpy(s->d_warm, mine, 16, hipMemcpyHostToDevice) != hipSuccess)
    ex = isl::fail(ISL_ERR_DEVICE, "hipMemcpy failed: %s", hipGetErrorString(hipGetLastError()));
  // (entered even then: the device buffer then holds an older outcome, the peers still get their collective)
  const isl_status ex2 = exchange(s, s->d_warm, s->d_warm + 16, s->h_warm, s->h_warm + 16, 16);
  if (ex == ISL_OK) ex = ex2;
  if (ex == ISL_OK) {
    hipError_t e = hipSuccess;
    const int w = wait_stream_bounded(s->side, &e);
    if (w == 1) ex = isl::fail(ISL_ERR_DEVICE, "the set-up exchange of isl_sharded_prepare did not complete within %llu ms "
                               "(a rank that never called it?)", (unsigned long long)exchange_timeout_ms());
    else if (w < 0) ex = isl::fail(ISL_ERR_DEVICE, "shard exchange failed: %s", hipGetErrorString(e));
  }
  if (local != ISL_OK) { isl::last_error() = keep; return local; }
  ISL_TRY(ex);
  std::vector<uint8_t> all(16 * (size_t)s->world);
  ISL_HIP(hipMemcpy(all.data(), s->d_warm + 16, all.size(), hipMemcpyDeviceToHost));
  for (int r = 0; r < s->world; ++r)
    if (all[16 * (size_t)r])
      return isl::fail(ISL_ERR_SEARCH, "Search error: rank %d failed its part of isl_sharded_prepare", r);
  return ISL_OK;
}

isl_status isl_sharded_submit(isl_sharded_searcher* s, const float* d_queries, uint64_t nq, uint64_t d,
                              uint64_t k, uint64_t ef, void* stream, uint64_t* handle) {
  if (!s || !handle) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL argument");
  *handle = 0;
  // Everything that can fail before the collective point is symmetric across ranks by the contract of
  // this call (every rank submits the same (nq, k) in the same order, with the same depth): argument
  // checks, buffer sizes, a free slot.
  if (nq == 0 || k == 0) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "nq and k must be positive");
  ISL_TRY(isl::use_device(s->device));
  std::lock_guard<std::mutex> lock(s->mu);
  ISL_TRY(size_slots(s, nq, k));
  reclaim_zombies(s);
  // any free slot, the one released longest ago first: a completed batch's answers stay untouched until
  // `depth` further batches have been submitted, whatever order the results were taken in
  Slot* slp = nullptr;
  for (Slot& c : s->slots)
    if (!slot_in_use(c) && (!slp || c.released_at < slp->released_at)) slp = &c;
  if (!slp)
    return isl::fail(ISL_ERR_SEARCH, "Search error: %d sharded batches already in flight; isl_sharded_result one first",
                     s->depth);
  Slot& sl = *slp;
  const uint64_t B = isl_shard_record_bytes(nq, k);
  uint32_t* rec_counts = (uint32_t*)(sl.rec + nq * k * 12);
  // ---- from here on this rank takes part in the batch's collective, whatever happens to its own search
  sl.nq = nq;
  sl.k = k;
  sl.token = 0;
  sl.recorded = false;
  sl.local_st = ISL_OK;
  uint64_t tok = 0;
  isl_status local = isl_search_batch_device_async(s->idx, d_queries, nq, d, k, ef, (uint64_t*)sl.rec,
                                                   (float*)(sl.rec + nq * k * 8), rec_counts, stream, &tok);
  sl.token = tok;
  if (local == ISL_OK) {
    if (tok) {
      local = isl_search_stream_wait(s->idx, tok, s->side);
      // per-query failures (NodeNotFound, scratch exhausted ...): those queries' counts are poisoned, so
      // that the other ranks' merges see them too
      if (local == ISL_OK) local = isl::poison_failed_queries(s->idx, tok, rec_counts, nq, s->side);
    } else {
      // an empty shard answered at once (counts zeroed on the NULL stream, leann.rs:875-877)
      if (hipStreamSynchronize(nullptr) != hipSuccess)
        local = isl::fail(ISL_ERR_DEVICE, "hipStreamSynchronize failed: %s", hipGetErrorString(hipGetLastError()));
    }
  }
  if (local != ISL_OK) {
    sl.local_st = local;
    sl.local_err = isl::last_error();
    // a search that did get enqueued may still write the record: complete it first, then poison
    if (tok) { (void)isl_search_wait(s->idx, tok); sl.token = 0; }
    hipLaunchKernelGGL(poison_record_kernel, dim3((uint32_t)((nq + 255) / 256)), dim3(256), 0, s->side, rec_counts,
                       (uint32_t)nq);
    (void)hipGetLastError();
  }
  isl_status ex = exchange(s, sl.rec, sl.gath, sl.h_rec, sl.h_gath, B);
  if (ex == ISL_OK && hipMemsetAsync(sl.d_flags, 0, 4, s->side) != hipSuccess)
    ex = isl::fail(ISL_ERR_DEVICE, "hipMemsetAsync failed: %s", hipGetErrorString(hipGetLastError()));
  if (ex == ISL_OK)
    ex = isl_merge_topk_packed_async((uint64_t)s->world, nq, k, sl.gath, B, s->d_base, k, sl.ids, sl.dist, sl.src, sl.cnt,
                                     sl.d_flags, s->device, s->side);
  if (ex == ISL_OK) {
    hipLaunchKernelGGL(publish_flags_kernel, dim3(1), dim3(1), 0, s->side, sl.d_flags, sl.h_flags);
    if (hipGetLastError() != hipSuccess || hipEventRecord(sl.done, s->side) != hipSuccess)
      ex = isl::fail(ISL_ERR_DEVICE, "enqueue on the exchange stream failed: %s", hipGetErrorString(hipGetLastError()));
    else
      sl.recorded = true;
  }
  if (local != ISL_OK || ex != ISL_OK) {
    // nobody will ask for this batch's result on this rank: the slot goes back behind its exchange
    sl.zombie = true;
    if (local != ISL_OK) { isl::last_error() = sl.local_err; return local; }
    return ex;
  }
  sl.busy = true;
  sl.handle = s->next_handle++;
  *handle = sl.handle;
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
  // (the batch's own event: the side stream also carries the exchanges of the batches behind it);
  // bounded -- a peer that never enters the collective is an error of this batch, not a hang
  hipError_t e = hipSuccess;
  const int w = sl->recorded ? wait_event_bounded(sl->done, &e) : wait_stream_bounded(s->side, &e);
  uint32_t flags = 0;
  {
    std::lock_guard<std::mutex> lock(s->mu);
    sl->busy = false;
    sl->token = 0;
    if (w == 1) {
      sl->lost = true;  // the collective may still complete later and write the slot: never reused
    } else {
      sl->released_at = s->release_clock++;
      if (w == 0) { flags = *sl->h_flags; s->flags_seen |= flags; }
    }
  }
  if (w == 1)
    return isl::fail(ISL_ERR_DEVICE, "the exchange of sharded batch %llu did not complete within %llu ms (a rank that "
                     "never submitted it?)", (unsigned long long)handle, (unsigned long long)exchange_timeout_ms());
  if (st != ISL_OK) { isl::last_error() = keep; return st; }
  if (w < 0) return isl::fail(ISL_ERR_DEVICE, "shard exchange failed: %s", hipGetErrorString(e));
  if (flags & 4u)
    return isl::fail(ISL_ERR_SEARCH, "Search error: the shard search of another rank failed for sharded batch %llu "
                     "(MultiIndexSearcher::search propagates an index's error, search.rs:215)", (unsigned long long)handle);
  if (flags & 1u) return isl::fail(ISL_ERR_SEARCH, "Search error: NaN score in merge (the reference panics here)");
  if (d_ids) *d_ids = sl->ids;
  if (d_dist) *d_dist = sl->dist;
  if (d_src) *d_src = sl->src;
  if (d_count) *d_count = sl->cnt;
  return ISL_OK;
}

isl_status isl_sharded_flags(isl_sharded_searcher* s, uint32_t* flags) {
  if (!s || !flags) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL argument");
  ISL_TRY(isl::use_device(s->device));
  hipError_t e = hipSuccess;
  const int w = wait_stream_bounded(s->side, &e);
  if (w == 1) return isl::fail(ISL_ERR_DEVICE, "the exchange stream did not drain within %llu ms",
                               (unsigned long long)exchange_timeout_ms());
  if (w < 0) return isl::fail(ISL_ERR_DEVICE, "shard exchange failed: %s", hipGetErrorString(e));
  std::lock_guard<std::mutex> lock(s->mu);
  *flags = s->flags_seen;
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
  // one host-buffer call at a time per searcher: the staging buffer is reallocated, filled, searched
  // from and its answers copied out under this lock (two threads would otherwise overwrite each
  // other's queries or free the buffer under a search in flight)
  std::lock_guard<std::mutex> host_lock(s->host_mu);
  const uint64_t bytes = nq * d * 4;
  if (s->d_q_bytes < bytes) {
    if (s->d_q) (void)hipFree(s->d_q);
    s->d_q = nullptr;
    s->d_q_bytes = 0;
    ISL_HIP(hipMalloc(&s->d_q, bytes));
    s->d_q_bytes = bytes;
  }
  ISL_HIP(hipMemcpy(s->d_q, queries, bytes, hipMemcpyHostToDevice));
  uint64_t h = 0;
  ISL_TRY(isl_sharded_submit(s, s->d_q, nq, d, k, ef, nullptr, &h));
  const uint64_t *ids = nullptr; const float* dist = nullptr; const uint32_t *src = nullptr, *cnt = nullptr;
  ISL_TRY(isl_sharded_result(s, h, &ids, &dist, &src, &cnt, nullptr));  // (this batch's own flags are checked there)
  ISL_HIP(hipMemcpy(out_ids, ids, nq * k * 8, hipMemcpyDeviceToHost));
  ISL_HIP(hipMemcpy(out_dist, dist, nq * k * 4, hipMemcpyDeviceToHost));
  if (out_src) ISL_HIP(hipMemcpy(out_src, src, nq * k * 4, hipMemcpyDeviceToHost));
  ISL_HIP(hipMemcpy(out_count, cnt, nq * 4, hipMemcpyDeviceToHost));
  return ISL_OK;
}

}  // extern "C"
