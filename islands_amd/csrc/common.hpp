// Internal declarations shared by the translation units of libislands_amd.so.
// Nothing here is part of the ABI (see include/islands_amd.h for that).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/islands_amd.h"

namespace isl {

// ---- thread-local error record (CoreError payloads, src/core/error.rs:9-62) ----
struct ErrorRecord {
  std::string message;
  uint64_t expected = 0, actual = 0, node = 0;
};
ErrorRecord& last_error();
isl_status fail(isl_status st, const char* fmt, ...);
isl_status fail_dim(uint64_t expected, uint64_t actual);
isl_status fail_node(uint64_t node);

#define ISL_HIP(expr)                                                                   \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess)                                                               \
      return ::isl::fail(ISL_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr,                \
                         hipGetErrorString(_e), __FILE__, __LINE__);                    \
  } while (0)

#define ISL_TRY(expr)            \
  do {                           \
    isl_status _s = (expr);      \
    if (_s != ISL_OK) return _s; \
  } while (0)

// Selects `device` after checking that it exists and is a gfx950 part.
isl_status use_device(int32_t device);
// compute units of a device use_device has verified (256 before that)
int device_cu_count(int32_t device);

// ---- per-index device workspace for the search kernels ----
// One lane = everything one search in flight needs: a stream, events, per-query device arrays,
// pinned host mirrors and (host-pointer entry points) staging buffers.  A lane is claimed under
// isl_index::mu and then touched by its owner alone until it is released, so the entry points do
// not hold the index mutex while they enqueue, wait or copy.
struct SearchWorkspace {
  uint32_t slots = 0;          // resident waves the scratch is sized for
  uint32_t ovf_bits = 0;       // log2 entries of the per-slot overflow visited table
  uint32_t* ovf_tab = nullptr; // [slots][1 << ovf_bits], EMPTY-filled between queries
  uint32_t cap_q = 0;          // per-query arrays sized for this many queries
  uint32_t* status = nullptr;  // [cap_q]
  uint64_t* payload = nullptr; // [cap_q]
  uint32_t* ctr = nullptr;     // [cap_q][4]  H,E,V,pushes
  uint32_t* ticket = nullptr;  // work-queue heads (fast, exact)
  uint32_t* redo = nullptr;    // [cap_q] query ids routed to the exact kernel
  uint32_t* replay = nullptr;  // [cap_q] query ids routed to the replay kernel
  uint32_t* qsel = nullptr;    // [cap_q] bf16 rows: queries whose elements are not all bf16 values
  uint32_t* qsel_h = nullptr;  // [cap_q] ... and the queries whose elements are
  uint64_t* plog = nullptr;    // [cap_q][plog_cap] push log (distance bits, id)
  uint64_t plog_entries = 0;
  // staging for the host-pointer entry points: device side ...
  float* q_stage = nullptr;
  uint64_t q_stage_bytes = 0;
  uint64_t* ids_stage = nullptr;
  float* dist_stage = nullptr;
  uint32_t* count_stage = nullptr;
  uint64_t out_stage_slots = 0;
  // ... and pinned host side (the caller's buffers are pageable: copied through these)
  float* h_q = nullptr;
  uint64_t h_q_bytes = 0;
  uint64_t* h_ids = nullptr;
  float* h_dist = nullptr;
  uint32_t* h_count = nullptr;
  uint64_t h_out_slots = 0;
  hipStream_t stream = nullptr;  // from the device's stream pool (search.hip): shared, never destroyed here
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_in = nullptr, ev_done = nullptr;
  // call in flight on this lane (claim .. release)
  bool busy = false;           // under isl_index::mu
  bool waiting = false;        // a thread is inside isl_search_wait for this lane (under mu)
  bool enqueued = false;       // kernels of a call are on the stream (owner only)
  bool ticket_clean = false;   // the work-queue heads are (or will be, in stream order) zero
  bool publish_results = false; // host-buffer call: the publish kernel also brings the answers home
  uint64_t token = 0;
  uint64_t nq_inflight = 0, k_inflight = 0;
  uint32_t ef_inflight = 0;  // the effective ef (max(ef, k)) of the call in flight
  bool fast_inflight = false;
  hipStream_t st_inflight = nullptr;
  // host-pointer call in flight: where the answers go at wait time
  uint64_t* u_ids = nullptr;
  float* u_dist = nullptr;
  uint32_t* u_count = nullptr;
  uint32_t* h_status = nullptr;  // pinned host mirrors of status / ctr / ticket
  uint32_t* h_ctr = nullptr;
  uint32_t* h_head = nullptr;
  uint64_t h_cap = 0;
  uint64_t* d_prof = nullptr;    // ISL_DEBUG phase timers of the call in flight
  uint64_t* d_tline = nullptr;   // ISL_TIMELINE: [nq][2] start / end ticks of every query of the call in flight
  uint32_t* q_entry = nullptr;   // HnswGraph: [2][nq] layer-0 entry and descent evaluations per query
  uint64_t q_entry_cap = 0;
  // recompute provider: the union of the calls this lane answers as one (search.hip, recompute_coalesced)
  float* co_q = nullptr;
  uint64_t co_q_cap = 0;
  uint64_t* co_ids = nullptr;
  uint64_t co_ids_cap = 0;
  float* co_dist = nullptr;
  uint64_t co_dist_cap = 0;
  uint32_t* co_cnt = nullptr;
  uint64_t co_cnt_cap = 0;
  // recompute provider: node ids whose rows a search round found absent, and their unique set
  uint32_t* miss = nullptr;
  uint32_t* uniq = nullptr;
  uint32_t* uniq_count = nullptr;
  uint64_t miss_cap = 0;
  uint64_t pref_cap = 0;         // entries behind miss[miss_cap]: ids a parked two-level query expects to promote next
  // ... and, for the searches that park and resume (fast kernel over the recompute provider): the
  // parked state of every query, its flag, the list of queries a round runs
  uint32_t* qstate = nullptr;
  uint64_t qstate_words = 0;     // allocated, in words
  uint32_t* qflag = nullptr;     // [qlist_cap]
  uint32_t* qlist = nullptr;     // [qlist_cap] device
  uint32_t* h_qlist = nullptr;   // [qlist_cap] pinned
  uint32_t* uslots = nullptr;    // [miss_cap] slab slots of the round's unique misses
  uint64_t qlist_cap = 0;
  uint32_t* xslot = nullptr;     // [qlist_cap] 1 + pool slot of a query parked in the heap-exact kernel
  uint32_t* h_xlist = nullptr;   // [64] pinned: the parked queries the next round hands to that kernel directly
  uint32_t round_x = 0;          // ... how many
  bool round_xpark = false;      // this call's queries park in the heap-exact kernel (bounded row cache)
  // the round search_sync is about to enqueue (recompute provider): 0 = an ordinary launch over
  // all queries; otherwise the RESUME kernel over `round_active` queries, listed in qlist unless
  // it is the first round
  uint32_t round_active = 0;
  uint32_t round_prefetch = 0;  // two-level search: ids a parked query names beyond its misses (0 = none)
  bool round_listed = false;
  // two-level search: per-query PQ distance tables [nq][m * K]
  float* tl_tables = nullptr;
  uint64_t tl_tables_cap = 0;
  bool tl_tables_built = false;  // ... of the call in flight (its later rounds / retries reuse them)
  uint32_t retry_count = 0;      // two-level search: queries re-run alone with a larger queue window (listed in qlist)
  // Asynchronous calls that cannot be split into "enqueue now, finish at wait" -- the rounds of the
  // recompute provider, the two-level search with its per-query retries -- run their synchronous form
  // on a host thread of their own; whoever waits for the token joins it and takes its status and
  // error record over.
  std::thread* worker = nullptr;
  bool threaded = false;         // the call in flight runs (or ran) on `worker`
  isl_status worker_status = ISL_OK;
  ErrorRecord worker_error;
  // device / pinned-host allocations, stream and event creations made for this lane so far: a
  // call's share of it is reported in isl_search_stats::allocations (0 after isl_index_prepare)
  uint64_t alloc_events = 0;
  uint64_t alloc_mark = 0;       // alloc_events when the call in flight was claimed
  isl_search_stats stats{};      // statistics of the call that finished last on this lane
};

constexpr int kSearchLanes = 32;  // independent workspaces = searches that may be in flight (on <= 16 pooled streams)

// Scratch of the heap-exact kernel, ONE pool per index shared by every lane: a workgroup that
// finds work in its redo queue claims a free slot (lock word per slot), so concurrent searches
// do not need a private copy each.
struct ExactPool {
  uint32_t slots = 0;
  uint64_t cand_cap = 0;        // entries per slot in the candidate heap
  float* cand_d = nullptr;      // [slots][cand_cap]
  uint32_t* cand_id = nullptr;
  uint32_t* vis_bits = nullptr; // [slots][vis_words] visited bitmap
  uint64_t vis_words = 0;
  uint32_t* ulist = nullptr;    // [slots][ulist_cap] unvisited ids of one hop
  uint32_t ulist_cap = 0;
  uint32_t* locks = nullptr;    // [slots] 0 = free, 1 = held by a workgroup (or by a query parked in the slot)
  uint32_t* xstate = nullptr;   // [slots][xstate_words] recompute provider: result heap + scalars of a parked query
  uint32_t xstate_words = 0;
};

}  // namespace isl

// ProductQuantizer (pq.rs:109-118) resident on a device: the codebooks.
struct isl_pq {
  uint64_t dimension = 0, m = 0, K = 0, dsub = 0, cstride = 0;
  int32_t metric = ISL_METRIC_EUCLIDEAN;
  int32_t device = 0;
  float* d_codebooks = nullptr;  // [m][K][cstride], rows 16-byte aligned, slack at the end
};

// The opaque handle of the ABI.  Host side mirrors LeannIndex (leann.rs:492-500).
struct isl_index {
  isl_leann_config cfg{};
  // CsrGraph, leann.rs:193-208 (host copy; may be absent for device-born graphs)
  bool host_csr_valid = true;
  std::vector<uint64_t> node_offsets{0};
  std::vector<uint64_t> neighbors;
  std::vector<uint64_t> levels;
  std::vector<uint64_t> degree_counts;
  bool has_entry = false;
  uint64_t entry_point = 0;
  uint64_t max_level = 0;
  uint64_t num_nodes = 0;
  bool has_dimension = false;
  uint64_t dimension = 0;

  // device residency
  int32_t device = -1;
  uint64_t* d_off = nullptr;  // [num_nodes + 1]
  uint32_t* d_adj = nullptr;  // [nnz] (duplicates within a row removed, first occurrence kept)
  uint64_t nnz = 0;
  uint32_t max_degree = 0;
  // distance evaluations per query of the most recent in-memory search call and the ef it ran with, packed
  // (ef << 32 | evaluations; 0 = none yet): the size of the visited table of later calls WITH THAT ef follows it
  // (search.hip, fast_geometry)
  mutable std::atomic<uint64_t> evals_hint{0};
  // in-memory provider (leann.rs:104-159): nvec rows, `stride` floats apart
  float* d_emb = nullptr;
  uint16_t* d_emb16 = nullptr;  // bf16 rows (ISL_DTYPE_BF16) instead of d_emb
  float* d_norm2 = nullptr;  // [nvec] sum of squares of every row, reference summation order
  uint64_t nvec = 0, emb_d = 0, emb_stride = 0;

  // graph under construction (build.hip): fixed-width adjacency rows, searched in place
  uint32_t* d_ell = nullptr;      // [num_nodes][ell_w]
  uint32_t* d_ell_deg = nullptr;  // [num_nodes]
  uint32_t ell_w = 0;
  bool ell_owned = false;         // the padded copy made at the first search (freed with the index)

  // recompute provider (EmbeddingProvider backed by the encoder, leann.rs:82-99): embeddings are
  // not stored (leann.rs:366-371); the search reports the rows it misses and the provider encodes
  // them from the resident token table into a bounded row cache
  struct isl_encoder* enc = nullptr;   // borrowed
  uint16_t* d_tokens = nullptr;        // [nvec][tok_L]
  uint16_t* d_lens = nullptr;          // [nvec] or NULL
  uint32_t tok_L = 0;
  // the rows live in a bounded slab (d_emb / d_norm2 indexed by SLOT): slot_of[id] = the node's
  // slot or 0xFFFFFFFF, owner[slot] = the node in it; slots are handed out round-robin, so the
  // oldest rows make room once the slab is full
  uint32_t* d_slot_of = nullptr;       // [nvec]
  uint32_t* d_owner = nullptr;         // [slab_rows]
  uint32_t* d_stamp = nullptr;         // [slab_rows] round in which a row was last asked for
  uint32_t* d_slab_head = nullptr;     // [1] where the clock hand of the slot allocator stands
  uint64_t slab_rows = 0;
  mutable uint32_t round_no = 1;       // rounds of recompute searches so far (under recompute_mu)
  bool recompute = false, keep_rows = false;
  int32_t enc_normalize = 1;

  // HnswGraph facade (hnsw.rs): distance-only heap order + upper layers for the greedy descent
  bool is_hnsw = false;
  uint64_t hnsw_layers = 0;
  const uint64_t** d_layer_off = nullptr;  // device array of device pointers, [max_level + 1]
  const uint32_t** d_layer_adj = nullptr;
  std::vector<void*> hnsw_owned;           // device allocations of the upper layers

  // two-level search (extension): PQ codes of every node, [ncodes][pq->m] u16 as ProductQuantizer::encode
  // writes them (pq.rs:221-244); the quantizer is borrowed
  const isl_pq* pq = nullptr;
  uint16_t* d_codes = nullptr;
  uint64_t ncodes = 0;

  mutable std::mutex mu;  // lane claims, the exact pool, index mutation -- never held across a search
  mutable std::mutex recompute_mu;  // searches over the recompute provider share its row table
  // asynchronous calls over the recompute provider that wait for their turn: the one that gets it answers every
  // compatible call waiting at that moment together with its own (search.hip, recompute_coalesced)
  struct RecJoin {
    std::mutex mu;        // guards `waiting`
    std::mutex leader;    // held by the call that is running the rounds
    std::vector<void*> waiting;
  };
  mutable RecJoin rec_join;
  mutable isl::SearchWorkspace ws[isl::kSearchLanes];
  mutable isl::ExactPool pool;
  mutable uint64_t next_token = 1;
};

namespace isl {
// One synchronous search over device buffers on a free lane (no argument checks): the path
// isl_search_batch_device takes, used by the graph builder for its construction searches.
isl_status search_device_sync(const isl_index* idx, const float* d_queries, uint64_t nq, uint64_t d,
                              uint64_t k, uint64_t ef, uint64_t* d_ids, float* d_dist,
                              uint32_t* d_count, hipStream_t stream);
// shard.hip: marks the queries of call `token` that failed with ISL_SHARD_POISON_COUNT in d_counts [nq]
isl_status poison_failed_queries(const isl_index* idx, uint64_t token, uint32_t* d_counts, uint64_t nq, hipStream_t stream);
isl_status materialise_host_csr(const isl_index* idx);
// build_distance_tables (pq.rs:307-338) for nq device-resident queries into d_tables [nq][m][K]
isl_status pq_launch_tables(const isl_pq* pq, const float* d_queries, uint64_t nq, float* d_tables,
                            hipStream_t st);
void free_workspace(SearchWorkspace& ws);
void free_exact_pool(ExactPool& pool);
// true while a search is in flight on any lane (call under idx->mu): provider / PQ setters and
// isl_index_free must not free tables such a search reads
bool any_lane_busy(const isl_index* idx);
// joins the host threads of asynchronous calls still running on the index's lanes (isl_index_free)
void join_lane_workers(const isl_index* idx);
// Builds the padded adjacency (64 ids per node + degrees) the traversal reads; under idx->mu.
isl_status ensure_padded_adjacency(isl_index* idx);
}  // namespace isl
