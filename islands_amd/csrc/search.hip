// LEANN best-first search on gfx950: one 64-lane wavefront per query.
//
// Replaces LeannIndex::search_with_params / search_layer_recompute
// (src/core/leann.rs:868-988) over an EmbeddingProvider (in-memory rows, leann.rs:104-159, or the
// recompute provider) with DistanceMetric::calculate (src/core/distance.rs:37-122); the same
// kernels serve HnswGraph::search (src/core/hnsw.rs:458-504) and the graph builder's
// construction searches.
//
// Parity design:
//   * distances are computed in the reference's exact operation order: the strictly sequential
//     f32 chain of a row (separate multiply and add roundings, -ffp-contract=off), so every
//     distance is bit-identical to the Rust scalar loop and every traversal decision (strict
//     float compares at leann.rs:925,959) matches.  The four lanes of a quad own one row, load 16
//     of every 64 bytes of it straight from global memory into a register ring and exchange the
//     products with DPP (device_common.hip.h, direct_distances).
//   * fast kernel: the result set R (<= ef entries, key = (OrderedFloat d, id)) lives in registers
//     as a sorted array spread over the wave; the candidate heap is implicit (live candidates are
//     exactly the unexpanded entries of R); the pushes of a hop are merged into R at once.
//     Situations where the reference's BinaryHeap internals become observable (equal distances
//     inside the returned prefix, or between an evicted entry and the new worst) are replayed from
//     a push log or handed to the exact kernel.
//   * exact kernel: emulates Rust's BinaryHeap push/pop/into_iter byte for byte (candidates in
//     HBM scratch, results in LDS) for those queries, for ef > 512, for adjacency rows longer
//     than 64 and for NaN / -0.0 distances.
#include "search_kernels.hip.h"

namespace {

struct FastGeom {
  uint32_t hbits;  // the table ef (and a long query) alone give: 1 << hbits entries; what parked queries' state blocks hold
  size_t lds;
  uint32_t hcap;   // entries of the table this launch runs with (1 << hbits unless the index's hint enlarged it)
};

// qbytes = bytes per query element in LDS: 4, or 2 for the instantiation that keeps a bf16-valued
// query as bf16 (bf16 rows; at d = 4096 the query is what bounds the waves per CU)
// vhint = distance evaluations per query this index's searches have been making (0 = unknown): every
// evaluated node is an entry of the visited table.
FastGeom fast_geometry(uint32_t ef, uint32_t d, uint32_t qbytes = 4, uint32_t vhint = 0) {
  // visited capacity grows with ef (V is roughly 10-30 x ef); overflow goes to HBM
  uint32_t hbits = ef <= 64 ? 10 : ef <= 160 ? 11 : ef <= 320 ? 12 : 13;
  static const int hbits_env = [] { const char* e = getenv("ISL_HBITS"); return e ? atoi(e) : 0; }();
  // A long query takes most of a wave's LDS (d = 4096: 16 KB as float32, 8 KB as bf16) and the waves
  // per CU with it; once it is at least as large as the visited table, half a table buys more
  // through occupancy than it costs through the overflow table in HBM (10M x 4096 bf16 rows,
  // ef = 128: 0.40 -> 0.45 of the HBM peak).  At d = 768 the full table wins and stays.
  const size_t qlds = (size_t)d * qbytes;
  if (qlds >= ((size_t)4 << hbits) && hbits > 9) hbits -= 1;
  // visited table, merge buffer, query (+ 64 bytes when d is not a multiple of 16: the operand
  // prefetch of direct_group may touch the rest of the last step)
  const size_t rest = (size_t)mbuf_entries(ef) * 8 +
                      (qbytes == 2 ? (size_t)((d + 7) / 8 * 8) * 2 + 64 : (size_t)((d + 3) / 4 * 4) * 4 + ((d & 15) ? 64 : 0));
  if (hbits_env >= 8 && hbits_env <= 14) hbits = (uint32_t)hbits_env;  // experiments only
  uint32_t hcap = 1u << hbits;
  // Round 4: a larger table when the index's queries have been filling it past its 7/8 limit on average.
  // How many nodes a query evaluates is a property of the graph and the data, not of ef alone (ef = 128: 1226
  // on the tree-of-clusters rows with the harness graph, 3100-3400 on manifold rows with an exact-kNN graph),
  // and a query past the limit pays an atomic round trip to its HBM overflow table for every further hop:
  // measured on the latter rows (1M x 768, 20 steps, profiles/r04_bench_M_knn_1m_hbits{11,12,13}.json)
  // 2048 entries 558 k queries/s, 4096 entries 686 k (12 -> 7 waves per CU and still +23 %), 8192 entries 602 k.
  // The table need not be a power of two (hslot_cap): it takes what the average query needs, in steps of 512
  // entries and at most four times the default, and then whatever else fits beside the same number of waves per CU.
  static const bool no_hint = getenv("ISL_NO_VISITED_HINT") != nullptr;  // A/B switch for measurements
  static const int hcap_env = [] { const char* e = getenv("ISL_HCAP"); return e ? atoi(e) : 0; }();  // experiments only
  if (vhint && !no_hint && hbits_env == 0) {
    const uint64_t need = ((uint64_t)vhint * 8 + 6) / 7;
    if (need > hcap) {
      const uint64_t most = (uint64_t)4 << hbits;  // (2 x until the densest graph of DESIGN section 4: 4096 entries 357 k, 5696 420 k queries/s)
      uint64_t want = std::min<uint64_t>((need + 511) / 512 * 512, most);
      auto lds_of = [&](uint64_t cap) { return (cap * 4 + rest + 511) / 512 * 512; };  // (LDS is handed out in 512-byte granules)
      auto room = [&](size_t waves) -> uint64_t {  // the largest table that leaves `waves` waves per CU
        const size_t each = (160 * 1024) / waves / 512 * 512;
        return each > rest ? std::min<uint64_t>((each - rest) / 4 / 64 * 64, most) : 0;
      };
      const size_t per_cu = std::max<size_t>(1, (160 * 1024) / lds_of(want));
      want = std::max(want, room(per_cu));           // what fits beside the same waves is free
      if (room(per_cu + 1) >= need) want = room(per_cu + 1);  // one more wave per CU if the average query still fits
      hcap = (uint32_t)want;
    }
  }
  if (hcap_env >= 256 && hcap_env <= 16384) hcap = (uint32_t)hcap_env / 64 * 64;
  const size_t lds = (size_t)hcap * 4 + rest;
  return {hbits, lds, hcap};
}


size_t exact_lds(uint32_t ef, uint32_t d) {
  return (size_t)TILE_ROWS * TILE_LD * 4 + 64 * 4 + 64 * 4 + 8 * 4 + (size_t)(ef + 1) * 8 + 16 +
         (size_t)((d + 3) / 4 * 4) * 4;
}

// Two-level search: LDS of one wave = visited table + approximate-queue window + R + staging + query
struct TwoLevelCall {
  float ratio;
  uint32_t window_scale = 1;  // the window grows 4x per retry after a query outgrew it
};
// qbytes = 2: the instantiation that keeps a bf16-valued query as bf16 (bf16 rows)
size_t two_level_lds(uint32_t hbits, uint32_t wcap, uint32_t ef, uint32_t d, uint32_t qbytes = 4) {
  return ((size_t)4 << hbits) + (size_t)(wcap + 64) * 8 + (size_t)tl_res_entries(ef) * 8 + 64 * 8 +
         128 * 4 + (size_t)kTlLdsWords * 4 +
         (qbytes == 2 ? (size_t)((d + 7) / 8 * 8) * 2 + 64 : (size_t)((d + 3) / 4 * 4) * 4 + ((d & 15) ? 64 : 0));
}

__global__ void fill_u32_kernel(uint32_t* p, uint64_t n, uint32_t v) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

// The search path moves its small host<->device traffic with kernels that read / write the
// lane's pinned host buffers directly (they are mapped into the device's address space) instead
// of hipMemcpyAsync: a call is then kernels and two event records only.  (The runtime's async
// copies draw completion signals from pools that grow one concurrent copy at a time, several
// milliseconds each -- measured as 7 ms stalls inside the first dozens of pipelined calls.)
struct PublishParams {
  const uint32_t* src[6];
  uint32_t* dst[6];
  uint32_t words[6];
  uint32_t ticket_seg;  // that segment's source is zeroed once copied: the next call's work-queue heads
};
__global__ void publish_kernel(PublishParams pp) {
  const uint32_t seg = blockIdx.y;
  const uint32_t n = pp.words[seg];
  const uint32_t* __restrict__ src = pp.src[seg];
  uint32_t* __restrict__ dst = pp.dst[seg];
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    dst[i] = src[i];
    if (seg == pp.ticket_seg) const_cast<uint32_t*>(src)[i] = 0u;
  }
}
__global__ void copy_u128_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
}
__global__ void copy_u32_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
}

// recompute provider: the queries parked in the heap-exact kernel go straight back to its queue
__global__ void seed_redo_kernel(const uint32_t* __restrict__ list, uint32_t n, uint32_t* __restrict__ redo,
                                 uint32_t* __restrict__ ticket) {
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) redo[i] = list[i];
  if (threadIdx.x == 0) ticket[1] = n;
}

constexpr uint32_t kExactSlots = 32;
constexpr uint32_t kTlPrefetchDefault = 0;  // (set from the measurement: DESIGN.md section 3.4)
constexpr uint32_t kOvfBits = 15;
constexpr uint32_t kMaxExactEf = 4096;

// Every allocation of the search path goes through these: a lane counts what it had to set up,
// and a call reports its share in isl_search_stats::allocations (0 once isl_index_prepare has run).
template <typename T>
isl_status lane_malloc(isl::SearchWorkspace& ws, T*& ptr, size_t bytes) {
  ptr = nullptr;
  ISL_HIP(hipMalloc(&ptr, bytes ? bytes : 4));
  ws.alloc_events++;
  return ISL_OK;
}
template <typename T>
isl_status lane_host_malloc(isl::SearchWorkspace& ws, T*& ptr, size_t bytes) {
  ptr = nullptr;
  ISL_HIP(hipHostMalloc(&ptr, bytes ? bytes : 4));
  ws.alloc_events++;
  return ISL_OK;
}
template <typename T>
isl_status ensure(isl::SearchWorkspace& ws, T*& ptr, uint64_t& have, uint64_t want) {
  if (have >= want && ptr) return ISL_OK;
  if (ptr) (void)hipFree(ptr);
  ptr = nullptr;
  have = 0;
  ISL_TRY(lane_malloc(ws, ptr, want * sizeof(T)));
  have = want;
  return ISL_OK;
}

// resident waves per CU the launch geometry may count on (ISL_WAVES_PER_CU: experiments only)
size_t waves_per_cu_cap() {
  static const size_t cap = [] {
    const char* wc = getenv("ISL_WAVES_PER_CU");
    return wc ? (size_t)std::max(1, atoi(wc)) : (size_t)16;
  }();
  return cap;
}
uint32_t push_log_cap(uint32_t ef) { return std::max<uint32_t>(1024, 12 * ef); }  // pushes per query ~ 3-6 x ef

// Streams, events, per-query arrays, overflow table, push log of one lane, sized for nq queries
// on `slots` resident waves.
// The lanes' streams come from ONE pool per device and process: a card serves only so many hardware
// queues side by side, and past about twenty streams in use the search rate collapses (24 streams on
// one card: 0.39 M queries/s where 16 reach 1.4 M; two processes with 12 each: 16 k against 145 k --
// DESIGN section 4).  More lanes than pool streams simply share: lane i of any index runs on stream
// i mod pool size, its calls in stream order behind the other lane's, every call waited for through
// its own event.  ISL_MAX_STREAMS (1..32, default 16) sizes the pool -- processes that share a card
// should divide the sixteen between them.  Pool streams live as long as the process.
hipStream_t pool_stream(int32_t device, uint32_t lane, bool* created) {
  static std::mutex mu;
  static hipStream_t pool[64][32];
  static const uint32_t size = [] {
    const char* e = getenv("ISL_MAX_STREAMS");
    const int v = e ? atoi(e) : 16;
    return (uint32_t)std::min(std::max(v, 1), 32);
  }();
  *created = false;
  if (device < 0 || device >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  hipStream_t& st = pool[device][lane % size];
  if (!st) {
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); st = nullptr; }
    else *created = true;
  }
  return st;
}

isl_status ensure_lane_stream(const isl_index* idx, isl::SearchWorkspace& ws) {
  if (ws.stream) return ISL_OK;
  bool created = false;
  hipStream_t st = pool_stream(idx->device, (uint32_t)(&ws - idx->ws), &created);
  if (!st) return isl::fail(ISL_ERR_DEVICE, "hipStreamCreate failed for a search lane");
  if (created) ws.alloc_events++;
  static const bool no_evdone = getenv("ISL_NO_EVDONE") != nullptr;  // A/B switch for measurements (stream synchronisation instead)
  if (!ws.ev_done && !no_evdone) { ISL_HIP(hipEventCreateWithFlags(&ws.ev_done, hipEventDisableTiming)); ws.alloc_events++; }
  if (!ws.ev0) { ISL_HIP(hipEventCreate(&ws.ev0)); ws.alloc_events++; }
  if (!ws.ev1) { ISL_HIP(hipEventCreate(&ws.ev1)); ws.alloc_events++; }
  if (!ws.ev_in) { ISL_HIP(hipEventCreateWithFlags(&ws.ev_in, hipEventDisableTiming)); ws.alloc_events++; }
  if (!ws.ticket) {
    ISL_TRY(lane_malloc(ws, ws.ticket, 64));
    ws.ticket_clean = false;
  }
  if (!ws.h_head) ISL_TRY(lane_host_malloc(ws, ws.h_head, 64));
  ws.stream = st;
  return ISL_OK;
}

isl_status prepare_workspace(const isl_index* idx, isl::SearchWorkspace& ws, uint32_t nq, uint32_t slots,
                             uint32_t plog_cap) {
  ISL_TRY(ensure_lane_stream(idx, ws));
  if (ws.h_cap < nq) {
    if (ws.h_status) (void)hipHostFree(ws.h_status);
    if (ws.h_ctr) (void)hipHostFree(ws.h_ctr);
    ws.h_status = ws.h_ctr = nullptr;
    ws.h_cap = 0;
    uint64_t cap = nq < 1024 ? 1024 : nq;
    ISL_TRY(lane_host_malloc(ws, ws.h_status, cap * 4));
    ISL_TRY(lane_host_malloc(ws, ws.h_ctr, cap * 16));
    ws.h_cap = cap;
  }
  if (ws.slots < slots || !ws.ovf_tab) {
    if (ws.ovf_tab) (void)hipFree(ws.ovf_tab);
    ws.ovf_tab = nullptr;
    ws.slots = 0;
    // (ISL_OVF_BITS: the overflow table's size for measurements; a query that fills 3/4 of it goes to the heap-exact kernel)
    static const uint32_t ovf_bits_cfg = [] {
      const char* e = getenv("ISL_OVF_BITS");
      const int v = e ? atoi(e) : (int)kOvfBits;
      return (uint32_t)std::min(15, std::max(10, v));
    }();
    ws.ovf_bits = ovf_bits_cfg;
    uint64_t n = (uint64_t)slots << ovf_bits_cfg;
    ISL_TRY(lane_malloc(ws, ws.ovf_tab, n * 4));
    hipLaunchKernelGGL(fill_u32_kernel, dim3(2048), dim3(256), 0, ws.stream, ws.ovf_tab, n, EMPTY);
    ISL_HIP(hipGetLastError());
    ISL_HIP(hipStreamSynchronize(ws.stream));
    ws.slots = slots;
  }
  if (ws.cap_q < nq) {
    void* ptrs[] = {ws.status, ws.payload, ws.ctr, ws.redo, ws.replay, ws.qsel, ws.qsel_h};
    for (void* q : ptrs)
      if (q) (void)hipFree(q);
    ws.status = nullptr; ws.payload = nullptr; ws.ctr = nullptr; ws.redo = nullptr;
    ws.replay = nullptr; ws.qsel = nullptr; ws.qsel_h = nullptr;
    ws.cap_q = 0;
    uint32_t cap = nq < 1024 ? 1024 : nq;
    ISL_TRY(lane_malloc(ws, ws.status, (size_t)cap * 4));
    ISL_TRY(lane_malloc(ws, ws.payload, (size_t)cap * 8));
    ISL_TRY(lane_malloc(ws, ws.ctr, (size_t)cap * 16));
    ISL_TRY(lane_malloc(ws, ws.redo, (size_t)cap * 4));
    ISL_TRY(lane_malloc(ws, ws.replay, (size_t)cap * 4));
    ISL_TRY(lane_malloc(ws, ws.qsel, (size_t)cap * 4));
    ISL_TRY(lane_malloc(ws, ws.qsel_h, (size_t)cap * 4));
    ws.cap_q = cap;
  }
  uint64_t want_log = (uint64_t)ws.cap_q * plog_cap;
  if (ws.plog_entries < want_log) {
    if (ws.plog) (void)hipFree(ws.plog);
    ws.plog = nullptr;
    ws.plog_entries = 0;
    ISL_TRY(lane_malloc(ws, ws.plog, want_log * 8));
    ws.plog_entries = want_log;
  }
  return ISL_OK;
}

// Staging of the host-pointer entry points: device buffers + pinned host mirrors.
isl_status prepare_host_staging(isl::SearchWorkspace& ws, uint64_t nq, uint64_t d, uint64_t k) {
  const uint64_t qbytes = nq * d * 4;
  if (ws.q_stage_bytes < qbytes) {
    if (ws.q_stage) (void)hipFree(ws.q_stage);
    ws.q_stage = nullptr;
    ws.q_stage_bytes = 0;
    ISL_TRY(lane_malloc(ws, ws.q_stage, qbytes));
    ws.q_stage_bytes = qbytes;
  }
  if (ws.h_q_bytes < qbytes) {
    if (ws.h_q) (void)hipHostFree(ws.h_q);
    ws.h_q = nullptr;
    ws.h_q_bytes = 0;
    ISL_TRY(lane_host_malloc(ws, ws.h_q, qbytes));
    ws.h_q_bytes = qbytes;
  }
  const uint64_t slots = nq * std::max<uint64_t>(k, 1);
  if (ws.out_stage_slots < slots) {
    void* ptrs[] = {ws.ids_stage, ws.dist_stage, ws.count_stage};
    for (void* q : ptrs)
      if (q) (void)hipFree(q);
    ws.ids_stage = nullptr; ws.dist_stage = nullptr; ws.count_stage = nullptr;
    ws.out_stage_slots = 0;
    ISL_TRY(lane_malloc(ws, ws.ids_stage, slots * 8));
    ISL_TRY(lane_malloc(ws, ws.dist_stage, slots * 4));
    ISL_TRY(lane_malloc(ws, ws.count_stage, slots * 4));
    ws.out_stage_slots = slots;
  }
  if (ws.h_out_slots < slots) {
    void* ptrs[] = {ws.h_ids, ws.h_dist, ws.h_count};
    for (void* q : ptrs)
      if (q) (void)hipHostFree(q);
    ws.h_ids = nullptr; ws.h_dist = nullptr; ws.h_count = nullptr;
    ws.h_out_slots = 0;
    ISL_TRY(lane_host_malloc(ws, ws.h_ids, slots * 8));
    ISL_TRY(lane_host_malloc(ws, ws.h_dist, slots * 4));
    ISL_TRY(lane_host_malloc(ws, ws.h_count, slots * 4));
    ws.h_out_slots = slots;
  }
  return ISL_OK;
}

// The shared scratch pool of the heap-exact kernel (under idx->mu).  Its sizes follow the index
// (node / row count, longest row); the setters that change those drop the pool.
isl_status ensure_pool(const isl_index* idx, isl::SearchWorkspace& ws) {
  isl::ExactPool& pl = idx->pool;
  if (pl.slots) return ISL_OK;
  uint64_t max_id = std::max(idx->num_nodes, idx->nvec);
  pl.vis_words = (max_id + 31) / 32 + 1;
  pl.cand_cap = std::min<uint64_t>(max_id + 1, 1ull << 21);
  pl.ulist_cap = std::max<uint32_t>(idx->max_degree, 64);
  isl_status st = ISL_OK;
  if ((st = lane_malloc(ws, pl.cand_d, (size_t)kExactSlots * pl.cand_cap * 4)) == ISL_OK &&
      (st = lane_malloc(ws, pl.cand_id, (size_t)kExactSlots * pl.cand_cap * 4)) == ISL_OK &&
      (st = lane_malloc(ws, pl.vis_bits, (size_t)kExactSlots * pl.vis_words * 4)) == ISL_OK &&
      (st = lane_malloc(ws, pl.ulist, (size_t)kExactSlots * pl.ulist_cap * 4)) == ISL_OK &&
      (st = lane_malloc(ws, pl.locks, (size_t)kExactSlots * 4)) == ISL_OK &&
      (st = lane_malloc(ws, pl.xstate, (size_t)kExactSlots * (16 + 2 * (kMaxExactEf + 1)) * 4)) == ISL_OK) {
    pl.xstate_words = 16 + 2 * (kMaxExactEf + 1);
    if (hipMemset(pl.locks, 0, (size_t)kExactSlots * 4) == hipSuccess) {
      pl.slots = kExactSlots;
      return ISL_OK;
    }
    st = isl::fail(ISL_ERR_DEVICE, "hipMemset failed for the exact-kernel pool");
  }
  isl::free_exact_pool(pl);
  return st;
}

// CSR -> W (64 or 128) ids per node (EMPTY-padded) + degree; one wave per row
__global__ void pad_rows_kernel(const uint64_t* __restrict__ off, const uint32_t* __restrict__ adj, uint64_t n,
                                uint32_t W, uint32_t* __restrict__ ell, uint32_t* __restrict__ deg) {
  const uint64_t row = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63;
  if (row >= n) return;
  const uint64_t o0 = off[row];
  const uint32_t d = (uint32_t)(off[row + 1] - o0);
  for (uint32_t i = lane; i < W; i += 64) ell[row * W + i] = i < d ? adj[o0 + i] : EMPTY;
  if (lane == 0) deg[row] = d;
}

// Stream a call runs on.  OWN: the lane's non-blocking stream (host-pointer entry point).
// USER: the caller's stream (NULL = legacy default stream, ordered after the caller's earlier
// work on it, e.g. torch kernels that produced the queries).  OWN_AFTER_USER: the lane's stream,
// made to wait for everything already enqueued on the caller's stream -- lets several searches
// overlap (asynchronous entry point).
enum class StreamMode { OWN, USER, OWN_AFTER_USER };

// Launch geometry of one call: which kernel answers it and what its lane must hold.
struct CallGeometry {
  uint32_t ef = 0;
  bool use_fast = false;
  FastGeom fg{};
  uint32_t slots = 0;      // resident waves of the launch
  uint32_t vhint = 0;      // evaluations per query the visited table was sized for (0 = by ef alone)
  uint32_t plog_cap = 0;
  uint32_t tl_wcap = 0;
  size_t tl_lds = 0;
  uint32_t tl_hbits_q = 0;   // the bf16-query instantiation: visited-table bits, LDS, resident waves
  size_t tl_lds_q = 0;
  uint32_t tl_slots_q = 0;
};

isl_status call_geometry(const isl_index* idx, uint64_t d, uint64_t k, uint64_t ef_in, const TwoLevelCall* tl,
                         CallGeometry& g) {
  g.ef = (uint32_t)std::min<uint64_t>(std::max(ef_in, k), 0xFFFFFFFFull);  // leann.rs:890
  if (std::max(ef_in, k) > kMaxExactEf)
    return isl::fail(ISL_ERR_UNSUPPORTED, "ef = %llu exceeds the device limit %u",
                     (unsigned long long)std::max(ef_in, k), kMaxExactEf);
  const uint32_t ef = g.ef;
  const int ncu = isl::device_cu_count(idx->device);
  // (searches over the recompute provider park their visited table in state blocks sized by ef alone; the
  // two-level search sizes its own LDS: neither takes the hint)
  g.vhint = 0;
  if (!tl && !idx->recompute) {  // (the evaluations of a call with another ef say nothing about this one)
    const uint64_t h = idx->evals_hint.load(std::memory_order_relaxed);
    if ((uint32_t)(h >> 32) == ef) g.vhint = (uint32_t)h;
  }
  g.fg = fast_geometry(ef, (uint32_t)d, 4, g.vhint);
  g.use_fast = ef <= 512 && ef >= 1 && idx->max_degree <= 128;
  // resident waves per CU: bounded by LDS (visited table + query) and by the kernel's VGPR
  // budget (<= 128 -> 4 per SIMD)
  const size_t cu_cap = waves_per_cu_cap();
  uint32_t per_cu = (uint32_t)std::min<size_t>(cu_cap, (160 * 1024) / g.fg.lds);
  if (per_cu == 0) g.use_fast = false;
  if (tl) {
    // Window of the approximate queue: ceil(a * |AQ|) must stay inside it.  |AQ| is bounded by the
    // node count and runs at about 10 x ef (1235 at ef = 128 on the 10M-node bench graph); 20 x ef
    // covers the long queries, and one that outgrows it is re-run alone with four times the window
    // (never answered differently).  The window is most of a wave's LDS: round 2 sized it for
    // 32 x ef and ran 3 waves per CU at d = 4096.
    g.use_fast = false;
    const float a = tl->ratio > 0.0f ? std::min(tl->ratio, 1.0f) : 0.0f;
    const double bound = (double)std::min<uint64_t>(idx->ncodes, (uint64_t)20 * ef * tl->window_scale);
    const uint64_t want = (uint64_t)(a * bound) + 64;
    g.tl_wcap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>((want + 63) / 64 * 64, 256), 16384);
    while (g.tl_wcap > 256 && two_level_lds(g.fg.hbits, g.tl_wcap, ef, (uint32_t)d) > 160 * 1024) g.tl_wcap -= 64;
    g.tl_lds = two_level_lds(g.fg.hbits, g.tl_wcap, ef, (uint32_t)d);
    if (g.tl_lds > 160 * 1024)
      return isl::fail(ISL_ERR_UNSUPPORTED, "two-level search: ef = %u, d = %llu do not fit the LDS", ef,
                       (unsigned long long)d);
    per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(cu_cap, (160 * 1024) / g.tl_lds));
    // bf16 rows: the queries whose elements are bf16 values keep their query as bf16 in LDS
    g.tl_hbits_q = fast_geometry(ef, (uint32_t)d, 2).hbits;
    g.tl_lds_q = two_level_lds(g.tl_hbits_q, g.tl_wcap, ef, (uint32_t)d, 2);
    g.tl_slots_q = (uint32_t)ncu * (uint32_t)std::max<size_t>(1, std::min<size_t>(cu_cap, (160 * 1024) / g.tl_lds_q));
  }
  g.slots = std::max<uint32_t>(1, (uint32_t)ncu * std::max<uint32_t>(per_cu, 1));
  g.plog_cap = push_log_cap(ef);
  return ISL_OK;
}

// status / counters / work-queue heads of the call -> the lane's pinned mirrors, and for a
// host-buffer call its answers too; one kernel behind the search kernels.
isl_status publish(isl::SearchWorkspace& ws, uint64_t nq, uint64_t k, hipStream_t st) {
  PublishParams pp{};
  uint32_t n = 0;
  auto seg = [&](const void* src, void* dst, uint64_t words) {
    pp.src[n] = (const uint32_t*)src; pp.dst[n] = (uint32_t*)dst; pp.words[n] = (uint32_t)words; ++n;
  };
  seg(ws.status, ws.h_status, nq);
  seg(ws.ctr, ws.h_ctr, nq * 4);
  pp.ticket_seg = n;
  seg(ws.ticket, ws.h_head, 16);
  if (ws.publish_results) {
    seg(ws.count_stage, ws.h_count, nq);
    if (k) {
      seg(ws.ids_stage, ws.h_ids, nq * k * 2);
      seg(ws.dist_stage, ws.h_dist, nq * k);
    }
  }
  hipLaunchKernelGGL(publish_kernel, dim3(16, n), dim3(256), 0, st, pp);
  ISL_HIP(hipGetLastError());
  ws.ticket_clean = true;
  return ISL_OK;
}

// Enqueues the kernels of one search on a claimed lane; every pointer is a device pointer.
// warm = true: the same launches over zero queries (isl_index_prepare: loads the code objects and
// brings the lane's stream up) -- nothing is read or written beyond the ticket words.
isl_status search_enqueue_impl(const isl_index* idx, isl::SearchWorkspace& ws, const float* d_queries,
                               uint64_t nq, uint64_t d, uint64_t k, uint64_t ef_in, uint64_t* d_ids,
                               float* d_dist, uint32_t* d_count, hipStream_t user_stream,
                               StreamMode mode, const TwoLevelCall* tl, bool warm) {
  if (nq > 0x7FFFFFFFull) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "too many queries");
  CallGeometry cg;
  ISL_TRY(call_geometry(idx, d, k, ef_in, tl, cg));
  const uint32_t ef = cg.ef;
  const bool use_fast = cg.use_fast;
  const FastGeom& fg = cg.fg;
  const uint32_t slots = cg.slots, plog_cap = cg.plog_cap;
  // searches over the recompute provider park and resume on the fast kernel (f32 rows; the
  // two-level and heap-exact kernels re-run a blocked query from its start instead)
  const bool resume = !warm && (ws.round_active != 0 || ws.round_x != 0);
  // two-level search over bf16 rows: first the instantiation that keeps a bf16-valued query as bf16
  // in LDS, then the float32-query one over the queries it passed on (not in a retry's list mode)
  static const bool no_tl_qh = getenv("ISL_NO_TL_QH") != nullptr;  // A/B switch for measurements
  const bool tl_qh = tl && !warm && !resume && idx->d_emb16 && ws.retry_count == 0 && !no_tl_qh &&
                     cg.tl_hbits_q == cg.fg.hbits;
  // bf16 rows: first the kernel that keeps the query as bf16 in LDS (half the LDS per wave, more
  // waves per CU), then the float32-query kernel over the queries that one passed on
  static const bool no_qh = getenv("ISL_NO_QH") != nullptr;  // A/B switch for measurements
  const bool qh = use_fast && !tl && !resume && idx->d_emb16 && idx->max_degree <= 64 && !no_qh;
  FastGeom fgq = fg;
  uint32_t slots_q = slots;
  if (qh) {
    fgq = fast_geometry(ef, (uint32_t)d, 2, cg.vhint);
    slots_q = (uint32_t)isl::device_cu_count(idx->device) *
              (uint32_t)std::max<size_t>(1, std::min<size_t>(waves_per_cu_cap(), (160 * 1024) / fgq.lds));
  }
  // per-slot state is indexed by blockIdx.x < min(nq, slots) -- by the query when it can come back on
  // another wave
  if (!warm)
    ISL_TRY(prepare_workspace(idx, ws, (uint32_t)nq,
                              (uint32_t)(idx->recompute ? nq : std::min<uint64_t>(nq, std::max(std::max(slots, slots_q), cg.tl_slots_q))),
                              plog_cap));
  if (!idx->pool.slots || ((use_fast || (tl && idx->max_degree <= 128)) && !idx->d_ell && idx->d_off && idx->num_nodes)) {
    // not prepared (isl_index_prepare / isl_index_upload do this ahead of time)
    std::lock_guard<std::mutex> lock(idx->mu);
    ISL_TRY(ensure_pool(idx, ws));
    const uint32_t* before = idx->d_ell;
    ISL_TRY(isl::ensure_padded_adjacency(const_cast<isl_index*>(idx)));
    if (idx->d_ell != before) ws.alloc_events += 2;
  }
  hipStream_t st = mode == StreamMode::USER ? user_stream : ws.stream;
  if (mode == StreamMode::OWN_AFTER_USER) {
    ISL_HIP(hipEventRecord(ws.ev_in, user_stream));
    ISL_HIP(hipStreamWaitEvent(ws.stream, ws.ev_in, 0));
  }

  SearchParams p{};
  p.off = idx->d_off;
  p.adj = idx->d_ell ? idx->d_ell : idx->d_adj;
  p.ell_w = idx->d_ell ? idx->ell_w : 0u;
  p.ell_deg = idx->d_ell_deg;
  p.num_nodes = idx->num_nodes;
  p.emb = idx->d_emb16 ? (const void*)idx->d_emb16 : (const void*)idx->d_emb;
  p.emb_bf16 = idx->d_emb16 ? 1u : 0u;
  p.norm2 = idx->d_norm2;
  p.nvec = idx->nvec;
  p.stride = idx->emb_stride;
  p.d = (uint32_t)d;
  p.queries = d_queries;
  p.nq = warm ? 0u : (uint32_t)nq;
  p.k = (uint32_t)k;
  p.ef = ef;
  p.prune_ratio = idx->cfg.prune_ratio;
  p.prune_strategy = idx->cfg.pruning_strategy;
  p.entry = (uint32_t)std::min<uint64_t>(idx->entry_point, 0x7FFFFFF0ull);
  p.out_ids = d_ids;
  p.out_dist = d_dist;
  p.out_count = d_count;
  p.status = ws.status;
  p.payload = ws.payload;
  p.ctr = ws.ctr;
  p.ticket = ws.ticket;
  p.redo = ws.redo;
  if (ws.d_prof) { (void)hipFree(ws.d_prof); ws.d_prof = nullptr; }
  static const bool debug_env = getenv("ISL_DEBUG") != nullptr;
  if (debug_env && !warm) {
    ISL_HIP(hipMalloc(&ws.d_prof, nq * 64));
    ISL_HIP(hipMemset(ws.d_prof, 0, nq * 64));
  }
  p.prof = ws.d_prof;
  // ISL_TIMELINE=<file>: start / end tick of every query of every call, appended at wait time
  // (measurement aid: where a run's fill and drain go; tools/timeline.py reads the file)
  static const char* tline_env = getenv("ISL_TIMELINE");
  if (ws.d_tline) { (void)hipFree(ws.d_tline); ws.d_tline = nullptr; }
  if (tline_env && !warm) {
    ISL_HIP(hipMalloc(&ws.d_tline, nq * 16));
    ISL_HIP(hipMemset(ws.d_tline, 0, nq * 16));
  }
  p.tline = ws.d_tline;
  p.replay = ws.replay;
  p.plog = reinterpret_cast<uint2*>(ws.plog);
  p.plog_cap = plog_cap;
  p.hbits = fg.hbits;
  p.hcap = fg.hcap;
  p.otab = ws.ovf_tab;
  p.obits = ws.ovf_bits;
  p.cand_d = idx->pool.cand_d;
  p.cand_id = idx->pool.cand_id;
  p.cand_cap = idx->pool.cand_cap;
  p.vis_bits = idx->pool.vis_bits;
  p.vis_words = idx->pool.vis_words;
  p.ulist = idx->pool.ulist;
  p.ulist_cap = idx->pool.ulist_cap;
  p.pool_locks = idx->pool.locks;
  p.pool_slots = idx->pool.slots;
  if (resume && !tl && ws.round_xpark) {  // the heap-exact kernel parks and resumes too (recompute provider, bounded cache)
    p.xslot = ws.xslot;
    p.xstate = idx->pool.xstate;
    p.xstate_words = idx->pool.xstate_words;
  }
  p.slot_of = idx->recompute ? idx->d_slot_of : nullptr;
  p.stamp = idx->d_stamp;
  p.round_no = idx->round_no;
  if (resume) {
    p.qstate = ws.qstate;
    p.qstate_words = tl ? isl_launch::tl_state_words(ef, cg.tl_wcap, fg.hbits)
                        : isl_launch::fast_state_words(ef <= 64 ? 1 : ef <= 128 ? 2 : ef <= 256 ? 4 : 8, fg.hbits);
    p.qflag = ws.qflag;
    p.qlist = ws.round_listed ? ws.qlist : nullptr;
  } else if (tl && ws.retry_count) {
    p.qlist = ws.qlist;  // the queries whose queue window was too small, alone, with a larger one
  }
  p.miss = ws.miss;
  p.miss_cap = (uint32_t)std::min<uint64_t>(ws.miss_cap, 0xFFFFFFFFull);
  p.pref = ws.miss ? ws.miss + ws.miss_cap : nullptr;
  p.pref_cap = (uint32_t)ws.pref_cap;
  p.tl_prefetch = (tl && resume && ws.pref_cap) ? ws.round_prefetch : 0u;
  p.layer_off = idx->d_layer_off;
  p.layer_adj = idx->d_layer_adj;
  p.max_level = idx->is_hnsw ? (uint32_t)idx->max_level : 0u;
  p.hnsw_order = idx->is_hnsw ? 1u : 0u;
  p.q_entry = nullptr;
  p.q_evals = nullptr;
  static const uint32_t seq_max_env = [] { const char* e = getenv("ISL_SEQ_MAX"); return e ? (uint32_t)atoi(e) : 0u; }();
  p.seq_max = seq_max_env;
  if (tl) {
    const isl_pq* pq = idx->pq;
    const uint64_t want = std::max<uint64_t>(nq, 1) * pq->m * pq->K;
    ISL_TRY(ensure(ws, ws.tl_tables, ws.tl_tables_cap, want));
    p.tl_tables = ws.tl_tables;
    p.tl_codes = idx->d_codes;
    p.tl_ncodes = idx->ncodes;
    p.tl_m = (uint32_t)pq->m;
    p.tl_K = (uint32_t)pq->K;
    p.tl_ratio = tl->ratio;
    p.tl_wcap = cg.tl_wcap;
  }
  const uint64_t nq_grid = warm ? 1 : nq;  // a warm launch needs one workgroup to exist

  // the work-queue heads are zero at this point: the publish kernel of the lane's previous call
  // left them so; only a lane's first call (or one after a failed enqueue) clears them itself
  if (!ws.ticket_clean) ISL_HIP(hipMemsetAsync(ws.ticket, 0, 64, st));
  ws.ticket_clean = false;
  ISL_HIP(hipEventRecord(ws.ev0, st));
  if (resume && !tl && ws.round_x) {
    hipLaunchKernelGGL(seed_redo_kernel, dim3(1), dim3(256), 0, st, ws.h_xlist, ws.round_x, ws.redo, ws.ticket);
    ISL_HIP(hipGetLastError());
  }
  if (tl) {
    // build_distance_tables for the whole batch (pq.rs:307-338; once per call: the rounds of the
    // recompute provider and a retry keep them), then one wave per query
    if (!warm && !ws.tl_tables_built) {
      ISL_TRY(isl::pq_launch_tables(idx->pq, d_queries, nq, ws.tl_tables, st));
      ws.tl_tables_built = true;
    }
    const uint64_t n_run = resume ? ws.round_active : ws.retry_count ? ws.retry_count : nq_grid;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(n_run, slots);
    const int metric = (int)idx->cfg.metric;
    p.nq = warm ? 0u : (uint32_t)n_run;
    if (tl_qh) {
      p.qsel = ws.qsel;
      p.qsel_h = ws.qsel_h;
      isl_launch::launch_classify((uint32_t)std::min<uint64_t>(nq_grid, 2048), st, &p);
      ISL_HIP(hipGetLastError());
      isl_launch::launch_two_level(metric, true, false, true, (uint32_t)std::min<uint64_t>(nq_grid, cg.tl_slots_q),
                                   cg.tl_lds_q, st, &p);
      ISL_HIP(hipGetLastError());
      p.qsel_mode = 1;  // the others
      isl_launch::launch_two_level(metric, true, false, false, grid, cg.tl_lds, st, &p);
      p.qsel_mode = 0;
    } else {
      isl_launch::launch_two_level(metric, p.emb_bf16 != 0, resume || (warm && idx->recompute), false, grid, cg.tl_lds, st, &p);
    }
    ISL_HIP(hipGetLastError());
    p.nq = warm ? 0u : (uint32_t)nq;
  } else
  if (use_fast && idx->is_hnsw && p.max_level > 0) {
    // HnswGraph::search: greedy descent through the upper layers first (its own kernel, so that
    // the traversal kernel keeps its register budget)
    ISL_TRY(ensure(ws, ws.q_entry, ws.q_entry_cap, std::max<uint64_t>(nq, 1) * 2));
    p.q_entry = ws.q_entry;
    p.q_evals = ws.q_entry + nq;
    const size_t dlds = (size_t)((d + 3) / 4 * 4) * 4 + 64;
    const uint32_t dgrid = (uint32_t)std::min<uint64_t>(nq_grid, 8192);
    isl_launch::launch_descent((int)idx->cfg.metric, dgrid, dlds, st, &p);
    ISL_HIP(hipGetLastError());
  }
  if (tl) {
  } else if (use_fast) {
    if (resume) p.nq = ws.round_active;  // the fast kernel runs this round's queries; nothing else reads nq
    uint32_t grid = (uint32_t)std::min<uint64_t>(resume ? ws.round_active : nq_grid, slots);
    const bool skip_fast = resume && ws.round_active == 0;  // only queries parked in the heap-exact kernel this round
    int S = ef <= 64 ? 1 : ef <= 128 ? 2 : ef <= 256 ? 4 : 8;
    const int metric = (int)idx->cfg.metric;
    const bool wide = idx->max_degree > 64;
    const bool bf16 = p.emb_bf16 != 0;
    auto launch_fast = [&](bool q16, uint32_t g, size_t lds_bytes) {
      switch (S) {
        case 1: isl_launch::launch_fast_s1(metric, wide, bf16, resume, q16, g, lds_bytes, st, &p); break;
        case 2: isl_launch::launch_fast_s2(metric, wide, bf16, resume, q16, g, lds_bytes, st, &p); break;
        case 4: isl_launch::launch_fast_s4(metric, wide, bf16, resume, q16, g, lds_bytes, st, &p); break;
        default: isl_launch::launch_fast_s8(metric, wide, bf16, resume, q16, g, lds_bytes, st, &p); break;
      }
    };
    if (qh) {
      p.qsel = ws.qsel;
      p.qsel_h = ws.qsel_h;
      isl_launch::launch_classify((uint32_t)std::min<uint64_t>(nq_grid, 2048), st, &p);
      ISL_HIP(hipGetLastError());
      p.hbits = fgq.hbits;
      p.hcap = fgq.hcap;
      launch_fast(true, (uint32_t)std::min<uint64_t>(nq_grid, slots_q), fgq.lds);  // the bf16-valued queries
      ISL_HIP(hipGetLastError());
      p.hbits = fg.hbits;
      p.hcap = fg.hcap;
      p.qsel_mode = 1;  // the others
      launch_fast(false, grid, fg.lds);
      p.qsel_mode = 0;
    } else if (!skip_fast) {
      launch_fast(false, grid, fg.lds);
    }
    ISL_HIP(hipGetLastError());
    p.nq = (uint32_t)nq;
  } else if (!warm && !resume) {
    // every query goes to the exact kernel: redo = [0, nq)
    std::vector<uint32_t> all(nq);
    for (uint64_t i = 0; i < nq; i++) all[i] = (uint32_t)i;
    uint32_t head0[4] = {0, (uint32_t)nq, 0, 0};
    ISL_HIP(hipMemcpyAsync(ws.redo, all.data(), nq * 4, hipMemcpyHostToDevice, st));
    ISL_HIP(hipMemcpyAsync(ws.ticket, head0, 16, hipMemcpyHostToDevice, st));
    ISL_HIP(hipStreamSynchronize(st));
  }
  if (!tl) {
    uint32_t grid = (uint32_t)std::min<uint64_t>(nq_grid, idx->pool.slots);
    isl_launch::launch_exact((int)idx->cfg.metric, idx->is_hnsw, grid, exact_lds(ef, (uint32_t)d), st, &p);
    ISL_HIP(hipGetLastError());
  }
  ISL_HIP(hipEventRecord(ws.ev1, st));
  ws.st_inflight = st;
  if (warm) return ISL_OK;

  ISL_TRY(publish(ws, nq, k, st));
  // the call's own completion event: lanes may share a stream, and a caller's stream carries its other work
  if (ws.ev_done) ISL_HIP(hipEventRecord(ws.ev_done, st));
  ws.enqueued = true;
  ws.nq_inflight = nq;
  ws.k_inflight = k;
  ws.ef_inflight = ef;
  ws.fast_inflight = use_fast;
  return ISL_OK;
}

// A failure part-way through an enqueue (a launch error, publish, a lane buffer that could not grow)
// may leave kernels of this call on the stream: they are drained before the error goes back, because
// the caller releases the lane next and the lane's next owner rewrites its pinned buffers / may
// reallocate what those kernels still read.
isl_status search_enqueue(const isl_index* idx, isl::SearchWorkspace& ws, const float* d_queries,
                          uint64_t nq, uint64_t d, uint64_t k, uint64_t ef_in, uint64_t* d_ids,
                          float* d_dist, uint32_t* d_count, hipStream_t user_stream,
                          StreamMode mode, const TwoLevelCall* tl = nullptr, bool warm = false) {
  const isl_status st = search_enqueue_impl(idx, ws, d_queries, nq, d, k, ef_in, d_ids, d_dist, d_count, user_stream,
                                            mode, tl, warm);
  if (st != ISL_OK) {
    const isl::ErrorRecord keep = isl::last_error();
    hipStream_t s = mode == StreamMode::USER ? user_stream : ws.stream;
    if (mode == StreamMode::USER || s) (void)hipStreamSynchronize(s);
    (void)hipGetLastError();
    ws.ticket_clean = false;
    ws.enqueued = false;
    isl::last_error() = keep;
  }
  return st;
}

// counters of the most recent call this thread completed, per index (isl_search_last_stats)
struct LastStats { const isl_index* idx = nullptr; isl_search_stats st{}; };
thread_local LastStats tl_last_stats;
void note_last_stats(const isl_index* idx, const isl_search_stats& st) {
  tl_last_stats.idx = idx;
  tl_last_stats.st = st;
}

// Waits for the call in flight on `ws`, leaves its counters in ws.stats and turns per-query
// failures into the CoreError the reference's sequential map would have returned.
isl_status search_statuses(isl::SearchWorkspace& ws, uint64_t nq, bool* window_short);

// defer_statuses: the per-query statuses are not final yet (a batch the recompute provider works
// through in rounds, some of its queries not even started): the caller runs search_statuses itself.
isl_status search_finish(const isl_index* idx, isl::SearchWorkspace& ws, uint32_t* misses = nullptr,
                         bool* window_short = nullptr, bool defer_statuses = false) {
  if (!ws.enqueued) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "no search in flight for this token");
  ws.enqueued = false;
  const uint64_t nq = ws.nq_inflight;
  const bool use_fast = ws.fast_inflight;
  static const bool sync_stream = getenv("ISL_SYNC_STREAM") != nullptr;  // A/B switch for measurements
  if (ws.ev_done && !sync_stream) ISL_HIP(hipEventSynchronize(ws.ev_done));
  else ISL_HIP(hipStreamSynchronize(ws.st_inflight));
  const uint32_t* status = ws.h_status;
  const uint32_t* ctr = ws.h_ctr;
  const uint32_t* head = ws.h_head;
  uint64_t* d_prof = ws.d_prof;
  ws.d_prof = nullptr;
  float ms = 0.0f;
  (void)hipEventElapsedTime(&ms, ws.ev0, ws.ev1);

  isl_search_stats& ss = ws.stats;
  ss = isl_search_stats{};
  ss.queries = nq;
  ss.exact_path = head[1];
  ss.replayed = head[3];
  ss.kernel_ms = ms;
  ss.allocations = ws.alloc_events - ws.alloc_mark;
  for (uint64_t i = 0; i < nq; i++) {
    ss.expansions += ctr[i * 4 + 0];
    ss.edges += ctr[i * 4 + 1];
    ss.evals += ctr[i * 4 + 2];
    ss.pushes += ctr[i * 4 + 3];
  }
  // what the next calls size their visited table by (fast_geometry): the evaluations per query of this one
  if (use_fast && !idx->recompute && !misses && nq >= 16)
    idx->evals_hint.store(((uint64_t)ws.ef_inflight << 32) | std::min<uint64_t>(ss.evals / nq, 0xFFFFFFFFull),
                          std::memory_order_relaxed);
  if (ws.d_tline) {
    std::vector<uint64_t> tl(nq * 2 + 2);
    tl[0] = 0x154C494E45ull;  // record header: magic, query count
    tl[1] = nq;
    ISL_HIP(hipMemcpy(tl.data() + 2, ws.d_tline, nq * 16, hipMemcpyDeviceToHost));
    (void)hipFree(ws.d_tline);
    ws.d_tline = nullptr;
    static std::mutex tl_mu;
    std::lock_guard<std::mutex> lock(tl_mu);
    if (FILE* f = fopen(getenv("ISL_TIMELINE"), "ab")) {
      fwrite(tl.data(), 8, tl.size(), f);
      fclose(f);
    }
  }
  if (d_prof) {
    std::vector<uint64_t> pr(nq * 8);
    ISL_HIP(hipMemcpy(pr.data(), d_prof, nq * 64, hipMemcpyDeviceToHost));
    (void)hipFree(d_prof);
    double sum[4] = {0, 0, 0, 0}, grp = 0, hr = 0;
    for (uint64_t i = 0; i < nq; i++) {
      for (int j = 0; j < 4; j++) sum[j] += pr[i * 8 + j] / 100.0;
      grp += pr[i * 8 + 4];
      hr += pr[i * 8 + 5];
    }
    fprintf(stderr, "[isl] per query: %.1f hops with new rows, %.1f distance passes of <= 16 rows\n", hr / nq,
            grp / nq);
    fprintf(stderr, "[isl] mean us per query by phase: select+adjacency %.0f, visited %.0f, rows+distance %.0f, "
            "insert %.0f\n", sum[0] / nq, sum[1] / nq, sum[2] / nq, sum[3] / nq);
  }
  if (getenv("ISL_DEBUG")) {
    std::vector<uint64_t> pay(nq);
    ISL_HIP(hipMemcpy(pay.data(), ws.payload, nq * 8, hipMemcpyDeviceToHost));
    std::vector<double> us;
    std::vector<std::pair<double, uint32_t>> byq;
    for (uint64_t i = 0; i < nq; i++)
      if (status[i] == QS_OK && pay[i]) { us.push_back(pay[i] / 100.0); byq.push_back({pay[i] / 100.0, ctr[i * 4]}); }
    std::sort(us.begin(), us.end());
    std::sort(byq.begin(), byq.end());
    if (!us.empty())
      fprintf(stderr, "[isl] per-query time in fast kernel (us): min %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f; "
              "hops of slowest %u, of median %u; kernel %.0f us\n", us.front(), us[us.size() / 2],
              us[us.size() * 9 / 10], us[us.size() * 99 / 100], us.back(), byq.back().second,
              byq[byq.size() / 2].second, ms * 1000.0);
  }
  if (getenv("ISL_DEBUG") && (head[1] || head[3])) {
    fprintf(stderr, "[isl] %u of %llu queries re-run by the exact kernel (fast kernel %s): "
            "long-row %u, visited-overflow %u, tie-candidate overflow %u, push-log overflow %u, "
            "NaN/-0 distance %u; %u re-ordered by the replay kernel\n", head[1], (unsigned long long)nq,
            use_fast ? "on" : "off", head[9], head[10], head[11], head[8], head[12], head[3]);
  }
  if (misses) {
    *misses = head[13];
    if (head[13]) return ISL_OK;  // a round of the recompute provider: statuses are not final yet
  }
  if (defer_statuses) return ISL_OK;
  return search_statuses(ws, nq, window_short);
}

// per-query failures -> the CoreError the reference's sequential map would have returned
isl_status search_statuses(isl::SearchWorkspace& ws, uint64_t nq, bool* window_short) {
  const uint32_t* status = ws.h_status;
  for (uint64_t i = 0; i < nq; i++) {  // first failing query wins, like the sequential map
    if (status[i] == QS_OK) continue;
    if (status[i] == QS_NODE_NOT_FOUND) {
      uint64_t node = 0;
      ISL_HIP(hipMemcpy(&node, ws.payload + i, 8, hipMemcpyDeviceToHost));
      return isl::fail_node(node);
    }
    if (status[i] == QS_SCRATCH && window_short) {  // two-level search: 7 = the queue window was too small
      uint64_t why = 0;
      ISL_HIP(hipMemcpy(&why, ws.payload + i, 8, hipMemcpyDeviceToHost));
      *window_short = why == 7;
    }
    if (status[i] == QS_SCRATCH)
      return isl::fail(ISL_ERR_SEARCH, "Search error: device scratch exhausted for query %llu (candidate heap, "
                       "visited table or the two-level search's approximate-queue window)",
                       (unsigned long long)i);
    return isl::fail(ISL_ERR_SEARCH, "Search error: query %llu left in state 0x%x",
                     (unsigned long long)i, status[i]);
  }
  return ISL_OK;
}

// ---- recompute provider (EmbeddingProvider backed by the encoder, leann.rs:82-99) ----
// lists each missed id once: the first reporter of an id claims its slot-map entry
__global__ void dedupe_misses_kernel(const uint32_t* __restrict__ miss, uint32_t n,
                                     uint32_t* __restrict__ slot_of, uint32_t* __restrict__ uniq,
                                     uint32_t* __restrict__ uniq_count) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t id = miss[i];
  if (atomicCAS(&slot_of[id], kNoSlot, kSlotClaim) == kNoSlot) uniq[atomicAdd(uniq_count, 1u)] = id;
}

// Hands the round's unique misses their slab slots (one wave): once the slab is full a clock hand walks it and
// skips the rows some query asked for in this round (stamp == round) -- those belong to hops that
// are waiting for their last rows, evicting them would make the hop wait for THEM next round.  The
// node that held a slot before loses its row.  Ids left over when a full turn finds no more free
// slots are un-claimed and reported again next round.
__global__ __launch_bounds__(64) void assign_slots_kernel(const uint32_t* __restrict__ uniq,
                                                          const uint32_t* __restrict__ n_ptr,
                                                          uint32_t round_no, uint32_t slab_rows,
                                                          uint32_t* __restrict__ head_word,
                                                          uint32_t* __restrict__ slot_of, uint32_t* __restrict__ owner,
                                                          uint32_t* __restrict__ stamp, uint32_t* __restrict__ uslots,
                                                          uint32_t* __restrict__ taken, uint32_t quantum,
                                                          uint32_t chunk) {
  const uint32_t lane = threadIdx.x;
  // How many of the round's misses are encoded now: the encoder's GEMMs run whole waves of tiles over the
  // chip's CUs, and a batch that ends a little past a full wave pays for a whole one more (860 nodes x 64
  // tokens: 645 tiles of the hidden x hidden GEMMs = 2.52 waves on 256 CUs, 84 % of them filled).  So a
  // round takes whole encoder passes of `chunk` nodes plus a multiple of `quantum` nodes (the largest batch
  // whose narrowest GEMM still fits ONE wave of tiles), plus the rest when that rest nearly fills a wave
  // anyway; what is left over is un-claimed below and reported again next round, when it is batched with
  // that round's misses.  quantum == 0: everything (a provider whose shapes were not analysed).
  const uint32_t n_all = *n_ptr;
  uint32_t n = n_all;
  if (quantum && n >= quantum) {
    const uint32_t whole = chunk ? (n / chunk) * chunk : 0u, rem = n - whole;
    const uint32_t r = rem % quantum;
    n = whole + (rem - r) + (r * 10u >= quantum * 9u ? r : 0u);
  }
  // never-used slots first (head_word[1] counts them): nothing is evicted before the slab is full
  uint32_t fill = head_word[1], done = 0;
  {
    const uint32_t t = n < slab_rows - fill ? n : slab_rows - fill;
    for (uint32_t i = lane; i < t; i += 64) {
      const uint32_t id = uniq[i], s = fill + i;
      owner[s] = id;
      slot_of[id] = s;
      stamp[s] = round_no;
      uslots[i] = s;
    }
    done = t;
    fill += t;
  }
  uint32_t pos = *head_word % slab_rows, walked = 0;
  while (done < n && walked < slab_rows) {
    const uint32_t step = slab_rows - walked < 64u ? slab_rows - walked : 64u;
    const uint32_t s = (pos + lane) % slab_rows;
    const bool free_ = lane < step && stamp[s] != round_no;
    const uint64_t fm = ballot(free_);
    const uint32_t i = done + (uint32_t)__popcll(fm & ((1ull << lane) - 1ull));
    if (free_ && i < n) {
      const uint32_t id = uniq[i];
      const uint32_t old = owner[s];
      if (old != kNoSlot) slot_of[old] = kNoSlot;  // (never one of this round's ids: those were absent)
      owner[s] = id;
      slot_of[id] = s;
      stamp[s] = round_no;
      uslots[i] = s;
    }
    done += (uint32_t)__popcll(fm);
    pos = (pos + step) % slab_rows;
    walked += step;
  }
  if (done > n) done = n;
  for (uint32_t i = done + lane; i < n_all; i += 64) slot_of[uniq[i]] = kNoSlot;  // (the quantum's left-overs too)
  if (lane == 0) { head_word[0] = pos; head_word[1] = fill; *taken = done; }
}

// norm2[id] = sum_j row[id][j]^2 in the reference's order for the freshly encoded rows
__global__ __launch_bounds__(64) void row_norm2_list_kernel(const float* __restrict__ emb, uint64_t stride,
                                                            uint32_t d, const uint32_t* __restrict__ ids,
                                                            uint32_t n, float* __restrict__ norm2) {
  extern __shared__ __align__(16) unsigned char smem[];
  float* tile = reinterpret_cast<float*>(smem);
  const uint32_t lane = threadIdx.x;
  for (uint32_t base = blockIdx.x * 64; base < n; base += gridDim.x * 64) {
    const uint32_t R = n - base < 64 ? n - base : 64;
    const uint32_t uid = lane < R ? ids[base + lane] : 0u;
    const float v = wave_distances<METRIC_SUMSQ_RAW>(emb, stride, d, uid, R, tile, tile, 0.f);
    if (lane < R) norm2[uid] = v;
  }
}

// query lists of a lane (rounds of the recompute provider, retries of the two-level search)
isl_status ensure_qlist(isl::SearchWorkspace& ws, uint64_t nq) {
  if (ws.qlist_cap >= nq && ws.qlist && ws.h_qlist && ws.qflag && ws.xslot && ws.h_xlist) return ISL_OK;
  if (ws.qflag) (void)hipFree(ws.qflag);
  if (ws.qlist) (void)hipFree(ws.qlist);
  if (ws.xslot) (void)hipFree(ws.xslot);
  if (ws.h_qlist) (void)hipHostFree(ws.h_qlist);
  if (ws.h_xlist) (void)hipHostFree(ws.h_xlist);
  ws.qflag = ws.qlist = ws.xslot = ws.h_qlist = ws.h_xlist = nullptr;
  ws.qlist_cap = 0;
  const uint64_t c = nq < 1024 ? 1024 : nq;
  ISL_TRY(lane_malloc(ws, ws.qflag, c * 4));
  ISL_TRY(lane_malloc(ws, ws.qlist, c * 4));
  ISL_TRY(lane_malloc(ws, ws.xslot, c * 4));
  ISL_TRY(lane_host_malloc(ws, ws.h_qlist, c * 4));
  ISL_TRY(lane_host_malloc(ws, ws.h_xlist, c * 4));
  ws.qlist_cap = c;
  return ISL_OK;
}

// Two-level search: the queries of the finished launch whose approximate-queue window was too small
// (QS_SCRATCH with payload 7) -> ws.h_qlist[0, *count).
isl_status tl_collect_short(isl::SearchWorkspace& ws, uint64_t nq, uint32_t* count) {
  *count = 0;
  bool any = false;
  for (uint64_t i = 0; i < nq && !any; ++i) any = ws.h_status[i] == QS_SCRATCH;
  if (!any) return ISL_OK;
  ISL_TRY(ensure_qlist(ws, nq));
  std::vector<uint64_t> pay(nq);
  ISL_HIP(hipMemcpy(pay.data(), ws.payload, nq * 8, hipMemcpyDeviceToHost));
  uint32_t n = 0;
  for (uint64_t i = 0; i < nq; ++i)
    if (ws.h_status[i] == QS_SCRATCH && pay[i] == 7) ws.h_qlist[n++] = (uint32_t)i;
  *count = n;
  return ISL_OK;
}

isl_status prepare_recompute(isl::SearchWorkspace& ws, uint64_t nq, uint64_t state_words_per_query) {
  const uint64_t cap = std::min<uint64_t>(nq * 128 + 64, 0xFFFFFFF0ull);  // a hop keeps up to 128 rows
  const uint64_t pcap = nq * 8 + 64;  // + the ids parked two-level queries expect to promote next (behind miss[cap])
  if (ws.miss_cap < cap || ws.pref_cap < pcap) {
    void* ptrs[] = {ws.miss, ws.uniq, ws.uniq_count, ws.uslots};
    for (void* q : ptrs)
      if (q) (void)hipFree(q);
    ws.miss = ws.uniq = ws.uniq_count = ws.uslots = nullptr;
    ws.miss_cap = 0;
    ws.pref_cap = 0;
    ISL_TRY(lane_malloc(ws, ws.miss, (cap + pcap) * 4));
    ISL_TRY(lane_malloc(ws, ws.uniq, (cap + pcap) * 4));
    ISL_TRY(lane_malloc(ws, ws.uslots, (cap + pcap) * 4));
    ISL_TRY(lane_malloc(ws, ws.uniq_count, 4));
    ws.miss_cap = cap;
    ws.pref_cap = pcap;
  }
  ISL_TRY(ensure_qlist(ws, nq));
  ISL_TRY(ensure(ws, ws.qstate, ws.qstate_words, std::max<uint64_t>(nq, 1) * state_words_per_query));
  return ISL_OK;
}

// One synchronous search on a claimed lane.  With the in-memory provider: enqueue + finish.  With
// the recompute provider: rounds of (search; every query that needs an absent row reports it and
// stops) -> (encode the reported nodes once each) until a round completes without a miss; that
// last round is an ordinary search over materialised rows, so ids, distances, counters and error
// behaviour are those of the in-memory provider holding the same embeddings.
isl_status search_sync(const isl_index* idx, isl::SearchWorkspace& ws, const float* d_queries,
                       uint64_t nq, uint64_t d, uint64_t k, uint64_t ef, uint64_t* d_ids,
                       float* d_dist, uint32_t* d_count, hipStream_t user_stream, StreamMode mode,
                       const TwoLevelCall* tl = nullptr) {
  // two-level search: a query whose approximate queue outgrew the LDS window is never answered
  // differently: the queries it happened to are run again, alone, with a window four times the size
  TwoLevelCall tcall;
  if (tl) { tcall = *tl; tl = &tcall; }
  struct CallReset {  // the lane goes back with its per-call two-level fields cleared whatever happens below
    isl::SearchWorkspace& w;
    ~CallReset() { w.retry_count = 0; w.tl_tables_built = false; }
  } call_reset{ws};
  ws.retry_count = 0;
  ws.tl_tables_built = false;
  if (!idx->recompute) {
    double ms_total = 0.0;
    for (;;) {
      ISL_TRY(search_enqueue(idx, ws, d_queries, nq, d, k, ef, d_ids, d_dist, d_count, user_stream, mode, tl));
      ISL_TRY(search_finish(idx, ws, nullptr, nullptr, tl != nullptr));
      if (!tl) return ISL_OK;  // (statuses evaluated by search_finish)
      ms_total += ws.stats.kernel_ms;
      uint32_t nshort = 0;
      ISL_TRY(tl_collect_short(ws, nq, &nshort));
      if (nshort && tcall.window_scale < 64) {
        tcall.window_scale *= 4;
        ws.retry_count = nshort;
        hipStream_t st = mode == StreamMode::USER ? user_stream : ws.stream;
        hipLaunchKernelGGL(copy_u32_kernel, dim3(16), dim3(256), 0, st, ws.h_qlist, ws.qlist, (uint64_t)nshort);
        ISL_HIP(hipGetLastError());
        continue;
      }
      ws.stats.kernel_ms = ms_total;
      return search_statuses(ws, nq, nullptr);
    }
  }
  // the rounds rewrite the provider's row cache: one recompute search at a time
  std::lock_guard<std::mutex> rlock(idx->recompute_mu);
  CallGeometry cg0;
  ISL_TRY(call_geometry(idx, d, k, ef, tl, cg0));
  const int S0 = cg0.ef <= 64 ? 1 : cg0.ef <= 128 ? 2 : cg0.ef <= 256 ? 4 : 8;
  // searches that park and resume: the wave-per-query traversal and the two-level search (the
  // heap-exact kernel alone -- ef > 512, rows past 128 ids -- re-runs a blocked query from its start)
  // searches park and resume: the wave-per-query traversal, the two-level search, and -- since round 3 --
  // the heap-exact kernel (ef > 512, rows past 128 ids, tie hand-overs), whose parked queries keep their
  // slot of the scratch pool across the rounds
  // -- when the row cache is bounded.  With a row for every node nothing is ever evicted, a blocked query
  // of that kernel simply starts over next round (all of them advance in parallel, where parked ones
  // would advance 32 at a time: the pool's slots).
  const bool exact_only = !tl && !cg0.use_fast;
  const bool x_park = !tl && idx->slab_rows < idx->nvec;
  const bool resumable = tl || cg0.use_fast || x_park;
  ISL_TRY(prepare_recompute(ws, nq, tl ? isl_launch::tl_state_words(cg0.ef, cg0.tl_wcap, cg0.fg.hbits)
                                       : cg0.use_fast ? isl_launch::fast_state_words(S0, cg0.fg.hbits) : 1));
  ISL_TRY(ensure_lane_stream(idx, ws));
  hipStream_t st = mode == StreamMode::USER ? user_stream : ws.stream;
  if (!idx->keep_rows) {  // every call starts from an empty cache: each node is encoded once per call
    ISL_HIP(hipMemsetAsync(idx->d_slot_of, 0xFF, (idx->nvec + 1) * 4, st));
    ISL_HIP(hipMemsetAsync(idx->d_owner, 0xFF, idx->slab_rows * 4, st));
    ISL_HIP(hipMemsetAsync(idx->d_stamp, 0, idx->slab_rows * 4, st));
    ISL_HIP(hipMemsetAsync(idx->d_slab_head, 0, 8, st));
  }
  ISL_HIP(hipMemsetAsync(ws.qflag, 0, nq * 4, st));
  ISL_HIP(hipMemsetAsync(ws.xslot, 0, nq * 4, st));
  if (!tl) {
    // no query is parked in the heap-exact kernel's pool yet (recompute calls run one at a time per index)
    if (!idx->pool.slots) {
      std::lock_guard<std::mutex> lock(idx->mu);
      ISL_TRY(ensure_pool(idx, ws));
    }
    ISL_HIP(hipMemsetAsync(idx->pool.locks, 0, (size_t)idx->pool.slots * 4, st));
  }
  uint64_t encoded = 0, rounds = 0;
  double kernel_ms = 0.0;
  // Queries in flight at a time: each may hold one hop (<= 128 rows) waiting for its last rows, and
  // those rows are exempt from eviction -- half the slab stays free for the rows being encoded, so
  // every round serves every miss and every query in flight advances by a hop per round.  (2^20
  // rows: 4096 queries; a smaller cache works through the batch a few queries at a time.)
  // (A slab with a row for every node never evicts: no limit.)
  // (a hop parked in the heap-exact kernel may hold a whole adjacency row of any length)
  const uint64_t hop_rows = 2 * std::max<uint64_t>(128, tl ? 128 : idx->max_degree);
  const uint32_t max_active = idx->slab_rows < idx->nvec
                                  ? (uint32_t)std::max<uint64_t>(1, idx->slab_rows / hop_rows) : (uint32_t)nq;
  uint32_t active = (uint32_t)std::min<uint64_t>(nq, max_active);
  uint32_t next_fresh = active;  // queries [next_fresh, nq) have not been started
  uint32_t nxl = 0;              // queries this round hands straight to the heap-exact kernel
  bool listed = active < nq;
  if (!resumable) {
    active = 0;  // an ordinary launch over all queries every round (round fields stay 0)
    listed = false;
  } else if (exact_only) {  // no traversal kernel in front: the round's queries are the heap-exact kernel's queue
    for (uint32_t i = 0; i < active; ++i) ws.h_xlist[i] = i;
    nxl = active;
    active = 0;
    listed = true;
  } else if (listed) {
    for (uint32_t i = 0; i < active; ++i) ws.h_qlist[i] = i;
    hipLaunchKernelGGL(copy_u32_kernel, dim3(16), dim3(256), 0, st, ws.h_qlist, ws.qlist, (uint64_t)active);
  }
  struct RoundReset {  // the lane goes back with its round fields cleared whatever happens below,
    const isl_index* idx;  // and no slot of the heap-exact kernel's pool stays with a query of this call
    isl::SearchWorkspace& w;
    hipStream_t st;
    ~RoundReset() {
      w.round_active = 0;
      w.round_prefetch = 0;
      w.round_x = 0;
      w.round_xpark = false;
      w.round_listed = false;
      if (idx->pool.slots && idx->pool.locks) {
        (void)hipMemsetAsync(idx->pool.locks, 0, (size_t)idx->pool.slots * 4, st);
        (void)hipStreamSynchronize(st);
      }
    }
  } reset{idx, ws, st};
  uint32_t* h_taken = ws.h_head + 15;  // (word 15 of the pinned ticket mirror is otherwise unused)
  // encoder batches in whole waves of GEMM tiles (assign_slots_kernel; ISL_RECOMPUTE_QUANTUM=0: every miss at once)
  uint32_t enc_quantum = 0, enc_chunk = 0;
  isl::encoder_batch_quantum(idx->enc, idx->tok_L, &enc_quantum, &enc_chunk);
  if (const char* qe = getenv("ISL_RECOMPUTE_QUANTUM")) enc_quantum = (uint32_t)std::max(0, atoi(qe));  // (read per call: A/B in one process)
  // Every query in flight advances by at least one hop per round, and a query makes at most a few
  // times ef hops with new rows: the cap scales with the number of groups the batch is worked
  // through in, so a 256-row cache (one query at a time) is not cut short and a bug still ends.
  const uint64_t max_rounds = 64 + ((nq + max_active - 1) / max_active) * ((uint64_t)64 * cg0.ef + 4096);
  uint32_t stalled = 0;
  // Two-level search: a parked query also names the nodes it expects to promote next (tl_prefetch of them), which
  // are encoded in the same round -- fewer, fuller rounds.  Only with a slab that has room to spare (the names are
  // guesses: under a small cache they would push out rows that hops are waiting for).  ISL_TL_PREFETCH=n overrides
  // (0 = off; read per call).
  uint32_t prefetch = (tl && idx->slab_rows >= (uint64_t)1024 * std::max<uint32_t>(1u, max_active)) ? kTlPrefetchDefault : 0u;
  if (const char* pe = getenv("ISL_TL_PREFETCH")) prefetch = tl ? (uint32_t)std::min(8, std::max(0, atoi(pe))) : 0u;
  std::vector<uint32_t> again;  // two-level search: queries to start over with a larger queue window
  for (;;) {
    ws.round_active = active;
    ws.round_x = nxl;
    ws.round_xpark = x_park;
    ws.round_listed = listed;
    ws.round_prefetch = prefetch;
    idx->round_no += 1;
    ISL_TRY(search_enqueue(idx, ws, d_queries, nq, d, k, ef, d_ids, d_dist, d_count, user_stream, mode, tl));
    uint32_t misses = 0;
    ISL_TRY(search_finish(idx, ws, &misses, nullptr, resumable));
    const uint32_t guesses = prefetch ? std::min<uint32_t>(ws.h_head[14], (uint32_t)ws.pref_cap) : 0u;
    kernel_ms += ws.stats.kernel_ms;
    rounds += 1;
    if (resumable) {  // next round: the queries that are waiting for rows, topped up with fresh ones
      uint32_t na = 0;
      nxl = 0;
      // queries parked in the heap-exact kernel go straight back to its queue (in front: they hold slots);
      // the others that wait for rows go through the traversal kernel again -- or, when there is none in
      // front (ef > 512, long rows), to that queue as well
      for (uint64_t i = 0; i < next_fresh; ++i)
        if (ws.h_status[i] == QS_BLOCKED_X) ws.h_xlist[nxl++] = (uint32_t)i;
      for (uint64_t i = 0; i < next_fresh; ++i)
        if (ws.h_status[i] == QS_BLOCKED) {
          if (exact_only) ws.h_xlist[nxl++] = (uint32_t)i;
          else ws.h_qlist[na++] = (uint32_t)i;
        }
      while (na + nxl < max_active && !again.empty()) { ws.h_qlist[na++] = again.back(); again.pop_back(); }
      while (na + nxl < max_active && next_fresh < nq) {
        if (exact_only) ws.h_xlist[nxl++] = next_fresh++;
        else ws.h_qlist[na++] = next_fresh++;
      }
      if (!na && tl) {
        // every query has run to its end; those whose queue window was too small start over -- alone,
        // nothing is parked now -- with a window four times the size (and a state block to match)
        uint32_t nshort = 0;
        ISL_TRY(tl_collect_short(ws, nq, &nshort));
        if (nshort && tcall.window_scale < 64) {
          tcall.window_scale *= 4;
          CallGeometry cg1;
          ISL_TRY(call_geometry(idx, d, k, ef, tl, cg1));
          ISL_TRY(ensure(ws, ws.qstate, ws.qstate_words,
                         std::max<uint64_t>(nq, 1) * isl_launch::tl_state_words(cg1.ef, cg1.tl_wcap, cg1.fg.hbits)));
          again.assign(ws.h_qlist, ws.h_qlist + nshort);
          while (na < max_active && !again.empty()) { ws.h_qlist[na++] = again.back(); again.pop_back(); }
        }
      }
      active = na;
      listed = true;
      if (!active && !nxl) {  // now the statuses are final
        ISL_TRY(search_statuses(ws, nq, nullptr));
        break;
      }
      if (na) hipLaunchKernelGGL(copy_u32_kernel, dim3(16), dim3(256), 0, st, ws.h_qlist, ws.qlist, (uint64_t)na);
    } else if (!misses) {
      break;
    }
    if (rounds > max_rounds)
      return isl::fail(ISL_ERR_SEARCH, "Search error: %llu recompute rounds without completing the batch (row cache "
                       "%llu rows, %u queries in flight at a time)", (unsigned long long)rounds,
                       (unsigned long long)idx->slab_rows, max_active);
    if (!misses) continue;  // only fresh queries to start
    if (misses > ws.miss_cap) misses = (uint32_t)ws.miss_cap;
    ISL_HIP(hipMemsetAsync(ws.uniq_count, 0, 4, st));
    hipLaunchKernelGGL(dedupe_misses_kernel, dim3((misses + 255) / 256), dim3(256), 0, st, ws.miss, misses,
                       idx->d_slot_of, ws.uniq, ws.uniq_count);
    // the guesses go behind the misses in the unique list: what a full slab or the quantum leaves out is theirs first
    if (guesses)
      hipLaunchKernelGGL(dedupe_misses_kernel, dim3((guesses + 255) / 256), dim3(256), 0, st, ws.miss + ws.miss_cap, guesses,
                         idx->d_slot_of, ws.uniq, ws.uniq_count);
    // slots for the new rows (clock hand over the slab; rows asked for in this round stay)
    hipLaunchKernelGGL(assign_slots_kernel, dim3(1), dim3(64), 0, st, ws.uniq, ws.uniq_count, idx->round_no,
                       (uint32_t)idx->slab_rows, idx->d_slab_head, idx->d_slot_of, idx->d_owner, idx->d_stamp,
                       ws.uslots, ws.ticket + 15, enc_quantum, enc_chunk);
    ISL_HIP(hipGetLastError());
    hipLaunchKernelGGL(copy_u32_kernel, dim3(1), dim3(64), 0, st, ws.ticket + 15, h_taken, (uint64_t)1);
    ISL_HIP(hipStreamSynchronize(st));
    const uint32_t take = *h_taken;
    // no row could be placed although rows are missing: every slot is held by a hop of this round.  A
    // resumable batch cannot get here (half the slab stays free by construction); a batch that re-runs
    // its blocked queries from their start needs their whole traversal resident and never will be.
    if (take == 0 && ++stalled >= (resumable ? 3u : 1u))
      return isl::fail(ISL_ERR_SEARCH, "Search error: the recompute provider's row cache (%llu rows) is too small "
                       "for this batch (no missing row could be placed)", (unsigned long long)idx->slab_rows);
    if (take) stalled = 0;
    ISL_TRY(isl::encoder_embed_nodes(idx->enc, idx->d_tokens, idx->d_lens, idx->tok_L, ws.uniq, take,
                                     idx->enc_normalize, idx->d_emb, idx->emb_stride, st, ws.uslots));
    const size_t lds = (size_t)TILE_ROWS * TILE_LD * 4 + 64;
    if (take)
      hipLaunchKernelGGL(row_norm2_list_kernel, dim3(std::min<uint32_t>((take + 63) / 64, 4096)), dim3(64), lds, st,
                         idx->d_emb, idx->emb_stride, (uint32_t)idx->emb_d, ws.uslots, take, idx->d_norm2);
    ISL_HIP(hipGetLastError());
    encoded += take;
  }
  ws.stats.encoded_nodes = encoded;
  ws.stats.recompute_rounds = rounds;
  ws.stats.kernel_ms = kernel_ms;
  ws.stats.allocations = ws.alloc_events - ws.alloc_mark;
  return ISL_OK;
}

// ---- concurrent asynchronous calls over the recompute provider, answered together ----
// Calls over the recompute provider run one at a time per index (the rounds rewrite the provider's row cache).
// A caller that keeps several batches in flight therefore used to get them answered one after the other, each
// with its own small encoder passes -- where ONE call over all their queries encodes a node once for all of
// them and hands the encoder fuller passes (8 x 1024 queries at 10M nodes: 89.0 against 74.9 queries/s,
// DESIGN.md section 3.4).  So the asynchronous device-buffer calls queue here: the call whose turn it is takes
// every compatible call (same d, k, ef, search kind, re-rank ratio) that is waiting at that moment, runs the
// rounds ONCE over the union of their queries on its own lane, and scatters the answers; the others wake up
// answered.  Every query's answer is what its own call would have computed (a query's traversal does not
// depend on what else is in the batch).  If the union fails -- one query's NodeNotFound fails the call it
// belongs to, not its neighbours' -- every member is run by itself and gets its own status.
struct RecCall {
  const float* dq;
  uint64_t nq, d, k, ef;
  uint64_t* ids;
  float* dist;
  uint32_t* cnt;
  bool has_tl;
  float ratio;
  isl::SearchWorkspace* ws;
  bool done = false;
  isl_status status = ISL_OK;
  isl::ErrorRecord err;
};

isl_status recompute_coalesced(const isl_index* idx, isl::SearchWorkspace& ws, const float* d_queries, uint64_t nq,
                               uint64_t d, uint64_t k, uint64_t ef, uint64_t* d_ids, float* d_dist, uint32_t* d_count,
                               const TwoLevelCall* tl) {
  static const bool off = getenv("ISL_NO_RECOMPUTE_COALESCE") != nullptr;  // A/B switch for measurements
  if (off) return search_sync(idx, ws, d_queries, nq, d, k, ef, d_ids, d_dist, d_count, nullptr, StreamMode::OWN, tl);
  RecCall me{d_queries, nq, d, k, ef, d_ids, d_dist, d_count, tl != nullptr, tl ? tl->ratio : 0.0f, &ws};
  auto& J = idx->rec_join;
  {
    std::lock_guard<std::mutex> l(J.mu);
    J.waiting.push_back(&me);
  }
  std::unique_lock<std::mutex> lead(J.leader);
  if (me.done) {  // answered by the call that had the turn before
    if (me.status != ISL_OK) isl::last_error() = me.err;
    return me.status;
  }
  constexpr uint64_t kMaxUnion = 1u << 17;  // queries one set of rounds works through
  std::vector<RecCall*> group{&me};
  uint64_t total = nq;
  {
    std::lock_guard<std::mutex> l(J.mu);
    std::vector<void*> rest;
    for (void* v : J.waiting) {
      RecCall* c = static_cast<RecCall*>(v);
      if (c == &me) continue;
      const bool same = c->d == d && c->k == k && c->ef == ef && c->has_tl == me.has_tl && (!me.has_tl || c->ratio == me.ratio);
      if (same && total + c->nq <= kMaxUnion) { group.push_back(c); total += c->nq; }
      else rest.push_back(v);
    }
    J.waiting.swap(rest);
  }
  if (group.size() == 1)
    return search_sync(idx, ws, d_queries, nq, d, k, ef, d_ids, d_dist, d_count, nullptr, StreamMode::OWN, tl);

  auto alone = [&](RecCall* c) {  // the member's own call, on the member's own lane
    c->status = search_sync(idx, *c->ws, c->dq, c->nq, c->d, c->k, c->ef, c->ids, c->dist, c->cnt, nullptr, StreamMode::OWN, tl);
    if (c->status != ISL_OK) c->err = isl::last_error();
  };
  auto fall_back = [&]() -> isl_status {  // every member by itself: its own answers, its own error
    for (RecCall* c : group)
      if (c != &me) { alone(c); c->done = true; }
    return search_sync(idx, ws, d_queries, nq, d, k, ef, d_ids, d_dist, d_count, nullptr, StreamMode::OWN, tl);
  };
  if (ensure(ws, ws.co_q, ws.co_q_cap, total * d) != ISL_OK || ensure(ws, ws.co_ids, ws.co_ids_cap, total * std::max<uint64_t>(k, 1)) != ISL_OK ||
      ensure(ws, ws.co_dist, ws.co_dist_cap, total * std::max<uint64_t>(k, 1)) != ISL_OK || ensure(ws, ws.co_cnt, ws.co_cnt_cap, total) != ISL_OK ||
      ensure_lane_stream(idx, ws) != ISL_OK)
    return fall_back();
  hipStream_t st = ws.stream;
  uint64_t o = 0;
  bool copied = true;
  for (RecCall* c : group) {  // a member's queries are there once its caller's stream has reached the call (ev_in)
    if (c != &me) copied = copied && hipStreamWaitEvent(st, c->ws->ev_in, 0) == hipSuccess;
    copied = copied && hipMemcpyAsync(ws.co_q + o * d, c->dq, c->nq * d * 4, hipMemcpyDeviceToDevice, st) == hipSuccess;
    o += c->nq;
  }
  if (!copied) { (void)hipStreamSynchronize(st); (void)hipGetLastError(); return fall_back(); }
  // (a host-buffer call's lane publishes its answers into pinned mirrors sized for THAT call: not for the union)
  const bool publishes = ws.publish_results;
  ws.publish_results = false;
  const isl_status rc = search_sync(idx, ws, ws.co_q, total, d, k, ef, ws.co_ids, ws.co_dist, ws.co_cnt, nullptr, StreamMode::OWN, tl);
  ws.publish_results = publishes;
  if (rc != ISL_OK) return fall_back();
  const isl_search_stats all = ws.stats;
  o = 0;
  bool scattered = true;
  for (RecCall* c : group) {
    if (k) {
      scattered = scattered && hipMemcpyAsync(c->ids, ws.co_ids + o * k, c->nq * k * 8, hipMemcpyDeviceToDevice, st) == hipSuccess;
      scattered = scattered && hipMemcpyAsync(c->dist, ws.co_dist + o * k, c->nq * k * 4, hipMemcpyDeviceToDevice, st) == hipSuccess;
    }
    scattered = scattered && hipMemcpyAsync(c->cnt, ws.co_cnt + o, c->nq * 4, hipMemcpyDeviceToDevice, st) == hipSuccess;
    if (c->ws->publish_results) {  // a host-buffer call (isl_search_batch_async): its wait copies out of the lane's pinned mirrors
      if (k) {
        scattered = scattered && hipMemcpyAsync(c->ws->h_ids, ws.co_ids + o * k, c->nq * k * 8, hipMemcpyDeviceToHost, st) == hipSuccess;
        scattered = scattered && hipMemcpyAsync(c->ws->h_dist, ws.co_dist + o * k, c->nq * k * 4, hipMemcpyDeviceToHost, st) == hipSuccess;
      }
      scattered = scattered && hipMemcpyAsync(c->ws->h_count, ws.co_cnt + o, c->nq * 4, hipMemcpyDeviceToHost, st) == hipSuccess;
      c->ws->nq_inflight = c->nq;
      c->ws->k_inflight = c->k;
    }
    o += c->nq;
  }
  scattered = scattered && hipStreamSynchronize(st) == hipSuccess;
  if (!scattered) { (void)hipGetLastError(); return fall_back(); }
  // each member's counters are its own queries' (the lane's pinned mirror holds the union's, query by query);
  // rounds, encoded nodes and kernel time are the union's
  o = 0;
  for (RecCall* c : group) {
    isl_search_stats ms = all;
    ms.queries = c->nq;
    ms.expansions = ms.edges = ms.evals = ms.pushes = 0;
    for (uint64_t i = o; i < o + c->nq; ++i) {
      ms.expansions += ws.h_ctr[i * 4 + 0];
      ms.edges += ws.h_ctr[i * 4 + 1];
      ms.evals += ws.h_ctr[i * 4 + 2];
      ms.pushes += ws.h_ctr[i * 4 + 3];
    }
    c->ws->stats = ms;
    o += c->nq;
    if (c != &me) { c->status = ISL_OK; c->done = true; }
  }
  return ISL_OK;
}

// Checks shared by the entry points; *done = 1 when the call is already answered.
isl_status precheck(const isl_index* idx, uint64_t nq, uint64_t d, uint64_t k, uint32_t* out_count,
                    bool count_on_device, int* done) {
  *done = 0;
  if (!idx) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "index is NULL");
  if (nq == 0) { *done = 1; return ISL_OK; }
  if (idx->num_nodes == 0) {  // is_empty() -> Ok(vec![]), leann.rs:875-877
    if (out_count) {
      if (count_on_device) {
        if (idx->device >= 0) {
          ISL_TRY(isl::use_device(idx->device));
          ISL_HIP(hipMemset(out_count, 0, nq * 4));
        }
      } else {
        memset(out_count, 0, nq * 4);
      }
    }
    *done = 1;
    return ISL_OK;
  }
  if (idx->has_dimension && d != idx->dimension)  // leann.rs:880-887
    return isl::fail_dim(idx->dimension, d);
  if (!idx->has_entry) return isl::fail(ISL_ERR_INDEX_NOT_BUILT, "Index not built");  // :889
  if (idx->device < 0 || !idx->d_off)
    return isl::fail(ISL_ERR_DEVICE, "index is not resident on a device (isl_index_upload)");
  if (!idx->d_emb && !idx->d_emb16)
    return isl::fail(ISL_ERR_EMBEDDING, "Embedding error: no embedding provider attached");
  if (d != idx->emb_d)  // metric.calculate length check, distance.rs:39-44
    return isl::fail_dim(d, idx->emb_d);
  if (k == 0) {
    *done = 2;  // nothing to write but counts
  }
  return ISL_OK;
}

isl_status precheck_two_level(const isl_index* idx, uint64_t d) {
  if (!idx->pq || !idx->d_codes)
    return isl::fail(ISL_ERR_PQ, "PQ error: no PQ codes attached (isl_index_set_pq_codes)");
  if (idx->is_hnsw) return isl::fail(ISL_ERR_UNSUPPORTED, "two-level search runs on a LeannIndex");
  if (d != idx->pq->dimension) return isl::fail_dim(idx->pq->dimension, d);  // pq.rs:308-313
  return ISL_OK;
}

__global__ void check_codes_kernel(const uint16_t* __restrict__ codes, uint64_t n, uint32_t K,
                                   uint32_t* __restrict__ flag) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  bool bad = false;
  for (; i < n; i += stride) bad |= codes[i] >= K;
  if (bad) atomicOr(flag, 1u);
}

// ---- lanes: claimed under idx->mu, then owned by the caller until released ----
isl::SearchWorkspace* claim_lane(const isl_index* idx) {
  std::lock_guard<std::mutex> lock(idx->mu);
  for (auto& w : idx->ws)
    if (!w.busy) {
      w.busy = true;
      w.waiting = false;
      w.threaded = false;
      w.enqueued = false;
      w.token = 0;
      w.alloc_mark = w.alloc_events;
      w.u_ids = nullptr; w.u_dist = nullptr; w.u_count = nullptr;
      w.publish_results = false;
      return &w;
    }
  return nullptr;
}
void release_lane(const isl_index* idx, isl::SearchWorkspace& ws) {
  std::lock_guard<std::mutex> lock(idx->mu);
  ws.busy = false;
  ws.waiting = false;
  ws.token = 0;
}
isl_status no_lane() {
  return isl::fail(ISL_ERR_SEARCH, "Search error: %d searches already in flight; isl_search_wait one first",
                   isl::kSearchLanes);
}
// RAII: the lane goes back unless a token took it over
struct LaneGuard {
  const isl_index* idx;
  isl::SearchWorkspace* ws;
  ~LaneGuard() { if (ws) release_lane(idx, *ws); }
  void keep() { ws = nullptr; }
};

// host-pointer calls: the queries go through the lane's pinned buffer, from where a kernel pulls
// them into HBM; the answers come back with the publish kernel (ws.publish_results)
isl_status host_stage_in(const isl_index* idx, isl::SearchWorkspace& ws, const float* queries, uint64_t nq, uint64_t d, uint64_t k) {
  ISL_TRY(prepare_host_staging(ws, nq, d, k));
  ISL_TRY(ensure_lane_stream(idx, ws));
  memcpy(ws.h_q, queries, nq * d * 4);
  const uint64_t bytes = nq * d * 4;
  if (bytes % 16 == 0)
    hipLaunchKernelGGL(copy_u128_kernel, dim3(256), dim3(256), 0, ws.stream, (const uint4*)ws.h_q, (uint4*)ws.q_stage,
                       bytes / 16);
  else
    hipLaunchKernelGGL(copy_u32_kernel, dim3(256), dim3(256), 0, ws.stream, (const uint32_t*)ws.h_q,
                       (uint32_t*)ws.q_stage, bytes / 4);
  ISL_HIP(hipGetLastError());
  ws.publish_results = true;
  return ISL_OK;
}
void host_copy_out(const isl::SearchWorkspace& ws, uint64_t nq, uint64_t k, uint64_t* out_ids, float* out_dist,
                   uint32_t* out_count) {
  if (k) {
    memcpy(out_ids, ws.h_ids, nq * k * 8);
    memcpy(out_dist, ws.h_dist, nq * k * 4);
  }
  memcpy(out_count, ws.h_count, nq * 4);
}

// Runs `body` (the synchronous form of a call, on the claimed lane `ws`) on a host thread; the
// token's wait joins it.  The thread selects the index's device first (the HIP device is per thread).
// Thread creation can fail (std::system_error, bad_alloc): nothing may cross the C ABI, so that is an
// ISL_ERR_DEVICE of the call and the lane goes back through the caller's LaneGuard; `threaded` is set
// only once the thread object exists.
template <typename F>
isl_status start_worker(const isl_index* idx, isl::SearchWorkspace* ws, F body) {
  ws->worker_status = ISL_OK;
  ws->threaded = false;
  std::thread* th = nullptr;
  try {
    th = new std::thread([idx, ws, body]() {
      isl_status st = isl::use_device(idx->device);
      try {  // nothing may leave the thread: an exception here would end the process
        if (st == ISL_OK) st = body();
      } catch (const std::exception& e) {
        st = isl::fail(ISL_ERR_SEARCH, "Search error: %s", e.what());
      } catch (...) {
        st = isl::fail(ISL_ERR_SEARCH, "Search error: unknown exception in the call's worker thread");
      }
      ws->worker_status = st;
      ws->worker_error = isl::last_error();
    });
  } catch (const std::exception& e) {
    return isl::fail(ISL_ERR_DEVICE, "the call's worker thread could not be started: %s", e.what());
  } catch (...) {
    return isl::fail(ISL_ERR_DEVICE, "the call's worker thread could not be started");
  }
  ws->worker = th;
  ws->threaded = true;
  return ISL_OK;
}
// true when the lane's call ran on a worker: *st = its status, the caller's error record = the worker's
bool join_worker(isl::SearchWorkspace& ws, isl_status* st) {
  if (!ws.threaded) return false;
  if (ws.worker) {
    ws.worker->join();
    delete ws.worker;
    ws.worker = nullptr;
  }
  ws.threaded = false;
  *st = ws.worker_status;
  if (*st != ISL_OK) isl::last_error() = ws.worker_error;
  return true;
}

}  // namespace

namespace isl {

void join_lane_workers(const isl_index* idx) {
  for (auto& w : idx->ws)
    if (w.worker) {
      w.worker->join();
      delete w.worker;
      w.worker = nullptr;
    }
}

bool any_lane_busy(const isl_index* idx) {
  for (const auto& w : idx->ws)
    if (w.busy) return true;
  return false;
}

void free_exact_pool(ExactPool& pl) {
  void* ptrs[] = {pl.cand_d, pl.cand_id, pl.vis_bits, pl.ulist, pl.locks, pl.xstate};
  for (void* q : ptrs)
    if (q) (void)hipFree(q);
  pl = ExactPool{};
}

// padded copy of the adjacency (64 ids per node + a degree array): 260 bytes per node buy the
// traversal one dependent memory round trip per hop.  Built where the CSR becomes resident
// (isl_index_upload, isl_index_from_device_csr, isl_index_prepare); the handle's fields are set
// only once the copy is complete.  Not enough memory -> the searches stay on the CSR.
isl_status ensure_padded_adjacency(isl_index* idx) {
  if (idx->d_ell || !idx->d_off || !idx->num_nodes || idx->max_degree > 128) return ISL_OK;
  const uint64_t n = idx->num_nodes;
  const uint32_t W = idx->max_degree > 64 ? 128u : 64u;
  uint32_t* ell = nullptr;
  uint32_t* deg = nullptr;
  if (hipMalloc(&ell, n * W * 4) != hipSuccess || hipMalloc(&deg, n * 4) != hipSuccess) {
    (void)hipGetLastError();
    if (ell) (void)hipFree(ell);
    return ISL_OK;
  }
  hipLaunchKernelGGL(pad_rows_kernel, dim3((uint32_t)((n + 3) / 4)), dim3(256), 0, nullptr, idx->d_off, idx->d_adj, n,
                     W, ell, deg);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    (void)hipFree(ell);
    (void)hipFree(deg);
    return fail(ISL_ERR_DEVICE, "padded adjacency: %s", hipGetErrorString(e));
  }
  idx->d_ell = ell;
  idx->d_ell_deg = deg;
  idx->ell_w = W;
  idx->ell_owned = true;
  return ISL_OK;
}

// multi-GPU exchange (shard.hip): the queries of call `token` that did not end in QS_OK get
// ISL_SHARD_POISON_COUNT as their count in the rank's record, so that every rank's merge sees which
// answers are incomplete.  Enqueued on `stream`, which must already wait for the call's kernels
// (isl_search_stream_wait).  Calls that run on a worker thread are finished by then (the stream wait
// joined them), their status array is final as well.
__global__ void poison_failed_kernel(const uint32_t* __restrict__ status, uint32_t* __restrict__ counts, uint32_t nq) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nq && status[i] != QS_OK) counts[i] = ISL_SHARD_POISON_COUNT;
}
isl_status poison_failed_queries(const isl_index* idx, uint64_t token, uint32_t* d_counts, uint64_t nq, hipStream_t stream) {
  const uint32_t* status = nullptr;
  {
    std::lock_guard<std::mutex> lock(idx->mu);
    for (auto& w : idx->ws)
      if (w.busy && w.token == token) { status = w.status; break; }
  }
  if (!status) return fail(ISL_ERR_INVALID_ARGUMENT, "unknown or already completed search token");
  hipLaunchKernelGGL(poison_failed_kernel, dim3((uint32_t)((nq + 255) / 256)), dim3(256), 0, stream, status, d_counts,
                     (uint32_t)nq);
  ISL_HIP(hipGetLastError());
  return ISL_OK;
}

isl_status search_device_sync(const isl_index* idx, const float* d_queries, uint64_t nq, uint64_t d,
                              uint64_t k, uint64_t ef, uint64_t* d_ids, float* d_dist,
                              uint32_t* d_count, hipStream_t stream) {
  SearchWorkspace* ws = claim_lane(idx);
  if (!ws) return no_lane();
  LaneGuard guard{idx, ws};
  return search_sync(idx, *ws, d_queries, nq, d, k, ef, d_ids, d_dist, d_count, stream, StreamMode::USER);
}

}  // namespace isl

extern "C" {

isl_status isl_index_prepare(isl_index* idx, uint64_t max_nq, uint64_t max_ef, uint64_t max_k, int32_t lanes) {
  if (!idx) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "index is NULL");
  if (idx->device < 0 || !idx->d_off)
    return isl::fail(ISL_ERR_DEVICE, "index is not resident on a device (isl_index_upload)");
  if (!idx->d_emb && !idx->d_emb16)
    return isl::fail(ISL_ERR_EMBEDDING, "Embedding error: no embedding provider attached");
  if (lanes < 1 || lanes > isl::kSearchLanes)
    return isl::fail(ISL_ERR_INVALID_ARGUMENT, "lanes must be 1..%d", isl::kSearchLanes);
  if (max_nq == 0 || max_nq > 0x7FFFFFFFull) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "max_nq out of range");
  max_ef = std::max<uint64_t>(std::max(max_ef, max_k), 1);
  if (max_ef > kMaxExactEf)
    return isl::fail(ISL_ERR_UNSUPPORTED, "ef = %llu exceeds the device limit %u", (unsigned long long)max_ef,
                     kMaxExactEf);
  ISL_TRY(isl::use_device(idx->device));
  const uint64_t d = idx->emb_d;
  {
    std::lock_guard<std::mutex> lock(idx->mu);
    if (isl::any_lane_busy(idx))
      return isl::fail(ISL_ERR_SEARCH, "Search error: isl_index_prepare while searches are in flight");
    ISL_TRY(isl::ensure_padded_adjacency(idx));
    ISL_TRY(ensure_pool(idx, idx->ws[0]));
    // the lanes are held for the duration: the last step exercises them together
    for (int i = 0; i < lanes; ++i) idx->ws[i].busy = true;
  }
  const int ncu = isl::device_cu_count(idx->device);
  // a smaller ef puts more waves on a CU: size the per-wave overflow tables for the most there can be
  const uint32_t ovf_slots = (uint32_t)std::min<uint64_t>(max_nq, (uint64_t)ncu * waves_per_cu_cap());
  struct ReleaseAll {
    isl_index* idx; int lanes;
    ~ReleaseAll() { for (int i = 0; i < lanes; ++i) release_lane(idx, idx->ws[i]); }
  } release_all{idx, lanes};
  for (int i = 0; i < lanes; ++i) {
    isl::SearchWorkspace& ws = idx->ws[i];
    ISL_TRY(prepare_workspace(idx, ws, (uint32_t)max_nq, idx->recompute ? (uint32_t)max_nq : ovf_slots,
                              push_log_cap((uint32_t)max_ef)));
    ISL_TRY(prepare_host_staging(ws, max_nq, d, std::max<uint64_t>(max_k, 1)));
    if (idx->recompute) {
      const uint32_t efm = (uint32_t)max_ef;
      ISL_TRY(prepare_recompute(ws, max_nq, isl_launch::fast_state_words(efm <= 64 ? 1 : efm <= 128 ? 2 : efm <= 256 ? 4 : 8,
                                                             fast_geometry(efm, (uint32_t)d).hbits)));
    }
    if (idx->is_hnsw) ISL_TRY(ensure(ws, ws.q_entry, ws.q_entry_cap, max_nq * 2));
    if (idx->pq && idx->d_codes && !idx->is_hnsw && d == idx->pq->dimension)
      ISL_TRY(ensure(ws, ws.tl_tables, ws.tl_tables_cap, max_nq * idx->pq->m * idx->pq->K));
    memset(ws.h_q, 0, max_nq * d * 4);
  }
  // The kernels this index will launch, over zero queries, and one staged copy each way, on all
  // lanes AT ONCE and twice over: code objects loaded, every stream's hardware queue and copy
  // queue up, the LDS opt-in attribute set, and the runtime's pools of completion signals grown to
  // what `lanes` calls in flight need (the runtime grows them one concurrent copy at a time, at
  // several milliseconds each -- measured inside the first calls otherwise).
  for (int round = 0; round < 2; ++round) {
    for (int i = 0; i < lanes; ++i) {
      isl::SearchWorkspace& ws = idx->ws[i];
      hipLaunchKernelGGL(copy_u128_kernel, dim3(256), dim3(256), 0, ws.stream, (const uint4*)ws.h_q,
                         (uint4*)ws.q_stage, max_nq * d * 4 / 16);
      const uint64_t efs[] = {max_ef, std::min<uint64_t>(max_ef, 64)};
      for (uint64_t e : efs)
        ISL_TRY(search_enqueue(idx, ws, nullptr, 0, d, std::min<uint64_t>(max_k, e), e, nullptr, nullptr, nullptr,
                               nullptr, StreamMode::OWN, nullptr, true));
      if (idx->pq && idx->d_codes && !idx->is_hnsw && d == idx->pq->dimension) {
        const TwoLevelCall tl{0.5f};
        ISL_TRY(search_enqueue(idx, ws, nullptr, 0, d, std::min<uint64_t>(max_k, max_ef), max_ef, nullptr, nullptr,
                               nullptr, nullptr, StreamMode::OWN, &tl, true));
      }
      ws.publish_results = true;
      ISL_TRY(publish(ws, max_nq, std::max<uint64_t>(max_k, 1), ws.stream));
    }
    for (int i = 0; i < lanes; ++i) ISL_HIP(hipStreamSynchronize(idx->ws[i].stream));
  }
  for (int i = 0; i < lanes; ++i) idx->ws[i].alloc_mark = idx->ws[i].alloc_events;
  return ISL_OK;
}

isl_status isl_search_batch_device(const isl_index* idx, const float* d_queries, uint64_t nq,
                                   uint64_t d, uint64_t k, uint64_t ef, uint64_t* d_out_ids,
                                   float* d_out_dist, uint32_t* d_out_count, void* stream) {
  int done = 0;
  ISL_TRY(precheck(idx, nq, d, k, d_out_count, true, &done));
  if (done == 1) return ISL_OK;
  if (!d_queries || !d_out_count || (k && (!d_out_ids || !d_out_dist)))
    return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL buffer");
  ISL_TRY(isl::use_device(idx->device));
  isl::SearchWorkspace* ws = claim_lane(idx);
  if (!ws) return no_lane();
  LaneGuard guard{idx, ws};
  const isl_status st = search_sync(idx, *ws, d_queries, nq, d, k, ef, d_out_ids, d_out_dist, d_out_count,
                                    (hipStream_t)stream, StreamMode::USER);
  note_last_stats(idx, ws->stats);
  return st;
}

isl_status isl_search_batch_device_async(const isl_index* idx, const float* d_queries, uint64_t nq,
                                         uint64_t d, uint64_t k, uint64_t ef, uint64_t* d_out_ids,
                                         float* d_out_dist, uint32_t* d_out_count, void* stream,
                                         uint64_t* token) {
  if (!token) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "token is NULL");
  *token = 0;
  int done = 0;
  ISL_TRY(precheck(idx, nq, d, k, d_out_count, true, &done));
  if (done == 1) return ISL_OK;  // token 0: nothing to wait for
  if (!d_queries || !d_out_count || (k && (!d_out_ids || !d_out_dist)))
    return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL buffer");
  ISL_TRY(isl::use_device(idx->device));
  isl::SearchWorkspace* ws = claim_lane(idx);
  if (!ws) return no_lane();
  LaneGuard guard{idx, ws};
  if (idx->recompute) {
    // the provider works through the batch in rounds (search, encode what was missed, resume): the
    // synchronous form on a thread of its own, ordered after the caller's stream
    ISL_TRY(ensure_lane_stream(idx, *ws));
    ISL_HIP(hipEventRecord(ws->ev_in, (hipStream_t)stream));
    ISL_HIP(hipStreamWaitEvent(ws->stream, ws->ev_in, 0));
    ISL_TRY(start_worker(idx, ws, [=]() {
      return recompute_coalesced(idx, *ws, d_queries, nq, d, k, ef, d_out_ids, d_out_dist, d_out_count, nullptr);
    }));
  } else
  ISL_TRY(search_enqueue(idx, *ws, d_queries, nq, d, k, ef, d_out_ids, d_out_dist, d_out_count,
                         (hipStream_t)stream, StreamMode::OWN_AFTER_USER));
  {
    std::lock_guard<std::mutex> lock(idx->mu);
    ws->token = idx->next_token++;
    *token = ws->token;
  }
  guard.keep();
  return ISL_OK;
}

isl_status isl_search_batch_async(const isl_index* idx, const float* queries, uint64_t nq, uint64_t d,
                                  uint64_t k, uint64_t ef, uint64_t* out_ids, float* out_dist,
                                  uint32_t* out_count, uint64_t* token) {
  if (!token) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "token is NULL");
  *token = 0;
  int done = 0;
  ISL_TRY(precheck(idx, nq, d, k, out_count, false, &done));
  if (done == 1) return ISL_OK;
  if (!queries || !out_count || (k && (!out_ids || !out_dist)))
    return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL buffer");
  ISL_TRY(isl::use_device(idx->device));
  isl::SearchWorkspace* ws = claim_lane(idx);
  if (!ws) return no_lane();
  LaneGuard guard{idx, ws};
  ISL_TRY(host_stage_in(idx, *ws, queries, nq, d, k));
  if (idx->recompute) {
    ISL_HIP(hipEventRecord(ws->ev_in, ws->stream));  // the staged queries are there (a call that answers this one with its own waits for it)
    ISL_TRY(start_worker(idx, ws, [=]() {
      return recompute_coalesced(idx, *ws, ws->q_stage, nq, d, k, ef, ws->ids_stage, ws->dist_stage, ws->count_stage, nullptr);
    }));
  } else
  ISL_TRY(search_enqueue(idx, *ws, ws->q_stage, nq, d, k, ef, ws->ids_stage, ws->dist_stage, ws->count_stage,
                         nullptr, StreamMode::OWN));
  ws->u_ids = out_ids;
  ws->u_dist = out_dist;
  ws->u_count = out_count;
  {
    std::lock_guard<std::mutex> lock(idx->mu);
    ws->token = idx->next_token++;
    *token = ws->token;
  }
  guard.keep();
  return ISL_OK;
}

isl_status isl_search_wait_stats(const isl_index* idx, uint64_t token, isl_search_stats* stats) {
  if (!idx) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "index is NULL");
  if (token == 0) {
    if (stats) *stats = isl_search_stats{};
    return ISL_OK;
  }
  ISL_TRY(isl::use_device(idx->device));
  isl::SearchWorkspace* ws = nullptr;
  {
    std::lock_guard<std::mutex> lock(idx->mu);
    for (auto& w : idx->ws)
      if (w.busy && w.token == token && !w.waiting) { ws = &w; w.waiting = true; break; }
  }
  if (!ws) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "unknown or already completed search token");
  LaneGuard guard{idx, ws};
  isl_status st = ISL_OK;
  if (!join_worker(*ws, &st)) st = search_finish(idx, *ws);  // the D2H result copies sit on the same stream
  if (st == ISL_OK && ws->u_count) host_copy_out(*ws, ws->nq_inflight, ws->k_inflight, ws->u_ids, ws->u_dist, ws->u_count);
  if (stats) *stats = ws->stats;
  note_last_stats(idx, ws->stats);
  return st;
}

isl_status isl_search_wait(const isl_index* idx, uint64_t token) { return isl_search_wait_stats(idx, token, nullptr); }

isl_status isl_search_stream_wait(const isl_index* idx, uint64_t token, void* stream) {
  if (!idx) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "index is NULL");
  if (token == 0) return ISL_OK;
  ISL_TRY(isl::use_device(idx->device));
  isl::SearchWorkspace* found = nullptr;
  {
    std::lock_guard<std::mutex> lock(idx->mu);
    for (auto& w : idx->ws)
      if (w.busy && w.token == token) {
        if (!w.threaded) {
          ISL_HIP(hipStreamWaitEvent((hipStream_t)stream, w.ev1, 0));  // recorded behind the last search kernel
          return ISL_OK;
        }
        if (!w.worker) return ISL_OK;  // already joined
        if (w.waiting) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "the call is being waited for on another thread");
        found = &w;
        w.waiting = true;  // (keeps isl_search_wait of another thread off the lane while we join)
        break;
      }
  }
  if (!found) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "unknown or already completed search token");
  // a call that runs on a worker thread (recompute provider, two-level search): its kernels are
  // enqueued round by round, so the host waits for the worker here; the call's status stays with the
  // lane for isl_search_wait
  found->worker->join();
  delete found->worker;
  found->worker = nullptr;
  {
    std::lock_guard<std::mutex> lock(idx->mu);
    found->waiting = false;
  }
  return ISL_OK;
}

// host-pointer entry: stage the queries, search on the lane's stream, copy the answers back
static isl_status search_batch_host(const isl_index* idx, const float* queries, uint64_t nq, uint64_t d,
                                    uint64_t k, uint64_t ef, uint64_t* out_ids, float* out_dist,
                                    uint32_t* out_count, const TwoLevelCall* tl) {
  int done = 0;
  ISL_TRY(precheck(idx, nq, d, k, out_count, false, &done));
  if (done == 1) return ISL_OK;
  if (tl) ISL_TRY(precheck_two_level(idx, d));
  if (!queries || !out_count || (k && (!out_ids || !out_dist)))
    return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL buffer");
  ISL_TRY(isl::use_device(idx->device));
  isl::SearchWorkspace* wsp = claim_lane(idx);
  if (!wsp) return no_lane();
  LaneGuard guard{idx, wsp};
  isl::SearchWorkspace& ws = *wsp;
  ISL_TRY(host_stage_in(idx, ws, queries, nq, d, k));
  const isl_status st = search_sync(idx, ws, ws.q_stage, nq, d, k, ef, ws.ids_stage, ws.dist_stage, ws.count_stage,
                                    nullptr, StreamMode::OWN, tl);
  note_last_stats(idx, ws.stats);
  ISL_TRY(st);
  host_copy_out(ws, nq, k, out_ids, out_dist, out_count);  // published with the last round's kernels
  return ISL_OK;
}

isl_status isl_search_batch(const isl_index* idx, const float* queries, uint64_t nq, uint64_t d,
                            uint64_t k, uint64_t ef, uint64_t* out_ids, float* out_dist,
                            uint32_t* out_count) {
  return search_batch_host(idx, queries, nq, d, k, ef, out_ids, out_dist, out_count, nullptr);
}

// ---- two-level search with a PQ filter (extension, see leann_search_two_level) ----
isl_status isl_index_set_pq_codes(isl_index* idx, const isl_pq* pq, const uint16_t* codes, uint64_t n,
                                  int32_t mem) {
  if (!idx || !pq || (!codes && n)) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL argument");
  if (n == 0) return isl::fail(ISL_ERR_EMPTY_COLLECTION, "Empty vector collection");
  if (idx->device < 0) return isl::fail(ISL_ERR_DEVICE, "call isl_index_upload before attaching PQ codes");
  if (pq->device != idx->device)
    return isl::fail(ISL_ERR_INVALID_ARGUMENT, "quantizer and index live on different devices");
  if (n > 0x7FFFFFFFull) return isl::fail(ISL_ERR_UNSUPPORTED, "more than 2^31 - 1 code rows");
  ISL_TRY(isl::use_device(idx->device));
  std::lock_guard<std::mutex> lock(idx->mu);
  if (isl::any_lane_busy(idx))
    return isl::fail(ISL_ERR_SEARCH, "Search error: PQ codes cannot be swapped while searches are in flight");
  if (idx->d_codes) { (void)hipFree(idx->d_codes); idx->d_codes = nullptr; }
  idx->pq = nullptr;
  idx->ncodes = 0;
  const size_t bytes = (size_t)n * pq->m * 2;
  ISL_HIP(hipMalloc(&idx->d_codes, bytes));
  ISL_HIP(hipMemcpy(idx->d_codes, codes, bytes, mem == ISL_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice));
  // tables[sq][code] of table_distance (pq.rs:345) would index out of bounds (panic) for a code >= K
  uint32_t* d_flag = nullptr;
  ISL_HIP(hipMalloc(&d_flag, 4));
  ISL_HIP(hipMemset(d_flag, 0, 4));
  hipLaunchKernelGGL(check_codes_kernel, dim3(1024), dim3(256), 0, nullptr, idx->d_codes, (uint64_t)n * pq->m,
                     (uint32_t)pq->K, d_flag);
  uint32_t flag = 0;
  hipError_t e = hipMemcpy(&flag, d_flag, 4, hipMemcpyDeviceToHost);
  (void)hipFree(d_flag);
  if (e != hipSuccess || flag) {
    (void)hipFree(idx->d_codes);
    idx->d_codes = nullptr;
    if (e != hipSuccess) return isl::fail(ISL_ERR_DEVICE, "code check failed: %s", hipGetErrorString(e));
    return isl::fail(ISL_ERR_PQ, "PQ error: a code is not below num_centroids = %llu", (unsigned long long)pq->K);
  }
  idx->pq = pq;
  idx->ncodes = n;
  return ISL_OK;
}

isl_status isl_search_two_level_batch(const isl_index* idx, const float* queries, uint64_t nq, uint64_t d,
                                      uint64_t k, uint64_t ef, float rerank_ratio, uint64_t* out_ids,
                                      float* out_dist, uint32_t* out_count) {
  const TwoLevelCall tl{rerank_ratio};
  return search_batch_host(idx, queries, nq, d, k, ef, out_ids, out_dist, out_count, &tl);
}

isl_status isl_search_two_level_batch_device(const isl_index* idx, const float* d_queries, uint64_t nq,
                                             uint64_t d, uint64_t k, uint64_t ef, float rerank_ratio,
                                             uint64_t* d_out_ids, float* d_out_dist, uint32_t* d_out_count,
                                             void* stream) {
  int done = 0;
  ISL_TRY(precheck(idx, nq, d, k, d_out_count, true, &done));
  if (done == 1) return ISL_OK;
  ISL_TRY(precheck_two_level(idx, d));
  if (!d_queries || !d_out_count || (k && (!d_out_ids || !d_out_dist)))
    return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL buffer");
  ISL_TRY(isl::use_device(idx->device));
  isl::SearchWorkspace* ws = claim_lane(idx);
  if (!ws) return no_lane();
  LaneGuard guard{idx, ws};
  const TwoLevelCall tl{rerank_ratio};
  const isl_status st = search_sync(idx, *ws, d_queries, nq, d, k, ef, d_out_ids, d_out_dist, d_out_count,
                                    (hipStream_t)stream, StreamMode::USER, &tl);
  note_last_stats(idx, ws->stats);
  return st;
}

isl_status isl_search_two_level_batch_device_async(const isl_index* idx, const float* d_queries, uint64_t nq,
                                                   uint64_t d, uint64_t k, uint64_t ef, float rerank_ratio,
                                                   uint64_t* d_out_ids, float* d_out_dist, uint32_t* d_out_count,
                                                   void* stream, uint64_t* token) {
  if (!token) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "token is NULL");
  *token = 0;
  int done = 0;
  ISL_TRY(precheck(idx, nq, d, k, d_out_count, true, &done));
  if (done == 1) return ISL_OK;
  ISL_TRY(precheck_two_level(idx, d));
  if (!d_queries || !d_out_count || (k && (!d_out_ids || !d_out_dist)))
    return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL buffer");
  ISL_TRY(isl::use_device(idx->device));
  isl::SearchWorkspace* ws = claim_lane(idx);
  if (!ws) return no_lane();
  LaneGuard guard{idx, ws};
  ISL_TRY(ensure_lane_stream(idx, *ws));
  ISL_HIP(hipEventRecord(ws->ev_in, (hipStream_t)stream));
  ISL_HIP(hipStreamWaitEvent(ws->stream, ws->ev_in, 0));
  const TwoLevelCall tl{rerank_ratio};
  ISL_TRY(start_worker(idx, ws, [=]() {
    if (idx->recompute)
      return recompute_coalesced(idx, *ws, d_queries, nq, d, k, ef, d_out_ids, d_out_dist, d_out_count, &tl);
    return search_sync(idx, *ws, d_queries, nq, d, k, ef, d_out_ids, d_out_dist, d_out_count, nullptr, StreamMode::OWN,
                       &tl);
  }));
  {
    std::lock_guard<std::mutex> lock(idx->mu);
    ws->token = idx->next_token++;
    *token = ws->token;
  }
  guard.keep();
  return ISL_OK;
}

isl_status isl_search(const isl_index* idx, const float* query, uint64_t d, uint64_t k,
                      uint64_t* out_ids, float* out_dist, uint32_t* out_count) {
  if (!idx) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "index is NULL");
  return isl_search_batch(idx, query, 1, d, k, idx->cfg.ef_search, out_ids, out_dist,
                          out_count);  // leann.rs:858-865
}

isl_status isl_search_last_stats(const isl_index* idx, isl_search_stats* out) {
  if (!idx || !out) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL argument");
  *out = tl_last_stats.idx == idx ? tl_last_stats.st : isl_search_stats{};
  return ISL_OK;
}

}  // extern "C"
