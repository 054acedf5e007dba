// The search kernels of libislands_amd.so (one 64-lane wavefront per query) -- included by the
// translation units that instantiate them: search_fast{1,2,4,8}.hip (the fast kernel per result-set
// size), search_aux.hip (heap-exact kernel, two-level search, HnswGraph descent) and search.hip
// (host side).  Everything lives in an anonymous namespace: each unit emits the instantiations
// it launches and nothing else.
#pragma once
#include "device_common.hip.h"
#include "encoder.hpp"

#include <algorithm>
#include <type_traits>

namespace {

using namespace isl_dev;

constexpr uint32_t EMPTY = 0xFFFFFFFFu;
constexpr uint32_t FLAG_EXP = 0x80000000u;
constexpr uint32_t ID_MASK = 0x7FFFFFFFu;

// per-query status words written by the kernels
enum : uint32_t {
  QS_OK = 0,
  QS_NODE_NOT_FOUND = 5,
  QS_REDO = 0x100,     // fast kernel gave up -> exact kernel
  QS_SCRATCH = 0x101,  // exact kernel ran out of candidate scratch
  QS_REPLAY = 0x102,   // result-heap order needed: replay kernel re-orders from the push log
  QS_BLOCKED = 0x103,  // recompute provider: a needed row is not materialised yet (ids reported)
  QS_BLOCKED_X = 0x104 // ... and the query is parked in the heap-exact kernel (it keeps its pool slot)
};

struct SearchParams {
  const uint64_t* off;
  const uint32_t* adj;
  uint64_t num_nodes;
  const void* emb;     // rows: f32, or bf16 bits when emb_bf16
  uint32_t emb_bf16;
  const float* norm2;  // per-row sum of squares (cosine only), reference order
  uint64_t nvec;
  uint64_t stride;  // floats between rows
  uint32_t d;
  const float* queries;
  uint32_t nq;
  uint32_t k, ef;
  float prune_ratio;
  uint32_t prune_strategy;
  uint32_t entry;
  uint64_t* out_ids;
  float* out_dist;
  uint32_t* out_count;
  uint32_t* status;
  uint64_t* payload;
  uint32_t* ctr;     // [nq][4]
  uint32_t* ticket;  // [0] fast head, [1] redo count, [2] exact head, [3] replay count,
                     // [4] replay head, [8..11] why the fast kernel gave a query up
  uint32_t* redo;    // [nq] queries for the exact kernel
  uint32_t* replay;  // [nq] queries for the replay kernel
  uint64_t* prof;    // optional [nq][8] phase timers (100 MHz ticks), ISL_DEBUG only
  uint64_t* tline;   // optional [nq][2]: start tick of every query (100 MHz); its duration | hops << 40; ISL_TIMELINE only
  uint2* plog;       // [nq][plog_cap] (distance bits, id) of every results.push, in order
  uint32_t plog_cap;
  uint32_t hbits;    // LDS visited table: 1 << hbits entries where its size is part of a parked query's state block
  uint32_t hcap;     // ... its entries (a multiple of 64; 1 << hbits unless sized by the index's evaluations per query)
  uint32_t* otab;    // overflow visited table in HBM, per slot
  uint32_t obits;
  // exact-kernel scratch
  float* cand_d;
  uint32_t* cand_id;
  uint64_t cand_cap;
  uint32_t* vis_bits;
  uint64_t vis_words;
  uint32_t* ulist;
  uint32_t ulist_cap;
  uint32_t* pool_locks;  // [pool_slots] lock word per slot of the shared exact-kernel scratch pool
  uint32_t pool_slots;
  // recompute provider: queries parked in the heap-exact kernel.  xslot[q] = 1 + the pool slot query q
  // keeps across rounds (0 = none); xstate = [pool_slots][xstate_words]: 16 scalars, then the result heap
  uint32_t* xslot;
  uint32_t* xstate;
  uint32_t xstate_words;
  // graph under construction (build.hip): row i = adj[i * ell_w .. + ell_deg[i]), `off` unused
  uint32_t ell_w;
  const uint32_t* ell_deg;
  // recompute provider: the rows live in a bounded slab, row of node id = slot_of[id] (kNoSlot =
  // not materialised); a query that needs an absent row appends the id to `miss` (count in
  // ticket[13]) and stops with QS_BLOCKED.  NULL = rows addressed by node id (in-memory provider).
  const uint32_t* slot_of;
  uint32_t* stamp;     // [slab rows] round in which the row was last asked for: such rows are not evicted
  uint32_t round_no;
  uint32_t* miss;
  uint32_t miss_cap;
  // resumable searches (recompute provider, fast kernel): a blocked query parks its state
  // (result set, visited table, the hop in progress) in qstate and carries on from there in the
  // next round; qlist = the queries of this round (NULL = 0 .. nq - 1)
  uint32_t* qstate;
  uint32_t qstate_words;  // per query
  uint32_t* qflag;        // [all queries] 1 = state parked
  const uint32_t* qlist;
  // bf16 rows: classify_queries_kernel sorts the batch into the queries whose elements are all bf16
  // values (qsel_h, count ticket[7], head ticket[4]: answered by the QH instantiation, which keeps
  // the query in LDS as bf16 bit patterns) and the others (qsel, count ticket[5], head ticket[6]:
  // the float32-query kernel in list mode, qsel_mode = 1)
  uint32_t* qsel;
  uint32_t* qsel_h;
  uint32_t qsel_mode;
  // HnswGraph facade (hnsw.rs): adjacency of the layers above 0 for the greedy descent
  const uint64_t* const* layer_off;  // [max_level + 1] device pointers (index 0 unused)
  const uint32_t* const* layer_adj;
  uint32_t max_level;
  uint32_t hnsw_order;  // fast kernel: heaps ordered on the distance alone, equal distances -> exact kernel
  uint32_t seq_max;     // hops with at most this many pushes insert one by one (cheaper than a merge)
  uint32_t* q_entry;    // [nq] layer-0 entry per query after the greedy descent (HnswGraph), or NULL
  uint32_t* q_evals;    // [nq] distance evaluations of the descent (+ 1 for the entry point)
  // two-level search (extension, leann_search_two_level)
  const float* tl_tables;     // [nq][tl_m][tl_K] distances of build_distance_tables
  const uint16_t* tl_codes;   // [tl_ncodes][tl_m]
  uint64_t tl_ncodes;
  uint32_t tl_m, tl_K;
  float tl_ratio;
  uint32_t tl_wcap;           // entries of the approximate queue kept in LDS (multiple of 64)
  // recompute provider, two-level search: a query that parks on an absent row also names the next tl_prefetch
  // unpromoted entries of its queue past the promoted prefix (count in ticket[14], list in pref) -- the nodes
  // it will most likely promote in its next hops, encoded in the same round.  Answers do not depend on it.
  uint32_t tl_prefetch;
  uint32_t* pref;
  uint32_t pref_cap;
};

// ------------------------------------------------------------- sorted result set
// R as a sorted array (ascending (OrderedFloat d, id)) of up to 64*S entries; entry e lives in
// slot e / 64 of lane e % 64.  Distances are held as their order-preserving integer image
// (ordkey) so that every comparison is an unsigned compare; the image is invertible because
// the fast kernel hands queries that meet a NaN or a -0.0 distance to the exact kernel.  id
// bit 31 marks "already expanded".  Entries at or past `len` hold the all-ones key and id
// (greater than every real key, and "expanded"), so neither the position count nor the search
// for the next candidate needs a length test.
constexpr uint32_t KEY_MAX = 0xFFFFFFFFu;

__device__ __forceinline__ float key_to_dist(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

__device__ __forceinline__ bool odd_distance(float d) {
  return d != d || __float_as_uint(d) == 0x80000000u;
}

// (every member is force-inlined: one call left out of line makes `this` escape and the whole set --
// kd, id, len -- lives in scratch memory instead of registers)
template <int S>
struct RSet {
  uint32_t kd[S];
  uint32_t id[S];
  uint32_t len;  // wave-uniform

  __device__ __forceinline__ void init() {
#pragma unroll
    for (int s = 0; s < S; ++s) { kd[s] = KEY_MAX; id[s] = KEY_MAX; }
    len = 0;
  }
  __device__ __forceinline__ uint32_t key_at(uint32_t e) const {
    uint32_t r = 0;
#pragma unroll
    for (int s = 0; s < S; ++s)
      if ((int)(e >> 6) == s) r = rl_u(kd[s], e & 63);
    return r;
  }
  __device__ __forceinline__ uint32_t id_at(uint32_t e) const {
    uint32_t r = 0;
#pragma unroll
    for (int s = 0; s < S; ++s)
      if ((int)(e >> 6) == s) r = rl_u(id[s], e & 63);
    return r;
  }
  // first entry not yet expanded, or 0xFFFFFFFF
  __device__ __forceinline__ uint32_t first_unexpanded() const {
#pragma unroll
    for (int s = 0; s < S; ++s) {
      uint64_t m = ballot(!(id[s] & FLAG_EXP));
      if (m) return s * 64 + (uint32_t)__ffsll((long long)m) - 1;
    }
    return 0xFFFFFFFFu;
  }
  __device__ __forceinline__ void mark_expanded(uint32_t e) {
    const int lane = threadIdx.x;
#pragma unroll
    for (int s = 0; s < S; ++s)
      if ((int)(e >> 6) == s && lane == (int)(e & 63)) id[s] |= FLAG_EXP;
  }
  // Inserts (nk, nid) keeping the order; entries at index >= cap fall off the end.
  __device__ __forceinline__ void insert(uint32_t nk, uint32_t nid, uint32_t cap) {
    const uint32_t lane = threadIdx.x;
    uint32_t pos = 0;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const uint64_t lt = ballot(kd[s] < nk);
      const uint64_t eq = ballot(kd[s] == nk);
      const uint64_t il = ballot((id[s] & ID_MASK) < nid);
      pos += (uint32_t)__popcll(lt | (eq & il));
    }
#pragma unroll
    for (int s = S - 1; s >= 0; --s) {
      if (pos >= 64u * (uint32_t)(s + 1)) continue;  // uniform: this slot stays as it is
      uint32_t uk = shr1_u(kd[s]);
      uint32_t ui = shr1_u(id[s]);
      if (s > 0) {  // lane 0 takes the last entry of the slot below
        const uint32_t pk = rl_u(kd[s - 1], 63), pi = rl_u(id[s - 1], 63);
        uk = lane == 0 ? pk : uk;
        ui = lane == 0 ? pi : ui;
      }
      const uint32_t e = (uint32_t)s * 64u + lane;
      const bool mv = e > pos && e < cap;
      kd[s] = mv ? uk : kd[s];
      id[s] = mv ? ui : id[s];
      const bool here = e == pos;
      kd[s] = here ? nk : kd[s];
      id[s] = here ? nid : id[s];
    }
  }
};

// One hop's pushes at once (leann.rs:953-970 run for every kept neighbour in CSR order).  With
// C = the neighbours whose key is below the worst result at the start of the hop, the sequential
// rule `nd < worst` admits exactly C as long as every j in C still finds fewer than ef smaller
// keys among R and the members of C before it (checked); the resulting R is the merge of R and
// C truncated to ef, and each of the |C| evictions lowers the worst distance when the largest
// |C| + 1 keys of the union have pairwise different distances (checked) -- then no evicted entry
// stays poppable and the tie list empties.  Everything else (R filling up inside the hop, equal
// distances, more than kBatchMax candidates) returns false and takes the one-by-one loop.
// The per-candidate loop has no serial SALU<->VALU round trip: its iterations only accumulate.
constexpr uint32_t kBatchMax = 32;
// entries of the merge buffer (two arrays of this many words); its head doubles as the compaction
// list of a hop's unvisited ids, up to 128 of them with WIDE rows
__host__ __device__ constexpr uint32_t mbuf_entries(uint32_t ef) { return ef + kBatchMax > 64u ? ef + kBatchMax : 64u; }
constexpr int BI_DONE = 1, BI_FALLBACK = 0, BI_TIE = 2;

template <int S>
__device__ __forceinline__ int batch_insert(RSet<S>& rs, uint32_t ef, uint32_t wk0, uint64_t C,
                                            uint32_t nk, uint32_t uid, uint32_t* mbuf, bool strict_ties) {
  const uint32_t lane = threadIdx.x;
  const bool full = rs.len >= ef;
  const uint32_t nC = (uint32_t)__popcll(C);
  const bool oversize = nC > kBatchMax || (!full && rs.len + nC > ef);
  if (oversize && !strict_ties) return BI_FALLBACK;
  const bool inC = (C >> lane) & 1ull;
  uint32_t c[S], idm[S];
#pragma unroll
  for (int s = 0; s < S; ++s) { c[s] = 0; idm[s] = rs.id[s] & ID_MASK; }
  uint32_t a = 0, r = 0, b = 0;
  bool tie = false;
  for (uint64_t rem = C; rem; rem &= rem - 1) {
    const int j = __ffsll((long long)rem) - 1;
    const uint32_t kj = rl_u(nk, j), ij = rl_u(uid, j);
    uint32_t aj = 0;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const bool eq = rs.kd[s] == kj;
      const bool less = rs.kd[s] < kj || (eq && idm[s] < ij);  // this entry sorts before j
      aj += (uint32_t)__popcll(ballot(less));
      c[s] += less ? 0u : 1u;
      tie |= eq;
    }
    const bool ceq = nk == kj && (int)lane != j;
    const bool after = nk > kj || (ceq && uid > ij);  // j sorts before this lane's candidate
    r += after ? 1u : 0u;
    b += (after && (int)lane > j) ? 1u : 0u;
    tie |= ceq && inC;
    a = (int)lane == j ? aj : a;
  }
  (void)wk0;
  // HnswGraph orders its heaps on the distance alone (hnsw.rs:136-141): with two equal distances
  // in play the pop / eviction order is the heap's, not (distance, id) -> the exact kernel decides
  if (strict_ties && ballot(tie)) return BI_TIE;
  if (oversize) return BI_FALLBACK;
  if (full) {
    if (ballot(inC && a + b >= ef)) return BI_FALLBACK;  // no longer below the worst at its turn
    if (ballot(tie)) return BI_FALLBACK;
    // R's own largest nC + 1 distances must differ pairwise
    bool rt = false;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      uint32_t prev = shr1_u(rs.kd[s]);
      if (s > 0) { const uint32_t pk = rl_u(rs.kd[s - 1], 63); prev = lane == 0 ? pk : prev; }
      const uint32_t e = (uint32_t)s * 64u + lane;
      rt |= e >= ef - nC && e < ef && e > 0 && rs.kd[s] == prev;
    }
    if (ballot(rt)) return BI_FALLBACK;
  }
  const uint32_t newlen = rs.len + nC < ef ? rs.len + nC : ef;
  uint32_t* mk = mbuf;
  uint32_t* mi = mbuf + (ef + kBatchMax);
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const uint32_t e = (uint32_t)s * 64u + lane;
    if (e < rs.len) { mk[e + c[s]] = rs.kd[s]; mi[e + c[s]] = rs.id[s]; }
  }
  if (inC) { mk[a + r] = nk; mi[a + r] = uid; }
  wave_sync();
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const uint32_t e = (uint32_t)s * 64u + lane;
    const bool live = e < newlen;
    const uint32_t e2 = live ? e : 0u;
    const uint32_t vk = mk[e2], vi = mi[e2];
    rs.kd[s] = live ? vk : KEY_MAX;
    rs.id[s] = live ? vi : KEY_MAX;
  }
  wave_sync();
  rs.len = newlen;
  return BI_DONE;
}

// Rust BinaryHeap ([external]: std): max-heap w.r.t. `less_eq`.  Operated by lane 0 only.
struct ResultOrder {  // (OrderedFloat<f32>, u64), leann.rs:908
  __device__ static bool le(float ad, uint32_t ai, float bd, uint32_t bi) {
    uint32_t ka = ordkey(ad), kb = ordkey(bd);
    return ka < kb || (ka == kb && ai <= bi);
  }
};
struct CandOrder {  // Reverse<(OrderedFloat<f32>, u64)>, leann.rs:907
  __device__ static bool le(float ad, uint32_t ai, float bd, uint32_t bi) {
    return ResultOrder::le(bd, bi, ad, ai);
  }
};

struct HnswCandOrder {  // hnsw.rs:136-141: Candidate::cmp = other.distance.cmp(self.distance)
  __device__ static bool le(float ad, uint32_t, float bd, uint32_t) { return ordkey(bd) <= ordkey(ad); }
};
struct HnswResultOrder {  // Reverse<Candidate>, hnsw.rs:349
  __device__ static bool le(float ad, uint32_t, float bd, uint32_t) { return ordkey(ad) <= ordkey(bd); }
};

template <class ORD>
__device__ void heap_sift_up(float* hd, uint32_t* hi, uint64_t start, uint64_t pos) {
  float ed = hd[pos];
  uint32_t ei = hi[pos];
  while (pos > start) {
    uint64_t parent = (pos - 1) / 2;
    if (ORD::le(ed, ei, hd[parent], hi[parent])) break;
    hd[pos] = hd[parent];
    hi[pos] = hi[parent];
    pos = parent;
  }
  hd[pos] = ed;
  hi[pos] = ei;
}

template <class ORD>
__device__ void heap_push(float* hd, uint32_t* hi, uint64_t& len, float d, uint32_t id) {
  hd[len] = d;
  hi[len] = id;
  len += 1;
  heap_sift_up<ORD>(hd, hi, 0, len - 1);
}

template <class ORD>
__device__ void heap_pop(float* hd, uint32_t* hi, uint64_t& len, float& od, uint32_t& oi) {
  // Vec::pop the last item; if the heap is not empty swap it with the root and
  // sift_down_to_bottom(0) + sift_up
  len -= 1;
  float itd = hd[len];
  uint32_t iti = hi[len];
  if (len > 0) {
    float rd = hd[0];
    uint32_t ri = hi[0];
    uint64_t end = len, pos = 0, child = 1;
    while (end >= 2 && child <= end - 2) {
      if (ORD::le(hd[child], hi[child], hd[child + 1], hi[child + 1])) child += 1;
      hd[pos] = hd[child];
      hi[pos] = hi[child];
      pos = child;
      child = 2 * pos + 1;
    }
    if (child == end - 1) {
      hd[pos] = hd[child];
      hi[pos] = hi[child];
      pos = child;
    }
    hd[pos] = itd;
    hi[pos] = iti;
    heap_sift_up<ORD>(hd, hi, 0, pos);
    itd = rd;
    iti = ri;
  }
  od = itd;
  oi = iti;
}

// Re-emits the first k results in the reference's order when equal distances make Rust's
// BinaryHeap array layout observable (results.into_iter() + stable sort, leann.rs:984-986):
// replays the logged sequence of results.push (and the pop that follows each push beyond ef)
// on an exact BinaryHeap emulation in LDS.  The result SET of the fast kernel is already exact.
__device__ void replay_result_order(const uint2* plog, uint32_t npush, uint32_t ef, uint32_t k,
                                    uint32_t qi, float* res_d, uint32_t* res_i, uint2* stage,
                                    uint64_t* out_ids, float* out_dist, uint32_t* out_count) {
  const int lane = threadIdx.x;
  uint64_t rlen = 0;
  for (uint32_t base = 0; base < npush; base += 64) {
    if (base + lane < npush) stage[lane] = plog[base + lane];
    __syncthreads();
    if (lane == 0) {
      uint32_t cnt = npush - base < 64 ? npush - base : 64;
      for (uint32_t i = 0; i < cnt; ++i) {
        heap_push<ResultOrder>(res_d, res_i, rlen, __uint_as_float(stage[i].x), stage[i].y);
        if (rlen > ef) {
          float dd;
          uint32_t di;
          heap_pop<ResultOrder>(res_d, res_i, rlen, dd, di);
        }
      }
    }
    __syncthreads();
  }
  if (lane == 0) {
    // only the first k entries of the stable sort are needed: k rounds of "first minimum"
    uint32_t outn = rlen < k ? (uint32_t)rlen : k;
    for (uint32_t o = 0; o < outn; ++o) {
      uint64_t best = o;
      for (uint64_t i = o + 1; i < rlen; ++i)
        if (res_d[i] < res_d[best]) best = i;  // strict: the earliest of equal distances wins
      float bd = res_d[best];
      uint32_t bi = res_i[best];
      for (uint64_t i = best; i > o; --i) {  // keep the relative order of the others (stable)
        res_d[i] = res_d[i - 1];
        res_i[i] = res_i[i - 1];
      }
      res_d[o] = bd;
      res_i[o] = bi;
      out_ids[(uint64_t)qi * k + o] = (uint64_t)bi;
      out_dist[(uint64_t)qi * k + o] = bd;
    }
    out_count[qi] = outn;
  }
  __syncthreads();
}

// HnswGraph::search, hnsw.rs:478-497: greedy descent from the top layer to layer 1 -- per round
// the neighbours of the node the round started at are scanned in order and `current` moves to
// every strictly closer one.  One wave per query; leaves the layer-0 entry in q_entry (a missing
// node id is reported through status / payload like everywhere else).
template <int METRIC_API>
__global__ __launch_bounds__(64) void hnsw_descent_kernel(SearchParams p) {
  constexpr int METRIC = METRIC_API == ISL_METRIC_COSINE ? METRIC_COSINE_PRE : METRIC_API;
  extern __shared__ __align__(16) unsigned char smem[];
  float* qs = reinterpret_cast<float*>(smem);
  const float* emb = reinterpret_cast<const float*>(p.emb);
  const int lane = threadIdx.x;
  for (uint32_t qi = blockIdx.x; qi < p.nq; qi += gridDim.x) {
    __syncthreads();
    const float q_norm = load_query<METRIC>(p.queries + (uint64_t)qi * p.d, p.d, qs);
    uint32_t status = QS_OK, cV = 1;
    uint64_t payload = 0;
    uint32_t entry = p.entry;
    float ed = 0.0f;
    if ((uint64_t)entry >= p.nvec) {
      status = QS_NODE_NOT_FOUND;
      payload = entry;
    } else {
      float e_aux = METRIC == METRIC_COSINE_PRE ? p.norm2[entry] : 0.0f;
      ed = rl_f(direct_distances<METRIC>(emb, p.stride, p.d, entry, 1, qs, q_norm, e_aux), 0);
      for (uint32_t layer = p.max_level; layer >= 1 && status == QS_OK; --layer) {
        const uint64_t* loff = p.layer_off[layer];
        const uint32_t* ladj = p.layer_adj[layer];
        for (;;) {
          const uint64_t g0 = loff[entry], g1 = loff[entry + 1];
          const uint32_t gdeg = (uint32_t)(g1 - g0);
          bool changed = false;
          uint32_t cur = entry;
          float cur_d = ed;
          for (uint32_t base = 0; base < gdeg && status == QS_OK; base += 64) {
            const uint32_t R = gdeg - base < 64 ? gdeg - base : 64;
            const uint32_t gid = (uint32_t)lane < R ? ladj[g0 + base + lane] : 0u;
            const uint64_t gbad = ballot((uint32_t)lane < R && (uint64_t)gid >= p.nvec);
            if (gbad) {  // HnswGraph::distance -> NodeNotFound, hnsw.rs:449-455
              status = QS_NODE_NOT_FOUND;
              payload = rl_u(gid, __ffsll((long long)gbad) - 1);
              break;
            }
            const float g_aux = (METRIC == METRIC_COSINE_PRE && (uint32_t)lane < R) ? p.norm2[gid] : 0.0f;
            const float gd = direct_distances<METRIC>(emb, p.stride, p.d, gid, R, qs, q_norm, g_aux);
            cV += R;
            for (uint32_t r = 0; r < R; ++r) {  // list order, strict `<` (hnsw.rs:485)
              const float dr = rl_f(gd, (int)r);
              if (dr < cur_d) { cur = rl_u(gid, (int)r); cur_d = dr; changed = true; }
            }
          }
          entry = cur;
          ed = cur_d;
          if (!changed || status != QS_OK) break;
        }
      }
    }
    if (lane == 0) {
      p.q_entry[qi] = entry;
      p.q_evals[qi] = cV;
      p.status[qi] = status;
      p.payload[qi] = payload;
    }
  }
}

// Recompute provider: true when every row of `uid` (lanes < n) is materialised; otherwise the
// absent ids are appended to the miss list and the caller stops the query with QS_BLOCKED.
constexpr uint32_t kNoSlot = 0xFFFFFFFFu;    // slot_of[id]: no row
constexpr uint32_t kSlotClaim = 0xFFFFFFFEu; // transient, while a round's misses are de-duplicated
__device__ __forceinline__ bool rows_present(const SearchParams& p, uint32_t uid, uint32_t n) {
  if (!p.slot_of) return true;
  const uint32_t lane = threadIdx.x;
  const uint32_t sl = lane < n ? p.slot_of[uid] : 0u;
  const bool absent = lane < n && sl >= kSlotClaim;
  if (lane < n && !absent) p.stamp[sl] = p.round_no;  // in use this round, whether or not the hop can run yet
  const uint64_t am = ballot(absent);
  if (!am) return true;
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(&p.ticket[13], (uint32_t)__popcll(am));
  base = uni(base);
  const uint32_t rank = (uint32_t)__popcll(am & ((1ull << lane) - 1ull));
  if (absent && base + rank < p.miss_cap) p.miss[base + rank] = uid;
  return false;
}
// The same for a hop of up to 128 rows (entry e in lane e % 64 of uid_lo / uid_hi): ONE run of the
// miss list per hop -- the provider serves a prefix of the list when its row cache is small, and a
// hop whose rows arrive in two rounds could lose the first part again before the second is there.
__device__ __forceinline__ bool rows_present2(const SearchParams& p, uint32_t uid_lo, uint32_t n_lo,
                                              uint32_t uid_hi, uint32_t n_hi) {
  if (!p.slot_of) return true;
  const uint32_t lane = threadIdx.x;
  const uint32_t s_lo = lane < n_lo ? p.slot_of[uid_lo] : 0u, s_hi = lane < n_hi ? p.slot_of[uid_hi] : 0u;
  const bool a_lo = lane < n_lo && s_lo >= kSlotClaim;
  const bool a_hi = lane < n_hi && s_hi >= kSlotClaim;
  if (lane < n_lo && !a_lo) p.stamp[s_lo] = p.round_no;
  if (lane < n_hi && !a_hi) p.stamp[s_hi] = p.round_no;
  const uint64_t m_lo = ballot(a_lo), m_hi = ballot(a_hi);
  if (!(m_lo | m_hi)) return true;
  const uint32_t c_lo = (uint32_t)__popcll(m_lo);
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(&p.ticket[13], c_lo + (uint32_t)__popcll(m_hi));
  base = uni(base);
  const uint64_t below = (1ull << lane) - 1ull;
  if (a_lo && base + (uint32_t)__popcll(m_lo & below) < p.miss_cap) p.miss[base + (uint32_t)__popcll(m_lo & below)] = uid_lo;
  if (a_hi && base + c_lo + (uint32_t)__popcll(m_hi & below) < p.miss_cap)
    p.miss[base + c_lo + (uint32_t)__popcll(m_hi & below)] = uid_hi;
  return false;
}
// row of node `id` in the provider's table (lanes >= n: any valid row)
__device__ __forceinline__ uint32_t row_index(const SearchParams& p, uint32_t id, bool live) {
  if (!p.slot_of) return id;
  return live ? p.slot_of[id] : 0u;
}

// visited.insert(id) of leann.rs:933-937 for one id per lane: true when the id was not in the set.
// The set is an open-addressing table in LDS; once that is 7/8 full (`ovf`) new ids go to the
// wave's overflow table in HBM.  (A function with value parameters, not a lambda: captures by
// reference cost the headline kernel 37 VGPRs and a scratch frame.)
__device__ __forceinline__ bool visited_insert(uint32_t* htab, uint32_t hcap, bool ovf,
                                               uint32_t* otab, uint32_t obits, uint32_t omask, uint32_t id,
                                               bool act) {
  bool fresh = false;
  if (act) {
    uint32_t h = hslot_cap(id, hcap);
    if (!ovf) {
      for (;;) {
        uint32_t old = atomicCAS(&htab[h], EMPTY, id);
        if (old == EMPTY) { fresh = true; break; }
        if (old == id) break;
        h = h + 1 == hcap ? 0u : h + 1;
      }
    } else {
      bool found = false;
      for (;;) {
        uint32_t cur = htab[h];
        if (cur == id) { found = true; break; }
        if (cur == EMPTY) break;
        h = h + 1 == hcap ? 0u : h + 1;
      }
      if (!found) {
        uint32_t g = hslot(id, obits);
        for (;;) {
          uint32_t old = atomicCAS(&otab[g], EMPTY, id);
          if (old == EMPTY) { fresh = true; break; }
          if (old == id) break;
          g = (g + 1) & omask;
        }
      }
    }
  }
  return fresh;
}

// bf16 rows: which queries of the batch consist of bf16 values only (one wave per query)
__global__ __launch_bounds__(64) void classify_queries_kernel(SearchParams p) {
  const uint32_t lane = threadIdx.x;
  for (uint32_t qi = blockIdx.x; qi < p.nq; qi += gridDim.x) {
    const uint32_t* q = reinterpret_cast<const uint32_t*>(p.queries + (uint64_t)qi * p.d);
    bool bad = false;
    for (uint32_t j = lane; j < p.d; j += 64) bad |= (q[j] & 0xFFFFu) != 0u;
    if (lane == 0) {
      if (ballot(bad)) p.qsel[atomicAdd(&p.ticket[5], 1u)] = qi;
      else p.qsel_h[atomicAdd(&p.ticket[7], 1u)] = qi;
    }
  }
}

// ------------------------------------------------------------------ fast kernel
// WIDE = adjacency rows of up to 128 ids (LeannConfig::accurate() has m0 = 96, leann.rs:419-429):
// a lane then holds two ids of the row, and the kept neighbours of a hop are evaluated and inserted
// 64 at a time in CSR order -- the same sequential rule, run over two slices.
// RESUME = searches over the recompute provider: a query that meets an absent row parks its whole
// state and is taken up again, in the hop it stopped at, once the provider has encoded the row.
// (amdgpu_waves_per_eu(3): the LDS footprint admits 12 waves per CU; 168 VGPRs keep three per SIMD,
// 169 round up to 176 and leave two)
template <int S, int METRIC_API, typename ROWT, bool WIDE, bool RESUME = false, bool QH = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) void leann_search_fast(SearchParams p) {
  static_assert(!QH || (sizeof(ROWT) == 2 && !RESUME), "the bf16 query operand goes with bf16 rows");
  constexpr uint32_t kMaxDeg = WIDE ? 128u : 64u;
  // parked state of one query, in words: 16 scalars, tie list, the hop's unvisited ids (2 x 64),
  // result set (S x (64 keys + 64 ids)), visited table
  constexpr uint32_t kStTie = 16, kStHop = 80, kStR = 208, kStTab = 208 + S * 128;
  constexpr int METRIC = METRIC_API == ISL_METRIC_COSINE ? METRIC_COSINE_PRE : METRIC_API;
  const ROWT* const emb = reinterpret_cast<const ROWT*>(p.emb);
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x;
  const uint32_t hcap = p.hcap;
  const uint32_t hlimit = hcap - hcap / 8;  // load factor 0.875
  uint32_t* htab = reinterpret_cast<uint32_t*>(smem);
  // merge buffer of batch_insert; its head doubles as the 64 words of the id compaction
  uint32_t* mbuf = htab + hcap;
  uint32_t* scratch = mbuf;
  float* qs = reinterpret_cast<float*>(mbuf + 2 * mbuf_entries(p.ef));
  const uint32_t ocap = 1u << p.obits;
  const uint32_t omask = ocap - 1;
  const uint32_t olimit = ocap - ocap / 4;
  uint32_t* otab = p.otab + (size_t)blockIdx.x * ocap;
  const uint32_t ef = p.ef;

  for (;;) {
    uint32_t qi = 0;
    if constexpr (sizeof(ROWT) == 2 && !QH && !RESUME) {
      if (p.qsel_mode) {  // the queries the bf16-query launch passed on
        if (lane == 0) qi = atomicAdd(&p.ticket[6], 1u);
        qi = uni(qi);
        if (qi >= *((volatile uint32_t*)&p.ticket[5])) break;
        qi = p.qsel[qi];
      } else {
        if (lane == 0) qi = atomicAdd(&p.ticket[0], 1u);
        qi = uni(qi);
        if (qi >= p.nq) break;
      }
    } else if constexpr (QH) {
      if (lane == 0) qi = atomicAdd(&p.ticket[4], 1u);
      qi = uni(qi);
      if (qi >= *((volatile uint32_t*)&p.ticket[7])) break;
      qi = p.qsel_h[qi];
    } else {
      if (lane == 0) qi = atomicAdd(&p.ticket[0], 1u);
      qi = uni(qi);
      if (qi >= p.nq) break;
    }
    if constexpr (RESUME) {
      if (p.qlist) qi = p.qlist[qi];
      otab = p.otab + (size_t)qi * ocap;  // a parked query may come back on another wave
    }

    if (p.q_entry && p.status[qi] != QS_OK) {  // the greedy descent already failed this query
      if (lane == 0) {
        p.out_count[qi] = 0;
        p.ctr[qi * 4 + 0] = 0; p.ctr[qi * 4 + 1] = 0; p.ctr[qi * 4 + 2] = p.q_evals[qi]; p.ctr[qi * 4 + 3] = 0;
      }
      continue;
    }
    const uint64_t t_start = __builtin_amdgcn_s_memrealtime();
    bool resumed = false;
    uint32_t* qst = nullptr;
    if constexpr (RESUME) {
      qst = p.qstate + (size_t)qi * p.qstate_words;
      resumed = uni(p.qflag[qi]) == 1u;
    }
    if (resumed) { for (uint32_t i = lane; i < hcap; i += 64) htab[i] = qst[kStTab + i]; }
    else { for (uint32_t i = lane; i < hcap; i += 64) htab[i] = EMPTY; }
    float q_norm;
    if constexpr (QH) {
      bool representable;
      q_norm = load_query_bf16<METRIC>(p.queries + (uint64_t)qi * p.d, p.d, qs, &representable);  // syncs
      if (!representable) {  // (cannot happen after classify_queries_kernel; kept as a guard)
        if (lane == 0) p.qsel[atomicAdd(&p.ticket[5], 1u)] = qi;
        continue;
      }
    } else {
      q_norm = load_query<METRIC>(p.queries + (uint64_t)qi * p.d, p.d, qs);  // syncs
    }

    RSet<S> rs;
    rs.init();
    uint32_t hcount = 0, ocount = 0;
    bool ovf = false;
    uint32_t status = QS_OK;
    uint64_t payload = 0;
    uint32_t cH = 0, cE = 0, cV = 0, cP = 0;
    // Tie-evicted candidates: entries pushed out of R whose distance
    // equals the new worst distance stay poppable in the reference's candidate heap.  Lane i <
    // tcount holds one id; they all share the current worst distance and die when it drops.
    uint32_t t_id = 0, tcount = 0;
    uint2* plog = p.plog + (size_t)qi * p.plog_cap;
    uint64_t tp0 = 0, tp1 = 0, tp2 = 0, tp3 = 0, tmark = 0, ngroups = 0, nhops_rows = 0;
#define ISL_MARK(acc) if (p.prof) { uint64_t now_ = __builtin_amdgcn_s_memrealtime(); acc += now_ - tmark; tmark = now_; }

    // the hop in progress: its unvisited ids in CSR order (entry e in lane e % 64 of uid / uid_hi)
    // and how many of them apply_pruning_strategy keeps
    uint32_t parked_nu = 0;
    bool have_hop = false;  // RESUME: the parked hop is evaluated before anything is popped
    if (resumed) {
      hcount = qst[0]; ocount = qst[1]; ovf = qst[2] != 0u; tcount = qst[3];
      cH = qst[4]; cE = qst[5]; cV = qst[6]; cP = qst[7];
      rs.len = qst[8]; parked_nu = qst[9];
      t_id = qst[kStTie + lane];
      scratch[lane] = qst[kStHop + lane];  // the parked hop's unvisited ids, where the compaction leaves them
      scratch[64 + lane] = qst[kStHop + 64 + lane];
#pragma unroll
      for (int s = 0; s < S; ++s) {
        rs.kd[s] = qst[kStR + s * 128 + lane];
        rs.id[s] = qst[kStR + s * 128 + 64 + lane];
      }
      have_hop = true;
      wave_sync();
    } else
    // entry point: provider.compute_embedding(entry) + distance, leann.rs:911-916
    if (!p.q_entry && (uint64_t)p.entry >= p.nvec) {
      status = QS_NODE_NOT_FOUND;
      payload = p.entry;
    } else if (!rows_present(p, p.entry, 1)) {
      status = QS_BLOCKED;
    } else {
      // HnswGraph: the layer-0 search starts where the greedy descent (hnsw_descent_kernel) ended
      const uint32_t entry = p.q_entry ? p.q_entry[qi] : p.entry;
      const uint32_t erow = row_index(p, entry, true);
      float e_aux = METRIC == METRIC_COSINE_PRE ? p.norm2[erow] : 0.0f;
      float ed = direct_distances<METRIC, ROWT, QH>(emb, p.stride, p.d, erow, 1, qs, q_norm, e_aux);
      ed = rl_f(ed, 0);
      cV = p.q_entry ? p.q_evals[qi] : 1;
      if (lane == 0) htab[hslot_cap(entry, hcap)] = entry;
      hcount = 1;
      if (odd_distance(ed) && status == QS_OK) { status = QS_REDO; payload = 5; }
      rs.insert(ordkey(ed), entry, ef);
      rs.len = 1;
      if (lane == 0) plog[0] = make_uint2(__float_as_uint(ed), entry);
      cP = 1;
      wave_sync();
    }

    if (p.prof) tmark = __builtin_amdgcn_s_memrealtime();
    while (status == QS_OK) {
     uint32_t nu;  // unvisited neighbours of the hop
     if constexpr (RESUME) {
       // a parked hop: its ids are in `scratch` already; R has not changed since, so neither has :944
       if (have_hop) { nu = parked_nu; goto hop_ready; }
     }
     {
      // candidates.pop(): the smallest unexpanded key of R (leann.rs:922); when none is
      // left every remaining candidate is farther than the worst result -> break (:924-928)
      uint32_t e = rs.first_unexpanded();
      uint32_t cid;
      if (e != 0xFFFFFFFFu) {
        cid = rs.id_at(e) & ID_MASK;
        rs.mark_expanded(e);
      } else if (tcount > 0) {
        // every key of R is expanded; the next candidates are the tie-evicted ones, whose
        // distance equals the worst result (`dist > worst` is false, leann.rs:925): smallest id first
        uint32_t best = rl_u(t_id, 0);
        int bl = 0;
        for (uint32_t i = 1; i < tcount; ++i) {
          uint32_t v = rl_u(t_id, (int)i);
          if (v < best) { best = v; bl = (int)i; }
        }
        uint32_t last = rl_u(t_id, (int)(tcount - 1));
        if (lane == bl) t_id = last;
        tcount -= 1;
        cid = best;
      } else {
        break;
      }
      if ((uint64_t)cid >= p.num_nodes) continue;  // get_neighbors -> None, leann.rs:227-229
      // adjacency: fixed-width rows (the padded copy made at the first search, or the table of a
      // graph under construction, build.hip) -- degree and ids are two independent loads, one
      // memory round trip per hop; plain CSR (offset, then ids) otherwise
      uint32_t deg, nid, nid1 = EMPTY;
      if (p.ell_w) {
        const uint32_t slot = (uint32_t)lane < p.ell_w ? (uint32_t)lane : p.ell_w - 1u;
        deg = p.ell_deg[cid];
        const uint32_t raw = p.adj[(uint64_t)cid * p.ell_w + slot];
        nid = (uint32_t)lane < deg ? raw : EMPTY;
        if constexpr (WIDE)
          if (deg > 64 && deg <= kMaxDeg && (uint32_t)lane + 64u < deg) nid1 = p.adj[(uint64_t)cid * p.ell_w + 64u + lane];
      } else {
        const uint64_t o0 = p.off[cid], o1 = p.off[cid + 1];
        deg = (uint32_t)(o1 - o0);
        nid = ((uint32_t)lane < deg && deg <= kMaxDeg) ? p.adj[o0 + lane] : EMPTY;
        if constexpr (WIDE)
          if (deg > 64 && deg <= kMaxDeg && (uint32_t)lane + 64u < deg) nid1 = p.adj[o0 + 64u + lane];
      }
      cH += 1;
      cE += deg;
      if (deg == 0) continue;
      if (deg > kMaxDeg) { status = QS_REDO; payload = 1; break; }  // long rows: exact kernel
      bool active = (uint32_t)lane < deg;

      ISL_MARK(tp0)  // selection + adjacency fetch
      // visited.insert(n), leann.rs:933-937 (rows hold no duplicate ids on the device)
      if (!ovf && hcount + deg > hlimit) ovf = true;
      const bool is_new = visited_insert(htab, hcap, ovf, otab, p.obits, omask, nid, active);
      uint64_t nm = ballot(is_new);
      nu = (uint32_t)__popcll(nm);
      bool is_new1 = false;
      uint64_t nm1 = 0;
      if constexpr (WIDE) {
        if (deg > 64) {
          is_new1 = visited_insert(htab, hcap, ovf, otab, p.obits, omask, nid1, (uint32_t)lane + 64u < deg);
          nm1 = ballot(is_new1);
          nu += (uint32_t)__popcll(nm1);
        }
      }
      if (!ovf) hcount += nu;
      else {
        ocount += nu;
        if (ocount > olimit) { status = QS_REDO; payload = 2; break; }
      }
      if (nu == 0) continue;  // leann.rs:939-941

      // compact the unvisited ids, CSR order preserved
      uint32_t rank = (uint32_t)__popcll(nm & ((1ull << lane) - 1ull));
      if (is_new) scratch[rank] = nid;
      if constexpr (WIDE)
        if (is_new1) scratch[(uint32_t)__popcll(nm) + (uint32_t)__popcll(nm1 & ((1ull << lane) - 1ull))] = nid1;
     }
    hop_ready:
      wave_sync();
      uint32_t uid = (uint32_t)lane < nu ? scratch[lane] : 0u;
      uint32_t uid_hi = 0u;  // entries 64.. of the list (the merge buffer reuses `scratch` below)
      if constexpr (WIDE) uid_hi = (uint32_t)lane + 64u < nu ? scratch[64 + lane] : 0u;
      wave_sync();

      const uint32_t keep_all = prune_keep(p.prune_ratio, p.prune_strategy, nu, rs.len, ef);  // :944
      // compute_embeddings_batch, leann.rs:947: the first missing id fails the query
      uint64_t bad = ballot((uint32_t)lane < keep_all && (uint64_t)uid >= p.nvec);
      if (bad) {
        int bl = __ffsll((long long)bad) - 1;
        status = QS_NODE_NOT_FOUND;
        payload = rl_u(uid, bl);
        break;
      }
      if constexpr (WIDE) {
        if (keep_all > 64) {
          bad = ballot((uint32_t)lane + 64u < keep_all && (uint64_t)uid_hi >= p.nvec);
          if (bad) {
            status = QS_NODE_NOT_FOUND;
            payload = rl_u(uid_hi, __ffsll((long long)bad) - 1);
            break;
          }
        }
      }
      {
        bool here;
        if constexpr (WIDE) here = rows_present2(p, uid, keep_all < 64 ? keep_all : 64, uid_hi, keep_all > 64 ? keep_all - 64 : 0);
        else here = rows_present(p, uid, keep_all);
        if (!here) {
          status = QS_BLOCKED;
          if constexpr (RESUME) {  // park: everything the rest of the search depends on
            if (lane == 0) {
              qst[0] = hcount; qst[1] = ocount; qst[2] = ovf ? 1u : 0u; qst[3] = tcount;
              qst[4] = cH; qst[5] = cE; qst[6] = cV; qst[7] = cP;
              qst[8] = rs.len; qst[9] = nu;
              p.qflag[qi] = 1u;
            }
            qst[kStTie + lane] = t_id;
            qst[kStHop + lane] = uid;
            qst[kStHop + 64 + lane] = uid_hi;
#pragma unroll
            for (int s = 0; s < S; ++s) {
              qst[kStR + s * 128 + lane] = rs.kd[s];
              qst[kStR + s * 128 + 64 + lane] = rs.id[s];
            }
            for (uint32_t i = lane; i < hcap; i += 64) qst[kStTab + i] = htab[i];
            have_hop = true;  // (marks "parked" for the epilogue below)
          }
          break;
        }
        have_hop = false;
      }
      cV += keep_all;
      nhops_rows += 1;
      ISL_MARK(tp1)  // visited set + compaction
     // (one slice unless WIDE: the loop folds away and the common kernel keeps its schedule)
     uint32_t sbase = 0;
     do {
      if constexpr (WIDE) { if (sbase) uid = uid_hi; }
      const uint32_t keep = WIDE ? (keep_all - sbase < 64 ? keep_all - sbase : 64) : keep_all;
      ngroups += (keep + 15) / 16;
      const uint32_t rix = row_index(p, uid, (uint32_t)lane < keep);
      float r_aux = (METRIC == METRIC_COSINE_PRE && (uint32_t)lane < keep) ? p.norm2[rix] : 0.0f;
      float nd = direct_distances<METRIC, ROWT, QH>(emb, p.stride, p.d, rix, keep, qs, q_norm, r_aux);
      ISL_MARK(tp2)  // row fetch + distances

      // leann.rs:953-970 in CSR order; worst = results.peek().  NaN / -0.0 distances have no
      // integer image: the exact kernel takes the query.
      if (ballot((uint32_t)lane < keep && odd_distance(nd))) { status = QS_REDO; payload = 5; break; }
      const uint32_t nk = ordkey(nd);
      uint64_t pending = keep >= 64 ? ~0ull : ((1ull << keep) - 1ull);
      {
        const bool full0 = rs.len >= ef;
        const uint32_t wk0 = full0 ? rs.key_at(ef - 1) : KEY_MAX;
        const uint64_t C = ballot(nk < wk0) & pending;
        if (!C) { ISL_MARK(tp3) continue; }
        const int bi = ((uint32_t)__popcll(C) <= p.seq_max && !p.hnsw_order)
                           ? BI_FALLBACK
                           : batch_insert<S>(rs, ef, wk0, C, nk, uid, mbuf, p.hnsw_order != 0);
        if (bi == BI_TIE) { status = QS_REDO; payload = 6; break; }
        if (bi == BI_DONE) {
          const uint32_t rank = (uint32_t)__popcll(C & ((1ull << lane) - 1ull));
          if (((C >> lane) & 1ull) && cP + rank < p.plog_cap)
            plog[cP + rank] = make_uint2(__float_as_uint(nd), uid);
          cP += (uint32_t)__popcll(C);
          if (full0) tcount = 0;
          ISL_MARK(tp3)
          continue;
        }
      }
      while (pending) {
        const bool full = rs.len >= ef;
        const uint32_t wk = full ? rs.key_at(ef - 1) : KEY_MAX;  // `nd < worst`, leann.rs:959
        const uint64_t pm = ballot(nk < wk) & pending;
        if (!pm) break;
        const int r = __ffsll((long long)pm) - 1;
        const uint32_t id_k = rl_u(nk, r);
        const uint32_t id_i = rl_u(uid, r);
        if (cP < p.plog_cap) {
          const uint32_t id_bits = rl_u(__float_as_uint(nd), r);
          if (lane == 0) plog[cP] = make_uint2(id_bits, id_i);
        }
        if (full) {
          // results.push + pop: the old worst leaves R but stays in the reference's candidate
          // heap.  It can only be popped again while its distance still equals the worst one.
          const uint32_t old_raw = rs.id_at(ef - 1);
          rs.insert(id_k, id_i, ef);
          const uint32_t new_wk = rs.key_at(ef - 1);
          if (wk != new_wk) {
            tcount = 0;
          } else if (!(old_raw & FLAG_EXP)) {
            if (tcount >= 64) { status = QS_REDO; payload = 3; }
            else {
              if (lane == (int)tcount) t_id = old_raw & ID_MASK;
              tcount += 1;
            }
          }
        } else {
          rs.insert(id_k, id_i, ef);
          rs.len += 1;
        }
        cP += 1;
        pending &= ~((2ull << r) - 1ull);
        if (r == 63) pending = 0;
      }
      ISL_MARK(tp3)  // result-set insertion
     } while (WIDE && (sbase += 64) < keep_all && status == QS_OK);  // slices of 64 kept neighbours
    }

    // results sorted by distance, take(k): leann.rs:984-986, :895
    uint32_t outn = rs.len < p.k ? rs.len : p.k;
    if (status == QS_OK) {
      // equal distances inside the returned prefix (or across its boundary) are ordered by
      // BinaryHeap array order in the reference: let the exact kernel reproduce that
      uint32_t chk = rs.len < p.k + 1 ? rs.len : p.k + 1;
      bool tie = false;
#pragma unroll
      for (int s = 0; s < S; ++s) {
        uint32_t e = s * 64 + lane;
        uint32_t nxt = (uint32_t)__shfl_down((int)rs.kd[s], 1);
        if (s + 1 < S) {
          uint32_t nd0 = rl_u(rs.kd[s + 1 < S ? s + 1 : s], 0);
          if (lane == 63) nxt = nd0;
        }
        if (e + 1 < chk && rs.kd[s] == nxt) tie = true;
      }
      if (ballot(tie)) {
        if (cP <= p.plog_cap) status = QS_REPLAY;
        else { status = QS_REDO; payload = 4; }
      }
    }
    if (status == QS_REPLAY) {
      __threadfence_block();
      __syncthreads();  // rare path: keep the full wait before re-reading the push log
      // the search is over: the visited table's LDS (>= 4 KiB) becomes the replay's heap + stage
      float* res_d = reinterpret_cast<float*>(htab);
      uint32_t* res_i = htab + (ef + 1);
      uint2* stage = reinterpret_cast<uint2*>(htab + 2 * (ef + 1));
      replay_result_order(plog, cP, ef, p.k, qi, res_d, res_i, stage, p.out_ids, p.out_dist,
                          p.out_count);
      status = QS_OK;
      outn = 0xFFFFFFFFu;  // outputs already written
      if (lane == 0) atomicAdd(&p.ticket[3], 1u);
    }
    if (status == QS_OK && outn != 0xFFFFFFFFu) {
#pragma unroll
      for (int s = 0; s < S; ++s) {
        uint32_t e = s * 64 + lane;
        if (e < outn) {
          p.out_ids[(uint64_t)qi * p.k + e] = (uint64_t)(rs.id[s] & ID_MASK);
          p.out_dist[(uint64_t)qi * p.k + e] = key_to_dist(rs.kd[s]);
        }
      }
    }
    if (lane == 0) {
      p.status[qi] = status;
      // payload of a successful query: its time in the kernel (100 MHz ticks), for ISL_DEBUG
      p.payload[qi] = status == QS_OK ? (__builtin_amdgcn_s_memrealtime() - t_start) : payload;
      if (outn != 0xFFFFFFFFu) p.out_count[qi] = status == QS_OK ? outn : 0u;
      p.ctr[qi * 4 + 0] = cH;
      p.ctr[qi * 4 + 1] = cE;
      p.ctr[qi * 4 + 2] = cV;
      p.ctr[qi * 4 + 3] = cP;
      if (p.tline) {
        p.tline[qi * 2] = t_start;
        p.tline[qi * 2 + 1] = ((__builtin_amdgcn_s_memrealtime() - t_start) & 0xFFFFFFFFFFull) | ((uint64_t)cH << 40);
      }
      if (p.prof) {
        p.prof[qi * 8 + 0] = tp0; p.prof[qi * 8 + 1] = tp1; p.prof[qi * 8 + 2] = tp2; p.prof[qi * 8 + 3] = tp3;
        p.prof[qi * 8 + 4] = ngroups; p.prof[qi * 8 + 5] = nhops_rows;
      }
      if (status == QS_REDO) {
        p.redo[atomicAdd(&p.ticket[1], 1u)] = qi;
        // why: 1 long row, 2 visited overflow, 3 tie-candidate overflow, 0 push-log overflow,
        // 5 a distance without an integer image (NaN, -0.0)
        atomicAdd(&p.ticket[payload == 5 ? 12u : payload == 6 ? 14u : 8u + ((uint32_t)payload & 3u)], 1u);
      }
    }
    bool parked = false;
    if constexpr (RESUME) {
      parked = status == QS_BLOCKED && have_hop;
      if (!parked && lane == 0) p.qflag[qi] = 0u;  // finished, failed, or handed to the exact kernel
    }
    if (ovf && !parked) {  // leave the overflow table empty for the next query of this slot
      for (uint32_t i = lane; i < ocap; i += 64) otab[i] = EMPTY;
    }
    wave_sync();
  }
}


// rows of the exact kernel: the LDS-tile routine for f32 rows, a plain per-lane walk for bf16
template <int METRIC>
__device__ __forceinline__ float exact_rows(const SearchParams& p, uint32_t rid, uint32_t R,
                                            const float* qs, float* tile, float q_norm, float aux) {
  if (p.emb_bf16)
    return lane_distances_bf16<METRIC>(reinterpret_cast<const uint16_t*>(p.emb), p.stride, p.d, rid, R, qs,
                                       q_norm, aux);
  return wave_distances<METRIC>(reinterpret_cast<const float*>(p.emb), p.stride, p.d, rid, R, qs, tile, q_norm, aux);
}

// ----------------------------------------------------------------- exact kernel
// HNSW = true: HnswGraph::search (hnsw.rs:458-504): greedy descent through the upper layers,
// then the same layer-0 loop with heaps ordered on distance only (hnsw.rs:136-141, 332-402).
template <int METRIC_API, bool HNSW>
__global__ __launch_bounds__(64) void leann_search_exact(SearchParams p) {
  constexpr int METRIC = METRIC_API == ISL_METRIC_COSINE ? METRIC_COSINE_PRE : METRIC_API;
  using CandOrd = typename std::conditional<HNSW, HnswCandOrder, CandOrder>::type;
  using ResOrd = typename std::conditional<HNSW, HnswResultOrder, ResultOrder>::type;
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x;
  const uint32_t ef = p.ef;
  float* tile = reinterpret_cast<float*>(smem);
  uint32_t* scratch = reinterpret_cast<uint32_t*>(tile + TILE_ROWS * TILE_LD);  // 64 ids
  float* dscratch = reinterpret_cast<float*>(scratch + 64);                     // 64 distances
  uint32_t* ctl = reinterpret_cast<uint32_t*>(dscratch + 64);                   // 8 control words
  float* res_d = reinterpret_cast<float*>(ctl + 8);                             // ef + 1
  uint32_t* res_i = reinterpret_cast<uint32_t*>(res_d + (ef + 1));
  float* qs = reinterpret_cast<float*>(res_i + (ef + 1));
  qs = reinterpret_cast<float*>(((uintptr_t)qs + 15) & ~(uintptr_t)15);

  // The scratch (candidate heap, visited bitmap, hop list) comes from the pool every lane of the
  // index shares: a workgroup that finds work claims a free slot and keeps it until its queue is
  // empty.  Holders never wait for anything, so a spinning claimant always gets one.
  uint32_t slot = 0xFFFFFFFFu;
  float* cand_d = nullptr;
  uint32_t* cand_i = nullptr;
  uint32_t* vis = nullptr;
  uint32_t* ulist = nullptr;

  for (;;) {
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(&p.ticket[2], 1u);
    t = uni(t);
    uint32_t nredo = *((volatile uint32_t*)&p.ticket[1]);
    if (t >= nredo) break;
    const uint32_t qi = p.redo[t];
    // Recompute provider (p.xslot): a query that meets an absent row parks here too -- candidate heap,
    // visited bitmap and hop list stay in its pool slot, which it keeps (lock held) across the rounds;
    // result heap and scalars go to the slot's state block.  A slot is then claimed per query, and a
    // query that finds none free gives up for this round (QS_BLOCKED: it starts over later) instead of
    // spinning -- every slot may be held by a parked query.
    bool xresumed = false;
    if (p.xslot) {
      const uint32_t xs = uni(p.xslot[qi]);
      uint32_t sl = 0xFFFFFFFFu;
      if (xs) {
        sl = xs - 1u;
        xresumed = true;
      } else if (slot != 0xFFFFFFFFu) {
        sl = slot;  // the slot this workgroup still holds from its previous (finished) query
      } else {
        if (lane == 0) {
          for (uint32_t tries = 0; tries < 2u * p.pool_slots && sl == 0xFFFFFFFFu; ++tries) {
            const uint32_t c = (blockIdx.x + tries) % p.pool_slots;
            if (atomicCAS(&p.pool_locks[c], 0u, 1u) == 0u) sl = c;
          }
        }
        sl = uni(sl);
        if (sl == 0xFFFFFFFFu) {
          if (lane == 0) {
            p.status[qi] = QS_BLOCKED;
            p.out_count[qi] = 0;
            p.ctr[qi * 4 + 0] = 0; p.ctr[qi * 4 + 1] = 0; p.ctr[qi * 4 + 2] = 0; p.ctr[qi * 4 + 3] = 0;
          }
          continue;
        }
      }
      if (sl != slot) {
        // (a parked query's slot while this workgroup holds another: that one goes back first)
        if (slot != 0xFFFFFFFFu && lane == 0) atomicExch(&p.pool_locks[slot], 0u);
        slot = sl;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        cand_d = p.cand_d + (size_t)slot * p.cand_cap;
        cand_i = p.cand_id + (size_t)slot * p.cand_cap;
        vis = p.vis_bits + (size_t)slot * p.vis_words;
        ulist = p.ulist + (size_t)slot * p.ulist_cap;
      }
    } else
    if (slot == 0xFFFFFFFFu) {
      uint32_t sl = 0;
      if (lane == 0) {
        sl = blockIdx.x % p.pool_slots;
        while (atomicCAS(&p.pool_locks[sl], 0u, 1u) != 0u) {
          sl = sl + 1 == p.pool_slots ? 0u : sl + 1;
          __builtin_amdgcn_s_sleep(16);
        }
      }
      slot = uni(sl);
      // the previous holder may have run on another CU: nothing of its bytes is read here (the
      // bitmap is cleared, heap and list entries are written before they are read), the acquire
      // only keeps this CU's L1 from serving lines it cached during an earlier tenure
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      cand_d = p.cand_d + (size_t)slot * p.cand_cap;
      cand_i = p.cand_id + (size_t)slot * p.cand_cap;
      vis = p.vis_bits + (size_t)slot * p.vis_words;
      ulist = p.ulist + (size_t)slot * p.ulist_cap;
    }

    if (!xresumed) { for (uint64_t i = lane; i < p.vis_words; i += 64) vis[i] = 0u; }
    const float q_norm = load_query<METRIC>(p.queries + (uint64_t)qi * p.d, p.d, qs);
    __threadfence_block();

    uint64_t clen = 0, rlen = 0;  // lane 0 only
    uint32_t status = QS_OK;
    uint64_t payload = 0;
    uint32_t cH = 0, cE = 0, cV = 0, cP = 0;
    uint32_t x_nu = 0, x_keep = 0;
    bool x_have_hop = false, x_parked = false;
    uint32_t* xst = p.xstate ? p.xstate + (size_t)slot * p.xstate_words : nullptr;

    if (xresumed) {
      // the parked hop: candidate heap, bitmap and the hop's unvisited ids are where they were left
      cH = xst[0]; cE = xst[1]; cV = xst[2]; cP = xst[3];
      x_nu = xst[4]; x_keep = xst[5];
      const uint32_t rl = xst[6];
      if (lane == 0) {
        rlen = rl;
        clen = (uint64_t)xst[7] | ((uint64_t)xst[8] << 32);
      }
      for (uint32_t i = lane; i < rl; i += 64) {
        res_d[i] = __uint_as_float(xst[16 + i]);
        res_i[i] = xst[16 + (ef + 1) + i];
      }
      x_have_hop = true;
      __threadfence_block();
      __syncthreads();
    } else
    if ((uint64_t)p.entry >= p.nvec) {
      status = QS_NODE_NOT_FOUND;
      payload = p.entry;
    } else if (!rows_present(p, p.entry, 1)) {
      status = QS_BLOCKED;
    } else {
      uint32_t entry = p.entry;
      const uint32_t erow = row_index(p, entry, true);
      float e_aux = METRIC == METRIC_COSINE_PRE ? p.norm2[erow] : 0.0f;
      float ed = rl_f(exact_rows<METRIC>(p, erow, 1, qs, tile, q_norm, e_aux), 0);
      cV = 1;
      if (HNSW) {
        // greedy search from the top layer down to layer 1, hnsw.rs:478-497: per round the
        // neighbours of the node the round STARTED at are scanned in order, `current` moves to
        // every strictly closer one (= first occurrence of the minimum if it beats current)
        for (uint32_t layer = p.max_level; layer >= 1 && status == QS_OK; --layer) {
          const uint64_t* loff = p.layer_off[layer];
          const uint32_t* ladj = p.layer_adj[layer];
          for (;;) {
            const uint64_t o0 = loff[entry], o1 = loff[entry + 1];
            const uint32_t deg = (uint32_t)(o1 - o0);
            bool changed = false;
            uint32_t cur = entry;
            float cur_d = ed;
            for (uint32_t base = 0; base < deg && status == QS_OK; base += 64) {
              const uint32_t R = deg - base < 64 ? deg - base : 64;
              const uint32_t nid = (uint32_t)lane < R ? ladj[o0 + base + lane] : 0u;
              const uint64_t bad = ballot((uint32_t)lane < R && (uint64_t)nid >= p.nvec);
              if (bad) {  // HnswGraph::distance -> NodeNotFound, hnsw.rs:449-455
                status = QS_NODE_NOT_FOUND;
                payload = rl_u(nid, __ffsll((long long)bad) - 1);
                break;
              }
              float r_aux = (METRIC == METRIC_COSINE_PRE && (uint32_t)lane < R) ? p.norm2[nid] : 0.0f;
              float nd = exact_rows<METRIC>(p, nid, R, qs, tile, q_norm, r_aux);
              cV += R;
              for (uint32_t r = 0; r < R; ++r) {  // in list order, strict `<` (hnsw.rs:485)
                float dr = rl_f(nd, (int)r);
                if (dr < cur_d) { cur = rl_u(nid, (int)r); cur_d = dr; changed = true; }
              }
            }
            entry = cur;
            ed = cur_d;
            if (!changed || status != QS_OK) break;
          }
        }
      }
      if (lane == 0 && status == QS_OK) {
        vis[entry >> 5] |= 1u << (entry & 31);
        heap_push<CandOrd>(cand_d, cand_i, clen, ed, entry);
        uint64_t rl = rlen;
        heap_push<ResOrd>(res_d, res_i, rl, ed, entry);
        rlen = rl;
      }
      cP = 1;
      __threadfence_block();
      __syncthreads();
    }

    while (status == QS_OK) {
      uint32_t nu = 0, keep = 0;
      if (x_have_hop) {  // a parked hop is evaluated before anything is popped
        x_have_hop = false;
        nu = x_nu;
        keep = x_keep;
        goto x_hop_ready;
      }
      {
      // lane 0: candidates.pop() + termination test, leann.rs:922-928
      if (lane == 0) {
        uint32_t go = 0, cid = 0;
        if (clen > 0) {
          float cd;
          heap_pop<CandOrd>(cand_d, cand_i, clen, cd, cid);
          go = 1;
          if (rlen > 0 && rlen >= ef && ordkey(cd) > ordkey(res_d[0])) go = 0;
        }
        ctl[0] = go;
        ctl[1] = cid;
      }
      __syncthreads();
      uint32_t go = ctl[0], cid = ctl[1];
      __syncthreads();
      if (!go) break;
      if ((uint64_t)cid >= p.num_nodes) continue;
      uint64_t o0, o1;
      if (p.ell_w) { o0 = (uint64_t)cid * p.ell_w; o1 = o0 + p.ell_deg[cid]; }
      else { o0 = p.off[cid]; o1 = p.off[cid + 1]; }
      uint32_t deg = (uint32_t)(o1 - o0);
      cH += 1;
      cE += deg;
      // unvisited = neighbors.filter(visited.insert), leann.rs:933-937
      for (uint32_t base = 0; base < deg; base += 64) {
        bool active = base + lane < deg;
        uint32_t nid = active ? p.adj[o0 + base + lane] : 0u;
        bool is_new = false;
        if (active) {
          if (((uint64_t)nid >> 5) < p.vis_words) {
            uint32_t bit = 1u << (nid & 31);
            uint32_t old = atomicOr(&vis[nid >> 5], bit);
            is_new = !(old & bit);
          } else {
            is_new = true;  // beyond every valid id: reported as NodeNotFound below
          }
        }
        uint64_t nm = ballot(is_new);
        uint32_t rank = (uint32_t)__popcll(nm & ((1ull << lane) - 1ull));
        if (is_new) ulist[nu + rank] = nid;
        nu += (uint32_t)__popcll(nm);
      }
      if (nu == 0) continue;
      __threadfence_block();
      __syncthreads();
      uint32_t rl_now = 0;
      if (lane == 0) ctl[2] = (uint32_t)rlen;
      __syncthreads();
      rl_now = ctl[2];
      keep = prune_keep(p.prune_ratio, p.prune_strategy, nu, rl_now, ef);
      // compute_embeddings_batch over all kept ids first, leann.rs:947
      uint32_t first_bad = 0xFFFFFFFFu;
      for (uint32_t base = 0; base < keep && first_bad == 0xFFFFFFFFu; base += 64) {
        uint32_t uid = base + lane < keep ? ulist[base + lane] : 0u;
        uint64_t bad = ballot(base + lane < keep && (uint64_t)uid >= p.nvec);
        if (bad) first_bad = rl_u(uid, __ffsll((long long)bad) - 1);
      }
      if (first_bad != 0xFFFFFFFFu) {
        status = QS_NODE_NOT_FOUND;
        payload = first_bad;
        break;
      }
      }
    x_hop_ready:
      if (p.slot_of) {  // recompute provider: every kept row must be materialised
        bool all_here = true;
        for (uint32_t base = 0; base < keep; base += 64) {
          const uint32_t R = keep - base < 64 ? keep - base : 64;
          const uint32_t uid = (uint32_t)lane < R ? ulist[base + lane] : 0u;
          if (!rows_present(p, uid, R)) all_here = false;
        }
        if (!all_here) {
          status = QS_BLOCKED;
          if (p.xslot) {  // park: the slot stays with the query
            status = QS_BLOCKED_X;
            if (lane == 0) {
              ctl[2] = (uint32_t)rlen;
              xst[0] = cH; xst[1] = cE; xst[2] = cV; xst[3] = cP;
              xst[4] = nu; xst[5] = keep; xst[6] = (uint32_t)rlen;
              xst[7] = (uint32_t)clen; xst[8] = (uint32_t)(clen >> 32);
              p.xslot[qi] = slot + 1u;
            }
            __syncthreads();
            const uint32_t rl = ctl[2];
            for (uint32_t i = lane; i < rl; i += 64) {
              xst[16 + i] = __float_as_uint(res_d[i]);
              xst[16 + (ef + 1) + i] = res_i[i];
            }
            x_parked = true;
          }
          break;
        }
      }
      cV += keep;
      for (uint32_t base = 0; base < keep && status == QS_OK; base += 64) {
        uint32_t R = keep - base < 64 ? keep - base : 64;
        uint32_t uid = (uint32_t)lane < R ? ulist[base + lane] : 0u;
        const uint32_t rix = row_index(p, uid, (uint32_t)lane < R);
        float r_aux = (METRIC == METRIC_COSINE_PRE && (uint32_t)lane < R) ? p.norm2[rix] : 0.0f;
        float nd = exact_rows<METRIC>(p, rix, R, qs, tile, q_norm, r_aux);
        if ((uint32_t)lane < R) {
          dscratch[lane] = nd;
          scratch[lane] = uid;
        }
        __syncthreads();
        if (lane == 0) {
          uint32_t pushes = 0, st = QS_OK;
          for (uint32_t r = 0; r < R; ++r) {  // leann.rs:953-970
            float d = dscratch[r];
            uint32_t id = scratch[r];
            bool should_add = rlen < ef || rlen == 0 || d < res_d[0];
            if (should_add) {
              if (clen >= p.cand_cap) { st = QS_SCRATCH; break; }
              heap_push<CandOrd>(cand_d, cand_i, clen, d, id);
              heap_push<ResOrd>(res_d, res_i, rlen, d, id);
              pushes++;
              if (rlen > ef) {
                float dd;
                uint32_t di;
                heap_pop<ResOrd>(res_d, res_i, rlen, dd, di);
              }
            }
          }
          ctl[3] = pushes;
          ctl[4] = st;
        }
        __syncthreads();
        cP += ctl[3];
        status = ctl[4];
        __syncthreads();
      }
    }

    // results.into_iter() (array order) + stable sort by distance, leann.rs:984-986
    if (lane == 0 && status != QS_OK) ctl[5] = 0;  // (nothing is returned for a failed or parked query)
    if (lane == 0 && status == QS_OK) {
      for (uint64_t i = 1; i < rlen; ++i) {
        float d = res_d[i];
        uint32_t id = res_i[i];
        uint64_t j = i;
        while (j > 0 && d < res_d[j - 1]) {  // partial_cmp == Less only
          res_d[j] = res_d[j - 1];
          res_i[j] = res_i[j - 1];
          j--;
        }
        res_d[j] = d;
        res_i[j] = id;
      }
      ctl[5] = (uint32_t)rlen;
    }
    __syncthreads();
    uint32_t rl = ctl[5];
    uint32_t outn = rl < p.k ? rl : p.k;
    if (status == QS_OK) {
      for (uint32_t e = lane; e < outn; e += 64) {
        p.out_ids[(uint64_t)qi * p.k + e] = (uint64_t)res_i[e];
        p.out_dist[(uint64_t)qi * p.k + e] = res_d[e];
      }
    }
    if (lane == 0) {
      p.status[qi] = status;
      p.payload[qi] = payload;
      p.out_count[qi] = status == QS_OK ? outn : 0u;
      p.ctr[qi * 4 + 0] = cH;
      p.ctr[qi * 4 + 1] = cE;
      p.ctr[qi * 4 + 2] = cV;
      p.ctr[qi * 4 + 3] = cP;
      if (p.xslot && !x_parked) p.xslot[qi] = 0u;  // (a resumed query that came to its end gives the slot up below)
    }
    __syncthreads();
    if (x_parked) slot = 0xFFFFFFFFu;  // the slot belongs to the parked query now, lock held
  }
  if (slot != 0xFFFFFFFFu) {
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (lane == 0) atomicExch(&p.pool_locks[slot], 0u);
  }
}


// ----------------------------------------------------------- two-level search (EXTENSION)
// "Algorithm 2: Two-Level Search with Hybrid Distance" of docs/leann-specification.md:223-275,
// which the reference promises (leann.rs:54-56, :855-857) and does not implement.  The rules
// the pseudo-code leaves open are fixed in oracle/islands_oracle.c (orc_two_level_search), the
// definition this kernel is tested against:
//   * R (<= ef exact results) and AQ (every node that got a PQ distance) are ascending arrays of
//     64-bit keys in LDS: (ordkey(distance) << 32) | (id << 1) | flag, flag = expanded (R) or
//     promoted (AQ).  EQ is implicit: the unexpanded members of R.
//   * per hop: the new neighbours get table_distance (pq.rs:341-348, left fold over the
//     subquantizers, one lane per neighbour) and are merged into AQ; the unpromoted members of
//     the first ceil(a * |AQ|) entries get their exact distance and are merged into R.
//   * only the smallest tl_wcap entries of AQ are kept: an entry that drops out is larger than
//     tl_wcap others for good, so it can only matter when ceil(a * |AQ|) outgrows the window --
//     then the query fails (QS_SCRATCH), it is never answered differently.
constexpr uint32_t kTlPairs = 16;                         // (neighbour, subquantizer) pairs per lane and chunk
constexpr uint32_t kTlLdsWords = kTlPairs * 64 + 64 + 4;   // their table entries in LDS (+ one pad word per row)
__device__ __forceinline__ uint64_t rl_u64(uint64_t v, int lane) {
  return ((uint64_t)rl_u((uint32_t)(v >> 32), lane) << 32) | (uint64_t)rl_u((uint32_t)v, lane);
}

// Merges the keys held by lanes [0, cnt) into the ascending array arr[0, len) in place (room for
// len + 64); returns len + cnt.  Keys are pairwise different above bit 0.
__device__ uint32_t tl_merge(uint64_t* arr, uint32_t len, uint64_t nk, uint32_t cnt, uint64_t* nbuf) {
  const uint32_t lane = threadIdx.x;
  const bool active = lane < cnt;
  uint32_t rank = 0;
  for (uint32_t j = 0; j < cnt; ++j) rank += rl_u64(nk, (int)j) < nk ? 1u : 0u;
  uint32_t lo = 0, hi = active ? len : 0u;  // old entries below this lane's key
  while (ballot(lo < hi)) {
    const uint32_t mid = (lo + hi) >> 1;
    const bool go = lo < hi;
    const uint64_t v = go ? arr[mid] : 0ull;
    if (go) { if (v < nk) lo = mid + 1; else hi = mid; }
  }
  if (active) nbuf[rank] = nk;
  wave_sync();
  // old entries move up by the number of new keys below them, last chunk first: a chunk is read
  // whole before any of it is written, and writes never reach below the chunk being moved
  uint32_t t = cnt;  // new keys not yet known to lie above everything still to be moved
  for (int c = len ? (int)((len - 1) & ~63u) : -1; c >= 0 && t > 0; c -= 64) {
    const uint32_t i = (uint32_t)c + lane;
    const uint64_t x = i < len ? arr[i] : ~0ull;
    const uint64_t first = rl_u64(x, 0);
    uint32_t sh = t, tt = t;
    while (tt > 0) {
      const uint64_t kb = nbuf[tt - 1];
      if (kb < first) break;
      sh -= kb > x ? 1u : 0u;
      tt -= 1;
    }
    wave_sync();
    if (i < len && sh > 0) arr[i + sh] = x;
    wave_sync();
    t = tt;
  }
  if (active) arr[lo + rank] = nk;
  wave_sync();
  return len + cnt;
}

// RESUME = searches over the recompute provider: a query whose promoted rows are not in the row
// cache reports them and parks -- R, the queue window, the visited table and its counters go to its
// state block -- and is taken up again in the promotion it stopped at once the provider has encoded
// them.  (Promotions are flagged in the window only when their rows were there, so the parked
// promotion is found again by the same scan; everything before it in the hop is complete.)
// QH = bf16 rows and a query whose elements are all bf16 values: the query stays bf16 in LDS
// (half the LDS of a long query), like leann_search_fast's instantiation of that name.
// words of one parked query: 16 scalars, R, the window, the visited table
__host__ __device__ inline uint32_t tl_res_entries(uint32_t ef) { return (ef + 63u) / 64u * 64u + 64u; }
__host__ __device__ inline uint32_t tl_state_words_of(uint32_t ef, uint32_t wcap, uint32_t hbits) {
  return 16u + 2u * tl_res_entries(ef) + 2u * (wcap + 64u) + (1u << hbits);
}

template <int METRIC_API, typename ROWT, bool RESUME = false, bool QH = false>
__global__ __launch_bounds__(64) void leann_search_two_level(SearchParams p) {
  static_assert(!QH || (sizeof(ROWT) == 2 && !RESUME), "the bf16 query operand goes with bf16 rows");
  constexpr int METRIC = METRIC_API == ISL_METRIC_COSINE ? METRIC_COSINE_PRE : METRIC_API;
  const ROWT* const emb = reinterpret_cast<const ROWT*>(p.emb);
  extern __shared__ __align__(16) unsigned char smem[];
  const uint32_t lane = threadIdx.x;
  const uint32_t hcap = p.hcap;
  const uint32_t hlimit = hcap - hcap / 8;
  const uint32_t ef = p.ef;
  const uint32_t wcap = p.tl_wcap;
  const uint32_t resn = tl_res_entries(ef);
  uint32_t* htab = reinterpret_cast<uint32_t*>(smem);
  uint64_t* win = reinterpret_cast<uint64_t*>(htab + hcap);  // hcap * 4 is a multiple of 8
  uint64_t* res = win + (wcap + 64);
  uint64_t* nbuf = res + resn;
  uint32_t* scratch = reinterpret_cast<uint32_t*>(nbuf + 64);  // 64 ids + 64 window positions
  float* tlv = reinterpret_cast<float*>(scratch + 128);  // table entries of one chunk of pairs, rows of m + 1
  float* qs = tlv + kTlLdsWords;
  const uint32_t ocap = 1u << p.obits;
  const uint32_t omask = ocap - 1;
  const uint32_t olimit = ocap - ocap / 4;
  uint32_t* otab = p.otab + (size_t)blockIdx.x * ocap;
  const uint32_t m = p.tl_m, K = p.tl_K;

  for (;;) {
    uint32_t qi = 0;
    if constexpr (QH) {
      if (lane == 0) qi = atomicAdd(&p.ticket[4], 1u);
      qi = uni(qi);
      if (qi >= *((volatile uint32_t*)&p.ticket[7])) break;
      qi = p.qsel_h[qi];
    } else {
      bool listed = false;
      if constexpr (sizeof(ROWT) == 2 && !RESUME) listed = p.qsel_mode != 0u;
      if (listed) {  // the queries the bf16-query launch passed on
        if (lane == 0) qi = atomicAdd(&p.ticket[6], 1u);
        qi = uni(qi);
        if (qi >= *((volatile uint32_t*)&p.ticket[5])) break;
        qi = p.qsel[qi];
      } else {
        if (lane == 0) qi = atomicAdd(&p.ticket[0], 1u);
        qi = uni(qi);
        if (qi >= p.nq) break;
        if (p.qlist) qi = p.qlist[qi];  // a round of the recompute provider / the queries of a retry
      }
    }
    uint32_t* qst = nullptr;
    bool resumed = false;
    if constexpr (RESUME) {
      qst = p.qstate + (size_t)qi * p.qstate_words;
      resumed = uni(p.qflag[qi]) == 1u;
      otab = p.otab + (size_t)qi * ocap;  // a parked query may come back on another wave
    }

    if (resumed) { for (uint32_t i = lane; i < hcap; i += 64) htab[i] = qst[16 + 2 * resn + 2 * (wcap + 64) + i]; }
    else { for (uint32_t i = lane; i < hcap; i += 64) htab[i] = EMPTY; }
    float q_norm;
    if constexpr (QH) {
      bool representable;
      q_norm = load_query_bf16<METRIC>(p.queries + (uint64_t)qi * p.d, p.d, qs, &representable);  // syncs
      if (!representable) {  // (cannot happen after classify_queries_kernel; kept as a guard)
        if (lane == 0) p.qsel[atomicAdd(&p.ticket[5], 1u)] = qi;
        continue;
      }
    } else {
      q_norm = load_query<METRIC>(p.queries + (uint64_t)qi * p.d, p.d, qs);  // syncs
    }
    const float* tables = p.tl_tables + (uint64_t)qi * m * K;

    uint32_t rlen = 0, wlen = 0, aq_total = 0;
    uint32_t hcount = 0, ocount = 0;
    bool ovf = false;
    uint32_t status = QS_OK;
    uint64_t payload = 0;
    uint32_t cH = 0, cE = 0, cV = 0, cP = 0;
    bool parked = false;
    bool in_promotion = false;  // RESUME: the parked hop's promotions come before anything is popped

    if (resumed) {
      rlen = qst[0]; wlen = qst[1]; aq_total = qst[2]; hcount = qst[3]; ocount = qst[4]; ovf = qst[5] != 0u;
      cH = qst[6]; cE = qst[7]; cV = qst[8]; cP = qst[9];
      const uint64_t* sres = reinterpret_cast<const uint64_t*>(qst + 16);
      const uint64_t* swin = sres + resn;
      for (uint32_t i = lane; i < rlen; i += 64) res[i] = sres[i];
      for (uint32_t i = lane; i < wlen; i += 64) win[i] = swin[i];
      in_promotion = true;
      wave_sync();
    } else
    if ((uint64_t)p.entry >= p.nvec) {  // provider.compute_embedding(entry), leann.rs:911
      status = QS_NODE_NOT_FOUND;
      payload = p.entry;
    } else if (!rows_present(p, p.entry, 1)) {
      status = QS_BLOCKED;  // nothing to park: the query starts over once the entry's row is there
    } else {
      const uint32_t entry = p.entry;
      const uint32_t erow = row_index(p, entry, true);
      const float e_aux = METRIC == METRIC_COSINE_PRE ? p.norm2[erow] : 0.0f;
      float ed = direct_distances<METRIC, ROWT, QH>(emb, p.stride, p.d, erow, 1, qs, q_norm, e_aux);
      ed = rl_f(ed, 0);
      cV = 1;
      if (lane == 0) {
        htab[hslot_cap(entry, hcap)] = entry;
        res[0] = ((uint64_t)ordkey(ed) << 32) | ((uint64_t)entry << 1);
      }
      hcount = 1;
      rlen = 1;
      wave_sync();
    }

    while (status == QS_OK) {
     if (!in_promotion) {
      // extract_min(EQ): the first unexpanded member of R; none left -> done (lines 5-9)
      uint32_t e = 0xFFFFFFFFu;
      for (uint32_t c = 0; c < rlen; c += 64) {
        const uint32_t i = c + lane;
        const uint64_t x = i < rlen ? res[i] : ~0ull;
        const uint64_t um = ballot(i < rlen && !(x & 1ull));
        if (um) { e = c + (uint32_t)__ffsll((long long)um) - 1u; break; }
      }
      if (e == 0xFFFFFFFFu) break;
      const uint64_t ekey = res[e];
      wave_sync();
      if (lane == 0) res[e] = ekey | 1ull;
      wave_sync();
      const uint32_t cid = (uint32_t)(ekey >> 1) & ID_MASK;
      if ((uint64_t)cid >= p.num_nodes) continue;  // get_neighbors -> None, leann.rs:227-229
      uint64_t o0;
      uint32_t deg;
      if (p.ell_w) { o0 = (uint64_t)cid * p.ell_w; deg = p.ell_deg[cid]; }
      else { o0 = p.off[cid]; deg = (uint32_t)(p.off[cid + 1] - o0); }
      cH += 1;
      cE += deg;

      // Phase 1 (lines 12-16): approximate distances of the unvisited neighbours, 64 at a time
      for (uint32_t base = 0; base < deg && status == QS_OK; base += 64) {
        const bool active = base + lane < deg;
        const uint32_t nid = active ? p.adj[o0 + base + lane] : EMPTY;
        const uint32_t batch = deg - base < 64 ? deg - base : 64;
        if (!ovf && hcount + batch > hlimit) ovf = true;
        const bool is_new = visited_insert(htab, hcap, ovf, otab, p.obits, omask, nid, active);
        const uint64_t nm = ballot(is_new);
        const uint32_t nu = (uint32_t)__popcll(nm);
        if (!ovf) hcount += nu;
        else {
          ocount += nu;
          if (ocount > olimit) { status = QS_SCRATCH; payload = 2; break; }
        }
        if (nu == 0) continue;
        const uint32_t rank = (uint32_t)__popcll(nm & ((1ull << lane) - 1ull));
        if (is_new) scratch[rank] = nid;
        wave_sync();
        const uint32_t uid = lane < nu ? scratch[lane] : 0u;
        wave_sync();
        const uint64_t bad = ballot(lane < nu && (uint64_t)uid >= p.tl_ncodes);
        if (bad) {
          status = QS_NODE_NOT_FOUND;
          payload = rl_u(uid, __ffsll((long long)bad) - 1);
          break;
        }
        // table_distance, pq.rs:341-348: s = sum over the subquantizers j of tables[j][code_j], a left
        // fold in j.  The nu * m (neighbour, subquantizer) pairs are spread over the lanes -- lane order
        // = j fastest, so a neighbour's codes are one contiguous run -- and fetched kTlPairs per lane at a
        // time: all code loads of a chunk are in flight together, then all table reads (two dependent
        // round trips per chunk of up to 1024 pairs, where one lane per neighbour made 2 * m / 8 of
        // them).  The entries go through LDS (rows padded by one word: lanes that fold different
        // neighbours hit different banks) and lane n folds neighbour n's entries in subquantizer
        // order: the same additions in the same order.
        float s = 0.0f;
        {
          const uint32_t per_chunk = m <= kTlPairs * 64u ? (kTlPairs * 64u) / m : 0u;  // neighbours per chunk
          if (per_chunk == 0) {  // a row of codes longer than a chunk (m > 1024): one lane per neighbour
            if (lane < nu) {
              const uint16_t* cr = p.tl_codes + (uint64_t)uid * m;
              for (uint32_t j = 0; j < m; ++j) s += tables[(uint64_t)j * K + cr[j]];
            }
          } else {
            for (uint32_t n0 = 0; n0 < nu; n0 += per_chunk) {
              const uint32_t cnt = nu - n0 < per_chunk ? nu - n0 : per_chunk;
              const uint32_t total = cnt * m;
              uint32_t cn[kTlPairs], cj[kTlPairs];
              uint32_t code[kTlPairs];
              // pair t = lane + 64 * i  ->  neighbour t / m, subquantizer t % m, kept incrementally
              uint32_t n = lane / m, j = lane - n * m;
#pragma unroll
              for (uint32_t i = 0; i < kTlPairs; ++i) {
                cn[i] = n;
                cj[i] = j;
                j += 64u;
                while (j >= m) { j -= m; n += 1u; }
              }
              // (pairs past the chunk's end load pair (0, 0) again: an `if` around a load makes the compiler
              // branch and wait per element -- 32 dependent round trips instead of 2)
#pragma unroll
              for (uint32_t i = 0; i < kTlPairs; ++i) {
                const bool ok = lane + 64u * i < total;
                cn[i] = ok ? cn[i] : 0u;
                cj[i] = ok ? cj[i] : 0u;
              }
#pragma unroll
              for (uint32_t i = 0; i < kTlPairs; ++i) code[i] = p.tl_codes[(uint64_t)scratch[n0 + cn[i]] * m + cj[i]];
              float tv[kTlPairs];
#pragma unroll
              for (uint32_t i = 0; i < kTlPairs; ++i) tv[i] = tables[(uint64_t)cj[i] * K + code[i]];
#pragma unroll
              for (uint32_t i = 0; i < kTlPairs; ++i) {
                const uint32_t t = lane + 64u * i;
                if (t < total) tlv[cn[i] * (m + 1u) + cj[i]] = tv[i];
              }
              wave_sync();
              if (lane >= n0 && lane < n0 + cnt) {
                // sixteen entries read together, then added in order: one LDS latency per sixteen additions
                // (an entry read and waited for per addition made the fold the longest part of the hop)
                const float* row = tlv + (lane - n0) * (m + 1u);
                uint32_t jj = 0;
                for (; jj + 16u <= m; jj += 16u) {
                  float v[16];
#pragma unroll
                  for (int e = 0; e < 16; ++e) v[e] = row[jj + e];
#pragma unroll
                  for (int e = 0; e < 16; ++e) s += v[e];
                }
                for (; jj < m; ++jj) s += row[jj];
              }
              wave_sync();
            }
          }
        }
        const float ad = sqrtf(s);
        const uint64_t key = ((uint64_t)ordkey(ad) << 32) | ((uint64_t)uid << 1);
        cP += nu;
        aq_total += nu;
        wlen = tl_merge(win, wlen, key, nu, nbuf);
        if (wlen > wcap) wlen = wcap;
      }
      if (status != QS_OK || aq_total == 0) continue;
     }
     in_promotion = false;

      // Phase 2 (lines 19-27): M = the first ceil(a * |AQ|) entries of AQ, at least one
      const float tf = ceilf(p.tl_ratio * (float)aq_total);
      uint32_t ntop = tf >= 1.0f ? (tf >= (float)aq_total ? aq_total : (uint32_t)tf) : 1u;
      if (ntop > aq_total) ntop = aq_total;
      if (ntop > wlen) { status = QS_SCRATCH; payload = 7; break; }
      // The unpromoted members of M are collected over the prefix in groups of up to 64 (a chunk of
      // 64 window positions never straddles two groups), so that a group's rows are fetched in one
      // distance pass and merged into R at once.  A group is flagged "promoted" only once its rows
      // were there: a parked promotion is found again by the same scan.
      uint32_t c = 0;
      while (c < ntop && status == QS_OK) {
        uint32_t pend = 0;
        for (; c < ntop; c += 64) {
          const uint32_t i = c + lane;
          const uint64_t x = i < ntop ? win[i] : ~0ull;
          const bool un = i < ntop && !(x & 1ull);
          const uint64_t um = ballot(un);
          const uint32_t pc = (uint32_t)__popcll(um);
          if (!pc) continue;
          if (pend + pc > 64) break;  // this chunk opens the next group
          const uint32_t rank = (uint32_t)__popcll(um & ((1ull << lane) - 1ull));
          if (un) {
            scratch[pend + rank] = (uint32_t)(x >> 1) & ID_MASK;
            scratch[64 + pend + rank] = i;
          }
          pend += pc;
        }
        if (!pend) break;
        const uint32_t pc = pend;
        wave_sync();
        const uint32_t pid = lane < pc ? scratch[lane] : 0u;
        const uint32_t ppos = lane < pc ? scratch[64 + lane] : 0u;
        wave_sync();
        const uint64_t bad = ballot(lane < pc && (uint64_t)pid >= p.nvec);
        if (bad) {
          status = QS_NODE_NOT_FOUND;
          payload = rl_u(pid, __ffsll((long long)bad) - 1);
          break;
        }
        if (!rows_present(p, pid, pc)) {
          status = QS_BLOCKED;
          if constexpr (RESUME) {  // park: everything the rest of the search depends on
            if (lane == 0) {
              qst[0] = rlen; qst[1] = wlen; qst[2] = aq_total; qst[3] = hcount; qst[4] = ocount; qst[5] = ovf ? 1u : 0u;
              qst[6] = cH; qst[7] = cE; qst[8] = cV; qst[9] = cP;
              p.qflag[qi] = 1u;
            }
            uint64_t* sres = reinterpret_cast<uint64_t*>(qst + 16);
            uint64_t* swin = sres + resn;
            for (uint32_t i = lane; i < rlen; i += 64) sres[i] = res[i];
            for (uint32_t i = lane; i < wlen; i += 64) swin[i] = win[i];
            for (uint32_t i = lane; i < hcap; i += 64) qst[16 + 2 * resn + 2 * (wcap + 64) + i] = htab[i];
            parked = true;
          }
          if constexpr (RESUME) {
            if (p.tl_prefetch) {  // what this query will most likely ask for next: the queue's entries right behind the prefix
              const uint32_t i = ntop + lane;
              const uint64_t x = i < wlen ? win[i] : ~0ull;
              const bool un = i < wlen && !(x & 1ull);
              const uint64_t um = ballot(un);
              const uint32_t r0 = (uint32_t)__popcll(um & ((1ull << lane) - 1ull));
              const uint32_t fid = (uint32_t)(x >> 1) & ID_MASK;
              const bool want = un && r0 < p.tl_prefetch && (uint64_t)fid < p.nvec && p.slot_of[fid] == kNoSlot;
              const uint64_t wm = ballot(want);
              if (wm) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&p.ticket[14], (uint32_t)__popcll(wm));
                base = uni(base);
                const uint32_t r1 = (uint32_t)__popcll(wm & ((1ull << lane) - 1ull));
                if (want && base + r1 < p.pref_cap) p.pref[base + r1] = fid;
              }
            }
          }
          break;
        }
        if (lane < pc) win[ppos] |= 1ull;
        cV += pc;
        const uint32_t prow = row_index(p, pid, lane < pc);
        const float r_aux = (METRIC == METRIC_COSINE_PRE && lane < pc) ? p.norm2[prow] : 0.0f;
        const float nd = direct_distances<METRIC, ROWT, QH>(emb, p.stride, p.d, prow, pc, qs, q_norm, r_aux);
        const uint64_t rkey = ((uint64_t)ordkey(nd) << 32) | ((uint64_t)pid << 1);
        rlen = tl_merge(res, rlen, rkey, pc, nbuf);
        if (rlen > ef) rlen = ef;  // lines 26-27
      }
    }

    const uint32_t outn = rlen < p.k ? rlen : p.k;
    if (status == QS_OK) {
      for (uint32_t e = lane; e < outn; e += 64) {
        const uint64_t x = res[e];
        p.out_ids[(uint64_t)qi * p.k + e] = (uint64_t)((uint32_t)(x >> 1) & ID_MASK);
        p.out_dist[(uint64_t)qi * p.k + e] = key_to_dist((uint32_t)(x >> 32));
      }
    }
    if (lane == 0) {
      p.status[qi] = status;
      p.payload[qi] = payload;
      p.out_count[qi] = status == QS_OK ? outn : 0u;
      p.ctr[qi * 4 + 0] = cH;
      p.ctr[qi * 4 + 1] = cE;
      p.ctr[qi * 4 + 2] = cV;
      p.ctr[qi * 4 + 3] = cP;
      if constexpr (RESUME) {
        if (!parked) p.qflag[qi] = 0u;  // finished, failed, or blocked on its entry point: a fresh start next time
      }
    }
    if (ovf && !parked) {
      for (uint32_t i = lane; i < ocap; i += 64) otab[i] = EMPTY;
    }
    wave_sync();
  }
}

// ------------------------------------------------------------------ launchers

template <typename K>
void launch_one(K kernel, uint32_t grid, size_t lds, hipStream_t st, const SearchParams& p) {
  // more than 64 KiB of dynamic LDS needs the opt-in attribute (set once per kernel and size)
  if (lds > 64 * 1024) {
    static std::mutex mu;
    static std::vector<std::pair<const void*, size_t>> done;
    const void* fn = reinterpret_cast<const void*>(kernel);
    std::lock_guard<std::mutex> lock(mu);
    bool have = false;
    for (auto& e : done) have |= e.first == fn && e.second >= lds;
    if (!have) {
      (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      done.emplace_back(fn, lds);
    }
  }
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(64), lds, st, p);
}

}  // namespace

// Launchers defined in the instantiating units; `params` points at a SearchParams (the struct is
// the same text in every unit).
namespace isl_launch {
// resume = the RESUME instantiation (recompute provider; f32 rows only); qh = the QH instantiation
// (bf16 rows of up to 64 ids per adjacency row, query held as bf16 in LDS)
void launch_fast_s1(int metric, bool wide, bool bf16, bool resume, bool qh, uint32_t grid, size_t lds, hipStream_t st, const void* params);
void launch_fast_s2(int metric, bool wide, bool bf16, bool resume, bool qh, uint32_t grid, size_t lds, hipStream_t st, const void* params);
void launch_fast_s4(int metric, bool wide, bool bf16, bool resume, bool qh, uint32_t grid, size_t lds, hipStream_t st, const void* params);
void launch_fast_s8(int metric, bool wide, bool bf16, bool resume, bool qh, uint32_t grid, size_t lds, hipStream_t st, const void* params);
// words of one parked query (RESUME) for result sets of S x 64 entries and a visited table of 1 << hbits
inline uint32_t fast_state_words(int S, uint32_t hbits) { return 208u + (uint32_t)S * 128u + (1u << hbits); }
void launch_exact(int metric, bool hnsw, uint32_t grid, size_t lds, hipStream_t st, const void* params);
// resume = the RESUME instantiation (recompute provider, f32 rows); qh = bf16 rows, query held as bf16 in LDS
void launch_two_level(int metric, bool bf16, bool resume, bool qh, uint32_t grid, size_t lds, hipStream_t st, const void* params);
inline uint32_t tl_state_words(uint32_t ef, uint32_t wcap, uint32_t hbits) { return tl_state_words_of(ef, wcap, hbits); }
void launch_descent(int metric, uint32_t grid, size_t lds, hipStream_t st, const void* params);
void launch_classify(uint32_t grid, hipStream_t st, const void* params);
}  // namespace isl_launch
