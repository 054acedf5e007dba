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
  QS_BLOCKED = 0x103   // recompute provider: a needed row is not materialised yet (ids reported)
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
  uint2* plog;       // [nq][plog_cap] (distance bits, id) of every results.push, in order
  uint32_t plog_cap;
  uint32_t hbits;    // LDS visited table: 1 << hbits entries
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
  // graph under construction (build.hip): row i = adj[i * ell_w .. + ell_deg[i]), `off` unused
  uint32_t ell_w;
  const uint32_t* ell_deg;
  // recompute provider: rows exist where `present` has a bit; a query that needs an absent row
  // appends the id to `miss` (count in ticket[13]) and stops with QS_BLOCKED
  const uint32_t* present;
  uint32_t* miss;
  uint32_t miss_cap;
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

template <int S>
struct RSet {
  uint32_t kd[S];
  uint32_t id[S];
  uint32_t len;  // wave-uniform

  __device__ void init() {
#pragma unroll
    for (int s = 0; s < S; ++s) { kd[s] = KEY_MAX; id[s] = KEY_MAX; }
    len = 0;
  }
  __device__ uint32_t key_at(uint32_t e) const {
    uint32_t r = 0;
#pragma unroll
    for (int s = 0; s < S; ++s)
      if ((int)(e >> 6) == s) r = rl_u(kd[s], e & 63);
    return r;
  }
  __device__ uint32_t id_at(uint32_t e) const {
    uint32_t r = 0;
#pragma unroll
    for (int s = 0; s < S; ++s)
      if ((int)(e >> 6) == s) r = rl_u(id[s], e & 63);
    return r;
  }
  // first entry not yet expanded, or 0xFFFFFFFF
  __device__ uint32_t first_unexpanded() const {
#pragma unroll
    for (int s = 0; s < S; ++s) {
      uint64_t m = ballot(!(id[s] & FLAG_EXP));
      if (m) return s * 64 + (uint32_t)__ffsll((long long)m) - 1;
    }
    return 0xFFFFFFFFu;
  }
  __device__ void mark_expanded(uint32_t e) {
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
__device__ __forceinline__ bool rows_present(const SearchParams& p, uint32_t uid, uint32_t n) {
  if (!p.present) return true;
  const uint32_t lane = threadIdx.x;
  const bool absent = lane < n && !((p.present[uid >> 5] >> (uid & 31)) & 1u);
  const uint64_t am = ballot(absent);
  if (!am) return true;
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(&p.ticket[13], (uint32_t)__popcll(am));
  base = uni(base);
  const uint32_t rank = (uint32_t)__popcll(am & ((1ull << lane) - 1ull));
  if (absent && base + rank < p.miss_cap) p.miss[base + rank] = uid;
  return false;
}

// ------------------------------------------------------------------ fast kernel
template <int S, int METRIC_API, typename ROWT>
__global__ __launch_bounds__(64) void leann_search_fast(SearchParams p) {
  constexpr int METRIC = METRIC_API == ISL_METRIC_COSINE ? METRIC_COSINE_PRE : METRIC_API;
  const ROWT* const emb = reinterpret_cast<const ROWT*>(p.emb);
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x;
  const uint32_t hcap = 1u << p.hbits;
  const uint32_t hmask = hcap - 1;
  const uint32_t hlimit = hcap - hcap / 8;  // load factor 0.875
  uint32_t* htab = reinterpret_cast<uint32_t*>(smem);
  // merge buffer of batch_insert; its head doubles as the 64 words of the id compaction
  uint32_t* mbuf = htab + hcap;
  uint32_t* scratch = mbuf;
  float* qs = reinterpret_cast<float*>(mbuf + 2 * (p.ef + kBatchMax));
  const uint32_t ocap = 1u << p.obits;
  const uint32_t omask = ocap - 1;
  const uint32_t olimit = ocap - ocap / 4;
  uint32_t* otab = p.otab + (size_t)blockIdx.x * ocap;
  const uint32_t ef = p.ef;

  for (;;) {
    uint32_t qi = 0;
    if (lane == 0) qi = atomicAdd(&p.ticket[0], 1u);
    qi = uni(qi);
    if (qi >= p.nq) break;

    if (p.q_entry && p.status[qi] != QS_OK) {  // the greedy descent already failed this query
      if (lane == 0) {
        p.out_count[qi] = 0;
        p.ctr[qi * 4 + 0] = 0; p.ctr[qi * 4 + 1] = 0; p.ctr[qi * 4 + 2] = p.q_evals[qi]; p.ctr[qi * 4 + 3] = 0;
      }
      continue;
    }
    const uint64_t t_start = __builtin_amdgcn_s_memrealtime();
    for (uint32_t i = lane; i < hcap; i += 64) htab[i] = EMPTY;
    const float q_norm = load_query<METRIC>(p.queries + (uint64_t)qi * p.d, p.d, qs);  // syncs

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

    // entry point: provider.compute_embedding(entry) + distance, leann.rs:911-916
    if (!p.q_entry && (uint64_t)p.entry >= p.nvec) {
      status = QS_NODE_NOT_FOUND;
      payload = p.entry;
    } else if (!rows_present(p, p.entry, 1)) {
      status = QS_BLOCKED;
    } else {
      // HnswGraph: the layer-0 search starts where the greedy descent (hnsw_descent_kernel) ended
      const uint32_t entry = p.q_entry ? p.q_entry[qi] : p.entry;
      float e_aux = METRIC == METRIC_COSINE_PRE ? p.norm2[entry] : 0.0f;
      float ed = direct_distances<METRIC, ROWT>(emb, p.stride, p.d, entry, 1, qs, q_norm, e_aux);
      ed = rl_f(ed, 0);
      cV = p.q_entry ? p.q_evals[qi] : 1;
      if (lane == 0) htab[hslot(entry, p.hbits)] = entry;
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
      uint32_t deg, nid;
      if (p.ell_w) {
        const uint32_t slot = (uint32_t)lane < p.ell_w ? (uint32_t)lane : p.ell_w - 1u;
        deg = p.ell_deg[cid];
        const uint32_t raw = p.adj[(uint64_t)cid * p.ell_w + slot];
        nid = (uint32_t)lane < deg ? raw : EMPTY;
      } else {
        const uint64_t o0 = p.off[cid], o1 = p.off[cid + 1];
        deg = (uint32_t)(o1 - o0);
        nid = ((uint32_t)lane < deg && deg <= 64) ? p.adj[o0 + lane] : EMPTY;
      }
      cH += 1;
      cE += deg;
      if (deg == 0) continue;
      if (deg > 64) { status = QS_REDO; payload = 1; break; }  // long rows: exact kernel
      bool active = (uint32_t)lane < deg;

      ISL_MARK(tp0)  // selection + adjacency fetch
      // visited.insert(n), leann.rs:933-937 (rows hold no duplicate ids on the device)
      if (!ovf && hcount + deg > hlimit) ovf = true;
      bool is_new = false;
      if (active) {
        uint32_t h = hslot(nid, p.hbits);
        if (!ovf) {
          for (;;) {
            uint32_t old = atomicCAS(&htab[h], EMPTY, nid);
            if (old == EMPTY) { is_new = true; break; }
            if (old == nid) break;
            h = (h + 1) & hmask;
          }
        } else {
          bool found = false;
          for (;;) {
            uint32_t cur = htab[h];
            if (cur == nid) { found = true; break; }
            if (cur == EMPTY) break;
            h = (h + 1) & hmask;
          }
          if (!found) {
            uint32_t g = hslot(nid, p.obits);
            for (;;) {
              uint32_t old = atomicCAS(&otab[g], EMPTY, nid);
              if (old == EMPTY) { is_new = true; break; }
              if (old == nid) break;
              g = (g + 1) & omask;
            }
          }
        }
      }
      uint64_t nm = ballot(is_new);
      uint32_t nu = (uint32_t)__popcll(nm);
      if (!ovf) hcount += nu;
      else {
        ocount += nu;
        if (ocount > olimit) { status = QS_REDO; payload = 2; break; }
      }
      if (nu == 0) continue;  // leann.rs:939-941

      // compact the unvisited ids, CSR order preserved
      uint32_t rank = (uint32_t)__popcll(nm & ((1ull << lane) - 1ull));
      if (is_new) scratch[rank] = nid;
      wave_sync();
      uint32_t uid = (uint32_t)lane < nu ? scratch[lane] : 0u;
      wave_sync();

      uint32_t keep = prune_keep(p.prune_ratio, p.prune_strategy, nu, rs.len, ef);  // :944
      // compute_embeddings_batch, leann.rs:947: the first missing id fails the query
      uint64_t bad = ballot((uint32_t)lane < keep && (uint64_t)uid >= p.nvec);
      if (bad) {
        int bl = __ffsll((long long)bad) - 1;
        status = QS_NODE_NOT_FOUND;
        payload = rl_u(uid, bl);
        break;
      }
      if (!rows_present(p, uid, keep)) { status = QS_BLOCKED; break; }
      cV += keep;
      ngroups += (keep + 15) / 16;
      nhops_rows += 1;
      ISL_MARK(tp1)  // visited set + compaction
      float r_aux = (METRIC == METRIC_COSINE_PRE && (uint32_t)lane < keep) ? p.norm2[uid] : 0.0f;
      float nd = direct_distances<METRIC, ROWT>(emb, p.stride, p.d, uid, keep, qs, q_norm, r_aux);
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
    if (ovf) {  // leave the overflow table empty for the next query of this slot
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

    for (uint64_t i = lane; i < p.vis_words; i += 64) vis[i] = 0u;
    const float q_norm = load_query<METRIC>(p.queries + (uint64_t)qi * p.d, p.d, qs);
    __threadfence_block();

    uint64_t clen = 0, rlen = 0;  // lane 0 only
    uint32_t status = QS_OK;
    uint64_t payload = 0;
    uint32_t cH = 0, cE = 0, cV = 0, cP = 0;

    if ((uint64_t)p.entry >= p.nvec) {
      status = QS_NODE_NOT_FOUND;
      payload = p.entry;
    } else if (!rows_present(p, p.entry, 1)) {
      status = QS_BLOCKED;
    } else {
      uint32_t entry = p.entry;
      float e_aux = METRIC == METRIC_COSINE_PRE ? p.norm2[entry] : 0.0f;
      float ed = rl_f(exact_rows<METRIC>(p, entry, 1, qs, tile, q_norm, e_aux), 0);
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
      uint32_t nu = 0;
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
      uint32_t keep = prune_keep(p.prune_ratio, p.prune_strategy, nu, rl_now, ef);
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
      if (p.present) {  // recompute provider: every kept row must be materialised
        bool all_here = true;
        for (uint32_t base = 0; base < keep; base += 64) {
          const uint32_t R = keep - base < 64 ? keep - base : 64;
          const uint32_t uid = (uint32_t)lane < R ? ulist[base + lane] : 0u;
          if (!rows_present(p, uid, R)) all_here = false;
        }
        if (!all_here) { status = QS_BLOCKED; break; }
      }
      cV += keep;
      for (uint32_t base = 0; base < keep && status == QS_OK; base += 64) {
        uint32_t R = keep - base < 64 ? keep - base : 64;
        uint32_t uid = (uint32_t)lane < R ? ulist[base + lane] : 0u;
        float r_aux = (METRIC == METRIC_COSINE_PRE && (uint32_t)lane < R) ? p.norm2[uid] : 0.0f;
        float nd = exact_rows<METRIC>(p, uid, R, qs, tile, q_norm, r_aux);
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
    if (lane == 0) {
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
    }
    __syncthreads();
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

template <int METRIC_API, typename ROWT>
__global__ __launch_bounds__(64) void leann_search_two_level(SearchParams p) {
  constexpr int METRIC = METRIC_API == ISL_METRIC_COSINE ? METRIC_COSINE_PRE : METRIC_API;
  const ROWT* const emb = reinterpret_cast<const ROWT*>(p.emb);
  extern __shared__ __align__(16) unsigned char smem[];
  const uint32_t lane = threadIdx.x;
  const uint32_t hcap = 1u << p.hbits;
  const uint32_t hmask = hcap - 1;
  const uint32_t hlimit = hcap - hcap / 8;
  const uint32_t ef = p.ef;
  const uint32_t wcap = p.tl_wcap;
  uint32_t* htab = reinterpret_cast<uint32_t*>(smem);
  uint64_t* win = reinterpret_cast<uint64_t*>(htab + hcap);  // hcap * 4 is a multiple of 8
  uint64_t* res = win + (wcap + 64);
  uint64_t* nbuf = res + ((ef + 63) / 64 * 64 + 64);
  uint32_t* scratch = reinterpret_cast<uint32_t*>(nbuf + 64);
  float* qs = reinterpret_cast<float*>(scratch + 64);
  const uint32_t ocap = 1u << p.obits;
  const uint32_t omask = ocap - 1;
  const uint32_t olimit = ocap - ocap / 4;
  uint32_t* otab = p.otab + (size_t)blockIdx.x * ocap;
  const uint32_t m = p.tl_m, K = p.tl_K;

  for (;;) {
    uint32_t qi = 0;
    if (lane == 0) qi = atomicAdd(&p.ticket[0], 1u);
    qi = uni(qi);
    if (qi >= p.nq) break;

    for (uint32_t i = lane; i < hcap; i += 64) htab[i] = EMPTY;
    const float q_norm = load_query<METRIC>(p.queries + (uint64_t)qi * p.d, p.d, qs);  // syncs
    const float* tables = p.tl_tables + (uint64_t)qi * m * K;

    uint32_t rlen = 0, wlen = 0, aq_total = 0;
    uint32_t hcount = 0, ocount = 0;
    bool ovf = false;
    uint32_t status = QS_OK;
    uint64_t payload = 0;
    uint32_t cH = 0, cE = 0, cV = 0, cP = 0;

    if ((uint64_t)p.entry >= p.nvec) {  // provider.compute_embedding(entry), leann.rs:911
      status = QS_NODE_NOT_FOUND;
      payload = p.entry;
    } else if (!rows_present(p, p.entry, 1)) {
      status = QS_BLOCKED;
    } else {
      const uint32_t entry = p.entry;
      const float e_aux = METRIC == METRIC_COSINE_PRE ? p.norm2[entry] : 0.0f;
      float ed = direct_distances<METRIC, ROWT>(emb, p.stride, p.d, entry, 1, qs, q_norm, e_aux);
      ed = rl_f(ed, 0);
      cV = 1;
      if (lane == 0) {
        htab[hslot(entry, p.hbits)] = entry;
        res[0] = ((uint64_t)ordkey(ed) << 32) | ((uint64_t)entry << 1);
      }
      hcount = 1;
      rlen = 1;
      wave_sync();
    }

    while (status == QS_OK) {
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
        bool is_new = false;
        if (active) {
          uint32_t h = hslot(nid, p.hbits);
          if (!ovf) {
            for (;;) {
              uint32_t old = atomicCAS(&htab[h], EMPTY, nid);
              if (old == EMPTY) { is_new = true; break; }
              if (old == nid) break;
              h = (h + 1) & hmask;
            }
          } else {
            bool found = false;
            for (;;) {
              uint32_t cur = htab[h];
              if (cur == nid) { found = true; break; }
              if (cur == EMPTY) break;
              h = (h + 1) & hmask;
            }
            if (!found) {
              uint32_t g = hslot(nid, p.obits);
              for (;;) {
                uint32_t old = atomicCAS(&otab[g], EMPTY, nid);
                if (old == EMPTY) { is_new = true; break; }
                if (old == nid) break;
                g = (g + 1) & omask;
              }
            }
          }
        }
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
        // table_distance, pq.rs:341-348: left fold over the subquantizers.  Eight codes per 16-byte
        // load, their eight table entries fetched together, then added in order.
        float s = 0.0f;
        if (lane < nu) {
          const uint16_t* cr = p.tl_codes + (uint64_t)uid * m;
          if ((m & 7u) == 0) {
            const uint4* cv = reinterpret_cast<const uint4*>(cr);
#pragma unroll 2
            for (uint32_t j0 = 0; j0 < m; j0 += 8) {
              const uint4 c8 = cv[j0 >> 3];
              const float* tb = tables + (uint64_t)j0 * K;
              const float t0 = tb[c8.x & 0xFFFFu], t1 = tb[K + (c8.x >> 16)];
              const float t2 = tb[2 * K + (c8.y & 0xFFFFu)], t3 = tb[3 * K + (c8.y >> 16)];
              const float t4 = tb[4 * K + (c8.z & 0xFFFFu)], t5 = tb[5 * K + (c8.z >> 16)];
              const float t6 = tb[6 * K + (c8.w & 0xFFFFu)], t7 = tb[7 * K + (c8.w >> 16)];
              s += t0; s += t1; s += t2; s += t3; s += t4; s += t5; s += t6; s += t7;
            }
          } else {
            for (uint32_t j = 0; j < m; ++j) s += tables[(uint64_t)j * K + cr[j]];
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

      // Phase 2 (lines 19-27): M = the first ceil(a * |AQ|) entries of AQ, at least one
      const float tf = ceilf(p.tl_ratio * (float)aq_total);
      uint32_t ntop = tf >= 1.0f ? (tf >= (float)aq_total ? aq_total : (uint32_t)tf) : 1u;
      if (ntop > aq_total) ntop = aq_total;
      if (ntop > wlen) { status = QS_SCRATCH; payload = 7; break; }
      // the unpromoted members of M are collected over the whole prefix (up to 64 at a time), so
      // that their rows are fetched in one distance pass and merged into R at once
      auto promote = [&](uint32_t pc) {
        wave_sync();
        const uint32_t pid = lane < pc ? scratch[lane] : 0u;
        wave_sync();
        const uint64_t bad = ballot(lane < pc && (uint64_t)pid >= p.nvec);
        if (bad) {
          status = QS_NODE_NOT_FOUND;
          payload = rl_u(pid, __ffsll((long long)bad) - 1);
          return;
        }
        if (!rows_present(p, pid, pc)) { status = QS_BLOCKED; return; }
        cV += pc;
        const float r_aux = (METRIC == METRIC_COSINE_PRE && lane < pc) ? p.norm2[pid] : 0.0f;
        const float nd = direct_distances<METRIC, ROWT>(emb, p.stride, p.d, pid, pc, qs, q_norm, r_aux);
        const uint64_t rkey = ((uint64_t)ordkey(nd) << 32) | ((uint64_t)pid << 1);
        rlen = tl_merge(res, rlen, rkey, pc, nbuf);
        if (rlen > ef) rlen = ef;  // lines 26-27
      };
      uint32_t pend = 0;
      for (uint32_t c = 0; c < ntop && status == QS_OK; c += 64) {
        const uint32_t i = c + lane;
        const uint64_t x = i < ntop ? win[i] : ~0ull;
        const bool un = i < ntop && !(x & 1ull);
        const uint64_t um = ballot(un);
        const uint32_t pc = (uint32_t)__popcll(um);
        if (!pc) continue;
        if (pend + pc > 64) {
          promote(pend);
          pend = 0;
          if (status != QS_OK) break;
        }
        const uint32_t rank = (uint32_t)__popcll(um & ((1ull << lane) - 1ull));
        if (un) {
          win[i] = x | 1ull;
          scratch[pend + rank] = (uint32_t)(x >> 1) & ID_MASK;
        }
        pend += pc;
      }
      if (pend && status == QS_OK) promote(pend);
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
    }
    if (ovf) {
      for (uint32_t i = lane; i < ocap; i += 64) otab[i] = EMPTY;
    }
    wave_sync();
  }
}

// ------------------------------------------------------------------ launchers
struct FastGeom {
  uint32_t hbits;
  size_t lds;
};

FastGeom fast_geometry(uint32_t ef, uint32_t d) {
  // visited capacity grows with ef (V is roughly 10-30 x ef); overflow goes to HBM
  uint32_t hbits = ef <= 64 ? 10 : ef <= 160 ? 11 : ef <= 320 ? 12 : 13;
  static const int hbits_env = [] { const char* e = getenv("ISL_HBITS"); return e ? atoi(e) : 0; }();
  if (hbits_env >= 8 && hbits_env <= 14) hbits = (uint32_t)hbits_env;  // experiments only
  // visited table, merge buffer, query (+ 64 bytes when d is not a multiple of 16: the operand
  // prefetch of direct_group may touch the rest of the last step)
  size_t lds = ((size_t)4 << hbits) + (size_t)(ef + kBatchMax) * 8 + (size_t)((d + 3) / 4 * 4) * 4 +
               ((d & 15) ? 64 : 0);
  return {hbits, lds};
}

size_t exact_lds(uint32_t ef, uint32_t d) {
  return (size_t)TILE_ROWS * TILE_LD * 4 + 64 * 4 + 64 * 4 + 8 * 4 + (size_t)(ef + 1) * 8 + 16 +
         (size_t)((d + 3) / 4 * 4) * 4;
}

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

template <int S, typename ROWT>
void launch_fast_t(int metric, uint32_t grid, size_t lds, hipStream_t st, const SearchParams& p) {
  switch (metric) {
    case ISL_METRIC_COSINE: launch_one(leann_search_fast<S, ISL_METRIC_COSINE, ROWT>, grid, lds, st, p); break;
    case ISL_METRIC_EUCLIDEAN: launch_one(leann_search_fast<S, ISL_METRIC_EUCLIDEAN, ROWT>, grid, lds, st, p); break;
    case ISL_METRIC_DOT: launch_one(leann_search_fast<S, ISL_METRIC_DOT, ROWT>, grid, lds, st, p); break;
    default: launch_one(leann_search_fast<S, ISL_METRIC_MANHATTAN, ROWT>, grid, lds, st, p); break;
  }
}
template <int S>
void launch_fast(int metric, uint32_t grid, size_t lds, hipStream_t st, const SearchParams& p) {
  if (p.emb_bf16) launch_fast_t<S, uint16_t>(metric, grid, lds, st, p);
  else launch_fast_t<S, float>(metric, grid, lds, st, p);
}

template <bool HNSW>
void launch_exact_t(int metric, uint32_t grid, size_t lds, hipStream_t st, const SearchParams& p) {
  switch (metric) {
    case ISL_METRIC_COSINE: launch_one(leann_search_exact<ISL_METRIC_COSINE, HNSW>, grid, lds, st, p); break;
    case ISL_METRIC_EUCLIDEAN: launch_one(leann_search_exact<ISL_METRIC_EUCLIDEAN, HNSW>, grid, lds, st, p); break;
    case ISL_METRIC_DOT: launch_one(leann_search_exact<ISL_METRIC_DOT, HNSW>, grid, lds, st, p); break;
    default: launch_one(leann_search_exact<ISL_METRIC_MANHATTAN, HNSW>, grid, lds, st, p); break;
  }
}
void launch_exact(int metric, bool hnsw, uint32_t grid, size_t lds, hipStream_t st,
                  const SearchParams& p) {
  if (hnsw) launch_exact_t<true>(metric, grid, lds, st, p);
  else launch_exact_t<false>(metric, grid, lds, st, p);
}

template <typename ROWT>
void launch_two_level_t(int metric, uint32_t grid, size_t lds, hipStream_t st, const SearchParams& p) {
  switch (metric) {
    case ISL_METRIC_COSINE: launch_one(leann_search_two_level<ISL_METRIC_COSINE, ROWT>, grid, lds, st, p); break;
    case ISL_METRIC_EUCLIDEAN: launch_one(leann_search_two_level<ISL_METRIC_EUCLIDEAN, ROWT>, grid, lds, st, p); break;
    case ISL_METRIC_DOT: launch_one(leann_search_two_level<ISL_METRIC_DOT, ROWT>, grid, lds, st, p); break;
    default: launch_one(leann_search_two_level<ISL_METRIC_MANHATTAN, ROWT>, grid, lds, st, p); break;
  }
}
void launch_two_level(int metric, uint32_t grid, size_t lds, hipStream_t st, const SearchParams& p) {
  if (p.emb_bf16) launch_two_level_t<uint16_t>(metric, grid, lds, st, p);
  else launch_two_level_t<float>(metric, grid, lds, st, p);
}

// Two-level search: LDS of one wave = visited table + approximate-queue window + R + staging + query
struct TwoLevelCall {
  float ratio;
  uint32_t window_scale = 1;  // the window grows 4x per retry after a query outgrew it
};
size_t two_level_lds(uint32_t hbits, uint32_t wcap, uint32_t ef, uint32_t d) {
  return ((size_t)4 << hbits) + (size_t)(wcap + 64) * 8 + (size_t)((ef + 63) / 64 * 64 + 64) * 8 + 64 * 8 +
         64 * 4 + (size_t)((d + 3) / 4 * 4) * 4 + ((d & 15) ? 64 : 0);
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

constexpr uint32_t kExactSlots = 32;
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
isl_status ensure_lane_stream(isl::SearchWorkspace& ws) {
  if (ws.stream) return ISL_OK;
  hipStream_t st = nullptr;
  ISL_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  ws.alloc_events++;
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

isl_status prepare_workspace(isl::SearchWorkspace& ws, uint32_t nq, uint32_t slots,
                             uint32_t plog_cap) {
  ISL_TRY(ensure_lane_stream(ws));
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
    ws.ovf_bits = kOvfBits;
    uint64_t n = (uint64_t)slots << kOvfBits;
    ISL_TRY(lane_malloc(ws, ws.ovf_tab, n * 4));
    hipLaunchKernelGGL(fill_u32_kernel, dim3(2048), dim3(256), 0, ws.stream, ws.ovf_tab, n, EMPTY);
    ISL_HIP(hipGetLastError());
    ISL_HIP(hipStreamSynchronize(ws.stream));
    ws.slots = slots;
  }
  if (ws.cap_q < nq) {
    void* ptrs[] = {ws.status, ws.payload, ws.ctr, ws.redo, ws.replay};
    for (void* q : ptrs)
      if (q) (void)hipFree(q);
    ws.status = nullptr; ws.payload = nullptr; ws.ctr = nullptr; ws.redo = nullptr;
    ws.replay = nullptr;
    ws.cap_q = 0;
    uint32_t cap = nq < 1024 ? 1024 : nq;
    ISL_TRY(lane_malloc(ws, ws.status, (size_t)cap * 4));
    ISL_TRY(lane_malloc(ws, ws.payload, (size_t)cap * 8));
    ISL_TRY(lane_malloc(ws, ws.ctr, (size_t)cap * 16));
    ISL_TRY(lane_malloc(ws, ws.redo, (size_t)cap * 4));
    ISL_TRY(lane_malloc(ws, ws.replay, (size_t)cap * 4));
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
      (st = lane_malloc(ws, pl.locks, (size_t)kExactSlots * 4)) == ISL_OK) {
    if (hipMemset(pl.locks, 0, (size_t)kExactSlots * 4) == hipSuccess) {
      pl.slots = kExactSlots;
      return ISL_OK;
    }
    st = isl::fail(ISL_ERR_DEVICE, "hipMemset failed for the exact-kernel pool");
  }
  isl::free_exact_pool(pl);
  return st;
}

// CSR -> 64 ids per node (EMPTY-padded) + degree; one wave per row
__global__ void pad_rows_kernel(const uint64_t* __restrict__ off, const uint32_t* __restrict__ adj, uint64_t n,
                                uint32_t* __restrict__ ell, uint32_t* __restrict__ deg) {
  const uint64_t row = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63;
  if (row >= n) return;
  const uint64_t o0 = off[row];
  const uint32_t d = (uint32_t)(off[row + 1] - o0);
  ell[row * 64 + lane] = lane < d ? adj[o0 + lane] : EMPTY;  // rows longer than 64 never use the copy
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
  uint32_t plog_cap = 0;
  uint32_t tl_wcap = 0;
  size_t tl_lds = 0;
};

isl_status call_geometry(const isl_index* idx, uint64_t d, uint64_t k, uint64_t ef_in, const TwoLevelCall* tl,
                         CallGeometry& g) {
  g.ef = (uint32_t)std::min<uint64_t>(std::max(ef_in, k), 0xFFFFFFFFull);  // leann.rs:890
  if (std::max(ef_in, k) > kMaxExactEf)
    return isl::fail(ISL_ERR_UNSUPPORTED, "ef = %llu exceeds the device limit %u",
                     (unsigned long long)std::max(ef_in, k), kMaxExactEf);
  const uint32_t ef = g.ef;
  const int ncu = isl::device_cu_count(idx->device);
  g.fg = fast_geometry(ef, (uint32_t)d);
  g.use_fast = ef <= 512 && ef >= 1 && idx->max_degree <= 64;
  // resident waves per CU: bounded by LDS (visited table + query) and by the kernel's VGPR
  // budget (<= 128 -> 4 per SIMD)
  const size_t cu_cap = waves_per_cu_cap();
  uint32_t per_cu = (uint32_t)std::min<size_t>(cu_cap, (160 * 1024) / g.fg.lds);
  if (per_cu == 0) g.use_fast = false;
  if (tl) {
    // window of the approximate queue: ceil(a * |AQ|) must stay inside it; |AQ| is bounded by the
    // node count and, in practice, by a few dozen times ef
    g.use_fast = false;
    const float a = tl->ratio > 0.0f ? std::min(tl->ratio, 1.0f) : 0.0f;
    const double bound = (double)std::min<uint64_t>(idx->ncodes, (uint64_t)32 * ef * tl->window_scale);
    const uint64_t want = (uint64_t)(a * bound) + 64;
    g.tl_wcap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>((want + 63) / 64 * 64, 256), 16384);
    while (g.tl_wcap > 256 && two_level_lds(g.fg.hbits, g.tl_wcap, ef, (uint32_t)d) > 160 * 1024) g.tl_wcap -= 64;
    g.tl_lds = two_level_lds(g.fg.hbits, g.tl_wcap, ef, (uint32_t)d);
    if (g.tl_lds > 160 * 1024)
      return isl::fail(ISL_ERR_UNSUPPORTED, "two-level search: ef = %u, d = %llu do not fit the LDS", ef,
                       (unsigned long long)d);
    per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(cu_cap, (160 * 1024) / g.tl_lds));
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
isl_status search_enqueue(const isl_index* idx, isl::SearchWorkspace& ws, const float* d_queries,
                          uint64_t nq, uint64_t d, uint64_t k, uint64_t ef_in, uint64_t* d_ids,
                          float* d_dist, uint32_t* d_count, hipStream_t user_stream,
                          StreamMode mode, const TwoLevelCall* tl = nullptr, bool warm = false) {
  if (nq > 0x7FFFFFFFull) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "too many queries");
  CallGeometry cg;
  ISL_TRY(call_geometry(idx, d, k, ef_in, tl, cg));
  const uint32_t ef = cg.ef;
  const bool use_fast = cg.use_fast;
  const FastGeom& fg = cg.fg;
  const uint32_t slots = cg.slots, plog_cap = cg.plog_cap;
  // per-slot state is indexed by blockIdx.x < min(nq, slots)
  if (!warm) ISL_TRY(prepare_workspace(ws, (uint32_t)nq, (uint32_t)std::min<uint64_t>(nq, slots), plog_cap));
  if (!idx->pool.slots || ((use_fast || (tl && idx->max_degree <= 64)) && !idx->d_ell && idx->d_off && idx->num_nodes)) {
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
  p.replay = ws.replay;
  p.plog = reinterpret_cast<uint2*>(ws.plog);
  p.plog_cap = plog_cap;
  p.hbits = fg.hbits;
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
  p.present = idx->recompute ? idx->d_present : nullptr;
  p.miss = ws.miss;
  p.miss_cap = (uint32_t)std::min<uint64_t>(ws.miss_cap, 0xFFFFFFFFull);
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
  if (tl) {
    // build_distance_tables for the whole batch (pq.rs:307-338), then one wave per query
    if (!warm) ISL_TRY(isl::pq_launch_tables(idx->pq, d_queries, nq, ws.tl_tables, st));
    const uint32_t grid = (uint32_t)std::min<uint64_t>(nq_grid, slots);
    launch_two_level((int)idx->cfg.metric, grid, cg.tl_lds, st, p);
    ISL_HIP(hipGetLastError());
  } else
  if (use_fast && idx->is_hnsw && p.max_level > 0) {
    // HnswGraph::search: greedy descent through the upper layers first (its own kernel, so that
    // the traversal kernel keeps its register budget)
    ISL_TRY(ensure(ws, ws.q_entry, ws.q_entry_cap, std::max<uint64_t>(nq, 1) * 2));
    p.q_entry = ws.q_entry;
    p.q_evals = ws.q_entry + nq;
    const size_t dlds = (size_t)((d + 3) / 4 * 4) * 4 + 64;
    const uint32_t dgrid = (uint32_t)std::min<uint64_t>(nq_grid, 8192);
    switch ((int)idx->cfg.metric) {
      case ISL_METRIC_COSINE: launch_one(hnsw_descent_kernel<ISL_METRIC_COSINE>, dgrid, dlds, st, p); break;
      case ISL_METRIC_EUCLIDEAN: launch_one(hnsw_descent_kernel<ISL_METRIC_EUCLIDEAN>, dgrid, dlds, st, p); break;
      case ISL_METRIC_DOT: launch_one(hnsw_descent_kernel<ISL_METRIC_DOT>, dgrid, dlds, st, p); break;
      default: launch_one(hnsw_descent_kernel<ISL_METRIC_MANHATTAN>, dgrid, dlds, st, p); break;
    }
    ISL_HIP(hipGetLastError());
  }
  if (tl) {
  } else if (use_fast) {
    uint32_t grid = (uint32_t)std::min<uint64_t>(nq_grid, slots);
    int S = ef <= 64 ? 1 : ef <= 128 ? 2 : ef <= 256 ? 4 : 8;
    const int metric = (int)idx->cfg.metric;
    switch (S) {
      case 1: launch_fast<1>(metric, grid, fg.lds, st, p); break;
      case 2: launch_fast<2>(metric, grid, fg.lds, st, p); break;
      case 4: launch_fast<4>(metric, grid, fg.lds, st, p); break;
      default: launch_fast<8>(metric, grid, fg.lds, st, p); break;
    }
    ISL_HIP(hipGetLastError());
  } else if (!warm) {
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
    launch_exact((int)idx->cfg.metric, idx->is_hnsw, grid, exact_lds(ef, (uint32_t)d), st, p);
    ISL_HIP(hipGetLastError());
  }
  ISL_HIP(hipEventRecord(ws.ev1, st));
  ws.st_inflight = st;
  if (warm) return ISL_OK;

  ISL_TRY(publish(ws, nq, k, st));
  ws.enqueued = true;
  ws.nq_inflight = nq;
  ws.k_inflight = k;
  ws.fast_inflight = use_fast;
  return ISL_OK;
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
isl_status search_finish(const isl_index* idx, isl::SearchWorkspace& ws, uint32_t* misses = nullptr,
                         bool* window_short = nullptr) {
  if (!ws.enqueued) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "no search in flight for this token");
  ws.enqueued = false;
  const uint64_t nq = ws.nq_inflight;
  const bool use_fast = ws.fast_inflight;
  ISL_HIP(hipStreamSynchronize(ws.st_inflight));
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
// marks the missed ids present and lists each of them once
__global__ void dedupe_misses_kernel(const uint32_t* __restrict__ miss, uint32_t n,
                                     uint32_t* __restrict__ present, uint32_t* __restrict__ uniq,
                                     uint32_t* __restrict__ uniq_count) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t id = miss[i], bit = 1u << (id & 31);
  const uint32_t old = atomicOr(&present[id >> 5], bit);
  if (!(old & bit)) uniq[atomicAdd(uniq_count, 1u)] = id;
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

isl_status prepare_recompute(isl::SearchWorkspace& ws, uint64_t nq) {
  const uint64_t cap = std::min<uint64_t>(nq * 64 + 64, 0xFFFFFFF0ull);
  if (ws.miss_cap < cap) {
    void* ptrs[] = {ws.miss, ws.uniq, ws.uniq_count};
    for (void* q : ptrs)
      if (q) (void)hipFree(q);
    ws.miss = ws.uniq = ws.uniq_count = nullptr;
    ws.miss_cap = 0;
    ISL_TRY(lane_malloc(ws, ws.miss, cap * 4));
    ISL_TRY(lane_malloc(ws, ws.uniq, cap * 4));
    ISL_TRY(lane_malloc(ws, ws.uniq_count, 4));
    ws.miss_cap = cap;
  }
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
  // differently, the batch is run again with a window four times the size
  TwoLevelCall tcall;
  if (tl) { tcall = *tl; tl = &tcall; }
  if (!idx->recompute) {
    for (;;) {
      ISL_TRY(search_enqueue(idx, ws, d_queries, nq, d, k, ef, d_ids, d_dist, d_count, user_stream, mode, tl));
      bool window_short = false;
      const isl_status st = search_finish(idx, ws, nullptr, tl ? &window_short : nullptr);
      if (st == ISL_ERR_SEARCH && window_short && tcall.window_scale < 64) { tcall.window_scale *= 4; continue; }
      return st;
    }
  }
  // the rounds rewrite the provider's row table and presence bitmap: one recompute search at a time
  std::lock_guard<std::mutex> rlock(idx->recompute_mu);
  ISL_TRY(prepare_recompute(ws, nq));
  ISL_TRY(ensure_lane_stream(ws));
  if (!idx->keep_rows)
    ISL_HIP(hipMemsetAsync(idx->d_present, 0, idx->present_words * 4, mode == StreamMode::USER ? user_stream : ws.stream));
  uint64_t encoded = 0, rounds = 0;
  double kernel_ms = 0.0;
  for (;;) {
    ISL_TRY(search_enqueue(idx, ws, d_queries, nq, d, k, ef, d_ids, d_dist, d_count, user_stream, mode, tl));
    hipStream_t st = mode == StreamMode::USER ? user_stream : ws.stream;
    uint32_t misses = 0;
    bool window_short = false;
    const isl_status fst = search_finish(idx, ws, &misses, tl ? &window_short : nullptr);
    if (fst == ISL_ERR_SEARCH && window_short && tcall.window_scale < 64) { tcall.window_scale *= 4; continue; }
    ISL_TRY(fst);
    kernel_ms += ws.stats.kernel_ms;
    rounds += 1;
    if (!misses) break;
    if (misses > ws.miss_cap) misses = (uint32_t)ws.miss_cap;
    ISL_HIP(hipMemsetAsync(ws.uniq_count, 0, 4, st));
    hipLaunchKernelGGL(dedupe_misses_kernel, dim3((misses + 255) / 256), dim3(256), 0, st, ws.miss, misses,
                       idx->d_present, ws.uniq, ws.uniq_count);
    uint32_t nu = 0;
    ISL_HIP(hipMemcpyAsync(&nu, ws.uniq_count, 4, hipMemcpyDeviceToHost, st));
    ISL_HIP(hipStreamSynchronize(st));
    ISL_TRY(isl::encoder_embed_nodes(idx->enc, idx->d_tokens, idx->d_lens, idx->tok_L, ws.uniq, nu,
                                     idx->enc_normalize, idx->d_emb, idx->emb_stride, st));
    const size_t lds = (size_t)TILE_ROWS * TILE_LD * 4 + 64;
    hipLaunchKernelGGL(row_norm2_list_kernel, dim3(std::min<uint32_t>((nu + 63) / 64, 4096)), dim3(64), lds, st,
                       idx->d_emb, idx->emb_stride, (uint32_t)idx->emb_d, ws.uniq, nu, idx->d_norm2);
    ISL_HIP(hipGetLastError());
    encoded += nu;
  }
  ws.stats.encoded_nodes = encoded;
  ws.stats.recompute_rounds = rounds;
  ws.stats.kernel_ms = kernel_ms;
  ws.stats.allocations = ws.alloc_events - ws.alloc_mark;
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
isl_status host_stage_in(isl::SearchWorkspace& ws, const float* queries, uint64_t nq, uint64_t d, uint64_t k) {
  ISL_TRY(prepare_host_staging(ws, nq, d, k));
  ISL_TRY(ensure_lane_stream(ws));
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

}  // namespace

namespace isl {

bool any_lane_busy(const isl_index* idx) {
  for (const auto& w : idx->ws)
    if (w.busy) return true;
  return false;
}

void free_exact_pool(ExactPool& pl) {
  void* ptrs[] = {pl.cand_d, pl.cand_id, pl.vis_bits, pl.ulist, pl.locks};
  for (void* q : ptrs)
    if (q) (void)hipFree(q);
  pl = ExactPool{};
}

// padded copy of the adjacency (64 ids per node + a degree array): 260 bytes per node buy the
// traversal one dependent memory round trip per hop.  Built where the CSR becomes resident
// (isl_index_upload, isl_index_from_device_csr, isl_index_prepare); the handle's fields are set
// only once the copy is complete.  Not enough memory -> the searches stay on the CSR.
isl_status ensure_padded_adjacency(isl_index* idx) {
  if (idx->d_ell || !idx->d_off || !idx->num_nodes || idx->max_degree > 64) return ISL_OK;
  const uint64_t n = idx->num_nodes;
  uint32_t* ell = nullptr;
  uint32_t* deg = nullptr;
  if (hipMalloc(&ell, n * 64 * 4) != hipSuccess || hipMalloc(&deg, n * 4) != hipSuccess) {
    (void)hipGetLastError();
    if (ell) (void)hipFree(ell);
    return ISL_OK;
  }
  hipLaunchKernelGGL(pad_rows_kernel, dim3((uint32_t)((n + 3) / 4)), dim3(256), 0, nullptr, idx->d_off, idx->d_adj, n,
                     ell, deg);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    (void)hipFree(ell);
    (void)hipFree(deg);
    return fail(ISL_ERR_DEVICE, "padded adjacency: %s", hipGetErrorString(e));
  }
  idx->d_ell = ell;
  idx->d_ell_deg = deg;
  idx->ell_w = 64;
  idx->ell_owned = true;
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
    ISL_TRY(prepare_workspace(ws, (uint32_t)max_nq, ovf_slots, push_log_cap((uint32_t)max_ef)));
    ISL_TRY(prepare_host_staging(ws, max_nq, d, std::max<uint64_t>(max_k, 1)));
    if (idx->recompute) ISL_TRY(prepare_recompute(ws, max_nq));
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
  if (idx->recompute)
    return isl::fail(ISL_ERR_UNSUPPORTED, "the recompute provider answers synchronously: use isl_search_batch_device");
  isl::SearchWorkspace* ws = claim_lane(idx);
  if (!ws) return no_lane();
  LaneGuard guard{idx, ws};
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
  if (idx->recompute)
    return isl::fail(ISL_ERR_UNSUPPORTED, "the recompute provider answers synchronously: use isl_search_batch");
  isl::SearchWorkspace* ws = claim_lane(idx);
  if (!ws) return no_lane();
  LaneGuard guard{idx, ws};
  ISL_TRY(host_stage_in(*ws, queries, nq, d, k));
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
  const isl_status st = search_finish(idx, *ws);  // the D2H result copies sit on the same stream
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
  std::lock_guard<std::mutex> lock(idx->mu);
  for (auto& w : idx->ws)
    if (w.busy && w.token == token) {
      ISL_HIP(hipStreamWaitEvent((hipStream_t)stream, w.ev1, 0));  // recorded behind the last search kernel
      return ISL_OK;
    }
  return isl::fail(ISL_ERR_INVALID_ARGUMENT, "unknown or already completed search token");
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
  ISL_TRY(host_stage_in(ws, queries, nq, d, k));
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
