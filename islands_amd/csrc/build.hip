// LeannIndex::build on the device (src/core/leann.rs:560-833), the step in front of the search
// path: nodes are inserted in id order; each insertion searches the graph built so far with
// ef_construction (search_layer_with_adjacency, :692-749 -- the search kernels of search.hip over
// fixed-width adjacency rows), selects neighbours with the high-degree-preserving rule
// (prune_with_degree_preservation_temp, :761-833) or plain truncation (:685), links them both
// ways (:592-607) and re-sorts a row by distance when it outgrows m0 (prune_neighbors_temp,
// :634-658).  `batch` nodes are inserted per step: with batch = 1 this IS the reference's
// sequential construction and the resulting CsrGraph is identical field by field (levels are
// an input because random_level draws from thread_rng, :549-554); with larger batches the
// searches of a step see the graph as of the step's start and the links of a step are applied
// under per-row locks in an unspecified order -- a throughput mode, same rules, no parity claim.
#include "device_common.hip.h"

#include <algorithm>
#include <vector>

namespace {

using namespace isl_dev;

struct BuildParams {
  const float* emb;
  const float* norm2;
  uint64_t stride;
  uint32_t d;
  uint32_t* ell;       // [n][W]
  uint32_t* ell_deg;   // [n]
  uint32_t* lock;      // [n]
  uint32_t W, m0;
  uint32_t ef;         // candidates per new node (row pitch of cand_*)
  const uint64_t* cand_ids;   // [B][ef] ascending distance (search output)
  const float* cand_dist;
  const uint32_t* cand_cnt;   // [B]
  uint32_t* sel;       // [B][m0] selected neighbours
  uint32_t* sel_cnt;   // [B]
  uint64_t id0;        // first node of the step
  uint32_t B;
  float hub_percentile;
  uint32_t high_degree;  // LeannConfig::high_degree_pruning
  uint32_t locking;      // batch > 1
};

// prune_with_degree_preservation_temp, leann.rs:761-833, for one new node per wave.
__global__ __launch_bounds__(64) void select_kernel(BuildParams p) {
  extern __shared__ uint32_t sm[];
  const uint32_t lane = threadIdx.x, b = blockIdx.x;
  const uint32_t n = p.cand_cnt[b] < p.ef ? p.cand_cnt[b] : p.ef;
  const uint64_t* cid = p.cand_ids + (uint64_t)b * p.ef;
  uint32_t* ids = sm;            // [ef]
  uint32_t* deg = sm + p.ef;     // [ef]
  uint32_t* pos = sm + 2 * p.ef; // [ef] final position in the selection, or ~0
  uint32_t* out = p.sel + (uint64_t)b * p.m0;
  const uint64_t node = p.id0 + b;
  for (uint32_t i = lane; i < n; i += 64) {
    ids[i] = (uint32_t)cid[i];
    // degrees of the candidates in the graph built so far, :768-771 (0 for unknown ids)
    deg[i] = cid[i] < node ? p.ell_deg[cid[i]] : 0u;
  }
  __syncthreads();
  uint32_t nsel = 0;
  if (n <= p.m0 || !p.high_degree) {  // :764-766 / :685 truncate(max_connections)
    nsel = n < p.m0 ? n : p.m0;
    for (uint32_t i = lane; i < nsel; i += 64) out[i] = ids[i];
  } else {
    // hub threshold = degree at the top hub_percentile of the candidate degrees, :774-785
    const uint32_t hub_count = (uint32_t)ceilf((float)n * p.hub_percentile);
    uint32_t thr = 0xFFFFFFFFu;
    if (hub_count > 0 && hub_count < n) {
      // the hub_count-th largest value v: #{> v} < hub_count <= #{>= v}
      uint32_t found = 0xFFFFFFFFu;
      for (uint32_t i = lane; i < n; i += 64) {
        uint32_t gt = 0, ge = 0;
        for (uint32_t j = 0; j < n; ++j) { gt += deg[j] > deg[i]; ge += deg[j] >= deg[i]; }
        if (gt < hub_count && hub_count <= ge) found = deg[i];
      }
      for (int o = 32; o >= 1; o >>= 1) { uint32_t t = (uint32_t)__shfl_xor((int)found, o); found = found < t ? found : t; }
      thr = found;
    }
    const bool have = thr != 0xFFFFFFFFu;
    // hubs keep candidate order, then a stable sort by degree descending (:788-800); regular
    // candidates stay in ascending distance (:802, already sorted; ties keep their order)
    uint32_t nh = 0;
    for (uint32_t i = 0; i < n; ++i) nh += (have && deg[i] >= thr);  // uniform loop, n <= 512
    const uint32_t hub_slots = p.m0 / 4 > 1 ? p.m0 / 4 : 1;  // :807
    const uint32_t first_hubs = nh < hub_slots ? nh : hub_slots;
    for (uint32_t i = lane; i < n; i += 64) {
      const bool hub = have && deg[i] >= thr;
      uint32_t r = 0;
      if (hub) {  // rank among the hubs: larger degree first, equal degree in candidate order
        for (uint32_t j = 0; j < n; ++j)
          if (have && deg[j] >= thr) r += (deg[j] > deg[i]) || (deg[j] == deg[i] && j < i);
        // first hub_slots hubs lead, the rest follow every regular candidate (:808-830)
        pos[i] = r < hub_slots ? r : r + (n - nh);
      } else {
        for (uint32_t j = 0; j < i; ++j) r += !(have && deg[j] >= thr);
        pos[i] = first_hubs + r;
      }
    }
    __syncthreads();
    nsel = n < p.m0 ? n : p.m0;
    for (uint32_t i = lane; i < n; i += 64)
      if (pos[i] < p.m0) out[pos[i]] = ids[i];
  }
  if (lane == 0) p.sel_cnt[b] = nsel;
}

// adjacency.push(neighbors) + bidirectional links + prune_neighbors_temp, leann.rs:592-607, 634-658
template <int METRIC_API>
__global__ __launch_bounds__(64) void link_kernel(BuildParams p) {
  constexpr int METRIC = METRIC_API == ISL_METRIC_COSINE ? METRIC_COSINE_PRE : METRIC_API;
  extern __shared__ __align__(16) unsigned char smem[];
  float* tile = reinterpret_cast<float*>(smem);
  float* qs = tile + TILE_ROWS * TILE_LD;
  const uint32_t lane = threadIdx.x, b = blockIdx.x;
  const uint32_t node = (uint32_t)(p.id0 + b);
  const uint32_t nsel = p.sel_cnt[b];
  const uint32_t* sel = p.sel + (uint64_t)b * p.m0;
  // adjacency.push(neighbors.clone()), :592
  for (uint32_t i = lane; i < nsel; i += 64) p.ell[(uint64_t)node * p.W + i] = sel[i];
  if (lane == 0) p.ell_deg[node] = nsel;
  for (uint32_t t = 0; t < nsel; ++t) {
    const uint32_t nid = sel[t];
    if (p.locking) {
      if (lane == 0) while (atomicCAS(&p.lock[nid], 0u, 1u) != 0u) __builtin_amdgcn_s_sleep(1);
      __threadfence();
      __syncthreads();
    }
    uint32_t* row = p.ell + (uint64_t)nid * p.W;
    uint32_t dg = *((volatile uint32_t*)&p.ell_deg[nid]);
    bool has = false;
    for (uint32_t i = lane; i < dg; i += 64) has |= ((volatile uint32_t*)row)[i] == node;
    if (!ballot(has)) {  // :596 if !adjacency[nid].contains(&id)
      if (lane == 0) row[dg] = node;
      dg += 1;
      if (dg > p.m0) {
        // prune_neighbors_temp: distances from nid to every neighbour, stable sort, keep m0
        __threadfence_block();
        __syncthreads();
        const float q_norm = load_query<METRIC>(p.emb + (uint64_t)nid * p.stride, p.d, qs);
        // the row holds dg = m0 + 1 <= 129 ids: up to three slices of 64, one id per lane each
        constexpr int CH = 3;
        uint32_t rid[CH];
        float dist[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const uint32_t base = 64u * c;
          const uint32_t cnt = dg > base ? (dg - base < 64u ? dg - base : 64u) : 0u;
          rid[c] = lane < cnt ? ((volatile uint32_t*)row)[base + lane] : 0u;
          dist[c] = 0.0f;
          if (cnt) {
            const float aux = (METRIC == METRIC_COSINE_PRE && lane < cnt) ? p.norm2[rid[c]] : 0.0f;
            dist[c] = wave_distances<METRIC>(p.emb, p.stride, p.d, rid[c], cnt, qs, tile, q_norm, aux);
          }
        }
        // stable sort by distance (`<` only, leann.rs:650): rank of every entry among all dg
        uint32_t rank[CH] = {0, 0, 0};
#pragma unroll
        for (int ci = 0; ci < CH; ++ci) {
          const uint32_t basei = 64u * ci;
          const uint32_t cnti = dg > basei ? (dg - basei < 64u ? dg - basei : 64u) : 0u;
          for (uint32_t l = 0; l < cnti; ++l) {
            const float di = rl_f(dist[ci], (int)l);
            const uint32_t i = basei + l;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
              const uint32_t me = 64u * c + lane;
              rank[c] += (di < dist[c]) || (!(dist[c] < di) && !(di < dist[c]) && i < me);
            }
          }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CH; ++c)
          if (64u * c + lane < dg && rank[c] < p.m0) row[rank[c]] = rid[c];
        dg = p.m0;
      }
      if (lane == 0) *((volatile uint32_t*)&p.ell_deg[nid]) = dg;
    }
    if (p.locking) {
      __threadfence();
      __syncthreads();
      if (lane == 0) atomicExch(&p.lock[nid], 0u);
    }
  }
}

__global__ void ell_to_csr_kernel(const uint32_t* __restrict__ ell, const uint32_t* __restrict__ deg,
                                  uint32_t W, const uint64_t* __restrict__ off, uint64_t n,
                                  uint32_t* __restrict__ adj) {
  const uint64_t row = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63;
  if (row >= n) return;
  for (uint32_t i = lane; i < deg[row]; i += 64) adj[off[row] + i] = ell[row * W + i];
}

__global__ void gather_rows_kernel(const float* __restrict__ emb, uint64_t stride, uint32_t d,
                                   uint64_t id0, uint32_t B, float* __restrict__ out) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (uint64_t)B * d) return;
  out[i] = emb[(id0 + i / d) * stride + i % d];
}

void launch_link(int metric, uint32_t grid, size_t lds, const BuildParams& p) {
  switch (metric) {
    case ISL_METRIC_COSINE: hipLaunchKernelGGL(link_kernel<ISL_METRIC_COSINE>, dim3(grid), dim3(64), lds, 0, p); break;
    case ISL_METRIC_EUCLIDEAN: hipLaunchKernelGGL(link_kernel<ISL_METRIC_EUCLIDEAN>, dim3(grid), dim3(64), lds, 0, p); break;
    case ISL_METRIC_DOT: hipLaunchKernelGGL(link_kernel<ISL_METRIC_DOT>, dim3(grid), dim3(64), lds, 0, p); break;
    default: hipLaunchKernelGGL(link_kernel<ISL_METRIC_MANHATTAN>, dim3(grid), dim3(64), lds, 0, p); break;
  }
}

}  // namespace

extern "C" isl_status isl_index_build(const isl_leann_config* cfg_in, const float* vectors, uint64_t n,
                                      uint64_t d, const uint64_t* levels, uint64_t batch, int32_t mem,
                                      int32_t device, isl_index** out) {
  using isl::fail;
  if (!out || (!vectors && n)) return fail(ISL_ERR_INVALID_ARGUMENT, "NULL argument");
  isl_leann_config cfg;
  if (cfg_in) cfg = *cfg_in;
  else isl_leann_config_paper_default(&cfg);
  ISL_TRY(isl_leann_config_validate(&cfg));
  if (n == 0) return isl_index_new(&cfg, out);  // build(&[]) -> Ok(()), leann.rs:565-567
  if (d == 0) return fail(ISL_ERR_EMPTY_COLLECTION, "Empty vector collection");
  if (cfg.m0 > 128) return fail(ISL_ERR_UNSUPPORTED, "the device builder keeps rows of up to 129 ids: m0 <= 128");
  if (cfg.ef_construction > 512) return fail(ISL_ERR_UNSUPPORTED, "ef_construction <= 512 on the device");
  if (n >= 0x7FFFFFF0ull) return fail(ISL_ERR_UNSUPPORTED, "num_nodes exceeds the device id range");
  if (batch == 0) batch = 1;
  ISL_TRY(isl::use_device(device));
  const uint32_t W = (uint32_t)cfg.m0 + 1, m0 = (uint32_t)cfg.m0, ef = (uint32_t)cfg.ef_construction;
  const uint64_t B = std::min<uint64_t>(batch, n);

  // the graph under construction: an index whose adjacency is the fixed-width table
  isl_index* g = nullptr;
  ISL_TRY(isl_index_new(&cfg, &g));
  auto bail = [&](isl_status st) { isl_index_free(g); return st; };
  g->cfg.prune_ratio = 0.0f;  // construction searches do not prune (leann.rs:692-749)
  g->host_csr_valid = false;
  g->num_nodes = n;
  g->device = device;
  g->has_dimension = true;
  g->dimension = d;
  g->max_degree = m0;  // what a construction search can meet: a row is back at <= m0 ids before the next search
  isl_status st = isl_set_embeddings(g, vectors, n, d, ISL_DTYPE_F32, mem);
  if (st != ISL_OK) return bail(st);
  std::vector<void*> tmp;
  auto dalloc = [&](size_t bytes) -> void* {
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 4) != hipSuccess) return nullptr;
    tmp.push_back(p);
    return p;
  };
  auto cleanup = [&]() { for (void* p : tmp) (void)hipFree(p); };
  auto bail2 = [&](isl_status s) { cleanup(); g->d_ell = nullptr; g->d_ell_deg = nullptr; return bail(s); };
  uint32_t* ell = (uint32_t*)dalloc(n * W * 4);
  uint32_t* ell_deg = (uint32_t*)dalloc(n * 4);
  uint32_t* lock = (uint32_t*)dalloc(n * 4);
  float* qbuf = (float*)dalloc(B * d * 4);
  uint64_t* cand_ids = (uint64_t*)dalloc(B * ef * 8);
  float* cand_dist = (float*)dalloc(B * ef * 4);
  uint32_t* cand_cnt = (uint32_t*)dalloc(B * 4);
  uint32_t* sel = (uint32_t*)dalloc(B * m0 * 4);
  uint32_t* sel_cnt = (uint32_t*)dalloc(B * 4);
  if (!ell || !ell_deg || !lock || !qbuf || !cand_ids || !cand_dist || !cand_cnt || !sel || !sel_cnt)
    return bail2(fail(ISL_ERR_DEVICE, "hipMalloc failed for the builder"));
  if (hipMemset(ell_deg, 0, n * 4) != hipSuccess || hipMemset(lock, 0, n * 4) != hipSuccess)
    return bail2(fail(ISL_ERR_DEVICE, "hipMemset failed"));
  g->d_ell = ell;
  g->d_ell_deg = ell_deg;
  g->ell_w = W;

  BuildParams p{};
  p.emb = g->d_emb; p.norm2 = g->d_norm2; p.stride = g->emb_stride; p.d = (uint32_t)d;
  p.ell = ell; p.ell_deg = ell_deg; p.lock = lock; p.W = W; p.m0 = m0; p.ef = ef;
  p.cand_ids = cand_ids; p.cand_dist = cand_dist; p.cand_cnt = cand_cnt; p.sel = sel; p.sel_cnt = sel_cnt;
  p.hub_percentile = cfg.hub_percentile; p.high_degree = cfg.high_degree_pruning;
  const size_t link_lds = (size_t)TILE_ROWS * TILE_LD * 4 + (size_t)((d + 3) / 4 * 4) * 4 + 64;

  bool has_entry = false;
  uint64_t entry = 0, max_level = 0;
  auto note_level = [&](uint64_t id) {  // :610-613
    const uint64_t lv = levels ? levels[id] : 0;
    if (!has_entry || lv > max_level) { has_entry = true; entry = id; max_level = lv; }
  };
  // node 0: no neighbours (adjacency is empty, :585), becomes the entry point
  note_level(0);
  for (uint64_t id0 = 1; id0 < n;) {
    // a step never inserts more than an eighth of the nodes already in the graph: the nodes of
    // a step cannot see each other, and a young graph would otherwise end up as a star
    const uint64_t nb = std::min<uint64_t>(std::min<uint64_t>(B, n - id0), std::max<uint64_t>(1, id0 / 8));
    g->has_entry = true;
    g->entry_point = entry;  // :669: entry_point.unwrap_or(0) as of the start of the step
    hipLaunchKernelGGL(gather_rows_kernel, dim3((uint32_t)((nb * d + 255) / 256)), dim3(256), 0, 0,
                       g->d_emb, g->emb_stride, (uint32_t)d, id0, (uint32_t)nb, qbuf);
    if (hipGetLastError() != hipSuccess) return bail2(fail(ISL_ERR_DEVICE, "gather launch failed"));
    st = isl::search_device_sync(g, qbuf, nb, d, ef, ef, cand_ids, cand_dist, cand_cnt, nullptr);
    if (st != ISL_OK) return bail2(st);
    p.id0 = id0;
    p.B = (uint32_t)nb;
    p.locking = nb > 1;
    hipLaunchKernelGGL(select_kernel, dim3((uint32_t)nb), dim3(64), (size_t)ef * 12, 0, p);
    launch_link((int)cfg.metric, (uint32_t)nb, link_lds, p);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess)
      return bail2(fail(ISL_ERR_DEVICE, "builder kernels failed"));
    for (uint64_t i = 0; i < nb; ++i) note_level(id0 + i);
    id0 += nb;
  }

  // flatten, :617-627
  std::vector<uint32_t> hdeg(n);
  if (hipMemcpy(hdeg.data(), ell_deg, n * 4, hipMemcpyDeviceToHost) != hipSuccess)
    return bail2(fail(ISL_ERR_DEVICE, "cannot read the degrees back"));
  std::vector<uint64_t> off(n + 1, 0);
  for (uint64_t i = 0; i < n; ++i) off[i + 1] = off[i] + hdeg[i];
  uint64_t* d_off = (uint64_t*)dalloc((n + 1) * 8);
  uint32_t* d_adj = (uint32_t*)dalloc((off[n] ? off[n] : 1) * 4);
  if (!d_off || !d_adj || hipMemcpy(d_off, off.data(), (n + 1) * 8, hipMemcpyHostToDevice) != hipSuccess)
    return bail2(fail(ISL_ERR_DEVICE, "hipMalloc failed for the CSR"));
  hipLaunchKernelGGL(ell_to_csr_kernel, dim3((uint32_t)((n + 3) / 4)), dim3(256), 0, 0, ell, ell_deg, W, d_off, n, d_adj);
  if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess)
    return bail2(fail(ISL_ERR_DEVICE, "CSR compaction failed"));
  isl_index* res = nullptr;
  st = isl_index_from_device_csr(&cfg, device, n, d_off, d_adj, 1, entry, 1, d, &res);
  if (st != ISL_OK) return bail2(st);
  res->max_level = max_level;
  if (levels) res->levels.assign(levels, levels + n);
  // the finished index takes over the rows (and their norms) of the construction graph
  res->d_emb = g->d_emb; res->d_norm2 = g->d_norm2;
  res->nvec = g->nvec; res->emb_d = g->emb_d; res->emb_stride = g->emb_stride;
  g->d_emb = nullptr; g->d_norm2 = nullptr;
  g->d_ell = nullptr; g->d_ell_deg = nullptr;
  cleanup();
  isl_index_free(g);
  *out = res;
  return ISL_OK;
}
