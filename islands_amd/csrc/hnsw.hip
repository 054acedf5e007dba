// HnswGraph search facade (src/core/hnsw.rs:149-515, search side) on gfx950.
// The graph is handed over layer by layer in CSR form; layer 0 and the node vectors live in an
// internal LeannIndex-like handle (so the same exact kernel, visited bitmap and row staging are
// used), the upper layers are kept as device CSR arrays for the greedy descent.
#include "common.hpp"

#include <algorithm>
#include <vector>

struct isl_hnsw {
  isl_index* core = nullptr;  // layer 0 + vectors + workspaces
  uint64_t m = 0, m0 = 0, ef_construction = 0;
  uint64_t dim = 0;
};

namespace {

__global__ void hnsw_convert_adj(const uint64_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t n,
                                 uint32_t* __restrict__ flag) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint64_t v = in[i];
    if (v > 0x7FFFFFF0ull) { atomicOr(flag, 1u); v = 0x7FFFFFF0ull; }
    out[i] = (uint32_t)v;
  }
}

}  // namespace

extern "C" {

void isl_hnsw_free(isl_hnsw* h) {
  if (!h) return;
  isl_index_free(h->core);
  delete h;
}

uint64_t isl_hnsw_len(const isl_hnsw* h) { return h && h->core ? h->core->num_nodes : 0; }

isl_status isl_hnsw_last_stats(const isl_hnsw* h, isl_search_stats* out) {
  if (!h || !h->core) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "hnsw handle is NULL");
  return isl_search_last_stats(h->core, out);
}

isl_status isl_hnsw_from_layers(uint64_t m, uint64_t m0, uint64_t ef_construction, int32_t metric,
                                uint64_t num_nodes, uint64_t d, uint64_t num_layers,
                                const uint64_t* const* layer_offsets,
                                const uint64_t* const* layer_neighbors, const uint64_t* levels,
                                int32_t has_entry, uint64_t entry_point, uint64_t max_level,
                                const float* vectors, int32_t device, isl_hnsw** out) {
  if (!out) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "out is NULL");
  // HnswConfig::validate, hnsw.rs:72-85
  if (m == 0) return isl::fail(ISL_ERR_INVALID_CONFIG, "Invalid configuration: M must be > 0");
  if (m0 < m) return isl::fail(ISL_ERR_INVALID_CONFIG, "Invalid configuration: M0 must be >= M");
  if (ef_construction < m)
    return isl::fail(ISL_ERR_INVALID_CONFIG, "Invalid configuration: ef_construction must be >= M");
  if (metric < 0 || metric > ISL_METRIC_MANHATTAN)
    return isl::fail(ISL_ERR_INVALID_ARGUMENT, "unknown metric");
  if (num_nodes && (num_layers == 0 || max_level >= num_layers || !layer_offsets || !layer_neighbors))
    return isl::fail(ISL_ERR_INVALID_ARGUMENT, "layers do not cover max_level");
  isl_hnsw* h = new isl_hnsw();
  h->m = m; h->m0 = m0; h->ef_construction = ef_construction; h->dim = d;
  auto bail = [&](isl_status st) { isl_hnsw_free(h); return st; };
  isl_leann_config cfg;
  isl_leann_config_paper_default(&cfg);
  cfg.metric = (uint32_t)metric;
  cfg.prune_ratio = 0.0f;
  isl_status st = isl_index_from_csr(&cfg, num_nodes, num_nodes ? layer_offsets[0] : nullptr,
                                     num_nodes ? layer_neighbors[0] : nullptr, levels, nullptr,
                                     has_entry, entry_point, max_level, num_nodes ? 1 : 0, d, &h->core);
  if (st != ISL_OK) return bail(st);
  h->core->is_hnsw = true;
  if (num_nodes == 0) { *out = h; return ISL_OK; }
  st = isl_index_upload(h->core, device);
  if (st != ISL_OK) return bail(st);
  st = isl_set_embeddings(h->core, vectors, num_nodes, d, ISL_DTYPE_F32, ISL_MEM_HOST);
  if (st != ISL_OK) return bail(st);
  // upper layers -> device CSR (u64 offsets, u32 ids)
  isl_index* c = h->core;
  std::vector<const uint64_t*> offs(max_level + 1, nullptr);
  std::vector<const uint32_t*> adjs(max_level + 1, nullptr);
  uint32_t* d_flag = nullptr;
  if (hipMalloc(&d_flag, 4) != hipSuccess || hipMemset(d_flag, 0, 4) != hipSuccess)
    return bail(isl::fail(ISL_ERR_DEVICE, "hipMalloc failed"));
  c->hnsw_owned.push_back(d_flag);
  for (uint64_t L = 1; L <= max_level; ++L) {
    const uint64_t* off = layer_offsets[L];
    uint64_t nnz = off[num_nodes];
    uint64_t* d_off = nullptr;
    uint32_t* d_adj = nullptr;
    uint64_t* d_tmp = nullptr;
    if (hipMalloc(&d_off, (num_nodes + 1) * 8) != hipSuccess ||
        hipMalloc(&d_adj, (nnz ? nnz : 1) * 4) != hipSuccess ||
        hipMalloc(&d_tmp, (nnz ? nnz : 1) * 8) != hipSuccess)
      return bail(isl::fail(ISL_ERR_DEVICE, "hipMalloc failed for layer %llu", (unsigned long long)L));
    c->hnsw_owned.push_back(d_off);
    c->hnsw_owned.push_back(d_adj);
    hipError_t e = hipMemcpy(d_off, off, (num_nodes + 1) * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess && nnz) e = hipMemcpy(d_tmp, layer_neighbors[L], nnz * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess && nnz) {
      hipLaunchKernelGGL(hnsw_convert_adj, dim3(256), dim3(256), 0, 0, d_tmp, d_adj, nnz, d_flag);
      e = hipDeviceSynchronize();
    }
    (void)hipFree(d_tmp);
    if (e != hipSuccess) return bail(isl::fail(ISL_ERR_DEVICE, "layer upload failed: %s", hipGetErrorString(e)));
    offs[L] = d_off;
    adjs[L] = d_adj;
  }
  uint32_t flag = 0;
  (void)hipMemcpy(&flag, d_flag, 4, hipMemcpyDeviceToHost);
  if (flag) return bail(isl::fail(ISL_ERR_UNSUPPORTED, "node ids above the device id range"));
  if (hipMalloc((void**)&c->d_layer_off, (max_level + 1) * sizeof(void*)) != hipSuccess ||
      hipMalloc((void**)&c->d_layer_adj, (max_level + 1) * sizeof(void*)) != hipSuccess ||
      hipMemcpy((void*)c->d_layer_off, offs.data(), (max_level + 1) * sizeof(void*), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy((void*)c->d_layer_adj, adjs.data(), (max_level + 1) * sizeof(void*), hipMemcpyHostToDevice) != hipSuccess)
    return bail(isl::fail(ISL_ERR_DEVICE, "layer table upload failed"));
  c->hnsw_layers = max_level + 1;
  *out = h;
  return ISL_OK;
}

// HnswGraph::from_bytes, hnsw.rs:511-514: bincode of the derive(Serialize) struct (hnsw.rs:150-164).
// bincode 1.x default layout as in api_index.hip (little-endian, fixed-width, usize = u64, Vec and
// HashMap = u64 length + items, Option = u8 tag, C-like enum = u32 variant index):
//   config { m, m0, ef_construction: u64; ml: f64; metric: u32; max_layers: u64 }          hnsw.rs:15-28
//   nodes: u64 count, then (key: u64, HnswNode { id: u64; vector: Vec<f32>; connections:
//          Vec<Vec<u64>>; level: u64 }) in the writer's HashMap order                      hnsw.rs:90-99
//   entry_point: Option<u64>; max_level: u64; dimension: Option<u64>; next_id: u64
// The byte layout is unpinned by the reference (tests round-trip only, hnsw.rs:689-709) and the
// pinned bincode is 3.0.0 (Cargo.lock:838-841); this is the documented 1.x-compatible intent.
// Node ids must be 0..n-1 (insert assigns them from next_id, hnsw.rs:218-219), in any order.
static isl_status hnsw_from_bytes_impl(const uint8_t* bytes, size_t len, int32_t device, isl_hnsw** out);

isl_status isl_hnsw_from_bytes(const uint8_t* bytes, size_t len, int32_t device, isl_hnsw** out) {
  if (!out || (!bytes && len)) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "NULL argument");
  // every length in the buffer is untrusted: nothing may throw across the C boundary
  try {
    return hnsw_from_bytes_impl(bytes, len, device, out);
  } catch (const std::exception& e) {
    return isl::fail(ISL_ERR_DESERIALIZATION, "Deserialization error: %s", e.what());
  }
}

static isl_status hnsw_from_bytes_impl(const uint8_t* bytes, size_t len, int32_t device, isl_hnsw** out) {
  size_t pos = 0;
  bool ok = true;
  auto need = [&](size_t n) { if (ok && len - pos < n) ok = false; return ok; };
  auto u64 = [&]() -> uint64_t { uint64_t v = 0; if (need(8)) { memcpy(&v, bytes + pos, 8); pos += 8; } return v; };
  auto u32 = [&]() -> uint32_t { uint32_t v = 0; if (need(4)) { memcpy(&v, bytes + pos, 4); pos += 4; } return v; };
  auto u8 = [&]() -> uint8_t { uint8_t v = 0; if (need(1)) { v = bytes[pos]; pos += 1; } return v; };
  auto bad = [&](const char* what) {
    return isl::fail(ISL_ERR_DESERIALIZATION, "Deserialization error: %s (offset %llu of %llu)", what,
                     (unsigned long long)pos, (unsigned long long)len);
  };
  const uint64_t m = u64(), m0 = u64(), efc = u64();
  (void)u64();  // ml: f64, only used by random_level (hnsw.rs:196-200)
  const uint32_t metric = u32();
  (void)u64();  // max_layers
  const uint64_t n = u64();
  if (!ok) return bad("truncated header");
  if (metric > ISL_METRIC_MANHATTAN) return bad("unknown DistanceMetric variant");
  if (n > (len - pos) / 32) return bad("node count exceeds the buffer");  // >= 32 bytes per node
  std::vector<uint64_t> levels(n, 0);
  std::vector<float> vectors;
  std::vector<uint8_t> seen(n, 0);
  std::vector<std::vector<std::vector<uint64_t>>> conn(n);
  uint64_t d = 0;
  for (uint64_t e = 0; e < n && ok; ++e) {
    const uint64_t key = u64(), id = u64(), vlen = u64();
    if (!ok) break;
    if (key != id) return bad("HashMap key differs from HnswNode::id");
    if (id >= n || seen[id]) return isl::fail(ISL_ERR_UNSUPPORTED, "HnswGraph node ids must be 0..n-1 (id %llu of %llu nodes)",
                                              (unsigned long long)id, (unsigned long long)n);
    seen[id] = 1;
    // vlen * 4 and n * d must not wrap: every node carries its vector inside the buffer
    if (vlen > (len - pos) / 4) return bad("vector length exceeds the buffer");
    if (e == 0) {
      d = vlen;
      if (d && n > (len / 4) / d) return bad("node count times vector length exceeds the buffer");
      vectors.assign((size_t)n * d, 0.0f);
    }
    if (vlen != d) return bad("nodes with different vector lengths");
    if (!need(vlen * 4)) break;
    memcpy(vectors.data() + (size_t)id * d, bytes + pos, vlen * 4);
    pos += vlen * 4;
    const uint64_t nl = u64();
    if (!ok || nl > (len - pos) / 8 + 1) { ok = false; break; }
    conn[id].resize(nl);
    for (uint64_t L = 0; L < nl && ok; ++L) {
      const uint64_t c = u64();
      if (!ok || c > (len - pos) / 8) { ok = false; break; }
      conn[id][L].resize(c);
      if (c) memcpy(conn[id][L].data(), bytes + pos, c * 8);
      pos += c * 8;
    }
    levels[id] = u64();
  }
  const uint8_t has_entry = u8();
  const uint64_t entry = has_entry ? u64() : 0;
  const uint64_t max_level = u64();
  const uint8_t has_dim = u8();
  const uint64_t dim = has_dim ? u64() : 0;
  (void)u64();  // next_id
  if (!ok) return bad("truncated input");
  if (has_entry > 1 || has_dim > 1) return bad("invalid Option tag");
  if (pos != len) return bad("trailing bytes");
  if (n && has_dim && dim != d) return bad("dimension differs from the node vectors");
  // per-layer CSR over all nodes
  uint64_t num_layers = max_level + 1;
  for (uint64_t i = 0; i < n; ++i) num_layers = std::max<uint64_t>(num_layers, conn[i].size());
  if (num_layers > 64) return bad("implausible layer count");
  std::vector<std::vector<uint64_t>> offs(num_layers, std::vector<uint64_t>(n + 1, 0)), adjs(num_layers);
  for (uint64_t L = 0; L < num_layers; ++L) {
    for (uint64_t i = 0; i < n; ++i) {
      if (L < conn[i].size()) adjs[L].insert(adjs[L].end(), conn[i][L].begin(), conn[i][L].end());
      offs[L][i + 1] = adjs[L].size();
    }
    if (adjs[L].empty()) adjs[L].push_back(0);
  }
  std::vector<const uint64_t*> po(num_layers), pa(num_layers);
  for (uint64_t L = 0; L < num_layers; ++L) { po[L] = offs[L].data(); pa[L] = adjs[L].data(); }
  return isl_hnsw_from_layers(m, m0, efc, (int32_t)metric, n, n ? d : dim, num_layers, po.data(), pa.data(),
                              levels.data(), has_entry, entry, max_level, vectors.data(), device, out);
}

isl_status isl_hnsw_search_batch(const isl_hnsw* h, const float* queries, uint64_t nq, uint64_t d,
                                 uint64_t k, uint64_t ef, uint64_t* out_ids, float* out_dist,
                                 uint32_t* out_count) {
  if (!h || !h->core) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "hnsw handle is NULL");
  // is_empty -> Ok(vec![]), dimension check, ef = max(ef, k): hnsw.rs:459-471, :500 -- all
  // shared with the LeannIndex entry point
  return isl_search_batch(h->core, queries, nq, d, k, ef, out_ids, out_dist, out_count);
}

}  // extern "C"
