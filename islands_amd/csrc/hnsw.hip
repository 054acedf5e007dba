// HnswGraph search facade (src/core/hnsw.rs:149-515, search side) on gfx950.
// The graph is handed over layer by layer in CSR form; layer 0 and the node vectors live in an
// internal LeannIndex-like handle (so the same exact kernel, visited bitmap and row staging are
// used), the upper layers are kept as device CSR arrays for the greedy descent.
#include "common.hpp"

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

isl_status isl_hnsw_search_batch(const isl_hnsw* h, const float* queries, uint64_t nq, uint64_t d,
                                 uint64_t k, uint64_t ef, uint64_t* out_ids, float* out_dist,
                                 uint32_t* out_count) {
  if (!h || !h->core) return isl::fail(ISL_ERR_INVALID_ARGUMENT, "hnsw handle is NULL");
  // is_empty -> Ok(vec![]), dimension check, ef = max(ef, k): hnsw.rs:459-471, :500 -- all
  // shared with the LeannIndex entry point
  return isl_search_batch(h->core, queries, nq, d, k, ef, out_ids, out_dist, out_count);
}

}  // extern "C"
