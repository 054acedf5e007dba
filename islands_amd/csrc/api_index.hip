// Host side of the C ABI: error records, LeannConfig, LeannIndex lifecycle,
// bincode (de)serialisation, device upload of the CSR graph and of the
// in-memory embedding provider.  Mirrors src/core/leann.rs of the reference;
// each function cites the lines it replaces.
#include "device_common.hip.h"
#include "encoder.hpp"

#include <algorithm>
#include <cmath>
#include <new>

namespace isl {

ErrorRecord& last_error() {
  thread_local ErrorRecord rec;
  return rec;
}

isl_status fail(isl_status st, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  last_error().message = buf;
  return st;
}

// CoreError::DimensionMismatch display text, error.rs:11
isl_status fail_dim(uint64_t expected, uint64_t actual) {
  ErrorRecord& r = last_error();
  r.expected = expected;
  r.actual = actual;
  return fail(ISL_ERR_DIMENSION_MISMATCH, "Vector dimension mismatch: expected %llu, got %llu",
              (unsigned long long)expected, (unsigned long long)actual);
}

// CoreError::NodeNotFound display text, error.rs:31
isl_status fail_node(uint64_t node) {
  last_error().node = node;
  return fail(ISL_ERR_NODE_NOT_FOUND, "Node not found: %llu", (unsigned long long)node);
}

// The device checks (count, architecture) are made once per device and process; later calls only
// select the device -- the search entry points run this on every call.
namespace {
struct DeviceInfo { int state = 0; int ncu = 256; };  // state: 0 unknown, 1 verified gfx950
DeviceInfo g_devices[64];
std::mutex g_devices_mu;
}  // namespace

isl_status use_device(int32_t device) {
  if (device >= 0 && device < 64 && g_devices[device].state == 1) {
    ISL_HIP(hipSetDevice(device));
    return ISL_OK;
  }
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return fail(ISL_ERR_DEVICE, "no HIP device visible (the gfx950 compute path has no CPU fallback)");
  }
  if (device < 0 || device >= n)
    return fail(ISL_ERR_DEVICE, "device %d out of range (%d visible)", device, n);
  ISL_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  ISL_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(ISL_ERR_DEVICE, "device %d is %s; this library is built for gfx950 only", device,
                prop.gcnArchName);
  if (device < 64) {
    std::lock_guard<std::mutex> lock(g_devices_mu);
    g_devices[device].ncu = prop.multiProcessorCount;
    g_devices[device].state = 1;
  }
  return ISL_OK;
}

int device_cu_count(int32_t device) {
  return (device >= 0 && device < 64 && g_devices[device].state == 1) ? g_devices[device].ncu : 256;
}

void free_workspace(SearchWorkspace& ws) {
  void* ptrs[] = {ws.ovf_tab, ws.status,  ws.payload,   ws.ctr,        ws.ticket,     ws.redo, ws.replay, ws.qsel, ws.qsel_h, ws.plog,
                  ws.q_stage, ws.ids_stage, ws.dist_stage, ws.count_stage, ws.d_prof, ws.d_tline, ws.q_entry, ws.miss, ws.uniq,
                  ws.uniq_count, ws.tl_tables, ws.qstate, ws.qflag, ws.qlist, ws.uslots, ws.xslot, ws.co_q, ws.co_ids, ws.co_dist,
                  ws.co_cnt};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  void* pinned[] = {ws.h_status, ws.h_ctr, ws.h_head, ws.h_q, ws.h_ids, ws.h_dist, ws.h_count, ws.h_qlist, ws.h_xlist};
  for (void* p : pinned)
    if (p) (void)hipHostFree(p);
  if (ws.ev0) (void)hipEventDestroy(ws.ev0);
  if (ws.ev1) (void)hipEventDestroy(ws.ev1);
  if (ws.ev_in) (void)hipEventDestroy(ws.ev_in);
  if (ws.ev_done) (void)hipEventDestroy(ws.ev_done);
  // (ws.stream belongs to the device's stream pool)
  ws = SearchWorkspace{};
}

}  // namespace isl

using namespace isl;

// Up to kSearchLanes searches overlap on their own streams; the runtime's default of 4 hardware
// queues per process would make them share queues and serialise.  Effective when the library is
// loaded before the HIP runtime initialises (an explicit setting in the environment wins).
__attribute__((constructor)) static void isl_default_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "32", 0); }

extern "C" {

const char* isl_last_error_message(void) { return last_error().message.c_str(); }
uint64_t isl_last_error_expected(void) { return last_error().expected; }
uint64_t isl_last_error_actual(void) { return last_error().actual; }
uint64_t isl_last_error_node(void) { return last_error().node; }
uint32_t isl_abi_version(void) { return ISL_ABI_VERSION; }

const char* isl_status_name(isl_status s) {
  switch (s) {
    case ISL_OK: return "Ok";
    case ISL_ERR_DIMENSION_MISMATCH: return "DimensionMismatch";
    case ISL_ERR_EMPTY_COLLECTION: return "EmptyCollection";
    case ISL_ERR_INVALID_CONFIG: return "InvalidConfig";
    case ISL_ERR_INDEX_NOT_BUILT: return "IndexNotBuilt";
    case ISL_ERR_NODE_NOT_FOUND: return "NodeNotFound";
    case ISL_ERR_SERIALIZATION: return "Serialization";
    case ISL_ERR_DESERIALIZATION: return "Deserialization";
    case ISL_ERR_IO: return "Io";
    case ISL_ERR_HNSW: return "HnswError";
    case ISL_ERR_PQ: return "PQError";
    case ISL_ERR_SEARCH: return "SearchError";
    case ISL_ERR_EMBEDDING: return "EmbeddingError";
    case ISL_ERR_DEVICE: return "Device";
    case ISL_ERR_INVALID_ARGUMENT: return "InvalidArgument";
    case ISL_ERR_UNSUPPORTED: return "Unsupported";
    default: return "Unknown";
  }
}

int32_t isl_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  int ok = 0;
  for (int i = 0; i < n; i++) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, i) == hipSuccess &&
        strncmp(prop.gcnArchName, "gfx950", 6) == 0)
      ok++;
  }
  return ok;
}

// ---------------------------------------------------------------- LeannConfig
void isl_leann_config_paper_default(isl_leann_config* c) {  // leann.rs:386-403
  if (!c) return;
  memset(c, 0, sizeof(*c));
  c->m = 30;
  c->m0 = 60;
  c->ef_construction = 128;
  c->ml = 1.0 / std::log(30.0);
  c->max_layers = 16;
  c->metric = ISL_METRIC_COSINE;
  c->ef_search = 64;
  c->beam_width = 1;
  c->prune_ratio = 0.0f;
  c->pruning_strategy = ISL_PRUNE_GLOBAL;
  c->high_degree_pruning = 1;
  c->hub_percentile = 0.02f;
  c->is_compact = 1;
  c->is_recompute = 1;
}

void isl_leann_config_fast(isl_leann_config* c) {  // leann.rs:406-416
  if (!c) return;
  isl_leann_config_paper_default(c);
  c->m = 16;
  c->m0 = 32;
  c->ef_construction = 100;
  c->ef_search = 32;
  c->beam_width = 1;
  c->prune_ratio = 0.3f;
}

void isl_leann_config_accurate(isl_leann_config* c) {  // leann.rs:419-429
  if (!c) return;
  isl_leann_config_paper_default(c);
  c->m = 48;
  c->m0 = 96;
  c->ef_construction = 400;
  c->ef_search = 128;
  c->beam_width = 1;
  c->prune_ratio = 0.0f;
}

isl_status isl_leann_config_validate(const isl_leann_config* c) {  // leann.rs:432-460
  if (!c) return fail(ISL_ERR_INVALID_ARGUMENT, "config is NULL");
  if (c->m == 0) return fail(ISL_ERR_INVALID_CONFIG, "Invalid configuration: M must be > 0");
  if (c->m0 < c->m) return fail(ISL_ERR_INVALID_CONFIG, "Invalid configuration: M0 must be >= M");
  if (c->ef_construction < c->m)
    return fail(ISL_ERR_INVALID_CONFIG, "Invalid configuration: ef_construction must be >= M");
  if (c->prune_ratio < 0.0f || c->prune_ratio > 1.0f)
    return fail(ISL_ERR_INVALID_CONFIG, "Invalid configuration: prune_ratio must be in [0.0, 1.0]");
  if (c->beam_width == 0)
    return fail(ISL_ERR_INVALID_CONFIG, "Invalid configuration: beam_width must be > 0");
  if (c->hub_percentile < 0.0f || c->hub_percentile > 1.0f)
    return fail(ISL_ERR_INVALID_CONFIG,
                "Invalid configuration: hub_percentile must be in [0.0, 1.0]");
  if (c->metric > ISL_METRIC_MANHATTAN || c->pruning_strategy > ISL_PRUNE_PROPORTIONAL)
    return fail(ISL_ERR_INVALID_ARGUMENT, "metric / pruning_strategy out of range");
  return ISL_OK;
}

// ----------------------------------------------------------------- LeannIndex
isl_status isl_index_new(const isl_leann_config* cfg, isl_index** out) {  // leann.rs:504-511
  if (!out) return fail(ISL_ERR_INVALID_ARGUMENT, "out is NULL");
  isl_leann_config c;
  if (cfg) c = *cfg; else isl_leann_config_paper_default(&c);
  ISL_TRY(isl_leann_config_validate(&c));
  isl_index* idx = new (std::nothrow) isl_index();
  if (!idx) return fail(ISL_ERR_IO, "out of memory");
  idx->cfg = c;
  *out = idx;
  return ISL_OK;
}

isl_status isl_index_from_csr(const isl_leann_config* cfg, uint64_t num_nodes,
                              const uint64_t* node_offsets, const uint64_t* neighbors,
                              const uint64_t* levels, const uint64_t* degree_counts,
                              int32_t has_entry, uint64_t entry_point, uint64_t max_level,
                              int32_t has_dimension, uint64_t dimension, isl_index** out) {
  if (!out || (!node_offsets && num_nodes))
    return fail(ISL_ERR_INVALID_ARGUMENT, "NULL argument");
  isl_index* idx = nullptr;
  ISL_TRY(isl_index_new(cfg, &idx));
  idx->num_nodes = num_nodes;
  if (num_nodes) {
    idx->node_offsets.assign(node_offsets, node_offsets + num_nodes + 1);
    for (uint64_t i = 0; i < num_nodes; i++)
      if (idx->node_offsets[i + 1] < idx->node_offsets[i]) {
        delete idx;
        return fail(ISL_ERR_INVALID_ARGUMENT, "node_offsets must be non-decreasing");
      }
    uint64_t nnz = idx->node_offsets[num_nodes];
    if (nnz && !neighbors) {
      delete idx;
      return fail(ISL_ERR_INVALID_ARGUMENT, "neighbors is NULL");
    }
    idx->neighbors.assign(neighbors, neighbors + nnz);
    if (levels) idx->levels.assign(levels, levels + num_nodes);
    else idx->levels.assign(num_nodes, 0);
    if (degree_counts) idx->degree_counts.assign(degree_counts, degree_counts + num_nodes);
    else {
      idx->degree_counts.resize(num_nodes);
      for (uint64_t i = 0; i < num_nodes; i++)
        idx->degree_counts[i] = idx->node_offsets[i + 1] - idx->node_offsets[i];
    }
  }
  idx->has_entry = has_entry != 0;
  idx->entry_point = has_entry ? entry_point : 0;
  idx->max_level = max_level;
  idx->has_dimension = has_dimension != 0;
  idx->dimension = has_dimension ? dimension : 0;
  *out = idx;
  return ISL_OK;
}

void isl_index_free(isl_index* idx) {
  if (!idx) return;
  if (idx->device >= 0) {
    (void)hipSetDevice(idx->device);
    join_lane_workers(idx);
    for (auto& w : idx->ws)
      if (w.busy && w.st_inflight) (void)hipStreamSynchronize(w.st_inflight);
    if (idx->ell_owned) { (void)hipFree(idx->d_ell); (void)hipFree(idx->d_ell_deg); }
    if (idx->d_emb16) (void)hipFree(idx->d_emb16);
    if (idx->d_tokens) (void)hipFree(idx->d_tokens);
    if (idx->d_lens) (void)hipFree(idx->d_lens);
    if (idx->d_slot_of) (void)hipFree(idx->d_slot_of);
    if (idx->d_owner) (void)hipFree(idx->d_owner);
    if (idx->d_stamp) (void)hipFree(idx->d_stamp);
    if (idx->d_slab_head) (void)hipFree(idx->d_slab_head);
    if (idx->d_off) (void)hipFree(idx->d_off);
    if (idx->d_adj) (void)hipFree(idx->d_adj);
    if (idx->d_emb) (void)hipFree(idx->d_emb);
    if (idx->d_norm2) (void)hipFree(idx->d_norm2);
    if (idx->d_codes) (void)hipFree(idx->d_codes);
    for (void* q : idx->hnsw_owned) (void)hipFree(q);
    if (idx->d_layer_off) (void)hipFree((void*)idx->d_layer_off);
    if (idx->d_layer_adj) (void)hipFree((void*)idx->d_layer_adj);
    for (auto& w : idx->ws) free_workspace(w);
    free_exact_pool(idx->pool);
  }
  delete idx;
}

uint64_t isl_index_len(const isl_index* idx) { return idx ? idx->num_nodes : 0; }
int32_t isl_index_is_empty(const isl_index* idx) { return !idx || idx->num_nodes == 0; }
int32_t isl_index_dimension(const isl_index* idx, uint64_t* dim) {
  if (!idx || !idx->has_dimension) return 0;
  if (dim) *dim = idx->dimension;
  return 1;
}
int32_t isl_index_is_recompute(const isl_index* idx) { return idx && idx->cfg.is_recompute; }
int32_t isl_index_is_compact(const isl_index* idx) { return idx && idx->cfg.is_compact; }
isl_status isl_index_config(const isl_index* idx, isl_leann_config* out) {
  if (!idx || !out) return fail(ISL_ERR_INVALID_ARGUMENT, "NULL argument");
  *out = idx->cfg;
  return ISL_OK;
}
int32_t isl_index_entry_point(const isl_index* idx, uint64_t* entry) {
  if (!idx || !idx->has_entry) return 0;
  if (entry) *entry = idx->entry_point;
  return 1;
}
uint64_t isl_index_max_level(const isl_index* idx) { return idx ? idx->max_level : 0; }

// CsrGraph::storage_bytes, leann.rs:296-301 (usize = u64 = 8 bytes each)
uint64_t isl_index_storage_bytes(const isl_index* idx) {
  if (!idx) return 0;
  if (materialise_host_csr(idx) != ISL_OK) return 0;
  return 8ull * (idx->node_offsets.size() + idx->neighbors.size() + idx->levels.size() +
                 idx->degree_counts.size());
}

int32_t isl_index_get_neighbors(const isl_index* idx, uint64_t node, const uint64_t** ptr,
                                size_t* len) {  // leann.rs:225-233
  if (!idx || !ptr || !len) return 0;
  if (node >= idx->num_nodes) return 0;
  if (materialise_host_csr(idx) != ISL_OK) return 0;
  uint64_t s = idx->node_offsets[node], e = idx->node_offsets[node + 1];
  *ptr = idx->neighbors.data() + s;
  *len = (size_t)(e - s);
  return 1;
}

// ------------------------------------------------------------------- bincode
// bincode 1.x default options: little-endian, fixed-width integers, usize as
// u64, Vec<T> = u64 length + items, Option<T> = u8 tag + payload, C-like enum =
// u32 variant index, bool = u8.  LeannIndex field order: config, graph,
// dimension (leann.rs:493-500).  The reference pins `bincode = "3.0.0"`
// (Cargo.toml:64) yet calls the 1.x free functions (leann.rs:1060,1065); the
// reference tests only round-trip (leann.rs:1347-1384) -> byte layout unpinned.
namespace {
struct Writer {
  std::vector<uint8_t> b;
  void raw(const void* p, size_t n) {
    const uint8_t* s = (const uint8_t*)p;
    b.insert(b.end(), s, s + n);
  }
  void u8(uint8_t v) { b.push_back(v); }
  void u32(uint32_t v) { raw(&v, 4); }
  void u64(uint64_t v) { raw(&v, 8); }
  void f32(float v) { raw(&v, 4); }
  void f64(double v) { raw(&v, 8); }
  void vec64(const std::vector<uint64_t>& v) {
    u64(v.size());
    raw(v.data(), v.size() * 8);
  }
};
struct Reader {
  const uint8_t* p;
  size_t n, off = 0;
  bool ok = true;
  bool take(void* dst, size_t k) {
    if (!ok || n - off < k) { ok = false; return false; }
    memcpy(dst, p + off, k);
    off += k;
    return true;
  }
  uint8_t u8() { uint8_t v = 0; take(&v, 1); return v; }
  uint32_t u32() { uint32_t v = 0; take(&v, 4); return v; }
  uint64_t u64() { uint64_t v = 0; take(&v, 8); return v; }
  float f32() { float v = 0; take(&v, 4); return v; }
  double f64() { double v = 0; take(&v, 8); return v; }
  bool vec64(std::vector<uint64_t>& v) {
    uint64_t len = u64();
    if (!ok || len > (n - off) / 8) { ok = false; return false; }
    v.resize(len);
    return take(v.data(), len * 8);
  }
};
}  // namespace

isl_status isl_index_to_bytes(const isl_index* idx, uint8_t** out, size_t* len) {  // leann.rs:1059
  if (!idx || !out || !len) return fail(ISL_ERR_INVALID_ARGUMENT, "NULL argument");
  ISL_TRY(materialise_host_csr(idx));
  Writer w;
  const isl_leann_config& c = idx->cfg;
  w.u64(c.m); w.u64(c.m0); w.u64(c.ef_construction); w.f64(c.ml); w.u64(c.max_layers);
  w.u32(c.metric); w.u64(c.ef_search); w.u64(c.beam_width); w.f32(c.prune_ratio);
  w.u32(c.pruning_strategy); w.u8(c.high_degree_pruning ? 1 : 0); w.f32(c.hub_percentile);
  w.u8(c.is_compact ? 1 : 0); w.u8(c.is_recompute ? 1 : 0);
  w.vec64(idx->node_offsets);
  w.vec64(idx->neighbors);
  w.vec64(idx->levels);
  if (idx->has_entry) { w.u8(1); w.u64(idx->entry_point); } else w.u8(0);
  w.u64(idx->max_level);
  w.u64(idx->num_nodes);
  w.vec64(idx->degree_counts);
  if (idx->has_dimension) { w.u8(1); w.u64(idx->dimension); } else w.u8(0);
  uint8_t* buf = (uint8_t*)malloc(w.b.size() ? w.b.size() : 1);
  if (!buf) return fail(ISL_ERR_SERIALIZATION, "Serialization error: out of memory");
  memcpy(buf, w.b.data(), w.b.size());
  *out = buf;
  *len = w.b.size();
  return ISL_OK;
}

void isl_free_bytes(uint8_t* p) { free(p); }

static isl_status index_from_bytes_impl(const uint8_t* bytes, size_t len, isl_index** out);

isl_status isl_index_from_bytes(const uint8_t* bytes, size_t len, isl_index** out) {  // leann.rs:1064
  if (!bytes || !out) return fail(ISL_ERR_INVALID_ARGUMENT, "NULL argument");
  // lengths inside the buffer are untrusted: nothing may throw across the C boundary
  try {
    return index_from_bytes_impl(bytes, len, out);
  } catch (const std::exception& e) {
    return fail(ISL_ERR_DESERIALIZATION, "Deserialization error: %s", e.what());
  }
}

static isl_status index_from_bytes_impl(const uint8_t* bytes, size_t len, isl_index** out) {
  Reader r{bytes, len};
  isl_index* idx = new (std::nothrow) isl_index();
  if (!idx) return fail(ISL_ERR_IO, "out of memory");
  isl_leann_config& c = idx->cfg;
  c.m = r.u64(); c.m0 = r.u64(); c.ef_construction = r.u64(); c.ml = r.f64();
  c.max_layers = r.u64(); c.metric = r.u32(); c.ef_search = r.u64(); c.beam_width = r.u64();
  c.prune_ratio = r.f32(); c.pruning_strategy = r.u32(); c.high_degree_pruning = r.u8();
  c.hub_percentile = r.f32(); c.is_compact = r.u8(); c.is_recompute = r.u8();
  bool enum_ok = c.metric <= ISL_METRIC_MANHATTAN && c.pruning_strategy <= ISL_PRUNE_PROPORTIONAL &&
                 c.high_degree_pruning <= 1 && c.is_compact <= 1 && c.is_recompute <= 1;
  r.vec64(idx->node_offsets);
  r.vec64(idx->neighbors);
  r.vec64(idx->levels);
  uint8_t tag = r.u8();
  if (tag == 1) { idx->has_entry = true; idx->entry_point = r.u64(); }
  else if (tag != 0) enum_ok = false;
  idx->max_level = r.u64();
  idx->num_nodes = r.u64();
  r.vec64(idx->degree_counts);
  uint8_t dtag = r.u8();
  if (dtag == 1) { idx->has_dimension = true; idx->dimension = r.u64(); }
  else if (dtag != 0) enum_ok = false;
  if (!r.ok || !enum_ok) {
    delete idx;
    return fail(ISL_ERR_DESERIALIZATION, "Deserialization error: %s",
                r.ok ? "invalid enum/option tag" : "unexpected end of input");
  }
  // bincode itself accepts trailing bytes with deserialize(); structural sanity below is
  // ours: get_neighbors (leann.rs:230-232) would index out of bounds otherwise.
  // (num_nodes is untrusted: num_nodes + 1 must not wrap before it is compared)
  if (idx->num_nodes >= (1ull << 61) || idx->node_offsets.size() != idx->num_nodes + 1 ||
      (idx->num_nodes && idx->node_offsets[idx->num_nodes] > idx->neighbors.size())) {
    delete idx;
    return fail(ISL_ERR_DESERIALIZATION,
                "Deserialization error: node_offsets inconsistent with num_nodes/neighbors");
  }
  for (uint64_t i = 0; i < idx->num_nodes; i++)
    if (idx->node_offsets[i + 1] < idx->node_offsets[i]) {
      delete idx;
      return fail(ISL_ERR_DESERIALIZATION, "Deserialization error: node_offsets not monotonic");
    }
  *out = idx;
  return ISL_OK;
}

}  // extern "C"

// ------------------------------------------------------------ device upload
namespace {

constexpr uint32_t kMaxDeviceId = 0x7FFFFFF0u;  // ids above this cannot carry the flag bit

__global__ void convert_adj_kernel(const uint64_t* __restrict__ in, uint32_t* __restrict__ out,
                                   uint64_t n, uint32_t* __restrict__ flags) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  bool bad = false;
  for (; i < n; i += stride) {
    uint64_t v = in[i];
    if (v > kMaxDeviceId) { bad = true; v = kMaxDeviceId; }
    out[i] = (uint32_t)v;
  }
  if (bad) atomicOr(flags, 1u);
}

// bf16 bit patterns -> their exact f32 images
__global__ void widen_bf16_kernel(const uint16_t* __restrict__ src, uint64_t sstride, uint32_t d, uint64_t n,
                                  float* __restrict__ dst, uint64_t dstride) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * d) return;
  const uint64_t r = i / d, c = i % d;
  dst[r * dstride + c] = __uint_as_float((uint32_t)src[r * sstride + c] << 16);
}

// norm2[i] = sum_j rows[i][j]^2, sequential in j (one lane per row, rows staged through LDS).
__global__ __launch_bounds__(64) void row_norm2_kernel(const float* __restrict__ rows, uint64_t n,
                                                       uint32_t d, uint64_t stride,
                                                       float* __restrict__ norm2) {
  extern __shared__ __align__(16) unsigned char smem[];
  float* tile = reinterpret_cast<float*>(smem);
  const int lane = threadIdx.x;
  for (uint64_t base = (uint64_t)blockIdx.x * 64; base < n; base += (uint64_t)gridDim.x * 64) {
    uint32_t R = (uint32_t)(n - base < 64 ? n - base : 64);
    // the query operand is ignored by SUMSQ_RAW; `tile` only serves as a valid LDS address
    float v = isl_dev::wave_distances<isl_dev::METRIC_SUMSQ_RAW>(rows + base * stride, stride, d,
                                                                  (uint32_t)lane, R, tile, tile, 0.f);
    if ((uint32_t)lane < R) norm2[base + lane] = v;
  }
}

// One thread per row: max degree, duplicate ids within a row (flag bit 1).
__global__ void row_stats_kernel(const uint64_t* __restrict__ off, const uint32_t* __restrict__ adj,
                                 uint64_t num_nodes, uint32_t* __restrict__ flags,
                                 uint32_t* __restrict__ max_degree) {
  uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= num_nodes) return;
  uint64_t s = off[row], e = off[row + 1];
  uint64_t deg = e - s;
  atomicMax(max_degree, (uint32_t)(deg > 0xFFFFFFFFull ? 0xFFFFFFFFu : deg));
  if (deg > 4096) { atomicOr(flags, 4u); return; }  // checked on the host instead
  bool dup = false;
  for (uint64_t i = s + 1; i < e && !dup; i++) {
    uint32_t v = adj[i];
    for (uint64_t j = s; j < i; j++)
      if (adj[j] == v) { dup = true; break; }
  }
  if (dup) atomicOr(flags, 2u);
}

}  // namespace

namespace isl {

isl_status materialise_host_csr(const isl_index* cidx) {
  isl_index* idx = const_cast<isl_index*>(cidx);
  if (idx->host_csr_valid) return ISL_OK;
  std::lock_guard<std::mutex> lock(idx->mu);
  if (idx->host_csr_valid) return ISL_OK;
  ISL_TRY(use_device(idx->device));
  idx->node_offsets.resize(idx->num_nodes + 1);
  ISL_HIP(hipMemcpy(idx->node_offsets.data(), idx->d_off, (idx->num_nodes + 1) * 8,
                    hipMemcpyDeviceToHost));
  std::vector<uint32_t> tmp(idx->nnz);
  if (idx->nnz)
    ISL_HIP(hipMemcpy(tmp.data(), idx->d_adj, idx->nnz * 4, hipMemcpyDeviceToHost));
  idx->neighbors.assign(tmp.begin(), tmp.end());
  if (idx->levels.size() != idx->num_nodes) idx->levels.assign(idx->num_nodes, 0);
  idx->degree_counts.resize(idx->num_nodes);
  for (uint64_t i = 0; i < idx->num_nodes; i++)
    idx->degree_counts[i] = idx->node_offsets[i + 1] - idx->node_offsets[i];
  idx->host_csr_valid = true;
  return ISL_OK;
}

static isl_status scan_device_csr(isl_index* idx, uint32_t* d_flags, uint32_t* flags_out) {
  uint32_t* d_maxdeg = d_flags + 1;
  if (idx->num_nodes) {
    uint32_t blocks = (uint32_t)((idx->num_nodes + 255) / 256);
    hipLaunchKernelGGL(row_stats_kernel, dim3(blocks), dim3(256), 0, 0, idx->d_off, idx->d_adj,
                       idx->num_nodes, d_flags, d_maxdeg);
    ISL_HIP(hipGetLastError());
  }
  uint32_t host[2] = {0, 0};
  ISL_HIP(hipMemcpy(host, d_flags, 8, hipMemcpyDeviceToHost));
  *flags_out = host[0];
  idx->max_degree = host[1];
  return ISL_OK;
}

}  // namespace isl

extern "C" {

isl_status isl_index_upload(isl_index* idx, int32_t device) {
  if (!idx) return fail(ISL_ERR_INVALID_ARGUMENT, "index is NULL");
  ISL_TRY(use_device(device));
  std::lock_guard<std::mutex> lock(idx->mu);
  if (idx->device >= 0 && idx->device != device)
    return fail(ISL_ERR_UNSUPPORTED, "index already resident on device %d", idx->device);
  if (idx->d_off) return ISL_OK;  // already uploaded
  if (!idx->host_csr_valid) return fail(ISL_ERR_INVALID_ARGUMENT, "no host CSR to upload");
  if (idx->num_nodes >= kMaxDeviceId)
    return fail(ISL_ERR_UNSUPPORTED, "num_nodes %llu exceeds the device id range",
                (unsigned long long)idx->num_nodes);
  idx->device = device;
  uint64_t nnz = idx->num_nodes ? idx->node_offsets[idx->num_nodes] : 0;

  // Duplicate ids inside one adjacency row are always "already visited" at their
  // second occurrence (visited.insert, leann.rs:933-937), so the device copy keeps
  // only the first occurrence; the host CSR is left untouched.
  std::vector<uint64_t> off = idx->node_offsets;
  const uint64_t* nb_src = idx->neighbors.data();
  std::vector<uint64_t> dedup;

  uint32_t* d_flags = nullptr;
  ISL_HIP(hipMalloc(&d_flags, 16));
  ISL_HIP(hipMemset(d_flags, 0, 16));
  for (int pass = 0; pass < 2; pass++) {
    ISL_HIP(hipMalloc(&idx->d_off, (idx->num_nodes + 1) * 8));
    ISL_HIP(hipMemcpy(idx->d_off, off.data(), (idx->num_nodes + 1) * 8, hipMemcpyHostToDevice));
    ISL_HIP(hipMalloc(&idx->d_adj, (nnz ? nnz : 1) * 4));
    if (nnz) {
      uint64_t* d_tmp = nullptr;
      ISL_HIP(hipMalloc(&d_tmp, nnz * 8));
      ISL_HIP(hipMemcpy(d_tmp, nb_src, nnz * 8, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(convert_adj_kernel, dim3(2048), dim3(256), 0, 0, d_tmp, idx->d_adj, nnz,
                         d_flags);
      ISL_HIP(hipGetLastError());
      ISL_HIP(hipDeviceSynchronize());
      ISL_HIP(hipFree(d_tmp));
    }
    idx->nnz = nnz;
    uint32_t flags = 0;
    ISL_TRY(scan_device_csr(idx, d_flags, &flags));
    if (flags & 1u) {
      (void)hipFree(d_flags);
      return fail(ISL_ERR_UNSUPPORTED, "neighbour ids above 0x%x are not representable on device",
                  kMaxDeviceId);
    }
    if (!(flags & 6u) || pass == 1) break;
    // rebuild without in-row duplicates on the host, then upload again
    (void)hipFree(idx->d_off);
    (void)hipFree(idx->d_adj);
    idx->d_off = nullptr;
    idx->d_adj = nullptr;
    ISL_HIP(hipMemset(d_flags, 0, 16));
    dedup.clear();
    dedup.reserve(nnz);
    std::vector<uint64_t> row;
    for (uint64_t i = 0; i < idx->num_nodes; i++) {
      uint64_t s = idx->node_offsets[i], e = idx->node_offsets[i + 1];
      row.assign(idx->neighbors.begin() + s, idx->neighbors.begin() + e);
      std::vector<uint64_t> sorted = row;
      std::sort(sorted.begin(), sorted.end());
      bool has_dup = std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end();
      off[i] = dedup.size();
      if (!has_dup) {
        dedup.insert(dedup.end(), row.begin(), row.end());
      } else {
        size_t base = dedup.size();
        for (uint64_t v : row) {
          bool seen = false;
          for (size_t j = base; j < dedup.size(); j++)
            if (dedup[j] == v) { seen = true; break; }
          if (!seen) dedup.push_back(v);
        }
      }
    }
    off[idx->num_nodes] = dedup.size();
    nnz = dedup.size();
    nb_src = dedup.data();
  }
  (void)hipFree(d_flags);
  return ensure_padded_adjacency(idx);
}

isl_status isl_index_from_device_csr(const isl_leann_config* cfg, int32_t device,
                                     uint64_t num_nodes, const uint64_t* d_node_offsets,
                                     const uint32_t* d_neighbors, int32_t has_entry,
                                     uint64_t entry_point, int32_t has_dimension,
                                     uint64_t dimension, isl_index** out) {
  if (!out || !d_node_offsets) return fail(ISL_ERR_INVALID_ARGUMENT, "NULL argument");
  ISL_TRY(use_device(device));
  if (num_nodes >= kMaxDeviceId)
    return fail(ISL_ERR_UNSUPPORTED, "num_nodes exceeds the device id range");
  isl_index* idx = nullptr;
  ISL_TRY(isl_index_new(cfg, &idx));
  idx->host_csr_valid = false;
  idx->node_offsets.clear();
  idx->num_nodes = num_nodes;
  idx->has_entry = has_entry != 0;
  idx->entry_point = has_entry ? entry_point : 0;
  idx->has_dimension = has_dimension != 0;
  idx->dimension = has_dimension ? dimension : 0;
  idx->device = device;
  auto bail = [&](isl_status st) { isl_index_free(idx); return st; };
  uint64_t nnz = 0;
  if (hipMemcpy(&nnz, d_node_offsets + num_nodes, 8, hipMemcpyDeviceToHost) != hipSuccess)
    return bail(fail(ISL_ERR_DEVICE, "cannot read node_offsets[num_nodes] from the device"));
  idx->nnz = nnz;
  if (hipMalloc(&idx->d_off, (num_nodes + 1) * 8) != hipSuccess ||
      hipMalloc(&idx->d_adj, (nnz ? nnz : 1) * 4) != hipSuccess)
    return bail(fail(ISL_ERR_DEVICE, "hipMalloc failed for the CSR copy"));
  if (hipMemcpy(idx->d_off, d_node_offsets, (num_nodes + 1) * 8, hipMemcpyDeviceToDevice) !=
          hipSuccess ||
      (nnz && hipMemcpy(idx->d_adj, d_neighbors, nnz * 4, hipMemcpyDeviceToDevice) != hipSuccess))
    return bail(fail(ISL_ERR_DEVICE, "device copy of the CSR failed"));
  uint32_t* d_flags = nullptr;
  if (hipMalloc(&d_flags, 16) != hipSuccess || hipMemset(d_flags, 0, 16) != hipSuccess)
    return bail(fail(ISL_ERR_DEVICE, "hipMalloc failed"));
  uint32_t flags = 0;
  isl_status st = scan_device_csr(idx, d_flags, &flags);
  (void)hipFree(d_flags);
  if (st != ISL_OK) return bail(st);
  if (flags & 6u)
    return bail(fail(ISL_ERR_UNSUPPORTED,
                     "device-born CSR rows must not repeat a neighbour id (or exceed 4096 ids)"));
  if ((st = ensure_padded_adjacency(idx)) != ISL_OK) return bail(st);
  *out = idx;
  return ISL_OK;
}

// InMemoryEmbeddingProvider::new, leann.rs:111-120
isl_status isl_set_embeddings(isl_index* idx, const void* rows, uint64_t n, uint64_t d,
                              int32_t dtype, int32_t mem) {
  if (!idx || (!rows && n)) return fail(ISL_ERR_INVALID_ARGUMENT, "NULL argument");
  if (n == 0) return fail(ISL_ERR_EMPTY_COLLECTION, "Empty vector collection");
  if (dtype != ISL_DTYPE_F32 && dtype != ISL_DTYPE_BF16)
    return fail(ISL_ERR_INVALID_ARGUMENT, "unknown row dtype");
  if (d == 0 || d > 65536) return fail(ISL_ERR_INVALID_ARGUMENT, "dimension out of range");
  if (idx->device < 0)
    return fail(ISL_ERR_DEVICE, "call isl_index_upload before attaching embeddings");
  ISL_TRY(use_device(idx->device));
  std::lock_guard<std::mutex> lock(idx->mu);
  if (any_lane_busy(idx))  // their kernels read the tables freed below
    return fail(ISL_ERR_SEARCH, "Search error: the embedding provider cannot be swapped while searches are in flight");
  if (idx->d_emb) { (void)hipFree(idx->d_emb); idx->d_emb = nullptr; }
  if (idx->d_emb16) { (void)hipFree(idx->d_emb16); idx->d_emb16 = nullptr; }
  free_exact_pool(idx->pool);  // sized by the row count: rebuilt by the next prepare / search
  idx->recompute = false;  // back to the in-memory provider
  if (dtype == ISL_DTYPE_BF16) {
    // rows are bf16 bit patterns; the provider's vectors are their exact f32 images.  Rows start
    // 16-byte aligned (stride = d rounded up to 8 elements), 1 KiB of slack like the f32 table.
    if (idx->is_hnsw) return fail(ISL_ERR_UNSUPPORTED, "the HnswGraph facade keeps f32 vectors");
    const uint64_t stride16 = (d + 7) / 8 * 8;
    const size_t bytes16 = (size_t)(n * stride16 + 512) * 2;
    ISL_HIP(hipMalloc(&idx->d_emb16, bytes16));
    ISL_HIP(hipMemset(idx->d_emb16, 0, bytes16));
    hipMemcpyKind kind16 = mem == ISL_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    ISL_HIP(hipMemcpy2D(idx->d_emb16, stride16 * 2, rows, d * 2, d * 2, n, kind16));
    idx->nvec = n;
    idx->emb_d = d;
    idx->emb_stride = stride16;
    if (idx->d_norm2) { (void)hipFree(idx->d_norm2); idx->d_norm2 = nullptr; }
    ISL_HIP(hipMalloc(&idx->d_norm2, (size_t)n * 4));
    // norm_b in the reference's order: widen a chunk of rows to f32 and reuse the f32 kernel
    using namespace isl_dev;
    const uint64_t stride32 = (d + 3) / 4 * 4;
    const uint64_t chunk = std::max<uint64_t>(1, std::min<uint64_t>(n, (256ull << 20) / (stride32 * 4)));
    float* tmp = nullptr;
    ISL_HIP(hipMalloc(&tmp, (size_t)(chunk * stride32 + 256) * 4));
    ISL_HIP(hipMemset(tmp, 0, (size_t)(chunk * stride32 + 256) * 4));
    const size_t lds = (size_t)TILE_ROWS * TILE_LD * 4 + 64;
    for (uint64_t o = 0; o < n; o += chunk) {
      const uint64_t c = std::min(chunk, n - o);
      hipLaunchKernelGGL(widen_bf16_kernel, dim3((uint32_t)((c * d + 255) / 256)), dim3(256), 0, 0,
                         idx->d_emb16 + o * stride16, stride16, (uint32_t)d, c, tmp, stride32);
      hipLaunchKernelGGL(row_norm2_kernel, dim3((uint32_t)std::min<uint64_t>((c + 63) / 64, 8192)), dim3(64), lds,
                         0, tmp, c, (uint32_t)d, stride32, idx->d_norm2 + o);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(ISL_ERR_DEVICE, "bf16 row upload failed: %s", hipGetErrorString(e));
    return ISL_OK;
  }
  uint64_t stride = (d + 3) / 4 * 4;  // rows start 16-byte aligned
  size_t bytes = (size_t)(n * stride + 256) * sizeof(float);  // slack for whole-slab reads
  ISL_HIP(hipMalloc(&idx->d_emb, bytes));
  if (stride != d) ISL_HIP(hipMemset(idx->d_emb, 0, bytes));
  else ISL_HIP(hipMemset(idx->d_emb + n * stride, 0, 256 * sizeof(float)));
  hipMemcpyKind kind = mem == ISL_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  if (stride == d) ISL_HIP(hipMemcpy(idx->d_emb, rows, (size_t)n * d * 4, kind));
  else ISL_HIP(hipMemcpy2D(idx->d_emb, stride * 4, rows, d * 4, d * 4, n, kind));
  idx->nvec = n;
  idx->emb_d = d;
  idx->emb_stride = stride;
  // norm_b of cosine_distance (distance.rs:79) depends on the row alone: computed once, in the
  // reference's left-to-right order, and reused by every search
  if (idx->d_norm2) { (void)hipFree(idx->d_norm2); idx->d_norm2 = nullptr; }
  ISL_HIP(hipMalloc(&idx->d_norm2, (size_t)n * 4));
  {
    using namespace isl_dev;
    size_t lds = (size_t)TILE_ROWS * TILE_LD * 4 + 64;
    uint32_t grid = (uint32_t)std::min<uint64_t>((n + 63) / 64, 8192);
    hipLaunchKernelGGL(row_norm2_kernel, dim3(grid), dim3(64), lds, 0, idx->d_emb, n, (uint32_t)d,
                       stride, idx->d_norm2);
    ISL_HIP(hipGetLastError());
    ISL_HIP(hipDeviceSynchronize());
  }
  return ISL_OK;
}

// (re)allocates the recompute provider's row cache: slab, per-slot norms, slot map, owners
static isl_status alloc_recompute_cache(isl_index* idx, uint64_t rows) {
  void* olds[] = {idx->d_emb, idx->d_norm2, idx->d_slot_of, idx->d_owner, idx->d_stamp, idx->d_slab_head};
  for (void* p : olds)
    if (p) (void)hipFree(p);
  idx->d_emb = nullptr; idx->d_norm2 = nullptr; idx->d_slot_of = nullptr; idx->d_owner = nullptr;
  idx->d_stamp = nullptr; idx->d_slab_head = nullptr;
  idx->slab_rows = 0;
  const uint64_t stride = idx->emb_stride, n = idx->nvec;
  ISL_HIP(hipMalloc(&idx->d_emb, (size_t)(rows * stride + 256) * sizeof(float)));
  ISL_HIP(hipMemset(idx->d_emb, 0, (size_t)(rows * stride + 256) * sizeof(float)));
  ISL_HIP(hipMalloc(&idx->d_norm2, (size_t)rows * 4));
  ISL_HIP(hipMemset(idx->d_norm2, 0, (size_t)rows * 4));
  ISL_HIP(hipMalloc(&idx->d_slot_of, (size_t)(n + 1) * 4));
  ISL_HIP(hipMemset(idx->d_slot_of, 0xFF, (size_t)(n + 1) * 4));
  ISL_HIP(hipMalloc(&idx->d_owner, (size_t)rows * 4));
  ISL_HIP(hipMemset(idx->d_owner, 0xFF, (size_t)rows * 4));
  ISL_HIP(hipMalloc(&idx->d_stamp, (size_t)rows * 4));
  ISL_HIP(hipMemset(idx->d_stamp, 0, (size_t)rows * 4));
  ISL_HIP(hipMalloc(&idx->d_slab_head, 8));  // [0] clock hand, [1] slots used so far
  ISL_HIP(hipMemset(idx->d_slab_head, 0, 8));
  idx->slab_rows = rows;
  return ISL_OK;
}

isl_status isl_index_set_recompute_cache_rows(isl_index* idx, uint64_t rows) {
  if (!idx) return fail(ISL_ERR_INVALID_ARGUMENT, "index is NULL");
  if (!idx->recompute) return fail(ISL_ERR_INVALID_ARGUMENT, "the index has no recompute provider");
  ISL_TRY(use_device(idx->device));
  std::lock_guard<std::mutex> lock(idx->mu);
  if (any_lane_busy(idx))
    return fail(ISL_ERR_SEARCH, "Search error: the row cache cannot be resized while searches are in flight");
  // a hop of one query needs up to 128 rows at once (plus the entry point): that is the floor
  rows = std::min<uint64_t>(idx->nvec, std::max<uint64_t>(rows, 256));
  return alloc_recompute_cache(idx, rows);
}

uint64_t isl_index_recompute_cache_bytes(const isl_index* idx) {
  if (!idx || !idx->recompute) return 0;
  return (idx->slab_rows * idx->emb_stride + 256) * 4 + idx->slab_rows * 12 + (idx->nvec + 1) * 4;
}

// EmbeddingProvider backed by the encoder (recompute mode), see islands_amd.h.
isl_status isl_set_recompute_provider(isl_index* idx, isl_encoder* enc, const uint16_t* tokens,
                                      const uint16_t* lengths, uint64_t n, uint64_t L,
                                      int32_t normalize, int32_t keep_rows, int32_t mem) {
  if (!idx || !enc || (!tokens && n)) return fail(ISL_ERR_INVALID_ARGUMENT, "NULL argument");
  if (n == 0) return fail(ISL_ERR_EMPTY_COLLECTION, "Empty vector collection");
  if (idx->device < 0) return fail(ISL_ERR_DEVICE, "call isl_index_upload before attaching a provider");
  if (enc->device != idx->device) return fail(ISL_ERR_INVALID_ARGUMENT, "encoder and index live on different devices");
  if (idx->is_hnsw) return fail(ISL_ERR_UNSUPPORTED, "the HnswGraph facade stores its vectors (hnsw.rs:97-112)");
  if (L == 0 || L > enc->cfg.max_position)
    return fail(ISL_ERR_EMBEDDING, "Embedding error: %llu token slots per node, the model takes 1..%u",
                (unsigned long long)L, enc->cfg.max_position);
  if (n > 0x7FFFFFF0ull) return fail(ISL_ERR_UNSUPPORTED, "node ids above the device id range");
  ISL_TRY(use_device(idx->device));
  std::lock_guard<std::mutex> lock(idx->mu);
  if (any_lane_busy(idx))
    return fail(ISL_ERR_SEARCH, "Search error: the embedding provider cannot be swapped while searches are in flight");
  const uint64_t d = enc->cfg.hidden, stride = (d + 3) / 4 * 4;
  // (d_emb16 too: bf16 rows of an earlier in-memory provider would otherwise stay the table the
  // searches read)
  void* olds[] = {idx->d_emb, idx->d_emb16, idx->d_norm2, idx->d_tokens, idx->d_lens, idx->d_slot_of, idx->d_owner,
                  idx->d_stamp, idx->d_slab_head};
  for (void* p : olds)
    if (p) (void)hipFree(p);
  idx->d_emb = nullptr; idx->d_emb16 = nullptr; idx->d_norm2 = nullptr; idx->d_tokens = nullptr;
  idx->d_lens = nullptr; idx->d_slot_of = nullptr; idx->d_owner = nullptr; idx->d_stamp = nullptr;
  idx->d_slab_head = nullptr;
  free_exact_pool(idx->pool);
  idx->recompute = false;
  // Recompute mode does not store embeddings (leann.rs:366-371): what is resident is the token
  // table plus a BOUNDED row cache -- a slab of slab_rows rows, 4 bytes of slot map per node.  The
  // default slab (2^20 rows, or every node of a smaller index) holds what a batch of a thousand
  // queries visits; isl_index_set_recompute_cache_rows changes it.
  idx->nvec = n;
  idx->emb_d = d;
  idx->emb_stride = stride;
  ISL_TRY(alloc_recompute_cache(idx, std::min<uint64_t>(n, 1ull << 20)));
  hipMemcpyKind kind = mem == ISL_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  ISL_HIP(hipMalloc(&idx->d_tokens, (size_t)n * L * 2));
  ISL_HIP(hipMemcpy(idx->d_tokens, tokens, (size_t)n * L * 2, kind));
  if (lengths) {
    ISL_HIP(hipMalloc(&idx->d_lens, (size_t)n * 2));
    ISL_HIP(hipMemcpy(idx->d_lens, lengths, (size_t)n * 2, kind));
  }
  idx->enc = enc;
  idx->tok_L = (uint32_t)L;
  idx->nvec = n;
  idx->emb_d = d;
  idx->emb_stride = stride;
  idx->enc_normalize = normalize;
  idx->keep_rows = keep_rows != 0;
  idx->recompute = true;
  idx->cfg.is_recompute = 1;
  if (!idx->has_dimension) { idx->has_dimension = true; idx->dimension = d; }
  return ISL_OK;
}

}  // extern "C"
