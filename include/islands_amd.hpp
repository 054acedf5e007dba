// islands_amd.hpp -- C++ host-side mirror of `islands::core` over the C ABI (islands_amd.h).
//
// The reference is compiled code (Rust) and its toolchain is absent from the build image, so
// the host side above the C ABI is mirrored in C++: same type and method names, argument
// meaning and error behaviour as src/core/{leann,distance,error}.rs.  Header-only; link with
// -lislands_amd.  CoreError variants become the `kind` of one exception type.
#pragma once

#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "islands_amd.h"

namespace islands::core {

// CoreError, src/core/error.rs:9-62
struct CoreError : std::runtime_error {
  isl_status status;
  uint64_t expected, actual, node;
  CoreError(isl_status s, const std::string& msg, uint64_t e = 0, uint64_t a = 0, uint64_t n = 0)
      : std::runtime_error(msg), status(s), expected(e), actual(a), node(n) {}
  std::string kind() const { return isl_status_name(status); }
};

inline void check(isl_status s) {
  if (s != ISL_OK)
    throw CoreError(s, isl_last_error_message(), isl_last_error_expected(),
                    isl_last_error_actual(), isl_last_error_node());
}

// DistanceMetric, src/core/distance.rs:9-19
enum class DistanceMetric : int32_t { Cosine = 0, Euclidean = 1, DotProduct = 2, Manhattan = 3 };
// PruningStrategy, src/core/leann.rs:168-178
enum class PruningStrategy : uint32_t { Global = 0, Local = 1, Proportional = 2 };

// Distance trait, src/core/distance.rs:22-35
inline float calculate(DistanceMetric m, const std::vector<float>& a, const std::vector<float>& b) {
  float out = 0;
  check(isl_distance((int32_t)m, a.data(), a.size(), b.data(), b.size(), &out));
  return out;
}
inline float calculate_squared(DistanceMetric m, const std::vector<float>& a,
                               const std::vector<float>& b) {
  float out = 0;
  check(isl_distance_squared((int32_t)m, a.data(), a.size(), b.data(), b.size(), &out));
  return out;
}
// rows: n contiguous rows of row_len floats
inline std::vector<float> batch_calculate(DistanceMetric m, const std::vector<float>& query,
                                          const std::vector<float>& rows, uint64_t row_len,
                                          int32_t device = 0) {
  uint64_t n = row_len ? rows.size() / row_len : 0;
  std::vector<float> out(n);
  check(isl_distance_batch((int32_t)m, query.data(), query.size(), rows.data(), n, row_len,
                           out.data(), ISL_MEM_HOST, device, nullptr));
  return out;
}

// LeannConfig, src/core/leann.rs:322-461
struct LeannConfig : isl_leann_config {
  LeannConfig() { isl_leann_config_paper_default(this); }
  static LeannConfig paper_default() { return LeannConfig(); }
  static LeannConfig fast() { LeannConfig c; isl_leann_config_fast(&c); return c; }
  static LeannConfig accurate() { LeannConfig c; isl_leann_config_accurate(&c); return c; }
  void validate() const { check(isl_leann_config_validate(this)); }
};

// CsrGraph, src/core/leann.rs:193-302 (public fields)
struct CsrGraph {
  std::vector<uint64_t> node_offsets{0};
  std::vector<uint64_t> neighbors;
  std::vector<uint64_t> levels;
  std::optional<uint64_t> entry_point;
  uint64_t max_level = 0;
  uint64_t num_nodes = 0;
  std::vector<uint64_t> degree_counts;

  uint64_t add_node(const std::vector<uint64_t>& nb, uint64_t level) {  // leann.rs:236-253
    uint64_t id = num_nodes++;
    levels.push_back(level);
    degree_counts.push_back(nb.size());
    neighbors.insert(neighbors.end(), nb.begin(), nb.end());
    node_offsets.push_back(neighbors.size());
    if (!entry_point || level > max_level) {
      entry_point = id;
      max_level = level;
    }
    return id;
  }
  std::optional<std::pair<const uint64_t*, size_t>> get_neighbors(uint64_t id) const {  // :225-233
    if (id >= num_nodes) return std::nullopt;
    return std::make_pair(neighbors.data() + node_offsets[id],
                          (size_t)(node_offsets[id + 1] - node_offsets[id]));
  }
  uint64_t storage_bytes() const {  // leann.rs:296-301
    return 8 * (node_offsets.size() + neighbors.size() + levels.size() + degree_counts.size());
  }
};

// InMemoryEmbeddingProvider, src/core/leann.rs:104-159 (row-major matrix)
struct InMemoryEmbeddingProvider {
  std::vector<float> embeddings;
  uint64_t dim = 0;
  InMemoryEmbeddingProvider(std::vector<float> rows, uint64_t dimension)
      : embeddings(std::move(rows)), dim(dimension) {
    if (embeddings.empty() || dim == 0) throw CoreError(ISL_ERR_EMPTY_COLLECTION, "Empty vector collection");
  }
  uint64_t dimension() const { return dim; }
  uint64_t len() const { return embeddings.size() / dim; }
};

// LeannIndex, src/core/leann.rs:492-1067
class LeannIndex {
 public:
  explicit LeannIndex(const LeannConfig& cfg = LeannConfig()) { check(isl_index_new(&cfg, &h_)); }
  static LeannIndex with_defaults() { return LeannIndex(); }
  static LeannIndex from_csr(const CsrGraph& g, const LeannConfig& cfg,
                             std::optional<uint64_t> dimension) {
    LeannIndex idx(nullptr);
    check(isl_index_from_csr(&cfg, g.num_nodes, g.node_offsets.data(), g.neighbors.data(),
                             g.levels.size() == g.num_nodes && g.num_nodes ? g.levels.data() : nullptr,
                             g.degree_counts.size() == g.num_nodes && g.num_nodes ? g.degree_counts.data() : nullptr,
                             g.entry_point ? 1 : 0, g.entry_point.value_or(0), g.max_level,
                             dimension ? 1 : 0, dimension.value_or(0), &idx.h_));
    return idx;
  }
  static LeannIndex from_bytes(const std::vector<uint8_t>& bytes) {  // leann.rs:1064
    LeannIndex idx(nullptr);
    check(isl_index_from_bytes(bytes.data(), bytes.size(), &idx.h_));
    return idx;
  }
  std::vector<uint8_t> to_bytes() const {  // leann.rs:1059
    uint8_t* p = nullptr;
    size_t n = 0;
    check(isl_index_to_bytes(h_, &p, &n));
    std::vector<uint8_t> out(p, p + n);
    isl_free_bytes(p);
    return out;
  }
  LeannIndex(LeannIndex&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
  LeannIndex& operator=(LeannIndex&& o) noexcept {
    if (this != &o) { isl_index_free(h_); h_ = o.h_; o.h_ = nullptr; }
    return *this;
  }
  LeannIndex(const LeannIndex&) = delete;
  LeannIndex& operator=(const LeannIndex&) = delete;
  ~LeannIndex() { isl_index_free(h_); }

  uint64_t len() const { return isl_index_len(h_); }
  bool is_empty() const { return isl_index_is_empty(h_) != 0; }
  std::optional<uint64_t> dimension() const {
    uint64_t d = 0;
    return isl_index_dimension(h_, &d) ? std::optional<uint64_t>(d) : std::nullopt;
  }
  uint64_t storage_bytes() const { return isl_index_storage_bytes(h_); }
  bool is_recompute() const { return isl_index_is_recompute(h_) != 0; }
  bool is_compact() const { return isl_index_is_compact(h_) != 0; }
  LeannConfig config() const { LeannConfig c; check(isl_index_config(h_, &c)); return c; }

  void upload(int32_t device = 0) { check(isl_index_upload(h_, device)); }
  void attach(const InMemoryEmbeddingProvider& p) {
    check(isl_set_embeddings(h_, p.embeddings.data(), p.len(), p.dim, ISL_DTYPE_F32, ISL_MEM_HOST));
  }
  // search / search_with_params, leann.rs:858-896
  std::vector<std::pair<uint64_t, float>> search(const std::vector<float>& query, uint64_t k) const {
    return search_with_params(query, k, config().ef_search);
  }
  std::vector<std::pair<uint64_t, float>> search_with_params(const std::vector<float>& query,
                                                             uint64_t k, uint64_t ef) const {
    std::vector<uint64_t> ids(k ? k : 1);
    std::vector<float> dist(k ? k : 1);
    uint32_t cnt = 0;
    check(isl_search_batch(h_, query.data(), 1, query.size(), k, ef, ids.data(), dist.data(), &cnt));
    std::vector<std::pair<uint64_t, float>> out;
    for (uint32_t i = 0; i < cnt; i++) out.emplace_back(ids[i], dist[i]);
    return out;
  }
  isl_index* handle() const { return h_; }

 private:
  explicit LeannIndex(std::nullptr_t) {}
  isl_index* h_ = nullptr;
};

}  // namespace islands::core
