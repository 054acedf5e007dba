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
  // recompute mode: EmbeddingProvider backed by the encoder (leann.rs:82-99); the embedder is
  // borrowed and must outlive the index
  void attach_recompute(isl_encoder* enc, const std::vector<uint16_t>& tokens, uint64_t n, uint64_t L,
                        bool normalize = true, bool keep_rows = false) {
    check(isl_set_recompute_provider(h_, enc, tokens.data(), nullptr, n, L, normalize ? 1 : 0,
                                     keep_rows ? 1 : 0, ISL_MEM_HOST));
  }
  // no reference counterpart: sets up every search lane ahead of time (isl_index_prepare)
  void prepare(uint64_t max_nq, uint64_t max_ef, uint64_t max_k = 10, int32_t lanes = 8) {
    check(isl_index_prepare(h_, max_nq, max_ef, max_k, lanes));
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
  // extension: the two-level search leann.rs:855-857 asks for (docs/leann-specification.md:223-275);
  // codes = ProductQuantizer::encode of every node, [n][m]; the quantizer is borrowed
  void attach_pq_codes(const isl_pq* pq, const std::vector<uint16_t>& codes, uint64_t n) {
    check(isl_index_set_pq_codes(h_, pq, codes.data(), n, ISL_MEM_HOST));
  }
  std::vector<std::pair<uint64_t, float>> search_two_level(const std::vector<float>& query, uint64_t k,
                                                           uint64_t ef, float rerank_ratio) const {
    std::vector<uint64_t> ids(k ? k : 1);
    std::vector<float> dist(k ? k : 1);
    uint32_t cnt = 0;
    check(isl_search_two_level_batch(h_, query.data(), 1, query.size(), k, ef, rerank_ratio, ids.data(),
                                     dist.data(), &cnt));
    std::vector<std::pair<uint64_t, float>> out;
    for (uint32_t i = 0; i < cnt; i++) out.emplace_back(ids[i], dist[i]);
    return out;
  }
  isl_index* handle() const { return h_; }

 private:
  explicit LeannIndex(std::nullptr_t) {}
  isl_index* h_ = nullptr;
};

// ---- hnsw.rs / search.rs facade ------------------------------------------------------------
// SearchResult, src/core/search.rs:54-103
struct SearchResult {
  uint64_t id = 0;
  float score = 0.f;
  std::optional<std::vector<float>> vector;
  std::optional<std::string> text;
  float to_similarity() const { return 1.0f / (1.0f + score); }  // search.rs:100-102
};

// SearchConfig, src/core/search.rs:8-52
struct SearchConfig {
  uint64_t top_k = 10, ef = 100;
  bool include_vectors = false, include_metadata = true;
  std::optional<float> min_similarity;
  static SearchConfig fast(uint64_t k) { SearchConfig c; c.top_k = k; c.ef = k * 2; return c; }
  static SearchConfig accurate(uint64_t k) { SearchConfig c; c.top_k = k; c.ef = k * 10; return c; }
};

// Search side of HnswGraph, src/core/hnsw.rs:149-515.  Insertion (hnsw.rs:214-329) is outside
// the search path: the graph is handed over layer by layer (layers[L][node] = neighbour ids).
class HnswGraph {
 public:
  HnswGraph(const std::vector<float>& vectors, uint64_t dim,
            const std::vector<std::vector<std::vector<uint64_t>>>& layers,
            const std::vector<uint64_t>& levels, std::optional<uint64_t> entry_point, uint64_t max_level,
            uint64_t m = 16, uint64_t m0 = 32, uint64_t ef_construction = 200,
            DistanceMetric metric = DistanceMetric::Cosine, int32_t device = 0)
      : vectors_(vectors), dim_(dim) {
    const uint64_t n = levels.size();
    std::vector<std::vector<uint64_t>> offs(layers.size()), adjs(layers.size());
    std::vector<const uint64_t*> po, pa;
    for (size_t L = 0; L < layers.size(); L++) {
      offs[L].assign(1, 0);
      for (uint64_t i = 0; i < n; i++) {
        adjs[L].insert(adjs[L].end(), layers[L][i].begin(), layers[L][i].end());
        offs[L].push_back(adjs[L].size());
      }
      if (adjs[L].empty()) adjs[L].push_back(0);
      po.push_back(offs[L].data());
      pa.push_back(adjs[L].data());
    }
    check(isl_hnsw_from_layers(m, m0, ef_construction, (int32_t)metric, n, dim, layers.size(), po.data(),
                               pa.data(), n ? levels.data() : nullptr, entry_point ? 1 : 0,
                               entry_point.value_or(0), max_level, n ? vectors.data() : nullptr, device, &h_));
  }
  // HnswGraph::from_bytes, hnsw.rs:511-514 (bincode image; entries in any HashMap order)
  static HnswGraph from_bytes(const std::vector<uint8_t>& bytes, int32_t device = 0) {
    HnswGraph g;
    check(isl_hnsw_from_bytes(bytes.data(), bytes.size(), device, &g.h_));
    return g;
  }
  HnswGraph(HnswGraph&& o) noexcept : h_(o.h_), vectors_(std::move(o.vectors_)), dim_(o.dim_) { o.h_ = nullptr; }
  HnswGraph(const HnswGraph&) = delete;
  HnswGraph& operator=(const HnswGraph&) = delete;
  ~HnswGraph() { isl_hnsw_free(h_); }
  uint64_t len() const { return isl_hnsw_len(h_); }
  bool is_empty() const { return len() == 0; }
  // HnswGraph::search, hnsw.rs:458-504
  std::vector<std::pair<uint64_t, float>> search(const std::vector<float>& query, uint64_t k, uint64_t ef) const {
    return search_batch(query, 1, k, ef)[0];
  }
  std::vector<std::vector<std::pair<uint64_t, float>>> search_batch(const std::vector<float>& queries,
                                                                    uint64_t nq, uint64_t k, uint64_t ef) const {
    std::vector<uint64_t> ids(nq * (k ? k : 1));
    std::vector<float> dist(nq * (k ? k : 1));
    std::vector<uint32_t> cnt(nq);
    check(isl_hnsw_search_batch(h_, queries.data(), nq, nq ? queries.size() / nq : 0, k, ef, ids.data(),
                                dist.data(), cnt.data()));
    std::vector<std::vector<std::pair<uint64_t, float>>> out(nq);
    for (uint64_t q = 0; q < nq; q++)
      for (uint32_t i = 0; i < cnt[q]; i++) out[q].emplace_back(ids[q * k + i], dist[q * k + i]);
    return out;
  }
  std::optional<std::vector<float>> get_vector(uint64_t id) const {  // get_node(id).vector, hnsw.rs:507-510
    if (id >= len() || vectors_.empty()) return std::nullopt;  // a graph read by from_bytes keeps no host copy
    return std::vector<float>(vectors_.begin() + id * dim_, vectors_.begin() + (id + 1) * dim_);
  }

 private:
  HnswGraph() = default;
  isl_hnsw* h_ = nullptr;
  std::vector<float> vectors_;
  uint64_t dim_ = 0;
};

// Searcher, src/core/search.rs:105-182; search_batch is one device launch instead of the
// reference's sequential map (:179-181).
class Searcher {
 public:
  explicit Searcher(const HnswGraph& g, SearchConfig c = SearchConfig()) : graph_(g), config_(c) {}
  Searcher& top_k(uint64_t k) { config_.top_k = k; return *this; }
  Searcher& ef(uint64_t e) { config_.ef = e; return *this; }
  Searcher& include_vectors() { config_.include_vectors = true; return *this; }
  Searcher& min_similarity(float t) { config_.min_similarity = t; return *this; }
  std::vector<SearchResult> search(const std::vector<float>& query) const { return search_batch(query, 1)[0]; }
  std::vector<std::vector<SearchResult>> search_batch(const std::vector<float>& queries, uint64_t nq) const {
    auto raw = graph_.search_batch(queries, nq, config_.top_k, config_.ef);
    std::vector<std::vector<SearchResult>> out(nq);
    for (uint64_t q = 0; q < nq; q++)
      for (auto& [id, d] : raw[q]) {
        SearchResult r;
        r.id = id;
        r.score = d;
        if (config_.include_vectors) r.vector = graph_.get_vector(id);
        if (!config_.min_similarity || r.to_similarity() >= *config_.min_similarity) out[q].push_back(std::move(r));
      }
    return out;
  }

 private:
  const HnswGraph& graph_;
  SearchConfig config_;
};

// ---- MultiIndexSearcher over id-range shards, one rank per GPU (search.rs:211-237) ------------
// ShardGroup = the ranks' communicator: RCCL from a unique id rank 0 made (unique_id(), carried to the
// other ranks by the host's own means), or a blocking host all-gather callback.
class ShardGroup {
 public:
  static std::vector<uint8_t> unique_id() {
    std::vector<uint8_t> id(ISL_SHARD_UNIQUE_ID_BYTES);
    check(isl_shard_unique_id(id.data()));
    return id;
  }
  ShardGroup(int32_t device, int32_t world, int32_t rank, const std::vector<uint8_t>& id) {
    check(isl_shard_group_create(device, world, rank, id.data(), &h_));
  }
  ShardGroup(int32_t device, int32_t world, int32_t rank, isl_shard_allgather_fn fn, void* user) {
    check(isl_shard_group_create_host(device, world, rank, fn, user, &h_));
  }
  ShardGroup(const ShardGroup&) = delete;
  ShardGroup& operator=(const ShardGroup&) = delete;
  ~ShardGroup() { isl_shard_group_free(h_); }
  int32_t comm_ranks() const { int32_t n = 0; check(isl_shard_group_info(h_, nullptr, nullptr, &n, nullptr)); return n; }
  isl_shard_group* handle() const { return h_; }

 private:
  isl_shard_group* h_ = nullptr;
};

struct ShardedResult { uint64_t id; float score; uint32_t shard; };

class ShardedSearcher {
 public:
  // `shard`: this rank's index over its id range (local ids); group = nullptr: a single shard
  ShardedSearcher(const LeannIndex& shard, ShardGroup* group, uint64_t n_total, int32_t depth = 8) {
    check(isl_sharded_searcher_new(shard.handle(), group ? group->handle() : nullptr, n_total, nullptr, depth, &h_));
  }
  ShardedSearcher(const ShardedSearcher&) = delete;
  ShardedSearcher& operator=(const ShardedSearcher&) = delete;
  ~ShardedSearcher() { isl_sharded_searcher_free(h_); }
  void prepare(uint64_t nq, uint64_t k, uint64_t ef) { check(isl_sharded_prepare(h_, nq, k, ef)); }
  // the same batch on every rank; every rank gets the merged answer (global ids, ascending, ties -> lower shard)
  std::vector<std::vector<ShardedResult>> search_batch(const std::vector<float>& queries, uint64_t nq, uint64_t k,
                                                       uint64_t ef) {
    std::vector<uint64_t> ids(nq * k);
    std::vector<float> dist(nq * k);
    std::vector<uint32_t> src(nq * k), cnt(nq);
    check(isl_sharded_search_batch(h_, queries.data(), nq, nq ? queries.size() / nq : 0, k, ef, ids.data(), dist.data(),
                                   src.data(), cnt.data()));
    std::vector<std::vector<ShardedResult>> out(nq);
    for (uint64_t q = 0; q < nq; q++)
      for (uint32_t j = 0; j < cnt[q]; j++) out[q].push_back({ids[q * k + j], dist[q * k + j], src[q * k + j]});
    return out;
  }
  isl_sharded_searcher* handle() const { return h_; }

 private:
  isl_sharded_searcher* h_ = nullptr;
};

// ---- embedding/candle_provider.rs ---------------------------------------------------------
// CandleEmbedder after tokenisation (candle_provider.rs:226-507): weights by checkpoint tensor
// name, embed = embed_texts_raw on padded token ids.
class CandleEmbedder {
 public:
  CandleEmbedder(const isl_bert_config& cfg, bool normalize = true, int32_t device = 0)
      : cfg_(cfg), normalize_(normalize) { check(isl_encoder_new(&cfg, device, &h_)); }
  CandleEmbedder(const CandleEmbedder&) = delete;
  CandleEmbedder& operator=(const CandleEmbedder&) = delete;
  ~CandleEmbedder() { isl_encoder_free(h_); }
  void set_weight(const std::string& name, const std::vector<float>& v) {
    check(isl_encoder_set_weight(h_, name.c_str(), v.data(), v.size(), ISL_MEM_HOST));
  }
  uint64_t dimension() const { return cfg_.hidden; }
  // ids / mask: B rows of L entries (padding: id 0, mask 0, candle_provider.rs:385-402)
  std::vector<float> embed(const std::vector<int64_t>& ids, const std::vector<float>& mask, uint64_t B,
                           uint64_t L) const {
    std::vector<float> out(B * cfg_.hidden);
    check(isl_encoder_embed(h_, ids.data(), nullptr, mask.empty() ? nullptr : mask.data(), B, L,
                            normalize_ ? 1 : 0, out.data(), ISL_MEM_HOST, nullptr));
    return out;
  }
  isl_encoder* handle() const { return h_; }

 private:
  isl_encoder* h_ = nullptr;
  isl_bert_config cfg_;
  bool normalize_;
};

}  // namespace islands::core
