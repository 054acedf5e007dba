// C++ mirror of the reference's inline tests for the path (src/core/leann.rs:1089-1572,
// src/core/distance.rs:148-440), written against include/islands_amd.hpp.
// `test_core_api cpu` runs the host-only part; `test_core_api gpu` adds the device part.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>

#include "islands_amd.hpp"

using namespace islands::core;

static int failures = 0;
#define EXPECT(cond)                                                        \
  do {                                                                      \
    if (!(cond)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); failures++; } \
  } while (0)

template <class F>
static bool throws(isl_status st, F f) {
  try { f(); } catch (const CoreError& e) { return e.status == st; }
  return false;
}

static std::vector<float> random_vectors(size_t n, size_t d, unsigned seed) {  // leann.rs:1078-1083 shape
  std::mt19937 rng(seed);
  std::uniform_real_distribution<float> u(-1.f, 1.f);
  std::vector<float> v(n * d);
  for (auto& x : v) x = u(rng);
  return v;
}

static void test_config() {  // leann.rs:1091-1143
  LeannConfig c = LeannConfig::paper_default();
  EXPECT(c.m == 30 && c.m0 == 60 && c.ef_construction == 128 && c.ef_search == 64);
  EXPECT(c.is_compact && c.is_recompute && c.high_degree_pruning);
  EXPECT(std::fabs(c.hub_percentile - 0.02f) < 1e-3f);
  c.validate();
  EXPECT(LeannConfig::fast().prune_ratio > 0 && LeannConfig::fast().m < 30);
  EXPECT(LeannConfig::accurate().m > 30 && LeannConfig::accurate().ef_construction > 128);
  LeannConfig bad; bad.m = 0;
  EXPECT(throws(ISL_ERR_INVALID_CONFIG, [&] { bad.validate(); }));
  bad = LeannConfig(); bad.m0 = 16;
  EXPECT(throws(ISL_ERR_INVALID_CONFIG, [&] { bad.validate(); }));
  bad = LeannConfig(); bad.prune_ratio = 1.5f;
  EXPECT(throws(ISL_ERR_INVALID_CONFIG, [&] { bad.validate(); }));
  bad = LeannConfig(); bad.beam_width = 0;
  EXPECT(throws(ISL_ERR_INVALID_CONFIG, [&] { bad.validate(); }));
}

static void test_csr_and_bytes() {  // leann.rs:1171-1217, 1347-1384
  CsrGraph g;
  EXPECT(g.num_nodes == 0 && !g.entry_point);
  EXPECT(g.add_node({}, 0) == 0 && *g.entry_point == 0);
  EXPECT(g.add_node({0}, 1) == 1 && *g.entry_point == 1);
  g.add_node({0, 1}, 0);
  EXPECT(g.get_neighbors(2)->second == 2 && !g.get_neighbors(999));
  LeannIndex idx = LeannIndex::from_csr(g, LeannConfig(), 16);
  EXPECT(idx.len() == 3 && *idx.dimension() == 16 && idx.storage_bytes() == g.storage_bytes());
  auto bytes = idx.to_bytes();
  LeannIndex r = LeannIndex::from_bytes(bytes);
  EXPECT(r.len() == idx.len() && r.dimension() == idx.dimension() && r.to_bytes() == bytes);
  bytes.resize(40);
  EXPECT(throws(ISL_ERR_DESERIALIZATION, [&] { LeannIndex::from_bytes(bytes); }));
  LeannIndex e = LeannIndex::with_defaults();  // leann.rs:1259-1267
  EXPECT(e.is_empty() && e.len() == 0 && !e.dimension() && e.is_recompute() && e.is_compact());
  EXPECT(throws(ISL_ERR_DIMENSION_MISMATCH, [&] { calculate(DistanceMetric::Cosine, {1, 2}, {1, 2, 3}); }));
}

static void test_gpu() {
  // distance.rs:150-229
  EXPECT(std::fabs(calculate(DistanceMetric::Cosine, {1, 2, 3}, {1, 2, 3})) < 1e-6f);
  EXPECT(std::fabs(calculate(DistanceMetric::Cosine, {1, 0}, {0, 1}) - 1.f) < 1e-6f);
  EXPECT(std::fabs(calculate(DistanceMetric::Euclidean, {0, 0}, {3, 4}) - 5.f) < 1e-6f);
  EXPECT(calculate(DistanceMetric::Manhattan, {0, 0}, {3, 4}) == 7.f);
  EXPECT(calculate(DistanceMetric::Cosine, {0, 0, 0}, {1, 2, 3}) == 1.f);
  EXPECT(std::fabs(calculate(DistanceMetric::DotProduct, {1, 2, 3}, {4, 5, 6}) + 32.f) < 1e-6f);
  EXPECT(std::fabs(calculate_squared(DistanceMetric::Euclidean, {0, 0}, {3, 4}) - 25.f) < 1e-6f);
  auto b = batch_calculate(DistanceMetric::Cosine, {1, 0}, {1, 0, 0, 1, -1, 0}, 2);
  EXPECT(b.size() == 3 && std::fabs(b[0]) < 1e-6f && std::fabs(b[1] - 1) < 1e-6f && std::fabs(b[2] - 2) < 1e-6f);

  // a ring graph with chords: self-query returns id 0 at distance ~0, ascending order, k results
  const size_t n = 100, d = 16;
  auto vecs = random_vectors(n, d, 42);
  CsrGraph g;
  for (size_t i = 0; i < n; i++)
    g.add_node({(i + 1) % n, (i + n - 1) % n, (i + 7) % n, (i * 13 + 5) % n == i ? (i + 3) % n : (i * 13 + 5) % n}, 0);
  g.entry_point = 0;
  LeannIndex idx = LeannIndex::from_csr(g, LeannConfig(), d);
  idx.upload(0);
  idx.attach(InMemoryEmbeddingProvider(vecs, d));
  std::vector<float> q(vecs.begin(), vecs.begin() + d);
  auto res = idx.search_with_params(q, 5, 200);  // leann.rs:1290-1304
  EXPECT(res.size() == 5 && res[0].first == 0 && res[0].second < 0.01f);
  for (size_t i = 1; i < res.size(); i++) EXPECT(res[i - 1].second <= res[i].second);  // :1327-1343
  EXPECT(throws(ISL_ERR_DIMENSION_MISMATCH, [&] { idx.search(std::vector<float>(8, 0.5f), 5); }));  // :1315-1325
  // pipelined host-buffer calls through the raw ABI: after isl_index_prepare no call allocates, every
  // token carries its own statistics, and the answers equal the synchronous call's
  {
    idx.prepare(4, 200, 5, 3);
    std::vector<float> qs;
    for (size_t b = 0; b < 4; ++b) qs.insert(qs.end(), vecs.begin() + b * 7 * d, vecs.begin() + (b * 7 + 1) * d);  // rows 0, 7, 14, 21
    uint64_t tok[3];
    std::vector<uint64_t> ids[3];
    std::vector<float> dist[3];
    std::vector<uint32_t> cnt[3];
    for (int c = 0; c < 3; ++c) {
      const uint64_t nqc = (uint64_t)c + 2;  // 2, 3, 4 queries
      ids[c].assign(nqc * 5, 0); dist[c].assign(nqc * 5, 0.f); cnt[c].assign(nqc, 0);
      check(isl_search_batch_async(idx.handle(), qs.data(), nqc, d, 5, 200, ids[c].data(), dist[c].data(),
                                   cnt[c].data(), &tok[c]));
    }
    for (int c = 2; c >= 0; --c) {
      isl_search_stats st{};
      check(isl_search_wait_stats(idx.handle(), tok[c], &st));
      EXPECT(st.queries == (uint64_t)c + 2 && st.allocations == 0);
      EXPECT(cnt[c][0] == 5 && ids[c][0] == 0 && dist[c][0] < 0.01f);
      EXPECT(ids[c][5] == 7);  // the second query is row 7
    }
    EXPECT(isl_search_wait(idx.handle(), tok[0]) == ISL_ERR_INVALID_ARGUMENT);  // a token completes once
  }
  // MultiIndexSearcher over id-range shards through the C++ mirror: the one rank this process is, with
  // the RCCL transport (communicator from a unique id, all-gather of the packed record, merge) --
  // the merged answer of a single shard is that shard's answer with global ids = local ids
  {
    ShardGroup grp(0, 1, 0, ShardGroup::unique_id());
    EXPECT(grp.comm_ranks() == 1);
    ShardedSearcher ss(idx, &grp, n, 2);
    ss.prepare(4, 5, 200);
    std::vector<float> qs(vecs.begin(), vecs.begin() + 3 * d);
    auto merged = ss.search_batch(qs, 3, 5, 200);
    for (size_t i = 0; i < 3; ++i) {
      auto want = idx.search_with_params(std::vector<float>(vecs.begin() + i * d, vecs.begin() + (i + 1) * d), 5, 200);
      EXPECT(merged[i].size() == want.size());
      for (size_t j = 0; j < want.size() && j < merged[i].size(); ++j)
        EXPECT(merged[i][j].id == want[j].first && merged[i][j].score == want[j].second && merged[i][j].shard == 0);
    }
  }
  LeannIndex empty = LeannIndex::with_defaults();  // :1306-1313
  EXPECT(empty.search(std::vector<float>(8, 0.5f), 5).empty());

  // HnswGraph facade + Searcher (hnsw.rs:458-504, search.rs:324-400) on a two-layer graph: layer 1
  // links every 10th node into a ring, layer 0 is the chord ring from above
  std::vector<std::vector<std::vector<uint64_t>>> layers(2, std::vector<std::vector<uint64_t>>(n));
  std::vector<uint64_t> levels(n, 0);
  for (size_t i = 0; i < n; i++) {
    layers[0][i] = {(i + 1) % n, (i + n - 1) % n, (i + 7) % n};
    if (i % 10 == 0) { levels[i] = 1; layers[1][i] = {(i + 10) % n, (i + n - 10) % n}; }
  }
  HnswGraph hg(vecs, d, layers, levels, 0, 1);
  EXPECT(hg.len() == n && !hg.is_empty());
  auto hres = hg.search(q, 5, 200);
  EXPECT(hres.size() == 5 && hres[0].first == 0 && hres[0].second < 0.01f);
  Searcher s(hg);
  EXPECT(s.top_k(5).search(q).size() == 5);
  for (auto& r : Searcher(hg).include_vectors().top_k(3).search(q)) EXPECT(r.vector && r.vector->size() == d);
  for (auto& r : Searcher(hg).min_similarity(0.5f).search(q)) EXPECT(r.to_similarity() >= 0.5f);
  EXPECT(SearchConfig::fast(5).ef == 10 && SearchConfig::accurate(5).ef == 50);  // search.rs:285-297
  SearchResult sr;
  sr.score = 1.0f;
  EXPECT(sr.to_similarity() == 0.5f);  // search.rs:311-324

  // CandleEmbedder (candle_provider.rs:353-507) with all-zero weights except LayerNorm scale 1:
  // every hidden state is LayerNorm(0) = bias = 0 -> pooled 0 -> the 1e-12 norm floor keeps it 0
  isl_bert_config bc{50, 32, 1, 2, 64, 16, 1, 1e-12f, 0};
  CandleEmbedder enc(bc);
  std::vector<int64_t> ids = {1, 2, 3, 0, 4, 5, 0, 0};
  std::vector<float> mask = {1, 1, 1, 0, 1, 1, 0, 0};
  auto emb = enc.embed(ids, mask, 2, 4);
  EXPECT(emb.size() == 64);
  bool all_zero = true;
  for (float v : emb) all_zero = all_zero && v == 0.0f;
  EXPECT(all_zero);
  EXPECT(throws(ISL_ERR_DIMENSION_MISMATCH, [&] { enc.set_weight("embeddings.LayerNorm.bias", std::vector<float>(5)); }));
  EXPECT(throws(ISL_ERR_EMBEDDING, [&] { enc.embed({99, 1, 1, 1}, {1, 1, 1, 1}, 1, 4); }));
}

int main(int argc, char** argv) {
  bool gpu = argc > 1 && !std::strcmp(argv[1], "gpu");
  test_config();
  test_csr_and_bytes();
  if (gpu) test_gpu();
  std::printf("%s: %d failure(s)\n", gpu ? "cpu+gpu" : "cpu", failures);
  return failures ? 1 : 0;
}
