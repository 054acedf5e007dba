"""GPU parity tests for the HnswGraph search facade (hnsw.rs:458-504), the Searcher /
MultiIndexSearcher layer above it (search.rs:105-256) and the mean-pool + normalise tail of
the recompute encoder (candle_provider.rs:434-488).  Graphs are built by the CPU oracle's
restatement of HnswGraph::insert with seeded levels, handed over layer by layer, searched on
the device, and compared id-for-id and bit-for-bit with the oracle's search."""
import numpy as np
import pytest

import islands_amd as ia
from _data import random_levels, uniform_vectors

pytestmark = pytest.mark.gpu

METRICS = [ia.DistanceMetric.Cosine, ia.DistanceMetric.Euclidean, ia.DistanceMetric.DotProduct,
           ia.DistanceMetric.Manhattan]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def build(orc, n, d, seed, m=16, m0=32, ef_construction=200, metric=ia.DistanceMetric.Cosine,
          vectors=None):
    v = uniform_vectors(n, d, seed) if vectors is None else vectors
    h = orc.Hnsw(m=m, m0=m0, ef_construction=ef_construction, metric=int(metric))
    lv = random_levels(n, m, seed + 3)
    for i in range(n):
        st, idx = h.insert(v[i], int(lv[i]))
        assert st == 0 and idx == i
    layers = [[(h.neighbors(i, L) or []) for i in range(n)] for L in range(h.max_level + 1)]
    levels = [h.level(i) for i in range(n)]
    g = ia.HnswGraph(v, layers, levels, h.entry_point, h.max_level, m=m, m0=m0,
                     ef_construction=ef_construction, metric=metric)
    return v, h, g


def assert_same(h, g, queries, k, ef):
    got = g.search_batch(queries, k, ef)
    for i, q in enumerate(queries):
        r = h.search(q, k, ef)
        assert r.status == 0
        ids, dd = got[i]
        assert ids.tolist() == r.ids.tolist(), (i, ids, r.ids)
        assert bits(dd).tolist() == bits(r.dist).tolist(), (i, dd, r.dist)


@pytest.mark.parametrize("metric", METRICS)
def test_hnsw_search_matches_oracle(orc, metric):
    v, h, g = build(orc, 600, 24, 11, metric=metric)
    assert len(g) == 600 and h.max_level >= 1
    q = uniform_vectors(40, 24, 99)
    assert_same(h, g, q, 10, 50)
    assert_same(h, g, v[:16], 5, 5)
    assert_same(h, g, q[:8], 20, 10)   # ef < k -> ef = k, hnsw.rs:500
    g.search_batch(q, 10, 50)
    st = g.last_stats()
    assert st["queries"] == 40 and st["exact_path"] == 0  # one wave per query, no tie met


def test_hnsw_d768_and_small_ef(orc):
    v, h, g = build(orc, 400, 768, 5, m=8, m0=16, ef_construction=40)
    q = uniform_vectors(24, 768, 6)
    for ef in (1, 3, 64):
        assert_same(h, g, q, 1 if ef < 10 else 10, ef)


def test_hnsw_ties_and_duplicates(orc):
    # duplicated vectors -> equal distances everywhere; order must still follow the heaps
    base = uniform_vectors(60, 8, 3)
    v = np.concatenate([base, base, base[:30]]).astype(np.float32)
    _, h, g = build(orc, v.shape[0], 8, 21, m=6, m0=12, ef_construction=30, vectors=v,
                    metric=ia.DistanceMetric.Euclidean)
    assert_same(h, g, base[:20], 10, 40)
    assert g.last_stats()["exact_path"] > 0  # equal distances: the heap-exact kernel decides


def test_hnsw_reference_shapes(orc):  # hnsw.rs:615-687
    v, h, g = build(orc, 100, 16, 42)
    r = g.search(v[0], 5, 50)
    assert len(r) == 5 and r[0][0] == 0 and r[0][1] < 0.01
    assert all(r[i][1] <= r[i + 1][1] for i in range(4))
    with pytest.raises(ia.CoreError) as e:
        g.search([0.5] * 8, 5, 50)
    assert e.value.kind == "DimensionMismatch" and (e.value.expected, e.value.actual) == (16, 8)
    empty = ia.HnswGraph(np.zeros((0, 0), np.float32), [], [], None, 0)
    assert empty.is_empty() and empty.search([0.5] * 8, 5, 50) == []
    for bad in (dict(m=0), dict(m=16, m0=8), dict(m=16, m0=32, ef_construction=8)):
        with pytest.raises(ia.CoreError) as e:
            ia.HnswGraph(np.zeros((0, 0), np.float32), [], [], None, 0, **bad)
        assert e.value.kind == "InvalidConfig"


def test_single_node_graph(orc):
    v, h, g = build(orc, 1, 4, 7)
    assert_same(h, g, uniform_vectors(3, 4, 8), 5, 10)


# ----------------------------------------------------------------- search.rs
def test_search_config():  # search.rs:276-297
    c = ia.SearchConfig()
    assert (c.top_k, c.ef, c.include_vectors, c.include_metadata) == (10, 100, False, True)
    assert (ia.SearchConfig.fast(5).top_k, ia.SearchConfig.fast(5).ef) == (5, 10)
    assert (ia.SearchConfig.accurate(5).top_k, ia.SearchConfig.accurate(5).ef) == (5, 50)
    assert ia.SearchResult(0, 0.0).to_similarity() == 1.0
    assert ia.SearchResult(0, 1.0).to_similarity() == 0.5
    assert ia.SearchResult(0, 9.0).to_similarity() == float(np.float32(0.1))


def test_searcher_facade(orc):  # search.rs:324-400
    v, h, g = build(orc, 50, 16, 42, m=8, m0=16, ef_construction=100)  # HnswConfig::fast()
    q = np.full(16, 0.5, np.float32)
    res = ia.Searcher(g).search(q)
    assert 0 < len(res) <= 10
    assert len(ia.Searcher(g).top_k(5).search(q)) == 5
    for r in ia.Searcher(g).include_vectors().top_k(3).search(q):
        assert r.vector is not None and r.vector.shape == (16,) and (r.vector == v[r.id]).all()
    res = ia.Searcher(g).min_similarity(0.5).search(q)
    want = [int(i) for i, s in zip(*(lambda r: (r.ids, r.dist))(h.search(q, 10, 100)))
            if orc.to_similarity(float(s)) >= np.float32(0.5)]
    assert [r.id for r in res] == want
    assert all(r.to_similarity() >= 0.5 for r in res)
    res = ia.Searcher(g).top_k(10).search(q)
    assert all(a.score <= b.score for a, b in zip(res, res[1:]))
    qs = np.stack([np.full(16, i * 0.1, np.float32) for i in range(3)])
    batch = ia.Searcher(g).top_k(5).search_batch(qs)
    assert len(batch) == 3
    for i, rs in enumerate(batch):
        if i == 0:
            continue  # the zero query: cosine distance is 1.0 to every node, still 5 results
        assert len(rs) == 5
        assert [r.id for r in rs] == h.search(qs[i], 5, 100).ids.tolist()
    assert len(batch[0]) == 5


def test_multi_index_searcher(orc):  # search.rs:402-430, 211-237
    v1, h1, g1 = build(orc, 20, 8, 42, m=8, m0=16, ef_construction=100)
    v2, h2, g2 = build(orc, 20, 8, 43, m=8, m0=16, ef_construction=100)
    s = ia.MultiIndexSearcher()
    s.add_index("index1", g1)
    s.add_index("index2", g2)
    assert s.num_indexes() == 2 and s.total_vectors() == 40
    q = np.full(8, 0.5, np.float32)
    res = s.search(q)
    r1, r2 = h1.search(q, 10, 100), h2.search(q, 10, 100)
    st, ei, es, esrc = orc.multi_index_merge([r1.ids, r2.ids], [r1.dist, r2.dist], 10)
    assert [(n, r.id) for n, r in res] == [(f"index{int(a) + 1}", int(b)) for a, b in zip(esrc, ei)]
    assert bits([r.score for _, r in res]).tolist() == bits(es).tolist()
    assert ia.MultiIndexSearcher().search(q) == []


def test_merge_service_matches_oracle(orc):  # indexer/service.rs:787-801
    rng = np.random.default_rng(8)
    nl, nq, k = 3, 11, 7
    dist = np.sort(rng.integers(0, 6, (nl, nq, k)).astype(np.float32) / 8, axis=2)
    dist[0, 0, :3] = [1e-9, 2e-9, 3e-9]  # different distances, equal scores after 1 - d
    ids = rng.integers(0, 50, (nl, nq, k)).astype(np.uint64)
    counts = rng.integers(0, k + 1, (nl, nq)).astype(np.uint32)
    counts[0, 0] = k
    files_len = np.array([50, 30, 40], np.uint64)
    for fl in (None, files_len):
        oi, osc, osrc, oc = ia.merge_service(ids, dist, counts, 5, files_len=fl)
        for q in range(nq):
            li, ld = [], []
            for l in range(nl):
                a, b = ids[l, q, :counts[l, q]], dist[l, q, :counts[l, q]]
                if fl is not None:
                    keep = a < fl[l]
                    a, b = a[keep], b[keep]
                li.append(a)
                ld.append(b)
            st, ei, es, esrc = orc.service_merge(li, ld, 5)
            n = int(oc[q])
            assert st == 0 and n == ei.size
            assert oi[q, :n].tolist() == ei.tolist() and osrc[q, :n].tolist() == esrc.tolist()
            assert bits(osc[q, :n]).tolist() == bits(es).tolist()


# --------------------------------------------------- embedding/candle_provider.rs
@pytest.mark.parametrize("B,L,H", [(1, 1, 8), (3, 17, 384), (5, 128, 768), (2, 512, 64)])
def test_mean_pool_normalize_matches_oracle(orc, B, L, H):
    rng = np.random.default_rng(B * 1000 + L)
    hidden = rng.standard_normal((B, L, H)).astype(np.float32)
    lens = rng.integers(1, L + 1, B)
    mask = (np.arange(L)[None, :] < lens[:, None]).astype(np.float32)
    if B > 1:
        mask[-1] = 0.0  # fully padded row -> clamp(1e-9) and the 1e-12 norm floor
    for normalize in (True, False):
        got = ia.mean_pool_normalize(hidden, mask, normalize)
        want = orc.mean_pool_normalize(hidden, mask, normalize)
        assert bits(got).tolist() == bits(want).tolist()
    if B > 1:
        n = np.linalg.norm(ia.mean_pool_normalize(hidden, mask, True)[:-1], axis=1)
        assert np.all(np.abs(n - 1) < 1e-5)
