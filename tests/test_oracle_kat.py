"""Pins the CPU oracle against every known-answer test the reference holds for the
hot path.  Each test names the reference test it restates."""
import math

import numpy as np
import pytest

from _data import clustered_vectors, random_csr, random_levels, uniform_vectors
import _pyref


# ------------------------------------------------------------ distance.rs KATs
def test_cosine_identical_orthogonal_opposite(orc):  # distance.rs:150-179
    st, d = orc.distance(orc.COSINE, [1, 2, 3], [1, 2, 3])
    assert st == 0 and abs(d - 0.0) < 1e-6
    st, d = orc.distance(orc.COSINE, [1, 0], [0, 1])
    assert abs(d - 1.0) < 1e-6
    st, d = orc.distance(orc.COSINE, [1, 0], [-1, 0])
    assert abs(d - 2.0) < 1e-6


def test_euclidean_kats(orc):  # distance.rs:181-204
    assert abs(orc.distance(orc.EUCLIDEAN, [1, 2, 3], [1, 2, 3])[1]) < 1e-6
    assert abs(orc.distance(orc.EUCLIDEAN, [0, 0], [3, 4])[1] - 5.0) < 1e-6
    assert orc.distance(orc.EUCLIDEAN, [1, 0], [0, 1])[1] == np.float32(1.414213562373095)
    assert orc.distance(orc.MANHATTAN, [1, 1], [1, 1])[1] == 0.0
    assert orc.distance(orc.MANHATTAN, [0, 0], [3, 4])[1] == 7.0


def test_dimension_mismatch_all_metrics(orc):  # distance.rs:206-212, 333-352
    for m in range(4):
        assert orc.distance(m, [1, 2], [1, 2, 3])[0] == orc.DIMENSION_MISMATCH
        assert orc.distance(m, [1, 2, 3, 4], [1, 2, 3, 4])[0] == 0
    assert orc.distance_squared(orc.EUCLIDEAN, [1, 2], [1, 2, 3])[0] == orc.DIMENSION_MISMATCH


def test_zero_vector_cosine_exactly_one(orc):  # distance.rs:214-220
    assert orc.distance(orc.COSINE, [0, 0, 0], [1, 2, 3])[1] == 1.0


def test_dot_product(orc):  # distance.rs:222-229
    assert abs(orc.distance(orc.DOT, [1, 2, 3], [4, 5, 6])[1] + 32.0) < 1e-6


def test_normalize(orc):  # distance.rs:231-248, 365-372
    v = orc.normalize([3, 4])
    assert abs(math.sqrt(float((v * v).sum())) - 1.0) < 1e-6
    assert (orc.normalize([0, 0, 0]) == 0).all()


def test_batch_calculate(orc):  # distance.rs:250-261, 374-382
    d = orc.batch_distance(orc.COSINE, [1, 0], [[1, 0], [0, 1], [-1, 0]])
    assert d.shape == (3,)
    assert abs(d[0]) < 1e-6 and abs(d[1] - 1) < 1e-6 and abs(d[2] - 2) < 1e-6
    assert orc.batch_distance(orc.COSINE, [1, 0], np.zeros((0, 2))).size == 0


def test_squared(orc):  # distance.rs:354-372
    assert abs(orc.distance_squared(orc.EUCLIDEAN, [0, 0], [3, 4])[1] - 25.0) < 1e-6
    d = orc.distance(orc.COSINE, [1, 0], [0, 1])[1]
    assert abs(orc.distance_squared(orc.COSINE, [1, 0], [0, 1])[1] - d * d) < 1e-6


def test_distance_properties(orc):  # distance.rs:264-328 (proptest invariants)
    rng = np.random.default_rng(0)
    for _ in range(200):
        a = (rng.random(8, dtype=np.float32) * 20 - 10)
        b = (rng.random(8, dtype=np.float32) * 20 - 10)
        c = (rng.random(8, dtype=np.float32) * 20 - 10)
        ab = orc.distance(orc.EUCLIDEAN, a, b)[1]
        assert ab >= 0 and orc.distance(orc.MANHATTAN, a, b)[1] >= 0
        assert abs(ab - orc.distance(orc.EUCLIDEAN, b, a)[1]) < 1e-5
        assert abs(orc.distance(orc.EUCLIDEAN, a, a)[1]) < 1e-6
        assert orc.distance(orc.EUCLIDEAN, a, c)[1] <= ab + orc.distance(orc.EUCLIDEAN, b, c)[1] + 1e-4
        pa, pb = np.abs(a) + 0.1, np.abs(b) + 0.1
        assert 0.0 <= orc.distance(orc.COSINE, pa, pb)[1] <= 2.0


def test_cosine_is_strict_sequential_f32(orc):
    """Operation order of distance.rs:71-88: three left-to-right f32 chains, mul then add."""
    rng = np.random.default_rng(1)
    for d in (3, 17, 128, 768):
        a = rng.standard_normal(d).astype(np.float32)
        b = rng.standard_normal(d).astype(np.float32)
        assert orc.distance(orc.COSINE, a, b)[1] == _pyref.cosine(a, b)


# ----------------------------------------------------------------- CSR (leann.rs)
def test_csr_get_neighbors(orc):  # leann.rs:1193-1204
    g = orc.Csr(node_offsets=[0, 0, 1, 3], neighbors=[0, 0, 1], entry_point=0)
    assert g.get_neighbors(0) == []
    assert g.get_neighbors(1) == [0]
    assert g.get_neighbors(2) == [0, 1]
    assert g.get_neighbors(999) is None


# ----------------------------------------------------------------- search (leann.rs)
def _build(orc, n, d, seed, **kw):
    v = uniform_vectors(n, d, seed)
    g = orc.leann_build(v, levels=random_levels(n, kw.get("m", 30), seed + 1), **kw)
    return v, g


def test_index_search_self_query(orc):  # leann.rs:1290-1304
    v, g = _build(orc, 100, 16, 42)
    r = orc.leann_search(g, v, v[0], 5, 64)
    assert r.status == 0 and r.ids.size == 5
    assert r.ids[0] == 0 and r.dist[0] < 0.01


def test_search_empty_index(orc):  # leann.rs:1306-1313
    g = orc.Csr(node_offsets=[0], neighbors=[], entry_point=None)
    r = orc.leann_search(g, np.zeros((1, 8), np.float32), [0.5] * 8, 5, 64, dimension=None)
    assert r.status == 0 and r.ids.size == 0


def test_search_dimension_mismatch(orc):  # leann.rs:1315-1325
    v, g = _build(orc, 10, 16, 42)
    r = orc.leann_search(g, v, [0.5] * 8, 5, 64)
    assert r.status == orc.DIMENSION_MISMATCH


def test_search_results_sorted(orc):  # leann.rs:1327-1343
    v, g = _build(orc, 50, 16, 123)
    r = orc.leann_search(g, v, [0.5] * 16, 10, 64)
    assert (np.diff(r.dist) >= 0).all()


@pytest.mark.parametrize("n,d", [(10, 8), (50, 16), (100, 32)])
def test_various_sizes(orc, n, d):  # leann.rs:1515-1531
    v, g = _build(orc, n, d, 42)
    assert g.num_nodes == n
    r = orc.leann_search(g, v, v[0], min(5, n), 64)
    assert r.ids.size == min(5, n)
    assert (r.ids < n).all()  # prop_search_returns_valid_ids leann.rs:1480-1493


@pytest.mark.parametrize("ratio", [0.0, 0.3, 0.5, 0.8])
@pytest.mark.parametrize("strategy", [0, 1])
def test_prune_ratios(orc, ratio, strategy):  # leann.rs:1437-1464, 1553-1572
    v, g = _build(orc, 50, 16, 42)
    r = orc.leann_search(g, v, v[0], 5, 64, prune_ratio=ratio, strategy=strategy)
    assert r.status == 0 and r.ids.size == 5


def test_recall_quality(orc):  # leann.rs:1388-1433 (accurate(): m=48,m0=96,ef_c=400,ef=128)
    n, d = 200, 32
    v = uniform_vectors(n, d, 42)
    g = orc.leann_build(v, m=48, m0=96, ef_construction=400, levels=random_levels(n, 48, 7))
    correct = 0
    for i in range(20):
        q = v[i * 10 % n]
        truth = int(np.argmin(orc.batch_distance(orc.COSINE, q, v)))
        r = orc.leann_search(g, v, q, 1, 128)
        correct += int(r.ids.size > 0 and r.ids[0] == truth)
    assert correct / 20 >= 0.35


def test_build_invariants(orc):
    """LeannIndex::build (leann.rs:560-631): degrees <= m0 after pruning, edges valid,
    degree_counts == row lengths, entry = first node of max level."""
    n, d, m0 = 300, 16, 20
    v = uniform_vectors(n, d, 5)
    lv = random_levels(n, 10, 6)
    g = orc.leann_build(v, m=10, m0=m0, ef_construction=40, levels=lv)
    deg = np.diff(g.node_offsets)
    assert (deg == g.degree_counts).all()
    assert deg.max() <= m0 and (g.neighbors < n).all()
    assert g.max_level == int(lv.max())
    assert g.entry_point == int(np.argmax(lv))  # first strictly-greater level wins (:610)
    for i in range(n):
        nb = g.get_neighbors(i)
        assert len(set(nb)) == len(nb) and i not in nb


def test_node_not_found(orc):  # provider miss, leann.rs:145-150 / :947
    g = orc.Csr(node_offsets=[0, 1, 1], neighbors=[7], entry_point=0)
    r = orc.leann_search(g, uniform_vectors(2, 4, 0), [0.1] * 4, 1, 4)
    assert r.status == orc.NODE_NOT_FOUND and r.payload == 7


def test_fewer_than_k_when_unreachable(orc):
    g = orc.Csr(node_offsets=[0, 1, 2, 2], neighbors=[1, 0], entry_point=0)
    r = orc.leann_search(g, uniform_vectors(3, 4, 0), [0.1] * 4, 3, 8)
    assert r.status == 0 and sorted(r.ids.tolist()) == [0, 1]


def test_c_oracle_matches_python_restatement(orc):
    """Two independent restatements (C and pure Python) of leann.rs:899-988 must agree
    bit-for-bit, including heap-array tie order with duplicated vectors."""
    for seed, dup_vectors in ((0, False), (1, True), (2, True)):
        n, d = 120, 12
        v = uniform_vectors(n, d, seed)
        if dup_vectors:
            v[n // 2:] = v[: n - n // 2]  # exact distance ties
        off, nb = random_csr(n, 10, seed + 10, dup=(seed == 2))
        g = orc.Csr(off, nb, entry_point=3)
        for qi in range(6):
            q = uniform_vectors(1, d, 100 + qi)[0]
            for ef in (1, 4, 16, 200):
                ids, dist = _pyref.leann_search_layer(off, nb, v, q, 3, ef)
                r = orc.leann_search(g, v, q, ef, ef)
                assert r.ids.tolist() == ids, (seed, qi, ef)
                assert r.dist.tolist() == [float(x) for x in dist]


# ------------------------------------------------------------------ hnsw.rs
def _hnsw(orc, n, d, seed, **kw):
    v = uniform_vectors(n, d, seed)
    h = orc.Hnsw(**kw)
    lv = random_levels(n, kw.get("m", 16), seed + 3)
    for i in range(n):
        st, idx = h.insert(v[i], int(lv[i]))
        assert st == 0 and idx == i
    return v, h, lv


def test_hnsw_search_basic(orc):  # hnsw.rs:615-687 shapes
    v, h, lv = _hnsw(orc, 100, 16, 42)
    assert len(h) == 100 and h.max_level == int(lv.max())
    r = h.search(v[0], 5, 50)
    assert r.status == 0 and r.ids.size == 5 and r.ids[0] == 0 and r.dist[0] < 0.01
    assert (np.diff(r.dist) >= 0).all()
    assert h.search([0.5] * 8, 5, 50).status == orc.DIMENSION_MISMATCH
    assert orc.Hnsw().search([0.5] * 8, 5, 50).ids.size == 0


def test_hnsw_recall(orc):  # hnsw.rs:806-854 (accurate(): m=32,m0=64,ef_c=400; ef=100)
    n, d = 200, 32
    v, h, _ = _hnsw(orc, n, d, 42, m=32, m0=64, ef_construction=400)
    correct = 0
    for i in range(20):
        q = v[i * 10 % n]
        truth = int(np.argmin(orc.batch_distance(orc.COSINE, q, v)))
        r = h.search(q, 1, 100)
        correct += int(r.ids[0] == truth)
    assert correct / 20 >= 0.35


def test_hnsw_degree_bounds(orc):
    v, h, lv = _hnsw(orc, 150, 8, 9, m=6, m0=12, ef_construction=30)
    for i in range(150):
        for layer in range(h.level(i) + 1):
            nb = h.neighbors(i, layer)
            assert len(nb) <= (12 if layer == 0 else 6)
        assert h.neighbors(i, h.level(i) + 1) is None


# ---------------------------------------------------------------- search.rs
def test_to_similarity(orc):  # search.rs:311-324
    assert orc.to_similarity(0.0) == 1.0
    assert orc.to_similarity(1.0) == 0.5
    assert abs(orc.to_similarity(9.0) - 0.1) < 1e-7


def test_multi_index_merge_stable(orc):  # search.rs:211-237
    st, ids, sc, src = orc.multi_index_merge(
        [[10, 11, 12], [20, 21]], [[0.1, 0.5, 0.9], [0.1, 0.5]], 4)
    assert st == 0
    assert ids.tolist() == [10, 20, 11, 21] and src.tolist() == [0, 1, 0, 1]
    st, *_ = orc.multi_index_merge([[1], [2]], [[float("nan")], [0.5]], 2)
    assert st == orc.PANIC  # partial_cmp().unwrap() on NaN


def test_service_merge(orc):  # indexer/service.rs:787-801
    st, ids, sc, src = orc.service_merge([[1, 2], [3]], [[0.2, 0.4], [0.2]], 2)
    assert ids.tolist() == [1, 3] and np.allclose(sc, [0.8, 0.8])


# --------------------------------------------------------------------- pq.rs
def test_pq_find_nearest(orc):  # pq.rs:787-809
    cents = np.array([[0, 0], [1, 1], [2, 2]], np.float32)
    assert orc.pq_find_nearest(orc.EUCLIDEAN, cents, [0.1, 0.1]) == (0, 0)
    assert orc.pq_find_nearest(orc.EUCLIDEAN, cents, [0.9, 0.9]) == (0, 1)
    assert orc.pq_find_nearest(orc.EUCLIDEAN, cents, [1, 2, 3])[0] == orc.DIMENSION_MISMATCH


def test_pq_table_vs_asymmetric(orc):  # pq.rs:639-669 (within 1e-3)
    rng = np.random.default_rng(3)
    m, K, dsub = 8, 256, 16
    cb = rng.standard_normal((m, K, dsub)).astype(np.float32)
    q = rng.standard_normal(m * dsub).astype(np.float32)
    v = rng.standard_normal(m * dsub).astype(np.float32)
    st, codes = orc.pq_encode(orc.EUCLIDEAN, cb, v)
    assert st == 0 and codes.size == m
    st, t = orc.pq_build_tables(cb, q)
    st, ad = orc.pq_asymmetric_distance(cb, q, codes)
    td = orc.pq_table_distance(t, codes)
    assert abs(ad - td) < 1e-3
    st, dec = orc.pq_decode(cb, codes)
    assert st == 0 and (dec.reshape(m, dsub) == cb[np.arange(m), codes]).all()
    assert orc.pq_decode(cb, codes[:3])[0] == orc.PQ_ERROR


def test_pq_encode_first_min_wins(orc):  # pq.rs:97-103 strict `<`
    cents = np.array([[1, 1], [1, 1], [0, 0]], np.float32)
    assert orc.pq_find_nearest(orc.EUCLIDEAN, cents, [1, 1]) == (0, 0)


# ---------------------------------------------------------------- pooling
def test_mean_pool_normalize(orc):  # candle_provider.rs:434-488
    rng = np.random.default_rng(4)
    h = rng.standard_normal((2, 5, 8)).astype(np.float32)
    mask = np.array([[1, 1, 1, 0, 0], [1, 1, 1, 1, 1]], np.float32)
    out = orc.mean_pool_normalize(h, mask, True)
    ref = (h * mask[:, :, None]).sum(1) / mask.sum(1, keepdims=True)
    ref /= np.linalg.norm(ref, axis=1, keepdims=True)
    assert np.allclose(out, ref, atol=1e-6)
    assert np.allclose(np.linalg.norm(out, axis=1), 1.0, atol=1e-6)


def test_clustered_helper_shapes():
    x = clustered_vectors(200, 16, 0)
    assert x.shape == (200, 16) and np.allclose(np.linalg.norm(x, axis=1), 1, atol=1e-5)
