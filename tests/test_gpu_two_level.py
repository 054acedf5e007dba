"""GPU parity of the two-level search (EXTENSION: docs/leann-specification.md:223-275, Algorithm 2,
not implemented by the reference).  The HIP path, through the C ABI, must return exactly what
oracle/islands_oracle.c::orc_two_level_search returns: ids, distance bits, counts, counters."""
import numpy as np
import pytest

import islands_amd as ia
from _data import clustered_vectors, knn_graph, random_csr, uniform_vectors
from test_gpu_parity import bits, make_index
from test_two_level_cpu import make_pq

pytestmark = pytest.mark.gpu


def attach_pq(idx, cb, codes, metric=ia.DistanceMetric.Euclidean):
    m, K, dsub = cb.shape
    pq = ia.ProductQuantizer(m * dsub, cb, metric)
    idx.set_pq_codes(pq, codes)
    return pq


def assert_same(orc, idx, csr, v, cb, codes, queries, k, ef, ratio, metric=0, skip=()):
    ids, dist, cnt = idx.search_two_level_batch(queries, k, ef, ratio)
    st = idx.last_stats()
    tot = {"expansions": 0, "edges": 0, "evals": 0, "pushes": 0}
    for i, q in enumerate(queries):
        r = orc.two_level_search(csr, v, cb, codes, q, k, ef, ratio, metric=metric)
        assert r.status == 0
        n = int(cnt[i])
        assert n == r.ids.size, (i, n, r.ids.size)
        assert ids[i, :n].tolist() == r.ids.tolist(), (i, ids[i, :n], r.ids)
        assert bits(dist[i, :n]).tolist() == bits(r.dist).tolist(), (i, dist[i, :n], r.dist)
        for f in tot:
            tot[f] += r.counters[f]
    for f in tot:
        if f not in skip:
            assert st[f] == tot[f], (f, st[f], tot[f])
    return st


@pytest.mark.parametrize("metric", [0, 1, 2, 3])
@pytest.mark.parametrize("ratio", [0.1, 0.25, 1.0])
def test_two_level_matches_oracle(orc, metric, ratio):
    n, d, m, K = 4000, 64, 16, 64
    v = clustered_vectors(n, d, 31)
    off, nb = knn_graph(v, 20, seed=3)
    cb, codes = make_pq(v, m, K, 6)
    csr = orc.Csr(off, nb, entry_point=5)
    idx = make_index(csr, v, ia.LeannConfig(metric=ia.DistanceMetric(metric)))
    pq = attach_pq(idx, cb, codes)
    q = clustered_vectors(24, d, 32)
    for k, ef in ((10, 64), (5, 5), (20, 130)):
        assert_same(orc, idx, csr, v, cb, codes, q, k, ef, ratio, metric=metric)
    del pq


def test_two_level_d768_m96(orc):
    """The shape of the headline workload: d = 768, PQ with 96 subquantizers of 8 dims."""
    n, d, m, K = 3000, 768, 96, 256
    v = clustered_vectors(n, d, 41)
    off, nb = knn_graph(v, 32, seed=4)
    cb, codes = make_pq(v, m, K, 7)
    csr = orc.Csr(off, nb, entry_point=0)
    idx = make_index(csr, v)
    pq = attach_pq(idx, cb, codes)
    q = clustered_vectors(16, d, 42)
    st = assert_same(orc, idx, csr, v, cb, codes, q, 10, 128, 0.15)
    assert st["evals"] < st["pushes"]
    del pq


def test_two_level_ties_long_rows_and_duplicates(orc):
    """Quantised rows (equal distances everywhere, exact and approximate), adjacency rows longer
    than 64 ids, repeated neighbour ids, few centroids (many equal code rows)."""
    n, d, m, K = 1500, 16, 4, 5
    rng = np.random.default_rng(8)
    v = np.round(uniform_vectors(n, d, 9) * 2) / 2
    v[v == 0] = 0.5
    v = v.astype(np.float32)
    off, nb = random_csr(n, 90, 10, dup=True)
    cb, codes = make_pq(v, m, K, 8)
    csr = orc.Csr(off, nb, entry_point=3)
    for metric in (0, 1, 3):
        idx = make_index(csr, v, ia.LeannConfig(metric=ia.DistanceMetric(metric)))
        pq = attach_pq(idx, cb, codes)
        q = v[rng.integers(0, n, 12)] + np.float32(0.25)
        for ratio in (0.05, 0.5, 1.0):
            # the device copy of a row holds each id once, so its edge count is smaller
            assert_same(orc, idx, csr, v, cb, codes, q, 10, 40, ratio, metric=metric, skip=("edges",))
        del pq


def test_two_level_bf16_rows(orc):
    n, d, m, K = 2000, 128, 16, 32
    v = clustered_vectors(n, d, 51)
    vb = (v.view(np.uint32) >> 16).astype(np.uint16)           # truncation to bf16
    vw = (vb.astype(np.uint32) << 16).view(np.float32)          # the provider's exact f32 images
    off, nb = knn_graph(vw, 16, seed=5)
    cb, codes = make_pq(vw, m, K, 9)
    csr = orc.Csr(off, nb, entry_point=0)
    g = ia.CsrGraph(node_offsets=csr.node_offsets, neighbors=csr.neighbors, levels=csr.levels,
                    entry_point=0, num_nodes=n, degree_counts=csr.degree_counts)
    idx = ia.LeannIndex.from_csr(g, None, dimension=d).upload(0)
    idx.set_embeddings_bf16(vb)
    pq = attach_pq(idx, cb, codes)
    q = clustered_vectors(10, d, 52)
    assert_same(orc, idx, csr, vw, cb, codes, q, 10, 64, 0.2)
    del pq


def test_two_level_errors(orc):
    n, d, m, K = 300, 16, 4, 8
    v = clustered_vectors(n, d, 61, per_cluster=10)
    off, nb = random_csr(n, 8, 11)
    cb, codes = make_pq(v, m, K, 10)
    csr = orc.Csr(off, nb, entry_point=0)
    idx = make_index(csr, v)
    q = v[:3]
    with pytest.raises(ia.CoreError) as e:       # nothing attached yet
        idx.search_two_level_batch(q, 3, 8, 0.5)
    assert e.value.kind == "PQError"
    pq = ia.ProductQuantizer(d, cb)
    with pytest.raises(ia.CoreError) as e:       # tables[sq][code] would panic in the reference
        idx.set_pq_codes(pq, np.full((n, m), K, np.uint16))
    assert e.value.kind == "PQError"
    idx.set_pq_codes(pq, codes[:40])             # neighbours beyond the code rows
    with pytest.raises(ia.CoreError) as e:
        idx.search_two_level_batch(q, 3, 8, 0.5)
    assert e.value.kind == "NodeNotFound"
    r = orc.two_level_search(csr, v, cb, codes[:40], q[0], 3, 8, 0.5)
    assert r.status == orc.NODE_NOT_FOUND and e.value.node == r.payload
    idx.set_pq_codes(pq, codes)
    with pytest.raises(ia.CoreError) as e:       # query dimension
        idx.search_two_level_batch(np.zeros((2, d + 1), np.float32), 3, 8, 0.5)
    assert e.value.kind == "DimensionMismatch"
    ids, dist, cnt = idx.search_two_level_batch(q, 0, 8, 0.5)   # k = 0
    assert cnt.tolist() == [0, 0, 0]
    # ratio <= 0, NaN, > 1: the oracle's rules
    for ratio in (0.0, float("nan"), 7.5):
        assert_same(orc, idx, csr, v, cb, codes, q, 5, 8, ratio)


def test_two_level_with_recompute_provider(orc):
    """Exact distances from the recompute provider: only promoted nodes are encoded, and the
    answers are those of the in-memory provider holding the same embeddings."""
    from test_gpu_encoder import _recompute_case
    cfg, enc, tok, lens, emb = _recompute_case(orc)
    n, d = emb.shape
    levels = np.zeros(n, np.uint64)
    csr = orc.leann_build(emb, m=8, m0=16, ef_construction=40, levels=levels)
    cb, codes = make_pq(emb, 8, 64, 12)
    g = ia.CsrGraph(node_offsets=csr.node_offsets, neighbors=csr.neighbors, levels=csr.levels,
                    entry_point=csr.entry_point, max_level=csr.max_level, num_nodes=csr.num_nodes,
                    degree_counts=csr.degree_counts)
    q = emb[::97] + np.float32(0.01)
    mem_idx = ia.LeannIndex.from_csr(g, None, dimension=d).upload(0)
    mem_idx.set_embeddings(emb)
    pq = attach_pq(mem_idx, cb, codes)
    want = mem_idx.search_two_level_batch(q, 10, 48, 0.2)
    want_stats = mem_idx.last_stats()
    plain = mem_idx.search_batch(q, 10, 48)
    plain_stats = mem_idx.last_stats()
    rec_idx = ia.LeannIndex.from_csr(g, None, dimension=d).upload(0)
    rec_idx.set_recompute_provider(enc, tok, lens)
    rec_idx.set_pq_codes(pq, codes)
    got = rec_idx.search_two_level_batch(q, 10, 48, 0.2)
    st = rec_idx.last_stats()
    assert got[2].tolist() == want[2].tolist()
    assert got[0].tolist() == want[0].tolist()
    assert got[1].view(np.uint32).tolist() == want[1].view(np.uint32).tolist()
    for f in ("expansions", "edges", "evals", "pushes"):
        assert st[f] == want_stats[f], f
    assert 0 < st["encoded_nodes"] <= min(want_stats["evals"], n)
    assert want_stats["evals"] < plain_stats["evals"]     # fewer embeddings recomputed
    for i in range(q.shape[0]):
        r = orc.two_level_search(csr, emb, cb, codes, q[i], 10, 48, 0.2)
        c = int(got[2][i])
        assert got[0][i, :c].tolist() == r.ids.tolist()
    del plain


def test_two_level_window_retry(orc):
    """A small ef sizes the LDS window of the approximate queue small; ratio 1 over long adjacency
    rows outgrows it, and the batch is answered by a run with a larger window -- same results."""
    n, d, m, K = 3000, 32, 8, 32
    v = clustered_vectors(n, d, 71)
    off, nb = random_csr(n, 64, 12)
    cb, codes = make_pq(v, m, K, 13)
    csr = orc.Csr(off, nb, entry_point=1)
    idx = make_index(csr, v)
    pq = attach_pq(idx, cb, codes)
    q = clustered_vectors(8, d, 72)
    st = assert_same(orc, idx, csr, v, cb, codes, q, 3, 6, 1.0)
    assert st["pushes"] / 8 > 256          # |AQ| beyond the first window (256 entries)
    del pq


def test_two_level_config5_shape(orc):
    """BASELINE config 5's shape at a small node count: d = 4096, bf16 rows, PQ m = 64 (dsub = 64),
    K = 256 -- the PQ re-rank path over the plain search's rows."""
    n, d, m, K = 700, 4096, 64, 256
    v = clustered_vectors(n, d, 81, per_cluster=35)
    vb = (v.view(np.uint32) >> 16).astype(np.uint16)
    vw = (vb.astype(np.uint32) << 16).view(np.float32)
    off, nb = knn_graph(vw, 14, seed=6)
    cb, codes = make_pq(vw, m, K, 14)
    csr = orc.Csr(off, nb, entry_point=0)
    g = ia.CsrGraph(node_offsets=csr.node_offsets, neighbors=csr.neighbors, levels=csr.levels,
                    entry_point=0, num_nodes=n, degree_counts=csr.degree_counts)
    idx = ia.LeannIndex.from_csr(g, None, dimension=d).upload(0)
    idx.set_embeddings_bf16(vb)
    pq = attach_pq(idx, cb, codes)
    q = clustered_vectors(6, d, 82, per_cluster=35)
    assert_same(orc, idx, csr, vw, cb, codes, q, 10, 48, 0.25)
    del pq


def test_two_level_fuzz_slice(orc):
    """A fixed slice of tests/fuzz_parity.py --mode two_level (random graphs, PQ shapes, ratios,
    ef / k, quantised and duplicated rows); the tool itself runs for minutes on the GPU box."""
    import fuzz_parity
    rng = np.random.default_rng(2027)
    for case in range(25):
        fuzz_parity.two_level_case(rng, case)


def test_two_level_many_queries_and_large_ef(orc):
    """More queries than resident waves (the kernel's work queue hands a slot several queries, the
    per-slot state must be clean each time), ef beyond 512 and k beyond ef."""
    n, d, m, K = 2500, 32, 8, 32
    v = clustered_vectors(n, d, 91)
    off, nb = knn_graph(v, 24, seed=7)
    cb, codes = make_pq(v, m, K, 15)
    csr = orc.Csr(off, nb, entry_point=2)
    idx = make_index(csr, v)
    pq = attach_pq(idx, cb, codes)
    q = clustered_vectors(4000, d, 92)
    assert_same(orc, idx, csr, v, cb, codes, q, 5, 24, 0.3)
    assert_same(orc, idx, csr, v, cb, codes, q[:12], 20, 600, 0.2)
    assert_same(orc, idx, csr, v, cb, codes, q[:12], 700, 16, 0.5)     # ef = max(ef, k)
    del pq
