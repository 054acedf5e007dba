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


def test_two_level_bf16_rows_bf16_valued_and_mixed_queries(orc):
    """bf16 rows: queries whose elements are all bf16 values are answered by the instantiation that
    keeps the query as bf16 in LDS, the others by the float32-query one -- in one call, with the same
    answers as the oracle either way (d large enough for the two to share a visited-table size)."""
    n, d, m, K = 1500, 1024, 16, 64
    v = clustered_vectors(n, d, 53, per_cluster=50)
    vb = (v.view(np.uint32) >> 16).astype(np.uint16)
    vw = (vb.astype(np.uint32) << 16).view(np.float32)
    off, nb = knn_graph(vw, 16, seed=5)
    cb, codes = make_pq(vw, m, K, 9)
    csr = orc.Csr(off, nb, entry_point=0)
    g = ia.CsrGraph(node_offsets=csr.node_offsets, neighbors=csr.neighbors, levels=csr.levels,
                    entry_point=0, num_nodes=n, degree_counts=csr.degree_counts)
    idx = ia.LeannIndex.from_csr(g, None, dimension=d).upload(0)
    idx.set_embeddings_bf16(vb)
    pq = attach_pq(idx, cb, codes)
    q = clustered_vectors(12, d, 54, per_cluster=50)
    qb = ((q.view(np.uint32) >> 16) << 16).view(np.float32).copy()   # bf16-valued
    mixed = qb.copy()
    mixed[::3] = q[::3]                                              # every third query is not
    for qq in (qb, mixed, q):
        assert_same(orc, idx, csr, vw, cb, codes, qq, 10, 64, 0.3)
    del pq


def _recompute_pair(orc, cache_rows=None, n=1600, deg=20):
    from test_gpu_encoder import _recompute_case
    cfg, enc, tok, lens, emb = _recompute_case(orc, n=n, seed=11, min_len=9)
    d = emb.shape[1]
    off, nb = random_csr(n, deg, 3)
    csr = orc.Csr(off, nb, entry_point=5)
    cb, codes = make_pq(emb, 8, 64, 12)
    g = ia.CsrGraph(node_offsets=csr.node_offsets, neighbors=csr.neighbors, levels=csr.levels, entry_point=5,
                    num_nodes=n, degree_counts=csr.degree_counts)
    mem_idx = ia.LeannIndex.from_csr(g, None, dimension=d).upload(0)
    mem_idx.set_embeddings(emb)
    pq = attach_pq(mem_idx, cb, codes)
    rec_idx = ia.LeannIndex.from_csr(g, None, dimension=d).upload(0)
    rec_idx.set_recompute_provider(enc, tok, lens, cache_rows=cache_rows)
    rec_idx.set_pq_codes(pq, codes)
    return emb, csr, cb, codes, mem_idx, rec_idx, pq, enc


def test_two_level_recompute_parks_and_resumes_with_a_small_row_cache(orc):
    """The two-level search over the recompute provider parks a query at the promotion whose rows
    are not in the row cache and resumes it there (round 2 re-ran it from its start and needed its
    whole traversal resident): a cache of 512 rows for 1600 nodes turns over inside the call, and
    ids, distance bits and counters still equal the in-memory provider's.  ratio 1 with a small ef
    also takes the window retry (alone, state block resized) over the recompute rounds."""
    emb, csr, cb, codes, mem_idx, rec_idx, pq, enc = _recompute_pair(orc, cache_rows=512)
    q = emb[::53] + np.float32(0.02)
    for (k, ef, ratio) in ((10, 64, 0.4), (10, 200, 0.3), (3, 6, 1.0)):
        want = mem_idx.search_two_level_batch(q, k, ef, ratio)
        want_stats = mem_idx.last_stats()
        got = rec_idx.search_two_level_batch(q, k, ef, ratio)
        st = rec_idx.last_stats()
        assert got[2].tolist() == want[2].tolist() and got[0].tolist() == want[0].tolist(), (k, ef, ratio)
        assert got[1].view(np.uint32).tolist() == want[1].view(np.uint32).tolist()
        for f in ("expansions", "edges", "evals", "pushes"):
            assert st[f] == want_stats[f], (f, k, ef, ratio)
        assert st["recompute_rounds"] > 2 and st["encoded_nodes"] > 0
    # the slab turned over: more rows were encoded than it holds
    got = rec_idx.search_two_level_batch(q, 10, 200, 0.6)
    assert rec_idx.last_stats()["encoded_nodes"] >= 512
    for i in range(0, q.shape[0], 7):
        r = orc.two_level_search(csr, emb, cb, codes, q[i], 10, 200, 0.6)
        c = int(got[2][i])
        assert got[0][i, :c].tolist() == r.ids.tolist()
    del pq


def test_two_level_recompute_names_the_nodes_it_will_promote_next(orc, monkeypatch):
    """ISL_TL_PREFETCH=n: a query that parks on an absent row also names the next n unpromoted entries of its
    queue, which the provider encodes in the same round -- fewer rounds, some nodes encoded that are never
    asked for, and every id, distance bit and counter of the answers unchanged (with a row for every node and
    with a cache that turns over)."""
    for cache_rows in (None, 512):
        emb, csr, cb, codes, mem_idx, rec_idx, pq, enc = _recompute_pair(orc, cache_rows=cache_rows)
        q = emb[::41] + np.float32(0.02)
        rounds = {}
        for (k, ef, ratio) in ((10, 64, 0.1), (10, 200, 0.05), (5, 100, 0.4)):
            want = mem_idx.search_two_level_batch(q, k, ef, ratio)
            want_stats = mem_idx.last_stats()
            for n in ("0", "2", "8"):
                monkeypatch.setenv("ISL_TL_PREFETCH", n)
                got = rec_idx.search_two_level_batch(q, k, ef, ratio)
                st = rec_idx.last_stats()
                assert got[2].tolist() == want[2].tolist() and got[0].tolist() == want[0].tolist(), (k, ef, ratio, n)
                assert got[1].view(np.uint32).tolist() == want[1].view(np.uint32).tolist(), (k, ef, ratio, n)
                for f in ("expansions", "edges", "evals", "pushes"):
                    assert st[f] == want_stats[f], (f, k, ef, ratio, n)
                rounds[(k, ef, ratio, n)] = (st["recompute_rounds"], st["encoded_nodes"])
            if cache_rows is None:  # nothing is evicted: naming nodes ahead can only save rounds and only add encodes
                assert rounds[(k, ef, ratio, "8")][0] <= rounds[(k, ef, ratio, "0")][0]
                assert rounds[(k, ef, ratio, "8")][1] >= rounds[(k, ef, ratio, "0")][1]
        del pq


def test_two_level_async_calls_overlap_and_retry(orc):
    """isl_search_two_level_batch_device_async: several calls in flight on the index's lanes, each
    completed by its token with its own counters; one of them needs the window retry."""
    import torch
    n, d, m, K = 3000, 32, 8, 32
    v = clustered_vectors(n, d, 71)
    off, nb = random_csr(n, 64, 12)
    cb, codes = make_pq(v, m, K, 13)
    csr = orc.Csr(off, nb, entry_point=1)
    idx = make_index(csr, v)
    pq = attach_pq(idx, cb, codes)
    calls = [(clustered_vectors(40, d, 100 + i), kk, ef, a) for i, (kk, ef, a) in
             enumerate([(5, 64, 0.3), (3, 6, 1.0), (10, 128, 0.5), (5, 64, 0.3), (10, 32, 0.2)])]
    outs, toks = [], []
    for (q, kk, ef, a) in calls:
        dq = torch.from_numpy(q).cuda()
        o = (torch.zeros((q.shape[0], kk), dtype=torch.int64, device="cuda"),
             torch.zeros((q.shape[0], kk), dtype=torch.float32, device="cuda"),
             torch.zeros(q.shape[0], dtype=torch.int32, device="cuda"))
        outs.append((dq, o))
        toks.append(idx.search_two_level_batch_device_async(dq.data_ptr(), q.shape[0], d, kk, ef, a, o[0].data_ptr(),
                                                           o[1].data_ptr(), o[2].data_ptr()))
    for (q, kk, ef, a), (dq, o), t in zip(calls, outs, toks):
        st = idx.wait_stats(t)
        assert st["queries"] == q.shape[0]
        ids, dist, cnt = o[0].cpu().numpy(), o[1].cpu().numpy(), o[2].cpu().numpy()
        ev = 0
        for i in range(q.shape[0]):
            r = orc.two_level_search(csr, v, cb, codes, q[i], kk, ef, a)
            c = int(cnt[i])
            assert ids[i, :c].tolist() == r.ids.tolist(), (kk, ef, a, i)
            assert bits(dist[i, :c]).tolist() == bits(r.dist).tolist()
            ev += r.counters["evals"]
        assert st["evals"] == ev
    with pytest.raises(ia.CoreError):
        idx.wait(toks[0])
    del pq


def test_recompute_index_on_the_async_entry_points(orc):
    """isl_search_batch_async / isl_search_batch_device_async accept an index with the recompute
    provider (round 2 refused them): the rounds run on a host thread of the library's, the token's
    wait hands over status, answers and counters -- those of the synchronous call."""
    import torch
    emb, csr, cb, codes, mem_idx, rec_idx, pq, enc = _recompute_pair(orc)
    q1, q2 = emb[::53] + np.float32(0.02), emb[5::71] + np.float32(0.01)
    want1, want2 = mem_idx.search_batch(q1, 10, 64), mem_idx.search_batch(q2, 5, 32)
    t1 = rec_idx.search_batch_async(q1, 10, 64)
    t2 = rec_idx.search_batch_async(q2, 5, 32)          # second call: queued behind the first one's rounds
    g2, g1 = rec_idx.wait(t2), rec_idx.wait(t1)
    for got, want in ((g1, want1), (g2, want2)):
        assert got[2].tolist() == want[2].tolist() and got[0].tolist() == want[0].tolist()
        assert got[1].view(np.uint32).tolist() == want[1].view(np.uint32).tolist()
    dq = torch.from_numpy(q1).cuda()
    o = (torch.zeros((q1.shape[0], 10), dtype=torch.int64, device="cuda"),
         torch.zeros((q1.shape[0], 10), dtype=torch.float32, device="cuda"),
         torch.zeros(q1.shape[0], dtype=torch.int32, device="cuda"))
    t = rec_idx.search_batch_device_async(dq.data_ptr(), q1.shape[0], emb.shape[1], 10, 64, o[0].data_ptr(),
                                          o[1].data_ptr(), o[2].data_ptr())
    st = rec_idx.wait_stats(t)
    assert st["recompute_rounds"] > 2 and st["encoded_nodes"] > 0
    assert o[0].cpu().numpy().astype(np.uint64).tolist() == want1[0].tolist()
    # isl_search_stream_wait on such a token waits for the worker on the host; the answers are then there
    # for work enqueued on the stream afterwards, and the token still completes once
    import ctypes as C
    from islands_amd import _check, _ffi
    o[0].zero_()
    t = rec_idx.search_batch_device_async(dq.data_ptr(), q1.shape[0], emb.shape[1], 10, 64, o[0].data_ptr(),
                                          o[1].data_ptr(), o[2].data_ptr())
    side = torch.cuda.Stream()
    _check(_ffi.lib().isl_search_stream_wait(rec_idx._h, t, C.c_void_p(side.cuda_stream)))
    with torch.cuda.stream(side):
        copy = o[0].clone()
    side.synchronize()
    assert copy.cpu().numpy().astype(np.uint64).tolist() == want1[0].tolist()
    rec_idx.wait(t)
    with pytest.raises(ia.CoreError):
        rec_idx.wait(t)
    del pq
