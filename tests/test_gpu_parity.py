"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  Integer/ID outputs must match exactly; distances are required to be
BIT-IDENTICAL (the kernels keep the reference's f32 operation order), which is stricter
than the 1e-5 the north star asks for."""
import numpy as np
import pytest

import islands_amd as ia
from _data import clustered_vectors, knn_graph, random_csr, random_levels, uniform_vectors

pytestmark = pytest.mark.gpu

METRICS = [ia.DistanceMetric.Cosine, ia.DistanceMetric.Euclidean, ia.DistanceMetric.DotProduct,
           ia.DistanceMetric.Manhattan]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def make_index(orc_csr, vectors, cfg=None, dimension=-1):
    g = ia.CsrGraph(node_offsets=orc_csr.node_offsets, neighbors=orc_csr.neighbors,
                    levels=orc_csr.levels, entry_point=orc_csr.entry_point,
                    max_level=orc_csr.max_level, num_nodes=orc_csr.num_nodes,
                    degree_counts=orc_csr.degree_counts)
    d = vectors.shape[1] if dimension == -1 else dimension
    idx = ia.LeannIndex.from_csr(g, cfg, dimension=d)
    idx.upload(0)
    idx.set_embeddings(vectors)
    return idx


def assert_same_search(orc, idx, csr, vectors, queries, k, ef, **okw):
    ids, dist, cnt = idx.search_batch(queries, k, ef)
    st = idx.last_stats()
    tot = {"expansions": 0, "edges": 0, "evals": 0, "pushes": 0}
    for i, q in enumerate(queries):
        r = orc.leann_search(csr, vectors, q, k, ef, **okw)
        assert r.status == 0
        n = int(cnt[i])
        assert n == r.ids.size, (i, n, r.ids.size)
        assert ids[i, :n].tolist() == r.ids.tolist(), (i, ids[i, :n], r.ids)
        assert bits(dist[i, :n]).tolist() == bits(r.dist).tolist(), (i, dist[i, :n], r.dist)
        for f in tot:
            tot[f] += r.counters[f]
    return st, tot


# ---------------------------------------------------------------- distance.rs
def test_distance_kats():  # distance.rs:150-261, 354-372
    C, E, D, M = METRICS
    assert abs(ia.calculate(C, [1, 2, 3], [1, 2, 3])) < 1e-6
    assert abs(ia.calculate(C, [1, 0], [0, 1]) - 1) < 1e-6
    assert abs(ia.calculate(C, [1, 0], [-1, 0]) - 2) < 1e-6
    assert abs(ia.calculate(E, [1, 2, 3], [1, 2, 3])) < 1e-6
    assert abs(ia.calculate(E, [0, 0], [3, 4]) - 5) < 1e-6
    assert ia.calculate(E, [1, 0], [0, 1]) == np.float32(1.414213562373095)
    assert ia.calculate(M, [1, 1], [1, 1]) == 0.0 and ia.calculate(M, [0, 0], [3, 4]) == 7.0
    assert ia.calculate(C, [0, 0, 0], [1, 2, 3]) == 1.0
    assert abs(ia.calculate(D, [1, 2, 3], [4, 5, 6]) + 32) < 1e-6
    assert abs(ia.calculate_squared(E, [0, 0], [3, 4]) - 25) < 1e-6
    d = ia.calculate(C, [1, 0], [0, 1])
    assert abs(ia.calculate_squared(C, [1, 0], [0, 1]) - d * d) < 1e-6
    b = ia.batch_calculate(C, [1, 0], [[1, 0], [0, 1], [-1, 0]])
    assert abs(b[0]) < 1e-6 and abs(b[1] - 1) < 1e-6 and abs(b[2] - 2) < 1e-6
    assert ia.batch_calculate(C, [1, 0], np.zeros((0, 2), np.float32)).size == 0
    for m in METRICS:
        with pytest.raises(ia.CoreError) as e:
            ia.calculate(m, [1, 2], [1, 2, 3])
        assert e.value.kind == "DimensionMismatch"
    with pytest.raises(ia.CoreError):
        ia.calculate_squared(E, [1, 2], [1, 2, 3])


@pytest.mark.parametrize("d", [1, 3, 5, 63, 64, 65, 128, 255, 768, 1000])
def test_batch_distance_bit_exact(orc, d):
    rng = np.random.default_rng(d)
    q = rng.standard_normal(d).astype(np.float32)
    rows = rng.standard_normal((131, d)).astype(np.float32)
    rows[7] = 0.0
    rows[9] = q
    for m in METRICS:
        got = ia.batch_calculate(m, q, rows)
        exp = orc.batch_distance(int(m), q, rows)
        assert bits(got).tolist() == bits(exp).tolist(), (m, d)


def test_sqrt_div_correctly_rounded(orc):
    """Euclidean ends in sqrtf and cosine in a division: both must be IEEE-correct on gfx950."""
    rng = np.random.default_rng(5)
    rows = (rng.standard_normal((4096, 8)) * np.exp(rng.uniform(-20, 20, (4096, 1)))).astype(np.float32)
    q = rng.standard_normal(8).astype(np.float32)
    for m in (ia.DistanceMetric.Euclidean, ia.DistanceMetric.Cosine):
        assert bits(ia.batch_calculate(m, q, rows)).tolist() == \
            bits(orc.batch_distance(int(m), q, rows)).tolist()


def test_normalize_rows(orc):  # distance.rs:231-248
    rng = np.random.default_rng(2)
    rows = rng.standard_normal((70, 32)).astype(np.float32)
    rows[3] = 0
    got = ia.normalize_rows(rows)
    for i in range(rows.shape[0]):
        assert bits(got[i]).tolist() == bits(orc.normalize(rows[i])).tolist()
    v = ia.normalize_rows([[3.0, 4.0, 0.0, 0.0]])[0]
    assert abs(float(np.sqrt((v * v).sum())) - 1) < 1e-6


# -------------------------------------------------------------------- search
def test_reference_search_tests(orc):
    """leann.rs:1290-1343, 1515-1531 restated against the HIP path."""
    v = uniform_vectors(100, 16, 42)
    csr = orc.leann_build(v, levels=random_levels(100, 30, 43))
    idx = make_index(csr, v)
    res = idx.search(v[0], 5, ia.InMemoryEmbeddingProvider(v))
    assert len(res) == 5 and res[0][0] == 0 and res[0][1] < 0.01
    res = idx.search_with_params([0.5] * 16, 10, 64)
    assert all(res[i][1] <= res[i + 1][1] for i in range(len(res) - 1))
    with pytest.raises(ia.CoreError) as e:  # leann.rs:1315-1325
        idx.search_with_params([0.5] * 8, 5, 64)
    assert e.value.kind == "DimensionMismatch" and (e.value.expected, e.value.actual) == (16, 8)
    for n, d in ((10, 8), (50, 16), (100, 32)):
        vv = uniform_vectors(n, d, 42)
        c2 = orc.leann_build(vv, levels=random_levels(n, 30, 1))
        i2 = make_index(c2, vv)
        assert len(i2.search(vv[0], min(5, n))) == min(5, n)


def test_empty_index_and_not_built():
    idx = ia.LeannIndex.with_defaults()  # leann.rs:1306-1313
    ids, dist, cnt = idx.search_batch(np.full((3, 8), 0.5, np.float32), 5, 64)
    assert cnt.tolist() == [0, 0, 0]
    g = ia.CsrGraph(node_offsets=np.array([0, 0], np.uint64), num_nodes=1, entry_point=None,
                    levels=np.zeros(1, np.uint64), degree_counts=np.zeros(1, np.uint64))
    i2 = ia.LeannIndex.from_csr(g, dimension=4).upload(0)
    i2.set_embeddings(np.ones((1, 4), np.float32))
    with pytest.raises(ia.CoreError) as e:  # leann.rs:889
        i2.search_with_params([1, 2, 3, 4], 1, 4)
    assert e.value.kind == "IndexNotBuilt"


@pytest.mark.parametrize("ef", [1, 8, 64, 100, 128, 200, 300, 600])
def test_search_matches_oracle_built_graph(orc, ef):
    n, d = 400, 24
    v = uniform_vectors(n, d, 7)
    csr = orc.leann_build(v, m=8, m0=16, ef_construction=40, levels=random_levels(n, 8, 8))
    idx = make_index(csr, v, ia.LeannConfig(m=8, m0=16, ef_construction=40))
    queries = np.concatenate([uniform_vectors(20, d, 9), v[:4], np.full((1, d), 0.5, np.float32)])
    st, tot = assert_same_search(orc, idx, csr, v, queries, min(10, ef), ef)
    for f in ("expansions", "edges", "evals", "pushes"):
        assert st[f] == tot[f], (f, st, tot)


@pytest.mark.parametrize("metric", METRICS)
def test_search_all_metrics(orc, metric):
    n, d = 300, 20
    v = uniform_vectors(n, d, 11)
    csr = orc.leann_build(v, m=8, m0=16, ef_construction=40, metric=int(metric),
                          levels=random_levels(n, 8, 12))
    idx = make_index(csr, v, ia.LeannConfig(m=8, m0=16, ef_construction=40, metric=metric))
    assert_same_search(orc, idx, csr, v, uniform_vectors(16, d, 13), 10, 48, metric=int(metric))


@pytest.mark.parametrize("strategy", [ia.PruningStrategy.Global, ia.PruningStrategy.Local])
@pytest.mark.parametrize("ratio", [0.3, 0.5, 0.8])
def test_search_with_pruning(orc, strategy, ratio):  # leann.rs:1437-1464, 1553-1572
    n, d = 200, 16
    v = uniform_vectors(n, d, 21)
    csr = orc.leann_build(v, m=8, m0=16, ef_construction=40, levels=random_levels(n, 8, 22))
    cfg = ia.LeannConfig(m=8, m0=16, ef_construction=40, prune_ratio=ratio,
                         pruning_strategy=strategy)
    idx = make_index(csr, v, cfg)
    assert_same_search(orc, idx, csr, v, uniform_vectors(12, d, 23), 5, 32,
                       prune_ratio=ratio, strategy=int(strategy))
    assert len(idx.search_with_params(v[0], 5, 64)) == 5


def test_ties_duplicates_and_dup_rows_take_exact_path(orc):
    """Duplicated vectors give exact distance ties; duplicated ids inside adjacency rows are
    always 'already visited' on their second occurrence.  IDs must still match the oracle's
    BinaryHeap-order tie-breaking."""
    for seed, dup_rows in ((0, False), (1, True)):
        n, d = 240, 12
        v = uniform_vectors(n, d, 30 + seed)
        v[n // 2:] = v[: n - n // 2]
        off, nb = random_csr(n, 12, 40 + seed, dup=dup_rows)
        csr = orc.Csr(off, nb, entry_point=5)
        idx = make_index(csr, v)
        queries = uniform_vectors(24, d, 50 + seed)
        for ef in (4, 16, 64, 130):
            st, _ = assert_same_search(orc, idx, csr, v, queries, min(10, ef), ef)
        assert st["exact_path"] + st["replayed"] > 0


@pytest.mark.parametrize("deg,ef,metric", [(96, 128, 0), (128, 64, 1), (65, 16, 2), (100, 300, 3), (127, 600, 0)])
def test_long_rows_on_the_fast_path(orc, deg, ef, metric):
    """LeannConfig::accurate() has m0 = 96 (leann.rs:419-429): adjacency rows of up to 128 ids are
    answered by the wave-per-query kernel (two ids per lane, neighbours evaluated 64 at a time in
    CSR order), not by the lane-0 heap emulation."""
    n, d = 1500, 24
    v = uniform_vectors(n, d, 60 + deg)
    off, nb = random_csr(n, deg, 61)
    csr = orc.Csr(off, nb, entry_point=0)
    cfg = ia.LeannConfig.accurate()
    cfg.metric = METRICS[metric]
    idx = make_index(csr, v, cfg)
    q = uniform_vectors(24, d, 62)
    st, tot = assert_same_search(orc, idx, csr, v, q, 10, ef, metric=metric)
    assert st["exact_path"] == 0 or ef > 512  # distinct distances: nothing for the heap-exact kernel
    for f in tot:
        assert st[f] == tot[f], f


def test_long_rows_with_pruning_and_ties(orc):
    """Rows of 65..128 ids with prune_ratio > 0 (the keep prefix is computed over the whole row,
    leann.rs:991-1016) and quantised vectors (equal distances everywhere: tie replay / exact kernel)."""
    n, d = 900, 8
    rng = np.random.default_rng(5)
    v = rng.integers(-2, 3, (n, d)).astype(np.float32)
    v[np.abs(v).sum(1) == 0, 0] = 1
    off, nb = random_csr(n, 110, 7)
    csr = orc.Csr(off, nb, entry_point=3)
    for strategy in (ia.PruningStrategy.Global, ia.PruningStrategy.Local):
        cfg = ia.LeannConfig(m=48, m0=110, ef_construction=400, prune_ratio=0.4, pruning_strategy=strategy,
                             metric=ia.DistanceMetric.Euclidean)
        idx = make_index(csr, v, cfg)
        q = rng.integers(-2, 3, (16, d)).astype(np.float32)
        assert_same_search(orc, idx, csr, v, q, 7, 40, metric=1, prune_ratio=0.4, strategy=int(strategy))


def test_rows_beyond_128_ids_use_the_exact_kernel(orc):
    n, d = 400, 16
    v = uniform_vectors(n, d, 60)
    off, nb = random_csr(n, 140, 61)
    csr = orc.Csr(off, nb, entry_point=0)
    idx = make_index(csr, v, ia.LeannConfig(m=48, m0=140, ef_construction=400))
    st, _ = assert_same_search(orc, idx, csr, v, uniform_vectors(8, d, 62), 10, 128)
    assert st["exact_path"] == 8


def test_node_not_found_and_unreachable(orc):
    v = uniform_vectors(3, 4, 0)
    csr = orc.Csr(node_offsets=[0, 1, 2, 2], neighbors=[1, 0], entry_point=0)
    idx = make_index(csr, v)
    res = idx.search_with_params([0.1] * 4, 3, 8)  # node 2 unreachable -> fewer than k
    assert sorted(r[0] for r in res) == [0, 1]
    bad = orc.Csr(node_offsets=[0, 1, 1], neighbors=[7], entry_point=0)
    i2 = make_index(bad, uniform_vectors(2, 4, 0))
    with pytest.raises(ia.CoreError) as e:  # provider miss, leann.rs:145-150
        i2.search_with_params([0.1] * 4, 1, 4)
    assert e.value.kind == "NodeNotFound" and e.value.node == 7
    # graph smaller than the provider: ids >= num_nodes have vectors but no adjacency
    v5 = uniform_vectors(5, 4, 1)
    g2 = orc.Csr(node_offsets=[0, 2, 3], neighbors=[1, 4, 3], entry_point=0)
    i3 = make_index(g2, v5)
    assert_same_search(orc, i3, g2, v5, uniform_vectors(3, 4, 2), 5, 8)


def test_bincode_roundtrip_then_search(orc):  # leann.rs:1347-1364
    n, d = 120, 16
    v = uniform_vectors(n, d, 70)
    csr = orc.leann_build(v, levels=random_levels(n, 30, 71))
    idx = make_index(csr, v)
    restored = ia.LeannIndex.from_bytes(idx.to_bytes())
    assert len(restored) == len(idx) and restored.dimension() == idx.dimension()
    restored.upload(0).set_embeddings(v)
    q = uniform_vectors(6, d, 72)
    a = idx.search_batch(q, 5, 64)
    b = restored.search_batch(q, 5, 64)
    assert a[0].tolist() == b[0].tolist() and bits(a[1]).tolist() == bits(b[1]).tolist()


def test_mid_size_d768_matches_oracle(orc):
    """BASELINE config 1 shape (10k nodes, d=768, M=30, efSearch=64) plus ef=128."""
    n, d = 10000, 768
    v = clustered_vectors(n, d, 80, per_cluster=100)
    off, nb = knn_graph(v, 60, seed=81)
    csr = orc.Csr(off, nb, entry_point=0)
    idx = make_index(csr, v)
    queries = clustered_vectors(48, d, 82, per_cluster=4)
    for ef in (64, 128):
        st, tot = assert_same_search(orc, idx, csr, v, queries, 10, ef)
        for f in ("expansions", "edges", "evals"):
            assert st[f] == tot[f]


def test_visited_table_grows_with_the_evaluations_the_index_sees(orc):
    """The size of a query's visited table in LDS follows the evaluations per query of the index's previous call
    (fast_geometry, round 4): the first call over this graph -- ~60 fresh neighbours per expansion, several
    thousand evaluations per query -- runs with the table ef alone gives (most queries spill into the HBM overflow
    table), the next ones with the next size up.  Same ids, distance bits and counters every time."""
    n, d, deg = 9000, 24, 60
    v = uniform_vectors(n, d, 17)
    off, nb = random_csr(n, deg, 23)
    csr = orc.Csr(off, nb, entry_point=7)
    idx = make_index(csr, v)
    q = uniform_vectors(48, d, 29)
    evals = []
    for rep in range(3):
        st, tot = assert_same_search(orc, idx, csr, v, q, 10, 128)
        for f in tot:
            assert st[f] == tot[f], (f, rep)
        evals.append(st["evals"] / q.shape[0])
    assert evals[0] > 1792 * 1.2  # (the case does exercise a table past its 7/8 limit)


def test_device_born_csr_and_device_buffers(orc):
    torch = pytest.importorskip("torch")
    n, d = 2000, 64
    v = clustered_vectors(n, d, 90)
    off, nb = knn_graph(v, 24, seed=91)
    csr = orc.Csr(off, nb, entry_point=17)
    dev = torch.device("cuda:0")
    t_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    t_nb = torch.from_numpy(nb.astype(np.int64)).to(torch.int32).to(dev)
    t_v = torch.from_numpy(v).to(dev)
    idx = ia.LeannIndex.from_device_csr(t_off.data_ptr(), t_nb.data_ptr(), n, 17, d)
    idx.set_embeddings(None, device_ptr=t_v.data_ptr(), n=n, d=d)
    q = clustered_vectors(32, d, 92)
    t_q = torch.from_numpy(q).to(dev)
    k, ef = 10, 64
    o_ids = torch.zeros((32, k), dtype=torch.int64, device=dev)
    o_dist = torch.zeros((32, k), dtype=torch.float32, device=dev)
    o_cnt = torch.zeros(32, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    idx.search_batch_device(t_q.data_ptr(), 32, d, k, ef, o_ids.data_ptr(), o_dist.data_ptr(),
                            o_cnt.data_ptr())
    for i in range(32):
        r = orc.leann_search(csr, v, q[i], k, ef)
        assert o_ids[i].cpu().numpy().tolist() == r.ids.tolist()
        assert bits(o_dist[i].cpu().numpy()).tolist() == bits(r.dist).tolist()
    # host CSR is materialised lazily from the device copy
    assert idx.get_neighbors(5).tolist() == csr.get_neighbors(5)
    assert ia.LeannIndex.from_bytes(idx.to_bytes()).get_neighbors(9).tolist() == csr.get_neighbors(9)


# ----------------------------------------------------------------- search.rs
def test_merge_topk_matches_multi_index_searcher(orc):  # search.rs:211-237
    rng = np.random.default_rng(3)
    nl, nq, k = 4, 9, 6
    scores = np.sort(rng.integers(0, 8, (nl, nq, k)).astype(np.float32) / 4, axis=2)
    ids = rng.integers(0, 1000, (nl, nq, k)).astype(np.uint64)
    counts = rng.integers(0, k + 1, (nl, nq)).astype(np.uint32)
    base = np.array([0, 1000, 2000, 3000], np.uint64)
    oi, osc, osrc, oc = ia.merge_topk(ids, scores, counts, 10, id_base=base)
    for q in range(nq):
        lists_i = [ids[l, q, :counts[l, q]] + base[l] for l in range(nl)]
        lists_s = [scores[l, q, :counts[l, q]] for l in range(nl)]
        st, ei, es, esrc = orc.multi_index_merge(lists_i, lists_s, 10)
        n = int(oc[q])
        assert n == ei.size and oi[q, :n].tolist() == ei.tolist()
        assert osrc[q, :n].tolist() == esrc.tolist() and bits(osc[q, :n]).tolist() == bits(es).tolist()


def test_merge_topk_packed_records_equal_the_dense_merge(orc):
    """The multi-GPU exchange form of the same merge (search.rs:211-237): every shard's answers as
    one packed record (ids | distances | counts, isl_shard_record_bytes), records a stride apart as
    an all-gather leaves them, merged by isl_merge_topk_packed_async with no host wait."""
    import ctypes as C

    import torch

    from islands_amd import _ffi
    from islands_amd.sharded import record_bytes, record_views

    rng = np.random.default_rng(8)
    for (nl, nq, k, top) in ((3, 17, 6, 6), (8, 5, 10, 10), (2, 33, 1, 1)):
        scores = np.sort(rng.integers(0, 8, (nl, nq, k)).astype(np.float32) / 4, axis=2)
        ids = rng.integers(0, 1000, (nl, nq, k)).astype(np.uint64)
        counts = rng.integers(0, k + 1, (nl, nq)).astype(np.uint32)
        base = (np.arange(nl) * 1000).astype(np.uint64)
        B = record_bytes(nq, k)
        assert B == _ffi.lib().isl_shard_record_bytes(nq, k) and B % 16 == 0
        gathered = torch.zeros((nl, B), dtype=torch.uint8)
        g_ids, g_dd, g_cnt = record_views(gathered, nq, k)
        g_ids.copy_(torch.from_numpy(ids.astype(np.int64)))
        g_dd.copy_(torch.from_numpy(scores))
        g_cnt.copy_(torch.from_numpy(counts.astype(np.int32)))
        dg = gathered.cuda()
        o_ids = torch.zeros((nq, top), dtype=torch.int64, device="cuda")
        o_sc = torch.zeros((nq, top), dtype=torch.float32, device="cuda")
        o_src = torch.zeros((nq, top), dtype=torch.int32, device="cuda")
        o_cnt = torch.zeros(nq, dtype=torch.int32, device="cuda")
        flags = torch.zeros(1, dtype=torch.int32, device="cuda")
        d_base = torch.from_numpy(base.astype(np.int64)).cuda()
        torch.cuda.synchronize()
        ia._check(_ffi.lib().isl_merge_topk_packed_async(
            nl, nq, k, C.c_void_p(dg.data_ptr()), B, C.c_void_p(d_base.data_ptr()), top, C.c_void_p(o_ids.data_ptr()),
            C.c_void_p(o_sc.data_ptr()), C.c_void_p(o_src.data_ptr()), C.c_void_p(o_cnt.data_ptr()),
            C.c_void_p(flags.data_ptr()), 0, None))
        torch.cuda.synchronize()
        assert int(flags.item()) == 0
        wi, ws, wsrc, wc = ia.merge_topk(ids, scores, counts, top, id_base=base)
        assert o_cnt.cpu().numpy().astype(np.uint32).tolist() == wc.tolist()
        for q in range(nq):
            n = int(wc[q])
            assert o_ids[q, :n].cpu().numpy().astype(np.uint64).tolist() == wi[q, :n].tolist()
            assert bits(o_sc[q, :n].cpu().numpy()).tolist() == bits(ws[q, :n]).tolist()
            assert o_src[q, :n].cpu().numpy().tolist() == wsrc[q, :n].tolist()


# --------------------------------------------------------------------- pq.rs
def test_pq_matches_oracle(orc):  # pq.rs:639-677, 787-809, 505-520
    rng = np.random.default_rng(4)
    for (m, K, dsub) in ((8, 256, 16), (4, 37, 5), (64, 256, 64)):
        d = m * dsub
        cb = rng.standard_normal((m, K, dsub)).astype(np.float32)
        pq = ia.ProductQuantizer(d, cb)
        vecs = rng.standard_normal((40, d)).astype(np.float32)
        vecs[:K if K < 40 else 40] = cb[:, :min(K, 40)].transpose(1, 0, 2).reshape(-1, d)[:40]
        q = rng.standard_normal((3, d)).astype(np.float32)
        codes = pq.encode(vecs)
        for i in range(vecs.shape[0]):
            assert codes[i].tolist() == orc.pq_encode(orc.EUCLIDEAN, cb, vecs[i])[1].tolist()
        tabs = pq.build_distance_tables(q)
        for i in range(3):
            assert bits(tabs[i]).tolist() == bits(orc.pq_build_tables(cb, q[i])[1]).tolist()
        td = pq.table_distance(tabs[0], codes)
        ad = pq.asymmetric_distance(q[0], codes)
        for i in range(vecs.shape[0]):
            assert bits(td[i:i + 1])[0] == bits([orc.pq_table_distance(tabs[0], codes[i])])[0]
            assert bits(ad[i:i + 1])[0] == bits([orc.pq_asymmetric_distance(cb, q[0], codes[i])[1]])[0]
        assert np.all(np.abs(td - ad) < 1e-3)
    pq8 = ia.ProductQuantizer(128, np.zeros((8, 256, 16), np.float32))
    assert pq8.compression_ratio() == 64.0 and pq8.bytes_per_vector() == 8
    assert ia.ProductQuantizer(128, np.zeros((8, 300, 16), np.float32)).bytes_per_vector() == 16
    with pytest.raises(ia.CoreError) as e:
        pq8.table_distance(np.zeros((8, 256), np.float32), np.full((1, 8), 300, np.uint16))
    assert e.value.kind == "PQError"
    with pytest.raises(ia.CoreError) as e:
        pq8.build_distance_tables(np.zeros(64, np.float32))
    assert e.value.kind == "DimensionMismatch"


# ------------------------------------------------ distance matrix on the matrix cores
@pytest.mark.parametrize("nq,n,d", [(300, 1030, 32), (513, 1024, 96), (700, 131074, 64), (600, 100001, 96),
                                    (2048, 4100, 64), (2048, 4100, 96)])
def test_distance_matrix_tile_walk(orc, nq, n, d):
    """The LDS-DMA float32 GEMM (d % 32 == 0): one slab and several, row counts that are and are not
    multiples of 4 (vector and per-element epilogue), ragged last tiles, and -- the two large cases --
    the persistent 256 x 256 variant whose workgroups walk several tiles with an even and an odd
    number of slabs (the buffer parity carried from tile to tile); the last two walk tiles in the
    128 x 128 variant (528 tiles on 512 resident workgroups).  Whole matrix against a float64
    product, sampled query rows against the reference's batch_calculate."""
    rows = clustered_vectors(n, d, 13)
    q = clustered_vectors(nq, d, 14)
    got = ia.distance_matrix(ia.DistanceMetric.Cosine, q, rows)
    qn = np.sqrt((q.astype(np.float64) ** 2).sum(1))
    rn = np.sqrt((rows.astype(np.float64) ** 2).sum(1))
    want = 1.0 - (q.astype(np.float64) @ rows.astype(np.float64).T) / (qn[:, None] * rn[None, :])
    assert got.shape == want.shape
    assert np.abs(got - want).max() < 1e-5
    for i in (0, nq // 2, nq - 1):
        assert np.abs(got[i] - orc.batch_distance(int(ia.DistanceMetric.Cosine), q[i], rows)).max() < 1e-5


@pytest.mark.parametrize("metric", [ia.DistanceMetric.Cosine, ia.DistanceMetric.DotProduct,
                                    ia.DistanceMetric.Euclidean])
@pytest.mark.parametrize("nq,n,d", [(5, 300, 128), (130, 257, 768), (3, 70, 30)])
def test_distance_matrix_matches_batch_calculate(orc, metric, nq, n, d):
    """float32 MFMA accumulation vs the reference's sequential sums: 1e-5 on normalised rows
    (north star tolerance); the Euclidean form |q|^2 + |r|^2 - 2 q.r loses digits near zero, so
    it is compared on the squared distance."""
    rows = clustered_vectors(n, d, 3)
    q = clustered_vectors(nq, d, 4)
    got = ia.distance_matrix(metric, q, rows)
    for i in range(nq):
        want = orc.batch_distance(int(metric), q[i], rows)
        if metric == ia.DistanceMetric.Euclidean:
            assert np.abs(got[i] ** 2 - want ** 2).max() < 1e-5
        else:
            assert np.abs(got[i] - want).max() < 1e-5
    with pytest.raises(ia.CoreError) as e:
        ia.distance_matrix(ia.DistanceMetric.Manhattan, q, rows)
    assert e.value.kind == "Unsupported"


def test_bruteforce_topk(orc):
    n, d, nq, k = 5000, 64, 37, 10
    rows = clustered_vectors(n, d, 8)
    q = clustered_vectors(nq, d, 9)
    ids, dist, cnt = ia.bruteforce_topk(ia.DistanceMetric.Cosine, q, rows, k)
    assert (cnt == k).all()
    for i in range(nq):
        exact = orc.batch_distance(orc.COSINE, q[i], rows)
        order = np.argsort(exact, kind="stable")
        assert np.abs(dist[i] - exact[order[:k]]).max() < 1e-5
        assert np.all(np.diff(dist[i]) >= 0)
        # same ids wherever the next distance is not within rounding of this one
        gaps = np.diff(exact[order[:k + 1]])
        for j in range(k):
            if (j == 0 or gaps[j - 1] > 1e-5) and gaps[j] > 1e-5:
                assert ids[i, j] == order[j]
    few = ia.bruteforce_topk(ia.DistanceMetric.DotProduct, q[:2], rows[:4], 10)
    assert few[2].tolist() == [4, 4]


def _check_topk(ids, dist, row, k):
    """ids / dist = the k smallest of `row` by (distance, id): distances to 2e-6 (another launch shape of
    the GEMM may round differently), ascending, ids exact wherever the order is not within that rounding,
    equal distances in id order."""
    n = row.shape[0]
    order = np.lexsort((np.arange(n), row))[:k + 1]
    assert np.abs(dist - row[order[:k]]).max() < 2e-6
    assert np.all(np.diff(dist) >= 0)
    assert len(set(ids.tolist())) == k and np.abs(row[ids.astype(np.int64)] - dist).max() < 2e-6
    ex = row[order]
    for j in range(k):
        lo_ok = j == 0 or ex[j] - ex[j - 1] > 1e-5
        hi_ok = j + 1 >= ex.size or ex[j + 1] - ex[j] > 1e-5
        if lo_ok and hi_ok:
            assert ids[j] == order[j]
    for j in range(k - 1):
        if dist[j] == dist[j + 1]:
            assert ids[j] < ids[j + 1]


def test_bruteforce_topk_scans_ragged_blocks_and_ties(orc):
    # column counts that are not a multiple of the scan's 4-wide loads or of its 256-column step, a
    # block boundary inside the row set (nq large enough to split the rows into chunks) and rows that
    # repeat (equal distances: the smaller id wins)
    d, k = 32, 12
    for n, nq in ((1, 3), (3, 3), (255, 5), (257, 5), (1030, 9), (70001, 4099)):
        rows = clustered_vectors(n, d, 21)
        if n > 600:
            rows[500:520] = rows[100:120]  # duplicates
        q = clustered_vectors(nq, d, 22)
        ids, dist, cnt = ia.bruteforce_topk(ia.DistanceMetric.Cosine, q, rows, k)
        kk = min(k, n)
        assert (cnt == kk).all()
        dm = ia.distance_matrix(ia.DistanceMetric.Cosine, q[:64], rows)  # the same GEMM (another launch shape)
        for i in range(min(nq, 64)):
            _check_topk(ids[i, :kk], dist[i, :kk], dm[i], kk)


# ---------------------------------------------------------------- bf16 row storage
def to_bf16_bits(a):
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def widen(bits):
    return (bits.astype(np.uint32) << 16).view(np.float32)


def test_bruteforce_topk_bf16_is_exact_under_the_bf16_gemm():
    d, k = 128, 11
    for n, nq in ((300, 7), (5000, 33), (66000, 4100)):
        rows = to_bf16_bits(clustered_vectors(n, d, 31))
        q = to_bf16_bits(clustered_vectors(nq, d, 32))
        ids, dist, cnt = ia.bruteforce_topk_bf16(ia.DistanceMetric.Cosine, q, rows, k)
        assert (cnt == k).all()
        dm = ia.distance_matrix_bf16(ia.DistanceMetric.Cosine, q[:48], rows)
        for i in range(min(nq, 48)):
            _check_topk(ids[i], dist[i], dm[i], k)
    with pytest.raises(ia.CoreError):
        ia.bruteforce_topk_bf16(ia.DistanceMetric.Cosine, q[:, :100], rows[:, :100], k)  # d % 64


@pytest.mark.parametrize("metric", METRICS)
@pytest.mark.parametrize("d", [40, 768, 2100])  # 2100: the query outgrows the visited table (smaller table, more waves)
def test_bf16_rows_match_oracle_on_widened_rows(orc, metric, d):
    """ISL_DTYPE_BF16: the provider's vectors are the exact f32 images of the stored bf16 values,
    the arithmetic is the reference's f32 chain -> ids and distance bits of the oracle run on
    the widened rows."""
    n = 900
    bits = to_bf16_bits(clustered_vectors(n, d, 31))
    rows = widen(bits)
    csr = orc.leann_build(rows, m=8, m0=16, ef_construction=40, metric=int(metric))
    cfg = ia.LeannConfig(m=8, m0=16, ef_construction=40, metric=metric)
    g = ia.CsrGraph(node_offsets=csr.node_offsets, neighbors=csr.neighbors, levels=csr.levels,
                    entry_point=csr.entry_point, max_level=csr.max_level, num_nodes=csr.num_nodes,
                    degree_counts=csr.degree_counts)
    idx = ia.LeannIndex.from_csr(g, cfg, dimension=d)
    idx.upload(0)
    idx.set_embeddings_bf16(bits)
    q = clustered_vectors(24, d, 32)
    st, tot = assert_same_search(orc, idx, csr, rows, q, 10, 48, metric=int(metric))
    for f in tot:
        assert st[f] == tot[f], f
    assert st["exact_path"] == 0
    # Queries whose elements are bf16 values are answered by the instantiation that keeps the query
    # as bf16 in LDS; the others are passed on to the float32-query kernel by the same call.  All
    # bf16-valued, none (above), and a mix:
    qb = widen(to_bf16_bits(clustered_vectors(24, d, 33)))
    mix = qb.copy()
    mix[::3] = clustered_vectors(24, d, 34)[::3]
    mix[5, d - 1] = np.float32(1.0) + np.float32(2.0 ** -20)  # one stray element in the last position
    for qq in (qb, mix):
        st, tot = assert_same_search(orc, idx, csr, rows, qq, 10, 48, metric=int(metric))
        for f in tot:
            assert st[f] == tot[f], f
        assert st["exact_path"] == 0


def test_bf16_rows_ties_take_the_exact_kernel(orc):
    base = to_bf16_bits(uniform_vectors(80, 24, 3))
    bits = np.concatenate([base, base, base[:40]])
    rows = widen(bits)
    csr = orc.leann_build(rows, m=6, m0=12, ef_construction=30)
    g = ia.CsrGraph(node_offsets=csr.node_offsets, neighbors=csr.neighbors, levels=csr.levels,
                    entry_point=csr.entry_point, max_level=csr.max_level, num_nodes=csr.num_nodes,
                    degree_counts=csr.degree_counts)
    idx = ia.LeannIndex.from_csr(g, ia.LeannConfig(m=6, m0=12, ef_construction=30), dimension=24)
    idx.upload(0)
    idx.set_embeddings_bf16(bits)
    st, _ = assert_same_search(orc, idx, csr, rows, rows[:16], 10, 30)
    assert st["exact_path"] + st["replayed"] > 0


def test_randomised_differential_slice(orc):
    """A fixed slice of tests/fuzz_parity.py (random graphs / vectors / metrics / ef / k / pruning,
    a third of the cases with quantised vectors so that equal distances are everywhere)."""
    import fuzz_parity
    rng = np.random.default_rng(2024)
    exact = 0
    for case in range(120):
        exact += fuzz_parity.one_case(rng, case)["exact_path"]
    assert exact > 0  # the slice reaches the heap-exact kernel too


@pytest.mark.parametrize("metric", [ia.DistanceMetric.Cosine, ia.DistanceMetric.DotProduct,
                                    ia.DistanceMetric.Euclidean])
@pytest.mark.parametrize("nq,n,d", [(5, 300, 128), (130, 257, 768), (33, 200, 4096)])
def test_distance_matrix_bf16_matches_batch_calculate(orc, metric, nq, n, d):
    """bf16 rows and queries (BASELINE config 5) on the bf16 matrix cores against the reference's
    sequential float32 sums over the same (exactly widened) values: 1e-5 on normalised rows."""
    def to_bf16(x):
        b = (x.view(np.uint32) >> 16).astype(np.uint16)
        return b, (b.astype(np.uint32) << 16).view(np.float32)
    rb, rw = to_bf16(clustered_vectors(n, d, 3))
    qb, qw = to_bf16(clustered_vectors(nq, d, 4))
    got = ia.distance_matrix_bf16(metric, qb, rb)
    for i in range(nq):
        want = orc.batch_distance(int(metric), qw[i], rw)
        if metric == ia.DistanceMetric.Euclidean:
            assert np.abs(got[i] ** 2 - want ** 2).max() < 1e-5
        else:
            assert np.abs(got[i] - want).max() < 1e-5
    with pytest.raises(ia.CoreError) as e:
        ia.distance_matrix_bf16(ia.DistanceMetric.Manhattan, qb, rb)
    assert e.value.kind == "Unsupported"
    with pytest.raises(ia.CoreError) as e:
        ia.distance_matrix_bf16(metric, qb[:, :40], rb[:, :40])
    assert e.value.kind == "Unsupported"


@pytest.mark.parametrize("metric", [ia.DistanceMetric.Cosine, ia.DistanceMetric.Euclidean, ia.DistanceMetric.DotProduct])
def test_distance_matrix_bf16_large_tile_forms_agree(orc, metric):
    """Shapes that take the 256 x 256 eight-phase kernel (>= 512 tiles, ragged edges): the call that computes the
    sums of squares itself, the one that is handed them (isl_row_sumsq_bf16) and the enqueue-only form over device
    buffers give the same bits; sampled rows agree with the reference's sequential sums to 1e-5.  Rows with
    zero / denormal-product norms sit in some tiles only: those blocks take the careful cosine branch, the
    others the branch-free one (gemm_tn_bf16_ph8, VAR 12)."""
    import ctypes as C
    import torch
    from islands_amd import _ffi
    nq, n, d = 300, 110_000, 128
    def to_bf16(x):
        b = (x.view(np.uint32) >> 16).astype(np.uint16)
        return b, (b.astype(np.uint32) << 16).view(np.float32)
    rows = clustered_vectors(n, d, 3)
    q = clustered_vectors(nq, d, 4)
    rows[5] = 0.0
    rows[70_001] *= np.float32(1e-20)
    rows[70_002] *= np.float32(3e-12)
    q[2] *= np.float32(1e-19)
    q[290] = 0.0
    rb, rw = to_bf16(rows)
    qb, qw = to_bf16(q)
    a = ia.distance_matrix_bf16(metric, qb, rb)
    qs, rs = ia.row_sumsq_bf16(qb), ia.row_sumsq_bf16(rb)
    b = ia.distance_matrix_bf16(metric, qb, rb, q_sumsq=qs, row_sumsq=rs)
    assert (a.view(np.uint32) == b.view(np.uint32)).all()
    dev = torch.device("cuda:0")
    tq, tr = torch.from_numpy(qb.view(np.int16)).to(dev), torch.from_numpy(rb.view(np.int16)).to(dev)
    tqs, trs = torch.from_numpy(qs).to(dev), torch.from_numpy(rs).to(dev)
    out = torch.full((nq, n), float("nan"), device=dev)
    torch.cuda.synchronize()
    ia._check(_ffi.lib().isl_distance_matrix_bf16_enqueue(
        int(metric), C.c_void_p(tq.data_ptr()), nq, C.c_void_p(tr.data_ptr()), n, d, C.c_void_p(tqs.data_ptr()),
        C.c_void_p(trs.data_ptr()), C.c_void_p(out.data_ptr()), 0, None))
    torch.cuda.synchronize()
    c = out.cpu().numpy()
    assert (a.view(np.uint32) == c.view(np.uint32)).all()
    for i in (0, 2, 131, 290, 299):
        want = orc.batch_distance(int(metric), qw[i], rw)
        if metric == ia.DistanceMetric.Euclidean:
            assert np.abs(a[i] ** 2 - want ** 2).max() < 1e-5
        else:
            assert np.abs(a[i] - want).max() < 1e-5
    if metric != ia.DistanceMetric.DotProduct:
        with pytest.raises(ia.CoreError) as e:
            ia._check(_ffi.lib().isl_distance_matrix_bf16_enqueue(
                int(metric), C.c_void_p(tq.data_ptr()), nq, C.c_void_p(tr.data_ptr()), n, d, None, None,
                C.c_void_p(out.data_ptr()), 0, None))
        assert e.value.kind == "InvalidArgument"


def test_distance_matrix_cosine_with_tiny_norms(orc):
    """Rows whose squared norms multiply to a denormal (or to zero): the 1-ulp rsq of the GEMM epilogue
    flushes such an input to zero -- those elements must take the reference's own form,
    1 - dot / sqrt(na * nb), and 1.0 when the product is exactly zero (distance.rs:82-87)."""
    d = 32
    rng = np.random.default_rng(5)
    rows = rng.standard_normal((40, d)).astype(np.float32)
    rows[3] *= np.float32(1e-12)
    rows[7] *= np.float32(3e-11)
    rows[9] = 0.0
    q = rng.standard_normal((6, d)).astype(np.float32)
    q[1] *= np.float32(1e-9)    # |q|^2 |r|^2 ~ 1e-18 * 1e-24 * d^2: below FLT_MIN, above zero
    q[2] *= np.float32(1e-12)   # ... and underflowing to exactly zero for the smallest rows
    got = ia.distance_matrix(ia.DistanceMetric.Cosine, q, rows)
    assert np.isfinite(got).all()
    for i in range(q.shape[0]):
        want = orc.batch_distance(orc.COSINE, q[i], rows)
        assert np.abs(got[i] - want).max() < 1e-5, (i, np.abs(got[i] - want).argmax())
