"""GPU: LeannIndex::build on the device (isl_index_build) against the oracle's restatement of
the reference builder (leann.rs:560-833).  With batch = 1 the construction is the reference's
sequential one: the whole LeannIndex must serialise to the same bytes (config, node_offsets,
neighbors, levels, entry_point, max_level, degree_counts, dimension)."""
import numpy as np
import pytest

import islands_amd as ia
from _data import clustered_vectors, random_levels, uniform_vectors

pytestmark = pytest.mark.gpu


def reference_bytes(orc, v, cfg, levels):
    csr = orc.leann_build(v, m=cfg.m, m0=cfg.m0, ef_construction=cfg.ef_construction,
                          metric=int(cfg.metric), high_degree_pruning=cfg.high_degree_pruning,
                          hub_percentile=cfg.hub_percentile, levels=levels)
    g = ia.CsrGraph(node_offsets=csr.node_offsets, neighbors=csr.neighbors, levels=csr.levels,
                    entry_point=csr.entry_point, max_level=csr.max_level, num_nodes=csr.num_nodes,
                    degree_counts=csr.degree_counts)
    return ia.LeannIndex.from_csr(g, cfg, dimension=v.shape[1]).to_bytes(), csr


@pytest.mark.parametrize("metric", [ia.DistanceMetric.Cosine, ia.DistanceMetric.Euclidean,
                                    ia.DistanceMetric.DotProduct, ia.DistanceMetric.Manhattan])
def test_sequential_build_is_the_reference_graph(orc, metric):
    n, d = 500, 24
    v = clustered_vectors(n, d, 7)
    cfg = ia.LeannConfig(m=8, m0=16, ef_construction=40, metric=metric)
    levels = random_levels(n, 8, 3)
    want, _ = reference_bytes(orc, v, cfg, levels)
    idx = ia.LeannIndex.build(v, cfg, levels=levels, batch=1)
    assert idx.to_bytes() == want


@pytest.mark.parametrize("hub_percentile,high_degree", [(0.02, True), (0.25, True), (0.02, False)])
def test_hub_rule_variants(orc, hub_percentile, high_degree):
    n, d = 700, 16
    v = uniform_vectors(n, d, 11)
    cfg = ia.LeannConfig(m=6, m0=12, ef_construction=48, hub_percentile=hub_percentile,
                         high_degree_pruning=high_degree)
    want, csr = reference_bytes(orc, v, cfg, None)
    idx = ia.LeannIndex.build(v, cfg, batch=1)
    assert idx.to_bytes() == want
    # and the built index answers like the oracle's search over the oracle's graph
    q = uniform_vectors(12, d, 12)
    ids, dist, cnt = idx.search_batch(q, 5, 30)
    for i in range(12):
        r = orc.leann_search(csr, v, q[i], 5, 30)
        assert ids[i, :cnt[i]].tolist() == r.ids.tolist()
        assert dist[i, :cnt[i]].view(np.uint32).tolist() == r.dist.view(np.uint32).tolist()


def test_paper_default_config_and_duplicates(orc):  # leann.rs:1437-1464 shapes; equal rows tie everywhere
    base = uniform_vectors(150, 32, 5)
    v = np.concatenate([base, base[:60]]).astype(np.float32)
    cfg = ia.LeannConfig.paper_default()  # m0 = 60, ef_construction = 128
    want, _ = reference_bytes(orc, v, cfg, None)
    assert ia.LeannIndex.build(v, cfg, batch=1).to_bytes() == want


def test_batched_build_keeps_the_invariants(orc):
    n, d = 3000, 16
    v = uniform_vectors(n, d, 21)  # the reference builder itself reaches recall@1 = 1.0 here
    cfg = ia.LeannConfig(m=8, m0=16, ef_construction=64)
    idx = ia.LeannIndex.build(v, cfg, batch=256)
    assert len(idx) == n and idx.dimension() == d and idx.entry_point == 0
    degs = [len(idx.get_neighbors(i)) for i in range(n)]
    assert max(degs) <= 16 and min(degs[1:]) >= 1
    for i in (1, 17, n - 1):
        nb = idx.get_neighbors(i).tolist()
        assert len(set(nb)) == len(nb) and i not in nb
    # recall@1 of self-queries, like leann.rs:1388-1433 (>= 0.35 there)
    q = v[::30]
    ids, dist, cnt = idx.search_batch(q, 1, 64)
    assert (ids[:, 0] == np.arange(0, n, 30)).mean() >= 0.9


def test_clustered_rows_build_equals_the_oracle(orc):
    """The clustered case round 1 took out of the test above: the reference's
    selection rule keeps the m0 closest candidates without any diversity, so on clustered rows the
    graph falls apart into its clusters and self-query recall collapses -- in the ORACLE's sequential
    build of the reference rule just as on the device.  What is asserted is therefore equality: the
    device builder at batch = 1 returns the oracle's index byte for byte, its searches return the
    oracle's answers, and its (low) recall is the oracle's recall."""
    n, d = 1200, 16
    v = clustered_vectors(n, d, 21)
    cfg = ia.LeannConfig(m=8, m0=16, ef_construction=64)
    levels = np.zeros(n, np.uint64)
    csr = orc.leann_build(v, m=8, m0=16, ef_construction=64, levels=levels)
    idx = ia.LeannIndex.build(v, cfg, levels=levels, batch=1)
    g = ia.CsrGraph(node_offsets=csr.node_offsets, neighbors=csr.neighbors, levels=csr.levels,
                    entry_point=csr.entry_point, max_level=csr.max_level, num_nodes=csr.num_nodes,
                    degree_counts=csr.degree_counts)
    assert idx.to_bytes() == ia.LeannIndex.from_csr(g, cfg, dimension=d).to_bytes()
    q = v[::20]
    ids, dist, cnt = idx.search_batch(q, 1, 64)
    hits_dev = hits_orc = 0
    for i in range(q.shape[0]):
        r = orc.leann_search(csr, v, q[i], 1, 64)
        assert ids[i, :int(cnt[i])].tolist() == r.ids.tolist()
        hits_orc += int(r.ids.size and r.ids[0] == i * 20)
        hits_dev += int(cnt[i] and ids[i, 0] == i * 20)
    assert hits_dev == hits_orc


def test_build_edge_cases():
    e = ia.LeannIndex.build(np.zeros((0, 0), np.float32))
    assert e.is_empty()
    one = ia.LeannIndex.build(np.ones((1, 8), np.float32))
    assert len(one) == 1 and one.entry_point == 0 and one.search(np.ones(8, np.float32), 3)[0][0] == 0
    with pytest.raises(ia.CoreError) as ex:
        ia.LeannIndex.build(np.ones((4, 8), np.float32), ia.LeannConfig(m=64, m0=129, ef_construction=200))
    assert ex.value.kind == "Unsupported"


@pytest.mark.parametrize("m,m0,efc", [(48, 96, 400), (64, 128, 256), (33, 65, 100)])
def test_accurate_preset_build_is_the_reference_graph(orc, m, m0, efc):
    """LeannConfig::accurate() (m = 48, m0 = 96, ef_construction = 400; leann.rs:419-429) and the
    edges of the wide-row range: rows outgrow 64 ids, the distance re-sort of prune_neighbors_temp
    (leann.rs:634-658) runs over up to 129 entries.  Byte-equal to the oracle's sequential build."""
    n, d = 420, 12
    v = uniform_vectors(n, d, 31 + m0)
    cfg = ia.LeannConfig.accurate()
    cfg.m, cfg.m0, cfg.ef_construction = m, m0, efc
    levels = random_levels(n, m, 5)
    want, csr = reference_bytes(orc, v, cfg, levels)
    idx = ia.LeannIndex.build(v, cfg, levels=levels, batch=1)
    assert idx.to_bytes() == want
    assert max(len(idx.get_neighbors(i)) for i in range(n)) > 64  # the case is what it claims to be
    q = uniform_vectors(10, d, 77)
    ids, dist, cnt = idx.search_batch(q, 10, cfg.ef_search)
    assert idx.last_stats()["exact_path"] == 0
    for i in range(10):
        r = orc.leann_search(csr, v, q[i], 10, cfg.ef_search)
        assert ids[i, :cnt[i]].tolist() == r.ids.tolist()
        assert dist[i, :cnt[i]].view(np.uint32).tolist() == r.dist.view(np.uint32).tolist()
