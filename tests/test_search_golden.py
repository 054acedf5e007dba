"""The frozen search cases of tests/golden/search_small.npz (written once by
tests/golden/make_search_golden.py from the CPU oracle): the oracle must keep reproducing them
(CPU), and the HIP path must return them bit for bit (GPU)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "search_small.npz")
CASES = ((10, 32, 0.0, 0), (5, 5, 0.0, 0), (10, 64, 0.5, 1))


def tag(metric, k, ef, ratio, strat):
    return f"m{metric}_k{k}_ef{ef}_p{int(ratio * 10)}{strat}"


def test_oracle_still_reproduces_the_golden_vectors(orc):
    z = np.load(GOLD)
    rows, q, levels = z["rows"], z["queries"], z["levels"]
    for metric in range(4):
        csr = orc.leann_build(rows, m=8, m0=16, ef_construction=40, metric=metric, levels=levels)
        assert np.array_equal(csr.node_offsets, z[f"m{metric}_offsets"])
        assert np.array_equal(csr.neighbors, z[f"m{metric}_neighbors"])
        assert csr.entry_point == int(z[f"m{metric}_entry"])
        for (k, ef, ratio, strat) in CASES:
            t = tag(metric, k, ef, ratio, strat)
            for i in range(q.shape[0]):
                r = orc.leann_search(csr, rows, q[i], k, ef, metric=metric, prune_ratio=ratio, strategy=strat)
                c = int(z[t + "_cnt"][i])
                assert r.ids.tolist() == z[t + "_ids"][i, :c].tolist()
                assert r.dist.view(np.uint32).tolist() == z[t + "_dist"][i, :c].view(np.uint32).tolist()


@pytest.mark.gpu
def test_hip_path_returns_the_golden_vectors():
    import islands_amd as ia
    z = np.load(GOLD)
    rows, q, levels = z["rows"], z["queries"], z["levels"]
    n = rows.shape[0]
    for metric in range(4):
        off, nb = z[f"m{metric}_offsets"], z[f"m{metric}_neighbors"]
        for (k, ef, ratio, strat) in CASES:
            cfg = ia.LeannConfig(m=8, m0=16, ef_construction=40, metric=ia.DistanceMetric(metric),
                                 prune_ratio=ratio, pruning_strategy=ia.PruningStrategy(strat))
            g = ia.CsrGraph(node_offsets=off, neighbors=nb, levels=levels, entry_point=int(z[f"m{metric}_entry"]),
                            max_level=int(z[f"m{metric}_max_level"]), num_nodes=n,
                            degree_counts=(off[1:] - off[:-1]).astype(np.uint64))
            idx = ia.LeannIndex.from_csr(g, cfg, dimension=rows.shape[1])
            idx.upload(0)
            idx.set_embeddings(rows)
            ids, dist, cnt = idx.search_batch(q, k, ef)
            t = tag(metric, k, ef, ratio, strat)
            assert cnt.tolist() == z[t + "_cnt"].tolist()
            for i in range(q.shape[0]):
                c = int(cnt[i])
                assert ids[i, :c].tolist() == z[t + "_ids"][i, :c].tolist(), (t, i)
                assert dist[i, :c].view(np.uint32).tolist() == z[t + "_dist"][i, :c].view(np.uint32).tolist()
        # the builder reproduces the golden graph as well
        cfg = ia.LeannConfig(m=8, m0=16, ef_construction=40, metric=ia.DistanceMetric(metric))
        built = ia.LeannIndex.build(rows, cfg, levels=levels, batch=1)
        assert built.get_neighbors(7).tolist() == nb[int(off[7]):int(off[8])].tolist()
        assert built.entry_point == int(z[f"m{metric}_entry"])
    nl = int(z["hnsw_layers"])
    layers = []
    for L in range(nl):
        lens, flat = z[f"hnsw_l{L}_lens"], z[f"hnsw_l{L}_flat"]
        o = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        layers.append([flat[o[i]:o[i + 1]].tolist() for i in range(n)])
    hg = ia.HnswGraph(rows, layers, z["hnsw_levels"].tolist(), int(z["hnsw_entry"]), nl - 1, m=8, m0=16,
                      ef_construction=40)
    got = hg.search_batch(q, 10, 50)
    for i in range(q.shape[0]):
        assert got[i][0].tolist() == z["hnsw_ids"][i].tolist()
        assert got[i][1].view(np.uint32).tolist() == z["hnsw_dist"][i].view(np.uint32).tolist()


# ---- two-level search (extension): tests/golden/two_level_small.npz, make_two_level_golden.py ----
GOLD2 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "two_level_small.npz")
CASES2 = ((10, 32, 0.1), (5, 5, 0.5), (10, 64, 0.25), (10, 48, 1.0))


def test_oracle_still_reproduces_the_two_level_golden_vectors(orc):
    z, z2 = np.load(GOLD), np.load(GOLD2)
    rows, q = z["rows"], z["queries"]
    for metric in range(4):
        csr = orc.Csr(z[f"m{metric}_offsets"], z[f"m{metric}_neighbors"], entry_point=int(z[f"m{metric}_entry"]))
        for (k, ef, a) in CASES2:
            t = f"m{metric}_k{k}_ef{ef}_a{int(a * 100)}"
            for i in range(q.shape[0]):
                r = orc.two_level_search(csr, rows, z2["codebooks"], z2["codes"], q[i], k, ef, a, metric=metric)
                c = int(z2[t + "_cnt"][i])
                assert r.ids.tolist() == z2[t + "_ids"][i, :c].tolist()
                assert r.dist.view(np.uint32).tolist() == z2[t + "_dist"][i, :c].view(np.uint32).tolist()
                assert [r.counters[f] for f in ("expansions", "edges", "evals", "pushes")] == z2[t + "_ctr"][i].tolist()


@pytest.mark.gpu
def test_hip_path_returns_the_two_level_golden_vectors():
    import islands_amd as ia
    z, z2 = np.load(GOLD), np.load(GOLD2)
    rows, q, levels = z["rows"], z["queries"], z["levels"]
    n = rows.shape[0]
    pq = ia.ProductQuantizer(rows.shape[1], z2["codebooks"])
    for metric in range(4):
        off, nb = z[f"m{metric}_offsets"], z[f"m{metric}_neighbors"]
        g = ia.CsrGraph(node_offsets=off, neighbors=nb, levels=levels, entry_point=int(z[f"m{metric}_entry"]),
                        max_level=int(z[f"m{metric}_max_level"]), num_nodes=n,
                        degree_counts=(off[1:] - off[:-1]).astype(np.uint64))
        idx = ia.LeannIndex.from_csr(g, ia.LeannConfig(metric=ia.DistanceMetric(metric)), dimension=rows.shape[1])
        idx.upload(0)
        idx.set_embeddings(rows)
        idx.set_pq_codes(pq, z2["codes"])
        for (k, ef, a) in CASES2:
            ids, dist, cnt = idx.search_two_level_batch(q, k, ef, a)
            t = f"m{metric}_k{k}_ef{ef}_a{int(a * 100)}"
            assert cnt.tolist() == z2[t + "_cnt"].tolist()
            for i in range(q.shape[0]):
                c = int(cnt[i])
                assert ids[i, :c].tolist() == z2[t + "_ids"][i, :c].tolist(), (t, i)
                assert dist[i, :c].view(np.uint32).tolist() == z2[t + "_dist"][i, :c].view(np.uint32).tolist()
