"""Two-level search with a PQ filter (EXTENSION: docs/leann-specification.md:223-275, Algorithm 2;
the reference ships no implementation).  The C oracle's orc_two_level_search is the definition
the device path is tested against; here it is cross-checked against an independent pure-Python
restatement and against properties that hold for any reading of the pseudo-code."""
import numpy as np
import pytest

import oracle as orc
from _data import clustered_vectors, knn_graph, random_csr
from _pyref import two_level_search as py_two_level


def make_pq(vectors, m, K, seed):
    """Codebooks = subvectors of K random rows; codes = nearest centroid (squared L2, numpy).
    Inputs only: no reference value depends on how they were made."""
    rng = np.random.default_rng(seed)
    n, d = vectors.shape
    dsub = d // m
    pick = rng.choice(n, size=K, replace=False)
    cb = np.stack([vectors[pick, j * dsub:(j + 1) * dsub] for j in range(m)]).astype(np.float32)
    codes = np.zeros((n, m), dtype=np.uint16)
    for j in range(m):
        sub = vectors[:, j * dsub:(j + 1) * dsub]
        dist = ((sub[:, None, :] - cb[j][None, :, :]) ** 2).sum(-1)
        codes[:, j] = dist.argmin(1)
    return cb, codes


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("ratio", [0.1, 0.34, 1.0])
def test_oracle_matches_python_restatement(ratio):
    n, d, m, K = 260, 16, 4, 16
    v = clustered_vectors(n, d, 11, per_cluster=20)
    off, nb = knn_graph(v, 10, seed=5, extra_random=3)
    cb, codes = make_pq(v, m, K, 3)
    g = orc.Csr(off, nb, entry_point=7)
    qs = clustered_vectors(6, d, 12, per_cluster=20)
    for q in qs:
        for k, ef in ((5, 12), (10, 40)):
            r = orc.two_level_search(g, v, cb, codes, q, k, ef, ratio)
            assert r.status == orc.OK
            ids, dist, n_exact, n_approx = py_two_level(off, nb, v, cb, codes, q, 7, k, ef, ratio)
            assert r.ids.tolist() == ids
            assert bits(r.dist).tolist() == bits(np.array(dist, dtype=np.float32)).tolist()
            assert r.counters["evals"] == n_exact and r.counters["pushes"] == n_approx


def test_properties():
    n, d, m, K = 3000, 32, 16, 256
    v = clustered_vectors(n, d, 21)
    off, nb = knn_graph(v, 16, seed=2)
    cb, codes = make_pq(v, m, K, 4)
    g = orc.Csr(off, nb, entry_point=0)
    qs = clustered_vectors(20, d, 22)
    hits = {0.2: 0, 1.0: 0}
    evals = {0.2: 0, 1.0: 0}
    for q in qs:
        truth = np.argsort(1.0 - v @ q / np.linalg.norm(q), kind="stable")[:10]
        for ratio in (0.2, 1.0):
            r = orc.two_level_search(g, v, cb, codes, q, 10, 64, ratio)
            assert r.status == orc.OK and r.ids.size == 10
            assert np.all(np.diff(r.dist) >= 0)                      # ascending
            assert len(set(r.ids.tolist())) == 10
            for i, dd in zip(r.ids, r.dist):                         # exact distances
                assert bits([dd])[0] == bits([orc.distance(orc.COSINE, q, v[int(i)])[1]])[0]
            assert r.counters["evals"] <= r.counters["pushes"] + 1   # exact <= approximate + entry
            hits[ratio] += len(set(truth.tolist()) & set(r.ids.tolist()))
            evals[ratio] += r.counters["evals"]
    assert evals[1.0] > 3 * evals[0.2]          # the filter saves exact evaluations (4.2x here) ...
    assert hits[1.0] >= 0.9 * 200               # ... ratio 1 is a plain best-first search
    assert hits[0.2] >= 0.85 * 200              # ... and ratio 0.2 keeps the recall (0.91 both)


def test_errors_and_edges():
    n, d, m, K = 50, 8, 2, 4
    v = clustered_vectors(n, d, 1, per_cluster=10)
    off, nb = random_csr(n, 6, 3)
    cb, codes = make_pq(v, m, K, 5)
    q = v[3]
    g = orc.Csr(off, nb, entry_point=0)
    # ef = max(ef, k); k beyond the reachable set -> fewer results, no error
    r = orc.two_level_search(g, v, cb, codes, q, 60, 4, 1.0)
    assert r.status == orc.OK and 1 <= r.ids.size <= n
    # wrong query dimension
    assert orc.two_level_search(g, v, cb, codes, np.zeros(5, np.float32), 3, 8, 0.5).status == \
        orc.DIMENSION_MISMATCH
    # a neighbour without a code row / a promoted id without an embedding row
    assert orc.two_level_search(g, v, cb, codes[:10], q, 3, 8, 0.5).status == orc.NODE_NOT_FOUND
    assert orc.two_level_search(g, v[:10], cb, codes, q, 3, 8, 1.0).status == orc.NODE_NOT_FOUND
    # empty graph -> empty result; no entry point -> IndexNotBuilt
    e = orc.Csr(np.zeros(1, np.uint64), np.zeros(0, np.uint64), entry_point=None)
    assert orc.two_level_search(e, v, cb, codes, q, 3, 8, 0.5).ids.size == 0
    ne = orc.Csr(off, nb, entry_point=None)
    assert orc.two_level_search(ne, v, cb, codes, q, 3, 8, 0.5).status == orc.INDEX_NOT_BUILT
    # ratio <= 0 or NaN promotes one entry per expansion; ratio > 1 behaves like 1
    a = orc.two_level_search(g, v, cb, codes, q, 5, 8, 0.0)
    b = orc.two_level_search(g, v, cb, codes, q, 5, 8, float("nan"))
    assert a.status == orc.OK and a.ids.tolist() == b.ids.tolist()
    c1 = orc.two_level_search(g, v, cb, codes, q, 5, 8, 1.0)
    c2 = orc.two_level_search(g, v, cb, codes, q, 5, 8, 7.5)
    assert c1.ids.tolist() == c2.ids.tolist()
