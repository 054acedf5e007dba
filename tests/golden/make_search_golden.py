"""Generates tests/golden/search_small.npz: a frozen set of search cases (graph, rows, queries,
parameters) with the answers of the CPU oracle (oracle/islands_oracle.c) at the time of
writing.  The reference holds no golden neighbour lists, so these vectors
pin the oracle against later drift and give the HIP path a second, immutable target; they are as
authoritative as the oracle's restatement of leann.rs:560-988 / hnsw.rs:214-504.
    python tests/golden/make_search_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import oracle as orc  # noqa: E402
from _data import clustered_vectors, random_levels, uniform_vectors  # noqa: E402


def main():
    orc.build()
    out = {}
    n, d = 400, 24
    rows = clustered_vectors(n, d, 101)
    rows[350:] = rows[:50]  # duplicated rows: equal distances, tie order matters
    q = np.concatenate([uniform_vectors(10, d, 102), rows[[3, 77, 351]]]).astype(np.float32)
    levels = random_levels(n, 8, 103)
    out["rows"], out["queries"], out["levels"] = rows, q, levels
    for metric in range(4):
        csr = orc.leann_build(rows, m=8, m0=16, ef_construction=40, metric=metric, levels=levels)
        out[f"m{metric}_offsets"], out[f"m{metric}_neighbors"] = csr.node_offsets, csr.neighbors
        out[f"m{metric}_entry"], out[f"m{metric}_max_level"] = np.uint64(csr.entry_point), np.uint64(csr.max_level)
        for (k, ef, ratio, strat) in ((10, 32, 0.0, 0), (5, 5, 0.0, 0), (10, 64, 0.5, 1)):
            ids = np.full((q.shape[0], k), np.iinfo(np.uint64).max, np.uint64)
            dist = np.zeros((q.shape[0], k), np.float32)
            cnt = np.zeros(q.shape[0], np.uint32)
            for i in range(q.shape[0]):
                r = orc.leann_search(csr, rows, q[i], k, ef, metric=metric, prune_ratio=ratio, strategy=strat)
                assert r.status == 0
                c = r.ids.size
                ids[i, :c], dist[i, :c], cnt[i] = r.ids, r.dist, c
            tag = f"m{metric}_k{k}_ef{ef}_p{int(ratio * 10)}{strat}"
            out[tag + "_ids"], out[tag + "_dist"], out[tag + "_cnt"] = ids, dist, cnt
    # HnswGraph facade: graph by the oracle's insert(), cosine
    h = orc.Hnsw(m=8, m0=16, ef_construction=40, metric=orc.COSINE)
    for i in range(n):
        st, _ = h.insert(rows[i], int(levels[i]))
        assert st == 0
    nl = h.max_level + 1
    out["hnsw_layers"] = np.uint64(nl)
    for L in range(nl):
        lens = np.array([len(h.neighbors(i, L) or []) for i in range(n)], np.uint64)
        flat = np.array([x for i in range(n) for x in (h.neighbors(i, L) or [])], np.uint64)
        out[f"hnsw_l{L}_lens"], out[f"hnsw_l{L}_flat"] = lens, flat
    out["hnsw_levels"] = np.array([h.level(i) for i in range(n)], np.uint64)
    out["hnsw_entry"] = np.uint64(h.entry_point)
    ids = np.zeros((q.shape[0], 10), np.uint64)
    dist = np.zeros((q.shape[0], 10), np.float32)
    for i in range(q.shape[0]):
        r = h.search(q[i], 10, 50)
        ids[i], dist[i] = r.ids, r.dist
    out["hnsw_ids"], out["hnsw_dist"] = ids, dist
    path = os.path.join(ROOT, "tests", "golden", "search_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
