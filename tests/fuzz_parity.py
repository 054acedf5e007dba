"""Randomised differential test of the search path: the HIP kernels (through the C ABI) against
the CPU oracle on random graphs, vectors (optionally quantised so that equal distances are
common), metrics, ef / k, pruning settings.  Ids, distance bits, result counts and the work
counters must all match.  Runs for `--seconds` on the GPU box:

    python tests/fuzz_parity.py --seconds 120 [--seed 0]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]  # run as a script

import numpy as np

import islands_amd as ia
import oracle as orc
from _data import clustered_vectors, knn_graph, random_csr, uniform_vectors


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def one_case(rng, case):
    n = int(rng.choice([30, 200, 900, 2500]))
    d = int(rng.choice([3, 8, 17, 32, 100, 768]))
    if d == 768:
        n = min(n, 900)
    seed = int(rng.integers(1 << 30))
    vec = clustered_vectors(n, d, seed) if rng.random() < 0.5 else uniform_vectors(n, d, seed)
    quant = rng.random()
    if quant < 0.3:  # coarse values -> many exactly equal distances
        vec = np.round(vec * 2) / 2
        vec[np.abs(vec).sum(1) == 0, 0] = 1.0
    elif quant < 0.4:  # duplicated rows
        vec[n // 2:] = vec[:n - n // 2]
    vec = vec.astype(np.float32)
    deg = int(rng.choice([4, 12, 30, 60, 64, 65, 96, 128, 140]))  # rows past 64 ids: two ids per lane; past 128: heap-exact kernel
    if rng.random() < 0.5 and n >= 64:
        off, nb = knn_graph(vec, min(deg, n - 1), seed)
    else:
        off, nb = random_csr(n, min(deg, n - 1), seed, dup=rng.random() < 0.2)
    metric = int(rng.integers(0, 4))
    ef = int(rng.choice([1, 2, 7, 33, 64, 100, 128, 200, 300]))
    k = int(rng.choice([1, 3, 10, 50]))
    ratio = float(rng.choice([0.0, 0.0, 0.3, 0.8]))
    strategy = int(rng.integers(0, 2))
    entry = int(rng.integers(0, n))
    nq = 24
    q = (vec[rng.integers(0, n, nq)] + (rng.random((nq, d), dtype=np.float32) - 0.5) *
         np.float32(rng.choice([0.0, 0.05, 0.5]))).astype(np.float32)
    levels = np.zeros(n, np.uint64)
    degs = (off[1:] - off[:-1]).astype(np.uint64)
    csr = orc.Csr(node_offsets=off, neighbors=nb, levels=levels, entry_point=entry, max_level=0,
                  degree_counts=degs)
    cfg = ia.LeannConfig(metric=ia.DistanceMetric(metric), prune_ratio=ratio,
                         pruning_strategy=ia.PruningStrategy(strategy))
    g = ia.CsrGraph(node_offsets=off, neighbors=nb, levels=levels, entry_point=entry, max_level=0,
                    num_nodes=n, degree_counts=degs)
    idx = ia.LeannIndex.from_csr(g, cfg, dimension=d)
    idx.upload(0)
    # a quarter of the cases store the rows as bf16 (the provider's vectors are then the exact f32
    # images of the rounded values); of their queries some, all or none are bf16-valued, which picks
    # between the two instantiations of the traversal kernel inside one call
    as_bf16 = rng.random() < 0.25
    if as_bf16:
        b16 = (vec.view(np.uint32) >> 16).astype(np.uint16)
        vec = (b16.astype(np.uint32) << 16).view(np.float32)
        idx.set_embeddings_bf16(b16)
        share = float(rng.choice([0.0, 0.5, 1.0]))
        pick = rng.random(nq) < share
        q = q.copy()
        q[pick] = ((q[pick].view(np.uint32) >> 16) << 16).view(np.float32)
    else:
        idx.set_embeddings(vec)
    ids, dist, cnt = idx.search_batch(q, k, ef)
    st = idx.last_stats()
    tot = {"expansions": 0, "edges": 0, "evals": 0, "pushes": 0}
    desc = (f"case {case}: n={n} d={d} deg={deg} metric={metric} ef={ef} k={k} prune={ratio}/{strategy} "
            f"quant={quant:.2f} bf16={as_bf16}")
    for i in range(nq):
        r = orc.leann_search(csr, vec, q[i], k, ef, metric=metric, prune_ratio=ratio, strategy=strategy)
        assert r.status == 0, desc
        c = int(cnt[i])
        assert c == r.ids.size, (desc, i, c, r.ids.size)
        assert ids[i, :c].tolist() == r.ids.tolist(), (desc, i, ids[i, :c], r.ids)
        assert bits(dist[i, :c]).tolist() == bits(r.dist).tolist(), (desc, i)
        for f in tot:
            tot[f] += r.counters[f]
    # the device removes repeated ids inside a row on upload: edges may differ then, the rest not
    for f in ("expansions", "evals", "pushes"):
        assert st[f] == tot[f], (desc, f, st[f], tot[f])
    return st


def hnsw_case(rng, case):
    """HnswGraph::search facade: graph built by the oracle's restatement of insert()."""
    from _data import random_levels
    n = int(rng.choice([1, 40, 300, 800]))
    d = int(rng.choice([4, 16, 48]))
    seed = int(rng.integers(1 << 30))
    v = uniform_vectors(n, d, seed)
    if rng.random() < 0.3:
        v = (np.round(v * 2) / 2).astype(np.float32)
        v[np.abs(v).sum(1) == 0, 0] = 1.0
    m = int(rng.choice([4, 8, 16]))
    metric = int(rng.integers(0, 4))
    h = orc.Hnsw(m=m, m0=2 * m, ef_construction=int(rng.choice([20, 60])), metric=metric)
    lv = random_levels(n, m, seed + 1)
    for i in range(n):
        st, _ = h.insert(v[i], int(lv[i]))
        assert st == 0
    layers = [[(h.neighbors(i, L) or []) for i in range(n)] for L in range(h.max_level + 1)]
    g = ia.HnswGraph(v, layers, [h.level(i) for i in range(n)], h.entry_point, h.max_level, m=m, m0=2 * m,
                     ef_construction=20, metric=ia.DistanceMetric(metric))
    k, ef = int(rng.choice([1, 5, 20])), int(rng.choice([1, 10, 64, 150]))
    q = (v[rng.integers(0, n, 16)] + (rng.random((16, d), dtype=np.float32) - 0.5) *
         np.float32(rng.choice([0.0, 0.3]))).astype(np.float32)
    got = g.search_batch(q, k, ef)
    for i in range(16):
        r = h.search(q[i], k, ef)
        assert got[i][0].tolist() == r.ids.tolist(), (case, n, d, m, metric, k, ef, i)
        assert bits(got[i][1]).tolist() == bits(r.dist).tolist(), (case, i)
    return g.last_stats()


def build_case(rng, case):
    """isl_index_build with batch = 1 against the oracle's restatement of LeannIndex::build."""
    from _data import random_levels
    n = int(rng.choice([2, 60, 400, 900]))
    d = int(rng.choice([4, 24, 64]))
    seed = int(rng.integers(1 << 30))
    v = uniform_vectors(n, d, seed) if rng.random() < 0.6 else clustered_vectors(n, d, seed)
    if rng.random() < 0.25:
        v = (np.round(v * 2) / 2).astype(np.float32)
        v[np.abs(v).sum(1) == 0, 0] = 1.0
    m0 = int(rng.choice([4, 12, 32, 60, 64, 96, 128]))  # rows past 64 ids included (accurate(): 96)
    cfg = ia.LeannConfig(m=max(2, m0 // 2), m0=m0, ef_construction=int(rng.choice([m0, 2 * m0, max(128, m0)])),
                         metric=ia.DistanceMetric(int(rng.integers(0, 4))),
                         hub_percentile=float(rng.choice([0.02, 0.1, 0.5])),
                         high_degree_pruning=bool(rng.random() < 0.8))
    levels = random_levels(n, max(2, cfg.m), seed + 5) if rng.random() < 0.5 else None
    csr = orc.leann_build(v, m=cfg.m, m0=cfg.m0, ef_construction=cfg.ef_construction, metric=int(cfg.metric),
                          high_degree_pruning=cfg.high_degree_pruning, hub_percentile=cfg.hub_percentile,
                          levels=levels)
    g = ia.CsrGraph(node_offsets=csr.node_offsets, neighbors=csr.neighbors, levels=csr.levels,
                    entry_point=csr.entry_point, max_level=csr.max_level, num_nodes=csr.num_nodes,
                    degree_counts=csr.degree_counts)
    want = ia.LeannIndex.from_csr(g, cfg, dimension=d).to_bytes()
    got = ia.LeannIndex.build(v, cfg, levels=levels, batch=1).to_bytes()
    assert got == want, (case, n, d, m0, cfg)
    return {"exact_path": 0, "replayed": 0}


def two_level_case(rng, case):
    """Two-level search (extension): random graphs, vectors, PQ shapes, ratios -- window sizes,
    merge batch sizes and ties all over the place -- against orc_two_level_search."""
    from test_two_level_cpu import make_pq
    n = int(rng.choice([40, 300, 1200, 3000]))
    m, dsub = int(rng.choice([1, 2, 4, 8, 16])), int(rng.choice([1, 3, 8]))
    d = m * dsub
    seed = int(rng.integers(1 << 30))
    vec = clustered_vectors(n, d, seed) if rng.random() < 0.5 else uniform_vectors(n, d, seed)
    quant = rng.random()
    if quant < 0.3:
        vec = np.round(vec * 2) / 2
        vec[np.abs(vec).sum(1) == 0, 0] = 1.0
    elif quant < 0.4:
        vec[n // 2:] = vec[:n - n // 2]
    vec = vec.astype(np.float32)
    deg = int(rng.choice([4, 12, 30, 64, 100]))
    if rng.random() < 0.5 and n >= 128 and deg <= 64:
        off, nb = knn_graph(vec, min(deg, n - 1), seed)
    else:
        off, nb = random_csr(n, min(deg, n - 1), seed, dup=rng.random() < 0.2)
    K = int(rng.choice([2, 7, 16, 64]))
    cb, codes = make_pq(vec, m, min(K, n), seed % 1000)
    metric = int(rng.integers(0, 4))
    ef = int(rng.choice([1, 2, 7, 33, 64, 128, 200, 300]))
    k = int(rng.choice([1, 3, 10, 50]))
    ratio = float(rng.choice([0.01, 0.1, 0.25, 0.5, 1.0, 0.0]))
    entry = int(rng.integers(0, n))
    nq = 16
    q = (vec[rng.integers(0, n, nq)] + (rng.random((nq, d), dtype=np.float32) - 0.5) *
         np.float32(rng.choice([0.0, 0.05, 0.5]))).astype(np.float32)
    levels = np.zeros(n, np.uint64)
    degs = (off[1:] - off[:-1]).astype(np.uint64)
    csr = orc.Csr(node_offsets=off, neighbors=nb, levels=levels, entry_point=entry, max_level=0,
                  degree_counts=degs)
    g = ia.CsrGraph(node_offsets=off, neighbors=nb, levels=levels, entry_point=entry, max_level=0,
                    num_nodes=n, degree_counts=degs)
    idx = ia.LeannIndex.from_csr(g, ia.LeannConfig(metric=ia.DistanceMetric(metric)), dimension=d)
    idx.upload(0)
    idx.set_embeddings(vec)
    pq = ia.ProductQuantizer(d, cb)
    idx.set_pq_codes(pq, codes)
    ids, dist, cnt = idx.search_two_level_batch(q, k, ef, ratio)
    st = idx.last_stats()
    tot = {"expansions": 0, "evals": 0, "pushes": 0}
    desc = f"case {case}: n={n} d={d} m={m} K={K} deg={deg} metric={metric} ef={ef} k={k} a={ratio} quant={quant:.2f}"
    for i in range(nq):
        r = orc.two_level_search(csr, vec, cb, codes, q[i], k, ef, ratio, metric=metric)
        assert r.status == 0, desc
        c = int(cnt[i])
        assert c == r.ids.size, (desc, i, c, r.ids.size)
        assert ids[i, :c].tolist() == r.ids.tolist(), (desc, i, ids[i, :c], r.ids)
        assert bits(dist[i, :c]).tolist() == bits(r.dist).tolist(), (desc, i)
        for f in tot:
            tot[f] += r.counters[f]
    for f in tot:
        assert st[f] == tot[f], (desc, f, st[f], tot[f])
    del pq
    return {"exact_path": 0, "replayed": 0}


_RC = {}


def recompute_case(rng, case):
    """Recompute provider (leann.rs:82-99: embeddings computed on the fly) against the in-memory
    provider holding the same embeddings: ids, distance bits, counts and work counters must be equal
    whatever the row cache holds -- random cache sizes (down to the 256-row floor, so that rows are
    evicted and re-encoded inside a call and the batch runs a few queries at a time), rows past 64
    ids, quantised token rows (equal embeddings -> ties -> heap-exact kernel, which re-runs blocked
    queries from their start), keep_rows on and off, every metric and result-set size."""
    import bert_ref
    if "enc" not in _RC:
        cfg = dict(vocab_size=200, hidden=32, layers=1, heads=2, intermediate=64, max_position=16, type_vocab=2)
        w = bert_ref.random_weights(cfg, seed=45, std=0.3)
        _RC["cfg"] = cfg
        _RC["enc"] = ia.CandleEmbedder(ia.BertConfig(**{k: cfg[k] for k in cfg}), w, normalize=True)
    enc = _RC["enc"]
    n = int(rng.choice([300, 800, 1500]))
    L = int(rng.choice([6, 12]))
    seed = int(rng.integers(1 << 30))
    r2 = np.random.default_rng(seed)
    ties = rng.random() < 0.2
    tok = r2.integers(1, 200, (n, L)).astype(np.uint16)
    topics = r2.integers(1, 200, (12, L // 2))
    tok[:, :L // 2] = topics[r2.integers(0, 12, n)]
    lens = None
    if ties:  # a quarter of the rows are copies: equal embeddings, equal distances
        tok[n // 2: n // 2 + n // 4] = tok[:n // 4]
    elif rng.random() < 0.5:
        lens = r2.integers(L // 2 + 1, L + 1, n).astype(np.uint16)
    mask = None if lens is None else (np.arange(L)[None, :] < lens[:, None]).astype(np.float32)
    emb = np.concatenate([enc.embed(tok[o:o + 512].astype(np.int64), None, None if mask is None else mask[o:o + 512])
                          for o in range(0, n, 512)])
    deg = int(rng.choice([8, 30, 64, 100]))
    off, nb = random_csr(n, min(deg, n - 1), seed)
    metric = int(rng.integers(0, 4))
    ef = int(rng.choice([4, 33, 64, 128, 200, 300, 600]))  # 600: past the wave-per-query kernel, heap-exact only
    k = int(rng.choice([1, 5, 10]))
    cfg = ia.LeannConfig(m=max(2, deg // 2), m0=max(deg, 4), ef_construction=max(deg, 128) if deg <= 128 else 200,
                         metric=ia.DistanceMetric(metric))
    entry = int(rng.integers(0, n))
    g = ia.CsrGraph(node_offsets=off, neighbors=nb, levels=np.zeros(n, np.uint64), entry_point=entry, num_nodes=n,
                    degree_counts=(off[1:] - off[:-1]).astype(np.uint64))
    nq = int(rng.choice([1, 7, 40]))
    q = (emb[r2.integers(0, n, nq)] + r2.standard_normal((nq, emb.shape[1])).astype(np.float32) * np.float32(0.05))
    mem = ia.LeannIndex.from_csr(g, cfg, dimension=emb.shape[1]).upload(0)
    mem.set_embeddings(emb)
    want = mem.search_batch(q, k, ef)
    ws = mem.last_stats()
    # (equal embeddings send queries to the heap-exact kernel: since round 3 it parks and resumes over a
    # bounded row cache like the others -- before, it needed its whole traversal resident)
    rows = int(rng.choice([256, max(256, n // 3), n]))
    keep = bool(rng.random() < 0.3)
    # round 4: the rounds' encoder batches cut to whole waves of GEMM tiles (left-over misses reported again) and the
    # encoder's passes as two halves side by side -- any quantum, any split threshold, the same answers
    quantum = rng.choice(["", "0", "8", "37", "100"])
    split = rng.choice(["", "0", "8", "40"])
    ahead = rng.choice(["", "0", "3", "8"])  # two-level search: nodes a parked query names beyond its misses
    for var, val in (("ISL_RECOMPUTE_QUANTUM", quantum), ("ISL_ENCODER_SPLIT", split), ("ISL_TL_PREFETCH", ahead)):
        if val == "":
            os.environ.pop(var, None)
        else:
            os.environ[var] = str(val)
    rec = ia.LeannIndex.from_csr(g, cfg, dimension=emb.shape[1]).upload(0)
    rec.set_recompute_provider(enc, tok, lens, keep_rows=keep, cache_rows=rows)
    desc = (f"case {case}: n={n} L={L} deg={deg} metric={metric} ef={ef} k={k} nq={nq} rows={rows} keep={keep} ties={ties} "
            f"quantum={quantum!r} split={split!r} ahead={ahead!r}")
    for rep in range(2):
        got = rec.search_batch(q, k, ef)
        st = rec.last_stats()
        assert got[2].tolist() == want[2].tolist(), desc
        assert got[0].tolist() == want[0].tolist(), desc
        assert bits(got[1]).tolist() == bits(want[1]).tolist(), desc
        for f in ("expansions", "edges", "evals", "pushes"):
            assert st[f] == ws[f], (desc, f, st[f], ws[f])
    # the two-level search over the same pair of providers (round 3: it parks and resumes too, any
    # cache size; a ratio of 1 with a small ef takes the window retry inside the recompute rounds)
    if rng.random() < 0.6 and ef <= 512:
        from test_two_level_cpu import make_pq
        m = int(rng.choice([2, 4, 8, 16]))
        K = int(rng.choice([8, 32, 64]))
        cb, codes = make_pq(emb, m, K, seed % 1000)
        pq = ia.ProductQuantizer(emb.shape[1], cb, ia.DistanceMetric.Euclidean)
        mem.set_pq_codes(pq, codes)
        rec.set_pq_codes(pq, codes)
        ratio = float(rng.choice([0.05, 0.2, 0.5, 1.0]))
        rows2 = int(rng.choice([256, max(256, n // 3), n]))
        rec.set_recompute_provider(enc, tok, lens, keep_rows=keep, cache_rows=rows2)
        rec.set_pq_codes(pq, codes)
        want2 = mem.search_two_level_batch(q, k, ef, ratio)
        ws2 = mem.last_stats()
        got2 = rec.search_two_level_batch(q, k, ef, ratio)
        st2 = rec.last_stats()
        d2 = desc + f" | two-level m={m} K={K} ratio={ratio} rows={rows2}"
        assert got2[2].tolist() == want2[2].tolist(), d2
        assert got2[0].tolist() == want2[0].tolist(), d2
        assert bits(got2[1]).tolist() == bits(want2[1]).tolist(), d2
        for f in ("expansions", "edges", "evals", "pushes"):
            assert st2[f] == ws2[f], (d2, f, st2[f], ws2[f])
        del pq
    return {"exact_path": ws["exact_path"], "replayed": ws["replayed"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--mode", choices=["leann", "hnsw", "build", "two_level", "recompute"], default="leann")
    args = ap.parse_args()
    orc.build()
    rng = np.random.default_rng(args.seed)
    t0, case, exact, replay = time.time(), 0, 0, 0
    fn = {"leann": one_case, "hnsw": hnsw_case, "build": build_case, "two_level": two_level_case,
          "recompute": recompute_case}[args.mode]
    while time.time() - t0 < args.seconds:
        st = fn(rng, case)
        exact += st["exact_path"]
        replay += st["replayed"]
        case += 1
        if case % 20 == 0:
            print(f"{case} cases ok ({exact} queries via the exact kernel, {replay} replayed)", flush=True)
    print(f"fuzz {args.mode} ok: {case} cases, {exact} queries via the exact kernel, {replay} replayed")


if __name__ == "__main__":
    main()
