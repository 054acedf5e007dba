"""Seeded synthetic inputs shared by the tests (numpy only, no reference code)."""
import numpy as np


def uniform_vectors(n, d, seed):
    """Reference-style test data: i.i.d. uniform [-1, 1) (leann.rs:1078-1083 shape;
    the Rust StdRng stream itself is not reproduced -- no reference value depends on it)."""
    rng = np.random.default_rng(seed)
    return (rng.random((n, d), dtype=np.float32) * 2.0 - 1.0).astype(np.float32)


def clustered_vectors(n, d, seed, per_cluster=50, noise=0.3):
    rng = np.random.default_rng(seed)
    nc = max(1, n // per_cluster)
    centres = rng.standard_normal((nc, d)).astype(np.float32)
    assign = rng.integers(0, nc, size=n)
    x = centres[assign] + noise * rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(np.float32)


def random_levels(n, m, seed, max_layers=16):
    """Stand-in for random_level() (leann.rs:549-554): floor(-ln(r) * 1/ln(M))."""
    rng = np.random.default_rng(seed)
    r = rng.random(n)
    lv = np.floor(-np.log(np.maximum(r, 1e-300)) * (1.0 / np.log(m))).astype(np.uint64)
    return np.minimum(lv, max_layers - 1)


def random_csr(n, deg, seed, dup=False):
    """Random regular-ish digraph in CSR form (not a good ANN graph; exercises traversal)."""
    rng = np.random.default_rng(seed)
    degs = rng.integers(max(1, deg // 2), deg + 1, size=n)
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum(degs)
    nb = np.empty(int(off[-1]), dtype=np.uint64)
    for i in range(n):
        k = int(degs[i])
        if dup:
            nb[int(off[i]):int(off[i + 1])] = rng.integers(0, n, size=k)
        else:
            nb[int(off[i]):int(off[i + 1])] = rng.choice(n, size=min(k, n), replace=False)[:k]
    return off, nb


def knn_graph(x, deg, seed=0, extra_random=4):
    """Brute-force cosine kNN digraph + a few random long-range edges, as CSR (u64).
    Test infrastructure only: gives the traversal a realistic graph at a few 10k nodes."""
    n = x.shape[0]
    rng = np.random.default_rng(seed)
    xn = x / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-30)
    k = min(deg - extra_random, n - 1)
    nbrs = np.empty((n, k), dtype=np.int64)
    step = 2048
    for s in range(0, n, step):
        sim = xn[s:s + step] @ xn.T
        sim[np.arange(sim.shape[0]), np.arange(s, min(s + step, n))] = -np.inf
        part = np.argpartition(-sim, k - 1, axis=1)[:, :k]
        order = np.argsort(-np.take_along_axis(sim, part, 1), axis=1)
        nbrs[s:s + step] = np.take_along_axis(part, order, 1)
    rows = []
    for i in range(n):
        row = list(dict.fromkeys(nbrs[i].tolist()))
        for v in rng.integers(0, n, size=extra_random):
            if int(v) != i and int(v) not in row:
                row.append(int(v))
        rows.append(row)
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in rows])
    nb = np.fromiter((v for r in rows for v in r), dtype=np.uint64, count=int(off[-1]))
    return off, nb
