"""ctypes binding of the CPU oracle (oracle/islands_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package `islands_amd`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libislands_oracle.so")

OK, DIMENSION_MISMATCH, EMPTY_COLLECTION, INVALID_CONFIG, INDEX_NOT_BUILT, NODE_NOT_FOUND = range(6)
PQ_ERROR = 10
PANIC = 99
COSINE, EUCLIDEAN, DOT, MANHATTAN = range(4)
PRUNE_GLOBAL, PRUNE_LOCAL, PRUNE_PROPORTIONAL = range(3)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "islands_oracle.c")
    hdr = os.path.join(_HERE, "islands_oracle.h")
    stale = (not os.path.exists(_SO)) or any(
        os.path.getmtime(p) > os.path.getmtime(_SO) for p in (src, hdr) if os.path.exists(p)
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class _Csr(C.Structure):
    _fields_ = [
        ("num_nodes", C.c_uint64),
        ("node_offsets", C.POINTER(C.c_uint64)),
        ("neighbors", C.POINTER(C.c_uint64)),
        ("degree_counts", C.POINTER(C.c_uint64)),
        ("has_entry", C.c_int),
        ("entry_point", C.c_uint64),
    ]


class _LeannParams(C.Structure):
    _fields_ = [
        ("metric", C.c_int),
        ("prune_ratio", C.c_float),
        ("pruning_strategy", C.c_int),
        ("has_dimension", C.c_int),
        ("dimension", C.c_uint64),
    ]


class _Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("expansions", "edges", "evals", "pushes")]


class _BuildParams(C.Structure):
    _fields_ = [
        ("m", C.c_uint64),
        ("m0", C.c_uint64),
        ("ef_construction", C.c_uint64),
        ("metric", C.c_int),
        ("high_degree_pruning", C.c_int),
        ("hub_percentile", C.c_float),
    ]


class _CsrOwned(C.Structure):
    _fields_ = [
        ("num_nodes", C.c_uint64),
        ("node_offsets", C.POINTER(C.c_uint64)),
        ("neighbors", C.POINTER(C.c_uint64)),
        ("degree_counts", C.POINTER(C.c_uint64)),
        ("levels", C.POINTER(C.c_uint64)),
        ("has_entry", C.c_int),
        ("entry_point", C.c_uint64),
        ("max_level", C.c_uint64),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_to_similarity.restype = C.c_float
        _lib.orc_to_similarity.argtypes = [C.c_float]
        _lib.orc_pq_table_distance.restype = C.c_float
        _lib.orc_hnsw_new.restype = C.c_void_p
        _lib.orc_hnsw_new.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int]
        _lib.orc_hnsw_free.argtypes = [C.c_void_p]
        _lib.orc_hnsw_len.restype = C.c_uint64
        _lib.orc_hnsw_len.argtypes = [C.c_void_p]
        _lib.orc_hnsw_max_level.restype = C.c_uint64
        _lib.orc_hnsw_max_level.argtypes = [C.c_void_p]
        _lib.orc_hnsw_level.restype = C.c_uint64
        _lib.orc_hnsw_level.argtypes = [C.c_void_p, C.c_uint64]
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


# ---------------------------------------------------------------- distance.rs
def distance(metric: int, a, b) -> tuple[int, float]:
    a, b = _f32(a), _f32(b)
    out = C.c_float()
    st = lib().orc_distance(metric, _p(a, C.c_float), C.c_size_t(a.size), _p(b, C.c_float),
                            C.c_size_t(b.size), C.byref(out))
    return st, out.value


def distance_squared(metric: int, a, b) -> tuple[int, float]:
    a, b = _f32(a), _f32(b)
    out = C.c_float()
    st = lib().orc_distance_squared(metric, _p(a, C.c_float), C.c_size_t(a.size),
                                    _p(b, C.c_float), C.c_size_t(b.size), C.byref(out))
    return st, out.value


def batch_distance(metric: int, q, rows) -> np.ndarray:
    q, rows = _f32(q), _f32(rows)
    n, d = rows.shape
    assert q.size == d
    out = np.empty(n, dtype=np.float32)
    lib().orc_batch_distance(metric, _p(q, C.c_float), C.c_size_t(d), _p(rows, C.c_float),
                             C.c_size_t(n), _p(out, C.c_float))
    return out


def normalize(v) -> np.ndarray:
    v = _f32(v).copy()
    lib().orc_normalize(_p(v, C.c_float), C.c_size_t(v.size))
    return v


# ------------------------------------------------------------------ leann.rs
@dataclass
class Csr:
    """Host CSR graph with the reference's field set (leann.rs:193-208)."""
    node_offsets: np.ndarray
    neighbors: np.ndarray
    entry_point: int | None
    levels: np.ndarray | None = None
    degree_counts: np.ndarray | None = None
    max_level: int = 0

    def __post_init__(self):
        self.node_offsets = _u64(self.node_offsets)
        self.neighbors = _u64(self.neighbors)
        n = self.num_nodes
        if self.levels is None:
            self.levels = np.zeros(n, dtype=np.uint64)
        if self.degree_counts is None:
            self.degree_counts = np.diff(self.node_offsets).astype(np.uint64)
        self.levels = _u64(self.levels)
        self.degree_counts = _u64(self.degree_counts)

    @property
    def num_nodes(self) -> int:
        return int(self.node_offsets.size - 1)

    def _c(self) -> _Csr:
        return _Csr(self.num_nodes, _p(self.node_offsets, C.c_uint64),
                    _p(self.neighbors, C.c_uint64), _p(self.degree_counts, C.c_uint64),
                    0 if self.entry_point is None else 1,
                    0 if self.entry_point is None else int(self.entry_point))

    def get_neighbors(self, node: int):
        ptr = C.POINTER(C.c_uint64)()
        ln = C.c_size_t()
        g = self._c()
        if lib().orc_csr_get_neighbors(C.byref(g), C.c_uint64(node), C.byref(ptr), C.byref(ln)):
            return None
        return [int(ptr[i]) for i in range(ln.value)]


@dataclass
class SearchOut:
    status: int
    ids: np.ndarray
    dist: np.ndarray
    counters: dict
    payload: int = 0


def leann_search(g: Csr, vectors, query, k: int, ef: int, metric: int = COSINE,
                 prune_ratio: float = 0.0, strategy: int = PRUNE_GLOBAL,
                 dimension: int | None = -1, copy_per_node: bool = False) -> SearchOut:
    vectors = _f32(vectors)
    query = _f32(query)
    nvec, d = vectors.shape
    if dimension == -1:
        dimension = d
    p = _LeannParams(metric, prune_ratio, strategy, 0 if dimension is None else 1,
                     0 if dimension is None else dimension)
    ids = np.zeros(max(k, 1), dtype=np.uint64)
    dist = np.zeros(max(k, 1), dtype=np.float32)
    cnt = C.c_size_t()
    ctr = _Counters()
    payload = C.c_uint64()
    gc = g._c()
    st = lib().orc_leann_search(C.byref(gc), C.byref(p), _p(vectors, C.c_float), C.c_uint64(nvec),
                                C.c_size_t(d), int(copy_per_node), _p(query, C.c_float),
                                C.c_size_t(query.size), C.c_size_t(k), C.c_size_t(ef),
                                _p(ids, C.c_uint64), _p(dist, C.c_float), C.byref(cnt),
                                C.byref(ctr), C.byref(payload))
    n = cnt.value
    return SearchOut(st, ids[:n].copy(), dist[:n].copy(),
                     {f: int(getattr(ctr, f)) for f, _ in _Counters._fields_}, payload.value)


def leann_search_batch(g: Csr, vectors, queries, k: int, ef: int, **kw):
    """Sequential map over queries, like Searcher::search_batch (search.rs:179-181)."""
    queries = _f32(queries)
    nq = queries.shape[0]
    ids = np.zeros((nq, k), dtype=np.uint64)
    dist = np.zeros((nq, k), dtype=np.float32)
    cnt = np.zeros(nq, dtype=np.uint32)
    tot = {"expansions": 0, "edges": 0, "evals": 0, "pushes": 0}
    for i in range(nq):
        r = leann_search(g, vectors, queries[i], k, ef, **kw)
        assert r.status == OK, r.status
        n = r.ids.size
        ids[i, :n], dist[i, :n], cnt[i] = r.ids, r.dist, n
        for f in tot:
            tot[f] += r.counters[f]
    return ids, dist, cnt, tot


def two_level_search(g: Csr, vectors, codebooks, codes, query, k: int, ef: int,
                     rerank_ratio: float, metric: int = COSINE,
                     dimension: int | None = -1) -> SearchOut:
    """EXTENSION (docs/leann-specification.md:223-275, Algorithm 2): the definition the
    device path is tested against -- see islands_oracle.c.  codebooks: [m][K][dsub],
    codes: [ncodes][m] u16.  counters: evals = exact, pushes = approximate evaluations."""
    vectors = _f32(vectors)
    query = _f32(query)
    cb = _f32(codebooks)
    codes = np.ascontiguousarray(codes, dtype=np.uint16)
    nvec, d = vectors.shape
    m, K, dsub = cb.shape
    if dimension == -1:
        dimension = d
    p = _LeannParams(metric, 0.0, PRUNE_GLOBAL, 0 if dimension is None else 1,
                     0 if dimension is None else dimension)
    ids = np.zeros(max(k, 1), dtype=np.uint64)
    dist = np.zeros(max(k, 1), dtype=np.float32)
    cnt = C.c_size_t()
    ctr = _Counters()
    payload = C.c_uint64()
    gc = g._c()
    fn = lib().orc_two_level_search
    fn.restype = C.c_int
    st = fn(C.byref(gc), C.byref(p), _p(vectors, C.c_float), C.c_uint64(nvec), C.c_size_t(d),
            _p(cb, C.c_float), C.c_size_t(m), C.c_size_t(K), C.c_size_t(dsub),
            _p(codes, C.c_uint16), C.c_uint64(codes.shape[0]), _p(query, C.c_float),
            C.c_size_t(query.size), C.c_size_t(k), C.c_size_t(ef), C.c_float(rerank_ratio),
            _p(ids, C.c_uint64), _p(dist, C.c_float), C.byref(cnt), C.byref(ctr),
            C.byref(payload))
    n = cnt.value
    return SearchOut(st, ids[:n].copy(), dist[:n].copy(),
                     {f: int(getattr(ctr, f)) for f, _ in _Counters._fields_}, payload.value)


def leann_build(vectors, m: int = 30, m0: int = 60, ef_construction: int = 128,
                metric: int = COSINE, high_degree_pruning: bool = True,
                hub_percentile: float = 0.02, levels=None) -> Csr:
    vectors = _f32(vectors)
    n, d = vectors.shape
    bp = _BuildParams(m, m0, ef_construction, metric, int(high_degree_pruning), hub_percentile)
    lv = _u64(levels if levels is not None else np.zeros(n))
    out = _CsrOwned()
    st = lib().orc_leann_build(_p(vectors, C.c_float), C.c_uint64(n), C.c_size_t(d), C.byref(bp),
                               _p(lv, C.c_uint64), C.byref(out))
    assert st == OK
    nn = int(out.num_nodes)
    if nn == 0:
        return Csr(np.zeros(1), np.zeros(0), None)
    off = np.ctypeslib.as_array(out.node_offsets, (nn + 1,)).copy()
    nb = np.ctypeslib.as_array(out.neighbors, (max(int(off[-1]), 1),)).copy()[: int(off[-1])]
    dg = np.ctypeslib.as_array(out.degree_counts, (nn,)).copy()
    lvl = np.ctypeslib.as_array(out.levels, (nn,)).copy()
    g = Csr(off, nb, int(out.entry_point) if out.has_entry else None, lvl, dg,
            int(out.max_level))
    lib().orc_csr_free(C.byref(out))
    return g


# ------------------------------------------------------------------- hnsw.rs
class Hnsw:
    def __init__(self, m=16, m0=32, ef_construction=200, metric=COSINE):
        self._h = C.c_void_p(lib().orc_hnsw_new(m, m0, ef_construction, metric))
        self.m, self.m0, self.ef_construction, self.metric = m, m0, ef_construction, metric
        self.dim = None

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_hnsw_free(self._h)
            self._h = None

    def insert(self, v, level: int) -> tuple[int, int]:
        v = _f32(v)
        out = C.c_uint64()
        st = lib().orc_hnsw_insert(self._h, _p(v, C.c_float), C.c_size_t(v.size),
                                   C.c_uint64(level), C.byref(out))
        if st == OK and self.dim is None:
            self.dim = v.size
        return st, out.value

    def __len__(self):
        return int(lib().orc_hnsw_len(self._h))

    @property
    def max_level(self):
        return int(lib().orc_hnsw_max_level(self._h))

    @property
    def entry_point(self):
        e = C.c_uint64()
        return None if lib().orc_hnsw_entry(self._h, C.byref(e)) else int(e.value)

    def level(self, node):
        return int(lib().orc_hnsw_level(self._h, C.c_uint64(node)))

    def neighbors(self, node, layer):
        ptr = C.POINTER(C.c_uint64)()
        ln = C.c_size_t()
        if lib().orc_hnsw_neighbors(self._h, C.c_uint64(node), C.c_uint64(layer), C.byref(ptr),
                                    C.byref(ln)):
            return None
        return [int(ptr[i]) for i in range(ln.value)]

    def search(self, q, k, ef) -> SearchOut:
        q = _f32(q)
        ids = np.zeros(max(k, 1), dtype=np.uint64)
        dist = np.zeros(max(k, 1), dtype=np.float32)
        cnt = C.c_size_t()
        ctr = _Counters()
        st = lib().orc_hnsw_search(self._h, _p(q, C.c_float), C.c_size_t(q.size), C.c_size_t(k),
                                   C.c_size_t(ef), _p(ids, C.c_uint64), _p(dist, C.c_float),
                                   C.byref(cnt), C.byref(ctr))
        n = cnt.value
        return SearchOut(st, ids[:n].copy(), dist[:n].copy(),
                         {f: int(getattr(ctr, f)) for f, _ in _Counters._fields_})


# ------------------------------------------------------------------ search.rs
def to_similarity(score: float) -> float:
    return lib().orc_to_similarity(C.c_float(score))


def _merge(fn, lists_ids, lists_vals, top_k):
    n = len(lists_ids)
    ids = [_u64(x) for x in lists_ids]
    vals = [_f32(x) for x in lists_vals]
    pid = (C.POINTER(C.c_uint64) * n)(*[_p(x, C.c_uint64) for x in ids])
    pv = (C.POINTER(C.c_float) * n)(*[_p(x, C.c_float) for x in vals])
    lens = (C.c_size_t * n)(*[x.size for x in ids])
    oi = np.zeros(max(top_k, 1), dtype=np.uint64)
    ov = np.zeros(max(top_k, 1), dtype=np.float32)
    osrc = np.zeros(max(top_k, 1), dtype=np.uint32)
    cnt = C.c_size_t()
    st = fn(C.c_size_t(n), pid, pv, lens, C.c_size_t(top_k), _p(oi, C.c_uint64),
            _p(ov, C.c_float), _p(osrc, C.c_uint32), C.byref(cnt))
    m = cnt.value
    return st, oi[:m].copy(), ov[:m].copy(), osrc[:m].copy()


def multi_index_merge(lists_ids, lists_scores, top_k):
    return _merge(lib().orc_multi_index_merge, lists_ids, lists_scores, top_k)


def service_merge(lists_ids, lists_dist, top_k):
    return _merge(lib().orc_service_merge, lists_ids, lists_dist, top_k)


# ---------------------------------------------------------------------- pq.rs
def pq_find_nearest(metric, centroids, sub):
    centroids, sub = _f32(centroids), _f32(sub)
    K, dsub = centroids.shape
    out = C.c_uint64()
    st = lib().orc_pq_find_nearest(metric, _p(centroids, C.c_float), C.c_size_t(K),
                                   C.c_size_t(dsub), _p(sub, C.c_float), C.c_size_t(sub.size),
                                   C.byref(out))
    return st, out.value


def pq_encode(metric, codebooks, v):
    cb, v = _f32(codebooks), _f32(v)
    m, K, dsub = cb.shape
    codes = np.zeros(m, dtype=np.uint16)
    st = lib().orc_pq_encode(metric, _p(cb, C.c_float), C.c_size_t(m), C.c_size_t(K),
                             C.c_size_t(dsub), _p(v, C.c_float), C.c_size_t(v.size),
                             _p(codes, C.c_uint16))
    return st, codes


def pq_decode(codebooks, codes):
    cb = _f32(codebooks)
    codes = np.ascontiguousarray(codes, dtype=np.uint16)
    m, K, dsub = cb.shape
    out = np.zeros(m * dsub, dtype=np.float32)
    st = lib().orc_pq_decode(_p(cb, C.c_float), C.c_size_t(m), C.c_size_t(K), C.c_size_t(dsub),
                             _p(codes, C.c_uint16), C.c_size_t(codes.size), _p(out, C.c_float))
    return st, out


def pq_asymmetric_distance(codebooks, q, codes):
    cb, q = _f32(codebooks), _f32(q)
    codes = np.ascontiguousarray(codes, dtype=np.uint16)
    m, K, dsub = cb.shape
    out = C.c_float()
    st = lib().orc_pq_asymmetric_distance(_p(cb, C.c_float), C.c_size_t(m), C.c_size_t(K),
                                          C.c_size_t(dsub), _p(q, C.c_float), C.c_size_t(q.size),
                                          _p(codes, C.c_uint16), C.c_size_t(codes.size),
                                          C.byref(out))
    return st, out.value


def pq_build_tables(codebooks, q):
    cb, q = _f32(codebooks), _f32(q)
    m, K, dsub = cb.shape
    t = np.zeros((m, K), dtype=np.float32)
    st = lib().orc_pq_build_tables(_p(cb, C.c_float), C.c_size_t(m), C.c_size_t(K),
                                   C.c_size_t(dsub), _p(q, C.c_float), C.c_size_t(q.size),
                                   _p(t, C.c_float))
    return st, t


def pq_table_distance(tables, codes) -> float:
    t = _f32(tables)
    codes = np.ascontiguousarray(codes, dtype=np.uint16)
    m, K = t.shape
    return lib().orc_pq_table_distance(_p(t, C.c_float), C.c_size_t(m), C.c_size_t(K),
                                       _p(codes, C.c_uint16))


def mean_pool_normalize(hidden, mask, normalize=True):
    h, mk = _f32(hidden), _f32(mask)
    B, L, H = h.shape
    out = np.zeros((B, H), dtype=np.float32)
    lib().orc_mean_pool_normalize(_p(h, C.c_float), _p(mk, C.c_float), C.c_size_t(B),
                                  C.c_size_t(L), C.c_size_t(H), int(normalize), _p(out, C.c_float))
    return out
