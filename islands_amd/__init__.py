"""islands_amd -- host-side mirror of `islands::core` for the MI355X LEANN search path.

Same names and argument meaning as the reference's re-exports
(src/core/mod.rs:60-99): DistanceMetric, LeannConfig, LeannIndex, CsrGraph,
InMemoryEmbeddingProvider, PruningStrategy, ProductQuantizer, CoreError.
Every compute call goes through the C ABI of libislands_amd.so (HIP kernels
for gfx950); there is no CPU implementation in this package.
"""
from __future__ import annotations

import ctypes as C
import enum
import os
from dataclasses import dataclass, field

import numpy as np

from . import _ffi
from ._ffi import BertConfigC, IndexMetadataC, LeannConfigC, SearchStatsC

__all__ = [
    "CoreError", "DistanceMetric", "PruningStrategy", "LeannConfig", "CsrGraph",
    "InMemoryEmbeddingProvider", "LeannIndex", "ProductQuantizer", "SearchResult",
    "batch_calculate", "calculate", "calculate_squared", "normalize_rows", "distance_matrix",
    "bruteforce_topk", "merge_topk", "merge_service",
    "device_count", "HnswGraph", "SearchConfig", "Searcher", "MultiIndexSearcher",
    "mean_pool_normalize", "service_search", "BertConfig", "CandleEmbedder", "IndexMetadata",
]

MEM_HOST, MEM_DEVICE = 0, 1


class CoreError(Exception):
    """CoreError, src/core/error.rs:9-62.  `kind` is the variant name."""

    def __init__(self, status: int, message: str, expected=0, actual=0, node=0):
        super().__init__(message)
        self.status = status
        self.kind = _ffi.lib().isl_status_name(status).decode()
        self.expected, self.actual, self.node = expected, actual, node


def _check(status: int):
    if status != 0:
        l = _ffi.lib()
        raise CoreError(status, l.isl_last_error_message().decode(), l.isl_last_error_expected(),
                        l.isl_last_error_actual(), l.isl_last_error_node())


def device_count() -> int:
    return int(_ffi.lib().isl_device_count())


class DistanceMetric(enum.IntEnum):
    """src/core/distance.rs:9-19"""
    Cosine = 0
    Euclidean = 1
    DotProduct = 2
    Manhattan = 3


class PruningStrategy(enum.IntEnum):
    """src/core/leann.rs:168-178"""
    Global = 0
    Local = 1
    Proportional = 2


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


# --------------------------------------------------------------- distance.rs
def calculate(metric: DistanceMetric, a, b) -> float:
    """Distance::calculate, distance.rs:38-52."""
    a, b = _f32(a).ravel(), _f32(b).ravel()
    out = C.c_float()
    _check(_ffi.lib().isl_distance(int(metric), _ptr(a), a.size, _ptr(b), b.size, C.byref(out)))
    return out.value


def calculate_squared(metric: DistanceMetric, a, b) -> float:
    """Distance::calculate_squared, distance.rs:54-66."""
    a, b = _f32(a).ravel(), _f32(b).ravel()
    out = C.c_float()
    _check(_ffi.lib().isl_distance_squared(int(metric), _ptr(a), a.size, _ptr(b), b.size,
                                           C.byref(out)))
    return out.value


def batch_calculate(metric: DistanceMetric, query, vectors, device: int = 0) -> np.ndarray:
    """Distance::batch_calculate, distance.rs:32-34 (vectors: n rows)."""
    q = _f32(query).ravel()
    v = _f32(vectors)
    if v.ndim != 2:
        v = v.reshape(0, q.size) if v.size == 0 else v.reshape(1, -1)
    n, row_len = v.shape
    out = np.empty(n, dtype=np.float32)
    _check(_ffi.lib().isl_distance_batch(int(metric), _ptr(q), q.size, _ptr(v), n, row_len,
                                         _ptr(out), MEM_HOST, device, None))
    return out


def distance_matrix(metric: DistanceMetric, queries, rows, device: int = 0) -> np.ndarray:
    """All query x row distances as one float32 GEMM on the matrix cores -> [nq, n]."""
    q, r = _f32(queries), _f32(rows)
    out = np.zeros((q.shape[0], r.shape[0]), dtype=np.float32)
    _check(_ffi.lib().isl_distance_matrix(int(metric), _ptr(q), q.shape[0], _ptr(r), r.shape[0],
                                          q.shape[1], _ptr(out), MEM_HOST, device, None))
    return out


def distance_matrix_bf16(metric: DistanceMetric, queries_bits, rows_bits, device: int = 0, q_sumsq=None,
                         row_sumsq=None) -> np.ndarray:
    """distance_matrix over bf16 bit patterns (u16) on the bf16 matrix cores -> [nq, n] f32.  q_sumsq /
    row_sumsq: row_sumsq_bf16 of the operands when the caller keeps them (same outputs, bit for bit)."""
    q = np.ascontiguousarray(queries_bits, dtype=np.uint16)
    r = np.ascontiguousarray(rows_bits, dtype=np.uint16)
    out = np.zeros((q.shape[0], r.shape[0]), dtype=np.float32)
    if q_sumsq is None and row_sumsq is None:
        _check(_ffi.lib().isl_distance_matrix_bf16(int(metric), _ptr(q), q.shape[0], _ptr(r), r.shape[0],
                                                   q.shape[1], _ptr(out), MEM_HOST, device, None))
    else:
        qs = None if q_sumsq is None else np.ascontiguousarray(q_sumsq, dtype=np.float32)
        rs = None if row_sumsq is None else np.ascontiguousarray(row_sumsq, dtype=np.float32)
        _check(_ffi.lib().isl_distance_matrix_bf16_norms(int(metric), _ptr(q), q.shape[0], _ptr(r), r.shape[0],
                                                         q.shape[1], None if qs is None else _ptr(qs),
                                                         None if rs is None else _ptr(rs), _ptr(out), MEM_HOST,
                                                         device, None))
    return out


def row_sumsq_bf16(rows_bits, device: int = 0) -> np.ndarray:
    """Sum of squares of every row of a bf16 matrix (u16 bit patterns) as the distance epilogues take it."""
    r = np.ascontiguousarray(rows_bits, dtype=np.uint16)
    out = np.zeros(r.shape[0], dtype=np.float32)
    _check(_ffi.lib().isl_row_sumsq_bf16(_ptr(r), r.shape[0], r.shape[1], _ptr(out), MEM_HOST, device, None))
    return out


def bruteforce_topk(metric: DistanceMetric, queries, rows, k: int, device: int = 0):
    """Exact k nearest rows per query (ids, distances, counts), ties towards the smaller id."""
    q, r = _f32(queries), _f32(rows)
    nq = q.shape[0]
    ids = np.zeros((nq, max(k, 1)), dtype=np.uint64)
    dd = np.zeros((nq, max(k, 1)), dtype=np.float32)
    cnt = np.zeros(nq, dtype=np.uint32)
    _check(_ffi.lib().isl_bruteforce_topk(int(metric), _ptr(q), nq, _ptr(r), r.shape[0], q.shape[1], k,
                                          _ptr(ids), _ptr(dd), _ptr(cnt), MEM_HOST, device, None))
    return ids[:, :k], dd[:, :k], cnt


def bruteforce_topk_bf16(metric: DistanceMetric, queries_bits, rows_bits, k: int, device: int = 0):
    """bruteforce_topk over bf16 bit patterns (u16): exact under distance_matrix_bf16's distances."""
    q = np.ascontiguousarray(queries_bits, dtype=np.uint16)
    r = np.ascontiguousarray(rows_bits, dtype=np.uint16)
    nq = q.shape[0]
    ids = np.zeros((nq, max(k, 1)), dtype=np.uint64)
    dd = np.zeros((nq, max(k, 1)), dtype=np.float32)
    cnt = np.zeros(nq, dtype=np.uint32)
    _check(_ffi.lib().isl_bruteforce_topk_bf16(int(metric), _ptr(q), nq, _ptr(r), r.shape[0], q.shape[1], k,
                                               _ptr(ids), _ptr(dd), _ptr(cnt), MEM_HOST, device, None))
    return ids[:, :k], dd[:, :k], cnt


def normalize_rows(rows, device: int = 0) -> np.ndarray:
    """normalize_vector (distance.rs:125-132) applied to every row; returns a copy."""
    r = _f32(rows).copy()
    if r.ndim == 1:
        r = r.reshape(1, -1)
    _check(_ffi.lib().isl_normalize_rows(_ptr(r), r.shape[0], r.shape[1], MEM_HOST, device, None))
    return r


# ------------------------------------------------------------------ leann.rs
@dataclass
class LeannConfig:
    """src/core/leann.rs:322-371; defaults = paper_default() (leann.rs:386-403)."""
    m: int = 30
    m0: int = 60
    ef_construction: int = 128
    ml: float = 1.0 / float(np.log(30.0))
    max_layers: int = 16
    metric: DistanceMetric = DistanceMetric.Cosine
    ef_search: int = 64
    beam_width: int = 1
    prune_ratio: float = 0.0
    pruning_strategy: PruningStrategy = PruningStrategy.Global
    high_degree_pruning: bool = True
    hub_percentile: float = 0.02
    is_compact: bool = True
    is_recompute: bool = True

    @classmethod
    def _from_c(cls, c: LeannConfigC) -> "LeannConfig":
        return cls(c.m, c.m0, c.ef_construction, c.ml, c.max_layers, DistanceMetric(c.metric),
                   c.ef_search, c.beam_width, c.prune_ratio, PruningStrategy(c.pruning_strategy),
                   bool(c.high_degree_pruning), c.hub_percentile, bool(c.is_compact),
                   bool(c.is_recompute))

    def _to_c(self) -> LeannConfigC:
        return LeannConfigC(self.m, self.m0, self.ef_construction, self.ml, self.max_layers,
                            int(self.metric), self.ef_search, self.beam_width, self.prune_ratio,
                            int(self.pruning_strategy), int(self.high_degree_pruning),
                            self.hub_percentile, int(self.is_compact), int(self.is_recompute))

    @classmethod
    def paper_default(cls) -> "LeannConfig":
        c = LeannConfigC()
        _ffi.lib().isl_leann_config_paper_default(C.byref(c))
        return cls._from_c(c)

    @classmethod
    def fast(cls) -> "LeannConfig":
        c = LeannConfigC()
        _ffi.lib().isl_leann_config_fast(C.byref(c))
        return cls._from_c(c)

    @classmethod
    def accurate(cls) -> "LeannConfig":
        c = LeannConfigC()
        _ffi.lib().isl_leann_config_accurate(C.byref(c))
        return cls._from_c(c)

    def validate(self) -> None:
        c = self._to_c()
        _check(_ffi.lib().isl_leann_config_validate(C.byref(c)))


@dataclass
class CsrGraph:
    """CsrGraph's public fields, src/core/leann.rs:193-208."""
    node_offsets: np.ndarray = field(default_factory=lambda: np.zeros(1, dtype=np.uint64))
    neighbors: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.uint64))
    levels: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.uint64))
    entry_point: int | None = None
    max_level: int = 0
    num_nodes: int = 0
    degree_counts: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.uint64))

    def add_node(self, neighbors, level: int) -> int:
        """leann.rs:236-253"""
        nid = self.num_nodes
        self.num_nodes += 1
        nb = np.asarray(list(neighbors), dtype=np.uint64)
        self.levels = np.append(self.levels, np.uint64(level))
        self.degree_counts = np.append(self.degree_counts, np.uint64(nb.size))
        self.neighbors = np.concatenate([self.neighbors.astype(np.uint64), nb])
        self.node_offsets = np.append(self.node_offsets, np.uint64(self.neighbors.size))
        if self.entry_point is None or level > self.max_level:
            self.entry_point = nid
            self.max_level = level
        return nid

    def get_neighbors(self, node_id: int):
        """leann.rs:225-233"""
        if node_id >= self.num_nodes:
            return None
        s, e = int(self.node_offsets[node_id]), int(self.node_offsets[node_id + 1])
        return self.neighbors[s:e]

    def storage_bytes(self) -> int:
        """leann.rs:296-301"""
        return 8 * (self.node_offsets.size + self.neighbors.size + self.levels.size +
                    self.degree_counts.size)


class InMemoryEmbeddingProvider:
    """leann.rs:104-159: id -> row of a resident matrix."""

    def __init__(self, embeddings):
        e = _f32(embeddings)
        if e.ndim != 2 or e.shape[0] == 0:
            raise CoreError(2, "Empty vector collection")  # leann.rs:112-114
        self.embeddings = e

    def dimension(self) -> int:
        return int(self.embeddings.shape[1])

    def __len__(self):
        return int(self.embeddings.shape[0])


@dataclass
class SearchResult:
    """search.rs:54-103"""
    id: int
    score: float
    vector: np.ndarray | None = None
    metadata: object | None = None
    text: str | None = None

    def to_similarity(self) -> float:
        return float(np.float32(1.0) / (np.float32(1.0) + np.float32(self.score)))


class LeannIndex:
    """LeannIndex, src/core/leann.rs:492-1067, backed by an isl_index handle."""

    def __init__(self, config: LeannConfig | None = None, _handle=None):
        self._h = C.c_void_p()
        self._provider_key = None
        self._pending = {}  # token -> output arrays of a host-buffer search in flight
        if _handle is not None:
            self._h = _handle
            return
        c = (config or LeannConfig())._to_c()
        _check(_ffi.lib().isl_index_new(C.byref(c), C.byref(self._h)))  # leann.rs:504-511

    @classmethod
    def with_defaults(cls) -> "LeannIndex":
        return cls(LeannConfig())

    @classmethod
    def from_csr(cls, graph: CsrGraph, config: LeannConfig | None = None,
                 dimension: int | None = None) -> "LeannIndex":
        cfg = (config or LeannConfig())._to_c()
        off = np.ascontiguousarray(graph.node_offsets, dtype=np.uint64)
        nb = np.ascontiguousarray(graph.neighbors, dtype=np.uint64)
        n = int(graph.num_nodes)
        lv = np.ascontiguousarray(graph.levels, dtype=np.uint64)
        dg = np.ascontiguousarray(graph.degree_counts, dtype=np.uint64)
        h = C.c_void_p()
        _check(_ffi.lib().isl_index_from_csr(
            C.byref(cfg), n, _ptr(off), _ptr(nb) if nb.size else None,
            _ptr(lv) if lv.size == n and n else None, _ptr(dg) if dg.size == n and n else None,
            0 if graph.entry_point is None else 1, graph.entry_point or 0, graph.max_level,
            0 if dimension is None else 1, dimension or 0, C.byref(h)))
        return cls(_handle=h)

    @classmethod
    def build(cls, vectors, config: LeannConfig | None = None, levels=None, batch: int = 1,
              device: int = 0) -> "LeannIndex":
        """LeannIndex::build (leann.rs:560-630) on the device; `levels` replaces random_level
        (thread_rng); batch = 1 is the reference's sequential insertion, larger batches insert
        that many nodes per step.  The vectors become the in-memory provider."""
        v = _f32(vectors)
        n, d = v.shape if v.ndim == 2 else (0, 0)
        c = (config or LeannConfig())._to_c()
        lv = None if levels is None else np.ascontiguousarray(levels, dtype=np.uint64)
        h = C.c_void_p()
        _check(_ffi.lib().isl_index_build(C.byref(c), _ptr(v) if n else None, n, d,
                                          None if lv is None else _ptr(lv), batch, MEM_HOST, device,
                                          C.byref(h)))
        return cls(_handle=h)

    @classmethod
    def from_device_csr(cls, d_offsets_ptr: int, d_neighbors_ptr: int, num_nodes: int,
                        entry_point: int | None, dimension: int | None,
                        config: LeannConfig | None = None, device: int = 0) -> "LeannIndex":
        cfg = (config or LeannConfig())._to_c()
        h = C.c_void_p()
        _check(_ffi.lib().isl_index_from_device_csr(
            C.byref(cfg), device, num_nodes, C.c_void_p(d_offsets_ptr),
            C.c_void_p(d_neighbors_ptr), 0 if entry_point is None else 1, entry_point or 0,
            0 if dimension is None else 1, dimension or 0, C.byref(h)))
        return cls(_handle=h)

    @classmethod
    def from_bytes(cls, data: bytes) -> "LeannIndex":
        """leann.rs:1064-1066"""
        buf = (C.c_uint8 * len(data)).from_buffer_copy(data) if data else (C.c_uint8 * 1)()
        h = C.c_void_p()
        _check(_ffi.lib().isl_index_from_bytes(buf, len(data), C.byref(h)))
        return cls(_handle=h)

    def save(self, path: str, metadata: "IndexMetadata | None" = None) -> None:
        """One file: chunk META (IndexWriter::write_metadata, storage.rs:119-124) + chunk LIDX
        (to_bytes); parent directories are created (storage.rs:68-74)."""
        m = None if metadata is None else metadata._to_c()
        _check(_ffi.lib().isl_index_save(self._h, os.fsencode(path), None if m is None else C.byref(m)))

    @classmethod
    def load(cls, path: str) -> "tuple[LeannIndex, IndexMetadata]":
        h = C.c_void_p()
        m = IndexMetadataC()
        _check(_ffi.lib().isl_index_load(os.fsencode(path), C.byref(h), C.byref(m)))
        return cls(_handle=h), IndexMetadata._from_c(m)

    def to_bytes(self) -> bytes:
        """leann.rs:1059-1061"""
        p = C.c_void_p()
        n = C.c_size_t()
        _check(_ffi.lib().isl_index_to_bytes(self._h, C.byref(p), C.byref(n)))
        try:
            if n.value < (1 << 30):
                data = C.string_at(p, n.value)
            else:  # (string_at takes a C int: a 10M-node index serialises to 5 GB)
                data = bytes((C.c_char * n.value).from_address(p.value))
        finally:
            _ffi.lib().isl_free_bytes(p)
        return data

    def close(self) -> None:
        """isl_index_free now (instead of whenever the object is collected)."""
        h = getattr(self, "_h", None)
        if h:
            try:
                _ffi.lib().isl_index_free(h)
            except Exception:
                pass
            self._h = None

    def __del__(self):
        self.close()

    # accessors, leann.rs:518-546
    def __len__(self) -> int:
        return int(_ffi.lib().isl_index_len(self._h))

    def len(self) -> int:
        return len(self)

    def is_empty(self) -> bool:
        return bool(_ffi.lib().isl_index_is_empty(self._h))

    def dimension(self) -> int | None:
        d = C.c_uint64()
        return int(d.value) if _ffi.lib().isl_index_dimension(self._h, C.byref(d)) else None

    def storage_bytes(self) -> int:
        return int(_ffi.lib().isl_index_storage_bytes(self._h))

    def is_recompute(self) -> bool:
        return bool(_ffi.lib().isl_index_is_recompute(self._h))

    def is_compact(self) -> bool:
        return bool(_ffi.lib().isl_index_is_compact(self._h))

    @property
    def config(self) -> LeannConfig:
        c = LeannConfigC()
        _check(_ffi.lib().isl_index_config(self._h, C.byref(c)))
        return LeannConfig._from_c(c)

    @property
    def entry_point(self) -> int | None:
        e = C.c_uint64()
        return int(e.value) if _ffi.lib().isl_index_entry_point(self._h, C.byref(e)) else None

    def get_neighbors(self, node: int):
        ptr = C.POINTER(C.c_uint64)()
        n = C.c_size_t()
        if not _ffi.lib().isl_index_get_neighbors(self._h, node, C.byref(ptr), C.byref(n)):
            return None
        return np.array([ptr[i] for i in range(n.value)], dtype=np.uint64)

    # device residency
    def upload(self, device: int = 0) -> "LeannIndex":
        _check(_ffi.lib().isl_index_upload(self._h, device))
        return self

    def set_recompute_provider(self, embedder: "CandleEmbedder", tokens=None, lengths=None,
                               keep_rows: bool = False, cache_rows: int | None = None,
                               device_ptr: int | None = None, n: int | None = None,
                               L: int | None = None) -> "LeannIndex":
        """EmbeddingProvider backed by the encoder (recompute mode, leann.rs:82-99): `tokens`
        [n, L] uint16 holds node i's tokenised text in row i, `lengths[i]` slots of it used (or a
        device pointer to such a table).  cache_rows = rows of the bounded embedding cache."""
        if device_ptr is not None:
            _check(_ffi.lib().isl_set_recompute_provider(
                self._h, embedder._h, C.c_void_p(device_ptr), None, n, L, int(embedder.normalize),
                int(keep_rows), MEM_DEVICE))
        else:
            t = np.ascontiguousarray(tokens, dtype=np.uint16)
            ln = None if lengths is None else np.ascontiguousarray(lengths, dtype=np.uint16)
            n, L = t.shape
            _check(_ffi.lib().isl_set_recompute_provider(
                self._h, embedder._h, _ptr(t), None if ln is None else _ptr(ln), n, L,
                int(embedder.normalize), int(keep_rows), MEM_HOST))
        if cache_rows is not None:
            _check(_ffi.lib().isl_index_set_recompute_cache_rows(self._h, cache_rows))
        self._embedder = embedder  # borrowed by the library: keep it alive
        return self

    def recompute_cache_bytes(self) -> int:
        return int(_ffi.lib().isl_index_recompute_cache_bytes(self._h))

    def set_embeddings(self, rows, device_ptr: int | None = None, n: int | None = None,
                       d: int | None = None) -> "LeannIndex":
        """Attach the in-memory provider (leann.rs:111-120).  Either a host matrix, or
        (device_ptr, n, d) for rows already resident on the index's device."""
        if device_ptr is not None:
            _check(_ffi.lib().isl_set_embeddings(self._h, C.c_void_p(device_ptr), n, d, 0,
                                                 MEM_DEVICE))
        else:
            r = _f32(rows)
            if r.ndim != 2 or r.shape[0] == 0:
                raise CoreError(2, "Empty vector collection")
            _check(_ffi.lib().isl_set_embeddings(self._h, _ptr(r), r.shape[0], r.shape[1], 0,
                                                 MEM_HOST))
        return self

    def set_embeddings_bf16(self, rows_bits, device_ptr: int | None = None, n: int | None = None,
                            d: int | None = None) -> "LeannIndex":
        """In-memory provider with bf16 storage: `rows_bits` [n, d] uint16 bit patterns (or a
        device pointer to them).  The provider's vectors are their exact f32 images."""
        if device_ptr is not None:
            _check(_ffi.lib().isl_set_embeddings(self._h, C.c_void_p(device_ptr), n, d, 1, MEM_DEVICE))
        else:
            r = np.ascontiguousarray(rows_bits, dtype=np.uint16)
            if r.ndim != 2 or r.shape[0] == 0:
                raise CoreError(2, "Empty vector collection")
            _check(_ffi.lib().isl_set_embeddings(self._h, _ptr(r), r.shape[0], r.shape[1], 1, MEM_HOST))
        return self

    def _attach(self, provider):
        if isinstance(provider, InMemoryEmbeddingProvider):
            key = id(provider.embeddings)
            if self._provider_key != key:
                self.set_embeddings(provider.embeddings)
                self._provider_key = key
        elif provider is not None:
            raise TypeError("provider must be an InMemoryEmbeddingProvider")

    # search, leann.rs:858-896
    def search(self, query, k: int, provider=None):
        return self.search_with_params(query, k, self.config.ef_search, provider)

    def search_with_params(self, query, k: int, ef: int, provider=None):
        ids, dist, cnt = self.search_batch(_f32(query).reshape(1, -1), k, ef, provider)
        return [(int(ids[0, i]), float(dist[0, i])) for i in range(int(cnt[0]))]

    def search_batch(self, queries, k: int, ef: int, provider=None):
        """Batch of independent queries (Searcher::search_batch semantics, search.rs:179-181).
        Returns (ids [nq,k] u64, dist [nq,k] f32, count [nq] u32)."""
        self._attach(provider)
        q = _f32(queries)
        if q.ndim == 1:
            q = q.reshape(1, -1)
        nq, d = q.shape
        ids = np.zeros((nq, max(k, 1)), dtype=np.uint64)
        dist = np.zeros((nq, max(k, 1)), dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.uint32)
        _check(_ffi.lib().isl_search_batch(self._h, _ptr(q), nq, d, k, ef, _ptr(ids), _ptr(dist),
                                           _ptr(cnt)))
        return ids[:, :k], dist[:, :k], cnt

    # ---- extension: two-level search with a PQ filter (docs/leann-specification.md:223-275) ----
    def set_pq_codes(self, pq: "ProductQuantizer", codes=None, device_ptr: int | None = None,
                     n: int | None = None):
        """Attach the quantizer (borrowed) and the code row of every node, [n][pq.m] u16 as
        ProductQuantizer::encode returns them (pq.rs:221-244)."""
        if device_ptr is not None:
            _check(_ffi.lib().isl_index_set_pq_codes(self._h, pq._h, C.c_void_p(device_ptr), n,
                                                     MEM_DEVICE))
        else:
            c = np.ascontiguousarray(codes, dtype=np.uint16)
            if c.ndim != 2 or c.shape[1] != pq.m:
                raise CoreError(10, "PQ error: codes must be [n][num_subquantizers]")
            _check(_ffi.lib().isl_index_set_pq_codes(self._h, pq._h, _ptr(c), c.shape[0], MEM_HOST))
        self._pq = pq  # keep the borrowed quantizer alive
        return self

    def search_two_level_batch(self, queries, k: int, ef: int, rerank_ratio: float, provider=None):
        """Algorithm 2 of the reference's specification (not implemented there): PQ distances for
        every new neighbour, exact distances for the top `rerank_ratio` of the approximate queue.
        Same return values as search_batch."""
        self._attach(provider)
        q = _f32(queries)
        if q.ndim == 1:
            q = q.reshape(1, -1)
        nq, d = q.shape
        ids = np.zeros((nq, max(k, 1)), dtype=np.uint64)
        dist = np.zeros((nq, max(k, 1)), dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.uint32)
        _check(_ffi.lib().isl_search_two_level_batch(self._h, _ptr(q), nq, d, k, ef,
                                                     C.c_float(rerank_ratio), _ptr(ids), _ptr(dist),
                                                     _ptr(cnt)))
        return ids[:, :k], dist[:, :k], cnt

    def search_two_level_batch_device(self, d_queries_ptr: int, nq: int, d: int, k: int, ef: int,
                                      rerank_ratio: float, d_ids_ptr: int, d_dist_ptr: int,
                                      d_count_ptr: int, stream: int = 0):
        _check(_ffi.lib().isl_search_two_level_batch_device(
            self._h, C.c_void_p(d_queries_ptr), nq, d, k, ef, C.c_float(rerank_ratio),
            C.c_void_p(d_ids_ptr), C.c_void_p(d_dist_ptr), C.c_void_p(d_count_ptr),
            C.c_void_p(stream)))

    def search_two_level_batch_device_async(self, d_queries_ptr: int, nq: int, d: int, k: int, ef: int,
                                            rerank_ratio: float, d_ids_ptr: int, d_dist_ptr: int,
                                            d_count_ptr: int, stream: int = 0) -> int:
        """isl_search_two_level_batch_device_async: returns a token for wait() / wait_stats()."""
        tok = C.c_uint64()
        _check(_ffi.lib().isl_search_two_level_batch_device_async(
            self._h, C.c_void_p(d_queries_ptr), nq, d, k, ef, C.c_float(rerank_ratio),
            C.c_void_p(d_ids_ptr), C.c_void_p(d_dist_ptr), C.c_void_p(d_count_ptr),
            C.c_void_p(stream), C.byref(tok)))
        return int(tok.value)

    def search_batch_device(self, d_queries_ptr: int, nq: int, d: int, k: int, ef: int,
                            d_ids_ptr: int, d_dist_ptr: int, d_count_ptr: int, stream: int = 0):
        _check(_ffi.lib().isl_search_batch_device(
            self._h, C.c_void_p(d_queries_ptr), nq, d, k, ef, C.c_void_p(d_ids_ptr),
            C.c_void_p(d_dist_ptr), C.c_void_p(d_count_ptr), C.c_void_p(stream)))

    def search_batch_device_async(self, d_queries_ptr: int, nq: int, d: int, k: int, ef: int,
                                  d_ids_ptr: int, d_dist_ptr: int, d_count_ptr: int,
                                  stream: int = 0) -> int:
        """Enqueue a search; returns a token for wait().  Up to 16 may be in flight."""
        tok = C.c_uint64()
        _check(_ffi.lib().isl_search_batch_device_async(
            self._h, C.c_void_p(d_queries_ptr), nq, d, k, ef, C.c_void_p(d_ids_ptr),
            C.c_void_p(d_dist_ptr), C.c_void_p(d_count_ptr), C.c_void_p(stream), C.byref(tok)))
        return int(tok.value)

    def wait(self, token: int):
        """Completes the call; for a host-buffer call returns its (ids, dist, count) arrays."""
        try:  # the library writes the answers during the wait: the arrays must outlive it
            _check(_ffi.lib().isl_search_wait(self._h, token))
        finally:
            out = self._pending.pop(token, None)
        return out

    def wait_stats(self, token: int) -> dict:
        """wait() that returns the counters of exactly that call (isl_search_wait_stats)."""
        s = SearchStatsC()
        try:
            _check(_ffi.lib().isl_search_wait_stats(self._h, token, C.byref(s)))
        finally:
            self._pending.pop(token, None)
        return {f: getattr(s, f) for f, _ in SearchStatsC._fields_}

    def prepare(self, max_nq: int, max_ef: int, max_k: int = 10, lanes: int = 8) -> "LeannIndex":
        """isl_index_prepare: every lane's buffers, the padded adjacency and the exact-kernel pool
        up front, so that no later search allocates (stats["allocations"] == 0)."""
        _check(_ffi.lib().isl_index_prepare(self._h, max_nq, max_ef, max_k, lanes))
        return self

    def search_batch_async(self, queries, k: int, ef: int, out=None) -> int:
        """Pipelined host-buffer search (isl_search_batch_async): returns a token; the arrays of
        `out` = (ids [nq,k] u64, dist [nq,k] f32, count [nq] u32) -- allocated here when None and
        available through result(token) -- are filled once wait(token) has returned."""
        q = _f32(queries)
        if q.ndim == 1:
            q = q.reshape(1, -1)
        nq, d = q.shape
        if out is None:
            out = (np.zeros((nq, max(k, 1)), dtype=np.uint64), np.zeros((nq, max(k, 1)), dtype=np.float32),
                   np.zeros(nq, dtype=np.uint32))
        ids, dist, cnt = out
        tok = C.c_uint64()
        _check(_ffi.lib().isl_search_batch_async(self._h, _ptr(q), nq, d, k, ef, _ptr(ids), _ptr(dist),
                                                 _ptr(cnt), C.byref(tok)))
        t = int(tok.value)
        if t:
            self._pending[t] = out  # the library writes into them at wait(): keep them alive
        return t

    def last_stats(self) -> dict:
        s = SearchStatsC()
        _check(_ffi.lib().isl_search_last_stats(self._h, C.byref(s)))
        return {f: getattr(s, f) for f, _ in SearchStatsC._fields_}


# ----------------------------------------------------------------- search.rs
def merge_topk(ids, scores, counts, top_k: int, id_base=None, device: int = 0):
    """MultiIndexSearcher::search merge, search.rs:211-237.
    ids/scores: [nlists, nq, k]; counts: [nlists, nq].  Returns (ids, scores, src, count)."""
    ids = np.ascontiguousarray(ids, dtype=np.uint64)
    scores = _f32(scores)
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    nl, nq, k = ids.shape
    base = None if id_base is None else np.ascontiguousarray(id_base, dtype=np.uint64)
    oi = np.zeros((nq, max(top_k, 1)), dtype=np.uint64)
    osc = np.zeros((nq, max(top_k, 1)), dtype=np.float32)
    osrc = np.zeros((nq, max(top_k, 1)), dtype=np.uint32)
    oc = np.zeros(nq, dtype=np.uint32)
    _check(_ffi.lib().isl_merge_topk(nl, nq, k, _ptr(ids), _ptr(scores), _ptr(counts),
                                     None if base is None else _ptr(base), top_k, _ptr(oi),
                                     _ptr(osc), _ptr(osrc), _ptr(oc), MEM_HOST, device, None))
    return oi[:, :top_k], osc[:, :top_k], osrc[:, :top_k], oc


def merge_service(ids, distances, counts, top_k: int, files_len=None, device: int = 0):
    """Cross-index merge of IndexerService::search_with_embeddings (indexer/service.rs:775-801):
    ids without a file entry dropped, score = 1 - distance, stable sort by score descending,
    truncate.  ids/distances: [nlists, nq, k]; counts: [nlists, nq].  Returns (ids, scores, src, count)."""
    ids = np.ascontiguousarray(ids, dtype=np.uint64)
    distances = _f32(distances)
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    nl, nq, k = ids.shape
    fl = None if files_len is None else np.ascontiguousarray(files_len, dtype=np.uint64)
    oi = np.zeros((nq, max(top_k, 1)), dtype=np.uint64)
    osc = np.zeros((nq, max(top_k, 1)), dtype=np.float32)
    osrc = np.zeros((nq, max(top_k, 1)), dtype=np.uint32)
    oc = np.zeros(nq, dtype=np.uint32)
    _check(_ffi.lib().isl_merge_service(nl, nq, k, _ptr(ids), _ptr(distances), _ptr(counts),
                                        None if fl is None else _ptr(fl), top_k, _ptr(oi),
                                        _ptr(osc), _ptr(osrc), _ptr(oc), MEM_HOST, device, None))
    return oi[:, :top_k], osc[:, :top_k], osrc[:, :top_k], oc


# --------------------------------------------------------------------- pq.rs
class ProductQuantizer:
    """ProductQuantizer with trained codebooks (pq.rs:116-348); training (k-means) is out of
    scope of the search path."""

    def __init__(self, dimension: int, codebooks, metric=DistanceMetric.Euclidean, device: int = 0):
        cb = _f32(codebooks)
        m, K, dsub = cb.shape
        if m * dsub != dimension:
            raise CoreError(3, f"Invalid configuration: dimension {dimension} must be divisible by "
                               f"num_subquantizers {m}")
        self.dimension, self.m, self.K, self.dsub = dimension, m, K, dsub
        self._h = C.c_void_p()
        _check(_ffi.lib().isl_pq_new(dimension, m, K, _ptr(cb), int(metric), device,
                                     C.byref(self._h)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _ffi.lib().isl_pq_free(h)
            except Exception:
                pass
            self._h = None

    def num_subquantizers(self) -> int:
        return self.m

    def bytes_per_vector(self) -> int:  # pq.rs:58-64
        return self.m if self.K <= 256 else self.m * 2

    def compression_ratio(self) -> float:  # pq.rs:168-172
        return float(np.float32(self.dimension * 4) / np.float32(self.bytes_per_vector()))

    def build_distance_tables(self, queries) -> np.ndarray:
        q = _f32(queries)
        single = q.ndim == 1
        if single:
            q = q.reshape(1, -1)
        nq, d = q.shape
        t = np.zeros((nq, self.m, self.K), dtype=np.float32)
        _check(_ffi.lib().isl_pq_build_distance_tables(self._h, _ptr(q), nq, d, _ptr(t), MEM_HOST,
                                                       None))
        return t[0] if single else t

    def table_distance(self, tables, codes) -> np.ndarray:
        t = _f32(tables)
        c = np.ascontiguousarray(codes, dtype=np.uint16).reshape(-1, self.m)
        out = np.zeros(c.shape[0], dtype=np.float32)
        _check(_ffi.lib().isl_pq_table_distance(self._h, _ptr(t), _ptr(c), c.shape[0], _ptr(out),
                                                MEM_HOST, None))
        return out

    def asymmetric_distance(self, query, codes) -> np.ndarray:
        q = _f32(query).ravel()
        c = np.ascontiguousarray(codes, dtype=np.uint16).reshape(-1, self.m)
        out = np.zeros(c.shape[0], dtype=np.float32)
        _check(_ffi.lib().isl_pq_asymmetric_distance(self._h, _ptr(q), q.size, _ptr(c),
                                                     c.shape[0], _ptr(out), MEM_HOST, None))
        return out

    def encode(self, vectors) -> np.ndarray:
        v = _f32(vectors)
        single = v.ndim == 1
        if single:
            v = v.reshape(1, -1)
        n, d = v.shape
        codes = np.zeros((n, self.m), dtype=np.uint16)
        _check(_ffi.lib().isl_pq_encode(self._h, _ptr(v), n, d, _ptr(codes), MEM_HOST, None))
        return codes[0] if single else codes


# ------------------------------------------------------------------- hnsw.rs
class HnswGraph:
    """Search side of HnswGraph (hnsw.rs:149-515) on the device.  Construction (insert,
    hnsw.rs:214-329) is outside the search path: the graph is handed over as per-layer adjacency
    (`layers[L][node]` = neighbour ids of `node` on layer L, empty above the node's level)."""

    def __init__(self, vectors, layers, levels, entry_point: int | None, max_level: int,
                 m: int = 16, m0: int = 32, ef_construction: int = 200,
                 metric: DistanceMetric = DistanceMetric.Cosine, device: int = 0):
        v = _f32(vectors)
        if v.ndim != 2:
            v = v.reshape(len(levels), -1) if len(levels) else v.reshape(0, 0)
        n, d = (v.shape if v.size else (0, 0))
        self.vectors, self.m, self.m0 = v, m, m0
        self.ef_construction, self.metric = ef_construction, DistanceMetric(metric)
        offs, adjs = [], []
        for L in range(len(layers)):
            lens = np.fromiter((len(layers[L][i]) for i in range(n)), dtype=np.uint64, count=n)
            off = np.zeros(n + 1, dtype=np.uint64)
            np.cumsum(lens, out=off[1:])
            flat = [x for i in range(n) for x in layers[L][i]]
            adj = np.asarray(flat if flat else [0], dtype=np.uint64)
            offs.append(off)
            adjs.append(adj)
        self._keep = (offs, adjs)
        nl = len(layers)
        poff = (C.c_void_p * max(nl, 1))(*[o.ctypes.data for o in offs])
        padj = (C.c_void_p * max(nl, 1))(*[a.ctypes.data for a in adjs])
        lv = np.ascontiguousarray(levels, dtype=np.uint64)
        h = C.c_void_p()
        _check(_ffi.lib().isl_hnsw_from_layers(
            m, m0, ef_construction, int(metric), n, d, nl, poff, padj,
            _ptr(lv) if n else None, 0 if entry_point is None else 1, entry_point or 0,
            max_level, _ptr(v) if n else None, device, C.byref(h)))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                _ffi.lib().isl_hnsw_free(self._h)
            except Exception:
                pass
            self._h = None

    def __len__(self) -> int:
        return int(_ffi.lib().isl_hnsw_len(self._h))

    def len(self) -> int:
        return len(self)

    def is_empty(self) -> bool:
        return len(self) == 0

    @classmethod
    def from_bytes(cls, data: bytes, device: int = 0) -> "HnswGraph":
        """HnswGraph::from_bytes (hnsw.rs:511-514): the bincode image of a HnswGraph."""
        g = cls.__new__(cls)
        buf = np.frombuffer(data, dtype=np.uint8)
        h = C.c_void_p()
        _check(_ffi.lib().isl_hnsw_from_bytes(_ptr(buf) if buf.size else None, buf.size, device,
                                              C.byref(h)))
        g._h = h
        g._keep = None
        g.vectors = None
        return g

    def last_stats(self) -> dict:
        s = SearchStatsC()
        _check(_ffi.lib().isl_hnsw_last_stats(self._h, C.byref(s)))
        return {f: getattr(s, f) for f, _ in SearchStatsC._fields_}

    def get_vector(self, node: int):
        """get_node(id).vector, hnsw.rs:507-510"""
        return self.vectors[node] if 0 <= node < self.vectors.shape[0] else None

    def search_batch(self, queries, k: int, ef: int):
        """Batched HnswGraph::search (hnsw.rs:458-504): per query the (id, distance) list."""
        q = _f32(queries)
        if q.ndim == 1:
            q = q.reshape(1, -1)
        nq, d = q.shape
        ids = np.zeros((nq, max(k, 1)), dtype=np.uint64)
        dd = np.zeros((nq, max(k, 1)), dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.uint32)
        _check(_ffi.lib().isl_hnsw_search_batch(self._h, _ptr(q), nq, d, k, ef, _ptr(ids),
                                                _ptr(dd), _ptr(cnt)))
        return [(ids[i, :cnt[i]].copy(), dd[i, :cnt[i]].copy()) for i in range(nq)]

    def search(self, query, k: int, ef: int):
        ids, dd = self.search_batch(query, k, ef)[0]
        return [(int(i), float(s)) for i, s in zip(ids, dd)]


# ----------------------------------------------------------------- search.rs
@dataclass
class SearchConfig:
    """SearchConfig, search.rs:8-52"""
    top_k: int = 10
    ef: int = 100
    include_vectors: bool = False
    include_metadata: bool = True
    min_similarity: float | None = None

    @classmethod
    def fast(cls, k: int) -> "SearchConfig":
        return cls(top_k=k, ef=k * 2)

    @classmethod
    def accurate(cls, k: int) -> "SearchConfig":
        return cls(top_k=k, ef=k * 10)


def _to_results(graph, ids, dd, cfg: SearchConfig):
    out = []
    for i, s in zip(ids, dd):
        r = SearchResult(int(i), float(s))
        if cfg.include_vectors:
            vec = graph.get_vector(int(i))
            if vec is not None:
                r.vector = vec.copy()
        out.append(r)
    return out


class Searcher:
    """Searcher, search.rs:105-182.  `search_batch` is one device launch instead of the
    reference's sequential map (search.rs:179-181)."""

    def __init__(self, graph: HnswGraph, config: SearchConfig | None = None):
        self.graph = graph
        self.config = config or SearchConfig()

    @classmethod
    def with_config(cls, graph: HnswGraph, config: SearchConfig) -> "Searcher":
        return cls(graph, config)

    def top_k(self, k: int) -> "Searcher":
        self.config.top_k = k
        return self

    def ef(self, ef: int) -> "Searcher":
        self.config.ef = ef
        return self

    def include_vectors(self) -> "Searcher":
        self.config.include_vectors = True
        return self

    def min_similarity(self, threshold: float) -> "Searcher":
        self.config.min_similarity = threshold
        return self

    def search_batch(self, queries):
        cfg = self.config
        out = []
        for ids, dd in self.graph.search_batch(queries, cfg.top_k, cfg.ef):
            res = _to_results(self.graph, ids, dd, cfg)
            if cfg.min_similarity is not None:  # retain(to_similarity() >= min_sim), f32
                thr = np.float32(cfg.min_similarity)
                res = [r for r in res
                       if np.float32(1.0) / (np.float32(1.0) + np.float32(r.score)) >= thr]
            out.append(res)
        return out

    def search(self, query):
        return self.search_batch(_f32(query).reshape(1, -1))[0]


class MultiIndexSearcher:
    """MultiIndexSearcher, search.rs:184-256: every index answers the query, the lists are
    concatenated in insertion order, stable-sorted by score and truncated (isl_merge_topk)."""

    def __init__(self, config: SearchConfig | None = None, device: int = 0):
        self.graphs: list[tuple[str, HnswGraph]] = []
        self.config = config or SearchConfig()
        self.device = device

    def add_index(self, name: str, graph: HnswGraph) -> None:
        self.graphs.append((name, graph))

    def with_config(self, config: SearchConfig) -> "MultiIndexSearcher":
        self.config = config
        return self

    def num_indexes(self) -> int:
        return len(self.graphs)

    def total_vectors(self) -> int:
        return sum(len(g) for _, g in self.graphs)

    def search_batch(self, queries):
        cfg = self.config
        q = _f32(queries)
        if q.ndim == 1:
            q = q.reshape(1, -1)
        nq, k, nl = q.shape[0], cfg.top_k, len(self.graphs)
        if nl == 0:
            return [[] for _ in range(nq)]
        ids = np.zeros((nl, nq, max(k, 1)), dtype=np.uint64)
        sc = np.zeros((nl, nq, max(k, 1)), dtype=np.float32)
        cnt = np.zeros((nl, nq), dtype=np.uint32)
        for li, (_, g) in enumerate(self.graphs):
            for qi, (a, b) in enumerate(g.search_batch(q, k, cfg.ef)):
                ids[li, qi, :len(a)], sc[li, qi, :len(a)], cnt[li, qi] = a, b, len(a)
        mi, ms, src, mc = merge_topk(ids, sc, cnt, k, device=self.device)
        out = []
        for qi in range(nq):
            row = []
            for j in range(int(mc[qi])):
                name, g = self.graphs[int(src[qi, j])]
                row.append((name, _to_results(g, [mi[qi, j]], [ms[qi, j]], cfg)[0]))
            out.append(row)
        return out

    def search(self, query):
        return self.search_batch(_f32(query).reshape(1, -1))[0]


# -------------------------------------------------------- indexer/service.rs
def service_search(embedder, indexes, input_ids, attention_mask, top_k: int, device: int = 0):
    """The device side of IndexerService::search_with_embeddings (indexer/service.rs:747-801) after
    tokenisation: embed the query (embed_texts_raw), search every index's HnswGraph with
    ef = max(top_k, 100) (:781), drop ids without a file entry, score = 1 - distance, sort by
    score descending, truncate.  `indexes`: list of (name, HnswGraph, number_of_files).
    Returns [(score, name, id)]."""
    q = embedder.embed(np.asarray(input_ids).reshape(1, -1),
                       None, None if attention_mask is None else np.asarray(attention_mask).reshape(1, -1))
    ef = max(top_k, 100)
    nl = len(indexes)
    if nl == 0:
        return []
    ids = np.zeros((nl, 1, max(top_k, 1)), dtype=np.uint64)
    dist = np.zeros((nl, 1, max(top_k, 1)), dtype=np.float32)
    cnt = np.zeros((nl, 1), dtype=np.uint32)
    for li, (_, graph, _) in enumerate(indexes):
        a, b = graph.search_batch(q, top_k, ef)[0]
        ids[li, 0, :len(a)], dist[li, 0, :len(a)], cnt[li, 0] = a, b, len(a)
    files = np.array([f for _, _, f in indexes], dtype=np.uint64)
    mi, ms, src, mc = merge_service(ids, dist, cnt, top_k, files_len=files, device=device)
    return [(float(ms[0, j]), indexes[int(src[0, j])][0], int(mi[0, j])) for j in range(int(mc[0]))]


# ------------------------------------------- embedding/candle_provider.rs
def mean_pool_normalize(hidden, mask, normalize: bool = True, device: int = 0) -> np.ndarray:
    """Masked mean pooling + optional L2 normalisation of encoder outputs
    (src/core/embedding/candle_provider.rs:434-488: sum(h * mask) / clamp(sum(mask), 1e-9), then
    x / sqrt(sum(x^2))).  hidden [B, L, H], mask [B, L] -> [B, H]."""
    h, mk = _f32(hidden), _f32(mask)
    B, L, H = h.shape
    out = np.zeros((B, H), dtype=np.float32)
    _check(_ffi.lib().isl_mean_pool_normalize(_ptr(h), _ptr(mk), B, L, H, int(normalize),
                                              _ptr(out), MEM_HOST, device, None))
    return out


@dataclass
class IndexMetadata:
    """IndexMetadata, src/core/storage.rs:16-47 (`now` replaces chrono::Utc::now())."""
    version: int = 1
    num_vectors: int = 0
    dimension: int = 0
    created_at: int = 0
    updated_at: int = 0
    description: str | None = None

    @classmethod
    def new(cls, num_vectors: int, dimension: int, now: int) -> "IndexMetadata":
        m = IndexMetadataC()
        _ffi.lib().isl_index_metadata_new(num_vectors, dimension, now, C.byref(m))
        return cls._from_c(m)

    @classmethod
    def _from_c(cls, m: IndexMetadataC) -> "IndexMetadata":
        return cls(m.version, m.num_vectors, m.dimension, m.created_at, m.updated_at,
                   m.description.decode("utf-8") if m.has_description else None)

    def _to_c(self) -> IndexMetadataC:
        d = (self.description or "").encode("utf-8")
        return IndexMetadataC(self.version, self.num_vectors, self.dimension, self.created_at,
                              self.updated_at, int(self.description is not None), d)

    def to_chunk(self) -> bytes:
        """IndexWriter::write_metadata into a buffer: b"META" + u64 LE length + JSON."""
        m, p, n = self._to_c(), C.c_void_p(), C.c_size_t()
        _check(_ffi.lib().isl_storage_write_metadata(C.byref(m), C.byref(p), C.byref(n)))
        data = C.string_at(p, n.value)
        _ffi.lib().isl_free_bytes(p)
        return data

    @classmethod
    def from_chunk(cls, data: bytes) -> "IndexMetadata":
        """IndexReader::read_metadata on a buffer."""
        buf = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data.ljust(1, b"\0"))
        m = IndexMetadataC()
        _check(_ffi.lib().isl_storage_read_metadata(buf, len(data), C.byref(m), None))
        return cls._from_c(m)


@dataclass
class BertConfig:
    """The config.json fields BertModel's forward pass uses (candle_provider.rs:258-284)."""
    vocab_size: int = 30522
    hidden: int = 384
    layers: int = 6
    heads: int = 12
    intermediate: int = 1536
    max_position: int = 512
    type_vocab: int = 2
    layer_norm_eps: float = 1e-12
    gelu_tanh: bool = False

    def _to_c(self) -> BertConfigC:
        return BertConfigC(self.vocab_size, self.hidden, self.layers, self.heads, self.intermediate,
                           self.max_position, self.type_vocab, self.layer_norm_eps, int(self.gelu_tanh))


class CandleEmbedder:
    """Device twin of CandleEmbedder (candle_provider.rs:226-507) after tokenisation: weights by
    their checkpoint names, `embed` = embed_texts_raw on padded token ids."""

    def __init__(self, config: BertConfig, weights: dict | None = None, normalize: bool = True,
                 device: int = 0):
        self.config, self.normalize, self.device = config, normalize, device
        h = C.c_void_p()
        c = config._to_c()
        _check(_ffi.lib().isl_encoder_new(C.byref(c), device, C.byref(h)))
        self._h = h
        for name, value in (weights or {}).items():
            self.set_weight(name, value)

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                _ffi.lib().isl_encoder_free(self._h)
            except Exception:
                pass
            self._h = None

    def set_weight(self, name: str, value) -> None:
        v = _f32(value)
        _check(_ffi.lib().isl_encoder_set_weight(self._h, name.encode(), _ptr(v), v.size, MEM_HOST))

    def set_precision(self, bf16: bool) -> None:
        """float32 Linear layers (default, the reference's arithmetic) or bf16 inputs with float32
        accumulation on the bf16 matrix cores; call after the weights are set."""
        _check(_ffi.lib().isl_encoder_set_precision(self._h, 1 if bf16 else 0))

    def dimension(self) -> int:
        return self.config.hidden

    @staticmethod
    def _inputs(input_ids, token_type_ids, attention_mask):
        ids = np.ascontiguousarray(input_ids, dtype=np.int64)
        tt = None if token_type_ids is None else np.ascontiguousarray(token_type_ids, dtype=np.int64)
        mk = None if attention_mask is None else _f32(attention_mask)
        return ids, tt, mk

    def forward(self, input_ids, token_type_ids=None, attention_mask=None) -> np.ndarray:
        ids, tt, mk = self._inputs(input_ids, token_type_ids, attention_mask)
        B, L = ids.shape
        out = np.zeros((B, L, self.config.hidden), dtype=np.float32)
        _check(_ffi.lib().isl_encoder_forward(self._h, _ptr(ids), None if tt is None else _ptr(tt),
                                              None if mk is None else _ptr(mk), B, L, _ptr(out),
                                              MEM_HOST, None))
        return out

    def embed(self, input_ids, token_type_ids=None, attention_mask=None) -> np.ndarray:
        ids, tt, mk = self._inputs(input_ids, token_type_ids, attention_mask)
        B, L = ids.shape
        out = np.zeros((B, self.config.hidden), dtype=np.float32)
        _check(_ffi.lib().isl_encoder_embed(self._h, _ptr(ids), None if tt is None else _ptr(tt),
                                            None if mk is None else _ptr(mk), B, L,
                                            int(self.normalize), _ptr(out), MEM_HOST, None))
        return out
