"""CPU-side checks of the C ABI: the library loads, exports every symbol that
include/islands_amd.h declares, and the host-only entry points (config,
bincode, CSR accessors) behave like the reference.  No compute call needs a GPU here."""
import os
import re

import numpy as np
import pytest

import islands_amd as ia
from islands_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "islands_amd.h")).read()
    declared = set(re.findall(r"\b(isl_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"isl_status"}
    lib = _ffi.lib()
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, missing
    assert declared == set(_ffi.SIGNATURES), declared ^ set(_ffi.SIGNATURES)
    assert lib.isl_abi_version() == 3


def test_config_presets():  # leann.rs:1091-1143
    c = ia.LeannConfig.paper_default()
    assert (c.m, c.m0, c.ef_construction, c.ef_search) == (30, 60, 128, 64)
    assert c.is_compact and c.is_recompute and c.high_degree_pruning
    assert abs(c.hub_percentile - 0.02) < 1e-3 and abs(c.ml - 1 / np.log(30)) < 1e-12
    c.validate()
    f = ia.LeannConfig.fast()
    f.validate()
    assert f.prune_ratio > 0 and f.m < 30
    a = ia.LeannConfig.accurate()
    a.validate()
    assert a.m > 30 and a.ef_construction > 128
    for kw in ({"m": 0}, {"m0": 16}, {"prune_ratio": 1.5}, {"beam_width": 0},
               {"hub_percentile": 1.5}, {"ef_construction": 3}):
        with pytest.raises(ia.CoreError) as e:
            ia.LeannConfig(**kw).validate()
        assert e.value.kind == "InvalidConfig"
    for s in ia.PruningStrategy:  # leann.rs:1153-1167
        ia.LeannConfig(pruning_strategy=s, prune_ratio=0.3).validate()


def test_index_new_accessors():  # leann.rs:1259-1267
    idx = ia.LeannIndex.with_defaults()
    assert idx.is_empty() and len(idx) == 0 and idx.dimension() is None
    assert idx.is_recompute() and idx.is_compact() and idx.entry_point is None


def test_csr_graph_host_mirror():  # leann.rs:1171-1217
    g = ia.CsrGraph()
    assert g.num_nodes == 0 and g.entry_point is None
    assert g.add_node([], 0) == 0 and g.entry_point == 0
    assert g.add_node([0], 1) == 1 and g.entry_point == 1
    g.add_node([0, 1], 0)
    assert list(g.get_neighbors(2)) == [0, 1] and g.get_neighbors(999) is None
    idx = ia.LeannIndex.from_csr(g, dimension=8)
    assert len(idx) == 3 and idx.dimension() == 8 and idx.entry_point == 1
    assert list(idx.get_neighbors(0)) == [] and list(idx.get_neighbors(1)) == [0]
    assert list(idx.get_neighbors(2)) == [0, 1] and idx.get_neighbors(999) is None
    assert idx.storage_bytes() == g.storage_bytes() > 0


def test_bincode_layout_and_roundtrip():  # leann.rs:1347-1384 + SURVEY 8f-2 layout
    g = ia.CsrGraph()
    g.add_node([1, 2], 0)
    g.add_node([0], 2)
    g.add_node([0, 1], 1)
    cfg = ia.LeannConfig(metric=ia.DistanceMetric.Euclidean, ef_search=77, prune_ratio=0.25,
                         pruning_strategy=ia.PruningStrategy.Local)
    idx = ia.LeannIndex.from_csr(g, cfg, dimension=16)
    b = idx.to_bytes()
    u64 = lambda o: int.from_bytes(b[o:o + 8], "little")
    assert (u64(0), u64(8), u64(16)) == (30, 60, 128)
    assert np.frombuffer(b[24:32], "<f8")[0] == cfg.ml and u64(32) == 16
    assert int.from_bytes(b[40:44], "little") == 1  # metric variant index
    assert u64(44) == 77 and u64(52) == 1
    assert np.frombuffer(b[60:64], "<f4")[0] == np.float32(0.25)
    assert int.from_bytes(b[64:68], "little") == 1 and b[68] == 1  # strategy, high_degree_pruning
    assert b[73] == 1 and b[74] == 1                                # is_compact, is_recompute
    assert u64(75) == 4 and [u64(83 + 8 * i) for i in range(4)] == [0, 2, 3, 5]  # node_offsets
    assert len(b) == 75 + (8 + 32) + (8 + 40) + (8 + 24) + 9 + 8 + 8 + (8 + 24) + 9
    r = ia.LeannIndex.from_bytes(b)
    assert len(r) == 3 and r.dimension() == 16 and r.entry_point == 1
    assert r.config == idx.config and r.to_bytes() == b
    assert list(r.get_neighbors(2)) == [0, 1]
    e = ia.LeannIndex.with_defaults()
    assert ia.LeannIndex.from_bytes(e.to_bytes()).is_empty()
    for cut in (0, 10, 74, 100, len(b) - 1):
        with pytest.raises(ia.CoreError) as ei:
            ia.LeannIndex.from_bytes(b[:cut])
        assert ei.value.kind == "Deserialization"


def test_from_bytes_rejects_wrapping_lengths():
    """Crafted input: num_nodes = 2^64 - 1 makes num_nodes + 1 wrap to 0, which an empty
    node_offsets used to satisfy; a Vec length beyond the buffer must fail, never allocate."""
    import struct
    cfg = ia.LeannIndex.with_defaults().to_bytes()[:75]
    empty_vec = struct.pack("<Q", 0)
    for num_nodes in ((1 << 64) - 1, 1 << 61, 5):
        blob = (cfg + empty_vec + empty_vec + empty_vec + b"\x00" + struct.pack("<Q", 0) +
                struct.pack("<Q", num_nodes) + empty_vec + b"\x00")
        with pytest.raises(ia.CoreError) as ei:
            ia.LeannIndex.from_bytes(blob)
        assert ei.value.kind == "Deserialization"
    blob = cfg + struct.pack("<Q", (1 << 63) + 7) + b"\x00" * 64
    with pytest.raises(ia.CoreError) as ei:
        ia.LeannIndex.from_bytes(blob)
    assert ei.value.kind == "Deserialization"


def test_provider_rules():  # leann.rs:111-120
    with pytest.raises(ia.CoreError) as e:
        ia.InMemoryEmbeddingProvider(np.zeros((0, 4), np.float32))
    assert e.value.kind == "EmptyCollection"
    p = ia.InMemoryEmbeddingProvider(np.ones((3, 4), np.float32))
    assert p.dimension() == 4 and len(p) == 3


def test_search_result_similarity():  # search.rs:311-324
    assert ia.SearchResult(0, 0.0).to_similarity() == 1.0
    assert ia.SearchResult(0, 1.0).to_similarity() == 0.5
    assert abs(ia.SearchResult(0, 9.0).to_similarity() - 0.1) < 1e-7


@pytest.mark.skipif(ia.device_count() > 0, reason="checks the no-GPU error path")
def test_compute_fails_loudly_without_gpu():
    """No CPU fallback: compute entry points report Device errors when no gfx950 is present."""
    g = ia.CsrGraph()
    g.add_node([], 0)
    idx = ia.LeannIndex.from_csr(g, dimension=4)
    with pytest.raises(ia.CoreError) as e:
        idx.upload(0)
    assert e.value.kind == "Device"
    with pytest.raises(ia.CoreError) as e:
        idx.search_with_params([0.1] * 4, 1, 4)
    assert e.value.kind == "Device"
    with pytest.raises(ia.CoreError) as e:
        ia.calculate(ia.DistanceMetric.Cosine, [1, 0], [0, 1])
    assert e.value.kind == "Device"
    # argument validation still follows the reference before any device work
    with pytest.raises(ia.CoreError) as e:
        ia.calculate(ia.DistanceMetric.Cosine, [1, 2], [1, 2, 3])
    assert e.value.kind == "DimensionMismatch" and (e.value.expected, e.value.actual) == (2, 3)


def test_every_compute_family_fails_loudly_without_gpu():
    """The entry points added after the first search path -- encoder, recompute provider, builder,
    HnswGraph facade, distance matrix, brute force, merges, pooling, PQ -- have no CPU path
    either: on a box without a gfx950 each of them reports Device."""
    import numpy as np
    if ia.device_count() > 0:
        pytest.skip("a GPU is present")
    v = np.ones((4, 8), np.float32)
    calls = [
        lambda: ia.LeannIndex.build(v),
        lambda: ia.CandleEmbedder(ia.BertConfig(vocab_size=10, hidden=32, layers=1, heads=2, intermediate=64,
                                                max_position=8, type_vocab=1)),
        lambda: ia.HnswGraph(v, [[[1], [0], [], []]], [0, 0, 0, 0], 0, 0),
        lambda: ia.distance_matrix(ia.DistanceMetric.Cosine, v, v),
        lambda: ia.bruteforce_topk(ia.DistanceMetric.Cosine, v, v, 2),
        lambda: ia.batch_calculate(ia.DistanceMetric.Euclidean, v[0], v),
        lambda: ia.normalize_rows(v),
        lambda: ia.merge_topk(np.zeros((1, 1, 2), np.uint64), np.zeros((1, 1, 2), np.float32), np.ones((1, 1), np.uint32), 2),
        lambda: ia.merge_service(np.zeros((1, 1, 2), np.uint64), np.zeros((1, 1, 2), np.float32), np.ones((1, 1), np.uint32), 2),
        lambda: ia.mean_pool_normalize(np.zeros((1, 2, 8), np.float32), np.ones((1, 2), np.float32)),
        lambda: ia.ProductQuantizer(8, np.zeros((2, 4, 4), np.float32)).encode(v),
    ]
    for i, call in enumerate(calls):
        with pytest.raises(ia.CoreError) as e:
            call()
        assert e.value.kind == "Device", (i, e.value.kind, str(e.value))


def test_shard_exchange_entry_points_fail_loudly_without_gpu():
    """The multi-GPU entry points are host-callable without a card only as far as they need no device:
    RCCL's unique id is handed out (librccl is loaded on first use), a group or a sharded searcher
    is not -- Device, never a host-side stand-in for the exchange."""
    import ctypes as C
    if ia.device_count() > 0:
        pytest.skip("a GPU is present")
    lib = _ffi.lib()
    uid = (C.c_uint8 * 128)()
    assert lib.isl_shard_unique_id(uid) == 0 and any(bytes(uid))
    h = C.c_void_p()
    cb = _ffi.SHARD_ALLGATHER_FN(lambda u, s, r, n: 0)
    assert lib.isl_shard_group_create_host(0, 2, 0, cb, None, C.byref(h)) == 100 and not h.value   # ISL_ERR_DEVICE
    assert lib.isl_shard_group_create(0, 1, 0, uid, C.byref(h)) == 100 and not h.value
    assert lib.isl_shard_group_create_host(0, 2, 5, cb, None, C.byref(h)) == 101                   # rank out of range
    idx = ia.LeannIndex.with_defaults()
    s = C.c_void_p()
    assert lib.isl_sharded_searcher_new(idx._h, None, 0, None, 2, C.byref(s)) == 100 and not s.value
    assert lib.isl_shard_record_bytes(1024, 10) == 1024 * 10 * 12 + 1024 * 4
