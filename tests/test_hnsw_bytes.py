"""HnswGraph::from_bytes (hnsw.rs:511-514) through isl_hnsw_from_bytes.  The bincode image is
written here by a restatement of bincode 1.x's default encoding of the derive(Serialize) structs
(hnsw.rs:15-28, 90-99, 150-164) -- the reference pins no bytes (its tests round-trip only) and
HashMap order is arbitrary, so the writer shuffles the node entries."""
import struct

import numpy as np
import pytest

import islands_amd as ia
from _data import random_levels, uniform_vectors


def hnsw_to_bincode(vectors, layers, levels, entry_point, max_level, m=16, m0=32, ef_construction=200,
                    metric=0, order=None, dimension=-1, next_id=None):
    n = len(levels)
    out = [struct.pack("<QQQdIQ", m, m0, ef_construction, 1.0 / np.log(m), metric, 16)]
    out.append(struct.pack("<Q", n))
    for i in (range(n) if order is None else order):
        v = np.ascontiguousarray(vectors[i], dtype="<f4")
        out.append(struct.pack("<QQQ", i, i, v.size) + v.tobytes())
        conns = [layers[L][i] for L in range(int(levels[i]) + 1)]
        out.append(struct.pack("<Q", len(conns)))
        for c in conns:
            out.append(struct.pack("<Q", len(c)) + np.asarray(c, dtype="<u8").tobytes())
        out.append(struct.pack("<Q", int(levels[i])))
    out.append(b"\x00" if entry_point is None else b"\x01" + struct.pack("<Q", entry_point))
    out.append(struct.pack("<Q", max_level))
    d = (vectors.shape[1] if n else None) if dimension == -1 else dimension
    out.append(b"\x00" if d is None else b"\x01" + struct.pack("<Q", d))
    out.append(struct.pack("<Q", n if next_id is None else next_id))
    return b"".join(out)


def test_from_bytes_rejects_bad_input():
    v = uniform_vectors(3, 4, 1)
    layers = [[[1], [0, 2], [1]]]
    good = hnsw_to_bincode(v, layers, [0, 0, 0], 0, 0)
    for cut in (0, 7, 40, len(good) - 1):
        with pytest.raises(ia.CoreError) as e:
            ia.HnswGraph.from_bytes(good[:cut])
        assert e.value.kind == "Deserialization"
    with pytest.raises(ia.CoreError) as e:
        ia.HnswGraph.from_bytes(good + b"\x00")
    assert e.value.kind == "Deserialization"
    bad_metric = bytearray(good)
    bad_metric[32:36] = struct.pack("<I", 9)
    with pytest.raises(ia.CoreError) as e:
        ia.HnswGraph.from_bytes(bytes(bad_metric))
    assert e.value.kind == "Deserialization"
    bad_key = bytearray(good)
    bad_key[52:60] = struct.pack("<Q", 2)          # first entry: key 2, id 0
    with pytest.raises(ia.CoreError) as e:
        ia.HnswGraph.from_bytes(bytes(bad_key))
    assert e.value.kind == "Deserialization"
    # lengths that wrap when multiplied: vlen = 2^62 + 1 makes vlen * 4 == 4 and, with
    # n = 4 nodes, n * vlen == 4 in 64 bits; a node count of 2^61 makes every size check wrap
    v4 = uniform_vectors(4, 1, 2)
    wrap = bytearray(hnsw_to_bincode(v4, [[[1], [0], [3], [2]]], [0, 0, 0, 0], 0, 0))
    wrap[68:76] = struct.pack("<Q", (1 << 62) + 1)   # vector length of the first entry
    with pytest.raises(ia.CoreError) as e:
        ia.HnswGraph.from_bytes(bytes(wrap))
    assert e.value.kind == "Deserialization"
    huge_n = bytearray(good)
    huge_n[44:52] = struct.pack("<Q", 1 << 61)
    with pytest.raises(ia.CoreError) as e:
        ia.HnswGraph.from_bytes(bytes(huge_n))
    assert e.value.kind == "Deserialization"
    # an empty graph needs no device
    empty = hnsw_to_bincode(np.zeros((0, 0), np.float32), [[]], [], None, 0, dimension=None)
    g = ia.HnswGraph.from_bytes(empty)
    assert len(g) == 0


@pytest.mark.gpu
def test_from_bytes_searches_like_the_oracle(orc):
    import test_gpu_hnsw as th
    n, d = 500, 24
    v, h, _ = th.build(orc, n, d, 17, m=8, m0=16, ef_construction=60, metric=ia.DistanceMetric.Euclidean)
    layers = [[(h.neighbors(i, L) or []) for i in range(n)] for L in range(h.max_level + 1)]
    levels = [h.level(i) for i in range(n)]
    order = np.random.default_rng(3).permutation(n).tolist()   # HashMap order is arbitrary
    data = hnsw_to_bincode(v, layers, levels, h.entry_point, h.max_level, m=8, m0=16,
                           ef_construction=60, metric=1, order=order)
    g = ia.HnswGraph.from_bytes(data)
    assert len(g) == n
    th.assert_same(h, g, uniform_vectors(30, d, 5), 10, 40)
    th.assert_same(h, g, v[:10], 3, 3)
