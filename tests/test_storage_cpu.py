"""CPU: index persistence (src/core/storage.rs) -- IndexMetadata, the META chunk and the
one-file save / load, against a Python restatement of the reference's serde_json + framing and
the reference's own storage tests (storage.rs:229-372)."""
import json
import struct

import numpy as np
import pytest

import islands_amd as ia


def ref_meta_chunk(version, num_vectors, dimension, created_at, updated_at, description):
    """serde_json::to_vec(&IndexMetadata) (compact, declaration order, storage.rs:16-29) inside
    tag || u64 LE length || payload (storage.rs:127-135)."""
    js = json.dumps({"version": version, "num_vectors": num_vectors, "dimension": dimension,
                     "created_at": created_at, "updated_at": updated_at, "description": description},
                    separators=(",", ":"), ensure_ascii=False).encode("utf-8")
    return b"META" + struct.pack("<Q", len(js)) + js


def test_metadata_new():  # storage.rs:229-238
    m = ia.IndexMetadata.new(100, 128, 1_700_000_000)
    assert (m.version, m.num_vectors, m.dimension) == (1, 100, 128)
    assert m.created_at == m.updated_at == 1_700_000_000 and m.description is None


@pytest.mark.parametrize("desc", [None, "test index", 'quote " backslash \\ tab\t newline\n', "ünïcödé ✓"])
def test_meta_chunk_bytes_and_roundtrip(desc):  # storage.rs:240-253, 302-323
    m = ia.IndexMetadata.new(50, 64, 1234567)
    m.description = desc
    chunk = m.to_chunk()
    assert chunk == ref_meta_chunk(1, 50, 64, 1234567, 1234567, desc)
    assert ia.IndexMetadata.from_chunk(chunk) == m


def test_reader_accepts_any_key_order_and_unknown_keys():
    js = b'{ "dimension": 8, "extra": [1, {"a": "}"}], "updated_at": -5, "version": 1,\n "num_vectors": 3, "created_at": 7 }'
    m = ia.IndexMetadata.from_chunk(b"META" + struct.pack("<Q", len(js)) + js)
    assert (m.dimension, m.num_vectors, m.created_at, m.updated_at, m.description) == (8, 3, 7, -5, None)


def test_reader_errors():  # storage.rs:337-361
    with pytest.raises(ia.CoreError) as e:
        ia.IndexMetadata.from_chunk(b"BAAD" + struct.pack("<Q", 8) + b"testdata")
    assert e.value.kind == "Deserialization" and "expected META chunk" in str(e.value)
    with pytest.raises(ia.CoreError) as e:
        ia.IndexMetadata.from_chunk(b"META" + struct.pack("<Q", 100) + b"{}")
    assert e.value.kind == "Io"
    with pytest.raises(ia.CoreError) as e:
        ia.IndexMetadata.from_chunk(b"META" + struct.pack("<Q", 13) + b'{"version":1}')
    assert e.value.kind == "Deserialization"


def test_index_save_load_nested_path(tmp_path):  # storage.rs:268-275 + leann.rs:1347-1364
    g = ia.CsrGraph()
    for i in range(9):
        g.add_node([(i + 1) % 9, (i + 4) % 9], 1 if i == 3 else 0)
    idx = ia.LeannIndex.from_csr(g, ia.LeannConfig.fast(), dimension=24)
    path = tmp_path / "nested" / "dir" / "index.leann"
    meta = ia.IndexMetadata.new(9, 24, 99)
    meta.description = "nine nodes"
    idx.save(str(path), meta)
    raw = path.read_bytes()
    head = ref_meta_chunk(1, 9, 24, 99, 99, "nine nodes")
    body = idx.to_bytes()
    assert raw == head + b"LIDX" + struct.pack("<Q", len(body)) + body
    back, m2 = ia.LeannIndex.load(str(path))
    assert m2 == meta and back.to_bytes() == body and len(back) == 9 and back.dimension() == 24
    assert back.get_neighbors(3).tolist() == [4, 7] and back.entry_point == 3
    idx.save(str(tmp_path / "auto.leann"))  # IndexMetadata::new(len, dimension) now
    _, m3 = ia.LeannIndex.load(str(tmp_path / "auto.leann"))
    assert (m3.version, m3.num_vectors, m3.dimension) == (1, 9, 24) and m3.created_at > 1_600_000_000
    with pytest.raises(ia.CoreError) as e:
        ia.LeannIndex.load(str(tmp_path / "missing.leann"))  # storage.rs:222-227
    assert e.value.kind == "Io"
