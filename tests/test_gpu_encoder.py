"""GPU: the recompute encoder (isl_encoder_*) against the golden fixture from HuggingFace's
BertModel and against the numpy oracle (oracle/bert_ref.py) on configurations the fixture does
not cover.  Float32 everywhere; tolerances are written next to each assertion."""
import os

import numpy as np
import pytest

import bert_ref
import islands_amd as ia

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bert_tiny.npz")


def to_cfg(c):
    return ia.BertConfig(vocab_size=c["vocab_size"], hidden=c["hidden"], layers=c["layers"],
                         heads=c["heads"], intermediate=c["intermediate"],
                         max_position=c["max_position"], type_vocab=c["type_vocab"],
                         layer_norm_eps=c.get("layer_norm_eps", 1e-12),
                         gelu_tanh=bool(c.get("gelu_tanh", False)))


def test_matches_hf_golden_fixture():
    z = np.load(GOLD)
    cfg = {k[5:]: z[k].item() for k in z.files if k.startswith("cfg::")}
    w = {k[3:]: z[k] for k in z.files if k.startswith("w::")}
    enc = ia.CandleEmbedder(to_cfg(cfg), w)
    hid = enc.forward(z["input_ids"], z["token_type_ids"], z["attention_mask"])
    # O(1) activations after LayerNorm; 3e-5 absolute covers float32 reassociation over K <= 160
    assert np.abs(hid - z["hidden"]).max() < 3e-5, np.abs(hid - z["hidden"]).max()
    emb = enc.embed(z["input_ids"], z["token_type_ids"], z["attention_mask"])
    want = bert_ref.mean_pool_normalize(z["hidden"], z["attention_mask"], True)
    assert np.abs(emb - want).max() < 1e-5
    assert np.all(np.abs(np.linalg.norm(emb, axis=1) - 1) < 1e-5)


@pytest.mark.parametrize("name,cfg,B,L", [
    ("minilm-l6", dict(vocab_size=500, hidden=384, layers=6, heads=12, intermediate=1536,
                       max_position=128, type_vocab=2), 5, 70),          # MiniLM-L6 shape, dh = 32
    ("config3", dict(vocab_size=300, hidden=768, layers=2, heads=12, intermediate=3072,
                     max_position=64, type_vocab=2), 3, 64),             # BASELINE config 3 shape, dh = 64
    ("tanh", dict(vocab_size=50, hidden=32, layers=1, heads=2, intermediate=44,
                  max_position=16, type_vocab=1, gelu_tanh=True), 2, 9),  # dh = 16, ragged tile edges
])
def test_matches_numpy_oracle(name, cfg, B, L):
    w = bert_ref.random_weights(cfg, seed=45, std=0.08)
    rng = np.random.default_rng(44)
    lens = rng.integers(1, L + 1, B)
    lens[0] = L
    ids, tt, mask = bert_ref.pad_batch([rng.integers(1, cfg["vocab_size"], n).tolist() for n in lens])
    enc = ia.CandleEmbedder(to_cfg(cfg), w, normalize=True)
    hid = enc.forward(ids, tt, mask)
    want = bert_ref.bert_forward(cfg, w, ids, tt, mask)
    err = np.abs(hid - want).max()
    assert err < 1e-4, (name, err)  # K up to 3072 in float32
    emb = enc.embed(ids, tt, mask)
    ewant = bert_ref.mean_pool_normalize(want, mask, True)
    assert np.abs(emb - ewant).max() < 2e-5, (name, np.abs(emb - ewant).max())


def test_embedding_independent_of_batch_composition():
    """A node's embedding must not depend on what it is batched with (the recompute provider
    re-encodes nodes in whatever batch a hop produces)."""
    cfg = dict(vocab_size=200, hidden=128, layers=2, heads=4, intermediate=256, max_position=32, type_vocab=2)
    w = bert_ref.random_weights(cfg, seed=3, std=0.1)
    enc = ia.CandleEmbedder(to_cfg(cfg), w)
    rng = np.random.default_rng(1)
    seqs = [rng.integers(1, 200, 20).tolist() for _ in range(9)]
    ids, tt, mask = bert_ref.pad_batch(seqs)
    all_at_once = enc.embed(ids, tt, mask)
    for i in (0, 4, 8):
        alone = enc.embed(ids[i:i + 1], tt[i:i + 1], mask[i:i + 1])
        assert alone.view(np.uint32).tolist() == all_at_once[i:i + 1].view(np.uint32).tolist()


def test_encoder_errors():
    cfg = ia.BertConfig(vocab_size=10, hidden=32, layers=1, heads=2, intermediate=64, max_position=8, type_vocab=1)
    enc = ia.CandleEmbedder(cfg)
    with pytest.raises(ia.CoreError) as e:
        enc.set_weight("encoder.layer.0.output.dense.bias", np.zeros(7, np.float32))
    assert e.value.kind == "DimensionMismatch" and (e.value.expected, e.value.actual) == (32, 7)
    with pytest.raises(ia.CoreError) as e:
        enc.embed(np.full((1, 4), 10, np.int64))  # id == vocab_size
    assert e.value.kind == "EmbeddingError"
    with pytest.raises(ia.CoreError) as e:
        enc.embed(np.zeros((1, 9), np.int64))  # longer than max_position
    assert e.value.kind == "EmbeddingError"
    with pytest.raises(ia.CoreError) as e:
        ia.CandleEmbedder(ia.BertConfig(hidden=30, heads=4))
    assert e.value.kind == "InvalidConfig"
    assert enc.embed(np.zeros((0, 4), np.int64)).shape == (0, 32)
