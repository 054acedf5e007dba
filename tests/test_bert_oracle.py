"""CPU: the encoder oracle (oracle/bert_ref.py) against the golden fixture generated from
HuggingFace transformers' BertModel (tests/golden/make_bert_golden.py)."""
import os

import numpy as np

import bert_ref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bert_tiny.npz")


def load_golden():
    z = np.load(GOLD)
    cfg = {k[5:]: z[k].item() for k in z.files if k.startswith("cfg::")}
    w = {k[3:]: z[k] for k in z.files if k.startswith("w::")}
    return cfg, w, z


def test_oracle_matches_hf_bert_fixture():
    cfg, w, z = load_golden()
    hid = bert_ref.bert_forward(cfg, w, z["input_ids"], z["token_type_ids"], z["attention_mask"])
    want = z["hidden"]
    m = z["attention_mask"].astype(bool)
    # float32 with different summation orders: 2e-5 absolute on O(1) activations; padded rows
    # (mask 0) are computed too and must agree as well
    assert hid.shape == want.shape
    assert np.abs(hid - want).max() < 2e-5, np.abs(hid - want).max()
    assert np.abs(hid[m] - want[m]).max() < 2e-5


def test_pooling_tensor_form_matches_sequential_restatement(orc):
    cfg, w, z = load_golden()
    hid, mask = z["hidden"], z["attention_mask"]
    a = bert_ref.mean_pool_normalize(hid, mask, True)
    b = orc.mean_pool_normalize(hid, mask, True)
    assert np.abs(a - b).max() < 1e-6
    assert np.all(np.abs(np.linalg.norm(a, axis=1) - 1) < 1e-5)


def test_padding_rule():  # candle_provider.rs:385-402
    ids, tt, mask = bert_ref.pad_batch([[5, 6, 7], [9]])
    assert ids.tolist() == [[5, 6, 7], [9, 0, 0]] and mask.tolist() == [[1, 1, 1], [1, 0, 0]]
    assert tt.tolist() == [[0, 0, 0], [0, 0, 0]]
    assert len(bert_ref.weight_names(6)) == 5 + 6 * 16
