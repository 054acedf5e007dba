"""Generates tests/golden/bert_tiny.npz: inputs, weights and outputs of HuggingFace
transformers' BertModel (float32, CPU, eval mode) for a tiny configuration.  candle-transformers'
BertModel -- the model islands' CandleEmbedder runs (src/core/embedding/candle_provider.rs:284,
:429-432; Cargo.lock:1113-1114, not vendored) -- is a port of this implementation, so the
fixture pins the encoder oracle (oracle/bert_ref.py) and the HIP encoder to the published
algorithm.  Run once, here: `python tests/golden/make_bert_golden.py`; the .npz is committed."""
import os
import sys

import numpy as np
import torch
from transformers import BertConfig, BertModel

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bert_ref  # noqa: E402  (weight naming and the padding rule only)

CFG = dict(vocab_size=97, hidden=64, layers=2, heads=4, intermediate=160, max_position=40,
           type_vocab=2, layer_norm_eps=1e-12, gelu_tanh=False)


def main():
    torch.manual_seed(0)
    hf = BertConfig(vocab_size=CFG["vocab_size"], hidden_size=CFG["hidden"],
                    num_hidden_layers=CFG["layers"], num_attention_heads=CFG["heads"],
                    intermediate_size=CFG["intermediate"], max_position_embeddings=CFG["max_position"],
                    type_vocab_size=CFG["type_vocab"], layer_norm_eps=CFG["layer_norm_eps"],
                    hidden_act="gelu", attn_implementation="eager")
    model = BertModel(hf, add_pooling_layer=False).eval()
    w = bert_ref.random_weights(CFG, seed=45, std=0.25)  # large std: attention far from uniform
    sd = model.state_dict()
    for name in bert_ref.weight_names(CFG["layers"]):
        sd[name].copy_(torch.from_numpy(w[name]))
    rng = np.random.default_rng(44)
    lens = [23, 7, 1, 16, 23]
    seqs = [rng.integers(1, CFG["vocab_size"], n).tolist() for n in lens]
    ids, tt, mask = bert_ref.pad_batch(seqs)
    tt[0, 10:23] = 1  # second segment in one row
    with torch.no_grad():
        hid = model(input_ids=torch.from_numpy(ids), token_type_ids=torch.from_numpy(tt),
                    attention_mask=torch.from_numpy(mask.astype(np.int64))).last_hidden_state.numpy()
    out = {"input_ids": ids, "token_type_ids": tt, "attention_mask": mask, "hidden": hid.astype(np.float32)}
    out.update({"w::" + k: v for k, v in w.items()})
    out.update({"cfg::" + k: np.asarray(v) for k, v in CFG.items()})
    path = os.path.join(ROOT, "tests", "golden", "bert_tiny.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; hidden", hid.shape)


if __name__ == "__main__":
    main()
