import sys, os, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import numpy as np
import islands_amd as ia
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
L = int(sys.argv[2]) if len(sys.argv) > 2 else 64
h = int(sys.argv[3]) if len(sys.argv) > 3 else 768
layers = 6
cfg = dict(vocab_size=30522, hidden=h, layers=layers, heads=12, intermediate=4 * h, max_position=512, type_vocab=2)
w = synth.bert_random_weights(cfg, seed=45, std=0.02)
enc = ia.CandleEmbedder(ia.BertConfig(**{k: cfg[k] for k in cfg}), w)
if os.environ.get("ISL_ENCODER_BF16"):
    enc.set_precision(bf16=True)
rng = np.random.default_rng(44)
ids = rng.integers(1, 30522, (B, L)).astype(np.int64)
mask = np.ones((B, L), np.float32)
flops = B * layers * (24 * h * h * L + 4 * L * L * h)
for it in range(4):
    t = time.time()
    e = enc.embed(ids, None, mask)
    dt = time.time() - t
    print(f"B={B} L={L} h={h}: {dt*1e3:.1f} ms  {flops/dt/1e12:.1f} TFLOP/s  ({B/dt:.0f} sequences/s)", flush=True)
