"""BASELINE config 3 at a reduced node count: LEANN search with on-the-fly recompute through
the 6-layer encoder (hidden 768, 12 heads, FFN 3072, L = 64 tokens per node), float32 MFMA.

    python tools/recompute_bench.py [--nodes 1000000] [--nq 256] [--ef 128]

Prints one JSON line: queries/s, encoder throughput and its fraction of the fp32 MFMA peak.
Synthetic data: node i's text = an 8-token topic prefix (topic = i // 1000, so neighbouring ids
share a topic) + 56 noise tokens; weights ~ N(0, 0.02^2).  The graph is
built by the harness of islands_amd/synth.py on the embeddings the encoder itself produces."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

import numpy as np
import torch

import islands_amd as ia
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth

MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--nq", type=int, default=256)
    ap.add_argument("--ef", type=int, default=128)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--tokens", type=int, default=64)
    ap.add_argument("--bf16", action="store_true",
                    help="optional bf16 mode of the encoder's Linear layers (not the reference's arithmetic)")
    ap.add_argument("--two-level", type=float, default=0.0, metavar="RATIO",
                    help="two-level search with a PQ filter (extension): promote this share of the "
                         "approximate queue to exact recomputation")
    ap.add_argument("--pq-m", type=int, default=96)
    args = ap.parse_args()
    N, L, h, layers = args.nodes, args.tokens, 768, 6
    cfg = dict(vocab_size=30522, hidden=h, layers=layers, heads=12, intermediate=3072,
               max_position=512, type_vocab=2)
    enc = ia.CandleEmbedder(ia.BertConfig(**cfg), synth.bert_random_weights(cfg, seed=45, std=0.02))
    if args.bf16:
        enc.set_precision(bf16=True)
    rng = np.random.default_rng(44)
    topics = rng.integers(1, cfg["vocab_size"], ((N + 999) // 1000, 8)).astype(np.uint16)
    tok = rng.integers(1, cfg["vocab_size"], (N, L)).astype(np.uint16)
    tok[:, :8] = topics[np.arange(N) // 1000]
    flops_per_node = layers * (24 * h * h * L + 4 * L * L * h)

    # all embeddings once, to build the graph (and as the in-memory twin for the recall check)
    t0 = time.time()
    emb = np.empty((N, h), np.float32)
    step = 2048
    for o in range(0, N, step):
        emb[o:o + step] = enc.embed(tok[o:o + step].astype(np.int64))
        if (o // step) % 256 == 255:
            print(f"[recompute_bench] encoded {o + step} of {N} nodes, {time.time() - t0:.0f}s",
                  file=sys.stderr, flush=True)
    t_all = time.time() - t0
    dev = torch.device("cuda:0")
    x = torch.from_numpy(emb).to(dev)
    off, nb, entry = synth.build_graph(x)
    qrng = np.random.default_rng(43)
    qtok = rng.integers(1, cfg["vocab_size"], (args.nq, L)).astype(np.uint16)
    qtok[:, :8] = topics[qrng.integers(0, topics.shape[0], args.nq)]
    q = enc.embed(qtok.astype(np.int64))
    ti, _ = synth.brute_force_topk(x, torch.from_numpy(q).to(dev), args.k)

    idx = ia.LeannIndex.from_device_csr(off.data_ptr(), nb.data_ptr(), N, entry, h)
    idx.set_recompute_provider(enc, tok, None, keep_rows=False)
    pq = None
    if args.two_level > 0:
        cb, codes = synth.train_pq(x, args.pq_m)
        pq = ia.ProductQuantizer(h, cb.cpu().numpy())
        idx.set_pq_codes(pq, None, device_ptr=codes.data_ptr(), n=N)
        del codes
    del x
    torch.cuda.synchronize()
    print("[recompute_bench] graph and index ready, searching", file=sys.stderr, flush=True)
    t0 = time.time()
    if pq is not None:
        ids, dist, cnt = idx.search_two_level_batch(q, args.k, args.ef, args.two_level)
    else:
        ids, dist, cnt = idx.search_batch(q, args.k, args.ef)
    dt = time.time() - t0
    st = idx.last_stats()
    hit = sum(len(set(ids[i, :cnt[i]].tolist()) & set(ti[i].tolist())) for i in range(args.nq))
    enc_tflops = st["encoded_nodes"] * flops_per_node / dt / 1e12
    mode_label = "bf16 Linear layers, float32 accumulation" if args.bf16 else "float32 MFMA"
    print(json.dumps({
        "metric": "queries/s, recompute provider (BASELINE config 3 at reduced N)",
        "value": round(args.nq / dt, 2), "unit": "queries/s",
        "config": {"workload": f"{N} nodes x {L} tokens, 6-layer encoder hidden 768 ({mode_label}), "
                               f"query batch {args.nq}, k={args.k}, ef={args.ef}, cosine",
                   "search": (f"two-level, rerank ratio {args.two_level}, PQ m={args.pq_m} K=256"
                              if pq is not None else "LeannIndex::search")},
        "recall_at_10": round(hit / (args.nq * args.k), 4),
        "seconds": round(dt, 2), "rounds": st["recompute_rounds"],
        "evals": st["evals"], "approx_evals": st["pushes"] if pq is not None else 0,
        "encoded_nodes": st["encoded_nodes"],
        "search_kernel_ms_all_rounds": round(st["kernel_ms"], 1),
        "roofline": {"bound": "mfma", "achieved": round(enc_tflops, 1),
                     "peak": 2500.0 if args.bf16 else MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(enc_tflops / (2500.0 if args.bf16 else MFMA_F32_PEAK_TFLOPS), 4),
                     "note": "encoder flops of the call / wall time of the call (rounds, gathers and "
                             "traversal included); layers*(24 h^2 L + 4 L^2 h) flops per node"},
        "encode_all_nodes_seconds": round(t_all, 1),
        "encode_all_tflops": round(N * flops_per_node / t_all / 1e12, 1),
    }))


if __name__ == "__main__":
    main()
