"""BASELINE config 3: LEANN search with on-the-fly recompute through the 6-layer encoder (hidden 768,
12 heads, FFN 3072, L = 64 tokens per node) on the fp32 matrix cores.

    python tools/recompute_bench.py [--nodes 10000000] [--nq 1024] [--ef 128]

Prints one JSON line: queries/s, recall@10, encoder throughput as a fraction of the fp32 MFMA peak,
rounds, the provider's HBM footprint next to what a dense N x d table would take.

Synthetic data (no checkpoint or corpus can be fetched): node i's "text" follows the same tree of
clusters as the headline rows (branching 10, 1000 nodes per leaf) -- 4 tokens for each upper tree
level and 28 for the leaf, all hashed from the node's path prefix at that level, the other 24 noise
tokens -- so that embeddings of nodes that share more of their path are closer; weights ~ N(0, 0.02^2).
The graph comes from the harness of tools/synth.py run on the embeddings the encoder itself
produces (all nodes encoded once, untimed set-up; the recompute index keeps none of them), with
its k-means assignments in float32: a randomly initialised encoder puts a large common component
into every vector (norm of the mean embedding 0.99) and the rows differ from each other only past
bfloat16's resolution -- what stopped the 10 M run of round 1.  Recall is measured against brute
force over the true embeddings."""
import argparse
import ctypes as C
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
from islands_amd import _check, _ffi
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth

MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md)


def log(msg):
    print(f"[recompute_bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def path_tokens(ids: torch.Tensor, L: int, vocab: int, per_leaf: int, seed: int, noise_seed: int):
    """[len(ids), L] int64 token rows: 4 tokens for each of the three upper tree levels and 28 for
    the leaf (a hash of the node's path prefix at that level: leaf // 1000, // 100, // 10, // 1 --
    branching 10), then noise tokens hashed from (node id, noise_seed), so that any id range can be
    produced independently.  Leaf-mates share 40 of 64 tokens, nodes of sibling leaves 12: like the
    headline rows, a query's nearest neighbours are in its own leaf."""
    dev = ids.device
    leaf = ids // per_leaf
    out = torch.empty((ids.numel(), L), dtype=torch.int64, device=dev)
    col = 0
    for li, (div, cnt) in enumerate(((1000, 4), (100, 4), (10, 4), (1, 28))):
        pref = leaf // div
        for j in range(cnt):
            if col >= L:
                break
            hsh = (pref * 1000003 + (li * 8 + j) * 7919 + seed * 104729) % 2147483647
            hsh = (hsh * 48271) % 2147483647
            out[:, col] = 1 + hsh % (vocab - 1)
            col += 1
    rest = L - col
    if rest > 0:
        base = (ids[:, None] * 2654435761 + torch.arange(rest, device=dev)[None, :] * 40503 + noise_seed * 97) % 2147483647
        base = (base * 48271) % 2147483647
        out[:, col:] = 1 + base % (vocab - 1)
    return out


def embed_device(enc, tok_i64: torch.Tensor, out: torch.Tensor):
    B, L = tok_i64.shape
    _check(_ffi.lib().isl_encoder_embed(enc._h, C.c_void_p(tok_i64.data_ptr()), None, None, B, L, int(enc.normalize),
                                        C.c_void_p(out.data_ptr()), ia.MEM_DEVICE, None))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--nq", type=int, default=1024)
    ap.add_argument("--ef", type=int, default=128)
    ap.add_argument("--ef-list", type=str, default="",
                    help="comma-separated ef values searched one after the other over the same set-up (one JSON "
                         "line each), stopping at the first that reaches recall@10 >= 0.95")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--tokens", type=int, default=64)
    ap.add_argument("--cache-rows", type=int, default=1 << 20)
    ap.add_argument("--bf16", action="store_true",
                    help="optional bf16 mode of the encoder's Linear layers (not the reference's arithmetic)")
    ap.add_argument("--setup-bf16", action="store_true",
                    help="encode the set-up embeddings (graph building only) in the bf16 mode; truth and search stay f32")
    ap.add_argument("--two-level", type=float, default=0.0, metavar="RATIO",
                    help="two-level search with a PQ filter (extension): promote this share of the "
                         "approximate queue to exact recomputation")
    ap.add_argument("--pq-m", type=int, default=96)
    ap.add_argument("--check-in-memory", action="store_true",
                    help="also run the batch over the in-memory provider holding the same embeddings and compare bits")
    args = ap.parse_args()
    N, L, h, layers = args.nodes, args.tokens, 768, 6
    dev = torch.device("cuda:0")
    cfg = dict(vocab_size=30522, hidden=h, layers=layers, heads=12, intermediate=3072,
               max_position=512, type_vocab=2)
    enc = ia.CandleEmbedder(ia.BertConfig(**cfg), synth.bert_random_weights(cfg, seed=45, std=0.02))
    if args.bf16:
        enc.set_precision(bf16=True)
    flops_per_node = layers * (24 * h * h * L + 4 * L * L * h)

    # the token table, resident on the device as the provider wants it (u16)
    t0 = time.time()
    tok16 = torch.empty((N, L), dtype=torch.int16, device=dev)
    step = 1 << 18
    for o in range(0, N, step):
        ids = torch.arange(o, min(N, o + step), device=dev)
        tok16[o:o + ids.numel()] = path_tokens(ids, L, cfg["vocab_size"], 1000, 44, 45).to(torch.int16)  # ids < 2^15
    torch.cuda.synchronize()
    log(f"token table {N} x {L} in {time.time() - t0:.1f}s")

    # all embeddings once (set-up): graph building and ground truth
    t0 = time.time()
    x = torch.empty((N, h), dtype=torch.float32, device=dev)
    step = 8192
    for o in range(0, N, step):
        embed_device(enc, tok16[o:o + step].to(torch.int64), x[o:o + step])
        if (o // step) % 128 == 127:
            torch.cuda.synchronize()
            log(f"encoded {o + step} of {N} nodes, {time.time() - t0:.0f}s")
    torch.cuda.synchronize()
    t_all = time.time() - t0
    log(f"all {N} nodes encoded in {t_all:.1f}s ({N * flops_per_node / t_all / 1e12:.1f} TFLOP/s)")

    t0 = time.time()
    common = float(x.mean(0).norm().item())
    # (float32 assignments: a randomly initialised encoder puts a large common component into every
    # vector and the rows differ only past bfloat16's resolution)
    off, nb, entry = synth.build_graph(x, precise=True)
    torch.cuda.synchronize()
    gst = synth.graph_stats(off)
    log(f"graph in {time.time() - t0:.1f}s: {gst} (norm of the mean embedding {common:.4f})")

    # out-of-sample queries: the text of a random node with its noise tokens drawn afresh
    qnode = torch.randint(0, N, (args.nq,), generator=torch.Generator().manual_seed(43)).to(dev)
    qt = path_tokens(qnode, L, cfg["vocab_size"], 1000, 44, 4545)
    q = torch.empty((args.nq, h), dtype=torch.float32, device=dev)
    embed_device(enc, qt, q)
    ti, _ = synth.brute_force_topk(x, q, args.k)
    qh = q.cpu().numpy()

    mem_res = None
    if args.check_in_memory:
        midx = ia.LeannIndex.from_device_csr(off.data_ptr(), nb.data_ptr(), N, entry, h)
        midx.set_embeddings(None, device_ptr=x.data_ptr(), n=N, d=h)
        mem_res = midx.search_batch(qh, args.k, args.ef)
        mem_stats = midx.last_stats()
        del midx
    idx = ia.LeannIndex.from_device_csr(off.data_ptr(), nb.data_ptr(), N, entry, h)
    idx.set_recompute_provider(enc, device_ptr=tok16.data_ptr(), n=N, L=L, keep_rows=False, cache_rows=args.cache_rows)
    pq = None
    if args.two_level > 0:
        cb, codes = synth.train_pq(x, args.pq_m)
        pq = ia.ProductQuantizer(h, cb.cpu().numpy())
        idx.set_pq_codes(pq, None, device_ptr=codes.data_ptr(), n=N)
        del codes
    del x
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    log("graph and index ready, searching")
    tih = ti.cpu().numpy()
    efs = [int(e) for e in args.ef_list.split(",") if e] or [args.ef]
    for ef in efs:
        t0 = time.time()
        if pq is not None:
            ids, dist, cnt = idx.search_two_level_batch(qh, args.k, ef, args.two_level)
        else:
            ids, dist, cnt = idx.search_batch(qh, args.k, ef)
        dt = time.time() - t0
        st = idx.last_stats()
        hit = sum(len(set(ids[i, :cnt[i]].tolist()) & set(tih[i].tolist())) for i in range(args.nq))
        enc_tflops = st["encoded_nodes"] * flops_per_node / dt / 1e12
        mode_label = "bf16 Linear layers, float32 accumulation" if args.bf16 else "float32 MFMA"
        res = {
            "metric": "queries/s, recompute provider (BASELINE config 3)",
            "value": round(args.nq / dt, 2), "unit": "queries/s",
            "config": {"workload": f"{N} nodes x {L} tokens, 6-layer encoder hidden 768 ({mode_label}), "
                                   f"query batch {args.nq}, k={args.k}, ef={ef}, cosine",
                       "search": (f"two-level, rerank ratio {args.two_level}, PQ m={args.pq_m} K=256"
                                  if pq is not None else "LeannIndex::search"),
                       "graph": gst},
            "recall_at_10": round(hit / (args.nq * args.k), 4),
            "seconds": round(dt, 2), "rounds": st["recompute_rounds"],
            "evals": st["evals"], "approx_evals": st["pushes"] if pq is not None else 0,
            "encoded_nodes": st["encoded_nodes"],
            "search_kernel_ms_all_rounds": round(st["kernel_ms"], 1),
            "provider_hbm_bytes": {"row_cache_and_slot_map": idx.recompute_cache_bytes(), "token_table": N * L * 2,
                                   "dense_table_would_be": N * h * 4},
            "roofline": {"bound": "mfma", "achieved": round(enc_tflops, 1),
                         "peak": 2500.0 if args.bf16 else MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(enc_tflops / (2500.0 if args.bf16 else MFMA_F32_PEAK_TFLOPS), 4),
                         "note": "encoder flops of the call / wall time of the call (rounds, gathers and "
                                 "traversal included); layers*(24 h^2 L + 4 L^2 h) flops per node"},
            "encode_all_nodes_seconds": round(t_all, 1),
            "encode_all_tflops": round(N * flops_per_node / t_all / 1e12, 1),
        }
        if mem_res is not None and ef == args.ef:
            res["equals_in_memory_provider"] = bool(
                (mem_res[0] == ids).all() and (mem_res[1].view(np.uint32) == dist.view(np.uint32)).all()
                and (mem_res[2] == cnt).all()
                and all(mem_stats[f] == st[f] for f in ("expansions", "edges", "evals", "pushes")))
        print(json.dumps(res), flush=True)
        log(f"ef={ef}: recall {res['recall_at_10']}, {dt:.1f}s")
        if res["recall_at_10"] >= 0.95:
            break


if __name__ == "__main__":
    main()
