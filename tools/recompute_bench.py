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
    ap.add_argument("--two-level-auto", action="store_true",
                    help="choose the two-level operating point: sweep (ratio, ef) over the in-memory provider holding the "
                         "same embeddings (identical traversals, no encoding) and take the point with the fewest exact "
                         "evaluations per query among those with recall@10 >= 0.95; then run it over the recompute provider")
    ap.add_argument("--also-plain", action="store_true", help="with --two-level[-auto]: the plain search over the recompute "
                    "provider too (at the first ef of --ef-list / --ef that reaches recall 0.95 in memory)")
    ap.add_argument("--warm", action="store_true",
                    help="afterwards, with keep_rows = 1 (the row cache survives the call): a first batch, a second batch "
                         "of other queries, and the first batch again")
    ap.add_argument("--quantum-ab", action="store_true",
                    help="run the searches twice: with every miss of a round encoded at once (ISL_RECOMPUTE_QUANTUM=0, "
                         "rounds 1-3's behaviour) and with the rounds' encoder batches in whole waves of GEMM tiles")
    ap.add_argument("--split-ab", action="store_true",
                    help="run the searches with the encoder's passes whole (ISL_ENCODER_SPLIT=0) and as two halves side by "
                         "side on two streams (default), each with and without the whole-tile-wave quantum, twice")
    ap.add_argument("--prefetch-ab", default="", metavar="LIST",
                    help="two-level search: run it under ISL_TL_PREFETCH = each value of the comma list (nodes a parked "
                         "query names beyond its misses; 0 = none), twice, interleaved in one process")
    ap.add_argument("--inflight", type=int, default=0, metavar="CALLS",
                    help="two-level search: answer the --nq queries as CALLS asynchronous device-buffer calls of nq / CALLS "
                         "queries each, all in flight (the library answers the ones that wait for their turn together); "
                         "with ISL_NO_RECOMPUTE_COALESCE=1 they run one after the other")
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

    tih = ti.cpu().numpy()
    flops_note = "encoder flops of the call / wall time of the call (rounds, gathers and traversal included); " \
                 "layers*(24 h^2 L + 4 L^2 h) flops per node"
    mode_label = "bf16 Linear layers, float32 accumulation" if args.bf16 else "float32 MFMA"
    peak = 2500.0 if args.bf16 else MFMA_F32_PEAK_TFLOPS

    def recall_of(ids, cnt, truth=None):
        truth = tih if truth is None else truth
        return sum(len(set(ids[i, :cnt[i]].tolist()) & set(truth[i].tolist())) for i in range(ids.shape[0])) / (ids.shape[0] * args.k)

    # the in-memory provider over the same embeddings: reference answers for the equality check and,
    # with --two-level-auto, the cheap place to look for the operating point (same traversal, no encoding)
    midx = None
    if args.check_in_memory or args.two_level_auto or args.also_plain:
        midx = ia.LeannIndex.from_device_csr(off.data_ptr(), nb.data_ptr(), N, entry, h)
        midx.set_embeddings(None, device_ptr=x.data_ptr(), n=N, d=h)
    pq = None
    ratio = args.two_level
    if args.two_level > 0 or args.two_level_auto:
        t0 = time.time()
        cb, codes = synth.train_pq(x, args.pq_m)
        pq = ia.ProductQuantizer(h, cb.cpu().numpy())
        log(f"PQ m={args.pq_m} trained, {N} rows encoded in {time.time() - t0:.1f}s")
    efs = [int(e) for e in args.ef_list.split(",") if e] or [args.ef]
    ef_tl, sweep = efs[0], None
    if pq is not None and midx is not None:
        midx.set_pq_codes(pq, None, device_ptr=codes.data_ptr(), n=N)
    if args.two_level_auto:
        sweep = []
        for efv in sorted(set(efs + [128, 192, 224, 256, 320, 384])):
            for a_ in (0.01, 0.02, 0.03, 0.05, 0.1, 0.15, 0.2, 0.3, 0.5, 0.7, 1.0):
                ids, dist, cnt = midx.search_two_level_batch(qh, args.k, efv, a_)
                st = midx.last_stats()
                sweep.append({"ef": efv, "ratio": a_, "recall_at_10": round(recall_of(ids, cnt), 4),
                              "exact_evals_per_query": round(st["evals"] / args.nq, 1),
                              "approx_evals_per_query": round(st["pushes"] / args.nq, 1)})
        good = [p_ for p_ in sweep if p_["recall_at_10"] >= 0.95]
        if not good:
            log("no two-level point reaches recall 0.95 on these embeddings; taking ratio 1.0 at the largest ef")
            good = [max(sweep, key=lambda p_: (p_["recall_at_10"]))]
        best = min(good, key=lambda p_: p_["exact_evals_per_query"])
        ratio, ef_tl = best["ratio"], best["ef"]
        log(f"two-level operating point from the in-memory sweep: {best}")
    ef_plain = None
    if args.also_plain or (pq is None):
        ef_plain = efs[-1]
        if midx is not None:
            for efv in efs:
                ids, dist, cnt = midx.search_batch(qh, args.k, efv)
                if recall_of(ids, cnt) >= 0.95:
                    ef_plain = efv
                    break

    idx = ia.LeannIndex.from_device_csr(off.data_ptr(), nb.data_ptr(), N, entry, h)
    idx.set_recompute_provider(enc, device_ptr=tok16.data_ptr(), n=N, L=L, keep_rows=False, cache_rows=args.cache_rows)
    if pq is not None:
        idx.set_pq_codes(pq, None, device_ptr=codes.data_ptr(), n=N)
    torch.cuda.synchronize()
    log("graph and index ready, searching")

    def run(label, queries, truth, ef, tl_ratio, check=False, extra=None):
        t0 = time.time()
        if tl_ratio:
            ids, dist, cnt = idx.search_two_level_batch(queries, args.k, ef, tl_ratio)
        else:
            ids, dist, cnt = idx.search_batch(queries, args.k, ef)
        dt = time.time() - t0
        st = idx.last_stats()
        nq_ = queries.shape[0]
        enc_tflops = st["encoded_nodes"] * flops_per_node / dt / 1e12
        res = {
            "metric": "queries/s, recompute provider (BASELINE config 3)", "run": label,
            "value": round(nq_ / dt, 2), "unit": "queries/s",
            "config": {"workload": f"{N} nodes x {L} tokens, 6-layer encoder hidden 768 ({mode_label}), "
                                   f"query batch {nq_}, k={args.k}, ef={ef}, cosine",
                       "search": (f"two-level, rerank ratio {tl_ratio}, PQ m={args.pq_m} K=256" if tl_ratio else "LeannIndex::search"),
                       "graph": gst},
            "recall_at_10": round(recall_of(ids, cnt, truth), 4),
            "seconds": round(dt, 2), "rounds": st["recompute_rounds"],
            "evals": st["evals"], "approx_evals": st["pushes"] if tl_ratio else 0,
            "encoded_nodes": st["encoded_nodes"], "encoded_nodes_per_query": round(st["encoded_nodes"] / nq_, 1),
            "search_kernel_ms_all_rounds": round(st["kernel_ms"], 1),
            "provider_hbm_bytes": {"row_cache_and_slot_map": idx.recompute_cache_bytes(), "token_table": N * L * 2,
                                   "pq_codes": (N * args.pq_m * 2 if pq is not None else 0), "dense_table_would_be": N * h * 4},
            "roofline": {"bound": "mfma", "achieved": round(enc_tflops, 1), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(enc_tflops / peak, 4), "note": flops_note},
            "encode_all_nodes_seconds": round(t_all, 1),
            "encode_all_tflops": round(N * flops_per_node / t_all / 1e12, 1),
        }
        if check and midx is not None:
            if tl_ratio:
                m_ids, m_dist, m_cnt = midx.search_two_level_batch(queries, args.k, ef, tl_ratio)
            else:
                m_ids, m_dist, m_cnt = midx.search_batch(queries, args.k, ef)
            ms = midx.last_stats()
            res["equals_in_memory_provider"] = bool(
                (m_ids == ids).all() and (m_dist.view(np.uint32) == dist.view(np.uint32)).all() and (m_cnt == cnt).all()
                and all(ms[f] == st[f] for f in ("expansions", "edges", "evals", "pushes")))
        if extra:
            res.update(extra)
        print(json.dumps(res), flush=True)
        log(f"{label}: ef={ef} ratio={tl_ratio}: recall {res['recall_at_10']}, {dt:.1f}s, {res['encoded_nodes_per_query']} nodes/query")
        return res

    if args.inflight > 1 and pq is not None:
        per = args.nq // args.inflight
        outs = (torch.zeros((args.nq, args.k), dtype=torch.int64, device=dev), torch.zeros((args.nq, args.k), dtype=torch.float32, device=dev),
                torch.zeros(args.nq, dtype=torch.int32, device=dev))
        torch.cuda.synchronize()
        t0 = time.time()
        toks = [idx.search_two_level_batch_device_async(q[i * per:(i + 1) * per].data_ptr(), per, h, args.k, ef_tl, ratio,
                                                        outs[0][i * per:(i + 1) * per].data_ptr(), outs[1][i * per:(i + 1) * per].data_ptr(),
                                                        outs[2][i * per:(i + 1) * per].data_ptr()) for i in range(args.inflight)]
        sts = [idx.wait_stats(t) for t in toks]
        torch.cuda.synchronize()
        dt = time.time() - t0
        nqa = per * args.inflight
        ids = outs[0][:nqa].cpu().numpy().astype(np.uint64)
        cnt = outs[2][:nqa].cpu().numpy().astype(np.uint32)
        dist = outs[1][:nqa].cpu().numpy()
        groups = {}
        for st in sts:  # calls answered together report the same rounds and encoded nodes
            groups.setdefault((st["recompute_rounds"], st["encoded_nodes"]), 0)
            groups[(st["recompute_rounds"], st["encoded_nodes"])] += 1
        encoded = sum(k_[1] for k_ in groups)
        res = {"metric": "queries/s, recompute provider (BASELINE config 3)",
               "run": f"two_level, {args.inflight} asynchronous calls of {per} queries in flight",
               "value": round(nqa / dt, 2), "unit": "queries/s", "seconds": round(dt, 2),
               "recall_at_10": round(recall_of(ids, cnt, tih[:nqa]), 4),
               "calls_answered_together": sorted(groups.values(), reverse=True),
               "encoded_nodes": encoded, "encoded_nodes_per_query": round(encoded / nqa, 1),
               "roofline": {"bound": "mfma", "achieved": round(encoded * flops_per_node / dt / 1e12, 1), "peak": peak, "unit": "TFLOP/s",
                            "frac": round(encoded * flops_per_node / dt / 1e12 / peak, 4), "note": flops_note},
               "config": {"search": f"two-level, rerank ratio {ratio}, PQ m={args.pq_m} K=256, ef {ef_tl}", "nodes": N}}
        if midx is not None:
            m_ids, m_dist, m_cnt = midx.search_two_level_batch(qh[:nqa], args.k, ef_tl, ratio)
            res["equals_in_memory_provider"] = bool((m_ids == ids).all() and (m_dist.view(np.uint32) == dist.view(np.uint32)).all()
                                                    and (m_cnt == cnt).all())
        print(json.dumps(res), flush=True)
        log(f"{res['run']}: {res['value']} queries/s, recall {res['recall_at_10']}, groups {res['calls_answered_together']}")
        return
    if args.prefetch_ab and pq is not None:
        import os
        for rep in range(2):
            for v in args.prefetch_ab.split(","):
                os.environ["ISL_TL_PREFETCH"] = v
                run(f"two_level, {v} nodes named ahead per parked query", qh, tih, ef_tl, ratio, check=True)
        os.environ.pop("ISL_TL_PREFETCH", None)
        return
    if args.split_ab:
        import os
        for rep in range(2):
            for split, quantum in (("0", None), ("256", None), ("256", "0"), ("0", "0")):
                for var, val in (("ISL_ENCODER_SPLIT", split), ("ISL_RECOMPUTE_QUANTUM", quantum)):
                    if val is None:
                        os.environ.pop(var, None)
                    else:
                        os.environ[var] = val
                label = (f"encoder passes {'whole' if split == '0' else 'as two halves side by side'}, "
                         f"batches {'every miss at once' if quantum == '0' else 'in whole tile waves'}")
                if pq is not None:
                    run(f"two_level, {label}", qh, tih, ef_tl, ratio, check=True)
                else:
                    run(f"plain, {label}", qh, tih, efs[0], 0.0, check=args.check_in_memory)
        os.environ.pop("ISL_ENCODER_SPLIT", None)
        os.environ.pop("ISL_RECOMPUTE_QUANTUM", None)
        return
    if args.quantum_ab:
        import os
        for label, env in (("every miss at once", "0"), ("whole tile waves", None), ("every miss at once", "0"),
                           ("whole tile waves", None)):
            if env is None:
                os.environ.pop("ISL_RECOMPUTE_QUANTUM", None)
            else:
                os.environ["ISL_RECOMPUTE_QUANTUM"] = env
            if pq is not None:
                run(f"two_level, encoder batches: {label}", qh, tih, ef_tl, ratio, check=True)
            else:
                run(f"plain, encoder batches: {label}", qh, tih, efs[0], 0.0, check=args.check_in_memory)
        os.environ.pop("ISL_RECOMPUTE_QUANTUM", None)
        return
    if pq is None:
        for ef in efs:  # round 2's behaviour: the ef values in turn, stopping at the first that reaches 0.95
            r = run("plain", qh, tih, ef, 0.0, check=args.check_in_memory and ef == args.ef)
            if r["recall_at_10"] >= 0.95:
                break
    else:
        if args.also_plain:
            run("plain", qh, tih, ef_plain, 0.0, check=True)
        run("two_level", qh, tih, ef_tl, ratio, check=True, extra={"in_memory_sweep": sweep} if sweep else None)
    if args.warm:
        # the row cache as an embedding cache that survives calls (keep_rows = 1): a first batch from an empty
        # cache, a second batch of other queries, the first batch again
        idx.set_recompute_provider(enc, device_ptr=tok16.data_ptr(), n=N, L=L, keep_rows=True, cache_rows=args.cache_rows)
        if pq is not None:
            idx.set_pq_codes(pq, None, device_ptr=codes.data_ptr(), n=N)
        qnode2 = torch.randint(0, N, (args.nq,), generator=torch.Generator().manual_seed(47)).to(dev)
        q2 = torch.empty((args.nq, h), dtype=torch.float32, device=dev)
        embed_device(enc, path_tokens(qnode2, L, cfg["vocab_size"], 1000, 44, 4747), q2)
        t2 = synth.brute_force_topk(x, q2, args.k)[0].cpu().numpy()
        q2h = q2.cpu().numpy()
        ef_w, r_w = (ef_tl, ratio) if pq is not None else (ef_plain, 0.0)
        run("keep_rows: first batch, empty cache", qh, tih, ef_w, r_w)
        run("keep_rows: second batch, other queries", q2h, t2, ef_w, r_w, check=True)
        run("keep_rows: first batch again", qh, tih, ef_w, r_w, check=True)


if __name__ == "__main__":
    main()
