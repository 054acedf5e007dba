"""BASELINE config 5 end to end: 10M x 4096 bf16 rows, query batch 4096 -- the MFMA distance-GEMM
stress (benches/vector_ops.rs batch_calculate as ONE query x row GEMM on the bf16 matrix cores),
PQ distance tables + codes (benches/pq_compression.rs shapes: build_distance_tables,
table_distance), the two-level re-rank search over them and the plain traversal over the bf16 rows.

    python tools/config5_bench.py [--nodes 10000000] [--dim 4096] [--nq 4096]

One JSON line with both rooflines: `mfma` for the distance GEMM (flops / kernel time against the
2.5 PFLOP/s dense bf16 peak), `hbm` for the traversal (SURVEY 8d bytes with s = 2 / wall time
against 8 TB/s).  Rows are generated and kept in bf16 only (82 GB at 10M x 4096; never a 164 GB
float32 copy): the harness of tools/synth.py builds the graph on the bf16 rows, ground truth is
the exact top-k under the library's own bf16 distance GEMM."""
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


def log(msg):
    print(f"[config5 {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=4096)
    ap.add_argument("--nq", type=int, default=4096)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--ef", type=int, default=128)
    ap.add_argument("--pq-m", type=int, default=64)
    ap.add_argument("--ratio", type=float, default=0.3, help="re-rank ratio of the two-level search")
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--depth", type=int, default=6)
    ap.add_argument("--skip-two-level", action="store_true")
    ap.add_argument("--graph-only", action="store_true", help="stop after the graph (harness check)")
    a = ap.parse_args()
    N, d, nq, k, ef = a.nodes, a.dim, a.nq, a.k, a.ef
    dev = torch.device("cuda:0")
    lib = _ffi.lib()

    # ---- rows: bf16 only, generated chunk by chunk
    t0 = time.time()
    x16 = torch.empty((N, d), dtype=torch.bfloat16, device=dev)
    step = synth.CHUNK * 4
    for o in range(0, N, step):
        c = min(step, N - o)
        x16[o:o + c] = synth.make_rows(N, d, o, c, device=dev).to(torch.bfloat16)
    torch.cuda.synchronize()
    log(f"{N} x {d} bf16 rows ({N * d * 2 / 1e9:.1f} GB) in {time.time() - t0:.1f}s")
    q = synth.make_rows(N, d, 0, nq, device=dev, query=True)
    q16 = q.to(torch.bfloat16).contiguous()
    qf = q16.to(torch.float32).contiguous()  # the queries' exact f32 images (what the traversal takes)

    # ---- the dense side: every (query, row) distance as a bf16 GEMM, block by block, + exact top-k
    t0 = time.time()
    block = 65536
    out = torch.empty((nq, block), dtype=torch.float32, device=dev)
    best_d = torch.full((nq, k), float("inf"), device=dev)
    best_i = torch.zeros((nq, k), dtype=torch.int64, device=dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gemm_ms = 0.0
    for o in range(0, N, block):
        c = min(block, N - o)
        torch.cuda.synchronize()
        ev0.record()
        _check(lib.isl_distance_matrix_bf16(0, C.c_void_p(q16.data_ptr()), nq, C.c_void_p(x16[o:o + c].data_ptr()), c, d,
                                            C.c_void_p(out.data_ptr()), ia.MEM_DEVICE, 0, None))
        ev1.record()
        torch.cuda.synchronize()
        gemm_ms += ev0.elapsed_time(ev1)
        dd, ii = torch.topk(out.view(-1)[:nq * c].view(nq, c), k, dim=1, largest=False)  # [nq][c], dense
        cat_d = torch.cat([best_d, dd], 1)
        cat_i = torch.cat([best_i, ii + o], 1)
        sel = torch.topk(cat_d, k, dim=1, largest=False).indices
        best_d, best_i = torch.gather(cat_d, 1, sel), torch.gather(cat_i, 1, sel)
    gemm_flops = 2.0 * nq * N * d
    gemm_tflops = gemm_flops / (gemm_ms * 1e-3) / 1e12
    log(f"distance GEMM over all rows: {gemm_ms:.1f} ms of kernels = {gemm_tflops:.0f} TFLOP/s; "
        f"with top-k {time.time() - t0:.1f}s")
    del out

    # ---- graph (harness) on the bf16 rows
    t0 = time.time()
    # (every harness GEMM in float32: the bf16 GEMMs torch picks for these shapes ended in a GPU memory
    # access fault at d = 4096, at 1M and at 10M rows alike)
    off, nb, entry = synth.build_graph(x16, m0=60, precise=True)
    torch.cuda.synchronize()
    gst = synth.graph_stats(off)
    log(f"graph in {time.time() - t0:.1f}s: {gst}")
    if a.graph_only:
        return
    idx = ia.LeannIndex.from_device_csr(off.data_ptr(), nb.data_ptr(), N, entry, d, ia.LeannConfig.paper_default(), device=0)
    del off, nb
    idx.set_embeddings_bf16(None, device_ptr=x16.view(torch.int16).data_ptr(), n=N, d=d)
    torch.cuda.synchronize()
    log("index resident (its own copy of the bf16 rows)")

    # ---- PQ side: codebooks + codes (harness training), tables in the timed search
    res_tl = None
    if not a.skip_two_level:
        t0 = time.time()
        cb, codes = synth.train_pq(x16, a.pq_m)
        pq = ia.ProductQuantizer(d, cb.cpu().numpy())
        idx.set_pq_codes(pq, None, device_ptr=codes.data_ptr(), n=N)
        del codes
        torch.cuda.synchronize()
        log(f"PQ m={a.pq_m} K=256 trained and {N} rows encoded in {time.time() - t0:.1f}s")
    del x16
    torch.cuda.empty_cache()

    depth = a.depth
    idx.prepare(nq, ef, k, depth)
    outs = [(torch.zeros((nq, k), dtype=torch.int64, device=dev), torch.zeros((nq, k), dtype=torch.float32, device=dev),
             torch.zeros(nq, dtype=torch.int32, device=dev)) for _ in range(depth)]

    def recall(ids, cnt):
        return synth.recall_at_k(ids, cnt, best_i)

    # ---- traversal over the bf16 rows (LeannIndex::search), `depth` batches in flight
    def run(steps):
        agg = {"expansions": 0, "edges": 0, "evals": 0, "queries": 0, "kernel_ms": 0.0, "exact_path": 0}
        pend = []
        for s in range(steps):
            o = outs[s % depth]
            pend.append(idx.search_batch_device_async(qf.data_ptr(), nq, d, k, ef, o[0].data_ptr(), o[1].data_ptr(),
                                                      o[2].data_ptr()))
            if len(pend) >= depth:
                st = idx.wait_stats(pend.pop(0))
                for f in agg:
                    agg[f] += st[f]
        while pend:
            st = idx.wait_stats(pend.pop(0))
            for f in agg:
                agg[f] += st[f]
        return agg

    run(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    agg = run(a.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rec = recall(outs[(a.steps - 1) % depth][0], outs[(a.steps - 1) % depth][2])
    bytes_total = agg["evals"] * d * 2 + 4 * agg["edges"] + 8 * agg["expansions"] + agg["queries"] * (4 * d + 12 * k)
    hbm_gbs = bytes_total / dt / 1e9
    log(f"traversal: {a.steps * nq / dt:.0f} q/s, recall {rec:.4f}, {hbm_gbs:.0f} GB/s algorithmic")

    if not a.skip_two_level:
        o = outs[0]
        idx.search_two_level_batch_device(qf.data_ptr(), nq, d, k, ef, a.ratio, o[0].data_ptr(), o[1].data_ptr(),
                                          o[2].data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            idx.search_two_level_batch_device(qf.data_ptr(), nq, d, k, ef, a.ratio, o[0].data_ptr(), o[1].data_ptr(),
                                              o[2].data_ptr())
        torch.cuda.synchronize()
        dtl = (time.perf_counter() - t0) / reps
        st = idx.last_stats()
        res_tl = {"value": round(nq / dtl, 1), "unit": "queries/s", "ms_per_batch": round(dtl * 1e3, 2),
                  "rerank_ratio": a.ratio, "pq": {"m": a.pq_m, "K": 256, "dsub": d // a.pq_m},
                  "exact_evals_per_query": round(st["evals"] / nq, 1),
                  "approx_evals_per_query": round(st["pushes"] / nq, 1),
                  "recall_at_10": round(recall(o[0], o[2]), 4),
                  "note": "build_distance_tables for the batch + two-level traversal (extension, DESIGN 3.6), one call at a time"}

    print(json.dumps({
        "metric": "BASELINE config 5: 10M x 4096 bf16, query batch 4096 (distance GEMM + PQ re-rank + traversal)",
        "value": round(a.steps * nq / dt, 1), "unit": "queries/s", "recall_at_10": round(rec, 4),
        "config": {"workload": f"{N} x {d} bf16 rows resident in HBM, query batch {nq}, k={k}, ef={ef}, cosine, "
                               f"{depth} batches in flight", "graph": gst,
                   "per_query": {"expansions": round(agg["expansions"] / agg["queries"], 1),
                                 "edges": round(agg["edges"] / agg["queries"], 1),
                                 "evals": round(agg["evals"] / agg["queries"], 1)},
                   "exact_path_queries": agg["exact_path"]},
        "roofline": {"bound": "hbm", "achieved": round(hbm_gbs, 1), "peak": 8000.0, "unit": "GB/s",
                     "frac": round(hbm_gbs / 8000.0, 4), "kernel": "leann_search_fast<2,cosine,bf16 rows>",
                     "per_launch_kernel_ms": round(agg["kernel_ms"] / a.steps, 3)},
        "roofline_mfma": {"bound": "mfma", "achieved": round(gemm_tflops, 1), "peak": 2500.0, "unit": "TFLOP/s",
                          "frac": round(gemm_tflops / 2500.0, 4), "kernel": "gemm_tn_bf16_dma<256x256> (isl_distance_matrix_bf16)",
                          "flops": gemm_flops, "kernel_ms_total": round(gemm_ms, 1),
                          "note": f"every (query, row) cosine distance of the batch: {nq} x {N} x {d}, in blocks of {block} rows"},
        "two_level": res_tl,
    }))


if __name__ == "__main__":
    main()
