"""Two-level search (extension) against the plain search on the same index, in-memory provider.

    python tools/two_level_perf.py [nodes] [nq] [ef] [m]

Prints one JSON line per rerank ratio: kernel time, exact / approximate evaluations per query,
recall@10 against brute force, and the plain search for comparison.  PQ codebooks come from the
harness's Lloyd iterations (synth.train_pq)."""
import json
import os
import sys

sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import torch

import islands_amd as ia
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
ef = int(sys.argv[3]) if len(sys.argv) > 3 else 128
m = int(sys.argv[4]) if len(sys.argv) > 4 else 96
d, k = 768, 10
x = synth.make_rows(N, d, 0, N, device=dev)
off, nb, entry = synth.build_graph(x)
idx = ia.LeannIndex.from_device_csr(off.data_ptr(), nb.data_ptr(), N, entry, d)
idx.set_embeddings(None, device_ptr=x.data_ptr(), n=N, d=d)
q = synth.make_rows(N, d, 0, nq, device=dev, query=True)
truth, _ = synth.brute_force_topk_native(x, q, k)
cb, codes = synth.train_pq(x, m)
pq = ia.ProductQuantizer(d, cb.cpu().numpy())
idx.set_pq_codes(pq, None, device_ptr=codes.data_ptr(), n=N)
oi = torch.zeros((nq, k), dtype=torch.int64, device=dev)
od = torch.zeros((nq, k), device=dev)
oc = torch.zeros(nq, dtype=torch.int32, device=dev)
torch.cuda.synchronize()


def report(name, call):
    best = None
    for _ in range(3):
        call()
        st = idx.last_stats()
        best = st if best is None or st["kernel_ms"] < best["kernel_ms"] else best
    print(json.dumps({"mode": name, "nodes": N, "nq": nq, "ef": ef, "pq_m": m,
                      "kernel_ms": round(best["kernel_ms"], 3),
                      "queries_per_s_one_launch": round(nq / best["kernel_ms"] * 1e3),
                      "exact_evals_per_query": round(best["evals"] / nq, 1),
                      "approx_evals_per_query": round(best["pushes"] / nq, 1),
                      "hops_per_query": round(best["expansions"] / nq, 1),
                      "recall_at_10": round(synth.recall_at_k(oi, oc, truth), 4)}), flush=True)


report("plain", lambda: idx.search_batch_device(q.data_ptr(), nq, d, k, ef, oi.data_ptr(), od.data_ptr(),
                                                oc.data_ptr()))
for ratio in (0.1, 0.2, 0.3, 0.5, 1.0):
    report(f"two-level a={ratio}",
           lambda: idx.search_two_level_batch_device(q.data_ptr(), nq, d, k, ef, ratio, oi.data_ptr(),
                                                     od.data_ptr(), oc.data_ptr()))
