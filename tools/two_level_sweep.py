"""Operating points of the two-level search (extension, spec Algorithm 2) on the headline rows, in memory.

    python tools/two_level_sweep.py [nodes] [nq] [m ...]

For every PQ shape m, result-set size ef and rerank ratio a: recall@10 against brute force, exact
and approximate distance evaluations per query, kernel time of one launch.  The exact evaluations
per query are what a recompute index would have to ENCODE (BASELINE config 3): the cheapest point
with recall@10 >= 0.95 next to the plain search says what the PQ filter can buy there.  One JSON
line per point; the last line is the summary."""
import json
import os
import sys

sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import torch

import islands_amd as ia
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
ms = [int(a) for a in sys.argv[3:]] or [96, 192]
d, k = 768, 10
x = synth.make_rows(N, d, 0, N, device=dev)
off, nb, entry = synth.build_graph(x)
idx = ia.LeannIndex.from_device_csr(off.data_ptr(), nb.data_ptr(), N, entry, d)
idx.set_embeddings(None, device_ptr=x.data_ptr(), n=N, d=d)
q = synth.make_rows(N, d, 0, nq, device=dev, query=True)
truth, _ = synth.brute_force_topk_native(x, q, k)
oi = torch.zeros((nq, k), dtype=torch.int64, device=dev)
od = torch.zeros((nq, k), device=dev)
oc = torch.zeros(nq, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
points = []


def point(name, m, ef, a, call):
    best = None
    for _ in range(2):
        call()
        st = idx.last_stats()
        best = st if best is None or st["kernel_ms"] < best["kernel_ms"] else best
    r = {"mode": name, "nodes": N, "nq": nq, "pq_m": m, "ef": ef, "a": a, "kernel_ms": round(best["kernel_ms"], 3),
         "exact_evals_per_query": round(best["evals"] / nq, 1), "approx_evals_per_query": round(best["pushes"] / nq, 1),
         "hops_per_query": round(best["expansions"] / nq, 1), "recall_at_10": round(synth.recall_at_k(oi, oc, truth), 4)}
    points.append(r)
    print(json.dumps(r), flush=True)


EFS = tuple(int(e) for e in os.environ.get("SWEEP_EFS", "128,192,256,384,512").split(","))
for ef in EFS[:3]:
    point("plain", 0, ef, None, lambda: idx.search_batch_device(q.data_ptr(), nq, d, k, ef, oi.data_ptr(), od.data_ptr(),
                                                                oc.data_ptr()))
for m in ms:
    cb, codes = synth.train_pq(x, m)
    pq = ia.ProductQuantizer(d, cb.cpu().numpy())
    idx.set_pq_codes(pq, None, device_ptr=codes.data_ptr(), n=N)
    for ef in EFS:
        for a in (0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 1.0):
            try:
                point("two-level", m, ef, a,
                      lambda: idx.search_two_level_batch_device(q.data_ptr(), nq, d, k, ef, a, oi.data_ptr(), od.data_ptr(),
                                                                oc.data_ptr()))
            except ia.CoreError as e:
                print(json.dumps({"mode": "two-level", "pq_m": m, "ef": ef, "a": a, "error": str(e)[:200]}), flush=True)
good = [p for p in points if p["recall_at_10"] >= 0.95]
plain = min((p for p in good if p["mode"] == "plain"), key=lambda p: p["exact_evals_per_query"], default=None)
tl = min((p for p in good if p["mode"] == "two-level"), key=lambda p: p["exact_evals_per_query"], default=None)
print(json.dumps({"summary": "cheapest points with recall@10 >= 0.95 by exact evaluations per query",
                  "plain": plain, "two_level": tl}), flush=True)
