"""Throughput of the device graph builder (isl_index_build) in its batched mode.
    python tools/build_perf.py [nodes] [dim] [batch]"""
import os, sys, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import numpy as np, torch
import islands_amd as ia
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
dev = torch.device("cuda:0")
x = synth.make_rows(N, d, 0, N, device=dev).cpu().numpy()
t = time.time()
idx = ia.LeannIndex.build(x, ia.LeannConfig.paper_default(), batch=batch)
dt = time.time() - t
q = synth.make_rows(N, d, 0, 512, device=dev, query=True)
ti, _ = synth.brute_force_topk(torch.from_numpy(x).to(dev), q, 10)
ids, dist, cnt = idx.search_batch(q.cpu().numpy(), 10, 128)
hit = sum(len(set(ids[i, :cnt[i]].tolist()) & set(ti[i].tolist())) for i in range(512))
st = idx.last_stats()
print(f"build N={N} d={d} batch={batch}: {dt:.1f} s = {N/dt:.0f} nodes/s; recall@10(ef=128) = {hit/5120:.3f}; "
      f"evals/q = {st['evals']/512:.0f} hops/q = {st['expansions']/512:.0f}")
