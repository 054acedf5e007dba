import sys, os
sys.path[:0] = ['.', 'oracle', 'tests']
import numpy as np, torch
import islands_amd as ia
from islands_amd import synth
dev = torch.device("cuda:0")
N, d, nq = int(sys.argv[1]), 768, 1024
x = synth.make_rows(N, d, 0, N, device=dev)
off, nb, entry = synth.build_graph(x)
idx = ia.LeannIndex.from_device_csr(off.data_ptr(), nb.data_ptr(), N, entry, d)
idx.set_embeddings(None, device_ptr=x.data_ptr(), n=N, d=d)
q = synth.make_rows(N, d, 0, nq, device=dev, query=True)
ti, td = synth.brute_force_topk(x, q, 10)
oi = torch.zeros((nq,10), dtype=torch.int64, device=dev); od = torch.zeros((nq,10), device=dev); oc = torch.zeros(nq, dtype=torch.int32, device=dev)
n_leaf = N // 1000; n_super = max(1, n_leaf // 100)  # same_super: leaves sharing the 2 top tree levels
for ef in (128, 256, 512):
    idx.search_batch_device(q.data_ptr(), nq, d, 10, ef, oi.data_ptr(), od.data_ptr(), oc.data_ptr())
    st = idx.last_stats()
    rec = synth.recall_at_k(oi, oc, ti)
    tl = synth._leaf_of(ti[:,0], N, 1000); fl = synth._leaf_of(oi[:,0], N, 1000)
    same_leaf = (tl == fl).float().mean().item()
    same_super = ((tl // 100) == (fl // 100)).float().mean().item()
    # recall among queries that reached the right leaf
    ok = tl == fl
    hit = ((oi[:, :, None] == ti[:, None, :]).any(2)).float().sum(1) / 10
    depth_hist = [int(((tl // (10**j)) == (fl // (10**j))).sum().item()) for j in (0,1,2,3,4)]
    print("  queries whose found leaf agrees with the truth up to 10^j-blocks (j=0 same leaf .. 4):", depth_hist)
    print(f"ef={ef} recall={rec:.4f} same_leaf={same_leaf:.3f} same_super={same_super:.3f} recall|leaf_ok={hit[ok].mean().item():.3f} evals/q={st['evals']/nq:.0f} ms={st['kernel_ms']:.2f}")
# truth structure: are the 10 true NN all in one leaf?
tl_all = synth._leaf_of(ti.reshape(-1), N, 1000).reshape(nq,10)
print("truth single-leaf frac", (tl_all == tl_all[:, :1]).all(1).float().mean().item())
deg = (off[1:]-off[:-1])
print("deg hist", torch.bincount(deg)[25:61].tolist())
