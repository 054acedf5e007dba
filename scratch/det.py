import sys, os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import numpy as np, torch
import islands_amd as ia
from islands_amd import synth
dev = torch.device("cuda:0")
N, d, nq, ef, k = 2000000, 768, 1024, 128, 10
x = synth.make_rows(N, d, 0, N, device=dev)
off, nb, entry = synth.build_graph(x)
idx = ia.LeannIndex.from_device_csr(off.data_ptr(), nb.data_ptr(), N, entry, d)
idx.set_embeddings(None, device_ptr=x.data_ptr(), n=N, d=d)
qs = [synth.make_rows(N, d, b * nq, nq, device=dev, query=True) for b in range(4)]
truth = [synth.brute_force_topk(x, q, k)[0] for q in qs]
torch.cuda.synchronize()
def mk(): return (torch.zeros((nq,k), dtype=torch.int64, device=dev), torch.zeros((nq,k), device=dev), torch.zeros(nq, dtype=torch.int32, device=dev))
ref = []
for b in range(4):
    o = mk()
    idx.search_batch_device(qs[b].data_ptr(), nq, d, k, ef, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
    ref.append(o)
    print("sync batch", b, "recall", synth.recall_at_k(o[0], o[2], truth[b]), idx.last_stats()["evals"])
# repeat sync
for b in range(4):
    o = mk()
    idx.search_batch_device(qs[b].data_ptr(), nq, d, k, ef, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
    print("sync again", b, "same ids", bool((o[0] == ref[b][0]).all()), "same dist", bool((o[1] == ref[b][1]).all()))
for depth in (2, 3, 4):
    outs = [mk() for _ in range(4)]
    toks = []
    for b in range(depth):
        toks.append(idx.search_batch_device_async(qs[b].data_ptr(), nq, d, k, ef, outs[b][0].data_ptr(), outs[b][1].data_ptr(), outs[b][2].data_ptr()))
    for b in range(depth):
        idx.wait(toks[b])
    for b in range(depth):
        diff = (outs[b][0] != ref[b][0]).any(1)
        print("async depth", depth, "batch", b, "same ids", not bool(diff.any()), "queries differing", int(diff.sum()), "recall", synth.recall_at_k(outs[b][0], outs[b][2], truth[b]))
