import sys, os, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import numpy as np, torch
import islands_amd as ia
from islands_amd import synth
dev = torch.device("cuda:0")
N, d, nq, ef, k = int(sys.argv[1]), 768, 1024, 128, 10
x = synth.make_rows(N, d, 0, N, device=dev)
qs = [synth.make_rows(N, d, b * nq, nq, device=dev, query=True) for b in range(2)]
truth = [synth.brute_force_topk(x, q, k)[0] for q in qs]
o = (torch.zeros((nq,k), dtype=torch.int64, device=dev), torch.zeros((nq,k), device=dev), torch.zeros(nq, dtype=torch.int32, device=dev))
for kw in [dict(), dict(k_upper=20, pool=96), dict(k0=32), dict(child_cap=36, k_upper=18), dict(k0=24, k_upper=20, pool=96, child_cap=32)]:
    t = time.time()
    off, nb, entry = synth.build_graph(x, **kw)
    torch.cuda.synchronize(); bt = time.time() - t
    idx = ia.LeannIndex.from_device_csr(off.data_ptr(), nb.data_ptr(), N, entry, d)
    idx.set_embeddings(None, device_ptr=x.data_ptr(), n=N, d=d)
    rec = []; ev = 0; ms = 0
    for b in range(2):
        idx.search_batch_device(qs[b].data_ptr(), nq, d, k, ef, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
        rec.append(synth.recall_at_k(o[0], o[2], truth[b])); st = idx.last_stats(); ev += st["evals"]; ms += st["kernel_ms"]
    print(kw, "build %.1fs" % bt, "recall", [round(r,4) for r in rec], "evals/q", ev // (2*nq), "ms", round(ms/2, 2), "deg", round(float((off[1:]-off[:-1]).float().mean()),1), flush=True)
    del idx
