import sys, os
sys.path[:0] = ['.', 'oracle', 'tests']
os.environ["ISL_DEBUG"]="1"
import numpy as np, torch
import islands_amd as ia
from islands_amd import synth
dev = torch.device("cuda:0")
N, d, nq = 1000000, 768, 1024
x = synth.make_rows(N, d, 0, N, device=dev)
off, nb, entry = synth.build_graph(x)
idx = ia.LeannIndex.from_device_csr(off.data_ptr(), nb.data_ptr(), N, entry, d)
idx.set_embeddings(None, device_ptr=x.data_ptr(), n=N, d=d)
q = synth.make_rows(N, d, 0, nq, device=dev, query=True)
oi = torch.zeros((nq,10), dtype=torch.int64, device=dev); od = torch.zeros((nq,10), device=dev); oc = torch.zeros(nq, dtype=torch.int32, device=dev)
for ef in (64, 128):
    for it in range(2):
        idx.search_batch_device(q.data_ptr(), nq, d, 10, ef, oi.data_ptr(), od.data_ptr(), oc.data_ptr())
        print(ef, idx.last_stats())
# look at ties in output
dd = od.cpu().numpy()
print("adjacent equal in top10:", int((dd[:,1:]==dd[:,:-1]).any(1).sum()))
print(dd[:3])
