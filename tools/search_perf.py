import sys, os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import numpy as np, torch
import islands_amd as ia
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
ef = int(sys.argv[3]) if len(sys.argv) > 3 else 128
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
d = 768
x = synth.make_rows(N, d, 0, N, device=dev)
off, nb, entry = synth.build_graph(x)
idx = ia.LeannIndex.from_device_csr(off.data_ptr(), nb.data_ptr(), N, entry, d)
idx.set_embeddings(None, device_ptr=x.data_ptr(), n=N, d=d)
q = synth.make_rows(N, d, 0, nq, device=dev, query=True)
oi = torch.zeros((nq,10), dtype=torch.int64, device=dev); od = torch.zeros((nq,10), device=dev); oc = torch.zeros(nq, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
for it in range(reps):
    idx.search_batch_device(q.data_ptr(), nq, d, 10, ef, oi.data_ptr(), od.data_ptr(), oc.data_ptr())
    st = idx.last_stats()
    print(f"N={N} nq={nq} ef={ef} kernel_ms={st['kernel_ms']:.3f} evals/q={st['evals']/nq:.0f} hops/q={st['expansions']/nq:.0f} GB/s={st['evals']*d*4/st['kernel_ms']/1e6:.0f} replay={st['replayed']} pushes/q={st['pushes']/nq:.0f}")
