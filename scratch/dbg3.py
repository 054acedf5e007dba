import sys, os
sys.path[:0] = ['.', 'oracle', 'tests']
import numpy as np, torch
from islands_amd import synth
dev = torch.device("cuda:0")
N, d = int(sys.argv[1]), 768
x = synth.make_rows(N, d, 0, N, device=dev)
off, nb, entry = synth.build_graph(x)
n_leaf = N // 1000; n_super = max(1, n_leaf // 100)
def sup(ids): return (synth._leaf_of(ids, N, 1000) % n_super)
g = torch.Generator(device=dev); g.manual_seed(7)
perm = torch.randperm(N, generator=g, device=dev)
levels=[perm]; cur=perm
while cur.numel() > 16:
    cur = cur[:max(1,cur.numel()//32)]; levels.append(cur)
print("levels", [l.numel() for l in levels], "entry", entry, "top", levels[-1].tolist())
def row(u): return nb[off[u]:off[u+1]].long()
top = levels[-1]
L3 = set(levels[-2].tolist())
reach1 = torch.unique(torch.cat([row(int(u)) for u in top.tolist()]))
in3 = torch.tensor([int(v) in L3 for v in reach1.tolist()], device=dev)
print("1-hop from top:", reach1.numel(), "of which L3:", int(in3.sum()), "supers covered by those L3:", torch.unique(sup(reach1[in3])).numel(), "of", n_super)
print("supers covered by all 1-hop:", torch.unique(sup(reach1)).numel())
print("entry row len", row(entry).numel(), "entry row supers", torch.unique(sup(row(entry))).numel())
l3 = levels[-2]
print("supers having an L3 node:", torch.unique(sup(l3)).numel())
lv_of = torch.zeros(N, dtype=torch.int64, device=dev)
for li, lv in enumerate(levels): lv_of[lv] = li
u = int(top[0]); r = row(u)
print("top node row levels:", torch.bincount(lv_of[r], minlength=len(levels)).tolist())
u = int(l3[20]); r = row(u)
print("L3 node row levels:", torch.bincount(lv_of[r], minlength=len(levels)).tolist(), "supers", torch.unique(sup(r)).numel(), "own super", int(sup(torch.tensor([u],device=dev))[0]))
