"""One-rank RCCL sanity check of the collectives bench.py / ShardedSearcher use (flat views of
[world, nq, k] outputs): run under torchrun with --nproc-per-node 1."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
torch.cuda.set_device(0)
world, nq, k = dist.get_world_size(), 16, 10
dev = torch.device("cuda:0")
ids = torch.arange(nq * k, dtype=torch.int64, device=dev).view(nq, k)
dd = torch.rand((nq, k), device=dev)
cnt = torch.full((nq,), k, dtype=torch.int32, device=dev)
g_ids = torch.zeros((world, nq, k), dtype=torch.int64, device=dev)
g_dd = torch.zeros((world, nq, k), dtype=torch.float32, device=dev)
g_cnt = torch.zeros((world, nq), dtype=torch.int32, device=dev)
dist.all_gather_into_tensor(g_ids.view(world * nq, k), ids.contiguous())
dist.all_gather_into_tensor(g_dd.view(world * nq, k), dd.contiguous())
dist.all_gather_into_tensor(g_cnt.view(world * nq), cnt.contiguous())
t = torch.tensor([1.5], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(g_ids[0], ids) and torch.equal(g_dd[0], dd) and torch.equal(g_cnt[0], cnt)
print("nccl sanity ok: world", world)
dist.destroy_process_group()
