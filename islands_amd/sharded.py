"""Multi-GPU search: one process per GPU, the index sharded by node-id range.

Semantics = MultiIndexSearcher::search (src/core/search.rs:211-237) and the product's
cross-index merge (src/indexer/service.rs:775-801): every shard answers the whole query
batch in its own sub-graph, the per-shard top-k lists are concatenated in shard order,
stable-sorted by distance and truncated to k.  The only data-path collective is one
all-gather of nq * k * (8 + 4) + nq * 4 bytes per rank (RCCL over xGMI when the process
group's backend is "nccl"; "gloo" in the CPU tests).

torch.distributed is plumbing here; the search and the merge run in libislands_amd.so.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _check, _ffi


def shard_range(n_total: int, rank: int, world: int) -> tuple[int, int]:
    """Node-id range [lo, hi) owned by `rank`."""
    return rank * n_total // world, (rank + 1) * n_total // world


def device_merge(g_ids, g_dist, g_cnt, id_base, k: int, device_index: int):
    """isl_merge_topk on tensors resident on the GPU: [world, nq, k] -> [nq, k]."""
    world, nq, kk = g_ids.shape
    dev = g_ids.device
    m_ids = torch.zeros((nq, k), dtype=torch.int64, device=dev)
    m_dist = torch.zeros((nq, k), dtype=torch.float32, device=dev)
    m_src = torch.zeros((nq, k), dtype=torch.int32, device=dev)
    m_cnt = torch.zeros(nq, dtype=torch.int32, device=dev)
    base = np.ascontiguousarray(id_base, dtype=np.uint64)
    _check(_ffi.lib().isl_merge_topk(
        world, nq, kk, C.c_void_p(g_ids.data_ptr()), C.c_void_p(g_dist.data_ptr()),
        C.c_void_p(g_cnt.data_ptr()), base.ctypes.data_as(C.c_void_p), k,
        C.c_void_p(m_ids.data_ptr()), C.c_void_p(m_dist.data_ptr()),
        C.c_void_p(m_src.data_ptr()), C.c_void_p(m_cnt.data_ptr()), 1, device_index, None))
    return m_ids, m_dist, m_src, m_cnt


class ShardedSearcher:
    """Search over R shards, one per rank.  `local_search(queries, k, ef)` answers the batch on
    this rank's shard with LOCAL ids and returns (ids [nq,k] int64, dist [nq,k] f32,
    count [nq] int32) on `device`; `merge(g_ids, g_dist, g_cnt, id_base, k)` turns the gathered
    [world, nq, k] lists into the global top-k.  In production both are the HIP entry points
    (LeannIndex.search_batch_device / device_merge); the CPU tests inject the oracle."""

    def __init__(self, n_total: int, local_search, merge, group=None, device="cpu"):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_total = n_total
        self.id_base = np.array([shard_range(n_total, r, self.world)[0] for r in range(self.world)],
                                dtype=np.uint64)
        self.local_search = local_search
        self.merge = merge
        self.device = torch.device(device)

    def search_batch(self, queries, k: int, ef: int):
        ids, dd, cnt = self.local_search(queries, k, ef)
        nq = ids.shape[0]
        if self.world == 1:
            g_ids, g_dist, g_cnt = ids[None], dd[None], cnt[None]
        else:
            g_ids = torch.zeros((self.world, nq, k), dtype=torch.int64, device=self.device)
            g_dist = torch.zeros((self.world, nq, k), dtype=torch.float32, device=self.device)
            g_cnt = torch.zeros((self.world, nq), dtype=torch.int32, device=self.device)
            # the one exchange step of the path
            # (rank-major concatenation along dim 0 == the [world, nq, k] stack)
            dist.all_gather_into_tensor(g_ids.view(self.world * nq, k), ids.contiguous(),
                                        group=self.group)
            dist.all_gather_into_tensor(g_dist.view(self.world * nq, k), dd.contiguous(),
                                        group=self.group)
            dist.all_gather_into_tensor(g_cnt.view(self.world * nq), cnt.contiguous(),
                                        group=self.group)
        return self.merge(g_ids, g_dist, g_cnt, self.id_base, k)
