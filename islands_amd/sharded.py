"""Multi-GPU search: one process per GPU, the index sharded by node-id range.

Semantics = MultiIndexSearcher::search (src/core/search.rs:211-237) and the product's
cross-index merge (src/indexer/service.rs:775-801): every shard answers the whole query
batch in its own sub-graph, the per-shard top-k lists are concatenated in shard order,
stable-sorted by distance and truncated to k.

The exchange is ONE collective per batch: every rank's answers form one packed record
(ids u64[nq][k] | distances f32[nq][k] | counts u32[nq], isl_shard_record_bytes) that the
search kernels write in place, all-gathered in rank order (RCCL over xGMI when the process
group's backend is "nccl"; "gloo" in the CPU tests and in the one-card rehearsal) and merged
by isl_merge_topk_packed_async.  On the GPU path nothing in a step waits on the host: the
collective and the merge are enqueued on a side stream behind an event of the search
(isl_search_stream_wait), so they overlap the traversals of the batches submitted after it.

torch.distributed is plumbing here; the search and the merge run in libislands_amd.so.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import CoreError, _check, _ffi


def shard_range(n_total: int, rank: int, world: int) -> tuple[int, int]:
    """Node-id range [lo, hi) owned by `rank`."""
    return rank * n_total // world, (rank + 1) * n_total // world


def record_bytes(nq: int, k: int) -> int:
    """Bytes of one shard's packed answer record (== isl_shard_record_bytes)."""
    return (nq * k * 12 + nq * 4 + 15) // 16 * 16


def record_views(buf: torch.Tensor, nq: int, k: int):
    """(ids [.., nq, k] int64, dist [.., nq, k] f32, count [.., nq] int32) views of a uint8 record
    buffer [B] or of gathered records [world, B]."""
    lead = buf.shape[:-1]
    a, b = nq * k * 8, nq * k * 12
    ids = buf[..., :a].view(torch.int64).reshape(*lead, nq, k)
    dd = buf[..., a:b].view(torch.float32).reshape(*lead, nq, k)
    cnt = buf[..., b:b + nq * 4].view(torch.int32).reshape(*lead, nq)
    return ids, dd, cnt


def device_merge(g_ids, g_dist, g_cnt, id_base, k: int, device_index: int):
    """isl_merge_topk on tensors resident on the GPU: [world, nq, k] -> [nq, k] (synchronous)."""
    world, nq, kk = g_ids.shape
    dev = g_ids.device
    m_ids = torch.zeros((nq, k), dtype=torch.int64, device=dev)
    m_dist = torch.zeros((nq, k), dtype=torch.float32, device=dev)
    m_src = torch.zeros((nq, k), dtype=torch.int32, device=dev)
    m_cnt = torch.zeros(nq, dtype=torch.int32, device=dev)
    base = np.ascontiguousarray(id_base, dtype=np.uint64)
    _check(_ffi.lib().isl_merge_topk(
        world, nq, kk, C.c_void_p(g_ids.contiguous().data_ptr()), C.c_void_p(g_dist.contiguous().data_ptr()),
        C.c_void_p(g_cnt.contiguous().data_ptr()), base.ctypes.data_as(C.c_void_p), k,
        C.c_void_p(m_ids.data_ptr()), C.c_void_p(m_dist.data_ptr()),
        C.c_void_p(m_src.data_ptr()), C.c_void_p(m_cnt.data_ptr()), 1, device_index, None))
    return m_ids, m_dist, m_src, m_cnt


class ShardedSearcher:
    """Search over R shards, one per rank.

    GPU path: `index` is this rank's LeannIndex over its node-id range (local ids).  submit()
    enqueues search -> all-gather of the packed records -> merge without waiting for anything;
    result() completes a submitted batch.  Up to `depth` batches may be in flight.

    Injection path (CPU tests): `local_search(queries, k, ef)` returns (ids [nq,k] int64,
    dist [nq,k] f32, count [nq] int32) with LOCAL ids and `merge(g_ids, g_dist, g_cnt, id_base, k)`
    turns the gathered [world, nq, k] lists into the global top-k; the sharding arithmetic, the
    record layout and the single collective are the same code."""

    def __init__(self, n_total: int, local_search=None, merge=None, group=None, device="cpu", index=None,
                 depth: int = 8):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.backend = dist.get_backend(group) if dist.is_initialized() else "none"
        self.n_total = n_total
        self.id_base = np.array([shard_range(n_total, r, self.world)[0] for r in range(self.world)],
                                dtype=np.uint64)
        self.local_search = local_search
        self.merge = merge
        self.device = torch.device(device)
        self.index = index
        self.depth = depth
        self._slots = {}      # (nq, k) -> list of per-slot buffers
        self._inflight = {}   # handle -> slot state
        self._next = 1
        if index is not None:
            self._side = torch.cuda.Stream(device=self.device)
            self._d_base = torch.from_numpy(self.id_base.astype(np.int64)).to(self.device)
            self._flags = torch.zeros(1, dtype=torch.int32, device=self.device)

    # ------------------------------------------------------------------ one collective
    def _all_gather_records(self, gathered: torch.Tensor, record: torch.Tensor):
        """gathered [world, B] <- every rank's record [B] (uint8), in rank order."""
        if self.world == 1:
            gathered[0].copy_(record)
        elif self.backend == "nccl" or record.device.type == "cpu":
            dist.all_gather_into_tensor(gathered.view(-1), record, group=self.group)
        else:
            # rehearsal on a box with fewer cards than ranks: the ranks share a card and the
            # exchange goes through host memory (gloo); synchronous by nature
            torch.cuda.current_stream(record.device).synchronize()
            host = torch.empty(gathered.shape, dtype=torch.uint8)
            dist.all_gather_into_tensor(host.view(-1), record.cpu(), group=self.group)
            gathered.copy_(host)

    # ------------------------------------------------------------------ injection path
    def search_batch(self, queries, k: int, ef: int):
        if self.index is not None:
            q = queries if torch.is_tensor(queries) else torch.as_tensor(np.ascontiguousarray(queries, np.float32))
            q = q.to(self.device).contiguous()
            h = self.submit(q, k, ef)
            return self.result(h)
        ids, dd, cnt = self.local_search(queries, k, ef)
        nq = ids.shape[0]
        B = record_bytes(nq, k)
        rec = torch.zeros(B, dtype=torch.uint8, device=self.device)
        r_ids, r_dd, r_cnt = record_views(rec, nq, k)
        r_ids.copy_(ids); r_dd.copy_(dd); r_cnt.copy_(cnt)
        gathered = torch.zeros((self.world, B), dtype=torch.uint8, device=self.device)
        self._all_gather_records(gathered, rec)  # the one exchange step of the path
        g_ids, g_dd, g_cnt = record_views(gathered, nq, k)
        return self.merge(g_ids, g_dd, g_cnt, self.id_base, k)

    # ------------------------------------------------------------------ GPU path
    def _slot(self, nq: int, k: int):
        pool = self._slots.setdefault((nq, k), [])
        for s in pool:
            if not s["busy"]:
                return s
        if len(pool) >= self.depth:
            raise CoreError(11, f"Search error: {self.depth} sharded batches already in flight; call result() first")
        B = record_bytes(nq, k)
        dev = self.device
        s = {"busy": False, "B": B,
             "rec": torch.zeros(B, dtype=torch.uint8, device=dev),
             "gath": torch.zeros((self.world, B), dtype=torch.uint8, device=dev),
             "ids": torch.zeros((nq, k), dtype=torch.int64, device=dev),
             "dist": torch.zeros((nq, k), dtype=torch.float32, device=dev),
             "src": torch.zeros((nq, k), dtype=torch.int32, device=dev),
             "cnt": torch.zeros(nq, dtype=torch.int32, device=dev),
             "done": torch.cuda.Event()}
        pool.append(s)
        return s

    def prepare(self, nq: int, k: int, ef: int):
        """Buffers for `depth` batches in flight and the index's lanes, ahead of time."""
        self.index.prepare(nq, ef, k, min(self.depth, 16))
        made = [self._slot(nq, k) for _ in range(self.depth)]
        for s in made:
            s["busy"] = True
        for s in made:
            s["busy"] = False
        return self

    def submit(self, d_queries: torch.Tensor, k: int, ef: int) -> int:
        """Enqueue one batch: search on a lane of the index, then -- on the side stream, behind the
        search's event -- the all-gather of the records and the merge.  Returns a handle."""
        nq, d = d_queries.shape
        s = self._slot(nq, k)
        rec = s["rec"]
        base = rec.data_ptr()
        tok = self.index.search_batch_device_async(d_queries.data_ptr(), nq, d, k, ef, base, base + nq * k * 8,
                                                   base + nq * k * 12)
        lib = _ffi.lib()
        side = self._side
        _check(lib.isl_search_stream_wait(self.index._h, tok, C.c_void_p(side.cuda_stream)))
        with torch.cuda.stream(side):
            self._all_gather_records(s["gath"], rec)
            _check(lib.isl_merge_topk_packed_async(
                self.world, nq, k, C.c_void_p(s["gath"].data_ptr()), s["B"], C.c_void_p(self._d_base.data_ptr()), k,
                C.c_void_p(s["ids"].data_ptr()), C.c_void_p(s["dist"].data_ptr()), C.c_void_p(s["src"].data_ptr()),
                C.c_void_p(s["cnt"].data_ptr()), C.c_void_p(self._flags.data_ptr()), self.device.index or 0,
                C.c_void_p(side.cuda_stream)))
            s["done"].record(side)
        s["busy"] = True
        h = self._next
        self._next += 1
        self._inflight[h] = (s, tok)
        return h

    def result(self, handle: int, with_stats: bool = False):
        """Completes a submitted batch: (ids [nq,k] int64 global, dist, src shard, count) on the
        device -- valid until the slot is reused `depth` submissions later -- [+ search counters]."""
        s, tok = self._inflight.pop(handle)
        try:
            st = self.index.wait_stats(tok)      # per-query failures of this rank's shard surface here
            s["done"].synchronize()
        finally:
            s["busy"] = False
        out = (s["ids"], s["dist"], s["src"], s["cnt"])
        return (out, st) if with_stats else out

    def check_flags(self):
        """NaN scores make the reference's merge panic (search.rs:231); lists must be ascending."""
        f = int(self._flags.item())
        if f & 1:
            raise CoreError(11, "Search error: NaN score in merge (the reference panics here)")
        if f & 2:
            raise CoreError(14, "per-list scores must be ascending")
