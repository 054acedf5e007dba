"""Multi-GPU search: one process per GPU, the index sharded by node-id range.

Semantics = MultiIndexSearcher::search (src/core/search.rs:211-237) and the product's
cross-index merge (src/indexer/service.rs:775-801): every shard answers the whole query
batch in its own sub-graph, the per-shard top-k lists are concatenated in shard order,
stable-sorted by distance and truncated to k.

The whole data path lives behind the C ABI (include/islands_amd.h, "multi-GPU"):
isl_shard_group (RCCL communicator created from a unique id, or a host all-gather callback)
and isl_sharded_searcher (shard search -> ONE all-gather of the packed records on a side
stream behind a device event of the search -> merge kernel).  This module binds it and does
the one thing the C ABI leaves to the host: carrying rank 0's 128-byte unique id to the other
ranks, here through torch.distributed (any backend; plumbing only).

ShardedSearcher keeps an injection path for CPU tests (local_search / merge callables over a
gloo group): the sharding arithmetic and the record layout are shared with the C side
(isl_shard_record_bytes) and checked against it.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import CoreError, _check, _ffi
from ._ffi import SearchStatsC


def shard_range(n_total: int, rank: int, world: int) -> tuple[int, int]:
    """Node-id range [lo, hi) owned by `rank`."""
    return rank * n_total // world, (rank + 1) * n_total // world


def record_bytes(nq: int, k: int) -> int:
    """Bytes of one shard's packed answer record (== isl_shard_record_bytes)."""
    return (nq * k * 12 + nq * 4 + 15) // 16 * 16


POISON_COUNT_I32 = -1  # ISL_SHARD_POISON_COUNT (0xFFFFFFFF) as the int32 the count views hold


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
    # the contiguous copies must outlive the call (a temporary's block could be handed to the next copy)
    gi, gd, gc = g_ids.contiguous(), g_dist.contiguous(), g_cnt.contiguous()
    _check(_ffi.lib().isl_merge_topk(
        world, nq, kk, C.c_void_p(gi.data_ptr()), C.c_void_p(gd.data_ptr()),
        C.c_void_p(gc.data_ptr()), base.ctypes.data_as(C.c_void_p), k,
        C.c_void_p(m_ids.data_ptr()), C.c_void_p(m_dist.data_ptr()),
        C.c_void_p(m_src.data_ptr()), C.c_void_p(m_cnt.data_ptr()), 1, device_index, None))
    del gi, gd, gc
    return m_ids, m_dist, m_src, m_cnt


class ShardGroup:
    """isl_shard_group: the communicator of the ranks.  transport "rccl": rank 0's unique id is
    broadcast over the torch process group (whatever its backend), then every rank runs
    ncclCommInitRank inside the library.  transport "host": the exchange is a blocking all-gather
    over host memory through the torch process group (gloo) -- for ranks that share one card."""

    def __init__(self, device_index: int, group=None, transport: str = "rccl"):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device_index = device_index
        self.transport = transport
        self._h = C.c_void_p()
        self._cb = None
        lib = _ffi.lib()
        if transport == "rccl":
            uid = (C.c_uint8 * 128)()
            if self.rank == 0:
                _check(lib.isl_shard_unique_id(uid))
            if self.world > 1:
                box = [bytes(uid)]
                dist.broadcast_object_list(box, src=0, group=group)
                uid = (C.c_uint8 * 128).from_buffer_copy(box[0])
            _check(lib.isl_shard_group_create(device_index, self.world, self.rank, uid, C.byref(self._h)))
        elif transport == "host":
            world, grp = self.world, group

            def allgather(_user, send, recv, nbytes):
                try:
                    src = torch.frombuffer((C.c_uint8 * nbytes).from_address(send), dtype=torch.uint8)
                    dst = torch.frombuffer((C.c_uint8 * (nbytes * world)).from_address(recv), dtype=torch.uint8)
                    if world == 1:
                        dst.copy_(src)
                    else:
                        dist.all_gather_into_tensor(dst, src.clone(), group=grp)
                    return 0
                except Exception:  # never let an exception cross the C frame
                    return 1

            self._cb = _ffi.SHARD_ALLGATHER_FN(allgather)
            _check(lib.isl_shard_group_create_host(device_index, self.world, self.rank, self._cb, None,
                                                   C.byref(self._h)))
        else:
            raise ValueError(transport)

    def info(self) -> dict:
        w, r, n, rc = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        _check(_ffi.lib().isl_shard_group_info(self._h, C.byref(w), C.byref(r), C.byref(n), C.byref(rc)))
        return {"world": w.value, "rank": r.value, "comm_ranks": n.value, "rccl": bool(rc.value)}

    def close(self):
        if self._h:
            _ffi.lib().isl_shard_group_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedSearcher:
    """Search over R shards, one per rank.

    GPU path: `index` is this rank's LeannIndex over its node-id range (local ids); everything
    runs in libislands_amd.so (isl_sharded_*).  submit() enqueues search -> all-gather of the packed
    records -> merge without waiting for anything; result() completes a submitted batch.  Up to
    `depth` batches may be in flight; `transport` "rccl" (default when the process group's backend
    is nccl, and for a single rank) or "host" (gloo process groups: ranks sharing a card).

    Injection path (CPU tests): `local_search(queries, k, ef)` returns (ids [nq,k] int64,
    dist [nq,k] f32, count [nq] int32) with LOCAL ids and `merge(g_ids, g_dist, g_cnt, id_base, k)`
    turns the gathered [world, nq, k] lists into the global top-k; the sharding arithmetic, the
    record layout and the single collective are the same."""

    def __init__(self, n_total: int, local_search=None, merge=None, group=None, device="cpu", index=None,
                 depth: int = 8, transport: str | None = None):
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
        self._h = C.c_void_p()
        self.shard_group = None
        self._shapes = {}
        if index is not None:
            # a single rank needs no group (the header's "world == 1 needs no group"): RCCL is loaded and a
            # communicator made only when the caller asks for transport="rccl" explicitly
            if transport is None and self.world > 1:
                transport = "rccl" if self.backend == "nccl" else "host"
            dev_index = self.device.index or 0
            if transport is not None:
                self.shard_group = ShardGroup(dev_index, group, transport)
            _check(_ffi.lib().isl_sharded_searcher_new(
                index._h, self.shard_group._h if self.shard_group else None, n_total,
                self.id_base.ctypes.data_as(C.c_void_p), depth, C.byref(self._h)))

    # ------------------------------------------------------------------ injection path (CPU tests)
    def _all_gather_records(self, gathered: torch.Tensor, record: torch.Tensor):
        """gathered [world, B] <- every rank's record [B] (uint8), in rank order."""
        if self.world == 1:
            gathered[0].copy_(record)
        else:
            dist.all_gather_into_tensor(gathered.view(-1), record, group=self.group)

    def search_batch(self, queries, k: int, ef: int):
        if self.index is not None:
            q = np.ascontiguousarray(queries.cpu().numpy() if torch.is_tensor(queries) else queries, np.float32)
            nq, d = q.shape
            ids = np.zeros((nq, k), np.uint64)
            dd = np.zeros((nq, k), np.float32)
            src = np.zeros((nq, k), np.uint32)
            cnt = np.zeros(nq, np.uint32)
            _check(_ffi.lib().isl_sharded_search_batch(
                self._h, q.ctypes.data_as(C.c_void_p), nq, d, k, ef, ids.ctypes.data_as(C.c_void_p),
                dd.ctypes.data_as(C.c_void_p), src.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p)))
            return ids, dd, src, cnt
        # A rank whose shard search fails still takes part in the exchange, with a record whose counts are
        # ISL_SHARD_POISON_COUNT; every rank then fails the batch (MultiIndexSearcher::search propagates an
        # index's error, search.rs:215) and nobody is left inside the collective -- the rule of shard.hip.
        nq = int(queries.shape[0])
        local_error = None
        try:
            ids, dd, cnt = self.local_search(queries, k, ef)
        except Exception as e:  # noqa: BLE001 -- whatever the local search raised is this rank's error
            local_error = e
        B = record_bytes(nq, k)
        assert B == int(_ffi.lib().isl_shard_record_bytes(nq, k))
        rec = torch.zeros(B, dtype=torch.uint8, device=self.device)
        r_ids, r_dd, r_cnt = record_views(rec, nq, k)
        if local_error is None:
            r_ids.copy_(ids); r_dd.copy_(dd); r_cnt.copy_(cnt)
        else:
            r_cnt.fill_(POISON_COUNT_I32)
        gathered = torch.zeros((self.world, B), dtype=torch.uint8, device=self.device)
        self._all_gather_records(gathered, rec)  # the one exchange step of the path
        if local_error is not None:
            raise local_error
        g_ids, g_dd, g_cnt = record_views(gathered, nq, k)
        failed = (g_cnt == POISON_COUNT_I32).any(dim=1)
        if bool(failed.any()):
            raise CoreError(11, "Search error: the shard search of rank(s) "
                                f"{torch.nonzero(failed).flatten().tolist()} failed for this batch")
        return self.merge(g_ids, g_dd, g_cnt, self.id_base, k)

    # ------------------------------------------------------------------ GPU path (C ABI)
    def prepare(self, nq: int, k: int, ef: int):
        """isl_sharded_prepare: buffers for `depth` batches in flight, the index's lanes, and (RCCL) the
        communicator's first collective.  Collective over the ranks."""
        _check(_ffi.lib().isl_sharded_prepare(self._h, nq, k, ef))
        return self

    def submit(self, d_queries: torch.Tensor, k: int, ef: int, stream: int = 0) -> int:
        """isl_sharded_submit: enqueue one batch (search, all-gather of the records, merge)."""
        nq, d = d_queries.shape
        h = C.c_uint64()
        _check(_ffi.lib().isl_sharded_submit(self._h, C.c_void_p(d_queries.data_ptr()), nq, d, k, ef,
                                             C.c_void_p(stream), C.byref(h)))
        self._shapes[int(h.value)] = (nq, k)
        return int(h.value)

    def result(self, handle: int, with_stats: bool = False):
        """isl_sharded_result: (ids [nq,k] int64 global, dist, src shard, count) as tensors over the
        library's device buffers -- valid until `depth` further batches have been submitted --
        [+ this rank's search counters]."""
        nq, k = self._shapes.pop(handle, (0, 0))  # (an unknown handle is the library's error to raise)
        p = [C.c_void_p() for _ in range(4)]
        st = SearchStatsC()
        _check(_ffi.lib().isl_sharded_result(self._h, handle, C.byref(p[0]), C.byref(p[1]), C.byref(p[2]),
                                             C.byref(p[3]), C.byref(st)))
        out = (_device_view(p[0].value, (nq, k), torch.int64, self.device),
               _device_view(p[1].value, (nq, k), torch.float32, self.device),
               _device_view(p[2].value, (nq, k), torch.int32, self.device),
               _device_view(p[3].value, (nq,), torch.int32, self.device))
        if with_stats:
            return out, {f: getattr(st, f) for f, _ in SearchStatsC._fields_}
        return out

    def check_flags(self):
        """NaN scores make the reference's merge panic (search.rs:231); lists must be ascending."""
        f = C.c_uint32()
        _check(_ffi.lib().isl_sharded_flags(self._h, C.byref(f)))
        if f.value & 1:
            raise CoreError(11, "Search error: NaN score in merge (the reference panics here)")
        if f.value & 2:
            raise CoreError(14, "per-list scores must be ascending")
        if f.value & 4:
            raise CoreError(11, "Search error: a rank failed one of the batches")

    def close(self):
        if self._h:
            _ffi.lib().isl_sharded_searcher_free(self._h)
            self._h = C.c_void_p()
        if self.shard_group is not None:
            self.shard_group.close()
            self.shard_group = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _DevArray:
    """__cuda_array_interface__ over a raw device pointer owned by the library."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def _device_view(ptr: int, shape, dtype, device) -> torch.Tensor:
    typestr = {torch.int64: "<i8", torch.float32: "<f4", torch.int32: "<i4"}[dtype]
    return torch.as_tensor(_DevArray(ptr, shape, typestr), device=device)
