"""Two-rank rehearsal of the multi-GPU path on ONE card (gloo rendezvous, both ranks on cuda:0):
the real HIP shard search writes its packed record in place, ShardedSearcher exchanges the
records with one all-gather (through host memory here; RCCL over xGMI on a multi-GPU node) and
isl_merge_topk_packed_async merges them.  Expected = the oracle's MultiIndexSearcher
(search.rs:211-237) over the same two sub-graphs, ids and f32 bits."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shards(n_total, d, world):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    import oracle as orc
    from _data import clustered_vectors, knn_graph
    from islands_amd.sharded import shard_range

    x = clustered_vectors(n_total, d, 5, per_cluster=40)
    out = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        xs = x[lo:hi]
        off, nb = knn_graph(xs, 14, seed=7 + r)
        out.append((lo, xs, orc.Csr(off, nb, entry_point=0)))
    qs = [clustered_vectors(40, d, 9 + b, per_cluster=3) for b in range(5)]
    return orc, x, out, qs


def _worker(rank, world, port, n_total, d, k, ef, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc, x, shards, qs = _shards(n_total, d, world)
    import islands_amd as ia
    from islands_amd.sharded import ShardedSearcher

    torch.cuda.set_device(0)
    lo, xs, csr = shards[rank]
    g = ia.CsrGraph(node_offsets=csr.node_offsets, neighbors=csr.neighbors, levels=csr.levels, entry_point=0,
                    num_nodes=csr.num_nodes, degree_counts=csr.degree_counts)
    idx = ia.LeannIndex.from_csr(g, dimension=d).upload(0)
    idx.set_embeddings(xs)
    s = ShardedSearcher(n_total, index=idx, device="cuda:0", depth=3)
    s.prepare(qs[0].shape[0], k, ef)
    assert s.world == world and s.rank == rank
    dq = [torch.from_numpy(q).cuda() for q in qs]
    # three batches in flight, then the rest
    res, handles = [], []
    for b, q in enumerate(dq):
        handles.append(s.submit(q, k, ef))
        if len(handles) == 3:
            (ids, dd, src, cnt), st = s.result(handles.pop(0), with_stats=True)
            assert st["allocations"] == 0 and st["queries"] == q.shape[0]
            res.append((ids.cpu().numpy().copy(), dd.cpu().numpy().copy(), src.cpu().numpy().copy(),
                        cnt.cpu().numpy().copy()))
    while handles:
        ids, dd, src, cnt = s.result(handles.pop(0))
        res.append((ids.cpu().numpy().copy(), dd.cpu().numpy().copy(), src.cpu().numpy().copy(), cnt.cpu().numpy().copy()))
    s.check_flags()
    if rank == 0:
        torch.save(res, out_path)
    # every rank holds the same merged answer
    gathered = [None] * world
    dist.all_gather_object(gathered, [r[0].tolist() for r in res])
    assert all(g_ == gathered[0] for g_ in gathered)
    dist.destroy_process_group()


def _failing_worker(rank, world, port, n_total, d, k, ef, out_path):
    """Batch 1: rank 1's shard search fails (queries of the wrong dimension on that rank only); batch 2:
    a query of rank 1's shard ends in NodeNotFound (an edge to a node without a row); batches 0 and 3
    are ordinary.  No rank may block, every rank must see batches 1 and 2 fail and 0 and 3 succeed."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["ISL_SHARD_TIMEOUT_MS"] = "20000"
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc, x, shards, qs = _shards(n_total, d, world)
    import islands_amd as ia
    from islands_amd.sharded import ShardedSearcher

    torch.cuda.set_device(0)
    lo, xs, csr = shards[rank]
    g = ia.CsrGraph(node_offsets=csr.node_offsets, neighbors=csr.neighbors, levels=csr.levels, entry_point=0,
                    num_nodes=csr.num_nodes, degree_counts=csr.degree_counts)
    idx = ia.LeannIndex.from_csr(g, dimension=d).upload(0)
    idx.set_embeddings(xs)
    s = ShardedSearcher(n_total, index=idx, device="cuda:0", depth=3)
    s.prepare(qs[0].shape[0], k, ef)
    log = []

    def run(q):
        try:
            h = s.submit(q, k, ef)
        except ia.CoreError as e:
            return ("submit", e.kind)
        try:
            ids, dd, src, cnt = s.result(h)
        except ia.CoreError as e:
            return ("result", e.kind)
        return ("ok", ids.cpu().numpy().copy().tolist())

    dq = [torch.from_numpy(q).cuda() for q in qs]
    log.append(run(dq[0]))
    # batch 1: the wrong dimension on rank 1 only (same nq, k: the exchange's record has the same size)
    bad = torch.zeros((dq[1].shape[0], d + 8), device="cuda:0") if rank == 1 else dq[1]
    log.append(run(bad))
    # batch 2: rank 1 loses the rows of the upper half of its shard -> its queries end in NodeNotFound
    if rank == 1:
        idx.set_embeddings(xs[: xs.shape[0] // 2])
    log.append(run(dq[2]))
    if rank == 1:
        idx.set_embeddings(xs)
    log.append(run(dq[3]))
    gathered = [None] * world
    dist.all_gather_object(gathered, log)
    if rank == 0:
        torch.save(gathered, out_path)
    s.close()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_a_failing_rank_fails_the_batch_on_every_rank_and_nobody_hangs(tmp_path):
    world, n_total, d, k, ef = 2, 3000, 32, 7, 40
    out = str(tmp_path / "log.pt")
    mp.spawn(_failing_worker, args=(world, _free_port(), n_total, d, k, ef, out), nprocs=world, join=True)
    logs = torch.load(out, weights_only=False)
    r0, r1 = logs
    # ordinary batches before and after: answered, identically on both ranks
    assert r0[0][0] == "ok" and r1[0] == r0[0]
    assert r0[3][0] == "ok" and r1[3] == r0[3]
    # batch 1: rank 1 reports its own error at submit (DimensionMismatch), rank 0 a SearchError at result
    assert r1[1] == ("submit", "DimensionMismatch"), r1[1]
    assert r0[1] == ("result", "SearchError"), r0[1]
    # batch 2: rank 1's own NodeNotFound at result, rank 0 a SearchError at result
    assert r1[2] == ("result", "NodeNotFound"), r1[2]
    assert r0[2] == ("result", "SearchError"), r0[2]


@pytest.mark.timeout(600)
def test_two_ranks_one_card_equal_multi_index_searcher(tmp_path):
    world, n_total, d, k, ef = 2, 3000, 32, 7, 40
    out = str(tmp_path / "merged.pt")
    mp.spawn(_worker, args=(world, _free_port(), n_total, d, k, ef, out), nprocs=world, join=True)
    res = torch.load(out, weights_only=False)
    orc, x, shards, qs = _shards(n_total, d, world)
    for b, q in enumerate(qs):
        ids_b, dd_b, src_b, cnt_b = res[b]
        for qi in range(q.shape[0]):
            li, ls = [], []
            for (lo, xs, csr) in shards:
                r = orc.leann_search(csr, xs, q[qi], k, ef)
                li.append(r.ids + np.uint64(lo))
                ls.append(r.dist)
            st, ids, sc, src = orc.multi_index_merge(li, ls, k)
            n = int(cnt_b[qi])
            assert n == ids.size
            assert ids_b[qi, :n].astype(np.uint64).tolist() == ids.tolist(), (b, qi)
            assert dd_b[qi, :n].view(np.uint32).tolist() == sc.view(np.uint32).tolist()
            assert src_b[qi, :n].tolist() == src.tolist()


def test_single_rank_rccl_group_and_host_entry():
    """The RCCL transport with the one rank this box has: ncclCommInitRank from a unique id, the
    communicator's all-gather of the packed record on the side stream, merge -- through
    isl_sharded_submit / _result and through the host-buffer entry isl_sharded_search_batch.
    Expected = the oracle's search of the one shard (MultiIndexSearcher over a single index)."""
    world, n_total, d, k, ef = 1, 2500, 32, 6, 48
    orc, x, shards, qs = _shards(n_total, d, world)
    import islands_amd as ia
    from islands_amd.sharded import ShardedSearcher

    lo, xs, csr = shards[0]
    g = ia.CsrGraph(node_offsets=csr.node_offsets, neighbors=csr.neighbors, levels=csr.levels, entry_point=0,
                    num_nodes=csr.num_nodes, degree_counts=csr.degree_counts)
    idx = ia.LeannIndex.from_csr(g, dimension=d).upload(0)
    idx.set_embeddings(xs)
    s = ShardedSearcher(n_total, index=idx, device="cuda:0", depth=2, transport="rccl")
    info = s.shard_group.info()
    assert info == {"world": 1, "rank": 0, "comm_ranks": 1, "rccl": True}
    s.prepare(qs[0].shape[0], k, ef)
    dq = [torch.from_numpy(q).cuda() for q in qs]
    got, handles = [], []
    for q in dq:
        handles.append(s.submit(q, k, ef))
        if len(handles) == 2:
            (ids, dd, src, cnt), st = s.result(handles.pop(0), with_stats=True)
            assert st["allocations"] == 0 and st["queries"] == q.shape[0]
            got.append((ids.cpu().numpy().copy(), dd.cpu().numpy().copy(), cnt.cpu().numpy().copy()))
    while handles:
        ids, dd, src, cnt = s.result(handles.pop(0))
        got.append((ids.cpu().numpy().copy(), dd.cpu().numpy().copy(), cnt.cpu().numpy().copy()))
    s.check_flags()
    h_ids, h_dd, h_src, h_cnt = s.search_batch(qs[0], k, ef)
    for b, q in enumerate(qs):
        for qi in range(q.shape[0]):
            r = orc.leann_search(csr, xs, q[qi], k, ef)
            n = int(got[b][2][qi])
            assert n == r.ids.size
            assert got[b][0][qi, :n].astype(np.uint64).tolist() == r.ids.tolist()
            assert got[b][1][qi, :n].view(np.uint32).tolist() == r.dist.view(np.uint32).tolist()
            if b == 0:
                assert h_ids[qi, :n].tolist() == r.ids.tolist() and int(h_cnt[qi]) == n
                assert h_dd[qi, :n].view(np.uint32).tolist() == r.dist.view(np.uint32).tolist()
                assert (h_src[qi, :n] == 0).all()
    # a handle is good once
    with pytest.raises(ia.CoreError):
        s.result(12345)
    s.close()


def test_empty_shard_answers_with_zero_counts():
    """A shard that holds no nodes (LeannIndex::search on an empty index is Ok(vec![]), leann.rs:875-877) still
    takes part in the exchange: its record carries zero counts and the merge returns what the others found --
    here, with one rank, nothing."""
    import islands_amd as ia
    from islands_amd.sharded import ShardedSearcher

    idx = ia.LeannIndex.with_defaults().upload(0)
    s = ShardedSearcher(0, index=idx, device="cuda:0", depth=2)
    q = np.zeros((5, 16), np.float32)
    ids, dd, src, cnt = s.search_batch(q, 3, 8)
    assert cnt.tolist() == [0] * 5
    s.close()
