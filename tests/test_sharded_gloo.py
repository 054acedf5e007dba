"""World-size-2 test of the multi-GPU exchange step on CPU (gloo): per-shard top-k lists are
all-gathered in shard order and merged with MultiIndexSearcher semantics (search.rs:211-237).
The shard searches and the merge are played by the CPU oracle here (no GPU in this test);
what is under test is the sharding arithmetic, the gathered layout and the id re-basing of
islands_amd.sharded, against a single-process oracle run over the same two sub-graphs."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup_shards(n_total, d, world):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    import oracle as orc
    from _data import clustered_vectors, knn_graph
    from islands_amd.sharded import shard_range

    x = clustered_vectors(n_total, d, 5, per_cluster=40)
    shards = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        xs = x[lo:hi]
        off, nb = knn_graph(xs, 12, seed=7 + r)
        shards.append((lo, xs, orc.Csr(off, nb, entry_point=0)))
    q = clustered_vectors(24, d, 9, per_cluster=3)
    return orc, x, shards, q


def _worker(rank, world, port, n_total, d, k, ef, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc, x, shards, q = _setup_shards(n_total, d, world)
    from islands_amd.sharded import ShardedSearcher

    lo, xs, csr = shards[rank]

    def local_search(queries, k, ef):
        ids, dd, cnt, _ = orc.leann_search_batch(csr, xs, queries, k, ef)
        return (torch.from_numpy(ids.astype(np.int64)), torch.from_numpy(dd),
                torch.from_numpy(cnt.astype(np.int32)))

    def merge(g_ids, g_dist, g_cnt, id_base, k):
        w, nq, kk = g_ids.shape
        out = []
        for qi in range(nq):
            li = [g_ids[l, qi, :g_cnt[l, qi]].numpy().astype(np.uint64) + id_base[l] for l in range(w)]
            ls = [g_dist[l, qi, :g_cnt[l, qi]].numpy() for l in range(w)]
            st, ids, sc, src = orc.multi_index_merge(li, ls, k)
            assert st == 0
            out.append((ids.tolist(), sc.tolist(), src.tolist()))
        return out

    s = ShardedSearcher(n_total, local_search, merge)
    assert s.world == world and s.rank == rank
    res = s.search_batch(q, k, ef)
    if rank == 0:
        torch.save(res, out_path)
    # every rank holds the same merged answer
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    assert all(g == gathered[0] for g in gathered)
    dist.destroy_process_group()


def test_shard_range_covers_everything():
    sys.path.insert(0, ROOT)
    from islands_amd.sharded import shard_range

    for n, w in ((10, 3), (100, 8), (7, 2), (10_000_000, 8)):
        parts = [shard_range(n, r, w) for r in range(w)]
        assert parts[0][0] == 0 and parts[-1][1] == n
        assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))


@pytest.mark.timeout(300)
def test_two_rank_gather_and_merge(tmp_path):
    world, n_total, d, k, ef = 2, 600, 16, 5, 24
    out = str(tmp_path / "merged.pt")
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_total, d, k, ef, out), nprocs=world, join=True)
    res = torch.load(out)
    # single-process reference: MultiIndexSearcher over the same two sub-graphs
    orc, x, shards, q = _setup_shards(n_total, d, world)
    for qi in range(q.shape[0]):
        li, ls = [], []
        for (lo, xs, csr) in shards:
            r = orc.leann_search(csr, xs, q[qi], k, ef)
            li.append(r.ids + np.uint64(lo))
            ls.append(r.dist)
        st, ids, sc, src = orc.multi_index_merge(li, ls, k)
        assert res[qi][0] == ids.tolist() and res[qi][2] == src.tolist()
        assert res[qi][1] == sc.tolist()
        # global ids index the unsharded matrix: distances must match a direct evaluation
        for gid, dd in zip(ids.tolist(), sc.tolist()):
            assert orc.distance(orc.COSINE, q[qi], x[gid])[1] == np.float32(dd)


def _poison_worker(rank, world, port, n_total, d, k, ef, out_path):
    """Batch 0 ordinary, batch 1: rank 1's local search raises, batch 2 ordinary again."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc, x, shards, q = _setup_shards(n_total, d, world)
    import islands_amd as ia
    from islands_amd.sharded import POISON_COUNT_I32, ShardedSearcher, record_bytes, record_views

    lo, xs, csr = shards[rank]
    calls = {"n": 0}

    def local_search(queries, k, ef):
        calls["n"] += 1
        if rank == 1 and calls["n"] == 2:
            raise ia.CoreError(5, "Node not found: 12345", node=12345)
        ids, dd, cnt, _ = orc.leann_search_batch(csr, xs, queries, k, ef)
        return (torch.from_numpy(ids.astype(np.int64)), torch.from_numpy(dd), torch.from_numpy(cnt.astype(np.int32)))

    seen = []

    def merge(g_ids, g_dist, g_cnt, id_base, k):
        seen.append(g_cnt.clone())
        return [g_ids[:, qi, :].tolist() for qi in range(g_ids.shape[1])]

    s = ShardedSearcher(n_total, local_search, merge)
    log = []
    for b in range(3):
        try:
            res = s.search_batch(q, k, ef)
            log.append(("ok", res))
        except ia.CoreError as e:
            log.append(("error", e.kind, str(e)))
    # the poisoned record never reaches the merge, and the count views read it as POISON_COUNT_I32
    assert len(seen) == 2 and all(bool((c != POISON_COUNT_I32).all()) for c in seen)
    rec = torch.zeros(record_bytes(4, k), dtype=torch.uint8)
    record_views(rec, 4, k)[2].fill_(POISON_COUNT_I32)
    assert rec[4 * k * 12:4 * k * 12 + 16].tolist() == [255] * 16  # == ISL_SHARD_POISON_COUNT, little endian
    gathered = [None] * world
    dist.all_gather_object(gathered, log)
    if rank == 0:
        torch.save(gathered, out_path)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_a_failing_rank_poisons_its_record_and_every_rank_fails_the_batch(tmp_path):
    world, n_total, d, k, ef = 2, 600, 16, 5, 24
    out = str(tmp_path / "log.pt")
    mp.spawn(_poison_worker, args=(world, _free_port(), n_total, d, k, ef, out), nprocs=world, join=True)
    r0, r1 = torch.load(out, weights_only=False)
    assert r0[0][0] == "ok" and r1[0] == r0[0]          # before: answered, the same on both ranks
    assert r1[1][:2] == ("error", "NodeNotFound")        # the failing rank: its own error
    assert r0[1][:2] == ("error", "SearchError") and "rank(s) [1]" in r0[1][2]  # the other: the peer's failure
    assert r0[2][0] == "ok" and r1[2] == r0[2]          # after: the next batch succeeds, nobody hung
