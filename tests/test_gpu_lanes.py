"""Search lanes: isl_index_prepare (no allocation on the search path afterwards), the pipelined
host-buffer entry point isl_search_batch_async (search.rs:150-181 / indexer/service.rs:781-785:
host slices in, Vec out), per-call statistics, the shared scratch pool of the heap-exact kernel
under concurrent launches, and the guards around provider swaps.  Everything through the C ABI,
results against the CPU oracle (ids and f32 bit patterns)."""
import threading

import numpy as np
import pytest

import islands_amd as ia
from _data import clustered_vectors, knn_graph
from test_gpu_parity import bits, make_index

pytestmark = pytest.mark.gpu


def _case(orc, n=3000, d=96, deg=20, seed=5):
    v = clustered_vectors(n, d, seed)
    off, nb = knn_graph(v, deg, seed=seed + 1)
    csr = orc.Csr(off, nb, entry_point=0)
    return v, csr


def _check(orc, csr, v, q, k, ef, ids, dist, cnt):
    for i in range(q.shape[0]):
        r = orc.leann_search(csr, v, q[i], k, ef)
        n = int(cnt[i])
        assert ids[i, :n].tolist() == r.ids.tolist(), i
        assert bits(dist[i, :n]).tolist() == bits(r.dist).tolist(), i


def test_prepare_then_no_allocations(orc):
    """After isl_index_prepare no search call allocates, creates a stream or event -- whichever
    entry point it comes through and however many are in flight."""
    v, csr = _case(orc)
    idx = make_index(csr, v)
    nq, k, ef, lanes = 48, 10, 64, 6
    idx.prepare(nq, ef, k, lanes)
    qs = [clustered_vectors(nq, v.shape[1], 100 + b) for b in range(lanes)]
    # host buffers, synchronous
    ids, dist, cnt = idx.search_batch(qs[0], k, ef)
    st = idx.last_stats()
    assert st["allocations"] == 0 and st["queries"] == nq
    _check(orc, csr, v, qs[0], k, ef, ids, dist, cnt)
    # host buffers, `lanes` calls in flight, waited out of order
    toks = [idx.search_batch_async(q, k, ef) for q in qs]
    outs = [idx._pending[t] for t in toks]
    stats = {}
    for j in reversed(range(lanes)):
        stats[j] = idx.wait_stats(toks[j])
    for j in range(lanes):
        assert stats[j]["allocations"] == 0, stats[j]
        assert stats[j]["queries"] == nq
        _check(orc, csr, v, qs[j], k, ef, *outs[j])
    # smaller batches / smaller ef / smaller k fit the prepared lanes too
    ids, dist, cnt = idx.search_batch(qs[1][:7], 3, 16)
    assert idx.last_stats()["allocations"] == 0
    _check(orc, csr, v, qs[1][:7], 3, 16, ids, dist, cnt)


def test_unprepared_calls_report_their_setup(orc):
    """Without prepare the first call on a lane sets it up and says so; the second does not."""
    v, csr = _case(orc, n=1500, seed=9)
    idx = make_index(csr, v)
    q = clustered_vectors(16, v.shape[1], 3)
    idx.search_batch(q, 5, 32)
    first = idx.last_stats()["allocations"]
    idx.search_batch(q, 5, 32)
    assert first > 0 and idx.last_stats()["allocations"] == 0
    # a bigger batch outgrows the lane once
    q2 = clustered_vectors(2000, v.shape[1], 4)
    ids, dist, cnt = idx.search_batch(q2, 5, 32)
    assert idx.last_stats()["allocations"] > 0
    _check(orc, csr, v, q2[:40], 5, 32, ids, dist, cnt)


def test_async_host_stats_are_per_call(orc):
    """Two calls of different size in flight: each token gets its own counters (the old
    isl_search_last_stats handed out 'the most recent call' of the index)."""
    v, csr = _case(orc, seed=21)
    idx = make_index(csr, v)
    idx.prepare(64, 64, 10, 4)
    qa, qb = clustered_vectors(64, v.shape[1], 31), clustered_vectors(5, v.shape[1], 32)
    ta = idx.search_batch_async(qa, 10, 64)
    tb = idx.search_batch_async(qb, 10, 32)
    sb = idx.wait_stats(tb)
    sa = idx.wait_stats(ta)
    assert sa["queries"] == 64 and sb["queries"] == 5
    ea = sum(orc.leann_search(csr, v, q, 10, 64).counters["evals"] for q in qa)
    eb = sum(orc.leann_search(csr, v, q, 10, 32).counters["evals"] for q in qb)
    assert sa["evals"] == ea and sb["evals"] == eb
    with pytest.raises(ia.CoreError):  # a token completes once
        idx.wait(ta)
    assert idx.search_batch_async(np.zeros((0, v.shape[1]), np.float32), 10, 64) == 0  # nothing to wait for


def test_last_stats_is_per_thread(orc):
    v, csr = _case(orc, n=1200, seed=40)
    idx = make_index(csr, v)
    idx.prepare(32, 32, 5, 4)
    q = clustered_vectors(32, v.shape[1], 41)
    seen = {}

    def work(name, n):
        idx.search_batch(q[:n], 5, 32)
        seen[name] = idx.last_stats()["queries"]

    ths = [threading.Thread(target=work, args=(i, n)) for i, n in enumerate([3, 17, 32, 9])]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert seen == {0: 3, 1: 17, 2: 32, 3: 9}


def test_exact_pool_shared_by_concurrent_launches(orc):
    """Rows longer than 128 ids send every query to the heap-exact kernel; eight such calls in
    flight share the one scratch pool (32 slots) and still answer like the oracle."""
    rng = np.random.default_rng(77)
    n, d, deg = 900, 48, 150
    v = clustered_vectors(n, d, 78)
    nb = np.stack([rng.permutation(n)[:deg] for _ in range(n)]).astype(np.uint64)
    off = (np.arange(n + 1) * deg).astype(np.uint64)
    csr = orc.Csr(off, nb.reshape(-1), entry_point=0)
    idx = make_index(csr, v)
    idx.prepare(24, 40, 8, 8)
    qs = [clustered_vectors(24, d, 200 + b) for b in range(8)]
    toks = [idx.search_batch_async(q, 8, 40) for q in qs]
    outs = [idx._pending[t] for t in toks]
    for j, t in enumerate(toks):
        st = idx.wait_stats(t)
        assert st["exact_path"] == 24 and st["allocations"] == 0
        _check(orc, csr, v, qs[j], 8, 40, *outs[j])


def test_provider_swap_refused_while_searching(orc):
    import torch

    v, csr = _case(orc, n=2500, seed=50)
    idx = make_index(csr, v)
    idx.prepare(256, 64, 10, 2)
    q = torch.from_numpy(clustered_vectors(256, v.shape[1], 51)).cuda()
    ids = torch.zeros((256, 10), dtype=torch.int64, device="cuda")
    dist = torch.zeros((256, 10), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(256, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    tok = idx.search_batch_device_async(q.data_ptr(), 256, v.shape[1], 10, 64, ids.data_ptr(),
                                        dist.data_ptr(), cnt.data_ptr())
    with pytest.raises(ia.CoreError) as e:
        idx.set_embeddings(v)
    assert e.value.kind == "SearchError"
    idx.wait(tok)
    idx.set_embeddings(v)  # fine once nothing is in flight
    ids2, dist2, cnt2 = idx.search_batch(q.cpu().numpy(), 10, 64)
    assert ids.cpu().numpy().astype(np.uint64).tolist() == ids2.tolist()


def test_bf16_rows_then_recompute_provider_drops_the_bf16_table(orc):
    """set_embeddings_bf16 followed by set_recompute_provider used to leave the bf16 table in
    place as the one the searches read."""
    from test_gpu_encoder import _recompute_case

    cfg, enc, tok, lens, emb = _recompute_case(orc, n=400, seed=9)
    csr = orc.leann_build(emb, m=6, m0=12, ef_construction=30, levels=np.zeros(400, np.uint64))
    q = emb[::41] + np.float32(0.01)
    want = make_index(csr, emb).search_batch(q, 5, 32)
    idx = make_index(csr, emb)
    idx.set_embeddings_bf16((emb.view(np.uint32) >> 16).astype(np.uint16))
    idx.set_recompute_provider(enc, tok, lens)
    got = idx.search_batch(q, 5, 32)
    assert got[0].tolist() == want[0].tolist() and bits(got[1]).tolist() == bits(want[1]).tolist()
