"""GPU: the recompute encoder (isl_encoder_*) against the golden fixture from HuggingFace's
BertModel and against the numpy oracle (oracle/bert_ref.py) on configurations the fixture does
not cover.  Float32 everywhere; tolerances are written next to each assertion."""
import os

import numpy as np
import pytest

import bert_ref
import islands_amd as ia

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bert_tiny.npz")


def to_cfg(c):
    return ia.BertConfig(vocab_size=c["vocab_size"], hidden=c["hidden"], layers=c["layers"],
                         heads=c["heads"], intermediate=c["intermediate"],
                         max_position=c["max_position"], type_vocab=c["type_vocab"],
                         layer_norm_eps=c.get("layer_norm_eps", 1e-12),
                         gelu_tanh=bool(c.get("gelu_tanh", False)))


def test_matches_hf_golden_fixture():
    z = np.load(GOLD)
    cfg = {k[5:]: z[k].item() for k in z.files if k.startswith("cfg::")}
    w = {k[3:]: z[k] for k in z.files if k.startswith("w::")}
    enc = ia.CandleEmbedder(to_cfg(cfg), w)
    hid = enc.forward(z["input_ids"], z["token_type_ids"], z["attention_mask"])
    # O(1) activations after LayerNorm; 3e-5 absolute covers float32 reassociation over K <= 160
    assert np.abs(hid - z["hidden"]).max() < 3e-5, np.abs(hid - z["hidden"]).max()
    emb = enc.embed(z["input_ids"], z["token_type_ids"], z["attention_mask"])
    want = bert_ref.mean_pool_normalize(z["hidden"], z["attention_mask"], True)
    assert np.abs(emb - want).max() < 1e-5
    assert np.all(np.abs(np.linalg.norm(emb, axis=1) - 1) < 1e-5)


@pytest.mark.parametrize("name,cfg,B,L", [
    ("minilm-l6", dict(vocab_size=500, hidden=384, layers=6, heads=12, intermediate=1536,
                       max_position=128, type_vocab=2), 5, 70),          # MiniLM-L6 shape, dh = 32
    ("config3", dict(vocab_size=300, hidden=768, layers=2, heads=12, intermediate=3072,
                     max_position=64, type_vocab=2), 3, 64),             # BASELINE config 3 shape, dh = 64
    ("tanh", dict(vocab_size=50, hidden=32, layers=1, heads=2, intermediate=44,
                  max_position=16, type_vocab=1, gelu_tanh=True), 2, 9),  # dh = 16, ragged tile edges
])
def test_matches_numpy_oracle(name, cfg, B, L):
    w = bert_ref.random_weights(cfg, seed=45, std=0.08)
    rng = np.random.default_rng(44)
    lens = rng.integers(1, L + 1, B)
    lens[0] = L
    ids, tt, mask = bert_ref.pad_batch([rng.integers(1, cfg["vocab_size"], n).tolist() for n in lens])
    enc = ia.CandleEmbedder(to_cfg(cfg), w, normalize=True)
    hid = enc.forward(ids, tt, mask)
    want = bert_ref.bert_forward(cfg, w, ids, tt, mask)
    err = np.abs(hid - want).max()
    assert err < 1e-4, (name, err)  # K up to 3072 in float32
    emb = enc.embed(ids, tt, mask)
    ewant = bert_ref.mean_pool_normalize(want, mask, True)
    assert np.abs(emb - ewant).max() < 2e-5, (name, np.abs(emb - ewant).max())


def test_embedding_independent_of_batch_composition():
    """A node's embedding must not depend on what it is batched with (the recompute provider
    re-encodes nodes in whatever batch a hop produces)."""
    cfg = dict(vocab_size=200, hidden=128, layers=2, heads=4, intermediate=256, max_position=32, type_vocab=2)
    w = bert_ref.random_weights(cfg, seed=3, std=0.1)
    enc = ia.CandleEmbedder(to_cfg(cfg), w)
    rng = np.random.default_rng(1)
    seqs = [rng.integers(1, 200, 20).tolist() for _ in range(9)]
    ids, tt, mask = bert_ref.pad_batch(seqs)
    all_at_once = enc.embed(ids, tt, mask)
    for i in (0, 4, 8):
        alone = enc.embed(ids[i:i + 1], tt[i:i + 1], mask[i:i + 1])
        assert alone.view(np.uint32).tolist() == all_at_once[i:i + 1].view(np.uint32).tolist()


def test_encoder_errors():
    cfg = ia.BertConfig(vocab_size=10, hidden=32, layers=1, heads=2, intermediate=64, max_position=8, type_vocab=1)
    enc = ia.CandleEmbedder(cfg)
    with pytest.raises(ia.CoreError) as e:
        enc.set_weight("encoder.layer.0.output.dense.bias", np.zeros(7, np.float32))
    assert e.value.kind == "DimensionMismatch" and (e.value.expected, e.value.actual) == (32, 7)
    with pytest.raises(ia.CoreError) as e:
        enc.embed(np.full((1, 4), 10, np.int64))  # id == vocab_size
    assert e.value.kind == "EmbeddingError"
    with pytest.raises(ia.CoreError) as e:
        enc.embed(np.zeros((1, 9), np.int64))  # longer than max_position
    assert e.value.kind == "EmbeddingError"
    with pytest.raises(ia.CoreError) as e:
        ia.CandleEmbedder(ia.BertConfig(hidden=30, heads=4))
    assert e.value.kind == "InvalidConfig"
    assert enc.embed(np.zeros((0, 4), np.int64)).shape == (0, 32)


# ------------------------------------------------- recompute provider (leann.rs:82-99)
def _recompute_case(orc, n=1200, L=12, seed=5, min_len=3):
    cfg = dict(vocab_size=400, hidden=64, layers=2, heads=4, intermediate=128, max_position=16, type_vocab=2)
    w = bert_ref.random_weights(cfg, seed=45, std=0.2)
    enc = ia.CandleEmbedder(to_cfg(cfg), w, normalize=True)
    rng = np.random.default_rng(seed)
    # "documents" = a topic prefix plus noise tokens, so that embeddings cluster
    topics = rng.integers(1, 400, (24, 6))
    tok = np.zeros((n, L), np.uint16)
    lens = rng.integers(min_len, L + 1, n).astype(np.uint16)  # (short rows are bare topic prefixes: equal embeddings)
    for i in range(n):
        row = np.concatenate([topics[rng.integers(0, 24)], rng.integers(1, 400, L - 6)])
        tok[i] = row
        tok[i, lens[i]:] = 0
    ids = tok.astype(np.int64)
    mask = (np.arange(L)[None, :] < lens[:, None]).astype(np.float32)
    emb = np.concatenate([enc.embed(ids[o:o + 256], None, mask[o:o + 256]) for o in range(0, n, 256)])
    return cfg, enc, tok, lens, emb


def test_recompute_provider_equals_in_memory_provider(orc):
    """The search over embeddings recomputed on the fly must return what the search over the
    same embeddings held in memory returns (which the parity tests pin to the oracle)."""
    cfg, enc, tok, lens, emb = _recompute_case(orc)
    n = emb.shape[0]
    levels = np.zeros(n, np.uint64)
    levels[0] = 2
    csr = orc.leann_build(emb, m=8, m0=16, ef_construction=40, levels=levels)
    g = ia.CsrGraph(node_offsets=csr.node_offsets, neighbors=csr.neighbors, levels=csr.levels,
                    entry_point=csr.entry_point, max_level=csr.max_level, num_nodes=csr.num_nodes,
                    degree_counts=csr.degree_counts)
    q = emb[::97] + np.float32(0.01)
    mem_idx = ia.LeannIndex.from_csr(g, None, dimension=64)
    mem_idx.upload(0)
    mem_idx.set_embeddings(emb)
    want = mem_idx.search_batch(q, 10, 48)
    want_stats = mem_idx.last_stats()
    rec_idx = ia.LeannIndex.from_csr(g, None, dimension=64)
    rec_idx.upload(0)
    rec_idx.set_recompute_provider(enc, tok, lens)
    assert rec_idx.is_recompute()
    for rep in range(2):  # keep_rows = False: the second call recomputes everything again
        got = rec_idx.search_batch(q, 10, 48)
        st = rec_idx.last_stats()
        assert got[2].tolist() == want[2].tolist()
        assert got[0].tolist() == want[0].tolist()
        assert got[1].view(np.uint32).tolist() == want[1].view(np.uint32).tolist()
        for f in ("expansions", "edges", "evals", "pushes"):
            assert st[f] == want_stats[f], f
        assert st["recompute_rounds"] > 2
        # every node is encoded once per call, however many queries visit it
        assert 0 < st["encoded_nodes"] <= min(want_stats["evals"], n)
    # and against the oracle directly
    for i in range(q.shape[0]):
        r = orc.leann_search(csr, emb, q[i], 10, 48)
        c = int(got[2][i])
        assert got[0][i, :c].tolist() == r.ids.tolist()


def test_recompute_provider_small_row_cache(orc):
    """Recompute mode stores no embeddings (leann.rs:366-371): the provider's row cache is a bounded
    slab.  With room for a quarter of the nodes rows get evicted and re-encoded inside one call,
    parked queries resume where they stopped -- and ids, distance bits and counters still equal the
    in-memory provider's.  ef = 300 and rows past 64 ids take the other instantiations."""
    # rows long enough to carry noise tokens: no two nodes share an embedding.  (Equal distances send
    # a query to the heap-exact kernel, which re-runs a blocked query from its start and therefore
    # needs the rows of its whole traversal in the cache -- the default cache, not a 400-row one.)
    cfg, enc, tok, lens, emb = _recompute_case(orc, n=1600, seed=11, min_len=9)
    n = emb.shape[0]
    from _data import random_csr
    q = emb[::53] + np.float32(0.02)
    for (deg, ef, rows) in ((20, 48, 512), (90, 200, 1024)):
        off, nb = random_csr(n, deg, 3)
        csr = orc.Csr(off, nb, entry_point=5)
        g = ia.CsrGraph(node_offsets=csr.node_offsets, neighbors=csr.neighbors, levels=csr.levels, entry_point=5,
                        num_nodes=n, degree_counts=csr.degree_counts)
        mem_idx = ia.LeannIndex.from_csr(g, None, dimension=64).upload(0)
        mem_idx.set_embeddings(emb)
        want = mem_idx.search_batch(q, 10, ef)
        want_stats = mem_idx.last_stats()
        rec_idx = ia.LeannIndex.from_csr(g, None, dimension=64).upload(0)
        rec_idx.set_recompute_provider(enc, tok, lens, cache_rows=rows)
        assert rows * 64 * 4 <= rec_idx.recompute_cache_bytes() < n * 64 * 4  # less than the dense table
        got = rec_idx.search_batch(q, 10, ef)
        st = rec_idx.last_stats()
        assert got[2].tolist() == want[2].tolist() and got[0].tolist() == want[0].tolist()
        assert got[1].view(np.uint32).tolist() == want[1].view(np.uint32).tolist()
        for f in ("expansions", "edges", "evals", "pushes"):
            assert st[f] == want_stats[f], f
        assert st["encoded_nodes"] >= rows  # the slab turned over at least once
        # with room for every row a parked query advances one hop per round: the rounds follow the
        # longest query of the batch (+ the entry point's round), whatever the batch size
        rec_idx.set_recompute_provider(enc, tok, lens)
        got = rec_idx.search_batch(q, 10, ef)
        st = rec_idx.last_stats()
        assert got[0].tolist() == want[0].tolist()
        longest = max(orc.leann_search(csr, emb, q[i], 10, ef).counters["expansions"] for i in range(q.shape[0]))
        assert st["recompute_rounds"] <= longest + 2, (st["recompute_rounds"], longest)
        assert st["encoded_nodes"] <= min(want_stats["evals"], n)  # each node once per call


def test_recompute_fuzz_slice(orc):
    """A fixed slice of tests/fuzz_parity.py --mode recompute (random graphs incl. rows past 64 ids,
    metrics, ef, cache sizes down to the 256-row floor, tied embeddings, keep_rows)."""
    import fuzz_parity
    rng = np.random.default_rng(2024)
    for case in range(10):
        fuzz_parity.recompute_case(rng, case)


def test_recompute_rounds_encode_in_quanta_and_answer_the_same(orc, monkeypatch):
    """The rounds hand the encoder whole multiples of a batch quantum (whole waves of GEMM tiles on the
    chip) and report the left-over misses again next round: more rounds, the same nodes encoded once each,
    the same answers bit for bit -- with the plain and the two-level search, and with a bounded row cache."""
    cfg, enc, tok, lens, emb = _recompute_case(orc, n=1600, seed=21, min_len=9)
    n = emb.shape[0]
    from _data import random_csr
    q = emb[::29] + np.float32(0.02)
    off, nb = random_csr(n, 24, 5)
    csr = orc.Csr(off, nb, entry_point=3)
    g = ia.CsrGraph(node_offsets=csr.node_offsets, neighbors=csr.neighbors, levels=csr.levels, entry_point=3,
                    num_nodes=n, degree_counts=csr.degree_counts)
    mem_idx = ia.LeannIndex.from_csr(g, None, dimension=64).upload(0)
    mem_idx.set_embeddings(emb)
    want = mem_idx.search_batch(q, 10, 64)
    want_stats = mem_idx.last_stats()
    runs = {}
    for quantum in ("0", "16", "100"):
        monkeypatch.setenv("ISL_RECOMPUTE_QUANTUM", quantum)
        for rows in (None, 512):
            rec_idx = ia.LeannIndex.from_csr(g, None, dimension=64).upload(0)
            if rows:
                rec_idx.set_recompute_provider(enc, tok, lens, cache_rows=rows)
            else:
                rec_idx.set_recompute_provider(enc, tok, lens)
            got = rec_idx.search_batch(q, 10, 64)
            st = rec_idx.last_stats()
            assert got[0].tolist() == want[0].tolist() and got[2].tolist() == want[2].tolist(), (quantum, rows)
            assert got[1].view(np.uint32).tolist() == want[1].view(np.uint32).tolist()
            for f in ("expansions", "edges", "evals", "pushes"):
                assert st[f] == want_stats[f], (f, quantum, rows)
            runs[(quantum, rows)] = st
    # with a row for every node each node is encoded once whatever the quantum; quanta take more rounds
    assert runs[("16", None)]["encoded_nodes"] == runs[("0", None)]["encoded_nodes"]
    assert runs[("16", None)]["recompute_rounds"] >= runs[("0", None)]["recompute_rounds"]
    assert runs[("100", None)]["encoded_nodes"] == runs[("0", None)]["encoded_nodes"]


def test_recompute_encoder_passes_split_in_two_answer_the_same(orc, monkeypatch):
    """encoder_embed_nodes runs the two halves of a pass side by side on two streams with a workspace each
    (ISL_ENCODER_SPLIT = the batch size from which it does; unset or 0 = never): the embeddings, and with them every
    answer, distance bit and counter, are those of the whole pass and of the in-memory provider."""
    cfg, enc, tok, lens, emb = _recompute_case(orc, n=1600, seed=23, min_len=5)
    n = emb.shape[0]
    from _data import random_csr
    q = emb[::17] + np.float32(0.015)
    off, nb = random_csr(n, 30, 9)
    csr = orc.Csr(off, nb, entry_point=11)
    g = ia.CsrGraph(node_offsets=csr.node_offsets, neighbors=csr.neighbors, levels=csr.levels, entry_point=11,
                    num_nodes=n, degree_counts=csr.degree_counts)
    mem_idx = ia.LeannIndex.from_csr(g, None, dimension=64).upload(0)
    mem_idx.set_embeddings(emb)
    want = mem_idx.search_batch(q, 10, 96)
    want_stats = mem_idx.last_stats()
    encoded = {}
    for split in ("0", "8", "50", "256", None):
        if split is None:
            monkeypatch.delenv("ISL_ENCODER_SPLIT", raising=False)
        else:
            monkeypatch.setenv("ISL_ENCODER_SPLIT", split)
        rec_idx = ia.LeannIndex.from_csr(g, None, dimension=64).upload(0)
        rec_idx.set_recompute_provider(enc, tok, lens)
        got = rec_idx.search_batch(q, 10, 96)
        st = rec_idx.last_stats()
        assert got[0].tolist() == want[0].tolist() and got[2].tolist() == want[2].tolist(), split
        assert got[1].view(np.uint32).tolist() == want[1].view(np.uint32).tolist(), split
        for f in ("expansions", "edges", "evals", "pushes"):
            assert st[f] == want_stats[f], (f, split)
        encoded[split] = st["encoded_nodes"]
    assert len(set(encoded.values())) == 1


def test_concurrent_recompute_calls_are_answered_together(orc):
    """Asynchronous device-buffer calls over the recompute provider queue for their turn; the call that gets it
    answers every compatible call waiting at that moment together with its own, in one set of rounds over the
    union of their queries (recompute_coalesced): each call's ids, distance bits, counts and counters are those
    of the in-memory provider, calls with another ef are not merged, and the calls that were merged report the
    same rounds."""
    import torch
    cfg, enc, tok, lens, emb = _recompute_case(orc, n=1600, seed=31, min_len=7)
    n, d = emb.shape
    from _data import random_csr
    off, nb = random_csr(n, 24, 13)
    csr = orc.Csr(off, nb, entry_point=2)
    g = ia.CsrGraph(node_offsets=csr.node_offsets, neighbors=csr.neighbors, levels=csr.levels, entry_point=2,
                    num_nodes=n, degree_counts=csr.degree_counts)
    mem_idx = ia.LeannIndex.from_csr(g, None, dimension=d).upload(0)
    mem_idx.set_embeddings(emb)
    rec_idx = ia.LeannIndex.from_csr(g, None, dimension=d).upload(0)
    rec_idx.set_recompute_provider(enc, tok, lens)
    rng = np.random.default_rng(3)
    calls = []
    for i, (nq, ef) in enumerate([(40, 64), (33, 64), (57, 64), (20, 32), (41, 64), (9, 64)]):
        q = (emb[rng.integers(0, n, nq)] + rng.standard_normal((nq, d)).astype(np.float32) * np.float32(0.03)).astype(np.float32)
        calls.append((q, 10, ef))
    dev = torch.device("cuda:0")
    bufs, toks = [], []
    for (q, k, ef) in calls:
        dq = torch.from_numpy(q).to(dev)
        o = (torch.zeros((q.shape[0], k), dtype=torch.int64, device=dev), torch.zeros((q.shape[0], k), dtype=torch.float32, device=dev),
             torch.zeros(q.shape[0], dtype=torch.int32, device=dev))
        bufs.append((dq, o))
    torch.cuda.synchronize()
    for (q, k, ef), (dq, o) in zip(calls, bufs):
        toks.append(rec_idx.search_batch_device_async(dq.data_ptr(), q.shape[0], d, k, ef, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr()))
    stats = [rec_idx.wait_stats(t) for t in toks]
    torch.cuda.synchronize()
    for (q, k, ef), (dq, o), st in zip(calls, bufs, stats):
        want = mem_idx.search_batch(q, k, ef)
        ws = mem_idx.last_stats()
        assert o[2].cpu().numpy().astype(np.uint32).tolist() == want[2].tolist()
        assert o[0].cpu().numpy().astype(np.uint64).tolist() == want[0].tolist()
        assert o[1].cpu().numpy().view(np.uint32).tolist() == want[1].view(np.uint32).tolist()
        for f in ("expansions", "edges", "evals", "pushes"):
            assert st[f] == ws[f], (f, q.shape[0], ef)
        assert st["queries"] == q.shape[0]
    # the first call had the turn alone or with whatever had arrived; the ef = 64 calls behind it were answered together
    merged_rounds = {st["recompute_rounds"] for (q, k, ef), st in zip(calls[1:], stats[1:]) if ef == 64}
    assert len(merged_rounds) <= 2
    # the host-buffer form (isl_search_batch_async: pageable arrays in and out) goes through the same turn
    outs = [(np.zeros((q.shape[0], k), np.uint64), np.zeros((q.shape[0], k), np.float32), np.zeros(q.shape[0], np.uint32))
            for (q, k, ef) in calls]
    toks = [rec_idx.search_batch_async(q, k, ef, out=o) for (q, k, ef), o in zip(calls, outs)]
    hstats = [rec_idx.wait_stats(t) for t in toks]
    for (q, k, ef), o, st in zip(calls, outs, hstats):
        want = mem_idx.search_batch(q, k, ef)
        ws = mem_idx.last_stats()
        assert o[2].tolist() == want[2].tolist() and o[0].tolist() == want[0].tolist(), (q.shape[0], ef)
        assert o[1].view(np.uint32).tolist() == want[1].view(np.uint32).tolist()
        for f in ("expansions", "edges", "evals", "pushes"):
            assert st[f] == ws[f], (f, q.shape[0], ef)


def test_concurrent_recompute_calls_that_fail_get_their_own_errors(orc):
    """A graph that names a node the provider has no tokens for: every search that reaches it fails with
    NodeNotFound (leann.rs:145-150).  Calls answered together must not share that fate blindly -- when the union
    fails, every member is run by itself: each call's outcome (answers or error) is what the same call gives alone."""
    import torch
    cfg, enc, tok, lens, emb = _recompute_case(orc, n=400, seed=5, min_len=7)
    n, d = emb.shape
    from _data import random_csr
    off, nb = random_csr(n, 8, 21)
    nb = nb.copy()
    nb[off[37]] = n + 5  # node 37's first neighbour does not exist
    csr = orc.Csr(off, nb, entry_point=1)
    g = ia.CsrGraph(node_offsets=csr.node_offsets, neighbors=csr.neighbors, levels=csr.levels, entry_point=1,
                    num_nodes=n, degree_counts=csr.degree_counts)
    rec_idx = ia.LeannIndex.from_csr(g, None, dimension=d).upload(0)
    rec_idx.set_recompute_provider(enc, tok, lens)
    rng = np.random.default_rng(8)
    qs = [(emb[rng.integers(0, n, m)] + np.float32(0.01)).astype(np.float32) for m in (12, 7, 30, 5)]
    alone = []
    for q in qs:  # what each call gives by itself
        try:
            alone.append(("ok", rec_idx.search_batch(q, 5, 6)))
        except ia.CoreError as e:
            alone.append((e.kind, e.node))
    assert any(a[0] == "NodeNotFound" for a in alone)
    dev = torch.device("cuda:0")
    bufs = []
    for q in qs:
        bufs.append((torch.from_numpy(q).to(dev), torch.zeros((q.shape[0], 5), dtype=torch.int64, device=dev),
                     torch.zeros((q.shape[0], 5), dtype=torch.float32, device=dev), torch.zeros(q.shape[0], dtype=torch.int32, device=dev)))
    torch.cuda.synchronize()
    toks = [rec_idx.search_batch_device_async(b[0].data_ptr(), q.shape[0], d, 5, 6, b[1].data_ptr(), b[2].data_ptr(), b[3].data_ptr())
            for q, b in zip(qs, bufs)]
    for q, b, t, a in zip(qs, bufs, toks, alone):
        try:
            rec_idx.wait(t)
            got = ("ok",)
        except ia.CoreError as e:
            got = (e.kind, e.node)
        if a[0] == "ok":
            assert got == ("ok",)
            torch.cuda.synchronize()
            assert b[1].cpu().numpy().astype(np.uint64).tolist() == a[1][0].tolist()
            assert b[2].cpu().numpy().view(np.uint32).tolist() == a[1][1].view(np.uint32).tolist()
        else:
            assert got == a


def test_recompute_provider_keeps_rows_when_asked(orc):
    cfg, enc, tok, lens, emb = _recompute_case(orc, n=400, seed=9)
    levels = np.zeros(400, np.uint64)
    csr = orc.leann_build(emb, m=6, m0=12, ef_construction=30, levels=levels)
    g = ia.CsrGraph(node_offsets=csr.node_offsets, neighbors=csr.neighbors, levels=csr.levels,
                    entry_point=csr.entry_point, max_level=csr.max_level, num_nodes=csr.num_nodes,
                    degree_counts=csr.degree_counts)
    idx = ia.LeannIndex.from_csr(g, None, dimension=64)
    idx.upload(0)
    idx.set_recompute_provider(enc, tok, lens, keep_rows=True)
    a = idx.search_batch(emb[:8], 5, 20)
    first = idx.last_stats()
    b = idx.search_batch(emb[:8], 5, 20)
    second = idx.last_stats()
    assert a[0].tolist() == b[0].tolist() and first["encoded_nodes"] > 0
    assert second["encoded_nodes"] == 0 and second["recompute_rounds"] == 1


# --------------------------------------------- indexer/service.rs:747-801, device side end to end
def test_service_search_composition(orc):
    import test_gpu_hnsw as th
    cfg = dict(vocab_size=300, hidden=64, layers=2, heads=4, intermediate=128, max_position=16, type_vocab=2)
    w = bert_ref.random_weights(cfg, seed=45, std=0.2)
    enc = ia.CandleEmbedder(to_cfg(cfg), w, normalize=True)
    rng = np.random.default_rng(12)
    indexes, oracles = [], []
    for name, n, seed in (("repo-a", 300, 1), ("repo-b", 200, 2)):
        tok = rng.integers(1, 300, (n, 10))
        emb = enc.embed(tok.astype(np.int64))
        _, h, g = th.build(orc, n, 64, seed, m=8, m0=16, ef_construction=60, vectors=emb)
        files = n - 7  # the last ids have no file entry (stored.files.get(id) == None, :788)
        indexes.append((name, g, files))
        oracles.append((name, h, files))
    qtok = rng.integers(1, 300, 10)
    got = ia.service_search(enc, indexes, qtok, np.ones(10, np.float32), top_k=5)
    q = enc.embed(qtok.reshape(1, -1).astype(np.int64))[0]
    li, ld = [], []
    for _, h, files in oracles:
        r = h.search(q, 5, 100)  # ef = max(top_k, 100), :781
        keep = r.ids < files
        li.append(r.ids[keep])
        ld.append(r.dist[keep])
    st, ei, es, esrc = orc.service_merge(li, ld, 5)
    assert st == 0 and len(got) == ei.size
    assert [(n, i) for _, n, i in got] == [(oracles[int(s)][0], int(i)) for s, i in zip(esrc, ei)]
    assert np.array([s for s, _, _ in got], np.float32).view(np.uint32).tolist() == es.view(np.uint32).tolist()


def test_bf16_linear_mode_is_close_and_switchable():
    """Optional bf16 mode of the Linear layers: embeddings within 3e-2 (absolute, unit vectors) of
    the float32 ones, cosine similarity to them above 0.999; switching back restores the float32
    bits."""
    cfg = dict(vocab_size=500, hidden=384, layers=6, heads=12, intermediate=1536, max_position=128, type_vocab=2)
    w = bert_ref.random_weights(cfg, seed=45, std=0.08)
    enc = ia.CandleEmbedder(to_cfg(cfg), w, normalize=True)
    rng = np.random.default_rng(5)
    ids, tt, mask = bert_ref.pad_batch([rng.integers(1, 500, n).tolist() for n in (70, 12, 33, 70)])
    f32 = enc.embed(ids, tt, mask)
    enc.set_precision(bf16=True)
    b16 = enc.embed(ids, tt, mask)
    assert not np.array_equal(f32, b16)
    assert np.abs(f32 - b16).max() < 3e-2, np.abs(f32 - b16).max()
    assert np.all((f32 * b16).sum(1) > 0.999)
    enc.set_precision(bf16=False)
    assert np.array_equal(enc.embed(ids, tt, mask).view(np.uint32), f32.view(np.uint32))


def test_recompute_indexes_release_their_device_memory(orc):
    """isl_index_free must give back everything a recompute index's lanes hold (parked query state,
    flags, lists, slot arrays, pinned lists): create / search / free in a loop and watch the
    device's free memory (hipMemGetInfo through torch)."""
    import torch
    cfg, enc, tok, lens, emb = _recompute_case(orc, n=1200, seed=3, min_len=9)
    n = emb.shape[0]
    from _data import random_csr
    off, nb = random_csr(n, 16, 3)
    csr = orc.Csr(off, nb, entry_point=0)
    g = ia.CsrGraph(node_offsets=csr.node_offsets, neighbors=csr.neighbors, levels=csr.levels, entry_point=0,
                    num_nodes=n, degree_counts=csr.degree_counts)
    q = emb[::11] + np.float32(0.01)

    def one():
        idx = ia.LeannIndex.from_csr(g, None, dimension=64).upload(0)
        idx.set_recompute_provider(enc, tok, lens)
        idx.prepare(q.shape[0], 128, 10, 4)
        idx.search_batch(q, 10, 128)
        idx.close()

    one()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    for _ in range(6):
        one()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info(0)[0]
    # four prepared lanes hold ~15 MB of parked-query state each here: a leak of that would show as
    # several hundred MB over six indexes
    assert free0 - free1 < 32 << 20, (free0, free1)


def test_recompute_heap_exact_kernel_parks_and_resumes(orc):
    """Queries answered by the heap-exact kernel -- ef beyond 512 (no traversal kernel in front), rows past
    128 ids, equal embeddings (ties hand a query over) -- over a row cache far smaller than their traversal:
    round 2 re-ran a blocked query from its start and failed with "row cache too small"; now the query
    parks in its slot of the scratch pool and resumes in the hop it stopped at.  ids, distance bits and
    counters equal the in-memory provider's."""
    from _data import random_csr
    # (the hand-over of single queries on ties is covered by fuzz_parity.py --mode recompute: copied nodes,
    # 256-row caches, several hundred queries through the heap-exact kernel per run)
    for (seed, dup, deg, ef, rows) in ((11, False, 20, 600, 512), (7, False, 150, 100, 700)):
        cfg, enc, tok, lens, emb = _recompute_case(orc, n=1600, seed=seed, min_len=9)
        n = emb.shape[0]
        if dup:  # 300 copies of one node: more equal distances at the edge of the result set than the
            # traversal kernel's tie list holds -> those queries are handed to the heap-exact kernel
            tok[100:400] = tok[0]
            lens[100:400] = lens[0]
            emb[100:400] = emb[0]   # (an embedding does not depend on what it is batched with)
        off, nb = random_csr(n, deg, 3)
        csr = orc.Csr(off, nb, entry_point=5)
        g = ia.CsrGraph(node_offsets=csr.node_offsets, neighbors=csr.neighbors, levels=csr.levels, entry_point=5,
                        num_nodes=n, degree_counts=csr.degree_counts)
        q = emb[::67] + np.float32(0.02)
        if dup:
            q[:8] = emb[0] + (np.arange(8, dtype=np.float32)[:, None] + 1) * np.float32(0.003)  # queries next to the copies
        mem_idx = ia.LeannIndex.from_csr(g, None, dimension=64).upload(0)
        mem_idx.set_embeddings(emb)
        want = mem_idx.search_batch(q, 10, ef)
        ws = mem_idx.last_stats()
        rec_idx = ia.LeannIndex.from_csr(g, None, dimension=64).upload(0)
        rec_idx.set_recompute_provider(enc, tok, lens, cache_rows=rows)
        got = rec_idx.search_batch(q, 10, ef)
        st = rec_idx.last_stats()
        assert got[2].tolist() == want[2].tolist() and got[0].tolist() == want[0].tolist(), (seed, ef, rows)
        assert got[1].view(np.uint32).tolist() == want[1].view(np.uint32).tolist()
        for f in ("expansions", "edges", "evals", "pushes"):
            assert st[f] == ws[f], (f, seed, ef, rows)
        assert ef > 512 or deg > 128   # no traversal kernel in front: every query is the heap-exact kernel's
        assert st["encoded_nodes"] >= rows          # the slab turned over
