"""Bench/test harness utilities (torch on the GPU): synthetic dataset, a scalable graph
builder that emits the reference's CsrGraph fields, and brute-force ground truth.

This module is plumbing around the hot path, not the hot path: the search itself always
runs in libislands_amd.so.  The builder here is what makes 1M-100M-node graphs available
within minutes (the reference's LeannIndex::build, leann.rs:560-631, is a sequential
O(n * ef_c * deg * d) CPU loop); a native HIP builder that follows the reference's
selection rule is SURVEY.md section 8f rank 1 ("next").

Dataset "H" (hierarchical Gaussian mixture, L2-normalised) -- SURVEY section 8d's dataset G
made hierarchical so that a proximity graph is navigable at all: with 10^4 i.i.d. N(0, I)
centres in d = 768 every centre is (almost) equidistant from every other and best-first
search has no gradient to follow between clusters.  Here the N / per_cluster leaf centres are
the leaves of a tree with branching factor 10 (depth 4 at 10M points); every tree node adds
an independent Gaussian offset whose scale shrinks with depth (1, 0.7, 0.5, 0.35, ...):
    leaf centre    c = sum_j SIGMAS[j] * g_j(ancestor_j)
    points         x = c + 0.25 * N(0, I_d), then x / ||x||
Cosine distance is ~0.03 inside a leaf, ~0.10 / 0.23 / 0.48 / 1.0 to leaves that split off
1 / 2 / 3 / 4 levels higher.
Point i belongs to leaf cluster perm(i) // per_cluster for a fixed pseudo-random
permutation, so neighbouring ids are unrelated.  Every chunk of 65536 rows has its own
seed: any id range (shard) can be generated independently and identically on any rank.
"""
from __future__ import annotations

import math

import torch

CHUNK = 65536
_MULT = 2654435761  # Knuth multiplicative hash, odd -> bijection mod 2^32


def _leaf_of(ids: torch.Tensor, n_total: int, per_cluster: int) -> torch.Tensor:
    """Pseudo-random but fixed cluster assignment of global ids."""
    h = (ids.to(torch.int64) * _MULT + 12345) & 0xFFFFFFFF
    n_leaf = max(1, n_total // per_cluster)
    return h % n_leaf


BRANCH = 10                              # children per internal centre of the mixture tree
SIGMAS = (1.0, 0.7, 0.5, 0.35, 0.25, 0.2)  # spread added at tree depth 0, 1, 2, ...
POINT_SIGMA = 0.25


def _centres(n_total: int, d: int, per_cluster: int, seed: int, device) -> torch.Tensor:
    """Leaf centres of the mixture tree: leaf l = sum over depths j of SIGMAS[j] * g_j(l // B^(L-1-j)),
    g_j ~ N(0, I_d) per tree node, L = number of base-BRANCH digits of the leaf count."""
    n_leaf = max(1, n_total // per_cluster)
    depth = max(1, math.ceil(math.log(n_leaf, BRANCH) - 1e-9)) if n_leaf > 1 else 1
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    leaf_idx = torch.arange(n_leaf, device=device)
    leaf = torch.zeros((n_leaf, d), device=device, dtype=torch.float32)
    for j in range(depth):
        anc = leaf_idx // (BRANCH ** (depth - 1 - j))
        n_nodes = int(anc.max().item()) + 1
        table = torch.randn((n_nodes, d), generator=g, device=device, dtype=torch.float32)
        leaf += SIGMAS[min(j, len(SIGMAS) - 1)] * table[anc]
    return leaf


def make_rows(n_total: int, d: int, start: int, count: int, seed: int = 42,
              per_cluster: int = 1000, device="cuda:0", query: bool = False,
              distinct_leaves: bool = False) -> torch.Tensor:
    """Rows [start, start+count) of dataset H (or of the query set when query=True: same
    mixture, independent noise, seed+1 -- out-of-sample queries).  distinct_leaves (queries only):
    query i of the stream comes from leaf (i * 7919) mod n_leaf instead of a random one, so that any
    n_leaf consecutive queries sit in n_leaf different leaf clusters -- a measurement aid: queries in
    flight together then share no neighbourhood (bench.py --distinct-leaves)."""
    dev = torch.device(device)
    centres = _centres(n_total, d, per_cluster, seed, dev)
    out = torch.empty((count, d), device=dev, dtype=torch.float32)
    c0 = start // CHUNK
    c1 = (start + count + CHUNK - 1) // CHUNK
    g = torch.Generator(device=dev)
    for c in range(c0, c1):
        lo, hi = c * CHUNK, (c + 1) * CHUNK
        g.manual_seed((seed + (1 if query else 0)) * 1000003 + c * 7919 + (5 if query else 0))
        noise = torch.randn((CHUNK, d), generator=g, device=dev, dtype=torch.float32)
        ids = torch.arange(lo, hi, device=dev)
        if query and distinct_leaves:
            leaf = (ids * 7919) % centres.shape[0]
        elif query:  # queries pick a leaf uniformly at random
            leaf = torch.randint(0, centres.shape[0], (CHUNK,), generator=g, device=dev)
        else:
            leaf = _leaf_of(ids, n_total, per_cluster)
        x = centres[leaf] + POINT_SIGMA * noise
        x = x / x.norm(dim=1, keepdim=True)
        a, b = max(lo, start), min(hi, start + count)
        out[a - start:b - start] = x[a - lo:b - lo]
    return out


def make_uniform(count: int, d: int, seed: int, device="cuda:0", start: int = 0) -> torch.Tensor:
    """Reference-style data: i.i.d. uniform [-1, 1) (benches/hnsw_benchmarks.rs:9-14), rows
    [start, start + count) of the stream `seed` names.  Every chunk of 65536 rows has its own seed, so
    a shard generates exactly its rows and nothing else (config 4: 100M rows never exist in one place)."""
    dev = torch.device(device)
    out = torch.empty((count, d), device=dev, dtype=torch.float32)
    g = torch.Generator(device=dev)
    for c in range(start // CHUNK, (start + count + CHUNK - 1) // CHUNK):
        lo, hi = c * CHUNK, (c + 1) * CHUNK
        g.manual_seed(seed * 1000003 + c * 7919 + 11)
        x = torch.rand((CHUNK, d), generator=g, device=dev, dtype=torch.float32) * 2 - 1
        a, b = max(lo, start), min(hi, start + count)
        out[a - start:b - start] = x[a - lo:b - lo]
    return out


def make_manifold(count: int, d: int, seed: int, device="cuda:0", start: int = 0, latent: int = 16,
                  noise: float = 0.1) -> torch.Tensor:
    """Dataset M: rows on a low-dimensional linear manifold, x = normalise(z W + noise * g) with z ~ N(0, I_latent),
    W a fixed [latent, d] matrix with N(0, 1 / latent) entries and g ~ N(0, I_d) -- no clusters, no tree, one smooth
    density of intrinsic dimension `latent` (what learned embeddings look like to a proximity graph far more than
    i.i.d. uniform rows in 768 dimensions, which no ANN index can navigate).  Nothing here knows about any graph
    builder.  Rows [start, start + count) of the stream `seed` names, chunk-seeded like the other datasets."""
    dev = torch.device(device)
    gw = torch.Generator(device=dev)
    gw.manual_seed(977)
    W = torch.randn((latent, d), generator=gw, device=dev, dtype=torch.float32) / math.sqrt(latent)
    out = torch.empty((count, d), device=dev, dtype=torch.float32)
    g = torch.Generator(device=dev)
    for c in range(start // CHUNK, (start + count + CHUNK - 1) // CHUNK):
        lo, hi = c * CHUNK, (c + 1) * CHUNK
        g.manual_seed(seed * 1000003 + c * 7919 + 23)
        z = torch.randn((CHUNK, latent), generator=g, device=dev, dtype=torch.float32)
        x = z @ W + noise * torch.randn((CHUNK, d), generator=g, device=dev, dtype=torch.float32)
        x = x / x.norm(dim=1, keepdim=True)
        a, b = max(lo, start), min(hi, start + count)
        out[a - start:b - start] = x[a - lo:b - lo]
    return out


# ------------------------------------------------------------------ ground truth
@torch.no_grad()
def brute_force_topk(x: torch.Tensor, q: torch.Tensor, k: int, metric: str = "cosine",
                     chunk: int = 1 << 20):
    """Exact top-k (ids int64, distances f32) of every query row against x, fp32."""
    nq = q.shape[0]
    best_d = torch.full((nq, k), float("inf"), device=x.device)
    best_i = torch.zeros((nq, k), dtype=torch.int64, device=x.device)
    qn = q / q.norm(dim=1, keepdim=True).clamp_min(1e-30) if metric == "cosine" else q
    for s in range(0, x.shape[0], chunk):
        xb = x[s:s + chunk]
        if metric == "cosine":
            sim = qn @ (xb / xb.norm(dim=1, keepdim=True).clamp_min(1e-30)).T
            dist = 1.0 - sim
        else:
            dist = torch.cdist(q, xb)
        kk = min(k, xb.shape[0])
        dd, ii = torch.topk(dist, kk, dim=1, largest=False)
        cat_d = torch.cat([best_d, dd], 1)
        cat_i = torch.cat([best_i, ii + s], 1)
        dd2, sel = torch.topk(cat_d, k, dim=1, largest=False)
        best_d, best_i = dd2, torch.gather(cat_i, 1, sel)
    return best_i, best_d


def brute_force_topk_native(x: torch.Tensor, q: torch.Tensor, k: int, metric: int = 0):
    """Exact top-k through the library's own brute force (isl_bruteforce_topk: float32 MFMA
    distance blocks + a running top-k), device buffers in and out."""
    import ctypes as C

    from islands_amd import _check, _ffi
    nq = q.shape[0]
    x, q = x.contiguous(), q.contiguous()
    ids = torch.zeros((nq, k), dtype=torch.int64, device=x.device)
    dd = torch.zeros((nq, k), dtype=torch.float32, device=x.device)
    cnt = torch.zeros(nq, dtype=torch.int32, device=x.device)
    torch.cuda.synchronize(x.device)
    _check(_ffi.lib().isl_bruteforce_topk(
        metric, C.c_void_p(q.data_ptr()), nq, C.c_void_p(x.data_ptr()), x.shape[0], x.shape[1], k,
        C.c_void_p(ids.data_ptr()), C.c_void_p(dd.data_ptr()), C.c_void_p(cnt.data_ptr()), 1,
        x.device.index or 0, None))
    return ids, dd


def brute_force_topk_bf16_native(x16: torch.Tensor, q16: torch.Tensor, k: int, metric: int = 0):
    """Exact top-k under the library's bf16 distance GEMM (isl_bruteforce_topk_bf16), device buffers."""
    import ctypes as C

    from islands_amd import _check, _ffi
    nq = q16.shape[0]
    assert x16.dtype == torch.bfloat16 and q16.dtype == torch.bfloat16
    x16, q16 = x16.contiguous(), q16.contiguous()
    ids = torch.zeros((nq, k), dtype=torch.int64, device=x16.device)
    dd = torch.zeros((nq, k), dtype=torch.float32, device=x16.device)
    cnt = torch.zeros(nq, dtype=torch.int32, device=x16.device)
    torch.cuda.synchronize(x16.device)
    _check(_ffi.lib().isl_bruteforce_topk_bf16(
        metric, C.c_void_p(q16.data_ptr()), nq, C.c_void_p(x16.data_ptr()), x16.shape[0], x16.shape[1], k,
        C.c_void_p(ids.data_ptr()), C.c_void_p(dd.data_ptr()), C.c_void_p(cnt.data_ptr()), 1,
        x16.device.index or 0, None))
    return ids, dd


@torch.no_grad()
def build_knn_graph(x: torch.Tensor, k: int = 30, m0: int = 60, block: int = 16384, bf16_above: int = 2_000_000,
                    progress=None):
    """A graph the harness did not tailor: every node's k exact nearest neighbours (cosine) by the
    library's own brute force -- float32 MFMA blocks up to `bf16_above` rows, the bf16 distance GEMM
    beyond (1.5e17 flops at 10M x 768) --, plus the reverse edges, rows truncated to m0 (own neighbours
    first, nearest first; then reverse edges, nearest first).  Entry point = the medoid (the row
    nearest to the mean of all rows), as NSG / Vamana take it.  No hierarchy, no diversification.
    Returns (offsets int64 [n+1], neighbours int32 [nnz], entry)."""
    dev = x.device
    n, d = x.shape
    use16 = n > bf16_above and d % 64 == 0
    x16 = x.to(torch.bfloat16) if use16 else None
    nb = torch.empty((n, k), dtype=torch.int64, device=dev)
    nd = torch.empty((n, k), dtype=torch.float32, device=dev)
    self_ids = torch.arange(n, device=dev)
    for s0 in range(0, n, block):
        s1 = min(n, s0 + block)
        if use16:
            ii, dd = brute_force_topk_bf16_native(x16, x16[s0:s1], k + 1)
        else:
            ii, dd = brute_force_topk_native(x, x[s0:s1], k + 1)
        # drop the node itself (first unless an exact duplicate row ties with it)
        own = ii == self_ids[s0:s1, None]
        has = own.any(1)
        drop = torch.where(has, own.to(torch.int64).argmax(1), torch.full_like(has, k, dtype=torch.int64))
        keep = torch.arange(k + 1, device=dev)[None, :] != drop[:, None]
        nb[s0:s1] = ii[keep].view(-1, k)
        nd[s0:s1] = dd[keep].view(-1, k)
        if progress and (s0 // block) % 16 == 0:
            progress(f"kNN lists of rows [{s0}, {s1}) of {n}")
    del x16
    src = self_ids[:, None].expand(n, k).reshape(-1)
    dst = nb.reshape(-1)
    dist = nd.reshape(-1)
    # forward edges rank before reverse edges, nearest first inside each class
    f_key = dist.double()
    r_key = dist.double() + 16.0
    a_src, a_dst, a_key = torch.cat([src, dst]), torch.cat([dst, src]), torch.cat([f_key, r_key])
    o = torch.argsort(a_key, stable=True)
    a_src, a_dst, a_key = a_src[o], a_dst[o], a_key[o]
    o = torch.argsort(a_src * (1 << 32) + a_dst, stable=True)  # (src, dst) groups, best key first inside
    a_src, a_dst, a_key = a_src[o], a_dst[o], a_key[o]
    first = torch.ones_like(a_src, dtype=torch.bool)
    first[1:] = (a_src[1:] != a_src[:-1]) | (a_dst[1:] != a_dst[:-1])
    a_src, a_dst, a_key = a_src[first], a_dst[first], a_key[first]
    o = torch.argsort(a_key, stable=True)
    a_src, a_dst = a_src[o], a_dst[o]
    o = torch.argsort(a_src, stable=True)  # rows, best key first
    a_src, a_dst = a_src[o], a_dst[o]
    start = torch.ones_like(a_src, dtype=torch.bool)
    start[1:] = a_src[1:] != a_src[:-1]
    seg_start = torch.nonzero(start).squeeze(1)
    seg_id = torch.cumsum(start.to(torch.int64), 0) - 1
    within = torch.arange(a_src.numel(), device=dev) - seg_start[seg_id]
    ok = within < m0
    a_src, a_dst = a_src[ok], a_dst[ok]
    deg = torch.bincount(a_src, minlength=n)
    offsets = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    offsets[1:] = torch.cumsum(deg, 0)
    mean = torch.zeros(d, dtype=torch.float64, device=dev)
    for s0 in range(0, n, 1 << 18):
        mean += x[s0:s0 + (1 << 18)].double().sum(0)
    mean = (mean / n).float()
    best, entry = -3.0, 0
    mn = mean / mean.norm().clamp_min(1e-30)
    for s0 in range(0, n, 1 << 18):
        xb = x[s0:s0 + (1 << 18)]
        sim = (xb @ mn) / xb.norm(dim=1).clamp_min(1e-30)
        v, i = sim.max(0)
        if float(v) > best:
            best, entry = float(v), s0 + int(i)
    return offsets, a_dst.to(torch.int32), entry


def recall_at_k(found_ids: torch.Tensor, found_cnt: torch.Tensor, truth_ids: torch.Tensor) -> float:
    k = truth_ids.shape[1]
    f = found_ids[:, :k].to(torch.int64)
    valid = torch.arange(k, device=f.device)[None, :] < found_cnt[:, None].to(torch.int64)
    hit = ((f[:, :, None] == truth_ids[:, None, :]) & valid[:, :, None]).any(2)
    return float(hit.sum().item()) / float(truth_ids.numel())


# ------------------------------------------------------------------ graph builder
def _trace(msg):
    """SYNTH_TRACE=1: phase markers with a device synchronisation in front (which phase a GPU fault
    belongs to)."""
    import os
    import sys
    if os.environ.get("SYNTH_TRACE"):
        torch.cuda.synchronize()
        print(f"[synth] {msg}", file=sys.stderr, flush=True)


# Every tensor the harness gathers or materialises per chunk is bounded in BYTES, computed from the
# element size of what is actually allocated (float32 operands: 4 bytes whatever the rows' storage
# type).  Round 2 lost a box to "Memory access fault by GPU" inside this builder on 1M x 4096 bf16 rows
# (gpurun_out/r02_config5_1m.err) with per-chunk tensors of 2^30 bf16 elements = 2 GiB less 512 KiB
# ([2730, 96, 4096] in _diversify: the row gather and the two batched bf16 GEMMs over it) after a
# first bound, on the element count, had already excluded 2^31 elements; the run passed once the
# operands were float32 AND the count was halved.  Which of the two mattered was never isolated (and
# must not be, by provoking the fault again): the bound below keeps every such tensor at or under
# 1 GiB -- a factor of two away from 2^31 in bytes and at least four in elements -- so that no 32-bit
# byte or element offset inside torch / rocBLAS / hipBLASLt kernels can come near its limit.
_CHUNK_BYTES = 1 << 30


def _rows_within(bytes_per_row: int, want: int, floor: int = 64) -> int:
    """Rows per chunk such that rows * bytes_per_row <= _CHUNK_BYTES (at least `floor`, at most `want`)."""
    return max(floor, min(want, _CHUNK_BYTES // max(1, bytes_per_row)))


# dtype of the k-means / bucket ASSIGNMENT GEMMs (build_graph(precise=True) switches to float32:
# embeddings with a large common component -- a randomly initialised encoder's -- differ from each
# other only past bfloat16's 8 bits)
_ASSIGN_DTYPE = torch.bfloat16


@torch.no_grad()
def _knn_in_buckets(x, member_ids, bucket_off, k, mem_budget=1.0e9):
    """For every (point, bucket) membership: the k nearest OTHER members of that bucket.
    member_ids: int64 [P] point ids grouped by bucket; bucket_off: int64 [B+1].
    Returns (nbr_ids int64 [P, k] (-1 = none), nbr_sim f32 [P, k])."""
    dev = x.device
    P = member_ids.numel()
    nbr = torch.full((P, k), -1, dtype=torch.int64, device=dev)
    sim_out = torch.full((P, k), -2.0, dtype=torch.float32, device=dev)
    sizes = (bucket_off[1:] - bucket_off[:-1])
    order = torch.argsort(sizes)
    sizes_s = sizes[order].tolist()
    order_l = order.tolist()
    off_l = bucket_off.tolist()
    d = x.shape[1]
    i = 0
    B = len(order_l)
    while i < B:
        if sizes_s[i] <= 1:
            i += 1
            continue
        # group buckets of similar size: padded batch within the memory budget
        smax = sizes_s[i]
        j = i
        while j < B and sizes_s[j] <= max(64, int(smax * 1.25)):
            smax2 = sizes_s[j]
            g = j - i + 1
            if g * smax2 * (smax2 + d) * 4 > mem_budget and g > 1:
                break
            j += 1
        j = max(j, i + 1)
        group = order_l[i:j]
        S = sizes_s[j - 1]
        G = len(group)
        idx = torch.zeros((G, S), dtype=torch.int64, device=dev)
        msk = torch.zeros((G, S), dtype=torch.bool, device=dev)
        pos = torch.zeros((G, S), dtype=torch.int64, device=dev)
        for gi, b in enumerate(group):
            s0, s1 = off_l[b], off_l[b + 1]
            idx[gi, : s1 - s0] = member_ids[s0:s1]
            msk[gi, : s1 - s0] = True
            pos[gi, : s1 - s0] = torch.arange(s0, s1, device=dev)
        kk = min(k, S - 1)
        rows_per = max(1, int(mem_budget // (G * S * 4)))
        X = x[idx].float()  # [G, S, d] (float32 whatever the rows' storage type)
        for r0 in range(0, S, rows_per):
            r1 = min(S, r0 + rows_per)
            sim = torch.bmm(X[:, r0:r1], X.transpose(1, 2))  # [G, r, S]
            sim.masked_fill_(~msk[:, None, :], -3.0)
            ar = torch.arange(r0, r1, device=dev)
            sim[:, ar - r0, ar] = -3.0  # self
            sv, si = torch.topk(sim, kk, dim=2)
            gid = torch.gather(idx[:, None, :].expand(G, r1 - r0, S), 2, si)
            gid = torch.where(sv > -2.5, gid, torch.full_like(gid, -1))
            rowmask = msk[:, r0:r1]
            p = pos[:, r0:r1][rowmask]
            nbr[p, :kk] = gid[rowmask]
            sim_out[p, :kk] = sv[rowmask]
        i = j
    return nbr, sim_out


@torch.no_grad()
def _knn_exact(x, ids, k, chunk=4096):
    """Exact cosine kNN among the points `ids` (rows L2-normalised), row-chunked GEMM + top-k.
    Returns int64 [len(ids), k] GLOBAL ids sorted by decreasing similarity."""
    n = ids.numel()
    X = x[ids].float()
    out = torch.empty((n, k), dtype=torch.int64, device=x.device)
    for s0 in range(0, n, chunk):
        sim = X[s0:s0 + chunk] @ X.T
        r = torch.arange(s0, min(n, s0 + chunk), device=x.device)
        sim[r - s0, r] = -3.0
        out[s0:s0 + chunk] = ids[torch.topk(sim, k, dim=1).indices]
    return out


@torch.no_grad()
def _medoids(x, ids, C, chunk=1 << 18):
    """For every centroid the most similar point among `ids` (global ids, duplicates removed)."""
    dev = x.device
    chunk = _rows_within(x.shape[1] * 4, chunk, 1024)  # x[sub] widened to at most float32
    k = C.shape[0]
    best_s = torch.full((k,), -3.0, device=dev)
    best_i = torch.zeros(k, dtype=torch.int64, device=dev)
    Ch = C.to(_ASSIGN_DTYPE)
    for s0 in range(0, ids.numel(), chunk):
        sub = ids[s0:s0 + chunk]
        sim = (x[sub].to(_ASSIGN_DTYPE) @ Ch.T).float()
        sv, si = sim.max(0)
        upd = sv > best_s
        best_s = torch.where(upd, sv, best_s)
        best_i = torch.where(upd, sub[si], best_i)
    return torch.unique(best_i)


@torch.no_grad()
def _lloyd_centroids(x, n_cent, iters=3, seed=11, chunk=1 << 18, ids=None):
    """n_cent unit-norm centroids of the rows `ids` (default: all): random rows refined by a few
    spherical k-means steps (bf16 assignment GEMM, f32 update)."""
    dev = x.device
    if ids is not None:
        x = x[ids]
    n, d = x.shape
    chunk = _rows_within(d * 4, chunk, 1024)  # xb[order].float(): d * 4 bytes per row
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    C = x[torch.randperm(n, generator=g, device=dev)[:n_cent]].float().clone()  # (x may be bf16 rows)
    for _ in range(iters):
        Ch = C.to(_ASSIGN_DTYPE)
        sums = torch.zeros_like(C)
        cnt = torch.zeros(n_cent, device=dev)
        for s0 in range(0, n, chunk):
            xb = x[s0:s0 + chunk]
            a = torch.argmax(xb.to(_ASSIGN_DTYPE) @ Ch.T, dim=1)
            # deterministic segmented sums (index_add_ uses float atomics: run-to-run noise
            # would change the buckets and with them the graph)
            order = torch.argsort(a, stable=True)
            lens = torch.bincount(a, minlength=n_cent)
            part = torch.segment_reduce(xb[order].float(), "sum", lengths=lens, unsafe=True)
            sums += part
            cnt += lens.to(torch.float32)
        alive = cnt > 0
        newC = sums / sums.norm(dim=1, keepdim=True).clamp_min(1e-20)
        C = torch.where(alive[:, None], newC, C)
    return C


@torch.no_grad()
def _knn_subset(x, ids, k, centroids=None, assign_chunk=1 << 18, exact_limit=400_000):
    """k nearest neighbours (cosine; rows are assumed L2-normalised) among the points `ids`.
    Up to exact_limit points: exact.  Larger sets: every point joins the buckets of its 2
    nearest centroids (`centroids`: [C, d] unit vectors), kNN inside buckets, lists merged.
    Returns int64 [len(ids), k] of GLOBAL ids (-1 = none), nearest first."""
    dev = x.device
    n = ids.numel()
    if n <= 1:
        return torch.full((n, k), -1, dtype=torch.int64, device=dev)
    if n <= exact_limit or centroids is None:
        return _knn_exact(x, ids, min(k, n - 1))
    C = centroids
    assign_chunk = _rows_within(x.shape[1] * 4, assign_chunk, 1024)
    a1 = torch.empty(n, dtype=torch.int64, device=dev)
    a2 = torch.empty(n, dtype=torch.int64, device=dev)
    Ch = C.to(_ASSIGN_DTYPE)
    for s in range(0, n, assign_chunk):
        sim = x[ids[s:s + assign_chunk]].to(_ASSIGN_DTYPE) @ Ch.T
        top = torch.topk(sim.float(), 2, dim=1).indices
        a1[s:s + assign_chunk], a2[s:s + assign_chunk] = top[:, 0], top[:, 1]
    nC = C.shape[0]
    bucket = torch.cat([a1, a2])
    member_local = torch.cat([torch.arange(n, device=dev), torch.arange(n, device=dev)])
    order = torch.argsort(bucket, stable=True)
    bucket_s, member_local_s = bucket[order], member_local[order]
    counts = torch.bincount(bucket_s, minlength=nC)
    boff = torch.zeros(nC + 1, dtype=torch.int64, device=dev)
    boff[1:] = torch.cumsum(counts, 0)
    nb, sm = _knn_in_buckets(x, ids[member_local_s], boff, k)
    # bring both memberships of every point side by side and keep its k best distinct ones
    inv = torch.empty_like(order)
    inv[order] = torch.arange(order.numel(), device=dev)
    p1, p2 = inv[:n], inv[n:]
    cat_i = torch.cat([nb[p1], nb[p2]], 1)
    cat_s = torch.cat([sm[p1], sm[p2]], 1)
    srt = torch.argsort(cat_i, dim=1)  # duplicates become adjacent
    ci, cs = torch.gather(cat_i, 1, srt), torch.gather(cat_s, 1, srt)
    dup = torch.zeros_like(ci, dtype=torch.bool)
    dup[:, 1:] = ci[:, 1:] == ci[:, :-1]
    cs = torch.where(dup | (ci < 0), torch.full_like(cs, -3.0), cs)
    sv, sel = torch.topk(cs, k, dim=1)
    out = torch.gather(ci, 1, sel)
    return torch.where(sv > -2.5, out, torch.full_like(out, -1))


@torch.no_grad()
def _diversify(x, ids, cand, m, chunk=8192):
    """HNSW-style neighbour selection on a candidate pool (cand: [n, K] global ids sorted by
    decreasing similarity to the base point, -1 = none): a candidate is kept only if it is
    closer to the base than to every neighbour kept so far; free slots are then refilled with
    the nearest rejected candidates.  Gives the long edges between clusters that a plain kNN
    list lacks.  Returns int64 [n, m] (-1 = none)."""
    dev = x.device
    n, K = cand.shape
    out = torch.full((n, m), -1, dtype=torch.int64, device=dev)
    # the gathered candidate rows Cv [chunk, K, d] as float32 stay within _CHUNK_BYTES (see there: the
    # recorded GPU fault sat in this function's neighbourhood -- chunk = 2730 at K = 96, d = 4096 made
    # Cv 2,146,959,360 bytes of bf16, 512 KiB short of 2^31, gathered with x[c] and fed to two bmm's)
    chunk = _rows_within(K * x.shape[1] * 4, chunk, 64)
    for s0 in range(0, n, chunk):
        c = cand[s0:s0 + chunk]
        valid = c >= 0
        B = x[ids[s0:s0 + chunk]].float()
        Cv = x[c.clamp_min(0)].float()
        sb = torch.bmm(Cv, B[:, :, None]).squeeze(2)          # similarity candidate <-> base
        pair = torch.bmm(Cv, Cv.transpose(1, 2))               # candidate <-> candidate
        sel = torch.zeros_like(valid)
        cnt = torch.zeros(c.shape[0], dtype=torch.int64, device=dev)
        for j in range(K):
            blocked = ((pair[:, j, :] >= sb[:, j:j + 1]) & sel).any(1)
            ok = valid[:, j] & (cnt < m) & ~blocked
            sel[:, j] = ok
            cnt += ok.to(torch.int64)
        rest = valid & ~sel
        fill_rank = torch.cumsum(rest.to(torch.int64), 1)
        filler = rest & (fill_rank <= (m - cnt)[:, None])
        keep = sel | filler
        # selected first (in distance order), then fillers (in distance order)
        pos = torch.arange(K, device=dev)[None, :].expand_as(c)
        key = torch.where(sel, pos, torch.where(filler, pos + K, torch.full_like(pos, 4 * K)))
        order = torch.argsort(key, dim=1)[:, :m]
        picked = torch.gather(c, 1, order)
        okk = torch.gather(keep, 1, order)
        out[s0:s0 + chunk, : picked.shape[1]] = torch.where(okk, picked, torch.full_like(picked, -1))
    return out


@torch.no_grad()
def _nearest_parent(x, child_ids, parent_ids, npar=2, chunk=1 << 16):
    """Indices (into parent_ids) of the npar most similar parents of every child: [n, npar]."""
    P = x[parent_ids].float()
    chunk = _rows_within(x.shape[1] * 4, chunk, 1024)
    npar = min(npar, parent_ids.numel())
    out = torch.empty((child_ids.numel(), npar), dtype=torch.int64, device=x.device)
    for s0 in range(0, child_ids.numel(), chunk):
        out[s0:s0 + chunk] = torch.topk(x[child_ids[s0:s0 + chunk]].float() @ P.T, npar, dim=1).indices
    return out


@torch.no_grad()
def build_graph(x: torch.Tensor, m0: int = 60, k0: int = 28, k_upper: int = 20, pool: int = 96,
                child_cap: int = 30, seed: int = 7, level_ratio: int = 32, precise: bool = False):
    global _ASSIGN_DTYPE
    saved = _ASSIGN_DTYPE
    _ASSIGN_DTYPE = torch.float32 if precise else torch.bfloat16
    try:
        return _build_graph(x, m0, k0, k_upper, pool, child_cap, seed, level_ratio)
    finally:
        _ASSIGN_DTYPE = saved


@torch.no_grad()
def _build_graph(x: torch.Tensor, m0: int = 60, k0: int = 28, k_upper: int = 20, pool: int = 96,
                 child_cap: int = 30, seed: int = 7, level_ratio: int = 32):
    """Flattened hierarchical proximity graph over the rows of x (L2-normalised, cosine).
    Returns (offsets int64 [n+1], neighbours int32 [nnz], entry_point int).

    Level l >= 1 is a random 1/level_ratio^l subset of the nodes (HNSW-like levels, but every
    edge lives in the single layer that the LEANN search walks).  Edges, by priority:
      1. parent -> child: every level-l node (l >= 1) is listed by its nearest level-(l+1)
         node (up to child_cap per parent) -- a navigating tree from the entry point down;
      2. k_upper diversified neighbours inside the node's top level (the top level is a clique);
      3. the node's k0 nearest neighbours among all nodes;
      4. diversified neighbours inside its lower levels;
      5. reverse edges of all of the above while room is left.
    Rows hold distinct ids, no self loops, at most m0 entries."""
    dev = x.device
    n = x.shape[0]
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    # Levels: level 1 is a random 1/level_ratio sample; from level 2 up the nodes are the
    # medoids of a spherical k-means over the level below, so that every region of the data
    # owns a node at every scale (a purely random sample leaves ~1/e of the natural
    # clusters without a representative two levels up, and those become unreachable).
    # the float32 copy x[ids] of an exact kNN stays within 2^30 bytes (d = 768: 349525 rows, 4096: 65536)
    exact_limit = min(400_000, _CHUNK_BYTES // (x.shape[1] * 4))
    all_ids = torch.arange(n, device=dev)
    perm = torch.randperm(n, generator=g, device=dev)
    cent = None
    levels = [all_ids]
    if n > k_upper:
        n1 = max(1, n // level_ratio)
        n2 = max(1, n1 // level_ratio)
        uppers = []
        if n2 > k_upper // 2:
            _trace(f"lloyd {n2} centroids")
            cent = _lloyd_centroids(x, n2)
            _trace("medoids")
            cur = _medoids(x, all_ids, cent)
            uppers.append(cur)
            while cur.numel() > k_upper:
                kk = max(1, cur.numel() // level_ratio)
                c = _lloyd_centroids(x, kk, iters=5, ids=cur)
                cur = _medoids(x, cur, c)
                uppers.append(cur)
        l1 = perm[:n1]
        if uppers:
            l1 = torch.unique(torch.cat([l1, uppers[0]]))
        levels.append(torch.sort(l1).values)
        levels.extend(uppers)
    if n <= exact_limit:
        cent = None
    top_level_of = torch.zeros(n, dtype=torch.int64, device=dev)
    for li, lv in enumerate(levels):
        top_level_of[lv] = li
    E_src, E_dst, E_prio = [], [], []

    def add_edges(src, dst, prio):
        ok = (dst >= 0) & (dst != src)
        E_src.append(src[ok])
        E_dst.append(dst[ok])
        E_prio.append(prio[ok].to(torch.int64))

    for li, ids in enumerate(levels):
        if ids.numel() < 2:
            continue
        _trace(f"level {li}: {ids.numel()} nodes")
        use_cent = cent
        if li == 0:
            nb = _knn_subset(x, ids, min(k0, ids.numel() - 1), use_cent, exact_limit=exact_limit)
        elif ids.numel() <= k_upper + 1:  # top level: clique
            nb = ids[None, :].expand(ids.numel(), ids.numel()).clone()
            nb[nb == ids[:, None]] = -1
        else:
            cand = _knn_subset(x, ids, min(pool, ids.numel() - 1), use_cent, exact_limit=exact_limit)
            nb = _diversify(x, ids, cand, min(k_upper, cand.shape[1]))
        src = ids[:, None].expand_as(nb).reshape(-1)
        rank = torch.arange(nb.shape[1], device=dev)[None, :].expand_as(nb).reshape(-1)
        if li == 0:  # upper-level nodes keep their row for navigation: their kNN list goes last
            upper = (top_level_of[ids] >= 2)[:, None].expand_as(nb).reshape(-1)
            base = torch.where(upper, torch.full_like(rank, 900), torch.full_like(rank, 200))
        else:  # the node's own top level ranks before its kNN list, lower levels after it
            is_top = (top_level_of[ids] == li)[:, None].expand_as(nb).reshape(-1)
            base = torch.where(is_top, torch.full_like(rank, 100), torch.full_like(rank, 300 + 20 * li))
        add_edges(src, nb.reshape(-1), base + rank)
        if li >= 1 and li + 1 < len(levels):  # parent -> child edges
            # children: nodes whose top level is li; parents: nodes whose top level is li + 1
            # (all nodes of the last level).  A node thus lists children of one level only.
            last = li + 1 == len(levels) - 1
            pl = levels[li + 1]
            parents = pl if last else pl[top_level_of[pl] == li + 1]
            ids_c = ids[top_level_of[ids] == li]
            if parents.numel() == 0 or ids_c.numel() == 0:
                continue
            ids_saved, ids = ids, ids_c
            par2 = _nearest_parent(x, ids, parents)
            par = par2.reshape(-1)
            chd = ids[:, None].expand_as(par2).reshape(-1)
            sim = (x[chd].float() * x[parents[par]].float()).sum(1)
            o = torch.argsort(par.double() * 4.0 - sim.double())  # by parent, most similar child first
            par_s, child_s = par[o], chd[o]
            first = torch.ones_like(par_s, dtype=torch.bool)
            first[1:] = par_s[1:] != par_s[:-1]
            seg_start = torch.nonzero(first).squeeze(1)
            seg_id = torch.cumsum(first.to(torch.int64), 0) - 1
            within = torch.arange(par_s.numel(), device=dev) - seg_start[seg_id]
            # candidate children per parent (nearest first), then the same diversified choice
            # as for neighbours: one child per direction, so that a parent lists children of
            # every cluster it is responsible for, not 30 siblings of its own cluster
            cpool = 6 * child_cap
            okc = within < cpool
            cand = torch.full((parents.numel(), cpool), -1, dtype=torch.int64, device=dev)
            cand[par_s[okc], within[okc]] = child_s[okc]
            _trace(f"level {li}: children of {parents.numel()} parents")
            ch = _diversify(x, parents, cand, child_cap)
            psrc = parents[:, None].expand_as(ch).reshape(-1)
            prank = torch.arange(ch.shape[1], device=dev)[None, :].expand_as(ch).reshape(-1)
            add_edges(psrc, ch.reshape(-1), prank)
            ids = ids_saved
    _trace("edges collected")
    src, dst, prio = torch.cat(E_src), torch.cat(E_dst), torch.cat(E_prio)
    # reverse edges rank behind every forward edge
    src, dst, prio = torch.cat([src, dst]), torch.cat([dst, src]), torch.cat([prio, prio + 1000])
    # dedupe (src, dst) keeping the best priority
    o = torch.argsort(prio, stable=True)
    src, dst, prio = src[o], dst[o], prio[o]
    o = torch.argsort(src * (1 << 32) + dst, stable=True)
    src, dst, prio = src[o], dst[o], prio[o]
    keep = torch.ones_like(src, dtype=torch.bool)
    keep[1:] = (src[1:] != src[:-1]) | (dst[1:] != dst[:-1])
    src, dst, prio = src[keep], dst[keep], prio[keep]
    # per source: best m0 by priority
    o = torch.argsort(src * 4096 + prio, stable=True)
    src, dst = src[o], dst[o]
    first = torch.ones_like(src, dtype=torch.bool)
    first[1:] = src[1:] != src[:-1]
    seg_start = torch.nonzero(first).squeeze(1)
    seg_id = torch.cumsum(first.to(torch.int64), 0) - 1
    within = torch.arange(src.numel(), device=dev) - seg_start[seg_id]
    ok = within < m0
    src, dst = src[ok], dst[ok]
    deg = torch.bincount(src, minlength=n)
    offsets = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    offsets[1:] = torch.cumsum(deg, 0)
    neighbours = dst.to(torch.int32)
    entry = int(levels[-1][0].item())
    return offsets, neighbours, entry


def train_pq(x: torch.Tensor, m: int, K: int = 256, iters: int = 6, seed: int = 13,
             sample: int = 262144, chunk: int = 1 << 18):
    """Harness only (PQ training, pq.rs:362-463, is build-time and draws from thread_rng): Lloyd
    iterations per subquantizer on a sample of the rows, then the code of every row = its nearest
    centroid under squared L2.  Returns (codebooks [m][K][dsub] f32, codes [n][m] u16) on x's
    device.  The codes are INPUT data of the two-level search; nothing is compared with them."""
    n, d = x.shape
    dsub = d // m
    g = torch.Generator(device=x.device)
    g.manual_seed(seed)
    sample = _rows_within(d * 4, sample, 4096)
    pick = torch.randperm(n, generator=g, device=x.device)[:min(sample, n)]
    xs = x[pick].float()  # (x may be bf16 rows)
    cb = torch.empty((m, K, dsub), dtype=torch.float32, device=x.device)
    codes = torch.empty((n, m), dtype=torch.int16, device=x.device)
    for j in range(m):
        sub = xs[:, j * dsub:(j + 1) * dsub].contiguous()
        cent = sub[torch.randperm(sub.shape[0], generator=g, device=x.device)[:K]].clone()
        if cent.shape[0] < K:
            cent = torch.cat([cent, cent[:1].expand(K - cent.shape[0], -1)])
        for _ in range(iters):
            a = torch.cdist(sub, cent).argmin(1)
            sums = torch.zeros_like(cent).index_add_(0, a, sub)
            cnt = torch.zeros(K, device=x.device).index_add_(0, a, torch.ones_like(a, dtype=torch.float32))
            cent = torch.where(cnt[:, None] > 0, sums / cnt[:, None].clamp_min(1.0), cent)
        cb[j] = cent
        for s in range(0, n, chunk):
            blk = x[s:s + chunk, j * dsub:(j + 1) * dsub].float()
            codes[s:s + chunk, j] = torch.cdist(blk, cent).argmin(1).to(torch.int16)
    return cb, codes


def graph_stats(offsets: torch.Tensor) -> dict:
    deg = (offsets[1:] - offsets[:-1]).float()
    return {"nodes": int(deg.numel()), "edges": int(offsets[-1].item()),
            "deg_mean": float(deg.mean().item()), "deg_max": int(deg.max().item()),
            "deg_min": int(deg.min().item())}


# ------------------------------------------------------------------ encoder weights
def bert_weight_names(layers: int):
    """Tensor names of a BERT checkpoint as candle's VarBuilder reads them
    (src/core/embedding/candle_provider.rs:267-284)."""
    names = ["embeddings.word_embeddings.weight", "embeddings.position_embeddings.weight",
             "embeddings.token_type_embeddings.weight", "embeddings.LayerNorm.weight",
             "embeddings.LayerNorm.bias"]
    for i in range(layers):
        p = f"encoder.layer.{i}."
        for lin in ("attention.self.query", "attention.self.key", "attention.self.value",
                    "attention.output.dense"):
            names += [p + lin + ".weight", p + lin + ".bias"]
        names += [p + "attention.output.LayerNorm.weight", p + "attention.output.LayerNorm.bias",
                  p + "intermediate.dense.weight", p + "intermediate.dense.bias",
                  p + "output.dense.weight", p + "output.dense.bias",
                  p + "output.LayerNorm.weight", p + "output.LayerNorm.bias"]
    return names


def bert_random_weights(cfg: dict, seed: int = 45, std: float = 0.02):
    """Synthetic encoder weights ~ N(0, std^2), LayerNorm scale 1 + noise (SURVEY.md section 8d,
    config 3: no checkpoint can be fetched offline)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    h, inter = cfg["hidden"], cfg["intermediate"]
    w = {}
    for name in bert_weight_names(cfg["layers"]):
        if name == "embeddings.word_embeddings.weight":
            shp = (cfg["vocab_size"], h)
        elif name == "embeddings.position_embeddings.weight":
            shp = (cfg["max_position"], h)
        elif name == "embeddings.token_type_embeddings.weight":
            shp = (cfg["type_vocab"], h)
        elif "LayerNorm" in name:
            shp = (h,)
        elif name.endswith("intermediate.dense.weight"):
            shp = (inter, h)
        elif name.endswith("intermediate.dense.bias"):
            shp = (inter,)
        elif name.endswith("output.dense.weight") and "attention" not in name:
            shp = (h, inter)
        elif name.endswith(".weight"):
            shp = (h, h)
        else:
            shp = (h,)
        v = rng.standard_normal(shp).astype(np.float32) * np.float32(std)
        if name.endswith("LayerNorm.weight"):
            v = (np.float32(1.0) + v).astype(np.float32)
        w[name] = v
    return w
