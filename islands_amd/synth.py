"""Bench/test harness utilities (torch on the GPU): synthetic dataset, a scalable graph
builder that emits the reference's CsrGraph fields, and brute-force ground truth.

This module is plumbing around the hot path, not the hot path: the search itself always
runs in libislands_amd.so.  The builder here is what makes 1M-100M-node graphs available
within minutes (the reference's LeannIndex::build, leann.rs:560-631, is a sequential
O(n * ef_c * deg * d) CPU loop); a native HIP builder that follows the reference's
selection rule is SURVEY.md section 8f rank 1 ("next").

Dataset "H" (hierarchical Gaussian mixture, L2-normalised) -- SURVEY section 8d's dataset G
with one more level so that a proximity graph is navigable at all: with i.i.d. N(0, I)
centres in d = 768 every centre is (almost) equidistant from every other, so best-first
search has no gradient to follow between 10^4 clusters.
    super-centres  s ~ N(0, I_d)                    one per 100 leaf clusters
    leaf centres   c = s + 0.5 * N(0, I_d)          one per `per_cluster` points
    points         x = c + 0.25 * N(0, I_d), then x / ||x||
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


def _centres(n_total: int, d: int, per_cluster: int, seed: int, device) -> torch.Tensor:
    n_leaf = max(1, n_total // per_cluster)
    n_super = max(1, n_leaf // 100)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    sup = torch.randn((n_super, d), generator=g, device=device, dtype=torch.float32)
    leaf_sup = torch.arange(n_leaf, device=device) % n_super
    leaf = sup[leaf_sup] + 0.5 * torch.randn((n_leaf, d), generator=g, device=device,
                                             dtype=torch.float32)
    return leaf


def make_rows(n_total: int, d: int, start: int, count: int, seed: int = 42,
              per_cluster: int = 1000, device="cuda:0", query: bool = False) -> torch.Tensor:
    """Rows [start, start+count) of dataset H (or of the query set when query=True: same
    mixture, independent noise, seed+1 -- out-of-sample queries)."""
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
        if query:  # queries pick a leaf uniformly at random
            leaf = torch.randint(0, centres.shape[0], (CHUNK,), generator=g, device=dev)
        else:
            leaf = _leaf_of(ids, n_total, per_cluster)
        x = centres[leaf] + 0.25 * noise
        x = x / x.norm(dim=1, keepdim=True)
        a, b = max(lo, start), min(hi, start + count)
        out[a - start:b - start] = x[a - lo:b - lo]
    return out


def make_uniform(count: int, d: int, seed: int, device="cuda:0") -> torch.Tensor:
    """Reference-style data: i.i.d. uniform [-1, 1) (benches/hnsw_benchmarks.rs:9-14)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return torch.rand((count, d), generator=g, device=device, dtype=torch.float32) * 2 - 1


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


def recall_at_k(found_ids: torch.Tensor, found_cnt: torch.Tensor, truth_ids: torch.Tensor) -> float:
    k = truth_ids.shape[1]
    f = found_ids[:, :k].to(torch.int64)
    valid = torch.arange(k, device=f.device)[None, :] < found_cnt[:, None].to(torch.int64)
    hit = ((f[:, :, None] == truth_ids[:, None, :]) & valid[:, :, None]).any(2)
    return float(hit.sum().item()) / float(truth_ids.numel())


# ------------------------------------------------------------------ graph builder
@torch.no_grad()
def _knn_in_buckets(x, member_ids, bucket_off, k, mem_budget=1.5e9):
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
        X = x[idx]  # [G, S, d]
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
def _knn_subset(x, ids, k, centroids_ids=None, assign_chunk=1 << 18):
    """k nearest neighbours (cosine; rows are assumed L2-normalised) among the points `ids`.
    Small sets: brute force.  Large sets: every point joins the buckets of its 2 nearest
    centroids, kNN inside buckets, lists merged.  Returns int64 [len(ids), k] of GLOBAL ids
    (-1 = none)."""
    dev = x.device
    n = ids.numel()
    if n <= 1:
        return torch.full((n, k), -1, dtype=torch.int64, device=dev)
    if n <= 32768 or centroids_ids is None:
        off = torch.tensor([0, n], dtype=torch.int64, device=dev)
        nb, _ = _knn_in_buckets(x, ids, off, k)
        return nb
    C = x[centroids_ids]
    a1 = torch.empty(n, dtype=torch.int64, device=dev)
    a2 = torch.empty(n, dtype=torch.int64, device=dev)
    Ch = C.to(torch.bfloat16)
    for s in range(0, n, assign_chunk):
        sim = x[ids[s:s + assign_chunk]].to(torch.bfloat16) @ Ch.T
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
    for s0 in range(0, n, chunk):
        c = cand[s0:s0 + chunk]
        valid = c >= 0
        B = x[ids[s0:s0 + chunk]]
        Cv = x[c.clamp_min(0)]
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
def _nearest_parent(x, child_ids, parent_ids, chunk=1 << 16):
    """Index (into parent_ids) of the most similar parent of every child."""
    P = x[parent_ids]
    out = torch.empty(child_ids.numel(), dtype=torch.int64, device=x.device)
    for s0 in range(0, child_ids.numel(), chunk):
        out[s0:s0 + chunk] = torch.argmax(x[child_ids[s0:s0 + chunk]] @ P.T, dim=1)
    return out


@torch.no_grad()
def build_graph(x: torch.Tensor, m0: int = 60, k0: int = 28, k_upper: int = 16, pool: int = 64,
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
    perm = torch.randperm(n, generator=g, device=dev)
    levels = [torch.sort(perm).values]
    cur = perm
    while cur.numel() > k_upper:
        cur = cur[: max(1, cur.numel() // level_ratio)]
        levels.append(torch.sort(cur).values)
    top_level_of = torch.zeros(n, dtype=torch.int64, device=dev)
    for li, lv in enumerate(levels):
        top_level_of[lv] = li
    cent = None  # centroids of the bucketed kNN: the coarsest level with ~n/1024 nodes
    for lv in levels:
        if lv.numel() <= max(64, n // 768):
            cent = lv
            break
    E_src, E_dst, E_prio = [], [], []

    def add_edges(src, dst, prio):
        ok = (dst >= 0) & (dst != src)
        E_src.append(src[ok])
        E_dst.append(dst[ok])
        E_prio.append(prio[ok].to(torch.int64))

    for li, ids in enumerate(levels):
        if ids.numel() < 2:
            continue
        use_cent = cent if (cent is not None and ids.numel() > 8 * cent.numel()) else None
        if li == 0:
            nb = _knn_subset(x, ids, min(k0, ids.numel() - 1), use_cent)
        elif ids.numel() <= k_upper + 1:  # top level: clique
            nb = ids[None, :].expand(ids.numel(), ids.numel()).clone()
            nb[nb == ids[:, None]] = -1
        else:
            cand = _knn_subset(x, ids, min(pool, ids.numel() - 1), use_cent)
            nb = _diversify(x, ids, cand, min(k_upper, cand.shape[1]))
        src = ids[:, None].expand_as(nb).reshape(-1)
        rank = torch.arange(nb.shape[1], device=dev)[None, :].expand_as(nb).reshape(-1)
        if li == 0:
            base = torch.full_like(rank, 200)
        else:  # the node's own top level ranks before its kNN list, lower levels after it
            is_top = (top_level_of[ids] == li)[:, None].expand_as(nb).reshape(-1)
            base = torch.where(is_top, torch.full_like(rank, 100), torch.full_like(rank, 300 + 20 * li))
        add_edges(src, nb.reshape(-1), base + rank)
        if li >= 1 and li + 1 < len(levels):  # parent -> child edges
            parents = levels[li + 1]
            par = _nearest_parent(x, ids, parents)
            sim = (x[ids] * x[parents[par]]).sum(1)
            o = torch.argsort(par * 4.0 - sim.double())  # by parent, most similar child first
            par_s, child_s = par[o], ids[o]
            first = torch.ones_like(par_s, dtype=torch.bool)
            first[1:] = par_s[1:] != par_s[:-1]
            seg_start = torch.nonzero(first).squeeze(1)
            seg_id = torch.cumsum(first.to(torch.int64), 0) - 1
            within = torch.arange(par_s.numel(), device=dev) - seg_start[seg_id]
            okc = within < child_cap
            add_edges(parents[par_s][okc], child_s[okc], within[okc])
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


def graph_stats(offsets: torch.Tensor) -> dict:
    deg = (offsets[1:] - offsets[:-1]).float()
    return {"nodes": int(deg.numel()), "edges": int(offsets[-1].item()),
            "deg_mean": float(deg.mean().item()), "deg_max": int(deg.max().item()),
            "deg_min": int(deg.min().item())}
