"""Second, independent restatement (pure Python + numpy.float32 scalars) of the
LEANN layer search, used ONLY to cross-check the C oracle on small cases.
Follows src/core/leann.rs:899-988 and distance.rs:71-88; heap rules are the
published Rust std BinaryHeap algorithm ([external], see oracle header)."""
import numpy as np

f32 = np.float32


def cosine(a, b):
    dot = f32(0); na = f32(0); nb = f32(0)
    for x, y in zip(a, b):
        dot = f32(dot + f32(x * y)); na = f32(na + f32(x * x)); nb = f32(nb + f32(y * y))
    norm = np.sqrt(f32(na * nb), dtype=f32)
    if norm == 0:
        return f32(1.0)
    return f32(f32(1.0) - f32(dot / norm))


class RustHeap:
    """Max-heap w.r.t. key(), mirroring BinaryHeap push/pop/into_vec."""

    def __init__(self, key):
        self.data = []
        self.key = key

    def _le(self, a, b):
        return self.key(a) <= self.key(b)

    def _sift_up(self, start, pos):
        elt = self.data[pos]
        while pos > start:
            parent = (pos - 1) // 2
            if self._le(elt, self.data[parent]):
                break
            self.data[pos] = self.data[parent]
            pos = parent
        self.data[pos] = elt

    def push(self, it):
        self.data.append(it)
        self._sift_up(0, len(self.data) - 1)

    def pop(self):
        if not self.data:
            return None
        item = self.data.pop()
        if self.data:
            item, self.data[0] = self.data[0], item
            end = len(self.data)
            pos = 0
            elt = self.data[0]
            child = 1
            while child <= max(end - 2, 0) and end >= 2:
                if self._le(self.data[child], self.data[child + 1]):
                    child += 1
                self.data[pos] = self.data[child]
                pos = child
                child = 2 * pos + 1
            if child == end - 1:
                self.data[pos] = self.data[child]
                pos = child
            self.data[pos] = elt
            self._sift_up(0, pos)
        return item


def leann_search_layer(off, nb, vectors, q, entry, ef):
    n = len(off) - 1
    visited = set()
    cand = RustHeap(lambda t: (-float(t[0]), -int(t[1])))  # Reverse<(d,id)>
    res = RustHeap(lambda t: (float(t[0]), int(t[1])))
    ed = cosine(q, vectors[entry])
    visited.add(entry)
    cand.push((ed, entry)); res.push((ed, entry))
    while True:
        cur = cand.pop()
        if cur is None:
            break
        d, cid = cur
        if res.data and len(res.data) >= ef and d > res.data[0][0]:
            break
        if cid >= n:
            continue
        unv = []
        for x in nb[int(off[cid]):int(off[cid + 1])]:
            x = int(x)
            if x not in visited:
                visited.add(x); unv.append(x)
        for x in unv:
            nd = cosine(q, vectors[x])
            if len(res.data) < ef or (not res.data) or nd < res.data[0][0]:
                cand.push((nd, x)); res.push((nd, x))
                if len(res.data) > ef:
                    res.pop()
    out = list(res.data)
    out.sort(key=lambda t: float(t[0]))  # Python sort is stable, like slice::sort_by
    return [t[1] for t in out], [t[0] for t in out]


# ---- extension: two-level search (docs/leann-specification.md:223-275, Algorithm 2) ----
def pq_tables(codebooks, q):
    """build_distance_tables, pq.rs:307-338: per (subquantizer, centroid) the left-fold sum of
    (a - b)^2 in f32."""
    m, K, dsub = codebooks.shape
    t = np.zeros((m, K), dtype=f32)
    for j in range(m):
        sub = q[j * dsub:(j + 1) * dsub]
        for c in range(K):
            s = f32(0)
            for a, b in zip(sub, codebooks[j, c]):
                df = f32(a - b)
                s = f32(s + f32(df * df))
            t[j, c] = s
    return t


def table_distance(tables, codes):
    s = f32(0)
    for j, c in enumerate(codes):
        s = f32(s + tables[j, int(c)])
    return np.sqrt(s, dtype=f32)


def two_level_search(off, nb, vectors, codebooks, codes, q, entry, k, ef, ratio):
    """Independent restatement of the rules fixed in oracle/islands_oracle.c
    (orc_two_level_search), written with sets and sorted() instead of arrays."""
    n = len(off) - 1
    ef = max(ef, k)
    tables = pq_tables(codebooks, q)
    visited = {entry}
    exact = {entry: cosine(q, vectors[entry])}     # every promoted node
    expanded = set()
    approx = {}                                     # AQ: id -> d_approx
    promoted = set()
    n_exact, n_approx = 1, 0

    def key(item):
        d, i = item
        d = float(d)
        return (d + 0.0 if d == d else float("inf"), d != d, i)

    def current_r():
        return sorted(((d, i) for i, d in exact.items()), key=key)[:ef]

    while True:
        r = current_r()
        cand = [i for _, i in r if i not in expanded]
        if not cand:
            break
        v = cand[0]
        expanded.add(v)
        if v >= n:
            continue
        for x in nb[int(off[v]):int(off[v + 1])]:
            x = int(x)
            if x in visited:
                continue
            visited.add(x)
            approx[x] = table_distance(tables, codes[x])
            n_approx += 1
        if not approx:
            continue
        ntop = int(np.ceil(f32(f32(ratio) * f32(len(approx)))))
        ntop = min(max(ntop, 1), len(approx))
        order = sorted(((d, i) for i, d in approx.items()), key=key)[:ntop]
        for _, i in order:
            if i in promoted:
                continue
            promoted.add(i)
            exact[i] = cosine(q, vectors[i])
            n_exact += 1
    r = current_r()[:k]
    return [i for _, i in r], [d for d, _ in r], n_exact, n_approx
