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
