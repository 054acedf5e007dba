"""Generates tests/golden/two_level_small.npz: frozen cases of the two-level search (EXTENSION:
docs/leann-specification.md:223-275; the reference has no implementation and no vectors) with the
answers of oracle/islands_oracle.c::orc_two_level_search at the time of writing, so that neither
the definition nor the HIP path can drift unnoticed.  The graphs are those of search_small.npz.
    python tests/golden/make_two_level_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import oracle as orc  # noqa: E402
from test_two_level_cpu import make_pq  # noqa: E402

CASES = ((10, 32, 0.1), (5, 5, 0.5), (10, 64, 0.25), (10, 48, 1.0))


def main():
    orc.build()
    z = np.load(os.path.join(ROOT, "tests", "golden", "search_small.npz"))
    rows, q = z["rows"], z["queries"]
    cb, codes = make_pq(rows, 6, 16, 104)
    out = {"codebooks": cb, "codes": codes}
    for metric in range(4):
        csr = orc.Csr(z[f"m{metric}_offsets"], z[f"m{metric}_neighbors"], entry_point=int(z[f"m{metric}_entry"]))
        for (k, ef, a) in CASES:
            ids = np.full((q.shape[0], k), np.iinfo(np.uint64).max, np.uint64)
            dist = np.zeros((q.shape[0], k), np.float32)
            cnt = np.zeros(q.shape[0], np.uint32)
            ctr = np.zeros((q.shape[0], 4), np.uint64)
            for i in range(q.shape[0]):
                r = orc.two_level_search(csr, rows, cb, codes, q[i], k, ef, a, metric=metric)
                assert r.status == 0
                c = r.ids.size
                ids[i, :c], dist[i, :c], cnt[i] = r.ids, r.dist, c
                ctr[i] = [r.counters[f] for f in ("expansions", "edges", "evals", "pushes")]
            tag = f"m{metric}_k{k}_ef{ef}_a{int(a * 100)}"
            out[tag + "_ids"], out[tag + "_dist"], out[tag + "_cnt"], out[tag + "_ctr"] = ids, dist, cnt, ctr
    path = os.path.join(ROOT, "tests", "golden", "two_level_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
