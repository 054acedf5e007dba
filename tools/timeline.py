#!/usr/bin/env python3
"""Where the fill and drain of a short run go: start / end tick of every query of a 20 x 1024 run
(ISL_TIMELINE, search.hip), as one launch of 20 480 queries and as 20 launches, turned into the
number of queries in flight over time and the distribution of a query's duration by start time.

    python tools/timeline.py [--nodes 10000000] > profiles/r04_timeline_20480.json"""
import argparse
import json
import os
import sys
import tempfile

os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
TL = os.path.join(tempfile.gettempdir(), f"isl_timeline_{os.getpid()}.bin")
os.environ["ISL_TIMELINE"] = TL
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import islands_amd as ia
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth


def read_records():
    raw = np.fromfile(TL, dtype=np.uint64)
    os.remove(TL)
    recs, i = [], 0
    while i < raw.size:
        assert raw[i] == 0x154C494E45, hex(int(raw[i]))
        n = int(raw[i + 1])
        r = raw[i + 2:i + 2 + 2 * n].reshape(n, 2)
        recs.append(np.stack([r[:, 0], r[:, 0] + (r[:, 1] & np.uint64(0xFFFFFFFFFF)), r[:, 1] >> np.uint64(40)], 1).astype(np.int64))
        i += 2 + 2 * n
    return recs


def summarise(recs):
    t = np.concatenate(recs, 0)
    hops = t[:, 2]
    ok = t[:, 1] > 0
    t0 = t[ok, 0].min()
    start = (t[:, 0] - t0) / 100.0  # us (100 MHz ticks)
    end = (t[:, 1] - t0) / 100.0
    dur = end - start
    total = end.max()
    grid = np.arange(0.0, total + 250.0, 250.0)
    inflight = [(int(g), int(((start <= g) & (end > g)).sum())) for g in grid]
    done = [(int(g), int((end <= g).sum())) for g in grid]
    # duration by start time, in 2 ms windows
    by_start = []
    for lo in np.arange(0.0, start.max() + 1.0, 2000.0):
        m = (start >= lo) & (start < lo + 2000.0)
        if m.sum():
            by_start.append({"start_ms": [lo / 1e3, (lo + 2000.0) / 1e3], "queries": int(m.sum()),
                             "dur_us_p50": round(float(np.median(dur[m])), 1),
                             "dur_us_p99": round(float(np.percentile(dur[m], 99)), 1),
                             "dur_us_max": round(float(dur[m].max()), 1),
                             "us_per_hop_p50": round(float(np.median(dur[m] / np.maximum(hops[m], 1))), 2)})
    last = np.argsort(end)[-5:]
    return {"queries": int(t.shape[0]), "end_us": round(float(total), 1),
            "last_start_us": round(float(start.max()), 1),
            "dur_us": {"p50": round(float(np.median(dur)), 1), "p90": round(float(np.percentile(dur, 90)), 1),
                       "p99": round(float(np.percentile(dur, 99)), 1), "max": round(float(dur.max()), 1)},
            "hops": {"p50": int(np.median(hops)), "p99": int(np.percentile(hops, 99)), "max": int(hops.max())},
            "slot_time_ms": round(float(dur.sum()) / 1e3, 1),
            "last_five_to_finish": [{"start_us": round(float(start[i]), 1), "end_us": round(float(end[i]), 1),
                                     "hops": int(hops[i])} for i in last],
            "in_flight_every_250us": inflight, "completed_every_250us": done, "by_start_window": by_start}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batches", type=int, default=20)
    ap.add_argument("--nq", type=int, default=1024)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    N, d, nq, k, ef, B = a.nodes, a.dim, a.nq, 10, 128, a.batches
    x = synth.make_rows(N, d, 0, N, device=dev)
    off, nb, entry = synth.build_graph(x, m0=60)
    idx = ia.LeannIndex.from_device_csr(off.data_ptr(), nb.data_ptr(), N, entry, d, ia.LeannConfig.paper_default(), device=0)
    idx.set_embeddings(None, device_ptr=x.data_ptr(), n=N, d=d)
    total = B * nq
    idx.prepare(total, ef, k, 32)
    warm = [synth.make_rows(N, d, b * nq, nq, device=dev, query=True).contiguous() for b in range(5)]
    qall = torch.cat([synth.make_rows(N, d, (5 + b) * nq, nq, device=dev, query=True) for b in range(B)], 0).contiguous()
    oi = torch.zeros((total, k), dtype=torch.int64, device=dev)
    od = torch.zeros((total, k), dtype=torch.float32, device=dev)
    oc = torch.zeros(total, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def run(plan, depth):
        toks = [idx.search_batch_device_async(w.data_ptr(), nq, d, k, ef, oi[:nq].data_ptr(), od[:nq].data_ptr(),
                                              oc[:nq].data_ptr()) for w in warm]
        for t in toks:
            idx.wait_stats(t)
        torch.cuda.synchronize()
        if os.path.exists(TL):
            os.remove(TL)  # the warm-up's records
        pend = []
        for (lo, n) in plan:
            pend.append((lo, n, idx.search_batch_device_async(qall[lo:lo + n].data_ptr(), n, d, k, ef, oi[lo:lo + n].data_ptr(),
                                                              od[lo:lo + n].data_ptr(), oc[lo:lo + n].data_ptr())))
            if len(pend) >= depth:
                lo_, n_, tok = pend.pop(0)
                idx.wait_stats(tok)
        while pend:
            lo_, n_, tok = pend.pop(0)
            idx.wait_stats(tok)
        torch.cuda.synchronize()
        return summarise(read_records())

    res = {"workload": f"{N} x {d} f32 rows, {B} x {nq} distinct queries, k={k}, ef={ef}; times in us from the first "
                       "query's start (s_memrealtime, 100 MHz)",
           "single_launch": run([(0, total)], 1),
           "separate_launches_all_in_flight": run([(b * nq, nq) for b in range(B)], 32)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
