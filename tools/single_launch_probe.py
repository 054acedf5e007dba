#!/usr/bin/env python3
"""How long do the driver's 20 batches of 1024 queries take when the device sees them as ONE
ticketed launch (the ideal device-side queue), next to 20 separate launches?

    python tools/single_launch_probe.py [--nodes 10000000] > profiles/r04_single_launch_20480.json

Legs, each after a warm-up of 5 ordinary batches, each over the same 20 x 1024 distinct queries:
  separate_16 / separate_32 : 20 calls of isl_search_batch_device_async, 16 / all 20 in flight
  single_20480              : ONE call of 20 480 queries
  split_16384_4096          : one call of 16 384 and one of 4 096, both in flight
End time = wall time from the first enqueue to the last completion (host clock around
torch.cuda.synchronize on both sides)."""
import argparse
import gc
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import islands_amd as ia
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batches", type=int, default=20)
    ap.add_argument("--nq", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--age-prio", default="", help="comma list of ISL_AGE_PRIO settings (0 = off) to run every leg under, "
                    "interleaved in one process: waves raise their issue priority every h expansions of their query "
                    "(needs the kernel of commit 18c59ad, which read that variable: measured without effect, "
                    "profiles/r04_age_prio_probe.json, and taken out again)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    N, d, nq, k, ef, B = a.nodes, a.dim, a.nq, 10, 128, a.batches
    x = synth.make_rows(N, d, 0, N, device=dev)
    off, nb, entry = synth.build_graph(x, m0=60)
    idx = ia.LeannIndex.from_device_csr(off.data_ptr(), nb.data_ptr(), N, entry, d, ia.LeannConfig.paper_default(), device=0)
    idx.set_embeddings(None, device_ptr=x.data_ptr(), n=N, d=d)
    total = B * nq
    idx.prepare(total, ef, k, 32)
    # warm-up batches come first in the query stream, the 20 timed ones after them
    warm = [synth.make_rows(N, d, b * nq, nq, device=dev, query=True).contiguous() for b in range(5)]
    qall = torch.cat([synth.make_rows(N, d, (5 + b) * nq, nq, device=dev, query=True) for b in range(B)], 0).contiguous()
    oi = torch.zeros((total, k), dtype=torch.int64, device=dev)
    od = torch.zeros((total, k), dtype=torch.float32, device=dev)
    oc = torch.zeros(total, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def submit(lo, n):
        return idx.search_batch_device_async(qall[lo:lo + n].data_ptr(), n, d, k, ef, oi[lo:lo + n].data_ptr(),
                                             od[lo:lo + n].data_ptr(), oc[lo:lo + n].data_ptr())

    def warmup():
        toks = [idx.search_batch_device_async(w.data_ptr(), nq, d, k, ef, oi[:nq].data_ptr(), od[:nq].data_ptr(),
                                              oc[:nq].data_ptr()) for w in warm]
        for t in toks:
            idx.wait_stats(t)
        torch.cuda.synchronize()

    def leg(plan, depth):
        """plan = [(lo, n), ...] calls in order, `depth` in flight."""
        warmup()
        pend, done_at, evals = [], [], 0
        t0 = time.perf_counter()
        for (lo, n) in plan:
            pend.append(submit(lo, n))
            if len(pend) >= depth:
                evals += idx.wait_stats(pend.pop(0))["evals"]
                done_at.append(round((time.perf_counter() - t0) * 1e3, 3))
        while pend:
            evals += idx.wait_stats(pend.pop(0))["evals"]
            done_at.append(round((time.perf_counter() - t0) * 1e3, 3))
        torch.cuda.synchronize()
        end = (time.perf_counter() - t0) * 1e3
        return {"end_ms": round(end, 3), "queries_per_s": round(total / end * 1e3), "completions_ms": done_at,
                "evals_per_query": round(evals / total, 1)}

    plans = {
        "separate_16": ([(b * nq, nq) for b in range(B)], 16),
        "separate_32": ([(b * nq, nq) for b in range(B)], 32),
        "single_%d" % total: ([(0, total)], 1),
        "split_%d_%d" % (total * 4 // 5, total - total * 4 // 5): ([(0, total * 4 // 5), (total * 4 // 5, total - total * 4 // 5)], 2),
        "four_calls_of_%d" % (total // 4): ([(i * (total // 4), total // 4) for i in range(4)], 4),
    }
    gc.collect()
    gc.disable()
    res = {"workload": f"{N} x {d} f32 rows, {B} batches of {nq} distinct queries, k={k}, ef={ef}", "legs": {}}
    ref = None
    ages = [int(v) for v in a.age_prio.split(",") if v != ""] or [None]
    for r in range(a.reps):
        for name, (plan, depth) in plans.items():
            for age in ages:
                if age is not None:
                    os.environ["ISL_AGE_PRIO"] = str(age)
                out = leg(plan, depth)
                ids = oi.clone()
                if ref is None:
                    ref = ids
                out["ids_equal_first_leg"] = bool((ids == ref).all().item())
                key = name if age is None else f"{name}@age_prio={age}"
                res["legs"].setdefault(key, []).append(out)
                print(f"[probe] rep {r} {key}: end {out['end_ms']} ms, {out['queries_per_s']} q/s", file=sys.stderr, flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
