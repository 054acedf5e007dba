#!/usr/bin/env python3
"""Where the time of the host-buffer search path goes: per call, the host time spent inside
isl_search_batch_async (staging memcpy + enqueue) and inside isl_search_wait_stats, next to the
device-resident asynchronous path, on the bench workload at a reduced node count."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import islands_amd as ia
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--nq", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--depth", type=int, default=8)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    N, d, nq, k, ef = a.nodes, a.dim, a.nq, 10, 128
    x = synth.make_rows(N, d, 0, N, device=dev)
    off, nb, entry = synth.build_graph(x, m0=60)
    idx = ia.LeannIndex.from_device_csr(off.data_ptr(), nb.data_ptr(), N, entry, d, ia.LeannConfig.paper_default(), device=0)
    idx.set_embeddings(None, device_ptr=x.data_ptr(), n=N, d=d)
    idx.prepare(nq, ef, k, a.depth)
    qs = [synth.make_rows(N, d, b * nq, nq, device=dev, query=True).contiguous() for b in range(4)]
    qh = [q.cpu().numpy() for q in qs]
    torch.cuda.synchronize()
    outs_d = [(torch.zeros((nq, k), dtype=torch.int64, device=dev), torch.zeros((nq, k), dtype=torch.float32, device=dev),
               torch.zeros(nq, dtype=torch.int32, device=dev)) for _ in range(a.depth)]
    outs_h = [(np.zeros((nq, k), np.uint64), np.zeros((nq, k), np.float32), np.zeros(nq, np.uint32)) for _ in range(a.depth)]

    def run(kind, steps):
        te = tw = 0.0
        pend = []
        trace = []
        t0 = time.perf_counter()
        for s in range(steps):
            t1 = time.perf_counter()
            if kind == "device":
                o = outs_d[s % a.depth]
                tok = idx.search_batch_device_async(qs[s % 4].data_ptr(), nq, d, k, ef, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
            else:
                tok = idx.search_batch_async(qh[s % 4], k, ef, out=outs_h[s % a.depth])
            te += time.perf_counter() - t1
            trace.append(round((time.perf_counter() - t1) * 1e6))
            pend.append(tok)
            if len(pend) >= a.depth:
                t1 = time.perf_counter()
                idx.wait_stats(pend.pop(0))
                tw += time.perf_counter() - t1
        while pend:
            t1 = time.perf_counter()
            idx.wait_stats(pend.pop(0))
            tw += time.perf_counter() - t1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return {"kind": kind, "steps": steps, "ms_per_step": round(dt / steps * 1e3, 3), "qps": round(steps * nq / dt),
                "enqueue_ms_per_call": round(te / steps * 1e3, 3), "wait_ms_per_call": round(tw / steps * 1e3, 3),
                "enqueue_us_trace": trace}

    for kind in ("host", "host", "device", "host"):
        print(json.dumps(run(kind, a.steps)), flush=True)
    # the staging memcpy alone
    buf = np.empty_like(qh[0])
    t0 = time.perf_counter()
    for _ in range(50):
        np.copyto(buf, qh[0])
    print(json.dumps({"numpy_copy_3MB_ms": round((time.perf_counter() - t0) / 50 * 1e3, 3)}))


if __name__ == "__main__":
    main()
