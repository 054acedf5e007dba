#!/usr/bin/env python3
"""Per-kernel statistics of the TIMED region of a bench.py run from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d DIR -o p -- python3 bench.py --steps K --warmup W \\
        --no-traffic --no-cpu-baseline --no-host-path
    python tools/kernel_stats_timed.py DIR --steps K --out profiles/r03_kernel_stats_timed.csv

`rocprofv3 --stats` averages over the whole process: the harness's graph build and ground truth
(hipBLASLt / at::native kernels) and the empty launches isl_index_prepare makes on every lane (one
workgroup each) dilute the product kernels' figures (round 2: 3.28 ms printed for a kernel whose real
launches averaged 5.52 ms).  Here: product kernels only (name filter), empty launches dropped by
grid size, and of the search kernel only the last K real dispatches -- the timed steps -- so that
average duration x dispatches is the kernel time of the timed region."""
import argparse
import csv
import glob
import os
import re
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--steps", type=int, required=True)
    ap.add_argument("--regex", default=r"leann_search|publish_kernel|classify|merge_topk|pq_tables|copy_u")
    ap.add_argument("--main", default="leann_search_fast", help="kernel whose last --steps real dispatches mark the timed region")
    ap.add_argument("--out", default="-")
    a = ap.parse_args()
    files = glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit(f"no *kernel_trace.csv under {a.dir}")
    rows = []
    for f in files:
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    rx = re.compile(a.regex)

    def grid(r):
        if "Grid_Size" in r and r["Grid_Size"]:
            return int(r["Grid_Size"])
        return int(r.get("Grid_Size_X", 0)) * max(1, int(r.get("Grid_Size_Y", 1))) * max(1, int(r.get("Grid_Size_Z", 1)))

    def wg(r):
        if "Workgroup_Size" in r and r["Workgroup_Size"]:
            return int(r["Workgroup_Size"])
        return int(r.get("Workgroup_Size_X", 1)) * max(1, int(r.get("Workgroup_Size_Y", 1))) * max(1, int(r.get("Workgroup_Size_Z", 1)))

    prod = [r for r in rows if rx.search(r["Kernel_Name"])]
    for r in prod:
        r["_s"], r["_e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        r["_real"] = grid(r) > wg(r)  # more than one workgroup: not an empty launch of isl_index_prepare
    main_real = sorted((r for r in prod if a.main in r["Kernel_Name"] and r["_real"]), key=lambda r: r["_s"])
    if len(main_real) < a.steps:
        sys.exit(f"{len(main_real)} real dispatches of {a.main}, expected at least {a.steps}")
    timed = main_real[-a.steps:]
    t0, t1 = timed[0]["_s"], max(r["_e"] for r in timed)
    inside = [r for r in prod if r["_real"] and r["_s"] >= t0 and r["_s"] <= t1]
    by = {}
    for r in inside:
        by.setdefault(r["Kernel_Name"], []).append(r["_e"] - r["_s"])
    span = (t1 - t0) / 1e6
    out = sys.stdout if a.out == "-" else open(a.out, "w")
    w = csv.writer(out)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "TimedRegionSpanMs",
                "SumOverSpan"])
    for name, d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([name, len(d), sum(d), round(sum(d) / len(d), 1), min(d), max(d), round(span, 3),
                    round(sum(d) / 1e6 / span, 3)])
    if out is not sys.stdout:
        out.close()
        print(f"{len(inside)} dispatches of {len(by)} product kernels in the timed region ({span:.3f} ms) -> {a.out}")


if __name__ == "__main__":
    main()
