#!/usr/bin/env python3
"""Headline benchmark of the LEANN search hot path on MI355X.

Metric (BASELINE.json): queries/sec at recall@10 >= 0.95 on synthetic 10M x 768 vectors,
ef = 128, k = 10, query batch 1024, embeddings resident in HBM (in-memory provider).
One "step" = one pass of the hot path (isl_search_batch_device) over one batch of queries
that is already resident in HBM.

    python bench.py --gpus 1 --steps K --warmup W
    python bench.py --gpus N ...            (no launcher: starts N ranks itself, before any GPU call)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU (N > 1), mode "shard" (default, the north star's layout): the index is sharded by
node-id range, every rank searches the whole query batch in its own sub-graph, the
per-shard top-k are exchanged with one RCCL all-gather and merged, all inside the library
(isl_sharded_submit / isl_sharded_result: MultiIndexSearcher::search semantics,
src/core/search.rs:211-237).  Total index size and
query count are fixed as N grows -> "strong" scaling.  Mode "replica": every rank holds the
full index and answers its own batch, no data-path collective -> "weak".

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys

# one hardware queue per search lane in flight (the HIP default is 4 queues per process)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
GRAPH_BUILDERS = {
    "harness": "the harness (tools/synth.py::build_graph: k-means medoid levels flattened into one layer, "
               "diversified links, parent->child and reverse edges)",
    "knn": "exact 30-NN lists (the library's brute force) + reverse edges, truncated to 60, entry = medoid",
    "product": "the library's isl_index_build = the reference's LeannIndex::build rule (leann.rs:560-833), "
               "4096 nodes per step",
}


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def algorithmic_bytes(st: dict, d: int, k: int, elem: int = 4) -> float:
    """V*d*s + 4*E + 8*H + 4*d + 12*k per query, summed over a batch."""
    return (st["evals"] * d * elem + 4 * st["edges"] + 8 * st["expansions"] +
            st["queries"] * (4 * d + 12 * k))


def host_cpu_info() -> dict:
    """What the CPU baseline ran on: `nproc` (logical CPUs this process may run on), the model name
    lscpu prints, the physical cores among them (distinct (package, core) pairs of /proc/cpuinfo), and
    the cgroup CPU quota when there is one (a box may hand a process fewer CPU-seconds per second than
    it has cores)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(os.cpu_count() or 1))
    model, cores, cur = "", set(), {}
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh.read().split("\n") + [""]:
                if ":" in line:
                    k_, v_ = (t.strip() for t in line.split(":", 1))
                    cur[k_] = v_
                elif cur:
                    if not model:
                        model = cur.get("model name", "")
                    if int(cur.get("processor", -1)) in allowed:
                        cores.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
                    cur = {}
    except OSError:
        pass
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q_, p_ = fh.read().split()[:2]
            quota = None if q_ == "max" else round(int(q_) / int(p_), 2)
    except (OSError, ValueError):
        pass
    return {"nproc": len(allowed), "logical_cpus_of_the_box": os.cpu_count(), "model": model,
            "physical_cores_available": len(cores) or len(allowed), "cgroup_cpu_quota": quota}


def cpu_baseline(x, offsets, neighbours, entry, qsets, k, ef, leg_s=10.0):
    """Times the CPU oracle (the restated reference algorithm) on this box's host cores: one thread
    (the reference is single-threaded per query and per batch, search.rs:179-181) and all cores, each
    leg for `leg_s` seconds of wall time over the run's query batches in turn (a leg of a second would
    be dominated by thread start-up).  The oracle is only the checker / baseline here, never the
    measured product."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc

    orc.build()
    t0 = time.time()
    xv = x.cpu().numpy()
    off = offsets.cpu().numpy().astype(np.uint64)
    nb = neighbours.cpu().numpy().astype(np.uint64)
    csr = orc.Csr(off, nb, entry_point=entry)
    q = np.concatenate([qb.cpu().numpy() for qb in qsets], 0)
    log(f"cpu_baseline: host copy of the index took {time.time() - t0:.1f}s")
    for i in range(4):  # page the index in
        orc.leann_search(csr, xv, q[i], k, ef, copy_per_node=True)
    host = host_cpu_info()
    threads = max(1, host["physical_cores_available"])  # all physical cores

    def leg(nthreads, seconds):
        done = [0] * nthreads
        start = threading.Barrier(nthreads + 1)

        def work(t):
            start.wait()
            end = time.perf_counter() + seconds
            i, n = t, 0
            while time.perf_counter() < end:  # thread t answers queries t, t + T, t + 2T, ... of the batches
                orc.leann_search(csr, xv, q[i % q.shape[0]], k, ef, copy_per_node=True)
                i += nthreads
                n += 1
            done[t] = n

        ths = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
        for t in ths:
            t.start()
        start.wait()  # the clock starts once every thread exists
        t0 = time.perf_counter()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
        return sum(done), dt

    n1, dt1 = leg(1, leg_s)
    nm, dtm = leg(threads, leg_s)
    return {
        "value": round(nm / dtm, 2), "unit": "queries/s", "cores": threads, "kind": "port",
        "value_1thread": round(n1 / dt1, 2),
        "threads_used": threads, "nproc": host["nproc"], "cpu_model": host["model"],
        "host": host,
        "sample": f"{nm} queries in {dtm:.1f} s on {threads} threads (thread t takes every {threads}th query of "
                  f"the run's {len(qsets)} batches; ctypes releases the GIL), {n1} queries in {dt1:.1f} s on 1 "
                  "thread; copy-per-node provider as in leann.rs:145-154; same graph, ef and k",
    }


def measure_traffic(args):
    """roofline.traffic for THIS configuration: FETCH_SIZE of the search kernel's dispatches in a
    child run of this script under rocprofv3 (MI355X_MICROARCH.md, HBM section: the counter is in
    KiB and tallies every 128-byte request as 64 bytes on gfx950, so bytes = value * 1024 * 2;
    the factor was re-checked on this kernel's access pattern, profiles/r01_pmc_fetch.json)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        raise RuntimeError("rocprofv3 not found")
    out = tempfile.mkdtemp(prefix="isl_pmc_", dir="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--traffic-child", "--gpus", "1", "--steps", "4",
             "--warmup", "1", "--pipeline", "1", "--nodes", str(args.nodes), "--dim", str(args.dim), "--nq",
             str(args.nq), "--k", str(args.k), "--ef", str(args.ef), "--per-cluster", str(args.per_cluster),
             "--row-dtype", args.row_dtype, "--dataset", args.dataset, "--latent", str(args.latent), "--distinct-batches",
             str(args.distinct_batches), "--graph", args.graph, "--no-cpu-baseline", "--no-host-path", "--no-traffic"] + \
        (["--distinct-leaves"] if args.distinct_leaves else [])
    cmd = [rp, "--pmc", "FETCH_SIZE", "--kernel-include-regex", "leann_search_fast", "-d", out, "-o", "p",
           "--output-format", "csv", "--"] + child
    env = dict(os.environ, TMPDIR="/tmp")
    env.pop("ISL_TRAFFIC_BYTES", None)
    try:
        pr = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
        if pr.returncode != 0 or not files:
            raise RuntimeError(f"rocprofv3 child rc={pr.returncode}: {pr.stderr.decode(errors='replace')[-200:]}")
        vals = []
        with open(files[0]) as fh:
            for row in csv.DictReader(fh):
                # (the empty launches isl_index_prepare makes have one workgroup: Grid_Size 64)
                if (row["Counter_Name"] == "FETCH_SIZE" and "leann_search_fast" in row["Kernel_Name"]
                        and int(row["Grid_Size"]) > 64):
                    vals.append(float(row["Counter_Value"]))
        line = [l for l in pr.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
        child_res = json.loads(line[-1]) if line else {}
    finally:
        shutil.rmtree(out, ignore_errors=True)
    if len(vals) < 2:
        raise RuntimeError(f"{len(vals)} profiled dispatches of the search kernel")
    vals = vals[1:]  # the first launch warms the caches
    traffic = float(np.mean(vals)) * 1024.0 * 2.0
    alg = child_res.get("roofline", {}).get("algorithmic_bytes_per_launch")
    return {"traffic": round(traffic, 0),
            "traffic_source": f"rocprofv3 --pmc FETCH_SIZE, child run of this configuration, mean of {len(vals)} "
                              "launches of leann_search_fast, x 1024 x 2 (KiB; gfx950 counts 128-B requests as 64 B)",
            "traffic_over_algorithmic": round(traffic / alg, 4) if alg else None}


def measure_neutral(args, graph):
    """`neutral_workload[_knn]` of the default line: this script once more, as a child, on 1M rows of dataset M with
    the same query batch, k, ef, steps and warm-up; graph = "product" (the library's own isl_index_build, i.e. the
    reference's LeannIndex::build rule: builder and search both the library's) or "knn" (exact nearest-neighbour
    lists by the library's brute force)."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--dataset", "M", "--graph", graph, "--nodes", "1000000", "--dim", str(args.dim), "--nq", str(args.nq),
           "--k", str(args.k), "--ef", str(args.ef), "--no-traffic", "--no-cpu-baseline", "--no-host-path", "--no-neutral-side"]
    t0 = time.time()
    pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    line = [l for l in pr.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
    if pr.returncode != 0 or not line:
        raise RuntimeError(f"child rc={pr.returncode}: {pr.stderr.decode(errors='replace')[-200:]}")
    r = json.loads(line[-1])
    return {"value": r["value"], "unit": r["unit"], "recall_at_10": r["recall_at_10"], "ms_per_step": r["ms_per_step"],
            "roofline_frac": r["roofline"]["frac"], "per_query": r["config"]["per_query"],
            "workload": r["config"]["workload"], "graph_build_s": r["config"]["graph_build_s"],
            "child_run_s": round(time.time() - t0, 1),
            "command": f"python bench.py --dataset M --graph {graph} --nodes 1000000 (same nq, k, ef, steps, warm-up)"}


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves -- as children of
    torch.distributed.run, one per GPU -- BEFORE this process has made any GPU call (it never does),
    and pass their exit code and rank 0's JSON line through.  Never a re-exec: a process that has
    initialised the GPU must not be replaced."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"--gpus {n} without WORLD_SIZE: launching {n} ranks on 127.0.0.1:{port}")
    return subprocess.run(cmd, env=env).returncode


def dry_run(args, world, rank, backend):
    """The launch plumbing without a card: rendezvous, ONE all-gather of a packed record of the
    run's (nq, k) over the process group, the count of ranks that got through it, the JSON line."""
    from islands_amd.sharded import record_bytes, record_views, shard_range

    nq, k = args.nq, args.k
    B = record_bytes(nq, k)
    rec = torch.zeros(B, dtype=torch.uint8)
    ids, dd, cnt = record_views(rec, nq, k)
    ids.fill_(rank); dd.fill_(float(rank)); cnt.fill_(k)
    gathered = torch.zeros((world, B), dtype=torch.uint8)
    if world > 1:
        dist.all_gather_into_tensor(gathered.view(-1), rec)
    else:
        gathered[0].copy_(rec)
    g_ids, g_dd, g_cnt = record_views(gathered, nq, k)
    ok = all(bool((g_ids[r] == r).all()) and bool((g_cnt[r] == k).all()) for r in range(world))
    done = torch.tensor([1 if ok else 0], dtype=torch.int64)
    if world > 1:
        dist.all_reduce(done)
    if rank == 0:
        print(json.dumps({"metric": "queries/sec @ recall@10>=0.95, 10Mx768 ef=128", "value": None,
                          "unit": "queries/s", "n_gpus": int(done.item()), "steps": args.steps, "warmup": args.warmup,
                          "dry_run": True, "backend": backend,
                          "config": {"workload": "dry run: no search", "shard_ranges": [
                              list(shard_range(args.nodes, r, world)) for r in range(world)]}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--nq", type=int, default=1024)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--ef", type=int, default=128)
    ap.add_argument("--per-cluster", type=int, default=1000)
    ap.add_argument("--mode", choices=["shard", "replica"], default="shard")
    ap.add_argument("--row-dtype", choices=["f32", "bf16"], default="f32",
                    help="storage type of the embedding rows (bf16: rows rounded to bf16, stored as "
                         "such, arithmetic still f32 on the widened values; not the headline config)")
    ap.add_argument("--dataset", choices=["G", "U", "M"], default="G",
                    help="G: the headline clustered mixture; U: i.i.d. uniform [-1,1) rows as in "
                         "benches/hnsw_benchmarks.rs:9-14 (SURVEY 8d asks for both; no cluster structure, "
                         "so recall at ef=128 is whatever the data allows); M: rows on a smooth --latent-dimensional "
                         "manifold (tools/synth.py::make_manifold: no clusters, no tree -- a dataset no graph builder "
                         "of this repository was designed around; use it with --graph knn)")
    ap.add_argument("--latent", type=int, default=16, help="dataset M: intrinsic dimension")
    ap.add_argument("--graph", choices=["harness", "knn", "product"], default="harness",
                    help="who builds the graph the search walks.  harness (headline): tools/synth.py::build_graph "
                         "(k-means medoid levels flattened into one layer, diversified links); knn: every node's 30 "
                         "exact nearest neighbours by the library's own brute force + reverse edges, truncated to 60 "
                         "-- no hierarchy, no diversification; product: the library's isl_index_build, i.e. the "
                         "reference's LeannIndex::build rule (leann.rs:560-833), batch mode")
    ap.add_argument("--distinct-leaves", action="store_true",
                    help="measurement aid (dataset G): consecutive queries come from different leaf clusters -- with "
                         "N / per-cluster leaves, any that many consecutive queries share no neighbourhood, so rows "
                         "fetched by one query in flight are not re-used by another (what part of the rate is "
                         "cross-query cache reuse: DESIGN section 4)")
    ap.add_argument("--ef-sweep", default="256,512",
                    help="when recall@10 at --ef misses 0.95: the larger ef values tried (one after the other, "
                         "until one reaches it); empty = none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-neutral-side", action="store_true",
                    help="skip the side measurement the default (headline) configuration carries in `neutral_workload`: "
                         "1M rows without a cluster tree (dataset M) on the exact-kNN graph the library's own brute force "
                         "builds, same nq / k / ef / steps, as a child process")
    ap.add_argument("--no-host-path", action="store_true", help="skip the host-buffer pipelined measurement")
    ap.add_argument("--no-traffic", action="store_true",
                    help="do not re-run this configuration under rocprofv3 --pmc FETCH_SIZE for roofline.traffic")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-replica", action="store_true",
                    help="multi-GPU shard runs: skip the extra replica-mode measurement")
    ap.add_argument("--distinct-batches", type=int, default=0,
                    help="query batches the steps cycle through, each with its own ground truth; default (0) = "
                         "steps + warmup, so that no two launches in flight (or anywhere in the run) traverse "
                         "the same queries (round 2 cycled 4 batches under 16 launches in flight)")
    ap.add_argument("--rehearse-shard", default="", metavar="r/R",
                    help="one GPU plays rank r of an R-rank sharded run of --nodes rows: it generates and indexes "
                         "only that shard's rows and answers the batches over them (no exchange) -- what one rank of "
                         "BASELINE config 4 (--nodes 100000000, 8 ranks) holds and does, measurable on one card")
    ap.add_argument("--dry-run", action="store_true",
                    help="rendezvous, one all-gather of a packed record over the process group and the JSON "
                         "line only (no GPU work): the launch plumbing of --gpus N, testable without a card")
    ap.add_argument("--pipeline", type=int, default=0,
                    help="searches kept in flight (isl_search_batch_device_async); 1 = synchronous; default 16 on "
                         "one GPU, 12 per rank in multi-GPU runs")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))  # this process stays off the GPU

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # ISL_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks (ranks
    # share a card, the exchange goes through host memory); the real runs use RCCL ("nccl").
    backend = os.environ.get("ISL_BENCH_BACKEND", "nccl")
    if args.dry_run:
        if world > 1:
            dist.init_process_group("gloo")
        return dry_run(args, world, rank, "gloo")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > ndev:
        raise SystemExit(f"bench.py: {world} ranks but {ndev} GPU(s) visible (RCCL needs one card per rank; "
                         "ISL_BENCH_BACKEND=gloo is the rehearsal mode for ranks that share a card)")
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{dev_index}"))
        else:
            dist.init_process_group("gloo")
    torch.cuda.set_device(dev_index)
    dev = torch.device(f"cuda:{dev_index}")
    local_rank = dev_index

    def all_gather_rows(out, inp):
        """out: [world * rows, ...] device tensor, inp: [rows, ...]; rank-major concatenation."""
        if backend == "nccl":
            dist.all_gather_into_tensor(out, inp)
        else:
            parts = [torch.zeros_like(inp, device="cpu") for _ in range(world)]
            dist.all_gather(parts, inp.cpu())
            out.copy_(torch.cat(parts, 0).to(out.device))

    import islands_amd as ia
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synth  # the synthetic-workload harness (data, bench graph, ground truth): not product code

    N, d, nq, k, ef = args.nodes, args.dim, args.nq, args.k, args.ef
    comm_info = {}

    def measure(mode):
        """One full measurement (setup, warm-up, timed steps) in `mode`; returns the result
        object and what the CPU baseline needs."""
        shard_mode = world > 1 and mode == "shard"
        if shard_mode:
            lo = rank * N // world
            hi = (rank + 1) * N // world
        elif args.rehearse_shard:
            r_, R_ = (int(v) for v in args.rehearse_shard.split("/"))
            lo, hi = r_ * N // R_, (r_ + 1) * N // R_
        else:
            lo, hi = 0, N
        n_local = hi - lo

        # ---------------- setup (untimed): data, graph, index upload, queries, ground truth
        t0 = time.time()
        if args.dataset == "U":
            x = synth.make_uniform(n_local, d, 42, device=dev, start=lo)  # this rank's rows only
        elif args.dataset == "M":
            x = synth.make_manifold(n_local, d, 42, device=dev, start=lo, latent=args.latent)
        else:
            x = synth.make_rows(N, d, lo, n_local, per_cluster=args.per_cluster, device=dev)
        x16 = None
        if args.row_dtype == "bf16":
            x16 = x.to(torch.bfloat16)       # the stored rows
            x = x16.to(torch.float32)        # their exact f32 images: graph, truth and CPU baseline use these
        torch.cuda.synchronize()
        log(f"rows [{lo},{hi}) generated in {time.time() - t0:.1f}s")
        t0 = time.time()
        cfg = ia.LeannConfig.paper_default()
        idx = None
        if args.graph == "knn":
            offsets, neighbours, entry = synth.build_knn_graph(x, k=30, m0=60, progress=log)
        elif args.graph == "product":
            # LeannIndex::build on the device (build.hip), 4096 nodes per step; its CsrGraph comes back through
            # to_bytes (bincode layout, api_index.hip): 75-byte config, then node_offsets and neighbors
            idx = ia.LeannIndex.build(x.cpu().numpy(), cfg, batch=4096, device=local_rank)
            raw = np.frombuffer(idx.to_bytes(), dtype=np.uint8)
            n_off = int(raw[75:83].view(np.uint64)[0])
            off_h = raw[83:83 + 8 * n_off].view(np.uint64)
            n_nb = int(raw[83 + 8 * n_off:91 + 8 * n_off].view(np.uint64)[0])
            nb_h = raw[91 + 8 * n_off:91 + 8 * n_off + 8 * n_nb].view(np.uint64)
            offsets = torch.from_numpy(off_h.astype(np.int64)).to(dev)
            neighbours = torch.from_numpy(nb_h.astype(np.int32)).to(dev)
            entry = idx.entry_point
            del raw
        else:
            offsets, neighbours, entry = synth.build_graph(x, m0=60)
        torch.cuda.synchronize()
        gst = synth.graph_stats(offsets)
        log(f"graph ({args.graph}) built in {time.time() - t0:.1f}s: {gst}")
        graph_build_s = time.time() - t0
        t0 = time.time()
        if idx is None:
            idx = ia.LeannIndex.from_device_csr(offsets.data_ptr(), neighbours.data_ptr(), n_local, entry,
                                                d, cfg, device=local_rank)
        if x16 is not None:
            idx.set_embeddings_bf16(None, device_ptr=x16.data_ptr(), n=n_local, d=d)
        else:
            idx.set_embeddings(None, device_ptr=x.data_ptr(), n=n_local, d=d)
        # every lane's buffers, the padded adjacency and the exact-kernel pool up front: nothing on
        # the search path allocates or synchronises for set-up afterwards (isl_index_prepare), so
        # the timed region does not depend on --warmup
        # One stream per search in flight: 16 (all lanes) is the best measured on one GPU; past ~20 streams
        # on a card the runtime's 32 hardware queues run out and the rate collapses (24 lanes: 0.39 M q/s),
        # so ranks that also run the exchange's side stream and RCCL's own keep a margin.
        # (a batch of fewer than 1024 queries fills a fraction of the chip's wave slots: more of them in
        # flight, up to the 32 lanes -- config 2's 256-query batches)
        # Round 4: all 32 lanes on one GPU.  Over a long run 16 and 32 in flight are the same rate;
        # over the driver's 20 steps all batches are then on the device from the start and the region ends
        # 0.3-0.8 ms earlier (profiles/r04_single_launch_20480.json: 14.6-14.9 ms against 15.0-15.7).
        auto_depth = 32 if world == 1 else 12
        depth = max(1, min(args.pipeline if args.pipeline > 0 else auto_depth, 32))
        idx.prepare(nq, ef, k, depth)
        torch.cuda.synchronize()
        log(f"index resident and {depth} search lanes prepared in {time.time() - t0:.1f}s")

        nb_batches = args.steps + args.warmup if args.distinct_batches <= 0 else \
            max(1, min(args.distinct_batches, args.steps + args.warmup))
        qsets, truths = [], []
        for b in range(nb_batches):
            # replica mode: every rank answers its own batches; shard mode: same batch on all ranks
            qoff = (b + (rank * nb_batches if (world > 1 and not shard_mode) else 0)) * nq
            if args.dataset == "U":
                q = synth.make_uniform(nq, d, 43 + qoff, device=dev)
            elif args.dataset == "M":  # out-of-sample points of the same density (their own seed stream)
                q = synth.make_manifold(nq, d, 4300 + qoff // max(nq, 1), device=dev, latent=args.latent)
            else:
                q = synth.make_rows(N, d, qoff, nq, per_cluster=args.per_cluster, device=dev, query=True,
                                    distinct_leaves=args.distinct_leaves)
            if args.row_dtype == "bf16":  # config 5: the queries are bf16 values too (their exact f32 images)
                q = q.to(torch.bfloat16).to(torch.float32)
            qsets.append(q.contiguous())
            ti, td = synth.brute_force_topk_native(x, q, k)  # exact truth: float32 MFMA brute force
            truths.append((ti + (lo if shard_mode else 0), td))  # (a rehearsed shard answers in local ids)
        torch.cuda.synchronize()

        # one output set per search in flight
        outs = [(torch.zeros((nq, k), dtype=torch.int64, device=dev),
                 torch.zeros((nq, k), dtype=torch.float32, device=dev),
                 torch.zeros(nq, dtype=torch.int32, device=dev)) for _ in range(depth)]
        if shard_mode:
            # search -> ONE all-gather of the packed per-shard records -> merge, enqueued without a
            # host wait (islands_amd/sharded.py); the exchange of a batch overlaps the traversals of
            # the batches submitted after it
            from islands_amd.sharded import ShardedSearcher
            # RCCL communicator of the library's own (unique id carried over the torch process group);
            # the gloo rehearsal (ranks sharing a card) exchanges through host memory instead
            searcher = ShardedSearcher(N, index=idx, device=dev, depth=depth,
                                       transport="rccl" if backend == "nccl" else "host").prepare(nq, k, ef)
            comm_info.update(searcher.shard_group.info())
            # exact global truth = merge of the per-shard exact top-k (same collective + merge)
            g_truth = []
            for (ti, td) in truths:
                gi_all = torch.zeros((world * nq, k), dtype=ti.dtype, device=dev)
                gd_all = torch.zeros((world * nq, k), dtype=td.dtype, device=dev)
                all_gather_rows(gi_all, ti.contiguous())
                all_gather_rows(gd_all, td.contiguous())
                ci = gi_all.view(world, nq, k).permute(1, 0, 2).reshape(nq, world * k)
                cd = gd_all.view(world, nq, k).permute(1, 0, 2).reshape(nq, world * k)
                sel = torch.topk(cd, k, dim=1, largest=False).indices
                g_truth.append(torch.gather(ci, 1, sel))

        def enqueue(b):
            """One step = one pass of the hot path over one resident query batch."""
            q = qsets[b % nb_batches]
            if shard_mode:
                return searcher.submit(q, k, ef)
            o = outs[b % depth]
            return idx.search_batch_device_async(q.data_ptr(), nq, d, k, ef, o[0].data_ptr(),
                                                 o[1].data_ptr(), o[2].data_ptr())

        def finish(b, token):
            if shard_mode:
                (m_ids, m_dist, m_src, m_cnt), st = searcher.result(token, with_stats=True)
                return st, (m_ids, m_cnt)
            st = idx.wait_stats(token)  # the counters of exactly this call
            o = outs[b % depth]
            return st, (o[0], o[2])

        def barrier():
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        def run(first, count, collect):
            """Runs steps [first, first+count) with `depth` searches in flight."""
            agg = {"queries": 0, "expansions": 0, "edges": 0, "evals": 0, "pushes": 0,
                   "exact_path": 0, "replayed": 0, "kernel_ms": 0.0, "allocations": 0}
            kept = []
            pending = []
            for s in range(first, first + count):
                pending.append((s, enqueue(s)))
                if len(pending) >= depth:
                    b, tok = pending.pop(0)
                    st, res = finish(b, tok)
                    if trace is not None:
                        trace.append((b, time.perf_counter(), st["kernel_ms"]))
                    for f in agg:
                        agg[f] += st[f]
                    if collect and len(kept) < nb_batches:
                        kept.append((b % nb_batches, res[0].clone(), res[1].clone()))
            while pending:
                b, tok = pending.pop(0)
                st, res = finish(b, tok)
                if trace is not None:
                    trace.append((b, time.perf_counter(), st["kernel_ms"]))
                for f in agg:
                    agg[f] += st[f]
                if collect and len(kept) < nb_batches:
                    kept.append((b % nb_batches, res[0].clone(), res[1].clone()))
            return agg, kept

        trace = None
        # The interpreter's cyclic garbage collector stays out of the timed regions: with torch imported a
        # full collection takes 35-45 ms -- measured as ONE 37-45 ms call among 600 pipelined host-buffer calls
        # (the 188th; none after), which is half of a 64-step timed region at query batch 256.  Collected
        # BEFORE the warm-up: 40 ms of host work between warm-up and timed region would let the chip drop
        # its clocks (measured: 6 % on the 20-step run).
        gc.collect()
        gc.disable()
        # A rank whose warm-up or timed steps fail must not leave the others inside a collective: the
        # library's exchange is entered by a failing rank too (poisoned record) and its waits are bounded
        # (shard.hip), so every rank gets here; what happened is then agreed on with one all-reduce of a
        # per-rank ok flag instead of assumed.
        failure = None
        agg, recalls, elapsed = None, [], float("inf")
        try:
            run(0, args.warmup, False)
            barrier()
            trace = [] if os.environ.get("ISL_BENCH_TRACE") else None  # completion times of the timed steps -> stderr
            t0 = time.perf_counter()
            agg, recalls = run(args.warmup, args.steps, True)
            barrier()
            elapsed = time.perf_counter() - t0
        except Exception as e:  # noqa: BLE001
            failure = e
        finally:
            gc.enable()
        ranks_ok = 1 if failure is None else 0
        if world > 1:
            okt = torch.tensor([ranks_ok], device=dev if backend == "nccl" else "cpu", dtype=torch.int64)
            dist.all_reduce(okt)
            ranks_ok = int(okt.item())
        if failure is not None:
            raise failure
        if ranks_ok != world:
            raise RuntimeError(f"only {ranks_ok} of {world} ranks came through the timed region")
        if trace:
            log("step completions (ms after the start of the timed region; kernel ms of the step): " +
                " ".join(f"{b}:{(t - t0) * 1e3:.2f}/{km:.2f}" for b, t, km in trace) + f"  end {elapsed * 1e3:.2f}")
        if shard_mode:
            searcher.check_flags()
        ranks_done = 1
        if world > 1:
            tt = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
            # ranks that came through every exchange of the timed region (the ok flags summed above)
            ranks_done = ranks_ok
            if shard_mode and comm_info.get("comm_ranks") not in (None, ranks_done):
                raise RuntimeError(f"communicator has {comm_info.get('comm_ranks')} ranks, {ranks_done} completed")

        ref_ids = {b: ids_b for (b, ids_b, cnt_b) in recalls} if not shard_mode else {}
        rec = []
        for (b, ids_b, cnt_b) in recalls:
            truth = g_truth[b] if shard_mode else truths[b][0]
            rec.append(synth.recall_at_k(ids_b, cnt_b, truth))
        recall = float(np.mean(rec)) if rec else 0.0
        if shard_mode:
            searcher.close()  # communicator, side stream and slots go before the next measurement

        queries_per_step = nq if (world == 1 or shard_mode) else nq * world
        value = queries_per_step * args.steps / elapsed
        kernel_ms = agg["kernel_ms"] / max(args.steps, 1)
        bytes_per_launch = algorithmic_bytes(agg, d, k, 2 if args.row_dtype == 'bf16' else 4) / max(args.steps, 1)
        achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        agg_gbs = algorithmic_bytes(agg, d, k, 2 if args.row_dtype == 'bf16' else 4) / elapsed / 1e9
        # HBM bytes per launch (rocprofv3 --pmc FETCH_SIZE) cannot be read from inside this process;
        # tools/measure_traffic.py runs this same configuration under the profiler and hands the
        # figure over in ISL_TRAFFIC_BYTES.  Without it the field is null -- never a stored constant.
        traffic_env = os.environ.get("ISL_TRAFFIC_BYTES")
        traffic_src = os.environ.get("ISL_TRAFFIC_SOURCE") if traffic_env else None

        result = {
            "metric": "queries/sec @ recall@10>=0.95, 10Mx768 ef=128",
            "value": round(value, 2),
            "unit": "queries/s",
            "n_gpus": ranks_done,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong" if (world == 1 or shard_mode) else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "recall_at_10": round(recall, 4),
            "config": {
                "workload": f"{n_local if args.rehearse_shard else N} x {d} {args.row_dtype} rows resident in HBM (in-memory provider), "
                            + ("hierarchical Gaussian mixture" if args.dataset == "G" else
                               f"dataset M: rows on a smooth {args.latent}-dimensional linear manifold + isotropic noise, "
                               "L2-normalised (tools/synth.py::make_manifold)" if args.dataset == "M" else
                               "dataset U: i.i.d. uniform [-1,1) rows (benches/hnsw_benchmarks.rs:9-14)") +
                            f", graph by {GRAPH_BUILDERS[args.graph]}, deg<= 60 (mean {gst['deg_mean']:.1f}), "
                            f"query batch {nq}, k={k}, ef={ef}, cosine",
                "nodes": N, "dim": d, "query_batch": nq, "k": k, "ef": ef,
                "graph": args.graph, "graph_build_s": round(graph_build_s, 1),
                "distinct_leaves": bool(args.distinct_leaves),
                "rehearsed_shard": ({"shard": args.rehearse_shard, "rows": [lo, hi]} if args.rehearse_shard else None),
                "parallelism": ("single" if world == 1 else
                                (f"shard{world}: node-id ranges, RCCL all-gather + top-k merge"
                                 if shard_mode else f"replica{world}")),
                "searches_in_flight": depth,
                "distinct_batches": nb_batches,
                "exchange": (dict(comm_info, ranks_completed=ranks_done) if shard_mode else None),
                "per_query": {"expansions": round(agg["expansions"] / max(agg["queries"], 1), 1),
                              "edges": round(agg["edges"] / max(agg["queries"], 1), 1),
                              "evals": round(agg["evals"] / max(agg["queries"], 1), 1)},
                "exact_path_queries": agg["exact_path"], "replayed_queries": agg["replayed"],
                "allocations_in_timed_region": agg["allocations"],
            },
            "roofline": {
                # `depth` launches of the search kernel overlap on the chip; the chip-level figure is
                # the algorithmic bytes of ALL launches of the timed region over its wall time.  The
                # per-launch figure (bytes of one launch / its own HIP-event duration, the number
                # rocprofv3 shows per dispatch) is given next to it.
                "bound": "hbm", "achieved": round(agg_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(agg_gbs / HBM_PEAK_GBS, 4),
                "traffic": float(traffic_env) if traffic_env else None,
                "traffic_source": traffic_src,
                "kernel": "leann_search_fast<2,cosine>",
                "launches_overlapped": depth,
                "per_launch": {"kernel_ms": round(kernel_ms, 3), "achieved": round(achieved, 1),
                               "frac": round(achieved / HBM_PEAK_GBS, 4)},
                "algorithmic_bytes_per_launch": round(bytes_per_launch, 0),
            },
        }
        if world == 1 and recall < 0.95 and not args.traffic_child and args.ef_sweep:
            # "if < 0.95 at ef=128 report the ef that reaches 0.95 and the QPS there" -- one batch
            # per ef, synchronous, up to the traversal kernel's limit (ef <= 512)
            sweep = []
            for ef2 in [int(v) for v in args.ef_sweep.split(",") if int(v) > ef]:
                o = outs[0]
                idx.search_batch_device(qsets[0].data_ptr(), nq, d, k, ef2, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
                st2 = idx.last_stats()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                toks = [idx.search_batch_device_async(qsets[b % nb_batches].data_ptr(), nq, d, k, ef2, outs[b][0].data_ptr(),
                                                      outs[b][1].data_ptr(), outs[b][2].data_ptr())
                        for b in range(min(depth, 8))]
                for t_ in toks:
                    idx.wait_stats(t_)
                torch.cuda.synchronize()
                dt2 = time.perf_counter() - t1
                r2 = synth.recall_at_k(outs[0][0], outs[0][2], truths[0][0])
                sweep.append({"ef": ef2, "recall_at_10": round(r2, 4), "queries_per_s": round(len(toks) * nq / dt2, 1),
                              "evals_per_query": round(st2["evals"] / nq, 1)})
                if r2 >= 0.95:
                    break
            result["ef_sweep"] = sweep
        if world == 1 and not args.no_host_path and not args.traffic_child:
            # QPS by SURVEY 8(d): host buffers in, host buffers out (the caller contract of
            # search.rs:150-181 / indexer/service.rs:781-785), `depth` calls in flight through
            # isl_search_batch_async -- H2D of the queries and D2H of the answers inside the timed
            # region, pageable numpy arrays on both sides.  Reported next to the headline, which the
            # bench contract defines on HBM-resident inputs.
            qh = [q.cpu().numpy() for q in qsets]
            houts = [(np.zeros((nq, k), np.uint64), np.zeros((nq, k), np.float32), np.zeros(nq, np.uint32))
                     for _ in range(depth)]

            def host_run(first, count):
                pend, allocs = [], 0
                for s_ in range(first, first + count):
                    pend.append(idx.search_batch_async(qh[s_ % nb_batches], k, ef, out=houts[s_ % depth]))
                    if len(pend) >= depth:
                        allocs += idx.wait_stats(pend.pop(0))["allocations"]
                while pend:
                    allocs += idx.wait_stats(pend.pop(0))["allocations"]
                return allocs

            gc.collect()
            gc.disable()
            host_run(0, min(args.warmup, 2))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            hall = host_run(0, args.steps)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            gc.enable()
            ref = ref_ids.get((args.steps - 1) % nb_batches)
            same = (bool((torch.from_numpy(houts[(args.steps - 1) % depth][0].astype(np.int64)).to(dev) == ref)
                         .all().item()) if ref is not None else None)
            # the QPS SURVEY 8(d) defines (H2D of the queries and D2H of the answers inside the timed region),
            # as a first-class field next to `value` (which this round's bench contract defines on
            # HBM-resident inputs)
            result["value_survey_8d"] = round(args.steps * nq / dt, 2)
            result["host_buffer_path"] = {
                "value": round(args.steps * nq / dt, 2), "unit": "queries/s",
                "ms_per_step": round(dt / args.steps * 1e3, 3),
                "ratio_to_device_resident": round((args.steps * nq / dt) / value, 4),
                "calls_in_flight": depth, "allocations": hall,
                "ids_equal_device_resident_path": same,
                "note": "isl_search_batch_async: pageable host queries in, pageable host results out, "
                        "PCIe both ways inside the timed region (SURVEY 8d's QPS definition)"}
        return result, (x, offsets, neighbours, entry, qsets)

    result, (x, offsets, neighbours, entry, qsets) = measure(args.mode)
    if world > 1 and args.mode == "shard" and not args.no_replica:
        # The 10M x 768 index also fits every GPU whole (30.7 GB of 288 GB): the query-parallel
        # deployment (full index per GPU, the batches split across ranks, no collective) is
        # measured next to the sharded one the north star names.
        del x, offsets, neighbours, qsets
        torch.cuda.empty_cache()
        rep, _ = measure("replica")
        result["replica_mode"] = {k_: rep[k_] for k_ in ("value", "unit", "ms_per_step", "scaling", "recall_at_10")}
        result["replica_mode"]["parallelism"] = rep["config"]["parallelism"]
        result["replica_mode"]["roofline_frac_per_gpu"] = rep["roofline"]["frac"]
        x = offsets = neighbours = entry = qsets = None
    if rank == 0 and world == 1 and not args.no_traffic and not args.traffic_child:
        # HBM bytes per launch of the dominant kernel: this same configuration once more, as a child
        # process under `rocprofv3 --pmc FETCH_SIZE` (its own pass, kernel trace only), one launch
        # in flight so that a dispatch's counter is that launch's.  Failure of any kind -> null.
        try:
            t0 = time.time()
            result["roofline"].update(measure_traffic(args))
            log(f"traffic pass under rocprofv3 took {time.time() - t0:.1f}s")
        except Exception as e:
            result["roofline"]["traffic"] = None
            result["roofline"]["traffic_source"] = f"not measured: {e!r}"[:300]
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.traffic_child:
        try:
            result["cpu_baseline"] = cpu_baseline(x, offsets, neighbours, entry, qsets, k, ef)
        except Exception as e:  # the baseline must never take the measured number down with it
            result["cpu_baseline"] = {"value": None, "unit": "queries/s", "cores": 0,
                                      "kind": "port", "sample": f"failed: {e!r}"}
    if (rank == 0 and world == 1 and not args.no_neutral_side and not args.traffic_child and args.dataset == "G"
            and args.graph == "harness" and not args.rehearse_shard and args.row_dtype == "f32"):
        # The headline's rows and graph come from one harness (a tree of clusters and a graph built from that
        # tree).  Beside it, in the same line: rows nobody designed a graph for and the graph anybody would build
        # first -- the library's own builder (isl_index_build: the reference's LeannIndex::build rule), and exact
        # nearest-neighbour lists by the library's brute force -- at 1M rows (at 10M the builds take two and five
        # minutes: profiles/r04_bench_M_product_10m.json, r04_bench_M_knn_10m.json).  Child processes; this one has
        # long finished timing.
        for key, graph in (("neutral_workload", "product"), ("neutral_workload_knn", "knn")):
            try:
                result[key] = measure_neutral(args, graph)
            except Exception as e:
                result[key] = {"value": None, "note": f"not measured: {e!r}"[:300]}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
