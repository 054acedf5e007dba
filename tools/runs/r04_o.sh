#!/bin/bash
# round 4, GPU call O: config 3 with eight 1024-query batches answered as ONE call of 8192 queries (what a coalescing caller gets):
# same operating point (PQ m = 192, ef 224, ratio 0.05)
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1170 python tools/recompute_bench.py --nodes 10000000 --nq 8192 --ef 224 --two-level 0.05 --pq-m 192 --check-in-memory > gpurun_out/r04_recompute_10m_nq8192.jsonl 2> gpurun_out/r04_recompute_10m_nq8192.err
rc=$?
tail -4 gpurun_out/r04_recompute_10m_nq8192.err | cut -c1-300
python3 - <<'PY'
import json
for l in open("gpurun_out/r04_recompute_10m_nq8192.jsonl"):
    if l.startswith("{"):
        d = json.loads(l)
        print({k: d.get(k) for k in ("run", "value", "seconds", "rounds", "encoded_nodes", "encoded_nodes_per_query", "recall_at_10", "equals_in_memory_provider")}, d.get("roofline", {}).get("frac"))
PY
exit $rc
