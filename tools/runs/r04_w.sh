#!/bin/bash
# round 4, GPU call W: concurrent recompute calls answered together -- the failing-member test, then BASELINE config 3 at full size with
# eight asynchronous 1024-query calls in flight (same operating point: PQ m = 192, ef 224, ratio 0.05)
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py -m gpu -x -q > gpurun_out/r04_w_tests.log 2>&1 || { tail -40 gpurun_out/r04_w_tests.log; exit 1; }
tail -2 gpurun_out/r04_w_tests.log
timeout -k 10 1100 python tools/recompute_bench.py --nodes 10000000 --nq 8192 --ef 224 --two-level 0.05 --pq-m 192 --check-in-memory --inflight 8 > gpurun_out/r04_recompute_10m_inflight8.jsonl 2> gpurun_out/r04_recompute_10m_inflight8.err
rc=$?
tail -3 gpurun_out/r04_recompute_10m_inflight8.err | cut -c1-300
python3 - <<'PY'
import json
for l in open("gpurun_out/r04_recompute_10m_inflight8.jsonl"):
    if l.startswith("{"):
        d = json.loads(l)
        print({k: d.get(k) for k in ("run", "value", "seconds", "recall_at_10", "calls_answered_together", "encoded_nodes_per_query", "equals_in_memory_provider")}, d.get("roofline", {}).get("frac"))
PY
exit $rc
