#!/bin/bash
# round 4, GPU call V: concurrent asynchronous recompute calls answered together -- tests, then config 3 at 1M with 8 calls of 256 in flight,
# with and without the coalescing
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_two_level.py tests/test_gpu_lanes.py -m gpu -x -q > gpurun_out/r04_v_tests.log 2>&1 || { tail -40 gpurun_out/r04_v_tests.log; exit 1; }
tail -2 gpurun_out/r04_v_tests.log
for off in 1 0; do
  if [ $off = 1 ]; then export ISL_NO_RECOMPUTE_COALESCE=1; else unset ISL_NO_RECOMPUTE_COALESCE; fi
  timeout -k 10 600 python tools/recompute_bench.py --nodes 1000000 --nq 2048 --ef 128 --two-level 0.05 --pq-m 192 --check-in-memory --inflight 8 > gpurun_out/r04_recompute_1m_inflight8_off$off.jsonl 2> gpurun_out/r04_recompute_1m_inflight8_off$off.err || { tail -20 gpurun_out/r04_recompute_1m_inflight8_off$off.err; exit 1; }
  python3 - $off <<'PY'
import json, sys
for l in open(f"gpurun_out/r04_recompute_1m_inflight8_off{sys.argv[1]}.jsonl"):
    if l.startswith("{"):
        d = json.loads(l)
        print("coalescing off" if sys.argv[1] == "1" else "coalescing on", {k: d.get(k) for k in ("run", "value", "seconds", "recall_at_10", "calls_answered_together", "encoded_nodes_per_query", "equals_in_memory_provider")}, d.get("roofline", {}).get("frac"))
PY
done
