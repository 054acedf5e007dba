#!/bin/bash
# round 4, GPU call E: BASELINE config 3 at full size (10M nodes x 64 tokens, nq 1024), two-level search over the recompute
# provider at the r03 operating point (PQ m = 192, ef 256, ratio 0.05), encoder batches "every miss at once" against "whole tile waves"
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1150 python tools/recompute_bench.py --nodes 10000000 --nq 1024 --ef 256 --two-level 0.05 --pq-m 192 --quantum-ab > gpurun_out/r04_recompute_10m_quantum_ab.jsonl 2> gpurun_out/r04_recompute_10m_quantum_ab.err
rc=$?
tail -5 gpurun_out/r04_recompute_10m_quantum_ab.err
python3 - <<'PY'
import json
for l in open("gpurun_out/r04_recompute_10m_quantum_ab.jsonl"):
    if l.startswith("{"):
        d = json.loads(l)
        print({k: d.get(k) for k in ("run", "value", "seconds", "rounds", "encoded_nodes", "recall_at_10", "equals_in_memory_provider")}, d.get("roofline"))
PY
exit $rc
