#!/bin/bash
# round 4, GPU call Q: float32 GEMM on four waves in the encoder -- tests, encoder end to end, config 3 at 1M, each with and without (same box)
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r04_q_tests.log 2>&1 || { tail -30 gpurun_out/r04_q_tests.log; exit 1; }
tail -2 gpurun_out/r04_q_tests.log
for w in 0 1 0 1; do
  ISL_GEMM_F32_W4=$w timeout -k 10 300 python tools/encoder_perf.py 8192 64 2>/dev/null | tail -1 | sed "s/^/four-wave tile $w: /"
done | tee gpurun_out/r04_encoder_perf_w4.log
for w in 0 1; do
  ISL_GEMM_F32_W4=$w timeout -k 10 600 python tools/recompute_bench.py --nodes 1000000 --nq 256 --ef 128 --two-level 0.05 --pq-m 192 --check-in-memory > gpurun_out/r04_recompute_1m_w4_$w.jsonl 2> gpurun_out/r04_recompute_1m_w4_$w.err || { tail -20 gpurun_out/r04_recompute_1m_w4_$w.err; exit 1; }
  python3 - $w <<'PY'
import json, sys
for l in open(f"gpurun_out/r04_recompute_1m_w4_{sys.argv[1]}.jsonl"):
    if l.startswith("{"):
        d = json.loads(l)
        print("four-wave tile", sys.argv[1], {k: d.get(k) for k in ("run", "value", "seconds", "rounds", "encoded_nodes", "recall_at_10", "equals_in_memory_provider", "encode_all_tflops")}, d.get("roofline", {}).get("frac"))
PY
done
