#!/bin/bash
# round 4, GPU call I: two-level search over the recompute provider naming nodes ahead (tests, then config 3 at 1M, A/B in one process)
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_two_level.py -m gpu -x -q > gpurun_out/r04_i_tests.log 2>&1 || { tail -30 gpurun_out/r04_i_tests.log; exit 1; }
tail -2 gpurun_out/r04_i_tests.log
timeout -k 10 900 python tools/recompute_bench.py --nodes 1000000 --nq 256 --ef 128 --two-level 0.05 --pq-m 192 --check-in-memory --prefetch-ab 0,1,2,4,8 > gpurun_out/r04_recompute_1m_prefetch_ab.jsonl 2> gpurun_out/r04_recompute_1m_prefetch_ab.err || { tail -20 gpurun_out/r04_recompute_1m_prefetch_ab.err; exit 1; }
python3 - <<'PY'
import json
for l in open("gpurun_out/r04_recompute_1m_prefetch_ab.jsonl"):
    if l.startswith("{"):
        d = json.loads(l)
        print({k: d.get(k) for k in ("run", "value", "seconds", "rounds", "encoded_nodes", "recall_at_10", "equals_in_memory_provider")}, d.get("roofline", {}).get("frac"))
PY
echo "== visited table size on dataset M (V ~ 3100 per query), 1M rows, kNN graph"
for hb in 11 12 13; do
  ISL_HBITS=$hb timeout -k 10 300 python bench.py --dataset M --graph knn --nodes 1000000 --steps 20 --warmup 5 --no-traffic --no-cpu-baseline --no-host-path > gpurun_out/r04_bench_M_knn_1m_hbits$hb.json 2> gpurun_out/r04_bench_M_knn_1m_hbits$hb.err || { tail -20 gpurun_out/r04_bench_M_knn_1m_hbits$hb.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/r04_bench_M_knn_1m_hbits$hb.json').read().strip().splitlines()[-1])
print('hbits $hb', d['value'], d['recall_at_10'], d['roofline']['frac'], d['config']['per_query'])"
done
