#!/bin/bash
# round 4, GPU call K: visited table sized by the index's evaluations per query -- parity file, dataset M at 1M, headline unchanged
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lanes.py tests/test_gpu_scale.py -m gpu -x -q > gpurun_out/r04_k_tests.log 2>&1 || { tail -30 gpurun_out/r04_k_tests.log; exit 1; }
tail -2 gpurun_out/r04_k_tests.log
timeout -k 10 300 python bench.py --dataset M --graph knn --nodes 1000000 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04_bench_M_knn_1m_hint.json 2> gpurun_out/r04_bench_M_knn_1m_hint.err || { tail -20 gpurun_out/r04_bench_M_knn_1m_hint.err; exit 1; }
python3 -c "
import json; d=json.loads(open('gpurun_out/r04_bench_M_knn_1m_hint.json').read().strip().splitlines()[-1])
print('M knn 1M with the hint', d['value'], d['recall_at_10'], d['roofline'], d['config']['per_query'], d.get('value_survey_8d'))"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-traffic --no-cpu-baseline > gpurun_out/r04_bench_G_after_hint.json 2> gpurun_out/r04_bench_G_after_hint.err || { tail -20 gpurun_out/r04_bench_G_after_hint.err; exit 1; }
python3 -c "
import json; d=json.loads(open('gpurun_out/r04_bench_G_after_hint.json').read().strip().splitlines()[-1])
print('G harness 10M', d['value'], d['recall_at_10'], d['roofline']['frac'], d['config']['per_query'], d['config']['allocations_in_timed_region'])"
