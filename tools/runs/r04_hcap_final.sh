#!/bin/bash
# round 4: the visited-table hint with its cap at four times the default -- parity file, then dataset M with the library's builder at 1M and 10M
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_build.py tests/test_gpu_hnsw.py -m gpu -x -q > gpurun_out/r04_hcap_final_tests.log 2>&1 || { tail -30 gpurun_out/r04_hcap_final_tests.log; exit 1; }
tail -2 gpurun_out/r04_hcap_final_tests.log
for n in 1000000 10000000; do
  timeout -k 10 900 python bench.py --dataset M --graph product --nodes $n --steps 20 --warmup 5 --no-traffic > gpurun_out/r04_bench_M_product_$n.json 2> gpurun_out/r04_bench_M_product_$n.err || { tail -20 gpurun_out/r04_bench_M_product_$n.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/r04_bench_M_product_$n.json').read().strip().splitlines()[-1])
print('M product $n', d['value'], d['recall_at_10'], d['roofline']['frac'], d['config']['per_query'], d['config']['graph_build_s'], d.get('value_survey_8d'), d['cpu_baseline']['value'])"
done
