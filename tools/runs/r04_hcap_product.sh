#!/bin/bash
# round 4: visited-table sizes on the densest graph of the table (dataset M, 1M rows, isl_index_build: 4895 evaluations per query)
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
for hc in 5696 6144 7104; do
  if [ $hc = 0 ]; then unset ISL_HCAP; else export ISL_HCAP=$hc; fi
  timeout -k 10 300 python bench.py --dataset M --graph product --nodes 1000000 --steps 20 --warmup 5 --no-traffic --no-cpu-baseline --no-host-path > gpurun_out/r04_bench_M_product_1m_hcap$hc.json 2> gpurun_out/r04_bench_M_product_1m_hcap$hc.err || { tail -20 gpurun_out/r04_bench_M_product_1m_hcap$hc.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/r04_bench_M_product_1m_hcap$hc.json').read().strip().splitlines()[-1])
print('hcap $hc (0 = by the hint)', d['value'], d['recall_at_10'], d['roofline']['frac'], d['config']['per_query']['evals'])"
done
