#!/bin/bash
# round 4, GPU call Z: dataset M on the graph the library's own builder makes (isl_index_build = the reference's LeannIndex::build rule), 1M rows
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1150 python bench.py --dataset M --graph product --steps 20 --warmup 5 --no-traffic > gpurun_out/r04_bench_M_product_10m.json 2> gpurun_out/r04_bench_M_product_10m.err || { tail -20 gpurun_out/r04_bench_M_product_10m.err; exit 1; }
python3 -c "
import json; d=json.loads(open('gpurun_out/r04_bench_M_product_10m.json').read().strip().splitlines()[-1])
print('M product 10M', d['value'], d['recall_at_10'], d['roofline']['frac'], d['config']['per_query'], d['config']['graph_build_s'], d.get('ef_sweep'), d['cpu_baseline']['value'])"
