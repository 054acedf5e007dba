#!/bin/bash
# round 4, GPU call G: dataset M (no cluster tree) on the exact-kNN graph and on the harness graph, 1M rows, then 10M on the kNN graph
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
for g in knn harness; do
  timeout -k 10 400 python bench.py --dataset M --graph $g --nodes 1000000 --steps 20 --warmup 5 --no-traffic > gpurun_out/r04_bench_M_${g}_1m.json 2> gpurun_out/r04_bench_M_${g}_1m.err || { tail -20 gpurun_out/r04_bench_M_${g}_1m.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/r04_bench_M_${g}_1m.json').read().strip().splitlines()[-1])
print('$g 1M', d['value'], d['recall_at_10'], d['roofline']['frac'], d['config']['per_query'], d.get('ef_sweep'))"
done
timeout -k 10 1000 python bench.py --dataset M --graph knn --steps 20 --warmup 5 --no-traffic > gpurun_out/r04_bench_M_knn_10m.json 2> gpurun_out/r04_bench_M_knn_10m.err || { tail -20 gpurun_out/r04_bench_M_knn_10m.err; exit 1; }
python3 -c "
import json; d=json.loads(open('gpurun_out/r04_bench_M_knn_10m.json').read().strip().splitlines()[-1])
print('knn 10M', d['value'], d['recall_at_10'], d['roofline']['frac'], d['config']['per_query'], d.get('ef_sweep'))"
