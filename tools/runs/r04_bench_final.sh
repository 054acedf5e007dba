#!/bin/bash
# round 4: the driver's bench command on the round's final library and bench.py (both neutral side fields)
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err || { tail -20 gpurun_out/r04_bench.err; exit 1; }
python3 -c "
import json; d=json.loads(open('gpurun_out/r04_bench.json').read().strip().splitlines()[-1])
print('bench', d['value'], d['recall_at_10'], d['roofline']['frac'], d['roofline'].get('traffic_over_algorithmic'), d.get('value_survey_8d'), d['cpu_baseline']['value'])
for k in ('neutral_workload', 'neutral_workload_knn'): print(k, {f: d[k].get(f) for f in ('value', 'recall_at_10', 'roofline_frac', 'graph_build_s', 'child_run_s')})"
