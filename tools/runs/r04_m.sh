#!/bin/bash
# round 4, GPU call M: the driver's bench on the round's final library (with the neutral side measurement), its kernel trace, and dataset M at 10M
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err || { tail -20 gpurun_out/r04_bench.err; exit 1; }
python3 -c "
import json; d=json.loads(open('gpurun_out/r04_bench.json').read().strip().splitlines()[-1])
print('bench', d['value'], d['recall_at_10'], d['roofline']['frac'], d['roofline'].get('traffic_over_algorithmic'), d.get('value_survey_8d'), d['cpu_baseline']['value'], d.get('neutral_workload'))"
rm -rf /tmp/kt; (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o p -- python3 $R/bench.py --steps 20 --warmup 5 --no-traffic --no-cpu-baseline --no-host-path --no-neutral-side > $R/gpurun_out/r04_bench_under_rocprofv3.json 2> $R/gpurun_out/r04_kt.err) || { tail -20 gpurun_out/r04_kt.err; exit 1; }
python tools/kernel_stats_timed.py /tmp/kt --steps 20 --out gpurun_out/r04_kernel_stats_timed.csv && cut -c1-200 gpurun_out/r04_kernel_stats_timed.csv
find /tmp/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r04_kernel_stats_whole_process.csv
timeout -k 10 1000 python bench.py --dataset M --graph knn --steps 20 --warmup 5 > gpurun_out/r04_bench_M_knn_10m.json 2> gpurun_out/r04_bench_M_knn_10m.err || { tail -20 gpurun_out/r04_bench_M_knn_10m.err; exit 1; }
python3 -c "
import json; d=json.loads(open('gpurun_out/r04_bench_M_knn_10m.json').read().strip().splitlines()[-1])
print('M knn 10M', d['value'], d['recall_at_10'], d['roofline']['frac'], d['roofline'].get('traffic_over_algorithmic'), d['config']['per_query'], d.get('value_survey_8d'), d['cpu_baseline']['value'], d['cpu_baseline']['value_1thread'])"
