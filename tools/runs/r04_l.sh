#!/bin/bash
# round 4, GPU call L: visited table of any capacity (hslot_cap) -- the whole GPU suite, then table sizes on dataset M at 1M
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputests.log 2>&1 || { tail -30 gpurun_out/r04_gputests.log; exit 1; }
tail -2 gpurun_out/r04_gputests.log
for hc in 0 3392 3584 4032 4096; do
  if [ $hc = 0 ]; then unset ISL_HCAP; else export ISL_HCAP=$hc; fi
  timeout -k 10 300 python bench.py --dataset M --graph knn --nodes 1000000 --steps 20 --warmup 5 --no-traffic --no-cpu-baseline --no-host-path > gpurun_out/r04_bench_M_knn_1m_hcap$hc.json 2> gpurun_out/r04_bench_M_knn_1m_hcap$hc.err || { tail -20 gpurun_out/r04_bench_M_knn_1m_hcap$hc.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/r04_bench_M_knn_1m_hcap$hc.json').read().strip().splitlines()[-1])
print('hcap $hc (0 = by the hint)', d['value'], d['recall_at_10'], d['roofline']['frac'], d['config']['per_query']['evals'])"
done
unset ISL_HCAP
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-traffic --no-cpu-baseline --no-host-path > gpurun_out/r04_bench_G_after_hcap.json 2> gpurun_out/r04_bench_G_after_hcap.err || { tail -20 gpurun_out/r04_bench_G_after_hcap.err; exit 1; }
python3 -c "
import json; d=json.loads(open('gpurun_out/r04_bench_G_after_hcap.json').read().strip().splitlines()[-1])
print('G harness 10M', d['value'], d['recall_at_10'], d['roofline']['frac'], d['config']['allocations_in_timed_region'])"
