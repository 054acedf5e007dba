#!/bin/bash
# round 4, last GPU call: the whole GPU suite and the smoke entry on the round's final library, dataset M's steady state
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputests.log 2>&1 || { tail -30 gpurun_out/r04_gputests.log; exit 1; }
tail -2 gpurun_out/r04_gputests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 400 python bench.py --dataset M --graph knn --nodes 1000000 --steps 600 --warmup 16 --distinct-batches 32 --no-traffic --no-cpu-baseline --no-host-path > gpurun_out/r04_bench_M_knn_1m_600_steps.json 2>/dev/null
python3 -c "
import json; d=json.loads(open('gpurun_out/r04_bench_M_knn_1m_600_steps.json').read().strip().splitlines()[-1])
print('M knn 1M, 600 steps', d['value'], d['recall_at_10'], d['roofline']['frac'])"
