#!/bin/bash
# round 4, GPU call D: where the search kernel's read requests go (TCC EA counters), recompute rounds in whole tile waves (1M A/B)
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
CHILD="--traffic-child --gpus 1 --steps 4 --warmup 1 --pipeline 1 --no-cpu-baseline --no-host-path --no-traffic"
for pass in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum" "TCC_EA0_RDREQ_GMI_32B_sum TCC_EA0_RDREQ_IO_32B_sum TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RD_UNCACHED_32B_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf /tmp/pmc_$tag
  (cd /tmp && timeout -k 10 300 rocprofv3 --pmc $pass --kernel-include-regex leann_search_fast -d /tmp/pmc_$tag -o p --output-format csv -- python3 $R/bench.py $CHILD > $R/gpurun_out/r04_pmc_$tag.json 2> $R/gpurun_out/r04_pmc_$tag.err) || { echo "pmc pass $tag failed"; tail -5 gpurun_out/r04_pmc_$tag.err; exit 1; }
  f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
  cp $f gpurun_out/r04_pmc_${tag}_counter_collection.csv
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "leann_search_fast" in r["Kernel_Name"] and int(r["Grid_Size"]) > 64:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, len(v), "dispatches, mean of all but the first:", sum(v[1:]) / max(1, len(v) - 1))
PY
done
echo "== recompute 1M quantum A/B"
timeout -k 10 600 python tools/recompute_bench.py --nodes 1000000 --nq 256 --ef 128 --two-level 0.3 --pq-m 192 --quantum-ab > gpurun_out/r04_recompute_1m_quantum_ab.jsonl 2> gpurun_out/r04_recompute_1m_quantum_ab.err || { tail -20 gpurun_out/r04_recompute_1m_quantum_ab.err; exit 1; }
python3 - <<'PY'
import json
for l in open("gpurun_out/r04_recompute_1m_quantum_ab.jsonl"):
    if l.startswith("{"):
        d = json.loads(l)
        print({k: d.get(k) for k in ("label", "queries_per_s", "wall_s", "rounds", "encoded_nodes", "encoder_tflops_over_call", "recall_at_10", "equals_in_memory_provider") if k in d} or list(d)[:12])
PY
