#!/bin/bash
# round 4, GPU call U: SQ counters of the search kernel, one launch of 1024 queries at a time: headline rows (10M, harness graph) and manifold rows (1M, kNN graph)
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
CH="--traffic-child --gpus 1 --steps 4 --warmup 2 --pipeline 1 --no-cpu-baseline --no-host-path --no-traffic --no-neutral-side"
i=0
for wl in "" "--dataset M --graph knn --nodes 1000000"; do
  i=$((i+1))
  for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
    tag=w${i}_$(echo $pass | cut -d' ' -f1)
    rm -rf /tmp/sq_$tag
    (cd /tmp && timeout -k 10 400 rocprofv3 --pmc $pass --kernel-include-regex leann_search_fast -d /tmp/sq_$tag -o p --output-format csv -- python3 $R/bench.py $CH $wl > $R/gpurun_out/r04_sq_$tag.json 2> $R/gpurun_out/r04_sq_$tag.err) || { echo "pass $tag failed"; tail -5 gpurun_out/r04_sq_$tag.err; continue; }
    f=$(find /tmp/sq_$tag -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$tag" <<'PY'
import csv, sys, collections, json
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "leann_search_fast" in r["Kernel_Name"] and int(r["Grid_Size"]) > 64:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: sum(v[1:]) / max(1, len(v) - 1) for k, v in acc.items()}
out["dispatches_averaged"] = max(0, len(next(iter(acc.values()))) - 1) if acc else 0
print(sys.argv[2], json.dumps(out))
PY
  done
done | tee gpurun_out/r04_pmc_sq_lines.txt
