#!/bin/bash
# round 4: the fuzz tool on the round's library, every mode (seed 404)
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r04_fuzz.log
for mode in recompute leann two_level hnsw build; do
  timeout -k 10 260 python tests/fuzz_parity.py --mode $mode --seconds 170 --seed 404 >> gpurun_out/r04_fuzz.log 2>&1 || { echo "fuzz $mode FAILED"; tail -20 gpurun_out/r04_fuzz.log; exit 1; }
  tail -1 gpurun_out/r04_fuzz.log
done
