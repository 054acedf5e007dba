#!/bin/bash
# round 4, GPU call S: the chain's addends by packed multiplies (v_pk_mul_f32) -- whole GPU suite, then A/B against the build before it
# (gpurun_ab/libislands_amd_scalar_terms.so), same box, alternating: headline 20 steps and 300 steps, dataset M 1M, config 5 traversal
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputests.log 2>&1 || { tail -30 gpurun_out/r04_gputests.log; exit 1; }
tail -2 gpurun_out/r04_gputests.log
OLD=$PWD/gpurun_ab/libislands_amd_scalar_terms.so
run() { # label, lib ('' = the tree's), args...
  local label=$1 lib=$2; shift 2
  if [ -n "$lib" ]; then export ISL_AMD_LIB=$lib; else unset ISL_AMD_LIB; fi
  timeout -k 10 400 python bench.py "$@" --no-traffic --no-cpu-baseline --no-host-path --no-neutral-side 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', d['value'], d['roofline']['frac'], d['recall_at_10'])"
}
for rep in 1 2; do
  run "G 20 steps, scalar" $OLD --steps 20 --warmup 5
  run "G 20 steps, packed" "" --steps 20 --warmup 5
done
run "G 300 steps, scalar" $OLD --steps 300 --warmup 16 --distinct-batches 32
run "G 300 steps, packed" "" --steps 300 --warmup 16 --distinct-batches 32
run "M 1M knn, scalar" $OLD --dataset M --graph knn --nodes 1000000 --steps 20 --warmup 5
run "M 1M knn, packed" "" --dataset M --graph knn --nodes 1000000 --steps 20 --warmup 5
run "bf16 rows 1M, scalar" $OLD --row-dtype bf16 --nodes 1000000 --steps 40 --warmup 8
run "bf16 rows 1M, packed" "" --row-dtype bf16 --nodes 1000000 --steps 40 --warmup 8
unset ISL_AMD_LIB
