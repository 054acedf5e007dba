#!/bin/bash
# round 4, GPU call T: size of the HBM overflow visited table (32768 entries per wave slot = 128 KiB, against 8192 and 4096: small enough to
# stay in L2 for the few queries that need it) -- headline 20 steps, alternating, same box; and dataset M with the visited-table hint off
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { # label, env assignments..., -- args
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 400 python bench.py "$@" --no-traffic --no-cpu-baseline --no-host-path --no-neutral-side 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', d['value'], d['roofline']['frac'], d['recall_at_10'], d['config']['exact_path_queries'])"
}
for rep in 1 2 3; do
  for b in 15 13 12; do run "G 20 steps, overflow table 2^$b" ISL_OVF_BITS=$b -- --steps 20 --warmup 5; done
done
for b in 15 12; do run "G 300 steps, overflow table 2^$b" ISL_OVF_BITS=$b -- --steps 300 --warmup 16 --distinct-batches 32; done
for b in 15 13 12; do run "M 1M knn, table by ef alone, overflow table 2^$b" ISL_OVF_BITS=$b ISL_NO_VISITED_HINT=1 -- --dataset M --graph knn --nodes 1000000 --steps 20 --warmup 5; done
