#!/bin/bash
# round 4, GPU call H: BASELINE config 3 at full size, operating point from the (finer) in-memory sweep, plain search beside it, warm cache
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 120 tools/microbench/gemm_f32_w4 300000 2304 768 > gpurun_out/r04_gemm_f32_w4_bits.log 2>&1; grep differing gpurun_out/r04_gemm_f32_w4_bits.log
timeout -k 10 1100 python tools/recompute_bench.py --nodes 10000000 --nq 1024 --ef-list 128,256 --two-level-auto --pq-m 192 --also-plain --warm > gpurun_out/r04_recompute_10m_two_level.jsonl 2> gpurun_out/r04_recompute_10m_two_level.err
rc=$?
tail -12 gpurun_out/r04_recompute_10m_two_level.err | cut -c1-300
exit $rc
