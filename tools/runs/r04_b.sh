#!/bin/bash
# round 4, GPU call B: GEMM epilogue changes (bf16 distance GEMM micro-benchmark, encoder end to end), then the GPU suite
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 tools/microbench/gemm_bf16_ph8 4096 65536 4096 30 > gpurun_out/r04_gemm_bf16_ph8_var12.log 2>&1 || { echo "microbench failed"; tail -5 gpurun_out/r04_gemm_bf16_ph8_var12.log; exit 1; }
grep -n "VAR 12\|round 0\|round 3\|tiny\|LDS-transposed epilogue: dot" gpurun_out/r04_gemm_bf16_ph8_var12.log | tail -12
timeout -k 10 300 python tools/encoder_perf.py 8192 64 > gpurun_out/r04_encoder_perf.log 2>&1 || { echo "encoder_perf failed"; tail -5 gpurun_out/r04_encoder_perf.log; exit 1; }
tail -2 gpurun_out/r04_encoder_perf.log
timeout -k 10 300 python tools/distance_gemm_perf.py > gpurun_out/r04_distance_gemm_perf.jsonl 2>&1 || { echo "distance_gemm_perf failed"; tail -5 gpurun_out/r04_distance_gemm_perf.jsonl; exit 1; }
cat gpurun_out/r04_distance_gemm_perf.jsonl | cut -c1-400
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputests.log 2>&1; rc=$?
tail -4 gpurun_out/r04_gputests.log
exit $rc
