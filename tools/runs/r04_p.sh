#!/bin/bash
# round 4, GPU call P: float32 GEMM, 256 x 256 tile on four waves (128 x 128 per wave) against eight, the encoder's four shapes
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r04_gemm_f32_w4.log
for shape in "524288 2304 768" "524288 768 768" "524288 3072 768" "524288 768 3072"; do
  timeout -k 10 200 tools/microbench/gemm_f32_w4 $shape >> gpurun_out/r04_gemm_f32_w4.log 2>&1 || { echo "failed on $shape"; tail -5 gpurun_out/r04_gemm_f32_w4.log; exit 1; }
done
cat gpurun_out/r04_gemm_f32_w4.log
