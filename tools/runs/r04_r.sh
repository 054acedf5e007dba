#!/bin/bash
# round 4, GPU call R: PMC pass of the bf16 distance GEMM (cosine epilogue, config-5 block shape) + kernel trace of the same command
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
rm -rf /tmp/pmc_g /tmp/kt_g
(cd /tmp && timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-include-regex "gemm_tn_bf16" -d /tmp/pmc_g -o p --output-format csv -- python3 $R/tools/distance_gemm_perf.py 65536 > $R/gpurun_out/r04_pmc_gemm.out 2> $R/gpurun_out/r04_pmc_gemm.err) || { tail -5 gpurun_out/r04_pmc_gemm.err; exit 1; }
python3 tools/pmc_sum.py /tmp/pmc_g | tee gpurun_out/r04_pmc_gemm_sums.txt
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_g -o p -- python3 $R/tools/distance_gemm_perf.py 65536 > $R/gpurun_out/r04_kt_gemm.out 2> $R/gpurun_out/r04_kt_gemm.err) || { tail -5 gpurun_out/r04_kt_gemm.err; exit 1; }
f=$(find /tmp/kt_g -name "*kernel_stats.csv" | head -1); grep -i "gemm_tn\|sumsq\|Name" $f | cut -c1-260 | tee gpurun_out/r04_kt_gemm_stats.csv
cat gpurun_out/r04_kt_gemm.out | cut -c1-300
