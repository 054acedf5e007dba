#!/bin/bash
# round 4, GPU call J: BASELINE config 5 at full size (10M x 4096 bf16, nq 4096): GEMM leg enqueued block by block, traversal, two-level at the r03 operating points
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1170 python tools/config5_bench.py --tl-ratios 0.3,0.85 --tl-efs 128 > gpurun_out/r04_bench_config5_10m.json 2> gpurun_out/r04_bench_config5_10m.err
rc=$?
tail -14 gpurun_out/r04_bench_config5_10m.err | cut -c1-300
exit $rc
