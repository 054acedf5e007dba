#!/bin/bash
# round 4, GPU call X: host-buffer asynchronous recompute calls through the coalescing; whole GPU suite on the result
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputests.log 2>&1 || { tail -40 gpurun_out/r04_gputests.log; exit 1; }
tail -2 gpurun_out/r04_gputests.log
