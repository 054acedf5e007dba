#!/bin/bash
# round 4, GPU call C: new distance-matrix forms, oldest-first wave priority probe, the driver's bench + kernel trace,
# counter list, steady state with and without cross-query row reuse
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
step() { echo "== $1"; }
step "tests"; timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "distance_matrix" > gpurun_out/r04_c_tests.log 2>&1 || { tail -30 gpurun_out/r04_c_tests.log; exit 1; }
tail -2 gpurun_out/r04_c_tests.log
step "age-prio probe"; timeout -k 10 400 python tools/single_launch_probe.py --age-prio 0,64,100,150 > gpurun_out/r04_age_prio_probe.json 2> gpurun_out/r04_age_prio_probe.err || { tail -20 gpurun_out/r04_age_prio_probe.err; exit 1; }
grep "rep 2" gpurun_out/r04_age_prio_probe.err
step "bench"; timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err || { tail -20 gpurun_out/r04_bench.err; exit 1; }
cut -c1-300 gpurun_out/r04_bench.json
step "counter list"; (cd /tmp && rocprofv3 -L > $OLDPWD/gpurun_out/r04_counters_list.txt 2>&1) || true
grep -c . gpurun_out/r04_counters_list.txt
step "kernel trace"; rm -rf /tmp/kt; (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o p -- python3 $OLDPWD/bench.py --steps 20 --warmup 5 --no-traffic --no-cpu-baseline --no-host-path > $OLDPWD/gpurun_out/r04_bench_under_rocprofv3.json 2> $OLDPWD/gpurun_out/r04_kt.err) || { tail -20 gpurun_out/r04_kt.err; exit 1; }
python tools/kernel_stats_timed.py /tmp/kt --steps 20 --out gpurun_out/r04_kernel_stats_timed.csv && cat gpurun_out/r04_kernel_stats_timed.csv | cut -c1-200
find /tmp/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r04_kernel_stats_whole_process.csv
step "steady state"; timeout -k 10 300 python bench.py --steps 1000 --warmup 16 --distinct-batches 32 --no-traffic --no-cpu-baseline --no-host-path > gpurun_out/r04_bench_1000_steps.json 2> gpurun_out/r04_1000.err || { tail -20 gpurun_out/r04_1000.err; exit 1; }
cut -c1-200 gpurun_out/r04_bench_1000_steps.json
timeout -k 10 300 python bench.py --steps 1000 --warmup 16 --distinct-batches 32 --distinct-leaves --no-traffic --no-cpu-baseline --no-host-path > gpurun_out/r04_bench_1000_steps_distinct_leaves.json 2> gpurun_out/r04_1000dl.err || { tail -20 gpurun_out/r04_1000dl.err; exit 1; }
cut -c1-200 gpurun_out/r04_bench_1000_steps_distinct_leaves.json
