"""`python bench.py --gpus N` with no launcher must start N ranks itself (the driver's N = 1 form of
the command, round 2: --gpus was parsed and never used).  CPU part: the launch plumbing in --dry-run
mode (rendezvous on 127.0.0.1, one all-gather of a packed record over gloo, the JSON line passed
through by the parent, which never touches the GPU).  GPU part: the real two-rank run on one card
(ISL_BENCH_BACKEND=gloo rehearsal: shard search, exchange and merge inside libislands_amd.so).
(80000 nodes, not fewer: the harness's graph over a 10000-row shard of 20 half-clusters reaches a
local recall of 0.56 -- a property of tools/synth.py at that size, measured per shard with no
exchange involved -- while 40000-row shards reach 1.0.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None, timeout=900):
    env = dict(os.environ, **(env_extra or {}))
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, cwd=ROOT,
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    lines = [l for l in pr.stdout.decode().splitlines() if l.startswith("{")]
    assert pr.returncode == 0, pr.stderr.decode()[-2000:]
    assert len(lines) == 1, (lines, pr.stderr.decode()[-1000:])
    return json.loads(lines[0])


@pytest.mark.timeout(600)
def test_gpus_flag_launches_the_ranks_itself_dry_run():
    res = _run(["--gpus", "2", "--dry-run", "--nodes", "20000", "--nq", "64"])
    assert res["n_gpus"] == 2 and res["dry_run"] is True
    assert res["config"]["shard_ranges"] == [[0, 10000], [10000, 20000]]
    one = _run(["--gpus", "1", "--dry-run", "--nodes", "20000", "--nq", "64"])
    assert one["n_gpus"] == 1


@pytest.mark.gpu
@pytest.mark.timeout(1200)
def test_two_ranks_on_one_card_without_a_launcher():
    res = _run(["--gpus", "2", "--nodes", "80000", "--nq", "64", "--steps", "4", "--warmup", "2", "--pipeline", "3",
                "--no-cpu-baseline", "--no-replica", "--no-traffic"], {"ISL_BENCH_BACKEND": "gloo"})
    assert res["n_gpus"] == 2
    ex = res["config"]["exchange"]
    assert ex["world"] == 2 and ex["comm_ranks"] == 2 and ex["ranks_completed"] == 2 and ex["rccl"] is False
    assert res["config"]["distinct_batches"] == 6
    assert res["recall_at_10"] > 0.9 and res["value"] > 0
    assert res["config"]["allocations_in_timed_region"] == 0
