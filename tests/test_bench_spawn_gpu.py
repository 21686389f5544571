"""`python bench.py --gpus N` with no rendezvous variables starts its own ranks (torch.distributed.run, one process per
GPU) -- the command line the driver uses for the scaling runs.  On a one-GPU box the same road is taken with one rank
(SPANGPU_BENCH_SPAWN=1) and the RCCL gathers forced on (SPANGPU_BENCH_FORCE_GATHER=1): the DTMF bank with its digit
gather, and run_echo() -- BASELINE configs[4]'s shard with the ERLE gather -- end to end."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["SPANGPU_BENCH_SPAWN"] = "1"
    env["SPANGPU_BENCH_FORCE_GATHER"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + extra, env=env, cwd="/tmp",
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks_dtmf(built):
    r = run_bench(["--steps", "50", "--warmup", "10", "--channels", "16384", "--no-cpu-baseline", "--no-e2e", "--no-paths",
                   "--min-timed-ms", "5"])
    assert r["n_gpus"] == 1 and r["value"] > 0 and r["unit"] == "Msamples/s"
    assert r["roofline"]["frac"] > 0


def test_bench_starts_its_own_ranks_echo_with_the_erle_gather(built):
    r = run_bench(["--workload", "echo", "--channels", "8192", "--echo-seconds", "3", "--no-cpu-baseline"])
    assert r["n_gpus"] == 1 and r["value"] > 0
    erle = r["config"]["erle_db_single_talk_channels"]
    assert erle["ranks"] == 1
    assert erle["median"] > 10.0        # the cancellers converged and their ERLE came through the gather


def test_bench_with_a_bank_in_queue_mode_and_the_gather(built):
    """--queues 2 at 131 072 channels with the RCCL digit gather forced on: every collective is queued behind a
    spangpu_bank_join() (spandsp_amd/parallel.py), and the digits that arrive are checked by bench.py itself."""
    r = run_bench(["--steps", "50", "--warmup", "10", "--channels", "131072", "--queues", "2", "--distinct-frames", "30",
                   "--no-cpu-baseline", "--no-e2e", "--no-paths", "--min-timed-ms", "5"])
    assert r["config"]["queues"] == 2 and r["value"] > 0
