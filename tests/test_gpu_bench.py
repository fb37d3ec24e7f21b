"""bench.py as the driver starts it: plain `python bench.py --gpus N` with no launcher around it must still run N ranks
(it spawns them as fresh child processes before touching the GPU).  On this one-GPU box the ranks share cuda:0 and talk
over gloo (SC_BENCH_REHEARSAL=1: the timings mean nothing, the code path is the N > 1 one)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("gpus,extra", [(2, []), (3, ["--shard", "replicated"])])
def test_bench_spawns_its_ranks_without_a_launcher(gpus, extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["SC_BENCH_REHEARSAL"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1",
                        "--config", "C1", "--no-cpu-baseline"] + extra, env=env, capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == gpus and out["winner"]["status"] == 0 and out["value"] > 0
    assert "roofline" in out and "REHEARSAL" in out["data"]
