"""bench.py's N > 1 path, rehearsed on the one-GPU box: `--gpus 2` without torchrun makes bench.py start its two
ranks itself; `--dist-backend gloo` keeps the control-path collectives (barrier, max of the times) on the CPU so
both ranks can share device 0.  Checks the launch protocol and the line, not a scaling figure."""
import json
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT

COMMON = ["--steps", "40", "--warmup", "4", "--settle-ms", "100", "--no-configs", "--no-cpu-baseline", "--no-generic"]


def run_bench(*extra):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(extra) + COMMON, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.fixture(scope="module")
def single():
    return run_bench("--gpus", "1")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["frames", "stripes"])
def test_two_ranks_launch_themselves_and_report_one_line(mode, single):
    two = run_bench("--gpus", "2", "--dist-backend", "gloo", "--mode", mode)
    assert two["n_gpus"] == 2 and single["n_gpus"] == 1
    assert len(two["per_rank_kernel_ms"]) == 2 and all(ms > 0 for ms in two["per_rank_kernel_ms"])
    assert two["scaling"] == ("strong" if mode == "stripes" else "weak")
    assert two["config"]["dist_backend"] == "gloo"
    # every rank verified its own rows against the oracle, and its host copies equal its device frames
    assert two["verified"] is True and two["verification"]["ranks_verified"] == 2
    assert two["host_delivered"]["n_gpus"] == 2 and two["host_delivered"]["host_copy_equals_device_frame"] is True
    assert single["verified"] is True
    # two ranks time-share one GPU here: the whole-job rate stays about that of one rank on it (two processes' kernels
    # overlap at each other's tails, so somewhat more is possible; much less would mean ranks waiting on each other)
    assert 0.90 <= two["value"] / single["value"] <= 1.35, (two["value"], single["value"])


def test_rank_count_mismatch_is_an_error():
    """No GPU needed: the check comes before anything is imported."""
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr
