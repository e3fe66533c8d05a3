"""bench.py's launch contract, checked without a GPU: `--gpus N` with no launcher around it starts the
ranks itself and REFUSES (non-zero exit, a message, no JSON line) when fewer than N devices are
visible -- it never prints an n_gpus: 1 line for an N-GPU request (missing #1)."""
import os
import subprocess
import sys

import pytest

from conftest import REPO


def _run(*args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *args], env=env, capture_output=True,
                          text=True, timeout=timeout)


def _visible():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def test_more_gpus_than_visible_is_refused_loudly():
    n = _visible() + 1 if _visible() else 2
    r = _run("--gpus", str(n), "--steps", "2", "--warmup", "1")
    assert r.returncode != 0
    assert "refusing to run" in r.stderr and f"--gpus {n}" in r.stderr
    assert r.stdout.strip() == ""                        # no JSON line of a smaller job


def test_world_size_mismatch_is_refused():
    r = _run("--gpus", "4", env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
    assert "n_gpus" not in r.stdout


def test_no_gpu_means_no_number():
    if _visible():
        pytest.skip("a GPU is visible")
    r = _run("--gpus", "1", "--steps", "2", "--warmup", "1")
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)
    assert "value" not in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [(), ("--force-shard", "--n", "2048")])
def test_exactly_one_json_line_on_stdout(extra):
    """The contract: rank 0 prints ONE JSON line.  RCCL writes its version banner to the process's
    stdout as late as library teardown; bench.py keeps fd 1 parked on stderr so that nothing but the
    line arrives -- with the process group (and the RCCL communicator) of the sharded code path too."""
    import json
    r = _run("--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline", *extra, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[:2000]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["warmup"] == 1 and d["steps"] == 2 and d["n_gpus"] == 1
    assert d["roofline"]["bound"] == "mfma" and 0.2 < d["roofline"]["frac"] < 1.0
