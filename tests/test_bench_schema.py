"""The JSON line of `bench.py --gpus N [--force-shard] [--sweep]` (the sharded code path), as a schema: checked on the
CPU against the committed line of the one-rank run (profiles/r04_bench_forceshard_sweep.json, written on a GPU box by
tools/r04_final.sh) and -- on a GPU -- against a live run (tests/test_bench_cli.py covers the N = 1 line).  VERDICT r03,
item 2c."""
import json
import os

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline")
SHARDED = ("rccl_ranks", "backend", "bcast_ms", "bcast_gbps", "gemm_ms_per_rank", "gemm_ms", "value_incl_bcast", "streamed_equals_plain",
           "model")


def check_sharded_line(d, world, sweep):
    for key in CONTRACT + SHARDED:
        assert key in d, key
    assert d["n_gpus"] == world == d["rccl_ranks"] == len(d["gemm_ms_per_rank"])
    assert d["scaling"] == "strong" and d["unit"] == "GFLOPS" and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert d["config"]["parallelism"] == f"row-panel x{world}"
    assert d["gemm_ms"] == max(d["gemm_ms_per_rank"]) > 0
    assert d["roofline"]["bound"] == "mfma" and 0.0 < d["roofline"]["frac"] < 1.0
    if "single_gpu_value" in d:
        assert abs(d["scaling_efficiency"] - d["value"] / (world * d["single_gpu_value"])) < 2e-3
        assert 0.2 < d["scaling_efficiency"] < 1.3
    m = d["model"]
    for key in ("bcast_flat_ms", "bcast_scatter_allgather_ms", "b_chunks", "streamed_ms", "value_at_linear_scaling"):
        assert key in m, key
    assert d["streamed_equals_plain"] is True
    if sweep:
        sw = d["sweep_gflops_sharded"]
        sizes = [int(k) for k in sw["gflops"]]
        assert sizes == sorted(sizes) and sizes[0] == 1024 and all(p % 128 == 0 for p in sizes)
        assert all(v is None or v > 0 for v in sw["gflops"].values())


def test_the_committed_one_rank_line_has_the_sharded_schema():
    path = os.path.join(REPO, "profiles", "r04_bench_forceshard_sweep.json")
    if not os.path.exists(path):
        pytest.skip("profiles/r04_bench_forceshard_sweep.json has not been collected yet")
    d = json.loads(open(path).read().strip().splitlines()[-1])
    check_sharded_line(d, 1, sweep=True)


@pytest.mark.gpu
def test_a_live_one_rank_sharded_sweep_has_the_schema():
    """`bench.py --gpus 1 --force-shard --sweep`: process group, RCCL broadcast, streamed B, per-rank times, the
    single-GPU reference and the square sweep -- on one rank, the code an N-rank run executes."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--force-shard", "--n", "4096", "--steps", "2",
                        "--warmup", "1", "--sweep", "--b-chunks", "4", "--no-extras", "--no-cpu-baseline"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    check_sharded_line(d, 1, sweep=True)
    assert d["model"]["b_chunks"] == 4
