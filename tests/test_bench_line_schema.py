"""The N = 1 JSON line of `bench.py` as a schema: the metric's WHOLE sweep (cuda/parameters.h:5-7,
25 sizes) for `auto`, both vendor libraries and the VALU rung, configs[2]'s literal 128x128 tile at 4096, the roofline
object with the matrix pipe's busy fraction and the HBM-side GB/s, and configs[4]'s own roofline object.  Checked on the
CPU against bench.py's key lists and the committed line of the round (profiles/r06_bench_line.json, written on a GPU box);
on a GPU against a live run."""
import json
import os

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(REPO, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def check_line(d, B, live_counters=True):
    for key in CONTRACT:
        assert key in d, key
    assert d["n_gpus"] == 1 and d["unit"] == "GFLOPS" and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert d["metric"] == B.METRIC and "workload" in d["config"] and "model" not in d["config"]
    rl = d["roofline"]
    for key in B.ROOFLINE_KEYS:
        assert key in rl, key
    assert rl["bound"] == "mfma" and rl["peak"] == B.PEAK_FP32_MFMA_TFLOPS and abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-3
    if live_counters:
        assert 0.3 < rl["mfma_busy_frac"] <= 1.0
        assert rl["traffic"] >= rl["algorithmic_bytes_per_launch"] * 0.9
        assert abs(rl["hbm_gbps"] - rl["traffic"] / (rl["kernel_ms"] * 1e-3) / 1e9) < 1.0
    cb = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert cb["kind"] in ("reference", "port") and "RESTATED" in cb["sample"]
    sw = d["extras"]["sweep_gflops"]
    for kern in B.SWEEP_KERNELS_ALL_SIZES:
        have = [p for p in B.SWEEP_SIZES if f"{kern}_{p}" in sw]
        assert have == list(B.SWEEP_SIZES), (kern, have)
        assert all(sw[f"{kern}_{p}"] > 0 for p in B.SWEEP_SIZES)
    assert len(B.SWEEP_SIZES) == 25 and B.SWEEP_SIZES[0] == 1024 and B.SWEEP_SIZES[-1] == 4096
    for key in B.SWEEP_KEYS_AT_4096:
        assert sw[key] > 0, key
    i8 = d["extras"]["int8_roofline"]
    for key in B.INT8_ROOFLINE_KEYS:
        assert key in i8, key
    assert i8["peak_spec"] == 5030.0 and abs(i8["frac"] - i8["achieved"] / 5030.0) < 1e-3
    assert i8["16x16x32_bit_equal_to_16x16x64"] is True
    assert d["extras"]["sweep_summary"]["sizes"] == 25


def test_bench_key_lists_name_what_the_verdict_asked_for():
    B = _bench()
    assert "mfma_busy_frac" in B.ROOFLINE_KEYS and "hbm_gbps" in B.ROOFLINE_KEYS
    assert B.SWEEP_KEYS_AT_4096 == ("mfma_tiles_4096", "mfma_128x128_dma5_4096")
    assert {"achieved", "peak_spec", "peak_measured_random", "frac", "kernel", "kernel_ms"} <= set(B.INT8_ROOFLINE_KEYS)
    assert tuple(B.SWEEP_SIZES) == tuple(range(1024, 4097, 128))
    assert B.PMC_PASSES[2] == ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE")       # its own pass, beside FETCH_SIZE / WRITE_SIZE


def test_the_committed_line_of_the_round_has_the_schema():
    path = os.path.join(REPO, "profiles", "r06_bench_line.json")
    if not os.path.exists(path):
        pytest.skip("profiles/r06_bench_line.json has not been collected yet")
    d = json.loads(open(path).read().strip().splitlines()[-1])
    check_line(d, _bench())


@pytest.mark.gpu
def test_a_live_default_line_has_the_schema():
    """`python bench.py` as the driver runs it (N = 1, defaults but fewer steps): one line, every key."""
    import subprocess
    import sys
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "10", "--warmup", "3"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert "error" not in d.get("extras", {}), d["extras"].get("error")
    check_line(d, _bench())
    assert time.time() - t0 < 240
