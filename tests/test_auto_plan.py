"""MMH_KERNEL_AUTO's tile choice as host arithmetic (mmh_auto_plan: the launch path's own auto_kernel + streamk_wanted
on a default handle, no device): pinned to what the GPU actually launched in the committed round-3 evidence, and held
to a few invariants over thousands of shapes, so that an edit of csrc/policy.hip that moves a decision shows up here,
on the CPU, before it shows up as a sweep point.  (The reference makes this choice by hand: `NEW := MMult_cuda_12` in
cuda/makefile:1-3.)"""
import json
import math
import os
import random
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _library_loads():
    try:
        import how_to_optimize_gemm_amd as H
        H.lib()
        return True
    except Exception:
        return False


pytestmark = pytest.mark.skipif(not _library_loads(), reason="libmmult_hip.so (or the HIP runtime it links) is not loadable here")

TILE = {"mfma_64x64_dma": (64, 64), "mfma_128x64_dma": (128, 64), "mfma_128x128_dma": (128, 128), "mfma_256x256": (256, 256),
        "mfma": (128, 128), "mfma_128x64": (128, 64), "mfma_64x64": (64, 64)}
FAMILY = {("mfma_dma", 64, 64): "mfma_64x64_dma", ("dma_streamk", 64, 64): "mfma_64x64_dma",
          ("mfma_dma", 128, 64): "mfma_128x64_dma", ("dma_streamk", 128, 64): "mfma_128x64_dma",
          ("mfma_dma", 128, 128): "mfma_128x128_dma", ("dma_streamk", 128, 128): "mfma_128x128_dma",
          ("mfma", 256, 256): "mfma_256x256", ("mfma_streamk", 256, 256): "mfma_256x256"}


def launched(text):
    """(kernel, tiles, persistent workgroups) out of an mmh_last_launch string."""
    m = re.match(r"sgemm_(\w+)_kernel<(\d+),(\d+)>", text)
    name = FAMILY[(m.group(1), int(m.group(2)), int(m.group(3)))]
    sk = re.search(r"(\d+) tiles on (\d+) persistent", text)
    if sk:
        return name, int(sk.group(1)), int(sk.group(2))
    return name, int(re.search(r"(\d+) workgroups", text).group(1)), 0


def test_the_plan_is_what_the_gpu_launched_on_the_reference_sweep():
    """profiles/r03_sweep_auto_launches.json: the harness's JSON sidecar of the sustained sweep, 25 sizes."""
    import how_to_optimize_gemm_amd as H
    rows = json.load(open(os.path.join(REPO, "profiles", "r03_sweep_auto_launches.json")))
    assert len(rows) == 25
    for r in rows:
        assert H.auto_plan(r["m"], r["n"], r["k"]) == launched(r["launched"]), r["p"]


def test_the_plan_is_what_the_gpu_launched_off_the_grid():
    """profiles/r03_offgrid_vs_vendor.json: 133 ragged / odd-leading-dimension / non-square shapes."""
    import how_to_optimize_gemm_amd as H
    rows = json.load(open(os.path.join(REPO, "profiles", "r03_offgrid_vs_vendor.json")))
    assert len(rows) >= 130
    for r in rows:
        got = H.auto_plan(r["m"], r["n"], r["k"], r["lda"], r["ldb"], r["ldc"])
        assert got == launched(r["launched"]), (r["m"], r["n"], r["k"], r["lda"], got, r["launched"])


def fill(m, n, bm, bn):
    return m * n / (math.ceil(m / bm) * bm * math.ceil(n / bn) * bn)


def test_invariants_over_many_shapes():
    import how_to_optimize_gemm_amd as H
    rng = random.Random(20260924)
    shapes = [(rng.randint(1, 9000), rng.randint(1, 9000), rng.randint(1, 9000)) for _ in range(3000)]
    shapes += [(n, n, n) for n in range(64, 8200, 37)]
    for (m, n, k) in shapes:
        name, tiles, grid = H.auto_plan(m, n, k)
        bm, bn = TILE[name]
        assert tiles == math.ceil(m / bm) * math.ceil(n / bn)
        # a stream-K grid is w workgroups per CU, never more ranges than tiles, and only for ragged counts
        assert grid in (0, -1) or (grid in (256, 512, 768) and tiles >= grid and tiles % grid != 0), (m, n, k, name, tiles, grid)
        # padding: never a tile that wastes much more of its area than the best-fitting one -- and next to nothing
        # more once the shape has rounds of tiles to spare
        best = max(fill(m, n, *t) for t in ((64, 64), (128, 64), (128, 128), (256, 256)))
        assert fill(m, n, bm, bn) >= best - (0.13 if tiles < 1024 else 0.07), (m, n, k, name, tiles)
        assert H.auto_plan(m, n, k) == (name, tiles, grid)


def test_one_row_and_column_less_keeps_the_tile():
    """N - 1 has N's tile counts: the guarded instantiation of the same tile, the same launch form (round 2 sent
    every off-grid shape to the register-staged kernels)."""
    import how_to_optimize_gemm_amd as H
    for n in range(1024, 4097, 128):
        assert H.auto_plan(n - 1, n - 1, n - 1) == H.auto_plan(n, n, n), n
        assert H.auto_plan(n, n, n, n + 1, n + 1, n + 1, base_align=4)[0] == H.auto_plan(n, n, n)[0], n   # odd rows, 4-byte bases


def test_named_decisions():
    import how_to_optimize_gemm_amd as H
    assert H.auto_plan(4096, 4096, 4096) == ("mfma_128x64_dma", 2048, 0)           # the headline: four whole rounds of two per CU, plain
    assert H.auto_plan(4096, 4096, 1024) == ("mfma_64x64_dma", 4096, 0)            # a short K loop keeps the small tile
    assert H.auto_plan(3072, 3072, 3072) == ("mfma_64x64_dma", 2304, 0)            # nine tiles per CU; 1152 of 128x64 are not whole rounds
    for rows in (16384, 8192, 4096, 2048):                                         # the config-4 panels: B beyond the Infinity Cache
        assert H.auto_plan(rows, 16384, 16384)[0] == "mfma_256x256", rows
    assert H.auto_plan(2817, 2817, 2817) == ("mfma_64x64_dma", 45 * 45, 0)          # pads less than the 128x64 grid
    assert H.auto_plan(2689, 2689, 2689)[0] == "mfma_128x64_dma"                    # 43 x 43 tiles: a last round a quarter full
    assert H.auto_plan(1024, 1024, 1024, cu_count=64)[0] != "mfma_64x64_dma" or H.auto_plan(1024, 1024, 1024, cu_count=64)[2] >= 0
    with pytest.raises(H.MMultError):
        H.auto_plan(0, 4, 4)
    with pytest.raises(H.MMultError):
        H.auto_plan(8, 8, 8, lda=4)
