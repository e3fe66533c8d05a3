#!/usr/bin/env python3
"""i8_ksweep.py -- where the int8 GEMM's time goes, from the outside: M = N = 4096 (one 256x256 tile per CU), K swept.
us per launch = fixed (launch, prologue, the 64 MB C store) + K x slope (the K loop); the slope against what the
matrix pipe sustains on random operands (mmh_probe_mfma_i8_sustained) is the loop's efficiency."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
import how_to_optimize_gemm_amd as H  # noqa: E402

modes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 8, 9]
mm = H.MMult(0, "auto")
g = torch.Generator(device="cuda").manual_seed(3)
n = 4096
probe = mm.probe_mfma_i8_sustained(True, 50.0)
print(json.dumps({"probe_mfma_i8_random_operands_tops": round(probe, 1)}))
for mode in modes:
    mm.set_igemm_mode(mode)
    pts = []
    for k in (1024, 2048, 4096, 8192, 16384):
        a = torch.randint(-127, 128, (n, k), device="cuda", dtype=torch.int8, generator=g)
        b = torch.randint(-127, 128, (k, n), device="cuda", dtype=torch.int8, generator=g)
        c = torch.empty((n, n), device="cuda", dtype=torch.int32)
        best = 1e9
        for rnd in range(3):
            for _ in range(max(20, 12000 // (k // 64))):
                mm.igemm_s8(a, b, out=c)
            reps = max(10, 6000 // (k // 64))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                mm.igemm_s8(a, b, out=c)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps * 1e3)
        pts.append((k, best))
    # least squares us = fixed + slope * k
    sx = sum(k for k, _ in pts); sy = sum(t for _, t in pts); sxx = sum(k * k for k, _ in pts); sxy = sum(k * t for k, t in pts)
    m_ = len(pts)
    slope = (m_ * sxy - sx * sy) / (m_ * sxx - sx * sx)
    fixed = (sy - slope * sx) / m_
    loop_tops = 2.0 * n * n / (slope * 1e-6) / 1e12          # ops per unit of K over the slope
    print(json.dumps({"mode": mode, "us_at_k": {k: round(t, 1) for k, t in pts}, "fixed_us": round(fixed, 1),
                      "slope_us_per_1024_k": round(slope * 1024, 2), "loop_tops": round(loop_tops, 1),
                      "loop_vs_pipe_on_random_operands": round(loop_tops / probe, 3),
                      "tops_at_4096": round(2.0 * n ** 3 / (dict(pts)[4096] * 1e-6) / 1e12, 1)}), flush=True)
