#!/usr/bin/env python3
"""i8_ab.py -- the int8 GEMM's kernel modes against each other: bit-equality on whole, ragged and K-tailed shapes,
then sustained TOPS at N = 4096 / 8192 (random operands in [-127, 127], interleaved rounds after ~20 ms of
warm-up launches of each mode).  mode 0 = what ships by default; 7 = the ping-pong schedule (igemm_s8_pp.hpp)."""
import json
import sys
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
import how_to_optimize_gemm_amd as H  # noqa: E402

modes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 6, 7]
mm = H.MMult(0, "auto")
g = torch.Generator(device="cuda").manual_seed(3)
bad = 0
for (m, n, k) in [(256, 256, 128), (256, 256, 256), (512, 768, 1024), (300, 260, 200), (1000, 1000, 1000), (257, 255, 129),
                  (4096, 4096, 4096), (1024, 2048, 515)]:
    a = torch.randint(-127, 128, (m, k), device="cuda", dtype=torch.int8, generator=g)
    b = torch.randint(-127, 128, (k, n), device="cuda", dtype=torch.int8, generator=g)
    c0 = torch.randint(-1000, 1000, (m, n), device="cuda", dtype=torch.int32, generator=g)
    mm.set_igemm_mode(2)
    want = mm.igemm_s8(a, b)
    wacc = c0.clone()
    mm.igemm_s8(a, b, out=wacc, accumulate=True)
    for mode in modes:
        mm.set_igemm_mode(mode)
        got = mm.igemm_s8(a, b)
        acc = c0.clone()
        mm.igemm_s8(a, b, out=acc, accumulate=True)
        ok = bool(torch.equal(got, want)) and bool(torch.equal(acc, wacc))
        bad += 0 if ok else 1
        print(f"exact  mode {mode}  {m}x{n}x{k}: {'ok' if ok else 'MISMATCH'}", flush=True)
rows = []
for n in (4096, 8192):
    a = torch.randint(-127, 128, (n, n), device="cuda", dtype=torch.int8, generator=g)
    b = torch.randint(-127, 128, (n, n), device="cuda", dtype=torch.int8, generator=g)
    c = torch.empty((n, n), device="cuda", dtype=torch.int32)
    res = {mode: [] for mode in modes}
    for rnd in range(4):
        for mode in modes:
            mm.set_igemm_mode(mode)
            warm = 300 if n == 4096 else 50
            for _ in range(warm):
                mm.igemm_s8(a, b, out=c)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 100 if n == 4096 else 20
            e0.record()
            for _ in range(reps):
                mm.igemm_s8(a, b, out=c)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                res[mode].append(2.0 * n ** 3 / (e0.elapsed_time(e1) / reps * 1e-3) / 1e12)
    row = {"n": n, **{f"mode{mode}_tops": round(sorted(v)[1], 1) for mode, v in res.items()}}
    rows.append(row)
    print(json.dumps(row), flush=True)
print("int8 A/B:", "ALL EXACT" if not bad else f"{bad} MISMATCHES")
