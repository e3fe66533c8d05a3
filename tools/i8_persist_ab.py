#!/usr/bin/env python3
"""i8_persist_ab.py -- the int8 ping-pong kernel K3p as a persistent launch (MMH_OPT_IGEMM_MODE 8: what mode 0 runs) against
one workgroup per tile (9: round 5's launch form), the lockstep K3t kernel (6) and the config-named 16x16x32 instruction
(7), on shapes with 1 .. 16 tiles of 256x256 per CU.  Sustained rates: interleaved rounds over the modes, ~10 ms of untimed launches in front
of every timed burst, median over the rounds; every mode's C compared with the first mode's, bit for bit."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
import how_to_optimize_gemm_amd as H  # noqa: E402

modes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [8, 9, 6, 7]
shapes = [tuple(int(v) for v in s.split("x")) for s in sys.argv[2].split(",")] if len(sys.argv) > 2 else \
    [(4096, 4096, 4096), (8192, 8192, 8192), (6144, 6144, 6144), (8192, 8192, 2048), (8192, 8192, 1024), (16384, 16384, 4096),
     (5120, 5120, 5120), (4096, 8192, 4096)]
ROUNDS = int(os.environ.get("ROUNDS", "7"))
mm = H.MMult(0, "auto")
g = torch.Generator(device="cuda").manual_seed(3)
probe = mm.probe_mfma_i8_sustained(True, 50.0)
print(json.dumps({"probe_mfma_i8_random_operands_tops": round(probe, 1)}), flush=True)
for (m, n, k) in shapes:
    a = torch.randint(-127, 128, (m, k), device="cuda", dtype=torch.int8, generator=g)
    b = torch.randint(-127, 128, (k, n), device="cuda", dtype=torch.int8, generator=g)
    c = torch.empty((m, n), device="cuda", dtype=torch.int32)
    first = None
    row = {"shape": [m, n, k], "tiles_per_cu": round(((m + 255) // 256) * ((n + 255) // 256) / 256, 2)}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    one = {}
    for mode in modes:
        mm.set_igemm_mode(mode)
        mm.igemm_s8(a, b, out=c)
        torch.cuda.synchronize()
        if first is None:
            first = c.clone()
        elif not torch.equal(c, first):
            row[f"mode{mode}_MISMATCH"] = True
        e0.record(); mm.igemm_s8(a, b, out=c); e1.record(); torch.cuda.synchronize()
        one[mode] = max(e0.elapsed_time(e1), 1e-3)
    # interleaved rounds (the first mode measured on a cool chip reads ~5 % fast: every round visits every mode, in an
    # order that rotates), ~10 ms of the mode's own launches in front of each timed burst; median over the rounds
    times = {mode: [] for mode in modes}
    for rnd in range(ROUNDS):
        for mode in modes[rnd % len(modes):] + modes[:rnd % len(modes)]:
            mm.set_igemm_mode(mode)
            for _ in range(max(3, int(10.0 / one[mode]))):
                mm.igemm_s8(a, b, out=c)
            reps = max(5, int(10.0 / one[mode]))
            e0.record()
            for _ in range(reps):
                mm.igemm_s8(a, b, out=c)
            e1.record()
            torch.cuda.synchronize()
            times[mode].append(e0.elapsed_time(e1) / reps * 1e3)
    for mode in modes:
        best = sorted(times[mode])[len(times[mode]) // 2]
        row[f"mode{mode}_us"] = round(best, 1)
        row[f"mode{mode}_tops"] = round(2.0 * m * n * k / (best * 1e-6) / 1e12, 1)
    print(json.dumps(row), flush=True)
    del a, b, c, first
mm.set_igemm_mode(0)
