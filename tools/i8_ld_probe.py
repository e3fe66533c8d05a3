#!/usr/bin/env python3
"""i8_ld_probe.py -- is the int8 GEMM's K loop bound by WHERE its operand lines live?  Same 4096^3 / 8192^3 problem with
leading dimensions lda = ldb = N (a power of two: every row of a K-slice of A is N bytes from the next, i.e. on the
same L2 channel if channels interleave below N) and with lda / ldb padded by 128 / 256 / 384 bytes."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
import how_to_optimize_gemm_amd as H  # noqa: E402

modes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 8]
mm = H.MMult(0, "auto")
g = torch.Generator(device="cuda").manual_seed(3)
for n in (4096, 8192):
    for pad in (0, 128, 256, 384, 64):
        abuf = torch.randint(-127, 128, (n, n + pad), device="cuda", dtype=torch.int8, generator=g)
        bbuf = torch.randint(-127, 128, (n, n + pad), device="cuda", dtype=torch.int8, generator=g)
        a, b = abuf[:, :n], bbuf[:, :n]
        c = torch.empty((n, n), device="cuda", dtype=torch.int32)
        row = {"n": n, "ld_pad_bytes": pad}
        for mode in modes:
            mm.set_igemm_mode(mode)
            best = 0.0
            for rnd in range(3):
                for _ in range(200 if n == 4096 else 40):
                    mm.igemm_s8(a, b, out=c)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 100 if n == 4096 else 20
                e0.record()
                for _ in range(reps):
                    mm.igemm_s8(a, b, out=c)
                e1.record()
                torch.cuda.synchronize()
                best = max(best, 2.0 * n ** 3 / (e0.elapsed_time(e1) / reps * 1e-3) / 1e12)
            row[f"mode{mode}_tops"] = round(best, 1)
        print(json.dumps(row), flush=True)
