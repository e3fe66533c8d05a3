#!/usr/bin/env python3
"""Small-N half of the reference sweep (N = 1024 .. 2048): the chain kernels (AUTO and its tiles),
the opt-in split-K launches, the VALU rung's two tiles, and the two vendor libraries, interleaved in
one process (cdna guide rule 24), median of ROUNDS rounds.  Writes a markdown table to stdout.
Every timed burst (REPS launches) follows WARM_MS milliseconds of untimed launches of the SAME variant
(default 30): the sustained rate, as the harness's WARMUP_MS measures it -- a variant inherits whatever
clock state the previous variant left, and the tiles differ in how many launches they need to settle
(profiles/r02_cold_start.txt: 18 .. 40).  --warm-ms 0 gives the old 5-launch warm-up.
usage: python tools/smalln_sweep.py [--sizes 1024,1152,...] [--rounds 5] [--warm-ms 30]"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import how_to_optimize_gemm_amd as H  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="1024,1152,1280,1408,1536,1664,1792,1920,2048")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--warm-ms", type=float, default=30.0)
ap.add_argument("--variants", default="auto,mfma_64x64_dma,mfma_128x64_dma,mfma_128x128_dma,mfma_64x64,mfma_128x64,mfma_tiles,"
                                       "mfma_splitk:0,auto:1,rocblas,hipblaslt,valu,valu_128x128,valu_64x64")
args = ap.parse_args()
if os.environ.get("MMH_AB") == "1":       # numeric variants = ids of the A/B library (e.g. 45/nosk)
    H.use_ab_library()
sizes = [int(x) for x in args.sizes.split(",")]
variants = args.variants.split(",")
mm = H.MMult(0)
stream = torch.cuda.current_stream().cuda_stream
torch.backends.cuda.matmul.allow_tf32 = False

w = torch.rand((4096, 4096), device="cuda")
wc = torch.empty_like(w)
for _ in range(200):                      # clock ramp
    mm.matmul(w, w, out=wc)
torch.cuda.synchronize()


def warm(fn):
    """args.warm_ms of untimed launches (at least 5)."""
    import time
    for _ in range(5):
        fn()
    if args.warm_ms > 0:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < args.warm_ms:
            for _ in range(5):
                fn()
            torch.cuda.synchronize()


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    warm(fn)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print("| N | " + " | ".join(variants) + " |")
print("|---|" + "---|" * len(variants))
for n in sizes:
    a = torch.rand((n, n), device="cuda") * 2 - 1
    b = torch.rand((n, n), device="cuda") * 2 - 1
    c = torch.empty((n, n), device="cuda")
    res = {v: [] for v in variants}
    launch = {}
    for r in range(args.rounds):
        for v in variants:
            if v == "hipblaslt":
                ms = timed(lambda: torch.mm(a, b, out=c), args.reps)
            elif v == "rocblas":
                ms = timed(lambda: mm.matmul_rocblas(a, b, out=c), args.reps)
            else:
                name, _, s = v.partition(":")
                nosk, sk2 = name.endswith("/nosk"), name.endswith("/sk2")   # without stream-K / stream-K whenever ragged
                base = name[:-5] if nosk else (name[:-4] if sk2 else name)
                if base.isdigit():
                    assert H.lib().mmh_set_kernel(mm._h, int(base)) == 0, base
                else:
                    mm.set_kernel(base)
                mm.set_streamk(0 if nosk else (2 if sk2 else 1))
                mm.set_splitk(int(s) if s else 0)
                warm(lambda: mm.matmul(a, b, out=c))
                ms = mm.time_sgemm(n, n, n, a.data_ptr(), n, b.data_ptr(), n, c.data_ptr(), n, warmup=2,
                                   reps=args.reps, stream=stream)
                launch[v] = H.last_launch()
            res[v].append(2.0 * n ** 3 / (ms * 1e-3) / 1e12)
    mm.set_splitk(0)
    mm.set_streamk(True)
    print(f"| {n} | " + " | ".join(f"{statistics.median(res[v]):.1f}" for v in variants) + " |", flush=True)
    if n == sizes[0]:
        for v in variants:
            if v in launch:
                print(f"<!-- {v}: {launch[v]} -->")
