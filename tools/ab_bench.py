#!/usr/bin/env python3
"""Within-process interleaved A/B of kernel variants (cdna guide rule 24):
N variants x R rounds, report median / best TFLOP/s per variant.
usage: python tools/ab_bench.py [--n 4096] [--rounds 7] [--reps 10] [--splitk S] mfma mfma_simple 16 17 ...
Loads libmmult_hip_ab.so (built on demand): numeric ids 16-19 are scheduling variants with valid
results, 21-24 / 32-44 TIMING-ONLY ablation builds whose results are wrong (profiles/r01_ablation.md).
A variant written name:S (e.g. mfma_splitk:4) runs with MMH_OPT_SPLITK = S."""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import how_to_optimize_gemm_amd as H  # noqa: E402

if os.environ.get("MMH_AB", "1") != "0":   # the A/B variants and ablation builds live in the tools-only library
    H.use_ab_library()

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--m", type=int, default=0)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--check", action="store_true")
ap.add_argument("variants", nargs="+")
args = ap.parse_args()
n = args.n
m = args.m or n
mm = H.MMult(0)
a = torch.rand((m, n), device="cuda") * 2 - 1
b = torch.rand((n, n), device="cuda") * 2 - 1
_w = torch.rand((4096, 4096), device="cuda")
for _ in range(200):                      # clock ramp before the first round
    mm.matmul(_w, _w, out=_w.new_empty(4096, 4096) if _ == 0 else None) if False else None
_c = torch.empty((4096, 4096), device="cuda")
for _ in range(200):
    mm.matmul(_w, _w, out=_c)
torch.cuda.synchronize()
c = torch.empty((m, n), device="cuda")
stream = torch.cuda.current_stream().cuda_stream


def vid(v):
    v = v.split(":")[0]
    return H.KERNELS[v] if v in H.KERNELS else int(v)


def vsplit(v):
    return int(v.split(":")[1]) if ":" in v else 0


ref = None
res = {v: [] for v in args.variants}
for r in range(args.rounds):
    for v in args.variants:
        H.lib().mmh_set_kernel(mm._h, vid(v))
        mm.set_splitk(vsplit(v))
        if v == "rocblas":
            continue
        ms = mm.time_sgemm(m, n, n, a.data_ptr(), n, b.data_ptr(), n, c.data_ptr(), n, warmup=1,
                           reps=args.reps, stream=stream)
        res[v].append(2.0 * m * n * n / (ms * 1e-3) / 1e12)
        if args.check and r == 0:
            torch.cuda.synchronize()
            if ref is None:
                ref = c.clone()
            else:
                print(f"  {v}: bit-equal to first variant: {torch.equal(ref, c)}")
for v in args.variants:
    x = res[v]
    print(f"{v:>14}: median {statistics.median(x):7.2f}  best {max(x):7.2f}  worst {min(x):7.2f} TFLOP/s  (m={m}, n=k={n})")
