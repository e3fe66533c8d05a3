#!/usr/bin/env python3
"""Run-to-run spread of one launch configuration: BURSTS independent bursts (another kernel in between),
each 5 warm-up + REPS timed launches; prints min / median / max TFLOP/s and how many bursts fell more
than 10 % under the median.  Numeric option pairs (A/B library): --opt 100=0
usage: python tools/bimodal.py --n 1536 [--kernel auto] [--bursts 40] [--opt ID=VALUE ...]"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import how_to_optimize_gemm_amd as H  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, nargs="+", default=[1536])
ap.add_argument("--kernel", default="auto")
ap.add_argument("--bursts", type=int, default=40)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--opt", action="append", default=[])
args = ap.parse_args()
if args.opt:
    H.use_ab_library()
mm = H.MMult(0, args.kernel)
for o in args.opt:
    k, v = o.split("=")
    assert H.lib().mmh_set_option(mm._h, int(k), int(v)) == 0, o
stream = torch.cuda.current_stream().cuda_stream
w = torch.rand((2048, 2048), device="cuda")
wc = torch.empty_like(w)
for n in args.n:
    a = torch.rand((n, n), device="cuda") * 2 - 1
    b = torch.rand((n, n), device="cuda") * 2 - 1
    c = torch.empty((n, n), device="cuda")
    for _ in range(100):
        mm.matmul(a, b, out=c)
    res = []
    for i in range(args.bursts):
        torch.mm(w, w, out=wc)                      # something else on the chip in between
        if i % 4 == 0:
            torch.cuda.synchronize()
        ms = mm.time_sgemm(n, n, n, a.data_ptr(), n, b.data_ptr(), n, c.data_ptr(), n, warmup=5, reps=args.reps, stream=stream)
        res.append(2.0 * n ** 3 / ms / 1e9)
    med = statistics.median(res)
    low = sum(1 for r in res if r < 0.9 * med)
    print(f"N={n} {args.kernel} opts={args.opt}: min {min(res):.1f} median {med:.1f} max {max(res):.1f} TFLOP/s, "
          f"{low}/{len(res)} bursts > 10 % under the median | {H.last_launch()}", flush=True)
