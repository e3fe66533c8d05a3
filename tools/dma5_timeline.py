#!/usr/bin/env python3
"""dma5_timeline.py -- per-workgroup wall-clock stamps of ONE plain K2W launch (libmmult_hip_tl.so, a build of its
own): entry, prologue done, K loop done, C stores done, reported apart for the workgroups of the whole tiles (the
first n_full block ids) and for the thin tiles a ragged shape dispatches last (sgemm_mfma_dma5_kernel).
usage: python tools/dma5_timeline.py [--kernel mfma_64x64_dma5] [--shape 1025,1025,1025] [--launches 20]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import how_to_optimize_gemm_amd as H  # noqa: E402

H.use_timeline_library()
ap = argparse.ArgumentParser()
ap.add_argument("--kernel", default="mfma_64x64_dma5")
ap.add_argument("--shape", nargs="+", default=["1025,1025,1025"])
ap.add_argument("--launches", type=int, default=20)
args = ap.parse_args()
L = H.lib()
L.mmh_ab_set_stamps5.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
mm = H.MMult(0, args.kernel)
mm.set_streamk(0)
bm, bn = (int(x) for x in args.kernel.split("_")[1].split("x"))
hip = ctypes.CDLL("libamdhip64.so")
v = ctypes.c_int(0)
hip.hipDeviceGetAttribute(ctypes.byref(v), 10017, 0)   # hipDeviceAttributeWallClockRate (kHz)
wall_khz = v.value or 100000
stream = torch.cuda.current_stream().cuda_stream
print(f"wall clock {wall_khz} kHz; kernel {args.kernel}")
for sh in args.shape:
    m, n, k = (int(x) for x in sh.split(","))
    a = torch.rand((m, k), device="cuda") * 2 - 1
    b = torch.rand((k, n), device="cuda") * 2 - 1
    c = torch.empty((m, n), device="cuda")
    for _ in range(300):
        mm.matmul(a, b, out=c)
    torch.cuda.synchronize()
    ms = mm.time_sgemm(m, n, k, a.data_ptr(), k, b.data_ptr(), n, c.data_ptr(), n, warmup=20, reps=100, stream=stream)
    print(f"{sh}: {H.last_launch()}")
    nbm, nbn = -(-m // bm), -(-n // bn)
    thin_row = 1 if nbm > 1 and m - (nbm - 1) * bm <= 16 else 0
    thin_col = 1 if nbn > 1 and n - (nbn - 1) * bn <= 16 else 0
    n_full = (nbm - thin_row) * (nbn - thin_col)
    if "rim wave" in H.last_launch():          # the RIM launch: the trimmed grid; report its last tile row / column apart
        nbm, nbn = (m - m % bm) // bm if m % bm == 1 else nbm, (n - n % bn) // bn if n % bn == 1 else nbn
        n_full = nbm * nbn
    stamps = torch.zeros((1 << 16, 4), device="cuda", dtype=torch.int64)
    assert L.mmh_ab_set_stamps5(mm._h, stamps.data_ptr()) == 0
    acc = []
    for it in range(args.launches):
        stamps.zero_()
        for _ in range(5):
            mm.matmul(a, b, out=c)      # the stamped launch is the last of a burst of five
        torch.cuda.synchronize()
        s = stamps[:nbm * nbn].cpu().double()
        t0 = s[:, 0].min()
        acc.append((s - t0) * 1e3 / wall_khz)
    assert L.mmh_ab_set_stamps5(mm._h, None) == 0
    us = torch.stack(acc).median(dim=0).values      # per workgroup, per stamp: median over the launches
    flops = 2.0 * m * n * k
    print(f"   back-to-back {ms * 1e3:.1f} us/launch = {flops / ms / 1e9:.1f} TFLOP/s; {nbm * nbn} workgroups, {n_full} whole tiles first")
    for label, sel in (("whole", us[:n_full]), ("thin ", us[n_full:])):
        if len(sel) == 0:
            continue
        q = lambda x: f"{x.min():.2f}/{x.median():.2f}/{x.max():.2f}"
        print(f"   {label}: entry {q(sel[:, 0])} | prologue {q(sel[:, 1] - sel[:, 0])} | K loop {q(sel[:, 2] - sel[:, 1])} | "
              f"store {q(sel[:, 3] - sel[:, 2])} | done {q(sel[:, 3])}   (min/median/max us)")
