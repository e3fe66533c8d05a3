#!/usr/bin/env python3
"""sk_timeline.py -- per-workgroup, per-part wall-clock stamps of ONE stream-K K2W launch (libmmult_hip_tl.so): where a
persistent workgroup's time goes between kernel entry and exit -- the prologue of each part (first slice in LDS), its K
loop (per slice), what follows the loop (C stores / partial-tile publish / hand-over poll) and the gaps between parts.
usage: python tools/sk_timeline.py [--kernel mfma_128x128_dma5] [--shape 2304,2304,2304 ...] [--launches 20]
The stamps perturb the kernel a little (thread 0 of consumer wave 0 reads the wall clock 4 times per part); the
back-to-back time of the stamped build is printed beside them.  Needs a GPU."""
import argparse
import ctypes
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import how_to_optimize_gemm_amd as H  # noqa: E402

H.use_timeline_library()
ap = argparse.ArgumentParser()
ap.add_argument("--kernel", default="mfma_128x128_dma5")
ap.add_argument("--shape", nargs="+", default=["2304,2304,2304"])
ap.add_argument("--launches", type=int, default=10)
ap.add_argument("--streamk", type=int, default=2)
ap.add_argument("--burst", type=int, default=150)
args = ap.parse_args()
L = H.lib()
L.mmh_ab_set_stamps5.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
L.mmh_ab_set_stamp_stride5.argtypes = [ctypes.c_void_p, ctypes.c_int]
mm = H.MMult(0, args.kernel)
mm.set_streamk(args.streamk)
hip = ctypes.CDLL("libamdhip64.so")
v = ctypes.c_int(0)
hip.hipDeviceGetAttribute(ctypes.byref(v), 10017, 0)   # hipDeviceAttributeWallClockRate (kHz)
wall_khz = v.value or 100000
stream = torch.cuda.current_stream().cuda_stream
KIND = {0: "HEAD", 1: "WHOLE", 2: "TAIL"}
print(f"wall clock {wall_khz} kHz; kernel {args.kernel}")


def stats(xs):
    xs = sorted(xs)
    if not xs:
        return "-"
    return f"{xs[0]:.2f}/{xs[len(xs) // 2]:.2f}/{xs[-1]:.2f}"


for sh in args.shape:
    m, n, k = (int(x) for x in sh.split(","))
    a = torch.rand((m, k), device="cuda") * 2 - 1
    b = torch.rand((k, n), device="cuda") * 2 - 1
    c = torch.empty((m, n), device="cuda")
    for _ in range(200):
        mm.matmul(a, b, out=c)
    torch.cuda.synchronize()
    ms = mm.time_sgemm(m, n, k, a.data_ptr(), k, b.data_ptr(), n, c.data_ptr(), n, warmup=20, reps=100, stream=stream)
    launched = H.last_launch()
    print(f"{sh}: {launched}")
    print(f"   back-to-back {ms * 1e3:.1f} us/launch = {2.0 * m * n * k / ms / 1e9:.1f} TFLOP/s")
    if "streamk" not in launched:
        print("   (not a stream-K launch)")
        continue
    mt = re.search(r"on (\d+) persistent", launched)
    G = int(mt.group(1)) if mt else 256
    stamps = torch.zeros((4096, 32), device="cuda", dtype=torch.int64)
    assert L.mmh_ab_set_stamp_stride5(mm._h, 32) == 0
    assert L.mmh_ab_set_stamps5(mm._h, stamps.data_ptr()) == 0
    rows = {}
    totals, entries = [], []
    for it in range(args.launches):
        stamps.zero_()
        for _ in range(args.burst):
            mm.matmul(a, b, out=c)      # every launch of the burst writes the slots: the last one's are read (sustained clocks)
        torch.cuda.synchronize()
        s = stamps.cpu()
        live = s[:, 0] > 0
        t0 = int(s[live, 0].min())
        us = lambda x: (int(x) - t0) * 1e3 / wall_khz
        for w in range(s.shape[0]):
            if not live[w]:
                continue
            entries.append(us(s[w, 0]))
            totals.append(us(s[w, 1]))
            n_parts = int(s[w, 2]) & 0xff
            prev_end = us(s[w, 0])
            for p in range(min(n_parts, 7)):
                kind = KIND[(int(s[w, 2]) >> (8 + 2 * p)) & 3]
                ln = (int(s[w, 3]) >> (8 * p)) & 0xff
                base = 4 + 4 * p
                st, pro, loop, end = (us(s[w, base + i]) for i in range(4))
                if int(s[w, base + 1]) == 0 or int(s[w, base + 2]) == 0:
                    continue           # a part that never ran its loop (a tail left to the head's owner)
                key = (p, kind, "chained" if p > 0 else "first")
                r = rows.setdefault(key, {"gap": [], "wait": [], "slice": [], "after": [], "len": [], "start": []})
                r["start"].append(st)
                r["gap"].append(st - prev_end)
                r["wait"].append(pro - st)
                r["slice"].append((loop - pro) / max(ln, 1))
                r["after"].append(end - loop)
                r["len"].append(ln)
                prev_end = end
    assert L.mmh_ab_set_stamps5(mm._h, None) == 0
    assert L.mmh_ab_set_stamp_stride5(mm._h, 4) == 0
    print(f"   {G} workgroups; entry {stats(entries)} us, exit (stores drained) {stats(totals)} us   (min/median/max)")
    print("   part  kind   | slices | start us | to first slice us | us per slice | loop end -> part end us")
    for key in sorted(rows):
        r = rows[key]
        print(f"   {key[0]}     {key[1]:6s} | {stats(r['len'])} | {stats(r['start'])} | {stats(r['wait'])} | "
              f"{stats(r['slice'])} | {stats(r['after'])}   ({len(r['len'])} samples)")
mm.close()
