#!/usr/bin/env python3
"""i8_timeline.py -- per-workgroup, per-tile wall-clock stamps of ONE launch of the persistent int8 ping-pong kernel K3p
(libmmult_hip_tl.so): where a workgroup's time goes at a tile boundary.  Stamps of wave 0 (older group) and wave 4
(younger group), kept in scalar registers and written behind the tile's last wait:
  0 top of the tile (the previous tile's C stores have just been issued)   1 prologue landed, barrier passed
  2..5 after the tile's first four phases (slice 0 step 0 / 1, slice 1 step 0 / 1)   6 K loop done   7 both groups in
  step, every request and store of this wave acknowledged
usage: python tools/i8_timeline.py [MxNxK ...]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import how_to_optimize_gemm_amd as H  # noqa: E402

H.use_timeline_library()
L = H.lib()
L.mmh_ab_set_stamps_i8.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
mm = H.MMult(0, "auto")
hip = ctypes.CDLL("libamdhip64.so")
v = ctypes.c_int(0)
hip.hipDeviceGetAttribute(ctypes.byref(v), 10017, 0)   # hipDeviceAttributeWallClockRate (kHz)
tick_us = 1e3 / (v.value or 100000)
shapes = [tuple(int(x) for x in s.split("x")) for s in sys.argv[1:]] or [(8192, 8192, 8192), (8192, 8192, 2048)]
mode = int(os.environ.get("I8_MODE", "8"))


def med(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2] if xs else float("nan")


for (m, n, k) in shapes:
    g = torch.Generator(device="cuda").manual_seed(3)
    a = torch.randint(-127, 128, (m, k), device="cuda", dtype=torch.int8, generator=g)
    b = torch.randint(-127, 128, (k, n), device="cuda", dtype=torch.int8, generator=g)
    c = torch.empty((m, n), device="cuda", dtype=torch.int32)
    mm.set_igemm_mode(mode)
    for _ in range(100):
        mm.igemm_s8(a, b, out=c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        mm.igemm_s8(a, b, out=c)
    e1.record()
    torch.cuda.synchronize()
    print(f"{m}x{n}x{k} mode {mode}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch (stamps off)")
    wgs = 256
    stamps = torch.zeros((wgs, 2, 16, 8), device="cuda", dtype=torch.int64)
    assert L.mmh_ab_set_stamps_i8(mm._h, stamps.data_ptr()) == 0
    for _ in range(30):
        mm.igemm_s8(a, b, out=c)
    torch.cuda.synchronize()
    stamps.zero_()
    e0.record()
    mm.igemm_s8(a, b, out=c)
    e1.record()
    torch.cuda.synchronize()
    print(f"   stamped launch: {e0.elapsed_time(e1) * 1e3:.1f} us")
    assert L.mmh_ab_set_stamps_i8(mm._h, None) == 0
    s = stamps.cpu().numpy().astype("float64") * tick_us
    t0 = s[s[:, :, 0, 0] > 0][:, 0, 0].min() if (s[:, :, 0, 0] > 0).any() else 0.0
    ntile = int((s[0, 0, :, 7] > 0).sum())
    print(f"   {ntile} tiles per workgroup; medians over workgroups, us (older group | younger group)")
    print("   tile  start-since-first  prologue-wait  ph1  ph2  ph3  ph4  rest-of-loop  resync+drain   tile-total")
    for t in range(ntile):
        cols = []
        for grp in range(2):
            x = s[:, grp, t, :]
            ok = x[:, 7] > 0
            x = x[ok]
            d = [med(x[:, 0] - t0)] + [med(x[:, i + 1] - x[:, i]) for i in range(7)] + [med(x[:, 7] - x[:, 0])]
            cols.append(d)
        print("   %2d  " % t + "  ".join(f"{a_:7.2f}|{b_:7.2f}" for a_, b_ in zip(cols[0], cols[1])))
    # spread of the tile boundaries over the chip
    for t in range(ntile):
        x = s[:, 0, t, 6]
        x = x[x > 0]
        print(f"   tile {t}: loop end min/med/max since first start {x.min() - t0:.1f} / {med(x) - t0:.1f} / {x.max() - t0:.1f} us")
