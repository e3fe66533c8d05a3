#!/usr/bin/env python3
"""Where does a short LDS-DMA launch spend its time?  Each workgroup of the plain K2L kernels (a build of its
own, libmmult_hip_tl.so) stamps the wall clock at entry, after its prologue, after its K loop and after its C stores; this
prints the distribution of each phase over the workgroups of ONE launch, next to the launch's
back-to-back hipEvent time.
usage: python tools/dma_timeline.py [--kernel mfma_64x64_dma] [--n 1024] [--launches 40]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import how_to_optimize_gemm_amd as H  # noqa: E402

H.use_timeline_library()
ap = argparse.ArgumentParser()
ap.add_argument("--kernel", default="mfma_64x64_dma")
ap.add_argument("--n", type=int, nargs="+", default=[1024])
ap.add_argument("--launches", type=int, default=40)
args = ap.parse_args()
L = H.lib()
L.mmh_ab_set_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
mm = H.MMult(0, args.kernel)
mm.set_streamk(0)
try:
    import ctypes.util
    hip = ctypes.CDLL("libamdhip64.so")
    v = ctypes.c_int(0)
    hip.hipDeviceGetAttribute(ctypes.byref(v), 10017, 0)   # hipDeviceAttributeWallClockRate (kHz)
    wall_khz = v.value or 100000
except Exception:
    wall_khz = 100000
stream = torch.cuda.current_stream().cuda_stream
print(f"wall clock {wall_khz} kHz; kernel {args.kernel}")
for n in args.n:
    a = torch.rand((n, n), device="cuda") * 2 - 1
    b = torch.rand((n, n), device="cuda") * 2 - 1
    c = torch.empty((n, n), device="cuda")
    for _ in range(300):
        mm.matmul(a, b, out=c)
    torch.cuda.synchronize()
    ms = mm.time_sgemm(n, n, n, a.data_ptr(), n, b.data_ptr(), n, c.data_ptr(), n, warmup=20, reps=200, stream=stream)
    launch = H.last_launch() if hasattr(H, "last_launch") else ""
    stamps = torch.zeros((1 << 16, 4), device="cuda", dtype=torch.int64)
    assert L.mmh_ab_set_stamps(mm._h, stamps.data_ptr()) == 0
    rows = []
    for it in range(args.launches):
        stamps.zero_()
        for _ in range(5):
            mm.matmul(a, b, out=c)      # the stamped launch is the last of a burst of five
        torch.cuda.synchronize()
        s = stamps.cpu()
        s = s[s[:, 0] > 0].double()
        t0 = s[:, 0].min()
        us = (s - t0) * 1e3 / wall_khz
        rows.append(torch.stack([us[:, 0].max(), (us[:, 1] - us[:, 0]).median(), (us[:, 2] - us[:, 1]).median(),
                                 (us[:, 2] - us[:, 1]).max(), (us[:, 3] - us[:, 2]).median(), us[:, 2].max(),
                                 us[:, 3].max(), torch.tensor(float(len(s)))]))
    assert L.mmh_ab_set_stamps(mm._h, None) == 0
    r = torch.stack(rows).median(dim=0).values
    flops = 2.0 * n ** 3
    print(f"N={n}: back-to-back {ms * 1e3:.1f} us/launch = {flops / ms / 1e9:.1f} TFLOP/s; {int(r[7])} workgroups; "
          f"MFMA time of the whole problem at peak {flops / 157.3e12 * 1e6:.1f} us")
    print(f"   last workgroup enters at +{r[0]:.2f} us | prologue median {r[1]:.2f} | K loop median {r[2]:.2f} max {r[3]:.2f}"
          f" | C store median {r[4]:.2f} | last loop end +{r[5]:.2f} | last store done +{r[6]:.2f} us")
