#!/usr/bin/env python3
"""Assorted timings on one GPU: int8 GEMM, guarded (EDGE) shapes, sweep sizes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import how_to_optimize_gemm_amd as H  # noqa: E402

if os.environ.get("MMH_AB", "1") != "0":   # the A/B variants and ablation builds live in the tools-only library
    H.use_ab_library()

mm = H.MMult(0, "mfma")
stream = torch.cuda.current_stream().cuda_stream


def ramp():
    a = torch.rand((4096, 4096), device="cuda")
    c = torch.empty_like(a)
    for _ in range(150):
        mm.matmul(a, a, out=c)
    torch.cuda.synchronize()


def time_f32(m, n, k, kernel="mfma", reps=20):
    if kernel == "rocblas":
        return time_rocblas(m, n, k, reps)
    if kernel == "torch.mm":
        return time_torch_mm(m, n, k, reps)
    H.lib().mmh_set_kernel(mm._h, H.KERNELS[kernel] if kernel in H.KERNELS else int(kernel))
    a = torch.rand((m, k), device="cuda") * 2 - 1
    b = torch.rand((k, n), device="cuda") * 2 - 1
    c = torch.empty((m, n), device="cuda")
    # ~40 ms of the same launch first: the clock a measurement sees otherwise depends on what ran
    # before it (a slow kernel leaves the next one a lower clock for tens of launches)
    est = mm.time_sgemm(m, n, k, a.data_ptr(), k, b.data_ptr(), n, c.data_ptr(), n, warmup=1, reps=3, stream=stream)
    ms = mm.time_sgemm(m, n, k, a.data_ptr(), k, b.data_ptr(), n, c.data_ptr(), n,
                       warmup=max(5, min(400, int(40.0 / max(est, 1e-3)))), reps=reps, stream=stream)
    return 2.0 * m * n * k / (ms * 1e-3) / 1e12


def time_torch_mm(m, n, k, reps=20):
    """hipBLASLt as PyTorch calls it (fp32, TF32 off): the second vendor comparator."""
    torch.backends.cuda.matmul.allow_tf32 = False
    a = torch.rand((m, k), device="cuda") * 2 - 1
    b = torch.rand((k, n), device="cuda") * 2 - 1
    c = torch.empty((m, n), device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        torch.mm(a, b, out=c)
    e1.record()
    torch.cuda.synchronize()
    for _ in range(max(5, min(400, int(40.0 / max(e0.elapsed_time(e1) / 3, 1e-3))))):
        torch.mm(a, b, out=c)
    e0.record()
    for _ in range(reps):
        torch.mm(a, b, out=c)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * m * n * k / (e0.elapsed_time(e1) / reps * 1e-3) / 1e12


def time_rocblas(m, n, k, reps=20):
    a = torch.rand((m, k), device="cuda") * 2 - 1
    b = torch.rand((k, n), device="cuda") * 2 - 1
    c = torch.empty((m, n), device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        mm.matmul_rocblas(a, b, out=c)
    e1.record()
    torch.cuda.synchronize()
    for _ in range(max(5, min(400, int(40.0 / max(e0.elapsed_time(e1) / 3, 1e-3))))):   # same ~40 ms ramp
        mm.matmul_rocblas(a, b, out=c)
    e0.record()
    for _ in range(reps):
        mm.matmul_rocblas(a, b, out=c)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * m * n * k / (e0.elapsed_time(e1) / reps * 1e-3) / 1e12


def time_i8(m, n, k, reps=20):
    a = torch.randint(-127, 128, (m, k), device="cuda", dtype=torch.int8)
    b = torch.randint(-127, 128, (k, n), device="cuda", dtype=torch.int8)
    c = torch.empty((m, n), device="cuda", dtype=torch.int32)
    for _ in range(5):
        mm.igemm_s8(a, b, out=c)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        mm.igemm_s8(a, b, out=c)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * m * n * k / (e0.elapsed_time(e1) / reps * 1e-3) / 1e12


ramp()
what = sys.argv[1:] or ["i8", "edge", "sweep"]
if "i8" in what:
    print(f"int8 MFMA-only probe: {mm.probe_mfma_i8():8.1f} TOPS (constant operands, 2.3 ms)")
    for rnd in (False, True):
        r = [mm.probe_mfma_i8_sustained(rnd, ms) for ms in (0.0, 20.0, 100.0, 300.0)]
        print(f"int8 MFMA-only sustained, {'random' if rnd else 'constant'} operands, after 0/20/100/300 ms: " +
              " ".join(f"{x:7.1f}" for x in r) + " TOPS")
    modes = [int(x) for x in os.environ.get("I8_MODES", "3,4,0").split(",")]
    for n in [int(x) for x in os.environ.get("I8_SIZES", "2048,4096,8192").split(",")]:
        r = []
        for mode in modes:
            mm.set_igemm_mode(mode)
            r.append(f"mode{mode} {time_i8(n, n, n):7.1f}")
        try:    # vendor comparator: hipBLASLt through torch._int_mm
            a8 = torch.randint(-127, 128, (n, n), device="cuda", dtype=torch.int8)
            b8 = torch.randint(-127, 128, (n, n), device="cuda", dtype=torch.int8)
            for _ in range(5):
                torch._int_mm(a8, b8)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                torch._int_mm(a8, b8)
            e1.record()
            torch.cuda.synchronize()
            r.append(f"hipBLASLt {2.0 * n ** 3 / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e12:7.1f}")
        except Exception:
            pass
        print(f"int8 N={n}: " + "  ".join(r) + " TOPS (incl. packing B)")
    mm.set_igemm_mode(0)
if "quant" in what:
    # the callers either side of the int8 GEMM: streaming passes, GB/s against 8 TB/s
    def ev(fn, reps=20):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3      # us
    for n in (2048, 4096, 8192):
        x = torch.rand((n, n), device="cuda") * 2 - 1
        y = torch.rand((n, n), device="cuda") * 2 - 1
        o = torch.empty((n, n), device="cuda")
        tq = ev(lambda: mm.quantize_sym_s8(x))
        tg = ev(lambda: mm.qgemm(x, y, out=o))
        # absmax reads 4 B/elt, quantise reads 4 + writes 1
        print(f"N={n}: quantize_sym_s8 {tq:8.1f} us = {9.0 * n * n / tq / 1e3:7.1f} GB/s   "
              f"qgemm {tg:8.1f} us = {2.0 * n ** 3 / tg / 1e6:7.1f} TOPS-equivalent")
if "edge" in what:
    print("| m x n x k | auto | rocblas | torch.mm |")
    print("|---|---|---|---|")
    for (m, n, k) in [(4096, 4096, 4096), (4000, 4000, 4000), (5000, 5000, 5000), (4097, 4095, 4099), (4096, 4096, 4100),
                      (3000, 5000, 2000), (6000, 4100, 3000),
                      (1000, 1000, 1000), (2049, 2049, 2049), (16384, 2048, 16384), (2048, 16384, 16384), (8192, 8192, 512),
                      (512, 512, 16384)]:
        print(f"| {m} x {n} x {k} | " + " | ".join(f"{time_f32(m, n, k, kk, reps=10):.1f}" for kk in ("auto", "rocblas", "torch.mm")) + " |",
              flush=True)
if "sweep" in what:
    for n in range(1024, 4097, 256):
        print(f"fp32 N={n}: mfma {time_f32(n, n, n):7.1f}  mfma256 {time_f32(n, n, n, 'mfma256'):7.1f}  "
              f"valu {time_f32(n, n, n, 'valu'):7.1f} TFLOP/s")
if "tiles" in what:
    ks = ["valu", "mfma_tiles", "mfma", "mfma_128x64", "mfma_64x64", "mfma_256x256", "auto", "rocblas", "torch.mm"]
    print("| N | " + " | ".join(ks) + " |")
    print("|---|" + "---|" * len(ks))
    for n in list(range(1024, 4097, 128)) + [4608, 5120, 6144, 8192]:
        print(f"| {n} | " + " | ".join(f"{time_f32(n, n, n, k, reps=10):.1f}" for k in ks) + " |", flush=True)
if "abl64" in what:
    ks = ["mfma_128x64", "36", "37", "38", "39", "40", "rocblas"]
    print("N      " + "  ".join(f"{k:>11}" for k in ks))
    for n in (1024, 1152, 1280, 1408, 1536, 1792, 2048):
        print(f"{n:5d}  " + "  ".join(f"{time_f32(n, n, n, k, reps=10):11.1f}" for k in ks), flush=True)
if "small" in what:
    ks = ["mfma", "mfma_128x64", "mfma_64x64", "rocblas"]
    print("N      " + "  ".join(f"{k:>11}" for k in ks))
    for n in list(range(512, 1024, 128)) + list(range(1024, 2945, 128)):
        print(f"{n:5d}  " + "  ".join(f"{time_f32(n, n, n, k, reps=10):11.1f}" for k in ks), flush=True)
if "mid" in what:
    ks = ["mfma", "mfma_tiles", "rocblas"]
    print("N      " + "  ".join(f"{k:>11}" for k in ks))
    for n in range(1920, 3201, 128):
        print(f"{n:5d}  " + "  ".join(f"{time_f32(n, n, n, k, reps=10):11.1f}" for k in ks), flush=True)
if "big" in what:
    for n in (8192, 16384):
        print(f"fp32 N={n}: mfma {time_f32(n, n, n, reps=5):7.1f}  mfma256 {time_f32(n, n, n, 'mfma256', reps=5):7.1f}")
