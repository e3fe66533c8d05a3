#!/usr/bin/env python3
"""Forced tiles against each other on whole-round shapes (interleaved rounds in one process, median of 3 bursts of 30
after 150 untimed launches each): which tile should AUTO give shapes whose 128x64 tile count is whole rounds of CUs?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import how_to_optimize_gemm_amd as H  # noqa: E402

mm = H.MMult(0, "auto")
shapes = [(4096, 4096, 4096), (6144, 6144, 6144), (8192, 8192, 8192), (4096, 8192, 4096), (8192, 4096, 2048), (2048, 4096, 8192), (4096, 4096, 1024)]
kernels = sys.argv[1].split(",") if len(sys.argv) > 1 else ["auto", "mfma_64x64_dma", "mfma_128x64_dma", "mfma_256x256"]
print("| m x n x k | " + " | ".join(kernels) + " |")
print("|---|" + "---|" * len(kernels))
for (m, n, k) in shapes:
    a = torch.rand((m, k), device="cuda") * 2 - 1
    b = torch.rand((k, n), device="cuda") * 2 - 1
    c = torch.empty((m, n), device="cuda")
    res = {kk: [] for kk in kernels}
    for rnd in range(3):
        for kk in kernels:
            mm.set_kernel(kk)
            ms = mm.time_sgemm(m, n, k, a.data_ptr(), k, b.data_ptr(), n, c.data_ptr(), n, warmup=150 if m * n * k < 2e11 else 40, reps=30)
            res[kk].append(2.0 * m * n * k / (ms * 1e-3) / 1e12)
    print(f"| {m} x {n} x {k} | " + " | ".join(f"{sorted(v)[1]:.1f}" for v in res.values()) + " |", flush=True)
mm.close()
