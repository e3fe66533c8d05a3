#!/usr/bin/env python3
"""Stream-K (MMH_KERNEL_MFMA on ragged tile counts) vs one-workgroup-per-tile
(MMH_KERNEL_MFMA_TILES): bit-equality and timing."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import how_to_optimize_gemm_amd as H  # noqa: E402

mm = H.MMult(0, "mfma")
stream = torch.cuda.current_stream().cuda_stream
a = torch.rand((4096, 4096), device="cuda")
c = torch.empty_like(a)
for _ in range(150):
    mm.matmul(a, a, out=c)
torch.cuda.synchronize()
sizes = [int(x) for x in sys.argv[1:]] or list(range(2048, 4097, 128))
for n in sizes:
    a = torch.rand((n, n), device="cuda") * 2 - 1
    b = torch.rand((n, n), device="cuda") * 2 - 1
    out = {}
    tf = {}
    for kern in ("mfma_tiles", "mfma"):
        mm.set_kernel(kern)
        c = torch.full((n, n), float("nan"), device="cuda")
        mm.matmul(a, b, out=c)
        torch.cuda.synchronize()
        out[kern] = c
        ms = mm.time_sgemm(n, n, n, a.data_ptr(), n, b.data_ptr(), n, c.data_ptr(), n, warmup=3, reps=10,
                           stream=stream)
        tf[kern] = 2.0 * n ** 3 / (ms * 1e-3) / 1e12
    # accumulate mode through stream-K as well
    mm.set_kernel("mfma")
    c0 = torch.rand((n, n), device="cuda")
    c1 = c0.clone()
    mm.matmul(a, b, out=c1, accumulate=True)
    mm.set_kernel("mfma_tiles")
    c2 = c0.clone()
    mm.matmul(a, b, out=c2, accumulate=True)
    torch.cuda.synchronize()
    print(f"N={n}: tiles {tf['mfma_tiles']:7.1f}  stream-K {tf['mfma']:7.1f} TFLOP/s  "
          f"equal={torch.equal(out['mfma'], out['mfma_tiles'])} acc_equal={torch.equal(c1, c2)}", flush=True)
