#!/usr/bin/env python3
"""Differential fuzz on one GPU: every kernel variant against the naive kernel (an
independent code path with the same chain semantics -> bit-equal) on random shapes,
leading dimensions, 4-byte-misaligned bases and accumulate flags; plus a stream-K
stress loop (ragged tile counts, repeated launches) against the one-tile-per-workgroup
kernel.  usage: python tools/fuzz.py [cases] [stress_reps] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import how_to_optimize_gemm_amd as H  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
stress = int(sys.argv[2]) if len(sys.argv) > 2 else 30
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rng = np.random.default_rng(seed)
mm = H.MMult(0)
stream = torch.cuda.current_stream().cuda_stream
VARIANTS = ["auto", "mfma", "mfma256", "mfma_256x256", "mfma_128x64", "mfma_64x64", "mfma_pipe", "mfma_simple", "valu",
            "valu_64x64", "valu_128x128", "mfma_64x64_dma", "mfma_128x64_dma", "mfma_128x128_dma",
            "mfma_64x64_dma5", "mfma_128x64_dma5", "mfma_128x128_dma5", "mfma_96x96_dma5",          # K2W (round 4)
            "mfma_64x64_dma5/sk2", "mfma_128x64_dma5/sk2", "mfma_128x128_dma5/sk2",                 # ... stream-K whenever ragged
            "mfma_96x64_dma5", "valu_128x64",                                                     # round 5: the 96x64 K2W tile, K1W's third tile
            "mfma_64x64_dma/sk2", "mfma_128x64_dma/sk2",                                          # ... K2L under stream-K (AUTO's candidates now)
            "valu_64x64/sk2", "valu_128x64/sk2", "valu_128x128/sk2",                             # round 6: K1W under K2W's stream-K body
            "mfma_160x160_dma5"]                                                                  # round 6: the 160x160 K2W tile (every K2W tile: fragment reads spread)


def strided(rows, cols, ld, off, fill=None):
    flat = torch.full((rows * ld + off + 8,), float("nan"), device="cuda")
    view = flat[off:off + rows * ld].view(rows, ld)
    if fill is not None:
        view[:, :cols] = fill
    return flat, view


bad = 0
for case in range(cases):
    kind = rng.integers(0, 7)
    aligned = kind == 4
    if kind == 4:      # whole tiles, 16-byte aligned bases and leading dimensions: what the LDS-DMA tiles take
        m, n = (int(rng.integers(1, 12)) * 128 for _ in range(2))
        k = int(rng.integers(1, 24)) * 64
    elif kind == 0:    # tile multiples
        m, n, k = (int(rng.integers(1, 9)) * 128 for _ in range(3))
    elif kind == 1:    # ragged small
        m, n, k = (int(rng.integers(1, 400)) for _ in range(3))
    elif kind == 2:    # ragged around tile edges
        m, n, k = (int(rng.integers(1, 6)) * 128 + int(rng.integers(-3, 4)) for _ in range(3))
    elif kind == 5:    # a few rows / columns past a 64-boundary: the K2W kernels' thin edge tiles
        m, n = (int(rng.integers(1, 20)) * 64 + int(rng.integers(0, 18)) for _ in range(2))
        k = int(rng.integers(1, 900))
    elif kind == 6:    # enough tiles for multi-part stream-K ranges, ragged
        m, n = (int(rng.integers(1100, 2600)) for _ in range(2))
        k = int(rng.integers(33, 700))
    else:              # thin
        m, n, k = int(rng.integers(1, 40)), int(rng.integers(1, 2000)), int(rng.integers(1, 1500))
    lda, ldb, ldc = k + int(rng.integers(0, 9)), n + int(rng.integers(0, 9)), n + int(rng.integers(0, 9))
    offs = [int(rng.integers(0, 4)) for _ in range(3)]
    if aligned:
        lda, ldb, ldc = k + 4 * int(rng.integers(0, 3)), n + 4 * int(rng.integers(0, 3)), n + 4 * int(rng.integers(0, 3))
        offs = [4 * int(rng.integers(0, 2)) for _ in range(3)]
    acc = bool(rng.integers(0, 2))
    a = torch.rand((m, k), device="cuda") * 2 - 1
    b = torch.rand((k, n), device="cuda") * 2 - 1
    c0 = torch.rand((m, n), device="cuda")
    _, av = strided(m, k, lda, offs[0], a)
    _, bv = strided(k, n, ldb, offs[1], b)
    results = {}
    for kern in ["naive"] + VARIANTS:
        mm.set_kernel(kern.split("/")[0])
        mm.set_streamk(2 if kern.endswith("/sk2") else 1)
        cflat, cv = strided(m, n, ldc, offs[2], c0)
        mm.sgemm(m, n, k, av.data_ptr(), lda, bv.data_ptr(), ldb, cv.data_ptr(), ldc, acc, stream)
        torch.cuda.synchronize()
        results[kern] = cv[:, :n].clone()
        pad_ok = bool(torch.isnan(cv[:, n:]).all()) and bool(torch.isnan(cflat[:offs[2]]).all())
        if not pad_ok:
            bad += 1
            print(f"case {case} {kern}: wrote outside C window  m,n,k={m},{n},{k} ld={lda},{ldb},{ldc}")
    for kern in VARIANTS:
        if not torch.equal(results[kern], results["naive"]):
            bad += 1
            d = (results[kern] - results["naive"]).abs().max().item()
            print(f"case {case} {kern}: != naive (max diff {d})  m,n,k={m},{n},{k} ld={lda},{ldb},{ldc} "
                  f"off={offs} acc={acc}")
mm.set_streamk(1)
print(f"fuzz: {cases} cases x {len(VARIANTS)} variants, {bad} failures")

# stream-K stress
sk_bad = 0
for n in (1152, 1536, 1792, 2176, 2432, 2944, 3072, 3456, 3712, 4352, 4608, 2049, 2305, 3001):   # the last three: guarded stream-K
    a = torch.rand((n, n), device="cuda") * 2 - 1
    b = torch.rand((n, n), device="cuda") * 2 - 1
    mm.set_kernel("mfma_tiles")
    ref = mm.matmul(a, b)
    # "auto": the LDS-DMA tiles under stream-K below 4096, the 256x256 tile above; "mfma": the register-staged 128x128 tile
    for kern in ("auto", "mfma", "mfma_128x128_dma5/sk2", "mfma_64x64_dma5/sk2", "valu_128x128/sk2", "valu_64x64/sk2", "valu"):
        mm.set_kernel(kern.split("/")[0])
        mm.set_streamk(2 if kern.endswith("/sk2") else 1)
        c = torch.empty_like(ref)
        for rep in range(stress):
            c.fill_(float("nan"))
            mm.matmul(a, b, out=c)
            if not torch.equal(c, ref):
                sk_bad += 1
                print(f"stream-K {kern} N={n} rep {rep}: mismatch, max diff {(c - ref).abs().max().item()}")
        if mm.streamk_timeouts():
            sk_bad += 1
            print(f"stream-K {kern} N={n}: hand-off timeouts reported")
mm.set_streamk(1)
print(f"stream-K stress: {sk_bad} failures")
sys.exit(1 if bad or sk_bad else 0)
