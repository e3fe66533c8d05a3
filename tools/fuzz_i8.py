#!/usr/bin/env python3
"""Differential fuzz of the int8 kernels on one GPU: every MMH_OPT_IGEMM_MODE (0 auto, 5 / 6 B read in
place, 7 / 8 / 9 the ping-pong kernel; FUZZ_I8_AB=1: the tools build's 1 in-kernel transpose and 3 / 4 packed B too) against mode 2 (the correctness-first kernel, an
independent code path; integers -> bit-equal; MMH_I8_GRID_CAP=3 in the environment makes the persistent ping-pong
kernel walk several tiles per workgroup on these small shapes) on random shapes, leading dimensions, byte-misaligned
bases and accumulate flags, with guard bands around C.  usage: python tools/fuzz_i8.py [cases] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import how_to_optimize_gemm_amd as H  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
MODES = [0, 5, 6, 7, 8, 9]   # (7: K3p on the config-named 16x16x32 instruction; 8 / 9: persistent / one workgroup per tile)
if os.environ.get("FUZZ_I8_AB") == "1":    # the tools build: K3 (1) and the packed-B kernel (3 / 4) too
    H.use_ab_library()
    MODES += [1, 3, 4]
mm = H.MMult(0)
GUARD = -2139062144          # 0x80808080: never a valid sum here


def strided(rows, cols, ld, off, dtype, fill, guard):
    flat = torch.full((rows * ld + off + 64,), guard, device="cuda", dtype=dtype)
    view = flat[off:off + rows * ld].view(rows, ld)
    view[:, :cols] = fill
    return flat, view


bad = 0
for case in range(cases):
    kind = rng.integers(0, 4)
    if kind == 0:      # whole tiles, odd and even slice counts
        m, n = (int(rng.integers(1, 5)) * 256 for _ in range(2))
        k = int(rng.integers(1, 9)) * 128
    elif kind == 1:    # ragged small
        m, n, k = (int(rng.integers(1, 500)) for _ in range(3))
    elif kind == 2:    # ragged around tile and slice edges
        m, n = (int(rng.integers(1, 4)) * 256 + int(rng.integers(-3, 4)) for _ in range(2))
        k = int(rng.integers(1, 6)) * 128 + int(rng.integers(-3, 4))
    else:              # thin / deep
        m, n, k = int(rng.integers(1, 40)), int(rng.integers(1, 1500)), int(rng.integers(1, 3000))
    # leading dimensions: mostly dword multiples (the fast paths), sometimes not
    pad4 = lambda x: (x + 3) & ~3
    lda = (pad4(k) + 4 * int(rng.integers(0, 5))) if rng.integers(0, 4) else k + int(rng.integers(0, 7))
    ldb = (pad4(n) + 4 * int(rng.integers(0, 5))) if rng.integers(0, 4) else n + int(rng.integers(0, 7))
    ldc = n + int(rng.integers(0, 9))
    offs = [4 * int(rng.integers(0, 4)) if rng.integers(0, 4) else int(rng.integers(0, 16)) for _ in range(2)]
    offc = int(rng.integers(0, 4))
    acc = bool(rng.integers(0, 2))
    a = torch.from_numpy(rng.integers(-127, 128, (m, k), dtype=np.int8)).cuda()
    b = torch.from_numpy(rng.integers(-127, 128, (k, n), dtype=np.int8)).cuda()
    c0 = torch.from_numpy(rng.integers(-100000, 100000, (m, n), dtype=np.int32)).cuda()
    _, av = strided(m, k, lda, offs[0], torch.int8, a, 77)
    _, bv = strided(k, n, ldb, offs[1], torch.int8, b, -55)
    results = {}
    for mode in [2] + MODES:
        mm.set_igemm_mode(mode)
        cflat, cv = strided(m, n, ldc, offc, torch.int32, c0, GUARD)
        mm.igemm_s8(av[:, :k], bv[:, :n], out=cv[:, :n], accumulate=acc)
        torch.cuda.synchronize()
        results[mode] = cv[:, :n].clone()
        pad_ok = bool((cv[:, n:] == GUARD).all()) and bool((cflat[:offc] == GUARD).all()) and \
            bool((cflat[offc + m * ldc:] == GUARD).all())
        if not pad_ok:
            bad += 1
            print(f"case {case} mode {mode}: wrote outside C window  m,n,k={m},{n},{k} ld={lda},{ldb},{ldc}")
    want = c0.double() * (1 if acc else 0) + a.double() @ b.double()
    if not torch.equal(results[2].double(), want):
        bad += 1
        print(f"case {case}: the correctness-first kernel itself != fp64 reference  m,n,k={m},{n},{k}")
    for mode in MODES:
        if not torch.equal(results[mode], results[2]):
            bad += 1
            nz = int((results[mode] != results[2]).sum())
            print(f"case {case} mode {mode}: != mode 2 ({nz} elements)  m,n,k={m},{n},{k} ld={lda},{ldb},{ldc} "
                  f"off={offs} acc={acc}")
mm.set_igemm_mode(0)
print(f"int8 fuzz: {cases} cases x {len(MODES)} modes, {bad} failures")
sys.exit(1 if bad else 0)
