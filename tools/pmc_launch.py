#!/usr/bin/env python3
"""pmc_launch.py -- the launch loop a `rocprofv3 --pmc ...` pass wraps when several launch forms of one shape are to be
compared by their counters (round 4: fabric traffic of the headline tile, plain against persistent launches).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o pmc -- python tools/pmc_launch.py --n 4096 \
        --variants mfma_128x64_dma,mfma_128x64_dma/p1,mfma_128x64_dma5,mfma_128x64_dma5/p1 [--ab]

A variant is a kernel's short name with tools/tile_sweep.py's suffixes (/sk0 /sk2 /p1 /nc) and, with --ab, /gN (raster
group height N of the plain K2W launch).  Every variant is launched
--warm times untimed and --reps times; the kernel names differ between the forms, so tools/pmc_by_kernel.py can tell
them apart in the counter CSV.  Needs a GPU."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--shape", default="")
    ap.add_argument("--variants", default="auto")
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--warm", type=int, default=30)
    ap.add_argument("--ab", action="store_true")
    ap.add_argument("--pad", type=int, default=0, help="leading dimensions = the row length + this many floats")
    args = ap.parse_args()
    import torch
    import how_to_optimize_gemm_amd as H
    if args.ab:
        H.use_ab_library()
    m, n, k = (int(x) for x in args.shape.split(",")) if args.shape else (args.n,) * 3
    mm = H.MMult(0, "auto")
    lda, ldb, ldc = k + args.pad, n + args.pad, n + args.pad
    a = torch.rand((m, lda), device="cuda") * 2 - 1
    b = torch.rand((k, ldb), device="cuda") * 2 - 1
    c = torch.empty((m, ldc), device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for v in args.variants.split(","):
        parts = v.split("/")
        mm.set_kernel(parts[0])
        sk = [x for x in parts[1:] if x.startswith("sk")]
        mm.set_streamk(int(sk[0][2:]) if sk else 1)
        mm.set_option(H.OPT_STREAMK_CHAIN, 0 if "nc" in parts[1:] else 1)
        mm.set_option(H.OPT_PERSIST, 1 if "p1" in parts[1:] else 0)
        mm.set_option(H.OPT_STREAMK_ORDER, 0 if "no" in parts[1:] else 1)   # /no: no phase-ordered stream-K tables
        gm = [x for x in parts[1:] if x.startswith("g") and x[1:].isdigit()]
        if args.ab:
            mm.set_option(101, int(gm[0][1:]) if gm else 0)   # tools build: raster group height of the K2W launches
            mm.set_option(102, 1 if "nd" in parts[1:] else 0)  # ... stream-K heads publish on the spot
            mm.set_option(103, 1 if "oo" in parts[1:] else 0)  # ... whole-tile stream-K grids by their own residency
            mm.set_option(105, 1 if "k1old" in parts[1:] else 0)   # /k1old: the register-staged K1 of rounds 1-4
            om = [x for x in parts[1:] if x.startswith("om") and x[2:].isdigit()]   # /omNN: phase-ordered tables from NN/10 tiles per workgroup
            mm.set_option(104, int(om[0][2:]) if om else 18)
        for _ in range(args.warm + args.reps):
            mm.sgemm(m, n, k, a.data_ptr(), lda, b.data_ptr(), ldb, c.data_ptr(), ldc, False, s)
        torch.cuda.synchronize()
        print(v, "->", H.last_launch(), flush=True)
    mm.close()


if __name__ == "__main__":
    main()
