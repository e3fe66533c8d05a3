#!/usr/bin/env python3
"""tile_sweep.py -- every tile family forced, plain and under stream-K, over a list of shapes: the table AUTO's
rules are read off (round 4; the reference's `NEW := MMult_xxx` switch, cuda/makefile:1-3, tried at every value).

    python tools/tile_sweep.py [--sizes 1024:4096:128 | --shapes m,n,k;m,n,k] [--variants a,b/sk0,c/sk2,...]
                               [--out gpurun_out/tile_sweep] [--check]

A variant is a kernel's short name (mmh_kernel_id) optionally followed by /sk0 (MMH_OPT_STREAMK = 0: one
workgroup per tile), /sk1 (the library's own policy, the default) or /sk2 (stream-K whenever the tile count is
ragged), with --ab also /gN, /nd, /oo, /omNN, /np and /k1old (tools build switches: raster group height, no deferred publish, a whole-tile
stream-K launch bounded by its own instantiation's residency, phase-ordered tables from NN/10 tiles per workgroup, no LDS pin of
persistent launches, the register-staged K1 of rounds 1-4 instead of K1W), /p1 (MMH_OPT_PERSIST = 1: whole rounds of the persistent grid run persistent too), and /nc (MMH_OPT_STREAMK_CHAIN = 0: the K2M tiles' stream-K parts unchained); `rocblas` / `hipblaslt` are the vendor comparators.  Protocol as tools/offgrid_sweep.py: every burst
through the C ABI after ~--warm-ms of untimed launches of its own variant, --rounds interleaved rounds, medians.
--check compares every variant's C with the first variant's, bit for bit.  Needs a GPU."""
from __future__ import annotations

import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

DEFAULT = ("auto,mfma_64x64_dma/sk0,mfma_64x64_dma/sk2,mfma32_64x64_dma/sk0,mfma32_64x64_dma/sk2,"
           "mfma_128x64_dma/sk0,mfma_128x64_dma/sk2,mfma32_128x64_dma/sk0,mfma32_128x64_dma/sk2,"
           "mfma32_64x128_dma/sk0,mfma32_64x128_dma/sk2,"
           "mfma_128x128_dma/sk0,mfma_128x128_dma/sk2,mfma32_128x128_dma/sk0,mfma32_128x128_dma/sk2,mfma_256x256")
VENDORS = ["rocblas", "hipblaslt"]


def parse_shapes(args):
    if args.shape_file:
        return [tuple(int(x) for x in line.split(",")) for line in open(args.shape_file) if line.strip()]
    if args.shapes:
        return [tuple(int(x) for x in s.split(",")) for s in args.shapes.split(";") if s]
    lo, hi, step = (int(x) for x in args.sizes.split(":"))
    return [(n, n, n) for n in range(lo, hi + 1, step)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="1024:4096:128")
    ap.add_argument("--shapes", default="")
    ap.add_argument("--shape-file", default="", help="one m,n,k per line (tools/policy_shapes.py)")
    ap.add_argument("--variants", default=DEFAULT)
    ap.add_argument("--out", default=os.path.join(REPO, "gpurun_out", "tile_sweep"))
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--warm-ms", type=float, default=20.0)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--ab", action="store_true", help="load libmmult_hip_ab.so (the exp5_* / ablation ids)")
    args = ap.parse_args()
    import torch
    import how_to_optimize_gemm_amd as H
    if args.ab:
        H.use_ab_library()
    mm = H.MMult(0, "auto")
    stream = torch.cuda.current_stream().cuda_stream
    variants = args.variants.split(",")
    shapes = parse_shapes(args)
    need_max = max(m * k + k * n + m * n for (m, n, k) in shapes)
    big = torch.rand((max(1 << 28, need_max),), device="cuda") * 2 - 1
    rows = []

    def select(v):
        parts = v.split("/")
        mm.set_kernel(parts[0])
        sk = [x for x in parts[1:] if x.startswith("sk")]
        mm.set_streamk(int(sk[0][2:]) if sk else 1)
        mm.set_option(H.OPT_STREAMK_CHAIN, 0 if "nc" in parts[1:] else 1)
        mm.set_option(H.OPT_PERSIST, 1 if "p1" in parts[1:] else 0)
        mm.set_option(H.OPT_STREAMK_ORDER, 0 if "no" in parts[1:] else 1)   # /no: stream-K ranges in chip order, no phase tables
        if args.ab:      # tools build: /gN raster group height, /nd publish stream-K heads on the spot
            gm = [x for x in parts[1:] if x.startswith("g") and x[1:].isdigit()]
            mm.set_option(101, int(gm[0][1:]) if gm else 0)
            mm.set_option(102, 1 if "nd" in parts[1:] else 0)
            mm.set_option(103, 1 if "oo" in parts[1:] else 0)   # whole-tile stream-K launches bounded by their own residency
            mm.set_option(100, 0 if "np" in parts[1:] else 1)   # /np: persistent launches ask for their own LDS only (no 160 KiB / w pin)
            mm.set_option(105, 1 if "k1old" in parts[1:] else 0)   # /k1old: the register-staged K1 of rounds 1-4
            mm.set_option(106, 1 if "wr" in parts[1:] else 0)      # /wr: persistent launches with whole-tile ranges (round 6)
            mm.set_option(107, 0 if "ns" in parts[1:] else 1)     # /ns: no tail split of plain K2W launches (the last round in the same launch: rounds 4-6)
            om = [x for x in parts[1:] if x.startswith("om") and x[2:].isdigit()]   # /omNN: phase-ordered tables from NN/10 tiles per workgroup
            mm.set_option(104, int(om[0][2:]) if om else 18)

    for (m, n, k) in shapes:
        need = m * k + k * n + m * n
        a = big[:m * k].view(m, k)
        b = big[m * k:m * k + k * n].view(k, n)
        c = big[m * k + k * n:need].view(m, n)
        pa, pb, pc = a.data_ptr(), b.data_ptr(), c.data_ptr()
        flops = 2.0 * m * n * k

        def burst(v, reps, warm):
            if v in VENDORS:
                return mm.time_comparator(v, m, n, k, pa, k, pb, n, pc, n, warmup=warm, reps=reps, stream=stream)
            select(v)
            return mm.time_sgemm(m, n, k, pa, k, pb, n, pc, n, warmup=warm, reps=reps, stream=stream)

        res, warm, launched, equal = {}, {}, {}, {}
        ref = None
        for v in variants:
            try:
                ms = burst(v, 3, 1)
                if v not in VENDORS:
                    launched[v] = H.last_launch()
                res[v] = []
                warm[v] = max(3, int(args.warm_ms / max(ms, 1e-3)))
                if args.check and v not in VENDORS:
                    torch.cuda.synchronize()
                    if ref is None:
                        ref = c.clone()
                    else:
                        equal[v] = bool(torch.equal(ref, c))
            except H.MMultError as e:
                res[v] = None
                launched[v] = f"error: {e}"[:120]
        for _ in range(args.rounds):
            for v in variants:
                if res.get(v) is None:
                    continue
                ms = burst(v, args.reps, warm[v])
                res[v].append(flops / (ms * 1e-3) / 1e12)
        row = {"m": m, "n": n, "k": k}
        for v in variants:
            xs = sorted(res[v]) if res.get(v) else None
            row[v] = round(xs[len(xs) // 2], 1) if xs else None
        row["launched"] = launched
        if args.check:
            row["bit_equal_to_first"] = equal
        rows.append(row)
        print(json.dumps({kk: vv for kk, vv in row.items() if kk != "launched"}), flush=True)
    mm.set_streamk(1)
    mm.set_option(H.OPT_PERSIST, 0)
    mm.close()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(rows, open(args.out + ".json", "w"), indent=1)
    with open(args.out + ".md", "w") as f:
        f.write("| m x n x k | " + " | ".join(variants) + " | best |\n")
        f.write("|---|" + "---|" * len(variants) + "---|\n")
        for r in rows:
            ours = {v: r[v] for v in variants if r.get(v) and v not in VENDORS}
            best = max(ours, key=ours.get) if ours else "-"
            f.write(f"| {r['m']} x {r['n']} x {r['k']} | " + " | ".join(str(r.get(v)) for v in variants) + f" | {best} |\n")
    print("wrote", args.out + ".md")


if __name__ == "__main__":
    main()
