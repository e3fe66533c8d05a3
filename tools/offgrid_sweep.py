#!/usr/bin/env python3
"""offgrid_sweep.py -- AUTO against rocBLAS and hipBLASLt (and against this library's own tiles, forced) on
shapes OFF the reference's 128-grid (the reference parameterises exactly
these: armv7/parameters.h:15-17,38-46 -- M, N, K, LDA, LDB, LDC -- and admits it skips boundaries,
README.md:80,93).

    python tools/offgrid_sweep.py [--set steps|pm1|ld|nonsquare|all] [--out profiles/r03_offgrid] [--quick]

Protocol: every variant is timed through the C ABI (mmh_time_sgemm / mmh_time_comparator: one event pair
around `reps` back-to-back calls issued from C), each burst after ~`--warm-ms` of untimed launches of its own
kernel; `--rounds` interleaved rounds, the median is reported.  Writes <out>.md (the table the judge asked
for) and <out>.json (every number).  Needs a GPU."""
from __future__ import annotations

import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

SWEEP = list(range(1024, 4097, 128))
OURS = ["auto", "mfma_64x64_dma", "mfma_128x64_dma", "mfma_128x128_dma", "mfma_256x256", "mfma"]
VENDORS = ["rocblas", "hipblaslt"]


def shapes(which: str, quick: bool):
    out = []
    if which in ("steps", "all"):
        for n in range(1000, 4101, 400 if quick else 100):
            out.append(("steps", n, n, n, n, n, n))
    if which in ("pm1", "all"):
        for n in (SWEEP[::6] if quick else SWEEP):
            for d in (-1, 0, 1):
                out.append(("pm1", n + d, n + d, n + d, n + d, n + d, n + d))
    if which in ("ld", "all"):
        for n in ((1024, 2048, 4096) if quick else (1024, 1536, 2048, 2560, 3072, 3584, 4096)):
            out.append(("ld+4", n, n, n, n + 4, n + 4, n + 4))
            out.append(("ld+1", n, n, n, n + 1, n + 1, n + 1))
    if which in ("nonsquare", "all"):
        for (m, n, k) in [(8192, 1024, 4096), (1024, 8192, 512), (16384, 128, 4096), (300, 5000, 7000), (4096, 4096, 4100),
                          (2049, 2049, 2049), (6000, 3000, 1000), (1000, 6000, 3000), (128, 16384, 4096), (5000, 5000, 5000),
                          (3000, 4000, 8192), (12288, 512, 2048)]:
            out.append(("nonsquare", m, n, k, k, n, n))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--set", default="all")
    ap.add_argument("--out", default=os.path.join(REPO, "gpurun_out", "offgrid"))
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--warm-ms", type=float, default=25.0)
    ap.add_argument("--variants", default=",".join(OURS + VENDORS))
    ap.add_argument("--vendor-harness", action="store_true",
                    help="take the rocblas / hipblaslt columns from harness/test_MMult.x, one C++ process per shape and library: "
                         "the image's ROCm libraries.  In THIS process (torch imported) dlopen finds the copies inside torch's "
                         "wheel -- an older hipBLASLt, 15-20 %% slower on its stream-K sizes; those figures are then kept "
                         "under <lib>_torch_bundled (round 6)")
    args = ap.parse_args()
    import torch
    import how_to_optimize_gemm_amd as H
    mm = H.MMult(0, "auto")
    stream = torch.cuda.current_stream().cuda_stream
    variants = args.variants.split(",")
    big = torch.rand((1 << 28,), device="cuda") * 2 - 1            # 1 GiB of operands to cut views from
    rows = []
    for (tag, m, n, k, lda, ldb, ldc) in shapes(args.set, args.quick):
        need = m * lda + k * ldb + m * ldc
        if need > big.numel():
            continue
        a = big[:m * lda].view(m, lda)
        b = big[m * lda:m * lda + k * ldb].view(k, ldb)
        c = big[m * lda + k * ldb:need].view(m, ldc)
        pa, pb, pc = a.data_ptr(), b.data_ptr(), c.data_ptr()
        flops = 2.0 * m * n * k

        def burst(v, reps, warm):
            if v in VENDORS:
                return mm.time_comparator(v, m, n, k, pa, lda, pb, ldb, pc, ldc, warmup=warm, reps=reps, stream=stream)
            mm.set_kernel(v)
            return mm.time_sgemm(m, n, k, pa, lda, pb, ldb, pc, ldc, warmup=warm, reps=reps, stream=stream)

        res, launched = {}, {}
        for v in variants:
            try:
                ms = burst(v, 3, 1)                                  # first contact: plans, tables
                if v not in VENDORS:
                    launched[v] = H.last_launch()
                res[v] = []
                est = max(ms, 1e-3)
                res[v + "_warm"] = max(3, int(args.warm_ms / est))
            except H.MMultError as e:
                res[v] = None
                launched[v] = f"error: {e}"[:120]
        for _ in range(args.rounds):
            for v in variants:
                if res.get(v) is None:
                    continue
                ms = burst(v, args.reps, res[v + "_warm"])
                res[v].append(flops / (ms * 1e-3) / 1e12)
        row = {"set": tag, "m": m, "n": n, "k": k, "lda": lda, "ldb": ldb, "ldc": ldc, "launched": launched.get("auto", "")}
        for v in variants:
            xs = sorted(res[v]) if res.get(v) else None
            row[v] = round(xs[len(xs) // 2], 1) if xs else None
        if args.vendor_harness:
            import subprocess
            exe = os.path.join(REPO, "how-to-optimize-gemm_amd", "harness", "test_MMult.x")
            torch.cuda.synchronize()
            for v in [x for x in variants if x in VENDORS]:
                row[v + "_torch_bundled"] = row[v]
                row[v] = None
                env = {**os.environ, "KERNEL": v, "REF": "skip", "WARMUP_MS": "50", "TRIALS": "3", "M": str(m), "N": str(n), "K": str(k),
                       "LDA": str(lda), "LDB": str(ldb), "LDC": str(ldc), "PFIRST": "1", "PLAST": "1", "PINC": "1"}
                try:
                    r = subprocess.run([exe], cwd=os.path.dirname(exe), env=env, capture_output=True, text=True, timeout=120)
                    for line in r.stdout.splitlines():
                        f = line.split()
                        if len(f) == 3 and f[0] == "1":
                            row[v] = round(float(f[1]) / 1e3, 1)
                except Exception:
                    pass
        rows.append(row)
        print(json.dumps(row), flush=True)
    mm.close()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(rows, open(args.out + ".json", "w"), indent=1)
    with open(args.out + ".md", "w") as f:
        f.write("| set | m x n x k (lda, ldb, ldc) | " + " | ".join(variants) + " | auto / max(vendor) | auto / best own | AUTO launched |\n")
        f.write("|---|---|" + "---|" * len(variants) + "---|---|---|\n")
        for r in rows:
            vend = [r[v] for v in VENDORS if r.get(v)]
            own = [r[v] for v in OURS if r.get(v)]
            ratio = f"{r['auto'] / max(vend):.3f}" if vend and r.get("auto") else "-"
            ratio_own = f"{r['auto'] / max(own):.3f}" if own and r.get("auto") else "-"
            ld = "" if (r["lda"], r["ldb"], r["ldc"]) == (r["k"], r["n"], r["n"]) else f" ({r['lda']}, {r['ldb']}, {r['ldc']})"
            f.write(f"| {r['set']} | {r['m']} x {r['n']} x {r['k']}{ld} | " + " | ".join(str(r.get(v)) for v in variants) +
                    f" | {ratio} | {ratio_own} | {r['launched'][:90]} |\n")
    print("wrote", args.out + ".md")


if __name__ == "__main__":
    main()
