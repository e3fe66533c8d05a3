#!/usr/bin/env python3
"""Plot output_*.m sweep files (the reference's result format, written by
`make -C how-to-optimize-gemm_amd/harness run`) with the MI355X fp32 MFMA roofline
as the top line -- the role of cuda/plot.py:30-40 and PlotAll.m's max_gflops axis.

    python tools/plot_sweep.py profiles/r01_output_MMult_hip_*.m -o profiles/r01_sweep.png

File format (cuda/makefile:43 + cuda/test_MMult.cpp:32-33,41,128,144): a `version = '...';`
line, the device line, a blank line, `MY_MMult = [`, rows `p gflops diff [extra...]`, `];`.
"""
import argparse
import re

PEAK_GFLOPS = 157300.0


def read_m(path):
    title, sizes, gflops = path, [], []
    with open(path) as f:
        for line in f:
            m = re.match(r"version = '(.*)';", line)
            if m:
                title = m.group(1)
                continue
            tok = line.split()
            if len(tok) >= 3 and tok[0].isdigit():
                sizes.append(int(tok[0]))
                gflops.append(float(tok[1]))
    return title, sizes, gflops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="+")
    ap.add_argument("-o", "--out", default="sweep.png")
    args = ap.parse_args()
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    fig, ax = plt.subplots(figsize=(9, 5.5))
    for path in args.files:
        title, x, y = read_m(path)
        ax.plot(x, y, marker="o", markersize=3, label=title)
    ax.axhline(PEAK_GFLOPS, color="k", linestyle="--", linewidth=1, label="fp32 MFMA peak 157.3 TFLOP/s")
    ax.axhline(0.8 * PEAK_GFLOPS, color="gray", linestyle=":", linewidth=1, label="80 % of peak")
    ax.set_xlabel("m = n = k")
    ax.set_ylabel("GFLOPS")
    ax.set_ylim(0, PEAK_GFLOPS * 1.05)
    ax.set_title("square SGEMM sweep on one MI355X (row-major fp32)")
    ax.legend(loc="lower right", fontsize=8)
    ax.grid(alpha=0.3)
    fig.tight_layout()
    fig.savefig(args.out, dpi=110)
    print(args.out)


if __name__ == "__main__":
    main()
