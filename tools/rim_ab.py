#!/usr/bin/env python3
"""The rim on / off (MMH_OPT_RIM = 8 / 0) through AUTO at shapes a few elements past a 64-boundary, beside the
on-grid neighbour: TFLOP/s, median of 3 bursts of 50 after 200 untimed launches."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import how_to_optimize_gemm_amd as H

AB = "--ab" in sys.argv      # the A/B library: also times the rim alone and the trimmed tiles alone (MMH_AB_RIM)
if AB:
    sys.argv.remove("--ab")
    H.use_ab_library()
mm = H.MMult(0, "auto")


def rate(n, rim, m=None, k=None):
    m = m or n
    k = k or n
    a = torch.rand((m, k), device="cuda") * 2 - 1
    b = torch.rand((k, n), device="cuda") * 2 - 1
    c = torch.empty((m, n), device="cuda")
    mm.set_option(H.OPT_RIM, rim)
    ms = sorted(mm.time_sgemm(m, n, k, a.data_ptr(), k, b.data_ptr(), n, c.data_ptr(), n, warmup=200, reps=50) for _ in range(3))[1]
    return 2.0 * m * n * k / (ms * 1e-3) / 1e12, ms * 1e3, H.last_launch()


sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1024, 1408, 2048, 2688, 3072, 4096]
extras = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 8]
print("| N | on-grid TF (us) | +e | rim on TF (us) | ratio | rim off TF (us) | launched with the rim on |" + (" rim alone us | tiles alone us |" if AB else ""))
print("|---|---|---|---|---|---|---|" + ("---|---|" if AB else ""))
for n in sizes:
    base, us0, _ = rate(n, 8)
    for e in extras:
        on, us1, what = rate(n + e, 8)
        off, us2, _ = rate(n + e, 0)
        extra = ""
        if AB:
            os.environ["MMH_AB_RIM"] = "only"
            _, us3, _ = rate(n + e, 8)
            os.environ["MMH_AB_RIM"] = "none"
            _, us4, _ = rate(n + e, 8)
            del os.environ["MMH_AB_RIM"]
            extra = f" {us3:.0f} | {us4:.0f} |"
        print(f"| {n} | {base:.1f} ({us0:.0f}) | {e} | {on:.1f} ({us1:.0f}) | {on / base:.2f} | {off:.1f} ({us2:.0f}) | {what[:34]}{' ..rim' if 'rim' in what else ''} |" + extra, flush=True)
mm.close()
