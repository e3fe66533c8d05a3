#!/usr/bin/env python3
"""Copy the round-2 evidence (gpurun_out/r02f/, written by tools/r02_final.sh and tools/r02_sweeps.sh on
the GPU box) into the tracked profiles/ directory under r02_ names, and refresh profiles/pmc_traffic.json
(what bench.py quotes as roofline.traffic) from the PMC passes."""
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "r02f")
DST = os.path.join(REPO, "profiles")


def cp(src, dst):
    s = os.path.join(SRC, src)
    if os.path.exists(s):
        shutil.copy(s, os.path.join(DST, dst))
        print("profiles/" + dst)
    else:
        print("missing:", src)


for k in ("auto", "auto_ref_convention", "auto_extended", "rocblas", "valu", "mfma", "auto_vs_blas", "auto_splitk"):
    cp(f"output_MMult_hip_{k}.m", f"r02_output_MMult_hip_{k}.m")
cp("clock_ramp.csv", "r02_clock_ramp.csv")
cp("bench.json", "r02_bench_line.json")
cp("bench_forceshard.json", "r02_bench_forceshard_line.json")
cp("sweep_vs_vendor.md", "r02_sweep_vs_vendor.md")
cp("hbm_patterns.txt", "r02_hbm_patterns.txt")
cp("host_flavour.txt", "r02_host_flavour.txt")
cp("prof4096_kernel_stats.csv", "r02_sgemm4096_kernel_stats.csv")
cp("prof4096_summary.json", "r02_sgemm4096_auto_rocprofv3.json")
cp("prof3072_summary.json", "r02_sgemm3072_dma64x64_rocprofv3.json")
cp("cold_start.txt", "r02_cold_start.txt")
cp("prof2048_summary.json", "r02_sgemm2048_dma128x64_rocprofv3.json")
cp("prof1024_summary.json", "r02_sgemm1024_dma64x64_rocprofv3.json")
cp(os.path.join("qprof", "summary.json"), "r02_qgemm_rocprofv3.json")
cp(os.path.join("i8prof", "summary.json"), "r02_igemm_s8_rocprofv3.json")

# roofline.traffic: FETCH_SIZE (KiB, x2 on gfx950 for 16 B/lane coalesced reads) + WRITE_SIZE (KiB)
traffic = {}
for n, name in ((4096, "prof4096_summary.json"), (3072, "prof3072_summary.json"),
                (2048, "prof2048_summary.json"), (1024, "prof1024_summary.json")):
    p = os.path.join(SRC, name)
    if not os.path.exists(p):
        continue
    d = json.load(open(p))
    pm = d.get("pmc_mean_per_dispatch", {})
    try:
        fetch, write = pm["pmc3"]["FETCH_SIZE"], pm["pmc4"]["WRITE_SIZE"]
        hit, miss = pm["pmc4"]["TCC_HIT_sum"], pm["pmc4"]["TCC_MISS_sum"]
    except KeyError:
        continue
    traffic[str(n)] = {
        "kernel": d["kernel_stats"][0]["name"],
        "FETCH_SIZE_KB_raw": fetch, "WRITE_SIZE_KB_raw": write,
        "correction": "FETCH_SIZE x2 on gfx950 for 16 B/lane coalesced reads (MI355X_MICROARCH.md, HBM)",
        "hbm_bytes_per_launch": int(round(fetch * 1024 * 2 + write * 1024)),
        "algorithmic_bytes_per_launch": 3 * 4 * n * n,
        "l2_hit_rate": round(hit / (hit + miss), 4),
        "avg_us": d["kernel_stats"][0]["avg_us"], "dispatches_in_trace": d["kernel_stats"][0]["calls"],
        "round": 2,
    }
old = json.load(open(os.path.join(DST, "pmc_traffic.json")))
for k, v in old.items():
    if k not in traffic:
        traffic[k] = v
json.dump(traffic, open(os.path.join(DST, "pmc_traffic.json"), "w"), indent=1)
print("profiles/pmc_traffic.json", {k: v["hbm_bytes_per_launch"] for k, v in traffic.items()})
