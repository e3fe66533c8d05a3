#!/usr/bin/env python3
"""Copy the round-3 evidence (gpurun_out/r03z/ written by tools/r03_final.sh on the GPU box, plus the named pieces of
the earlier round-3 calls) into the tracked profiles/ directory under r03_ names, and refresh
profiles/pmc_traffic.json (what bench.py quotes as roofline.traffic when the live passes are off) from the PMC passes."""
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "gpurun_out")
SRC = os.path.join(OUT, sys.argv[1] if len(sys.argv) > 1 else "r03z")
DST = os.path.join(REPO, "profiles")


def cp(src, dst, base=SRC):
    s = os.path.join(base, src)
    if os.path.exists(s) and os.path.getsize(s) > 0:
        shutil.copy(s, os.path.join(DST, dst))
        print("profiles/" + dst)
    else:
        print("missing:", src)


for k in ("auto", "auto_ref_convention", "rocblas", "hipblaslt", "valu", "mfma", "auto_vs_blas"):
    cp(f"output_MMult_hip_{k}.m", f"r03_output_MMult_hip_{k}.m")
cp("sweep_auto_launches.json", "r03_sweep_auto_launches.json")
cp("clock_ramp.csv", "r03_clock_ramp.csv")
cp("bench.json", "r03_bench_line.json")
cp("bench_noramp.json", "r03_bench_line_no_ramp.json")
cp("bench_forceshard.json", "r03_bench_forceshard_sweep.json")
cp("cold_start.txt", "r03_cold_start.txt")
cp("offgrid.md", "r03_offgrid_vs_vendor.md")
cp("offgrid.json", "r03_offgrid_vs_vendor.json")
cp("shard_dryrun.md", "r03_shard_dryrun.md")
cp("harness_sharded_1gpu.txt", "r03_harness_sharded_shared_device.txt")
cp("prof4096_kernel_stats.csv", "r03_sgemm4096_kernel_stats.csv")
cp("prof4096_summary.json", "r03_sgemm4096_auto_dma128x64_rocprofv3.json")
cp("prof4096_64_summary.json", "r03_sgemm4096_dma64x64_rocprofv3.json")
cp("prof4096_mfma128_summary.json", "r03_sgemm4096_mfma128x128_rocprofv3.json")
cp("prof4096_256_summary.json", "r03_sgemm4096_mfma256x256_rocprofv3.json")
cp("prof3584_summary.json", "r03_sgemm3584_dma_streamk128x64_rocprofv3.json")
cp("prof1023_summary.json", "r03_sgemm1023_dma64x64_guarded_rocprofv3.json")
cp("vendor_kernels.md", "r03_vendor_kernels.md")
cp(os.path.join("i8prof", "summary.json"), "r03_igemm_s8_rocprofv3.json")
cp(os.path.join("qprof", "summary.json"), "r03_qgemm_rocprofv3.json")
cp("i8_ksweep.txt", "r03_igemm_s8_ksweep.txt")
cp("i8_ab.txt", "r03_igemm_s8_ab.txt")
# pieces of the other calls of the round that the notes cite
cp(os.path.join("r03o", "probe_align.txt"), "r03_lds_dma_align_probe.txt", OUT)
cp(os.path.join("r03o", "ab_r02.txt"), "r03_streamk_ab_vs_r02.txt", OUT)
cp(os.path.join("r03o", "i8_ld_probe.txt"), "r03_igemm_s8_ld_probe.txt", OUT)
cp(os.path.join("r03n", "rim_ab.md"), "r03_rim_ab.md", OUT)
cp(os.path.join("r03k", "harness_2944_context.txt"), "r03_harness_2944_context.txt", OUT)
cp(os.path.join("r03p", "eight_sweeps.txt"), "r03_eight_sweeps_ref_skip.txt", OUT)
cp(os.path.join("r03s", "prof2304_summary.json"), "r03_sgemm2304_dma_streamk128x128_rocprofv3.json", OUT)
cp(os.path.join("r03s", "prof1152_summary.json"), "r03_sgemm1152_dma_streamk64x64_rocprofv3.json", OUT)
cp(os.path.join("r03s", "prof1024_summary.json"), "r03_sgemm1024_dma64x64_rocprofv3.json", OUT)
cp(os.path.join("r03s", "prof4096_valu_summary.json"), "r03_sgemm4096_valu128x128_rocprofv3.json", OUT)

# roofline.traffic: FETCH_SIZE (KiB, x2 on gfx950 for 16 B/lane coalesced reads) + WRITE_SIZE (KiB)
traffic = {}
for n, name in ((4096, "prof4096_summary.json"), (3584, "prof3584_summary.json"), (1023, "prof1023_summary.json")):
    p = os.path.join(SRC, name)
    if not os.path.exists(p):
        continue
    try:
        d = json.load(open(p))
        pm = d.get("pmc_mean_per_dispatch", {})
        fetch, write = pm["pmc3"]["FETCH_SIZE"], pm["pmc4"]["WRITE_SIZE"]
        hit, miss = pm["pmc4"]["TCC_HIT_sum"], pm["pmc4"]["TCC_MISS_sum"]
    except (KeyError, ValueError):
        continue
    traffic[str(n)] = {
        "kernel": d["kernel_stats"][0]["name"],
        "FETCH_SIZE_KB_raw": fetch, "WRITE_SIZE_KB_raw": write,
        "correction": "FETCH_SIZE x2 on gfx950 for 16 B/lane coalesced reads (MI355X_MICROARCH.md, HBM)",
        "hbm_bytes_per_launch": int(round(fetch * 1024 * 2 + write * 1024)),
        "algorithmic_bytes_per_launch": 3 * 4 * n * n,
        "l2_hit_rate": round(hit / (hit + miss), 4),
        "avg_us": d["kernel_stats"][0]["avg_us"], "dispatches_in_trace": d["kernel_stats"][0]["calls"],
        "round": 3,
    }
old = json.load(open(os.path.join(DST, "pmc_traffic.json")))
for k, v in old.items():
    if k not in traffic:
        traffic[k] = v
json.dump(traffic, open(os.path.join(DST, "pmc_traffic.json"), "w"), indent=1)
print("profiles/pmc_traffic.json", {k: v["hbm_bytes_per_launch"] for k, v in traffic.items()})
