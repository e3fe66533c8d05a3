#!/usr/bin/env python3
"""Copy the round-4 evidence (gpurun_out/r04z/ written by tools/r04_final.sh on the GPU box, plus the named pieces of
the earlier round-4 calls under gpurun_out/r04*/) into the tracked profiles/ directory under r04_ names, refresh
profiles/pmc_traffic.json (what bench.py quotes as roofline.traffic when the live passes are off) from the PMC passes,
and draw profiles/r04_sweep.png (tools/plot_sweep.py).  No GPU."""
import json
import os
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "gpurun_out")
SRC = os.path.join(OUT, sys.argv[1] if len(sys.argv) > 1 else "r04z")
DST = os.path.join(REPO, "profiles")


def cp(src, dst, base=SRC):
    s = os.path.join(base, src)
    if os.path.exists(s) and os.path.getsize(s) > 0:
        shutil.copy(s, os.path.join(DST, dst))
        print("profiles/" + dst)
    else:
        print("missing:", src)


for k in ("auto", "auto_ref_convention", "rocblas", "hipblaslt", "valu", "mfma", "auto_vs_blas", "auto_refskip", "rocblas_refskip",
          "hipblaslt_refskip", "auto_nonsquare", "rocblas_nonsquare", "hipblaslt_nonsquare"):
    cp(f"output_MMult_hip_{k}.m", f"r04_output_MMult_hip_{k}.m")
cp("sweep_auto_launches.json", "r04_sweep_auto_launches.json")
cp("clock_ramp.csv", "r04_clock_ramp.csv")
cp("bench.json", "r04_bench_line.json")
cp("bench_noramp.json", "r04_bench_line_no_ramp.json")
cp("bench_forceshard.json", "r04_bench_forceshard_sweep.json")
cp("cold_start.txt", "r04_cold_start.txt")
cp("offgrid.md", "r04_offgrid_vs_vendor.md")
cp("offgrid.json", "r04_offgrid_vs_vendor.json")
cp("shard_dryrun.md", "r04_shard_dryrun.md")
cp("harness_sharded_1gpu.txt", "r04_harness_sharded_shared_device.txt")
cp("harness_sharded_rccl1.txt", "r04_harness_sharded_one_rank_rccl.txt")
cp("prof4096_kernel_stats.csv", "r04_sgemm4096_kernel_stats.csv")
cp("prof4096_summary.json", "r04_sgemm4096_auto_k2w128x64_rocprofv3.json")
cp("prof2560_summary.json", "r04_sgemm2560_k2w_streamk128x128_rocprofv3.json")
cp("prof1152_summary.json", "r04_sgemm1152_k2w_streamk64x64_rocprofv3.json")
cp("prof1024_summary.json", "r04_sgemm1024_k2w64x64_rocprofv3.json")
cp("prof1536_summary.json", "r04_sgemm1536_k2w96x96_rocprofv3.json")
cp("i8_instr_ab.md", "r04_i8_instr_ab.md")
cp("create_time.json", "r04_create_time.json")
cp("pytest_gpu.log", "r04_pytest_gpu.log")
# pieces of the other calls of the round that the notes cite
for src, dst in (("r04/exp1_small.md", "r04_k2w_variants_small.md"), ("r04/exp1_mid.md", "r04_k2w_variants_mid.md"),
                 ("r04/exp1_big.md", "r04_persistent_vs_plain.md"), ("r04/exp1_pmc.json", "r04_persistent_traffic.json"),
                 ("r04/exp2_small.md", "r04_loader_count_small.md"), ("r04/exp2_mid.md", "r04_loader_count_mid.md"),
                 ("r04/exp2_pmc.json", "r04_raster_group_traffic.json"), ("r04/exp2_pmcpad.json", "r04_raster_group_traffic_padded_rows.json"),
                 ("r04/ab3_mid.md", "r04_deferred_publish_ab_mid.md"), ("r04/ab3_small.md", "r04_deferred_publish_ab_small.md"),
                 ("r04/edge2.md", "r04_thin_tiles_edge.md"), ("r04/tl5_mfma_64x64_dma5.txt", "r04_rim_timeline.txt")):
    cp(src, dst, OUT)

# roofline.traffic: FETCH_SIZE (KiB, x2 on gfx950 for 16 B/lane coalesced reads) + WRITE_SIZE (KiB)
traffic = {}
for n, name in ((4096, "prof4096_summary.json"), (2560, "prof2560_summary.json"), (1152, "prof1152_summary.json")):
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
        "round": 4,
    }
old = json.load(open(os.path.join(DST, "pmc_traffic.json")))
for k, v in old.items():
    if k not in traffic:
        traffic[k] = v
json.dump(traffic, open(os.path.join(DST, "pmc_traffic.json"), "w"), indent=1)
print("profiles/pmc_traffic.json", {k: v["hbm_bytes_per_launch"] for k, v in traffic.items()})

# the round's plot: auto (sustained and under the reference's convention), both vendor libraries, the VALU rung, the peak line
files = [os.path.join(DST, f"r04_output_MMult_hip_{k}.m") for k in ("auto", "auto_ref_convention", "rocblas", "hipblaslt", "valu", "mfma")]
files = [f for f in files if os.path.exists(f)]
if files:
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "plot_sweep.py"), *files, "-o", os.path.join(DST, "r04_sweep.png")],
                       capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr.strip()[-300:])
