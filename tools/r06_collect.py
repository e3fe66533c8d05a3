#!/usr/bin/env python3
"""Copy the round-6 evidence (gpurun_out/r06z/ written by tools/r06_final.sh on the GPU box, plus the named pieces of
the earlier round-6 calls under gpurun_out/r05*/) into the tracked profiles/ directory under r06_ names, refresh
profiles/pmc_traffic.json (what bench.py quotes as roofline.traffic when the live passes are off) from the PMC passes,
and draw profiles/r06_sweep.png (tools/plot_sweep.py).  No GPU."""
import json
import os
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "gpurun_out")
SRC = os.path.join(OUT, sys.argv[1] if len(sys.argv) > 1 else "r06z")
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
    cp(f"output_MMult_hip_{k}.m", f"r06_output_MMult_hip_{k}.m")
cp("sweep_auto_launches.json", "r06_sweep_auto_launches.json")
cp("clock_ramp.csv", "r06_clock_ramp.csv")
cp("bench.json", "r06_bench_line.json")
cp("bench_noramp.json", "r06_bench_line_no_ramp.json")
cp("bench_forceshard.json", "r06_bench_forceshard_sweep.json")
cp("cold_start.txt", "r06_cold_start.txt")
cp("offgrid.md", "r06_offgrid_vs_vendor.md")
cp("offgrid.json", "r06_offgrid_vs_vendor.json")
cp("shard_dryrun.md", "r06_shard_dryrun.md")
cp("harness_sharded_1gpu.txt", "r06_harness_sharded_shared_device.txt")
cp("harness_sharded_rccl1.txt", "r06_harness_sharded_one_rank_rccl.txt")
cp("prof4096_kernel_stats.csv", "r06_sgemm4096_kernel_stats.csv")
cp("prof4096_summary.json", "r06_sgemm4096_auto_k2w128x64_rocprofv3.json")
cp("prof2560_summary.json", "r06_sgemm2560_k2w160x160_rocprofv3.json")     # (second session: the 160x160 tile; the first session's stream-K profile keeps its name)
cp("prof3584_summary.json", "r06_sgemm3584_k2w_streamk128x128_rocprofv3.json")
cp("prof1152_summary.json", "r06_sgemm1152_k2w96x64_rocprofv3.json")
cp("prof1024_summary.json", "r06_sgemm1024_k2w64x64_rocprofv3.json")
cp("prof1536_summary.json", "r06_sgemm1536_k2w96x96_rocprofv3.json")
cp("create_time.json", "r06_create_time.json")
cp("pytest_gpu.log", "r06_pytest_gpu.log")
cp("prof_valu1024_summary.json", "r06_sgemm1024_k1w_valu64x64_rocprofv3.json")
# the held-out shapes measured once more with the final table in the library
held = os.path.join(SRC, "dataset_heldout.json")
if os.path.exists(held):
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import numpy as np
    import policy_fit as P
    table = P.fit(P.rows_of(json.load(open(os.path.join(DST, "r06_policy_dataset_fit.json")))))
    rows = json.load(open(held))
    same, ratio, reg_auto = 0, [], []
    for r in rows:
        (fam, form), _ = P.choose(table, r["m"], r["n"], r["k"])
        la = r["launched"]["auto"]
        ok = P.family_of_auto(r) == (fam, form)
        same += ok
        forced = P.measured(r, fam, form)
        if ok and forced:
            ratio.append(r["auto"] / forced)
        best = max(v for v in (P.measured(r, f, fo) for f in P.FAMILIES for fo in ("plain", "sk")) if v)
        reg_auto.append(1.0 - r["auto"] / best)
    rg = np.array([x["regret"] for x in P.regret(table, rows)])
    worst = sorted(P.regret(table, rows), key=lambda x: -x["regret"])[:10]
    with open(os.path.join(DST, "r06_auto_regret_confirmation_pass.md"), "w") as f:
        f.write("# The 500 held-out shapes measured again with the FINAL table in the library (tools/r06_final.sh, part `regret`)\n\n"
                "The table was fitted on profiles/r06_policy_dataset_fit.json; this pass (another gpurun call, another box)\n"
                "took no part in it.\n\n"
                f"* MMH_KERNEL_AUTO launched the tile family and launch form tools/policy_fit.py::choose names on **{same} of {len(rows)}** shapes\n"
                "  (the launch strings of the `auto` column against the Python evaluation of the committed table).\n"
                f"* Regret of those choices against the best FORCED candidate measured in this pass: **mean {rg.mean() * 100:.2f} %, "
                f"p90 {np.percentile(rg, 90) * 100:.2f} %, max {rg.max() * 100:.2f} %**.\n"
                f"* The `auto` column itself reads lower than the same kernel forced a few bursts later in the same rotation (median "
                f"{np.median(ratio):.3f}, mean {np.mean(ratio):.3f} of it): `auto` is the first variant measured after every change of shape, and the\n"
                "  dataset protocol warms each variant for 10 ms only -- taken at face value that column gives mean "
                f"{np.mean(reg_auto) * 100:.2f} %, max {max(reg_auto) * 100:.2f} %; it measures the rotation, not the choice.\n\n"
                "Worst ten choices:\n\n```\n" +
                "\n".join(f"{x['shape']}: chose {x['chosen']} {x['tf']} TF, best {x['best_is'][0]}/{x['best_is'][1]} {x['best']} ({x['regret'] * 100:.1f} %)" for x in worst) +
                "\n```\n")
    print("profiles/r06_auto_regret_confirmation_pass.md")
# pieces of the other calls of the round that the notes cite (copied by hand as they were taken: profiles/r06_own_residency_ab.md,
# r06_streamk_l2_phase_order.md, r06_phase_order_and_96x64_ab.md, r06_k1w_variants.md, r06_k1w_sweep.md, r06_valu_probe.json)
cp("harness_sharded_rccl1_streamed.txt", "r06_harness_sharded_one_rank_rccl_streamed.txt")
# round 6: configs[4] and configs[1]
cp("i8_persist_ab.txt", "r06_i8_persist_ab_final_pass.txt")
cp("i8_timeline.txt", "r06_i8_timeline_final_pass.txt")
cp("i8prof_summary.json", "r06_igemm_s8_rocprofv3.json")
cp("i8_instr_ab.md", "r06_i8_instr_ab.md")
cp("fuzz.txt", "r06_fuzz.txt")
cp("prof_valu2176_summary.json", "r06_sgemm2176_k1w_valu_streamk128x128_rocprofv3.json")
cp("valu_probe.json", "r06_valu_probe.json")
cp("sweep_valu_launches.json", "r06_sweep_valu_launches.json")
cp("prof_valu2048_summary.json", "r06_sgemm2048_k1w_valu128x128_rocprofv3.json")

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
        "round": 6,
    }
old = json.load(open(os.path.join(DST, "pmc_traffic.json")))
for k, v in old.items():
    if k not in traffic:
        traffic[k] = v
json.dump(traffic, open(os.path.join(DST, "pmc_traffic.json"), "w"), indent=1)
print("profiles/pmc_traffic.json", {k: v["hbm_bytes_per_launch"] for k, v in traffic.items()})

# the round's plot: auto (sustained and under the reference's convention), both vendor libraries, the VALU rung, the peak line
files = [os.path.join(DST, f"r06_output_MMult_hip_{k}.m") for k in ("auto", "auto_ref_convention", "rocblas", "hipblaslt", "valu", "mfma")]
files = [f for f in files if os.path.exists(f)]
if files:
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "plot_sweep.py"), *files, "-o", os.path.join(DST, "r06_sweep.png")],
                       capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr.strip()[-300:])
