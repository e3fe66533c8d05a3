#!/usr/bin/env python3
"""Summarise gpurun_out/prof/* (written by tools/gpu_profile.sh) into the
committed profiles/ files: per-kernel average duration from the kernel trace
and per-dispatch PMC means for the dominant kernel."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

prof = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
match = sys.argv[2] if len(sys.argv) > 2 else "sgemm_mfma_kernel"
csv.field_size_limit(1 << 30)


def short(name):
    return name.split("(")[0].replace("void ", "")[:90]


out = {"kernel_match": match}
stats = glob.glob(os.path.join(prof, "trace", "*kernel_stats.csv"))
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    out["kernel_stats"] = [
        {"name": short(r["Name"]), "calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2),
         "min_us": round(float(r["MinNs"]) / 1e3, 2), "max_us": round(float(r["MaxNs"]) / 1e3, 2),
         "pct": float(r["Percentage"])} for r in rows[:8]]
counters = {}
for d in sorted(glob.glob(os.path.join(prof, "pmc*"))):
    f = glob.glob(os.path.join(d, "*counter_collection.csv"))
    if not f:
        continue
    acc = defaultdict(list)
    dur = []
    seen = set()
    for r in csv.DictReader(open(f[0])):
        if match not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    if acc:
        c = {k: sum(v) / len(v) for k, v in acc.items()}
        c["_dispatches"] = len(seen)
        c["_avg_us_under_pmc"] = round(sum(dur) / len(dur), 2)
        counters[os.path.basename(d)] = c
out["pmc_mean_per_dispatch"] = counters
print(json.dumps(out, indent=1))
