#!/usr/bin/env python3
"""pmc_by_kernel.py -- per-kernel-name means of the counters in rocprofv3 counter_collection CSVs (no GPU).

    python tools/pmc_by_kernel.py DIR [DIR ...] [--last N] [--group G]

Every DIR is searched for *counter_collection.csv.  Dispatches much shorter than a kernel name's longest (the one-tile
warm launches of mmh_create carry the same names) are dropped; of the rest the LAST N dispatches per name are
averaged (default 6: the timed repetitions of tools/pmc_launch.py).  --group G: variants that share a kernel name
(raster A/Bs) were launched G = warm + reps times each, one after the other: the name's dispatches are cut into
runs of G and every run is reported on its own (name#0, name#1, ...).  FETCH_SIZE / WRITE_SIZE (KiB) are also reported
in bytes, FETCH_SIZE doubled as MI355X_MICROARCH.md (HBM section) prescribes for 16-byte-per-lane reads on gfx950."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict, OrderedDict

csv.field_size_limit(1 << 30)


def main():
    args = sys.argv[1:]
    last, group = 6, 0
    for flag in ("--last", "--group"):
        if flag in args:
            i = args.index(flag)
            v = int(args[i + 1])
            del args[i:i + 2]
            if flag == "--last":
                last = v
            else:
                group = v
    dirs = args
    per = defaultdict(lambda: defaultdict(OrderedDict))   # name -> counter -> dispatch -> value
    dur = defaultdict(OrderedDict)
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                key = (f, r["Dispatch_Id"])
                per[name][r["Counter_Name"]][key] = per[name][r["Counter_Name"]].get(key, 0.0) + float(r["Counter_Value"])
                dur[name][key] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    out = {}
    todo = []
    for name, ctrs in per.items():
        if "sgemm" not in name and "igemm" not in name:
            continue
        longest = max(dur[name].values())
        if group:
            n_runs = max(len([k for k in vals if dur[name][k] >= 0.5 * longest]) // group for vals in ctrs.values())
            for g in range(max(1, n_runs)):
                todo.append((name + (f"#{g}" if n_runs > 1 else ""), name, g))
        else:
            todo.append((name, name, -1))
    for label, name, g in todo:
        ctrs = per[name]
        longest = max(dur[name].values())
        row = {}
        for cname, vals in ctrs.items():
            keep = [k for k in vals if dur[name][k] >= 0.5 * longest]
            keep = keep[g * group:(g + 1) * group][-last:] if g >= 0 else keep[-last:]
            if not keep:
                continue
            v = sum(vals[k] for k in keep) / len(keep)
            row[cname] = round(v, 1)
            row["_us_" + cname] = round(sum(dur[name][k] for k in keep) / len(keep), 2)
            row["_n_" + cname] = len(keep)
        if "FETCH_SIZE" in row:
            row["fetch_bytes"] = int(row["FETCH_SIZE"] * 1024 * 2)   # KiB, x2: the guide's gfx950 correction
        if "WRITE_SIZE" in row:
            row["write_bytes"] = int(row["WRITE_SIZE"] * 1024)
        if "TCC_HIT_sum" in row and "TCC_MISS_sum" in row:
            row["l2_hit"] = round(row["TCC_HIT_sum"] / max(1.0, row["TCC_HIT_sum"] + row["TCC_MISS_sum"]), 4)
        out[label[:118]] = row
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
