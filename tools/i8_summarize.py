#!/usr/bin/env python3
"""Per int8 kernel and problem size: median duration from the kernel trace, and from the counter
passes the effective clock (GRBM_GUI_ACTIVE / 8 XCDs / duration) and the matrix-pipe busy fraction
(SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / cycles)."""
import csv, glob, json, os, re, sys
from collections import defaultdict
from statistics import median
csv.field_size_limit(1 << 30)
prof = sys.argv[1]
def short(name):
    m = re.search(r"(igemm_s8\w*|pack_bt_s8_kernel)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else None
out = {}
f = glob.glob(os.path.join(prof, "trace", "*kernel_trace.csv"))
if f:
    d = defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        k = short(r["Kernel_Name"])
        if k:
            d[(k, r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""))].append(
                (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for (k, g), v in sorted(d.items()):
        out.setdefault(k + " grid " + g, {})["median_us"] = round(median(v), 2)
        out[k + " grid " + g]["calls"] = len(v)
for p in ("pmc1", "pmc2"):
    f = glob.glob(os.path.join(prof, p, "*counter_collection.csv"))
    if not f:
        continue
    d = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = short(r["Kernel_Name"])
        if k:
            d[k + " grid " + r.get("Grid_Size_X", r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in d.items():
        for c, v in cs.items():
            out.setdefault(k, {})[c] = round(median(v), 1)
for k, v in out.items():
    if "GRBM_GUI_ACTIVE" in v and "median_us" in v:
        cyc = v["GRBM_GUI_ACTIVE"] / 8
        v["clock_ghz"] = round(cyc / v["median_us"] / 1e3, 3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v:
            v["mfma_busy_frac"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc, 4)
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(prof, "summary.json"), "w"), indent=1)
