#!/bin/bash
# round 3, call 11 (after the container was re-created): the GPU suite on HEAD, the seven harness sweeps, N +- 1 with the rim
set -u
O=gpurun_out/r03z; mkdir -p $O
export TMPDIR=/tmp
PARTS="tests sweeps" bash tools/r03_final.sh
timeout 500 python tools/offgrid_sweep.py --set pm1 --variants auto,rocblas,hipblaslt --out $O/offgrid_pm1 > $O/offgrid_pm1.log 2>&1
tail -1 $O/offgrid_pm1.log | cut -c1-200
python - $O/offgrid_pm1.json <<'PY'
import json, sys
rows = json.load(open(sys.argv[1]))
by = {r["m"]: r for r in rows}
for n in range(1024, 4097, 128):
    a, b, c = by[n - 1], by[n], by[n + 1]
    print(n, "N-1 %.1f (%.2f)  N %.1f  N+1 %.1f (%.2f)  vendors at N+1: %.1f %.1f   %s" % (a["auto"], a["auto"] / b["auto"], b["auto"], c["auto"], c["auto"] / b["auto"], c["rocblas"], c["hipblaslt"], c["launched"][:40] + ("..rim" if "rim" in c["launched"] else "")))
PY
