#!/bin/bash
# round 3, call 10: the rim (N = 1025, 2049, ...), the VALU rung with one LDS slice (three waves per SIMD)
set -u
O=gpurun_out/r03i; mkdir -p $O
export TMPDIR=/tmp
H=how-to-optimize-gemm_amd/harness
timeout 600 python -m pytest tests/test_gpu_round3.py -m gpu -q -x > $O/pytest_round3.txt 2>&1; tail -5 $O/pytest_round3.txt
timeout 300 python -m pytest tests -m gpu -q -k "valu or VALU or ladder or kernels_agree" > $O/pytest_valu.txt 2>&1; tail -2 $O/pytest_valu.txt
timeout 600 python tools/offgrid_sweep.py --set pm1 --variants auto,rocblas,hipblaslt --out $O/offgrid_pm1 > $O/offgrid_pm1.log 2>&1
tail -1 $O/offgrid_pm1.log | cut -c1-200
python - $O/offgrid_pm1.json <<'PY'
import json, sys
rows = json.load(open(sys.argv[1]))
by = {r["m"]: r for r in rows}
for n in range(1024, 4097, 128):
    a, b, c = by[n - 1], by[n], by[n + 1]
    print(n, "N-1 %.1f (%.2f)  N %.1f  N+1 %.1f (%.2f)  vendors at N+1: %.1f %.1f   %s" % (a["auto"], a["auto"] / b["auto"], b["auto"], c["auto"], c["auto"] / b["auto"], c["rocblas"], c["hipblaslt"], c["launched"][:40] + ("..rim" if "rim" in c["launched"] else "")))
PY
for nb in 1 2; do
  ( cd $H && MMH_VALU_NBUF=$nb KERNEL=valu REF=skip WARMUP_MS=50 TRIALS=3 timeout 300 ./test_MMult.x ) > $O/output_MMult_hip_valu_nbuf$nb.m 2> $O/valu$nb.err
done
paste <(grep -E "^[0-9]+ " $O/output_MMult_hip_valu_nbuf1.m | awk '{print $1, $2}') <(grep -E "^[0-9]+ " $O/output_MMult_hip_valu_nbuf2.m | awk '{print $2}')
