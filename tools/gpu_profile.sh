#!/bin/bash
# tools/gpu_profile.sh -- run ON THE GPU BOX (via gpurun) from the repo root.
#   PASSES="trace pmc1 pmc2 pmc3 pmc4"  (default: all)   KERNEL=mfma   TAG=prof
# trace: kernel trace + stats of the default bench command (what bench.py's
#        roofline object must agree with).
# pmcN : counter passes, each in its own run with --kernel-trace only (never mixed
#        with sys/hip traces) and at most what the hardware can collect at once
#        (SQ 8 / TCC 4 slots; FETCH_SIZE alone costs 3 TCC slots).
# Every rocprofv3 run is wrapped in `timeout`: a rejected counter set aborts the
# child but can leave rocprofv3 waiting forever.
# Results land in gpurun_out/$TAG/<pass>/ ; tools/summarize_profile.py turns them
# into the committed profiles/*.json.
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
TAG=${TAG:-prof}
OUT=$REPO/gpurun_out/$TAG
PASSES=${PASSES:-"trace pmc1 pmc2 pmc3 pmc4"}
KERNEL=${KERNEL:-mfma}
rm -rf "$OUT"; mkdir -p "$OUT"
BENCH="python $REPO/bench.py --no-extras --no-cpu-baseline --kernel $KERNEL ${BENCH_ARGS:-}"
cd /tmp
declare -A CTRS
CTRS[pmc1]="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
CTRS[pmc2]="SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
CTRS[pmc3]="FETCH_SIZE"
CTRS[pmc5]="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"   # VALU rung only (ask for it in PASSES)
CTRS[pmc4]="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
for p in $PASSES; do
  if [ "$p" = trace ]; then
    timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $BENCH > "$OUT/trace.log" 2>&1
    echo "trace rc=$?"
  else
    timeout ${PMC_TIMEOUT:-180} rocprofv3 --kernel-trace --pmc ${CTRS[$p]} --output-format csv -d "$OUT/$p" -o pmc -- $BENCH --steps 3 --warmup 1 > "$OUT/$p.log" 2>&1
    echo "$p rc=$? (${CTRS[$p]})"
  fi
done
du -sh "$OUT"
