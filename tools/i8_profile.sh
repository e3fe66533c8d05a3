#!/bin/bash
# tools/i8_profile.sh -- run ON THE GPU BOX: kernel trace + two counter passes of tools/i8_trace.py.
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/${TAG:-i8prof}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python $REPO/tools/i8_trace.py > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
export I8_REPS=4
timeout 240 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc1" -o pmc -- python $REPO/tools/i8_trace.py > "$OUT/pmc1.log" 2>&1
echo "pmc1 rc=$?"
timeout 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc2" -o pmc -- python $REPO/tools/i8_trace.py > "$OUT/pmc2.log" 2>&1
echo "pmc2 rc=$?"
python $REPO/tools/i8_summarize.py "$OUT"
