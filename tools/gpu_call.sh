#!/bin/bash
# gpu_call.sh -- the single-purpose gpurun calls of a round, one script (round 3 had 25 copies of this):
#   gpurun --timeout 900 -- 'STEPS="k32tests sweep" bash tools/gpu_call.sh'
# Runs ON THE GPU BOX from the repo root; every step writes under gpurun_out/<TAG>/ (TAG defaults to r04).
set -u
TAG=${TAG:-r04}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
STEPS=${STEPS:-"k32tests sweep"}
has() { [[ " $STEPS " == *" $1 "* ]]; }
if has k32tests; then   # the parity tests of the 32x32x2 LDS-DMA tiles only
  ( time timeout 900 python -m pytest tests -m gpu -x -q -k "mfma32" ) > $OUT/pytest_k32.log 2>&1
  tail -5 $OUT/pytest_k32.log
fi
if has tests; then
  ( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.log 2>&1
  tail -5 $OUT/pytest_gpu.log
fi
if has sweep; then      # every tile family forced, plain and stream-K, over the reference sweep
  timeout 900 python tools/tile_sweep.py --check ${SWEEP_ARGS:-} --out $OUT/tile_sweep${SWEEP_TAG:-} > $OUT/tile_sweep${SWEEP_TAG:-}.log 2>&1
  tail -30 $OUT/tile_sweep${SWEEP_TAG:-}.log | cut -c1-400
fi
du -sh $OUT
