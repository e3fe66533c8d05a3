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
  ( time timeout 900 python -m pytest tests -m gpu -x -q -k "${K32_FILTER:-mfma32 or dma5}" ) > $OUT/pytest_k32.log 2>&1
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
if has abl; then        # timing-only ablations of the K2M loop (tools build)
  for n in ${ABL_SIZES:-4096 1024}; do
    timeout 300 python tools/ab_bench.py --n $n ${ABL_VARIANTS:-mfma_128x64_dma mfma32_128x64_dma 52 53 54 55 mfma_64x64_dma mfma32_64x64_dma 56 57 58 59 mfma32_64x128_dma} > $OUT/abl_$n.txt 2>&1
    cat $OUT/abl_$n.txt | grep -v amdgpu.ids
  done
fi
if has pmc; then        # counters (each group its own run) for a list of kernels at one size
  for kk in ${PMC_KERNELS:-mfma_128x64_dma mfma32_128x64_dma mfma32_64x128_dma}; do
    TAG=$TAG/pmc_$kk KERNEL=$kk PASSES="${PMC_PASSES:-pmc1 pmc2}" BENCH_ARGS="${PMC_BENCH_ARGS:-}" bash tools/gpu_profile.sh > $OUT/pmc_$kk.log 2>&1
    python tools/summarize_profile.py $OUT/pmc_$kk "${PMC_MATCH:-sgemm_}" > $OUT/pmc_$kk.json 2>> $OUT/pmc_$kk.log
    python - $OUT/pmc_$kk.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for p, c in d.get("pmc_mean_per_dispatch", {}).items():
    print(sys.argv[1].split("/")[-1], p, {k: (round(v, 1) if isinstance(v, float) else v) for k, v in c.items()})
PY
  done
  find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
fi
du -sh $OUT
