#!/bin/bash
# which box is this, and how do the three headline candidates run on it?  (tools: box-to-box variance of the headline)
set -u
mkdir -p gpurun_out/boxes
{
echo "box $(hostname) $(date +%H:%M:%S) $(rocm-smi --showserial 2>/dev/null | grep -i serial | head -1 | tr -s ' ')"
for kk in auto mfma_64x64_dma mfma_256x256; do
  timeout 200 python bench.py --steps 20 --warmup 5 --ramp 0 --no-extras --no-cpu-baseline --no-live-traffic --kernel $kk 2> /dev/null | \
    python -c "import json,sys; d=json.load(sys.stdin); print('  $kk', d['value'], d['roofline']['frac'])"
done
} | tee -a gpurun_out/boxes/probe_$(date +%s).txt
