#!/bin/bash
# round 3, call 23: the headline size on the 128x64 LDS-DMA tile (2048 tiles = four whole rounds of two workgroups per CU):
# sustained / cold beside the 64x64 and 256x256 tiles in alternating fresh processes, and its fabric traffic
set -u
O=gpurun_out/r03t; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2; do
  for kk in auto mfma_128x64_dma mfma_256x256; do
    timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-live-traffic --kernel $kk 2> /dev/null | \
      python -c "import json,sys; d=json.load(sys.stdin); c=d['cold']; print('$kk', 'sustained', d['value'], 'launch1_ms', c['launch_1_ms'], 'first20', c['reference_convention_20_launches_no_warmup_tflops'], 'launches2to21', c['launches_2_to_21_tflops'], 'within1pct_after', c['launches_until_within_1pct_of_sustained'], d['roofline']['kernel'][:40])" >> $O/cold_start3.txt
    sleep 2
  done
done
cat $O/cold_start3.txt
TAG=r03t/prof4096_128x64 KERNEL=mfma_128x64_dma PASSES="trace pmc1 pmc3 pmc4" bash tools/gpu_profile.sh > $O/prof4096_128x64.log 2>&1
python tools/summarize_profile.py $O/prof4096_128x64 "sgemm_mfma_dma_kernel" > $O/prof4096_128x64_summary.json 2>> $O/prof4096_128x64.log
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
python - $O/prof4096_128x64_summary.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); ks=d['kernel_stats'][0]; pm=d['pmc_mean_per_dispatch']
print(ks['name'][:60], ks['calls'], ks['avg_us'])
print('fetch MB %.0f write MB %.0f L2 hit %.3f'%(pm['pmc3']['FETCH_SIZE']*2/1024, pm['pmc4']['WRITE_SIZE']/1024, pm['pmc4']['TCC_HIT_sum']/(pm['pmc4']['TCC_HIT_sum']+pm['pmc4']['TCC_MISS_sum'])))
p1=pm['pmc1']; print('mfma busy %.3f'%(p1['SQ_VALU_MFMA_BUSY_CYCLES']/1024/(p1['GRBM_GUI_ACTIVE']/8)))
PY
