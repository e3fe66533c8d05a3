#!/bin/bash
# round 3, call 22: counters for the sizes the sweep still shows below 0.92 of peak: 2304 (stream-K on 128x128 tiles, 1.27
# tiles per workgroup), 1152 (stream-K on 64x64 tiles), 1024 (one 64x64 tile per CU), and the VALU rung at 4096
set -u
O=gpurun_out/r03s; mkdir -p $O
export TMPDIR=/tmp
for n in 2304 1152 1024; do
  TAG=r03s/prof$n KERNEL=auto BENCH_ARGS="--n $n" PASSES="trace pmc1 pmc3 pmc4" bash tools/gpu_profile.sh > $O/prof$n.log 2>&1
done
TAG=r03s/prof4096_valu KERNEL=valu PASSES="trace pmc1 pmc5" bash tools/gpu_profile.sh > $O/prof4096_valu.log 2>&1
python tools/summarize_profile.py $O/prof2304 "sgemm_dma_streamk_kernel" > $O/prof2304_summary.json 2>> $O/prof2304.log
python tools/summarize_profile.py $O/prof1152 "sgemm_dma_streamk_kernel" > $O/prof1152_summary.json 2>> $O/prof1152.log
python tools/summarize_profile.py $O/prof1024 "sgemm_mfma_dma_kernel" > $O/prof1024_summary.json 2>> $O/prof1024.log
python tools/summarize_profile.py $O/prof4096_valu "sgemm_valu_kernel" > $O/prof4096_valu_summary.json 2>> $O/prof4096_valu.log
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
head -c 600 $O/prof2304_summary.json; echo; head -c 400 $O/prof4096_valu_summary.json; du -sh $O
